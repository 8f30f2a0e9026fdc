// device_agg.h -- accumulator device code shared by the AOT aggregation kernels (agg.hip) and the JIT-fused
// project+accumulate kernels (jit.cpp embeds this file verbatim).  Self-contained.
#pragma once

#define TG_LIMBS 68          // 32-bit limbs (held in int64 words) covering 2^-1074 .. 2^2101
#define TG_AGG_BLOCK 256
#define TG_MAX_AGGS 16

// aggregate function codes = tgpu_agg_function
#define TG_AGG_COUNT_ALL 1
#define TG_AGG_COUNT_COLUMN 2
#define TG_AGG_SUM_BIGINT 3
#define TG_AGG_SUM_DOUBLE 4
#define TG_AGG_AVG_BIGINT 5
#define TG_AGG_AVG_DOUBLE 6

// per-aggregate state arrays in HBM, indexed by group id
struct TgAggState {
    int function;
    int pad;
    long long *counts;            // int64[g]
    long long *limbs;             // int64[g][TG_LIMBS]   exact accumulator of double sums
    unsigned int *special;        // uint32[g]            NaN / +inf / -inf seen
    unsigned long long *i128;     // uint64[g][2]         bigint sums
    double *dsum;                 // double[g]            ORDERED mode: plain running sum (limbs / special are null then)
};

// exact, order-independent accumulation of one double into the limb array of its group
__device__ inline void tg_kulisch_add(long long *limbs, unsigned int *special, double v)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned int e = (unsigned int)((bits >> 52) & 0x7ff);
    unsigned long long m = bits & 0xfffffffffffffULL;
    const bool neg = (bits >> 63) != 0;
    if (e == 0x7ff) {
        atomicOr(special, m ? 1u : (neg ? 4u : 2u));
        return;
    }
    int p = 0;
    if (e) { m |= 1ULL << 52; p = (int)e - 1; }
    if (m == 0) return;
    const int j = p >> 5, s = p & 31;
    const unsigned long long lo64 = m << s;                      // (m << s) as three 32-bit limbs
    const unsigned long long hi64 = s ? (m >> (64 - s)) : 0ULL;
    long long l0 = (long long)(lo64 & 0xffffffffULL), l1 = (long long)(lo64 >> 32), l2 = (long long)hi64;
    if (neg) { l0 = -l0; l1 = -l1; l2 = -l2; }
    if (l0) atomicAdd((unsigned long long *)&limbs[j], (unsigned long long)l0);
    if (l1) atomicAdd((unsigned long long *)&limbs[j + 1], (unsigned long long)l1);
    if (l2) atomicAdd((unsigned long long *)&limbs[j + 2], (unsigned long long)l2);
}

__device__ inline void tg_i128_add(unsigned long long *acc, long long v)
{
    const unsigned long long uv = (unsigned long long)v;
    const unsigned long long old = atomicAdd(&acc[0], uv);
    const unsigned long long carry = (old + uv) < old ? 1ULL : 0ULL;
    const unsigned long long hi_add = (v < 0 ? ~0ULL : 0ULL) + carry;
    if (hi_add) atomicAdd(&acc[1], hi_add);
}

// double-double add (hi, lo) += (h2, l2)
__device__ inline void tg_dd_add(double &hi, double &lo, double h2, double l2)
{
    const double s = hi + h2;
    const double bb = s - hi;
    double e = (hi - (s - bb)) + (h2 - bb);
    e += lo + l2;
    hi = s + e;
    lo = e - (hi - s);
}

// ---- low-cardinality path: lane-private accumulators in LDS ---------------------------------------------------------
// layout per group: hi[n_wide][256] doubles, lo[n_wide][256] doubles, cnt[n_cnt][256] uint32
struct TgLowCardPlan {
    int n_aggs;
    int n_wide;                       // distinct 16-byte states (double sums, bigint sums); aggregates over the same input share one
    int wide_slot[TG_MAX_AGGS];       // aggregate -> wide state, or -1
    int n_cnt;                        // count slots (cnt[n_cnt][256])
    int cnt_slot[TG_MAX_AGGS];        // aggregate -> count slot
    int count_from_rows[TG_MAX_AGGS]; // 1: this aggregate's count is the group's row count (slot rows_slot), its own slot is not maintained
    int rows_slot;                    // count slot holding the number of rows of the group, or -1
    int per_group_bytes;
    int n_groups;
};

__device__ inline double *tg_lc_hi(unsigned char *lds, const TgLowCardPlan &p, int g) { return (double *)(lds + (size_t)g * p.per_group_bytes); }
__device__ inline double *tg_lc_lo(unsigned char *lds, const TgLowCardPlan &p, int g)
{
    return (double *)(lds + (size_t)g * p.per_group_bytes + p.n_wide * TG_AGG_BLOCK * 8);
}
__device__ inline unsigned int *tg_lc_cnt(unsigned char *lds, const TgLowCardPlan &p, int g)
{
    return (unsigned int *)(lds + (size_t)g * p.per_group_bytes + p.n_wide * 2 * TG_AGG_BLOCK * 8);
}

__device__ inline void tg_lc_zero(unsigned char *lds, const TgLowCardPlan &p)
{
    const int total_words = p.n_groups * p.per_group_bytes / 4;
    for (int i = threadIdx.x; i < total_words; i += TG_AGG_BLOCK) ((unsigned int *)lds)[i] = 0u;
    __syncthreads();
}

// one double into the lane's (hi, lo) slot: two-sum keeps the rounding error in lo
__device__ inline void tg_lc_add_double(double *hi_base, double *lo_base, int w, double v)
{
    const int i = w * TG_AGG_BLOCK + threadIdx.x;
    const double hi = hi_base[i];
    const double s = hi + v;
    const double bb = s - hi;
    const double err = (hi - (s - bb)) + (v - bb);
    hi_base[i] = s;
    lo_base[i] += err;
}

__device__ inline void tg_lc_add_bigint(double *hi_base, double *lo_base, int w, long long v)
{
    unsigned long long *lo64 = (unsigned long long *)&hi_base[w * TG_AGG_BLOCK + threadIdx.x];
    long long *hi64 = (long long *)&lo_base[w * TG_AGG_BLOCK + threadIdx.x];
    const unsigned long long old = *lo64, nw = old + (unsigned long long)v;
    *lo64 = nw;
    *hi64 += (v < 0 ? -1 : 0) + (nw < old ? 1 : 0);
}

__device__ inline void tg_flag_special(unsigned int *special, double v)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    atomicOr(special, (bits & 0xfffffffffffffULL) ? 1u : ((bits >> 63) ? 4u : 2u));
}

// end of block: fold the 256 lane partials of each (group, aggregate) and add them EXACTLY into the global state
__device__ inline void tg_lc_fold(unsigned char *lds, const TgLowCardPlan &p, const TgAggState *st)
{
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int g = 0; g < p.n_groups; g++) {
        double *hi_base = tg_lc_hi(lds, p, g), *lo_base = tg_lc_lo(lds, p, g);
        unsigned int *cnt_base = tg_lc_cnt(lds, p, g);
        for (int k = 0; k < p.n_aggs; k++) {
            const TgAggState &a = st[k];
            unsigned long long c = cnt_base[(p.count_from_rows[k] ? p.rows_slot : p.cnt_slot[k]) * TG_AGG_BLOCK + threadIdx.x];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
            const bool any = __shfl(c, 0, 64) != 0;
            if (!any) continue;
            if (lane == 0) atomicAdd((unsigned long long *)&a.counts[g], c);
            const int w = p.wide_slot[k];
            if (w < 0) continue;
            if (a.function == TG_AGG_SUM_BIGINT) {
                const unsigned long long lo64 = *(unsigned long long *)&hi_base[w * TG_AGG_BLOCK + threadIdx.x];
                const long long hi64 = *(long long *)&lo_base[w * TG_AGG_BLOCK + threadIdx.x];
                if (lo64 || hi64) {
                    const unsigned long long old = atomicAdd(&a.i128[g * 2], lo64);
                    const unsigned long long carry = (old + lo64) < old ? 1ULL : 0ULL;
                    const unsigned long long add_hi = (unsigned long long)hi64 + carry;
                    if (add_hi) atomicAdd(&a.i128[g * 2 + 1], add_hi);
                }
                continue;
            }
            double hi = hi_base[w * TG_AGG_BLOCK + threadIdx.x], lo = lo_base[w * TG_AGG_BLOCK + threadIdx.x];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const double h2 = __shfl_down(hi, d, 64), l2 = __shfl_down(lo, d, 64);
                tg_dd_add(hi, lo, h2, l2);
            }
            if (lane == 0) {
                tg_kulisch_add(&a.limbs[(size_t)g * TG_LIMBS], &a.special[g], hi);
                tg_kulisch_add(&a.limbs[(size_t)g * TG_LIMBS], &a.special[g], lo);
            }
        }
    }
}
