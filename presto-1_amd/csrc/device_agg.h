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
#define TG_AGG_MIN_BIGINT 7
#define TG_AGG_MAX_BIGINT 8
#define TG_AGG_MIN_DOUBLE 9
#define TG_AGG_MAX_DOUBLE 10

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

// min / max over BIGINT and DOUBLE (AbstractMinMaxAggregationFunction.java:274-306: keep the value the comparison prefers).  The state word --
// word 0 of the aggregate's two i128 words -- holds the running extreme in an unsigned code that preserves (max) or reverses (min) the
// order, so one unsigned atomicMax serves both and a zero word is the identity (no value yet; the count says whether there is one).
// Order independent and exact: every path (lane-private, global, row-order, one-pass re-runs) may apply a row any number of times.
__device__ inline bool tg_is_minmax(int function) { return function >= TG_AGG_MIN_BIGINT && function <= TG_AGG_MAX_DOUBLE; }
// raw = the value's 8 bytes (BIGINT or DOUBLE).  DOUBLE: the usual total-order key of the bits (negatives reversed below the positives: -0.0 <
// +0.0, as Double.compare has it), with NaN where the reference's comparison puts it -- min: MinMaxCompare.min over Double.compare
// (DoubleType.java:194-198), NaN above +inf, so NaN maps to the top key = code 0: it wins only when nothing else comes; max:
// MinMaxCompare.maxDouble (`left > right || isNaN(right)`), a NaN state gives way to any value and never replaces one: key 0 = code 0.
// Either way "only NaNs" is code 0 with a count, decoded as NaN (the canonical one: a NaN's payload is not kept).
__device__ inline unsigned long long tg_minmax_encode(int function, unsigned long long raw)
{
    const unsigned long long sign = 0x8000000000000000ULL;
    if (function <= TG_AGG_MAX_BIGINT) {
        const unsigned long long u = raw ^ sign;
        return function == TG_AGG_MIN_BIGINT ? ~u : u;
    }
    if ((raw & ~sign) > 0x7ff0000000000000ULL) return 0ULL;                     // NaN
    const unsigned long long key = (raw & sign) ? ~raw : (raw | sign);
    return function == TG_AGG_MIN_DOUBLE ? ~key : key;
}
__device__ inline unsigned long long tg_minmax_decode(int function, unsigned long long code)
{
    const unsigned long long sign = 0x8000000000000000ULL;
    if (function <= TG_AGG_MAX_BIGINT) return (function == TG_AGG_MIN_BIGINT ? ~code : code) ^ sign;
    if (code == 0ULL) return 0x7ff8000000000000ULL;                             // NaN (or no value: the count decides)
    const unsigned long long key = function == TG_AGG_MIN_DOUBLE ? ~code : code;
    return (key & sign) ? (key & ~sign) : ~key;
}
// (the plain read first: after a group's first few rows almost no row improves the extreme, and a stale read can only be too small)
__device__ inline void tg_minmax_update(unsigned long long *word, unsigned long long code)
{
    if (code > *(volatile unsigned long long *)word) atomicMax(word, code);
}

// acc (128-bit, two words) += v, v a partial sum of many rows
__device__ inline void tg_i128_add_wide(unsigned long long *acc, __int128 v)
{
    const unsigned long long lo = (unsigned long long)v, hi = (unsigned long long)((unsigned __int128)v >> 64);
    const unsigned long long old = atomicAdd(&acc[0], lo);
    const unsigned long long hi_add = hi + ((old + lo) < old ? 1ULL : 0ULL);
    if (hi_add) atomicAdd(&acc[1], hi_add);
}

// ---- strict row-order sums (ORDERED mode with few groups: jit.cpp fa_ordered_chain, agg.hip agg_ordered_chain_kernel) -----------------------
// One workgroup of TG_ORD_WAVES waves per group: waves 1.. produce the DOUBLE addends of TG_ORD_TILE rows into one of two LDS tiles
// ([chain][TG_ORD_STRIDE] doubles each), wave 0 adds the tile before -- lane d = the d-th DOUBLE sum of the operator.
#define TG_ORD_WAVES 16
#define TG_ORD_TILE ((TG_ORD_WAVES - 1) * 64)
#define TG_ORD_STRIDE (TG_ORD_TILE + 2)      // even: 16-byte LDS reads; neighbouring lanes' chains lie 4 banks apart
#define TG_ORD_MAX_DOUBLES 8
// os + in[0] + in[1] + ... + in[cnt - 1], strictly left to right (`in`: 16-byte aligned LDS).  16 values per batch (8 x 16-byte reads); the
// reads of the next batch are issued BETWEEN this batch's additions (one read after every second addition: a dependent v_add_f64 waits
// ~12 cycles for its input, tools/exp_dep_add.hip, and a read fits in that shadow); two batches per trip so that no registers are copied
__device__ inline double tg_chain_add_tile(const double *in, int cnt, double os)
{
#define TG_ORD_ADDS(a) _Pragma("unroll") for (int u = 0; u < 8; u++) { os += a[u].x; os += a[u].y; }
#define TG_ORD_MIX() _Pragma("unroll") for (int u = 0; u < 8; u++) { __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    int j = 0;
    if (cnt >= 16) {
        const double2 *in2 = (const double2 *)in;
        double2 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = in2[u];
        for (; j + 48 <= cnt; j += 32) {
#pragma unroll
            for (int u = 0; u < 8; u++) b[u] = in2[(j >> 1) + 8 + u];
            TG_ORD_ADDS(a)
            TG_ORD_MIX()
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = in2[(j >> 1) + 16 + u];
            TG_ORD_ADDS(b)
            TG_ORD_MIX()
        }
        TG_ORD_ADDS(a)
        j += 16;
        for (; j + 16 <= cnt; j += 16) {
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = in2[(j >> 1) + u];
            TG_ORD_ADDS(a)
        }
    }
    for (; j < cnt; j++) os += in[j];
    return os;
#undef TG_ORD_ADDS
#undef TG_ORD_MIX
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

// Folded partials of the low-cardinality launches: workgroup b owns slot row b -- `stride` (group, aggregate) items of 3 words: count,
// and the (hi, lo) double-double sum or the (low, high) words of a 128-bit bigint sum -- and ADDS each launch's block total to it
// (plain read-modify-write: the launches of an operator are ordered on the stream and a row has one owner per launch).  Nothing
// crosses workgroups per page; tg_fold_flush adds the rows EXACTLY into the global state when the state is read (evaluate).
// (Round 1 had every wave add its partials to the global state with atomics at the end of each launch: ~7 atomics per wave and
// (group, aggregate) onto the same few cache lines = ~100 us per launch whatever the page size; a last-workgroup reduction inside
// the launch was no better -- device-scope fences write back / invalidate the XCD's L2, remote partials arrive one latency at a
// time, and the cold reduction code is fetched at instruction-cache-miss speed.)
struct TgFoldScratch {
    unsigned long long *partials;   // [workgroups][stride][3]
    int stride;                     // items per row = group capacity x n_aggs
    // One-pass launches (filter + group lookup + accumulate in ONE kernel, no read-back in front of the next page): a page's block totals
    // are PENDING here, same shape as `partials`, until the page is known to be clean (no row met a group the kernel did not know); the
    // next one-pass launch of the operator -- or the host, for the last page -- adds them to `partials`, or drops them.
    unsigned long long *pending;
};

// a += b for a (hi, lo) double-double pair, or for the (low, high) words of a 128-bit integer
__device__ inline void tg_fold_pair(bool bigint, unsigned long long &a0, unsigned long long &a1, unsigned long long b0, unsigned long long b1)
{
    if (bigint) {
        const unsigned long long s = a0 + b0;
        a1 += b1 + (s < a0 ? 1ULL : 0ULL);
        a0 = s;
    }
    else {
        double hi = __longlong_as_double((long long)a0), lo = __longlong_as_double((long long)a1);
        tg_dd_add(hi, lo, __longlong_as_double((long long)b0), __longlong_as_double((long long)b1));
        a0 = (unsigned long long)__double_as_longlong(hi);
        a1 = (unsigned long long)__double_as_longlong(lo);
    }
}

__device__ inline void tg_fold_wave(bool bigint, unsigned long long &c, unsigned long long &a0, unsigned long long &a1)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        c += __shfl_down(c, d, 64);
        const unsigned long long b0 = __shfl_down(a0, d, 64), b1 = __shfl_down(a1, d, 64);
        tg_fold_pair(bigint, a0, a1, b0, b1);
    }
}

// end of block: the 256 lane partials of each (group, aggregate) item are folded by 8 threads -- each sums 32 of them from LDS, a
// rotated start keeps the wave's reads on distinct banks, then three shuffle rounds fold the 8 sums -- 32 items per pass with every
// lane busy (one wave per item with six full-width shuffle rounds was issue bound: ~2600 cycles per item, 10 us for TPCH Q1's 32
// items).  The item's total is added to the block's slot: plain read-modify-write, no atomics, no cross-workgroup traffic.
template <bool OVERWRITE>
__device__ inline void tg_lc_fold_to(unsigned char *lds, const TgLowCardPlan &p, const TgAggState *st, unsigned long long *mine)
{
    __syncthreads();
    const int t = threadIdx.x, chunk = t & 7;
    const int items = p.n_groups * p.n_aggs;
    for (int base = 0; base < items; base += TG_AGG_BLOCK / 8) {
        int it = base + (t >> 3);
        const bool live = it < items;
        it = live ? it : 0;
        const int g = it / p.n_aggs, k = it - g * p.n_aggs;
        const unsigned int *cnt = tg_lc_cnt(lds, p, g) + (p.count_from_rows[k] ? p.rows_slot : p.cnt_slot[k]) * TG_AGG_BLOCK + chunk * 32;
        const int w = p.wide_slot[k];
        const bool bigint = st[k].function == TG_AGG_SUM_BIGINT;
        const unsigned long long *hi = (const unsigned long long *)tg_lc_hi(lds, p, g) + (w < 0 ? 0 : w) * TG_AGG_BLOCK + chunk * 32;
        const unsigned long long *lo = (const unsigned long long *)tg_lc_lo(lds, p, g) + (w < 0 ? 0 : w) * TG_AGG_BLOCK + chunk * 32;
        unsigned long long c = 0, a0 = 0, a1 = 0;
        for (int i = 0; i < 32; i++) {
            const int e = (i + t) & 31;
            c += cnt[e];
            if (w >= 0) {
                if (i == 0) { a0 = hi[e]; a1 = lo[e]; }
                else tg_fold_pair(bigint, a0, a1, hi[e], lo[e]);
            }
        }
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1) {   // the 8 threads of an item are neighbours
            c += __shfl_down(c, d, 64);
            const unsigned long long b0 = __shfl_down(a0, d, 64), b1 = __shfl_down(a1, d, 64);
            tg_fold_pair(bigint, a0, a1, b0, b1);
        }
#ifdef FA_DEBUG_SKIP
        if (FA_DEBUG_SKIP & 8) continue;
#endif
        if (OVERWRITE) {
            if (chunk == 0 && live) {   // the slot takes this launch's totals whatever it held (a zero count marks "nothing")
                unsigned long long *slot = mine + (size_t)it * 3;
                slot[0] = c;
                slot[1] = w >= 0 ? a0 : 0ULL;
                slot[2] = w >= 0 ? a1 : 0ULL;
            }
        }
        else if (chunk == 0 && live && c != 0) {
            unsigned long long *slot = mine + (size_t)it * 3;
            slot[0] += c;
            if (w >= 0) {
                unsigned long long s0 = slot[1], s1 = slot[2];
                tg_fold_pair(bigint, s0, s1, a0, a1);
                slot[1] = s0;
                slot[2] = s1;
            }
        }
    }
}

__device__ inline void tg_lc_fold(unsigned char *lds, const TgLowCardPlan &p, const TgAggState *st, TgFoldScratch fs)
{
    tg_lc_fold_to<false>(lds, p, st, fs.partials + (size_t)blockIdx.x * fs.stride * 3);
}

// a workgroup row of pending totals joins the same row of the folded partials (the page they belong to turned out clean); the pending
// row is cleared when `clear` is set (host-side commit of the last page; a one-pass launch overwrites its row anyway)
__device__ inline void tg_commit_pending(TgFoldScratch fs, int n_aggs, const TgAggState *st, bool clear)
{
    unsigned long long *mainr = fs.partials + (size_t)blockIdx.x * fs.stride * 3, *pend = fs.pending + (size_t)blockIdx.x * fs.stride * 3;
    for (int it = threadIdx.x; it < fs.stride; it += TG_AGG_BLOCK) {
        unsigned long long *src = pend + (size_t)it * 3, *dst = mainr + (size_t)it * 3;
        const unsigned long long c = src[0];
        if (c != 0) {
            const bool bigint = st[it % n_aggs].function == TG_AGG_SUM_BIGINT;
            dst[0] += c;
            unsigned long long s0 = dst[1], s1 = dst[2];
            tg_fold_pair(bigint, s0, s1, src[1], src[2]);
            dst[1] = s0;
            dst[2] = s1;
        }
        if (clear) src[0] = src[1] = src[2] = 0;
    }
}

// one wave per item: folds the item's slot of every workgroup row (in row order: deterministic), adds the total exactly into the
// global state and clears the slots
__device__ inline void tg_fold_flush(TgFoldScratch fs, int rows, int n_aggs, const TgAggState *st)
{
    const int lane = threadIdx.x & 63;
    const int it = blockIdx.x * (TG_AGG_BLOCK / 64) + (threadIdx.x >> 6);
    if (it >= fs.stride) return;
    const int g = it / n_aggs, k = it - g * n_aggs;
    const TgAggState &a = st[k];
    const bool bigint = a.function == TG_AGG_SUM_BIGINT;
    unsigned long long c = 0, a0 = 0, a1 = 0;
    for (int b = lane; b < rows; b += 64) {
        unsigned long long *slot = fs.partials + ((size_t)b * fs.stride + it) * 3;
        const unsigned long long v = slot[0];
        if (v == 0) continue;
        c += v;
        tg_fold_pair(bigint, a0, a1, slot[1], slot[2]);
        slot[0] = slot[1] = slot[2] = 0;
    }
    tg_fold_wave(bigint, c, a0, a1);
    if (lane != 0 || c == 0) return;
    atomicAdd((unsigned long long *)&a.counts[g], c);
    if (a.function == TG_AGG_COUNT_ALL || a.function == TG_AGG_COUNT_COLUMN || tg_is_minmax(a.function)) return;   // (min / max: the rows went to the state word)
    if (bigint) {
        if (a0 || a1) {
            const unsigned long long old = atomicAdd(&a.i128[g * 2], a0);
            const unsigned long long add_hi = a1 + ((old + a0) < old ? 1ULL : 0ULL);
            if (add_hi) atomicAdd(&a.i128[g * 2 + 1], add_hi);
        }
        return;
    }
    tg_kulisch_add(&a.limbs[(size_t)g * TG_LIMBS], &a.special[g], __longlong_as_double((long long)a0));
    tg_kulisch_add(&a.limbs[(size_t)g * TG_LIMBS], &a.special[g], __longlong_as_double((long long)a1));
}
