// agg.hip -- grouped accumulators (K6) for count / sum / avg.
#include "agg.h"
#include "kernels.h"
#include "device_agg.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>

namespace tgpu {

namespace {

constexpr int kBlock = 256;

struct AggView {
    int32_t function;
    int32_t pad;
    const void *input;           // values of the input channel (or nullptr for count(*))
    const uint8_t *input_nulls;
    const uint8_t *mask;         // BOOLEAN mask channel values
    const uint8_t *mask_nulls;
    const void *input2;          // FINAL: the sum channel (input = the count channel)
    const uint8_t *input2_nulls;
    long long *counts;
    long long *limbs;
    unsigned int *special;
    unsigned long long *i128;
    double *dsum;                // ORDERED mode: plain running sum per group
};

struct AggArgs {
    int32_t n_aggs;
    int32_t pad;
    AggView a[kMaxAggs];
};

#define kulisch_add tg_kulisch_add
#define i128_add tg_i128_add

template <bool INTERMEDIATE>
__global__ void __launch_bounds__(kBlock) agg_accumulate_kernel(AggArgs args, const int32_t *__restrict__ gids, int64_t n)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        const int64_t g = gids ? gids[r] : 0;
        if (g < 0) continue;  // row excluded by a filter fused in front of the table
        for (int k = 0; k < args.n_aggs; k++) {
            const AggView &a = args.a[k];
            if (INTERMEDIATE) {
                // combine(): DoubleSumAggregation.java:48-52, AverageAggregations.java:63-67, CountAggregation.java:46-49
                const long long cnt = ((const long long *)a.input)[r];
                if (a.function == TGPU_AGG_COUNT_ALL || a.function == TGPU_AGG_COUNT_COLUMN) {
                    if (cnt) atomicAdd((unsigned long long *)&a.counts[g], (unsigned long long)cnt);
                    continue;
                }
                if (cnt == 0) continue;  // empty partial state
                atomicAdd((unsigned long long *)&a.counts[g], (unsigned long long)cnt);
                if (a.function == TGPU_AGG_SUM_BIGINT) i128_add(&a.i128[g * 2], ((const long long *)a.input2)[r]);
                else if (tg_is_minmax(a.function)) tg_minmax_update(&a.i128[g * 2], tg_minmax_encode(a.function, ((const unsigned long long *)a.input2)[r]));
                else kulisch_add(&a.limbs[g * kLimbs], &a.special[g], ((const double *)a.input2)[r]);
                continue;
            }
            if (a.mask && ((a.mask_nulls && a.mask_nulls[r]) || !a.mask[r])) continue;
            if (a.function == TGPU_AGG_COUNT_ALL) {
                atomicAdd((unsigned long long *)&a.counts[g], 1ULL);
                continue;
            }
            if (a.input_nulls && a.input_nulls[r]) continue;
            atomicAdd((unsigned long long *)&a.counts[g], 1ULL);
            switch (a.function) {
            case TGPU_AGG_SUM_BIGINT: i128_add(&a.i128[g * 2], ((const long long *)a.input)[r]); break;
            case TGPU_AGG_MIN_BIGINT:
            case TGPU_AGG_MAX_BIGINT:
            case TGPU_AGG_MIN_DOUBLE:
            case TGPU_AGG_MAX_DOUBLE: tg_minmax_update(&a.i128[g * 2], tg_minmax_encode(a.function, ((const unsigned long long *)a.input)[r])); break;
            case TGPU_AGG_SUM_DOUBLE:
            case TGPU_AGG_AVG_DOUBLE: kulisch_add(&a.limbs[g * kLimbs], &a.special[g], ((const double *)a.input)[r]); break;
            case TGPU_AGG_AVG_BIGINT: kulisch_add(&a.limbs[g * kLimbs], &a.special[g], (double)((const long long *)a.input)[r]); break;
            default: break;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// ORDERED mode (many groups): keys[i] = group id + 1 of the i-th row in (group, row) order (0 = row excluded by a filter),
// rows[i] = its row number.  The lane that owns the first row of a group walks the group's rows in order.
// ---------------------------------------------------------------------------------------------------------------------
// *unsorted is set when some key is smaller than its predecessor: otherwise the pairs are already in (group, row) order -- the
// usual case behind a join whose probe side is clustered by the group key (first-seen ids then ascend with the rows) -- and the
// sort is skipped
__global__ void __launch_bounds__(kBlock) ordered_keys_kernel(const int32_t *__restrict__ gids, int64_t n, unsigned int *__restrict__ keys, int *__restrict__ rows,
                                                               unsigned int *__restrict__ unsorted)
{
    bool bad = false;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        const int32_t g = gids[r];
        keys[r] = (unsigned int)(g + 1);
        rows[r] = (int)r;
        bad = bad || (r > 0 && gids[r - 1] > g);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *unsorted = 1u;   // idempotent plain store
}

// {first, end} of every group's stretch of the sorted keys (the words start at 0)
__global__ void __launch_bounds__(kBlock) ordered_stretches_kernel(const unsigned int *__restrict__ keys, int64_t n, int *__restrict__ stretches)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const unsigned int k = keys[i];
        if (k == 0) continue;
        if (i == 0 || keys[i - 1] != k) stretches[(size_t)(k - 1) * 2] = (int)i;
        if (i == n - 1 || keys[i + 1] != k) stretches[(size_t)(k - 1) * 2 + 1] = (int)(i + 1);
    }
}

// handoff > 0: a lane walks at most that many rows of its group and leaves the rest ({group, next index} appended to list) to
// agg_ordered_chain_kernel -- a group far longer than the others would otherwise be the whole launch
template <bool INTERMEDIATE>
__global__ void __launch_bounds__(kBlock) agg_ordered_kernel(AggArgs args, const unsigned int *__restrict__ keys, const int *__restrict__ rows, int64_t n,
                                                              unsigned int *error, int64_t handoff, long long *list, unsigned int *list_count)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const unsigned int key = keys[i];
        if (key == 0 || (i > 0 && keys[i - 1] == key)) continue;
        const int64_t g = (int64_t)key - 1;
        const int64_t stop = handoff > 0 ? (i + handoff < n ? i + handoff : n) : n;   // (the walk ends there at the latest)
        if (handoff > 0 && stop < n && keys[stop] == key) {
            const unsigned int at = atomicAdd(list_count, 1u);
            list[(size_t)at * 2] = g;
            list[(size_t)at * 2 + 1] = stop;
        }
#pragma unroll
        for (int k = 0; k < kMaxAggs; k++) {   // constant indices into the by-value argument block (no scratch copy)
            if (k >= args.n_aggs) break;
            const AggView &a = args.a[k];
            long long cnt = 0;
            double s = a.dsum ? a.dsum[g] : 0.0;
            __int128 big = 0;
            unsigned long long best = 0;   // min / max: the best code among the rows walked (0 = none)
            for (int64_t j = i; j < stop && keys[j] == key; j++) {
                const int64_t r = rows[j];
                if (INTERMEDIATE) {
                    // combine(): DoubleSumAggregation.java:48-52, AverageAggregations.java:63-67, CountAggregation.java:46-49
                    const long long c = ((const long long *)a.input)[r];
                    if (c == 0) continue;   // empty partial state
                    cnt += c;
                    if (a.function == TGPU_AGG_SUM_BIGINT) big += ((const long long *)a.input2)[r];
                    else if (tg_is_minmax(a.function)) {
                        const unsigned long long c_ = tg_minmax_encode(a.function, ((const unsigned long long *)a.input2)[r]);
                        best = c_ > best ? c_ : best;
                    }
                    else if (a.function != TGPU_AGG_COUNT_ALL && a.function != TGPU_AGG_COUNT_COLUMN) s += ((const double *)a.input2)[r];
                    continue;
                }
                if (a.mask && ((a.mask_nulls && a.mask_nulls[r]) || !a.mask[r])) continue;
                if (a.function != TGPU_AGG_COUNT_ALL && a.input_nulls && a.input_nulls[r]) continue;
                cnt++;
                switch (a.function) {
                case TGPU_AGG_SUM_BIGINT: big += ((const long long *)a.input)[r]; break;
                case TGPU_AGG_MIN_BIGINT:
                case TGPU_AGG_MAX_BIGINT:
                case TGPU_AGG_MIN_DOUBLE:
                case TGPU_AGG_MAX_DOUBLE: {
                    const unsigned long long c_ = tg_minmax_encode(a.function, ((const unsigned long long *)a.input)[r]);
                    best = c_ > best ? c_ : best;
                    break;
                }
                case TGPU_AGG_SUM_DOUBLE:
                case TGPU_AGG_AVG_DOUBLE: s += ((const double *)a.input)[r]; break;
                case TGPU_AGG_AVG_BIGINT: s += (double)((const long long *)a.input)[r]; break;
                default: break;
                }
            }
            if (cnt) a.counts[g] += cnt;
            if (a.dsum) a.dsum[g] = s;
            if (best) tg_minmax_update(&a.i128[g * 2], best);
            if (a.i128 && big != 0) {
                const unsigned __int128 cur = ((unsigned __int128)a.i128[g * 2 + 1] << 64) | a.i128[g * 2];
                const unsigned __int128 nxt = cur + (unsigned __int128)big;
                a.i128[g * 2] = (unsigned long long)nxt;
                a.i128[g * 2 + 1] = (unsigned long long)(nxt >> 64);
            }
        }
    }
    (void)error;
}

// ORDERED mode with few groups (many rows per group): one workgroup per group, see device_agg.h "strict row-order sums".  Waves 1..15 take
// the group's rows 960 at a time: counts and BIGINT sums are reduced per wave and added to LDS words (any order gives the same bits), the
// DOUBLE addends go to the LDS tile -- a row an aggregate skips travels as -0.0, the identity of IEEE addition; wave 0, lane d, adds the
// tile before to the d-th DOUBLE sum, value after value in row order.  Same bits as agg_ordered_kernel<false> (tested against it).
struct OrdChainPlan {
    int32_t slot[kMaxAggs];   // aggregate -> its chain (lane of wave 0), or -1
};

__device__ inline long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ inline void ordered_chain_group(const AggArgs &args, const OrdChainPlan &plan, const int *__restrict__ rows, int64_t g, long long s, long long e, double *vals,
                                           unsigned long long *cnt_lds, unsigned long long *lo_lds, long long *hi_lds)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < kMaxAggs) {
        cnt_lds[threadIdx.x] = 0;
        lo_lds[threadIdx.x] = 0;
        hi_lds[threadIdx.x] = 0;
    }
    double *sum = nullptr;
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int k = 0; k < kMaxAggs; k++)
            if (k < args.n_aggs && plan.slot[k] == lane) sum = args.a[k].dsum;
    }
    double os = sum ? sum[g] : 0.0;
    __syncthreads();
    const long long tiles = (e - s + TG_ORD_TILE - 1) / TG_ORD_TILE;
    const int r = (wave - 1) * 64 + lane;
    long long row_next = (wave > 0 && s + r < e) ? rows[s + r] : 0;
    for (long long t = 0; t <= tiles; t++) {
        if (wave > 0 && t < tiles) {
            const long long j = s + t * TG_ORD_TILE + r;
            const long long row = row_next;
            if (j + TG_ORD_TILE < e) row_next = rows[j + TG_ORD_TILE];
            double *out = vals + (t & 1) * (TG_ORD_MAX_DOUBLES * TG_ORD_STRIDE) + r;
            const bool live = j < e;
#pragma unroll
            for (int k = 0; k < kMaxAggs; k++) {   // constant indices into the by-value argument block (no scratch copy)
                if (k >= args.n_aggs) break;
                const AggView &a = args.a[k];
                bool take = live;
                if (take && a.mask && ((a.mask_nulls && a.mask_nulls[row]) || !a.mask[row])) take = false;
                if (take && a.function != TGPU_AGG_COUNT_ALL && a.input_nulls && a.input_nulls[row]) take = false;
                const unsigned long long taken = __ballot(take);
                if (lane == 0 && taken) atomicAdd(&cnt_lds[k], (unsigned long long)__popcll(taken));
                if (a.function == TGPU_AGG_SUM_BIGINT) {
                    // sum = 2^32 x (sum of the signed high halves) + (sum of the unsigned low halves): neither leaves 64 bits below 2^31 rows
                    const long long v = take ? ((const long long *)a.input)[row] : 0;
                    const long long lo = wave_sum_i64((long long)(unsigned int)v), hi = wave_sum_i64(v >> 32);
                    if (lane == 0 && taken) {
                        atomicAdd(&lo_lds[k], (unsigned long long)lo);
                        atomicAdd((unsigned long long *)&hi_lds[k], (unsigned long long)hi);
                    }
                }
                else if (tg_is_minmax(a.function)) {
                    if (take) tg_minmax_update(&a.i128[g * 2], tg_minmax_encode(a.function, ((const unsigned long long *)a.input)[row]));
                }
                else if (plan.slot[k] >= 0 && live) {
                    double x = -0.0;
                    if (take) x = a.function == TGPU_AGG_AVG_BIGINT ? (double)((const long long *)a.input)[row] : ((const double *)a.input)[row];
                    out[plan.slot[k] * TG_ORD_STRIDE] = x;
                }
            }
        }
        else if (wave == 0 && t > 0 && sum) {
            const long long left = e - (s + (t - 1) * TG_ORD_TILE);
            os = tg_chain_add_tile(vals + ((t - 1) & 1) * (TG_ORD_MAX_DOUBLES * TG_ORD_STRIDE) + lane * TG_ORD_STRIDE, left < TG_ORD_TILE ? (int)left : TG_ORD_TILE, os);
        }
        __syncthreads();
    }
    if (sum) sum[g] = os;
    if (threadIdx.x == 64) {
#pragma unroll
        for (int k = 0; k < kMaxAggs; k++) {
            if (k >= args.n_aggs) break;
            const AggView &a = args.a[k];
            if (cnt_lds[k]) a.counts[g] += (long long)cnt_lds[k];
            if (a.function == TGPU_AGG_SUM_BIGINT && a.i128) {
                const __int128 big = (__int128)hi_lds[k] * ((__int128)1 << 32) + (__int128)lo_lds[k];
                const unsigned __int128 cur = ((unsigned __int128)a.i128[g * 2 + 1] << 64) | a.i128[g * 2];
                const unsigned __int128 nxt = cur + (unsigned __int128)big;
                a.i128[g * 2] = (unsigned long long)nxt;
                a.i128[g * 2 + 1] = (unsigned long long)(nxt >> 64);
            }
        }
    }
    __syncthreads();   // (the LDS words are reset by the next group of this workgroup)
}

// stretches: workgroup b = group b (every group of the page); else the workgroups share the list the lane-per-group kernel handed over
__global__ void __launch_bounds__(TG_ORD_WAVES * 64) agg_ordered_chain_kernel(AggArgs args, OrdChainPlan plan, const int *__restrict__ stretches,
                                                                               const unsigned int *__restrict__ keys, const int *__restrict__ rows, int64_t n,
                                                                               const long long *__restrict__ list, const unsigned int *__restrict__ list_count)
{
    __shared__ __attribute__((aligned(16))) double vals[2 * TG_ORD_MAX_DOUBLES * TG_ORD_STRIDE];
    __shared__ unsigned long long cnt_lds[kMaxAggs], lo_lds[kMaxAggs];
    __shared__ long long hi_lds[kMaxAggs];
    if (stretches) {
        const long long s = stretches[(size_t)blockIdx.x * 2], e = stretches[(size_t)blockIdx.x * 2 + 1];
        if (e > s) ordered_chain_group(args, plan, rows, blockIdx.x, s, e, vals, cnt_lds, lo_lds, hi_lds);
        return;
    }
    const unsigned int count = *list_count;
    for (unsigned int b = blockIdx.x; b < count; b += gridDim.x) {
        const long long g = list[(size_t)b * 2], s = list[(size_t)b * 2 + 1];
        long long lo = s, hi = n;            // the end of the group's stretch: first index whose key is larger
        while (lo < hi) { const long long mid = (lo + hi) >> 1; if (keys[mid] <= (unsigned int)(g + 1)) lo = mid + 1; else hi = mid; }
        if (lo > s) ordered_chain_group(args, plan, rows, g, s, lo, vals, cnt_lds, lo_lds, hi_lds);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Low-cardinality path (TPCH Q1: 4 groups x 8 aggregates over 600 M rows).  With a handful of groups every lane of the
// chip would hammer the same few accumulators, so the accumulators are privatised PER LANE in LDS (device_agg.h):
//     slot(g, a, lane) -> { hi, lo } double-double running sum (two-sum: the pair carries ~106 bits) + uint32 count
// laid out [g][a][256 lanes] so a wave's access is 64 consecutive 8-byte words (conflict-free, no atomics, and the order
// in which a lane adds its rows is fixed -> deterministic).  At the end of the block the 256 lane partials of each (g, a)
// are folded with double-double adds and added EXACTLY into the global limb accumulator, so both paths feed the same
// state and the final rounding is still the exact sum's.
// ---------------------------------------------------------------------------------------------------------------------
using LowCardPlan = TgLowCardPlan;

struct LowCardStates {
    TgAggState st[kMaxAggs];
};

__global__ void __launch_bounds__(kBlock) agg_lowcard_kernel(AggArgs args, LowCardStates states, LowCardPlan plan, const int32_t *__restrict__ gids, int64_t n,
                                                              TgFoldScratch fold)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    tg_lc_zero(lds, plan);
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        const int g = gids ? gids[r] : 0;
        if (g < 0) continue;
        double *hi_base = tg_lc_hi(lds, plan, g), *lo_base = tg_lc_lo(lds, plan, g);
        unsigned int *cnt_base = tg_lc_cnt(lds, plan, g);
        for (int k = 0; k < args.n_aggs; k++) {
            const AggView &a = args.a[k];
            if (a.mask && ((a.mask_nulls && a.mask_nulls[r]) || !a.mask[r])) continue;
            if (a.function != TGPU_AGG_COUNT_ALL && a.input_nulls && a.input_nulls[r]) continue;
            cnt_base[plan.cnt_slot[k] * kBlock + threadIdx.x] += 1u;
            if (tg_is_minmax(a.function)) {   // (no lane-private slot: almost no row improves a group's extreme, see tg_minmax_update)
                tg_minmax_update(&a.i128[(size_t)g * 2], tg_minmax_encode(a.function, ((const unsigned long long *)a.input)[r]));
                continue;
            }
            const int w = plan.wide_slot[k];
            if (w < 0) continue;
            if (a.function == TGPU_AGG_SUM_BIGINT) {
                tg_lc_add_bigint(hi_base, lo_base, w, ((const long long *)a.input)[r]);
                continue;
            }
            const double v = a.function == TGPU_AGG_AVG_BIGINT ? (double)((const long long *)a.input)[r] : ((const double *)a.input)[r];
            if (!(fabs(v) <= 1.7976931348623157e308)) {  // NaN / +-inf: flagged globally, not summed
                tg_flag_special(&a.special[g], v);
                continue;
            }
            tg_lc_add_double(hi_base, lo_base, w, v);
        }
    }
    tg_lc_fold(lds, plan, states.st, fold);
}

__global__ void __launch_bounds__(kBlock) agg_fold_flush_kernel(TgFoldScratch fs, int rows, int n_aggs, LowCardStates states)
{
    tg_fold_flush(fs, rows, n_aggs, states.st);
}

// limbs -> correctly rounded double (round half to even).  One pass over the limbs with O(1) state: the carry-normalised
// number is a string of 32-bit digits whose sign is only known after the top limb, so the pass tracks, for the number AND for
// its two's complement, the highest non-zero digit seen so far together with the two digits below it and whether anything
// further down is non-zero -- 96 bits around the leading one are all the rounding needs (53 + guard + sticky).
struct KulischWindow {
    int top;                  // index of the highest non-zero digit, -1 = none
    unsigned int w0, w1, w2;  // digits top, top - 1, top - 2
    bool sticky;              // a digit below top - 2 is non-zero
    unsigned int p1, p2;      // the last two digits seen
    bool below;               // a digit older than p2 is non-zero
    __device__ void init() { top = -1; w0 = w1 = w2 = 0; sticky = false; p1 = p2 = 0; below = false; }
    __device__ void push(int i, unsigned int digit)
    {
        if (digit) { top = i; w0 = digit; w1 = p1; w2 = p2; sticky = below; }
        below = below || p2 != 0;
        p2 = p1;
        p1 = digit;
    }
};

__device__ double kulisch_round(const long long *limbs, unsigned int special)
{
    if (special) {
        if ((special & 1u) || ((special & 2u) && (special & 4u))) return __longlong_as_double(0x7ff8000000000000LL);
        return (special & 2u) ? __longlong_as_double(0x7ff0000000000000LL) : __longlong_as_double((long long)0xfff0000000000000ULL);
    }
    KulischWindow pos, negw;
    pos.init();
    negw.init();
    long long carry = 0;
    unsigned int negc = 1;   // carry of the "+ 1" of the two's complement: alive while every lower digit was zero
    for (int i = 0; i < kLimbs; i++) {
        const long long v = limbs[i] + carry;
        const unsigned int digit = (unsigned int)(v & 0xffffffffLL);
        carry = v >> 32;
        pos.push(i, digit);
        const unsigned int nd = ~digit + negc;
        negc = (negc && digit == 0) ? 1u : 0u;
        negw.push(i, nd);
    }
    const bool neg = carry < 0;
    const KulischWindow &w = neg ? negw : pos;
    if (w.top < 0) return 0.0;
    const int hb = 31 - __clz((int)w.w0);
    const int t = w.top * 32 + hb;  // index of the highest set bit
    unsigned long long bits;
    if (t <= 52) {
        // denormal or the smallest normals: exactly representable, the integer IS the bit pattern
        bits = w.top == 1 ? (((unsigned long long)w.w0 << 32) | w.w1) : (unsigned long long)w.w0;
    }
    else {
        // window = digits top .. top-2, bit (64 + hb) is the leading one; t >= 53 so the 54 bits below it are inside the window
        const unsigned __int128 win = ((unsigned __int128)w.w0 << 64) | ((unsigned __int128)w.w1 << 32) | (unsigned __int128)w.w2;
        const int gpos = 64 + hb - 53;   // position of the guard bit (>= 11)
        unsigned long long mant = (unsigned long long)(win >> (gpos + 1));   // 53 bits including the leading one
        const bool guard = (unsigned long long)(win >> gpos) & 1ULL;
        bool sticky = w.sticky || (win & (((unsigned __int128)1 << gpos) - 1)) != 0;
        // digits below the window exist only when top >= 3; when top < 2 the missing digits are zero (w1 / w2 were pushed as 0)
        long long e = (long long)t - 51;
        if (guard && (sticky || (mant & 1ULL))) {
            mant++;
            if (mant >> 53) { mant >>= 1; e++; }
        }
        if (e >= 2047) bits = 0x7ff0000000000000ULL;
        else bits = ((unsigned long long)e << 52) | (mant & 0xfffffffffffffULL);
    }
    if (neg) bits |= 1ULL << 63;
    return __longlong_as_double((long long)bits);
}

struct EvalView {
    int32_t function;
    int32_t partial;
    const long long *counts;
    const long long *limbs;
    const unsigned int *special;
    const unsigned long long *i128;
    const double *dsum;   // ORDERED mode
    void *out0;           // final value, or (partial) the count channel
    uint8_t *out0_nulls;
    void *out1;           // partial: the sum channel
};
struct EvalArgs {
    int32_t n_aggs;
    int32_t pad;
    EvalView a[kMaxAggs];
};

__global__ void __launch_bounds__(kBlock) agg_evaluate_kernel(EvalArgs args, int64_t groups, unsigned int *error)
{
    const int64_t total = groups * args.n_aggs;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int k = (int)(i % args.n_aggs);
        const int64_t g = i / args.n_aggs;
        const EvalView &a = args.a[k];
        const long long cnt = a.counts[g];
        if (a.function == TGPU_AGG_COUNT_ALL || a.function == TGPU_AGG_COUNT_COLUMN) {
            ((long long *)a.out0)[g] = cnt;  // CountAggregation.java:52-56 never null
            continue;
        }
        double dsum = 0.0;
        long long lsum = 0;
        if (tg_is_minmax(a.function)) lsum = (long long)tg_minmax_decode(a.function, a.i128[g * 2]);   // (DOUBLE: the value's bits)
        else if (a.function == TGPU_AGG_SUM_BIGINT) {
            const unsigned long long lo = a.i128[g * 2], hi = a.i128[g * 2 + 1];
            lsum = (long long)lo;
            const unsigned long long expect_hi = lsum < 0 ? ~0ULL : 0ULL;
            if (hi != expect_hi) atomicOr(error, 1u);  // BigintOperators.add overflow (M/type/BigintOperators.java:47-57)
        }
        else {
            dsum = a.dsum ? a.dsum[g] : kulisch_round(&a.limbs[g * kLimbs], a.special[g]);
        }
        if (a.partial) {
            ((long long *)a.out0)[g] = cnt;
            if (a.function == TGPU_AGG_SUM_BIGINT) ((long long *)a.out1)[g] = lsum;
            else if (tg_is_minmax(a.function)) ((long long *)a.out1)[g] = cnt ? lsum : 0;
            else ((double *)a.out1)[g] = dsum;
            continue;
        }
        a.out0_nulls[g] = cnt == 0 ? 1 : 0;  // DoubleSumAggregation.java:54-63, AverageAggregations.java:69-80
        switch (a.function) {
        case TGPU_AGG_MIN_BIGINT:
        case TGPU_AGG_MAX_BIGINT:
        case TGPU_AGG_MIN_DOUBLE:
        case TGPU_AGG_MAX_DOUBLE:
        case TGPU_AGG_SUM_BIGINT: ((long long *)a.out0)[g] = cnt ? lsum : 0; break;
        case TGPU_AGG_SUM_DOUBLE: ((double *)a.out0)[g] = cnt ? dsum : 0.0; break;
        default: ((double *)a.out0)[g] = cnt ? dsum / (double)cnt : 0.0; break;
        }
    }
}

int grid_for(Context *ctx, int64_t n)
{
    int64_t blocks = ceil_div(n, kBlock);
    int64_t cap = (int64_t)ctx->cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

bool is_double_state(int32_t f) { return f == TGPU_AGG_SUM_DOUBLE || f == TGPU_AGG_AVG_DOUBLE || f == TGPU_AGG_AVG_BIGINT; }
bool is_count(int32_t f) { return f == TGPU_AGG_COUNT_ALL || f == TGPU_AGG_COUNT_COLUMN; }
bool is_minmax(int32_t f) { return f >= TGPU_AGG_MIN_BIGINT && f <= TGPU_AGG_MAX_DOUBLE; }
bool is_bigint_state(int32_t f) { return f == TGPU_AGG_SUM_BIGINT || is_minmax(f); }   // two 64-bit words per group (128-bit sum / extreme code)
// the type of the aggregate's value channel (input, PARTIAL's second channel, output)
bool bigint_valued(int32_t f) { return f == TGPU_AGG_SUM_BIGINT || f == TGPU_AGG_MIN_BIGINT || f == TGPU_AGG_MAX_BIGINT; }

// the regions a round of state growth has to clear, zeroed by ONE launch (an operator with 8 aggregates grows ~24 arrays: one
// memset each would cost more in launch gaps than the clearing itself)
struct ZeroList {
    static constexpr int kMax = 5 * kMaxAggs;
    unsigned long long *ptr[kMax];
    unsigned long long words[kMax];   // 8-byte words
    int n;
};

__global__ void __launch_bounds__(kBlock) multi_zero_kernel(ZeroList z)
{
    for (int i = 0; i < z.n; i++)
        for (unsigned long long w = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; w < z.words[i]; w += (unsigned long long)gridDim.x * kBlock) z.ptr[i][w] = 0ULL;
}

// new array of new_elems elements: the old contents are carried over, the tail is queued for clearing (sizes are multiples of 8)
BufferPtr grow(Context *ctx, BufferPtr old, int64_t old_elems, int64_t new_elems, int elem_bytes, ZeroList &zeros)
{
    BufferPtr nb = ctx->alloc((size_t)new_elems * elem_bytes);
    const size_t keep = old ? (size_t)old_elems * elem_bytes : 0;
    if (keep) HIP_CHECK(hipMemcpyAsync(nb->ptr(), old->ptr(), keep, hipMemcpyDeviceToDevice, ctx->stream()));
    TG_CHECK_STATE(zeros.n < ZeroList::kMax && keep % 8 == 0 && ((size_t)new_elems * elem_bytes) % 8 == 0, "accumulator growth list");
    zeros.ptr[zeros.n] = (unsigned long long *)(nb->as<uint8_t>() + keep);
    zeros.words[zeros.n] = ((size_t)new_elems * elem_bytes - keep) / 8;
    zeros.n++;
    return nb;
}

}  // namespace

GroupedAccumulators::GroupedAccumulators(Context *ctx, std::vector<tgpu_agg_spec> specs, int32_t step) : ctx_(ctx), step_(step)
{
    TG_CHECK_ARG((int)specs.size() <= kMaxAggs, "at most 16 aggregates per operator");
    for (auto &s : specs) {
        TG_CHECK_ARG(s.function >= TGPU_AGG_COUNT_ALL && s.function <= TGPU_AGG_MAX_DOUBLE, "unknown aggregate function");
        State st;
        st.spec = s;
        states_.push_back(st);
    }
    error_ = ctx_->alloc_zero(4);
}

GroupedAccumulators::FoldScratch GroupedAccumulators::fold_scratch(int64_t blocks, int64_t group_capacity)
{
    const int64_t stride = group_capacity * (int64_t)states_.size();
    if (!fold_partials_ || stride != fold_stride_ || blocks > fold_rows_) {
        flush_fold();   // (into the states, before the rows change shape)
        fold_rows_ = std::max<int64_t>(blocks, fold_rows_);
        fold_stride_ = stride;
        fold_partials_ = ctx_->alloc_zero((size_t)fold_rows_ * (size_t)stride * 24);
        fold_pending_ = ctx_->alloc_zero((size_t)fold_rows_ * (size_t)stride * 24);
    }
    ensure(group_capacity);   // the flush addresses the states of every group a row has room for
    fold_dirty_ = true;
    return {fold_partials_->as<unsigned long long>(), (int32_t)stride, fold_pending_->as<unsigned long long>()};
}

__global__ void __launch_bounds__(kBlock) agg_resolve_pending_kernel(TgFoldScratch fs, int n_aggs, LowCardStates states, int commit)
{
    if (commit) tg_commit_pending(fs, n_aggs, states.st, true);
    else {
        unsigned long long *pend = fs.pending + (size_t)blockIdx.x * fs.stride * 3;
        for (int i = threadIdx.x; i < fs.stride * 3; i += kBlock) pend[i] = 0;
    }
}

void GroupedAccumulators::resolve_pending(int64_t blocks, bool commit)
{
    if (!fold_pending_ || blocks <= 0) return;
    TG_CHECK_STATE(blocks <= fold_rows_, "pending rows beyond the fold scratch");
    LowCardStates states{};
    for (size_t k = 0; k < states_.size(); k++) states.st[k].function = device_state((int)k).function;
    agg_resolve_pending_kernel<<<(int)blocks, kBlock, 0, ctx_->stream()>>>(
        TgFoldScratch{fold_partials_->as<unsigned long long>(), (int)fold_stride_, fold_pending_->as<unsigned long long>()}, (int)states_.size(), states, commit ? 1 : 0);
    check_launch("agg_resolve_pending");
    fold_dirty_ = true;
}

void GroupedAccumulators::flush_fold()
{
    if (!fold_dirty_) return;
    fold_dirty_ = false;
    LowCardStates states{};
    for (size_t k = 0; k < states_.size(); k++) {
        const DeviceState d = device_state((int)k);
        states.st[k].function = d.function;
        states.st[k].counts = d.counts;
        states.st[k].limbs = d.limbs;
        states.st[k].special = d.special;
        states.st[k].i128 = d.i128;
        states.st[k].dsum = nullptr;
    }
    ProfileScope ps(ctx_, "agg_fold_flush");
    agg_fold_flush_kernel<<<(int)ceil_div(fold_stride_, kBlock / 64), kBlock, 0, ctx_->stream()>>>(TgFoldScratch{fold_partials_->as<unsigned long long>(), (int)fold_stride_, nullptr},
                                                                                                (int)fold_rows_, (int)states_.size(), states);
    check_launch("agg_fold_flush");
}

int64_t GroupedAccumulators::HostStates::bytes() const
{
    int64_t b = 0;
    for (const Agg &a : aggs) b += (int64_t)(a.counts.size() * 8 + a.limbs.size() * 8 + a.special.size() * 4 + a.i128.size() * 8 + a.dsum.size() * 8);
    return b;
}

GroupedAccumulators::HostStates GroupedAccumulators::dump(int64_t groups)
{
    HostStates h;
    h.groups = groups;
    if (groups <= 0) return h;
    ensure(groups);
    flush_fold();
    for (State &st : states_) {
        HostStates::Agg a;
        a.counts.resize((size_t)groups);
        ctx_->download(a.counts.data(), st.counts->ptr(), (size_t)groups * 8);
        if (st.limbs) {
            a.limbs.resize((size_t)groups * kLimbs);
            ctx_->download(a.limbs.data(), st.limbs->ptr(), (size_t)groups * kLimbs * 8);
            a.special.resize((size_t)groups);
            ctx_->download(a.special.data(), st.special->ptr(), (size_t)groups * 4);
        }
        if (st.dsum) {
            a.dsum.resize((size_t)groups);
            ctx_->download(a.dsum.data(), st.dsum->ptr(), (size_t)groups * 8);
        }
        if (st.i128) {
            a.i128.resize((size_t)groups * 2);
            ctx_->download(a.i128.data(), st.i128->ptr(), (size_t)groups * 16);
        }
        h.aggs.push_back(std::move(a));
    }
    return h;
}

namespace {
struct MergeArgs {
    int32_t n_aggs;
    struct A {
        int32_t function, pad;
        long long *counts, *limbs;
        unsigned int *special;
        unsigned long long *i128;
        const long long *r_counts, *r_limbs;
        const unsigned int *r_special;
        const unsigned long long *r_i128;
        const double *r_dsum;
        double *dsum;   // ORDERED target: the run's sum joins it with ONE double add -- DoubleSumAggregation.combine, run after run
    } a[kMaxAggs];
};

// one thread per (run group, limb): the run's limbs stream in, the live group's limbs are contiguous.  A live group receives at most
// one run group per launch (the keys of a run are distinct), so plain adds suffice.
__global__ void __launch_bounds__(kBlock) agg_merge_states_kernel(MergeArgs args, const int32_t *__restrict__ gids, int64_t n)
{
    const int64_t total = n * kLimbs;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t i = idx / kLimbs;
        const int l = (int)(idx - i * kLimbs);
        const int64_t g = gids ? gids[i] : 0;
        for (int k = 0; k < args.n_aggs; k++) {
            const MergeArgs::A &a = args.a[k];
            if (a.r_limbs) a.limbs[g * kLimbs + l] += a.r_limbs[idx];
            if (l != 0) continue;
            a.counts[g] += a.r_counts[i];
            if (a.r_special) a.special[g] |= a.r_special[i];
            if (a.r_dsum && a.dsum) a.dsum[g] = a.dsum[g] + a.r_dsum[i];   // state.setDouble(state.getDouble() + otherState.getDouble()) (DoubleSumAggregation.java:48-52)
            else if (a.r_dsum && a.limbs) {
                const double v = a.r_dsum[i];
                if (!(fabs(v) <= 1.7976931348623157e308)) tg_flag_special(&a.special[g], v);
                else tg_kulisch_add(&a.limbs[g * kLimbs], &a.special[g], v);   // (atomic adds: lane l == 0 of this group races with the limb lanes above)
            }
            if (a.r_i128 && tg_is_minmax(a.function)) {   // combine(): the better of the two extremes (AbstractMinMaxAggregationFunction.java:251-254)
                if (a.r_i128[i * 2] > a.i128[g * 2]) a.i128[g * 2] = a.r_i128[i * 2];
            }
            else if (a.r_i128) {
                const unsigned long long lo = a.i128[g * 2], add = a.r_i128[i * 2];
                a.i128[g * 2] = lo + add;
                a.i128[g * 2 + 1] += a.r_i128[i * 2 + 1] + ((lo + add) < lo ? 1ULL : 0ULL);
            }
        }
    }
}
}  // namespace

void GroupedAccumulators::merge(const int32_t *gids, const HostStates &run, int64_t live_groups)
{
    if (states_.empty() || run.groups <= 0) return;
    TG_CHECK_STATE(run.aggs.size() == states_.size(), "spilled run has a different number of aggregates");
    // Runs whose double sums are running (row-order) sums -- ORDERED runs -- are combined the reference's way when `combine_ordered` says
    // every run is one: the target keeps plain doubles and each run's sum joins it with one double add, run after run in spill order
    // (SpillableHashAggregationBuilder.mergeFromDisk :229-240 feeds the runs' intermediate states to addIntermediate = combine).  Else the
    // target is EXACT: limbs add limb-wise (exact, bit-identical to the unspilled result), a stray ORDERED run's sum is one exact addend.
    bool run_ordered = false;
    for (size_t k = 0; k < states_.size(); k++) run_ordered = run_ordered || !run.aggs[k].dsum.empty();
    if (mode_ == Mode::UNDECIDED) mode_ = (combine_ordered_ && run_ordered) ? Mode::ORDERED : Mode::EXACT;
    TG_CHECK_STATE(mode_ == Mode::EXACT || run_ordered, "an exact run cannot join row-order sums");
    ensure(live_groups > 0 ? live_groups : 1);
    // an aggregate of a run carries either limbs (EXACT run) or running double sums (ORDERED run), never both: the plain limb adds and
    // the atomic adds of a double never meet on one group's limbs
    MergeArgs args{};
    args.n_aggs = (int32_t)states_.size();
    std::vector<BufferPtr> keep;
    auto up = [&](const void *src, size_t bytes) -> void * {
        if (!bytes) return nullptr;
        BufferPtr b = ctx_->alloc(bytes);
        ctx_->upload(b->ptr(), src, bytes);
        keep.push_back(b);
        return b->ptr();
    };
    for (size_t k = 0; k < states_.size(); k++) {
        State &st = states_[k];
        const HostStates::Agg &r = run.aggs[k];
        MergeArgs::A &a = args.a[k];
        a.function = st.spec.function;
        a.counts = st.counts->as<long long>();
        a.limbs = st.limbs ? st.limbs->as<long long>() : nullptr;
        a.special = st.special ? st.special->as<unsigned int>() : nullptr;
        a.i128 = st.i128 ? st.i128->as<unsigned long long>() : nullptr;
        a.dsum = st.dsum ? st.dsum->as<double>() : nullptr;
        a.r_counts = (const long long *)up(r.counts.data(), r.counts.size() * 8);
        a.r_limbs = (const long long *)up(r.limbs.data(), r.limbs.size() * 8);
        a.r_special = (const unsigned int *)up(r.special.data(), r.special.size() * 4);
        a.r_i128 = (const unsigned long long *)up(r.i128.data(), r.i128.size() * 8);
        a.r_dsum = (const double *)up(r.dsum.data(), r.dsum.size() * 8);
    }
    ctx_->sync();   // the uploads read host vectors owned by the caller
    ProfileScope ps(ctx_, "agg_merge_states");
    agg_merge_states_kernel<<<grid_for(ctx_, run.groups * kLimbs), kBlock, 0, ctx_->stream()>>>(args, gids, run.groups);
    check_launch("agg_merge_states");
    ctx_->sync();
}

GroupedAccumulators::DeviceState GroupedAccumulators::device_state(int k) const
{
    const State &st = states_[(size_t)k];
    DeviceState d;
    d.function = st.spec.function;
    d.counts = st.counts ? st.counts->as<long long>() : nullptr;
    d.limbs = st.limbs ? st.limbs->as<long long>() : nullptr;
    d.special = st.special ? st.special->as<unsigned int>() : nullptr;
    d.i128 = st.i128 ? st.i128->as<unsigned long long>() : nullptr;
    d.dsum = st.dsum ? st.dsum->as<double>() : nullptr;
    return d;
}

const std::vector<tgpu_agg_spec> GroupedAccumulators::specs() const
{
    std::vector<tgpu_agg_spec> v;
    for (auto &s : states_) v.push_back(s.spec);
    return v;
}

int GroupedAccumulators::output_channel_count() const
{
    return step_ == TGPU_STEP_PARTIAL ? intermediate_channel_count() : (int)states_.size();
}

int GroupedAccumulators::intermediate_channel_count() const
{
    int n = 0;
    for (auto &s : states_) n += is_count(s.spec.function) ? 1 : 2;
    return n;
}

int64_t GroupedAccumulators::estimated_size() const
{
    int64_t s = 0;
    for (auto &st : states_) {
        s += st.cap * 8;
        if (st.limbs) s += st.cap * (kLimbs * 8 + 4);
        if (st.dsum) s += st.cap * 8;
        if (st.i128) s += st.cap * 16;
    }
    return s;
}

void GroupedAccumulators::ensure(int64_t groups)
{
    ZeroList zeros{};
    for (auto &st : states_) {
        if (groups <= st.cap) continue;
        int64_t cap = st.cap ? st.cap : 256;
        while (cap < groups) cap <<= 1;
        st.counts = grow(ctx_, st.counts, st.cap, cap, 8, zeros);
        if (is_double_state(st.spec.function)) {
            if (mode_ == Mode::ORDERED) st.dsum = grow(ctx_, st.dsum, st.cap, cap, 8, zeros);
            else {
                st.limbs = grow(ctx_, st.limbs, st.cap * kLimbs, cap * kLimbs, 8, zeros);
                st.special = grow(ctx_, st.special, st.cap, cap, 4, zeros);
            }
        }
        if (is_bigint_state(st.spec.function)) st.i128 = grow(ctx_, st.i128, st.cap * 2, cap * 2, 8, zeros);
        st.cap = cap;
    }
    if (zeros.n) {
        unsigned long long most = 0;
        for (int i = 0; i < zeros.n; i++) most = std::max(most, zeros.words[i]);
        multi_zero_kernel<<<grid_for(ctx_, (int64_t)most), kBlock, 0, ctx_->stream()>>>(zeros);
        check_launch("multi_zero");
    }
}

static void check_channel(const DevicePage &page, int ch, int32_t want_type, const char *what)
{
    TG_CHECK_ARG(ch >= 0 && ch < (int)page.cols.size(), std::string(what) + ": channel out of range");
    if (want_type) TG_CHECK_ARG(page.cols[ch].type == want_type, std::string(what) + ": unexpected channel type " + type_name(page.cols[ch].type));
}

// lane-private LDS bytes one group needs on the low-cardinality path (AOT layout: every aggregate has its own slots)
static int64_t lowcard_bytes_per_group(const std::vector<tgpu_agg_spec> &specs)
{
    int64_t wide = 0;
    for (auto &s : specs) wide += (is_count(s.function) || is_minmax(s.function)) ? 0 : 1;
    return wide * 2 * kBlock * 8 + (int64_t)specs.size() * kBlock * 4;
}

// The mode is fixed by the first page: ORDERED when it is allowed, the page has group ids and its groups do not fit the
// low-cardinality LDS path; else EXACT.
void GroupedAccumulators::decide_mode(int64_t groups, int64_t lowcard_max_groups)
{
    if (mode_ != Mode::UNDECIDED) return;
    const bool many = groups > lowcard_max_groups || force_ordered_;
    mode_ = (allow_ordered_ && many && getenv("TGPU_DISABLE_ORDERED") == nullptr) ? Mode::ORDERED : Mode::EXACT;
}

void GroupedAccumulators::decide(int64_t groups_in_prefix, int64_t lowcard_max_groups)
{
    if (lowcard_max_groups <= 0) lowcard_max_groups = (160 * 1024) / std::max<int64_t>(lowcard_bytes_per_group(specs()), 1);
    decide_mode(groups_in_prefix > 0 ? groups_in_prefix : 1, lowcard_max_groups);
}

static __global__ void __launch_bounds__(kBlock) max_gid_kernel(const int32_t *__restrict__ gids, const uint8_t *__restrict__ gids8, int64_t m, int *out)
{
    int best = -1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
        const int g = gids8 ? (int)gids8[i] - 1 : gids[i];
        best = g > best ? g : best;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_down(best, d, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best >= 0) atomicMax(out, best);
}

int64_t GroupedAccumulators::groups_among(Context *ctx, const int32_t *gids, const uint8_t *gids8, int64_t m)
{
    if (m <= 0 || (!gids && !gids8)) return 0;
    BufferPtr out = ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(out->ptr(), 0xff, 4, ctx->stream()));   // -1
    max_gid_kernel<<<(int)std::min<int64_t>(ceil_div(m, kBlock), 64), kBlock, 0, ctx->stream()>>>(gids, gids8, m, out->as<int>());
    check_launch("max_gid");
    return (int64_t)ctx->read_scalar(out->as<int>()) + 1;
}

bool GroupedAccumulators::begin_ordered(const int32_t *gids, int64_t n, int64_t groups, int64_t lowcard_max_groups, BufferPtr &keys, BufferPtr &rows)
{
    if (n <= 0) return false;
    decide_mode(groups > 0 ? groups : 1, lowcard_max_groups);   // the first page fixes the mode, whatever path it takes
    if (mode_ != Mode::ORDERED) return false;
    TG_CHECK_STATE(gids != nullptr, "ordered accumulation needs int32 group ids");
    ensure(groups > 0 ? groups : 1);
    sort_rows_by_group(gids, n, groups, keys, rows);
    return true;
}

BufferPtr GroupedAccumulators::group_stretches(const unsigned int *keys, int64_t n, int64_t ids)
{
    BufferPtr st = ctx_->alloc_zero((size_t)ids * 8);
    ordered_stretches_kernel<<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(keys, n, st->as<int>());
    check_launch("ordered_stretches");
    return st;
}

// (group id + 1, row) pairs of the page in (group, row) order: stable LSD radix sort on the bits the group ids use
void GroupedAccumulators::sort_rows_by_group(const int32_t *gids, int64_t n, int64_t groups, BufferPtr &keys, BufferPtr &rows)
{
    BufferPtr keys_in = ctx_->alloc((size_t)n * 4), rows_in = ctx_->alloc((size_t)n * 4), unsorted = ctx_->alloc_zero(4);
    ordered_keys_kernel<<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(gids, n, keys_in->as<unsigned int>(), rows_in->as<int>(), unsorted->as<unsigned int>());
    check_launch("ordered_keys");
    // one small read-back (~25 us) against three or four radix sort passes over the page
    if (getenv("TGPU_ALWAYS_SORT") == nullptr && ctx_->read_scalar(unsorted->as<unsigned int>()) == 0) {
        keys = keys_in;
        rows = rows_in;
        return;
    }
    keys = ctx_->alloc((size_t)n * 4);
    rows = ctx_->alloc((size_t)n * 4);
    unsigned int end_bit = 1;
    while (end_bit < 32 && (1ull << end_bit) <= (unsigned long long)groups) end_bit++;   // keys are in [0, groups]
    size_t temp_bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in->as<unsigned int>(), keys->as<unsigned int>(), rows_in->as<int>(), rows->as<int>(), (size_t)n, 0,
                                        end_bit, ctx_->stream()));
    BufferPtr temp = ctx_->alloc(temp_bytes ? temp_bytes : 1);
    HIP_CHECK(rocprim::radix_sort_pairs(temp->ptr(), temp_bytes, keys_in->as<unsigned int>(), keys->as<unsigned int>(), rows_in->as<int>(), rows->as<int>(), (size_t)n, 0,
                                        end_bit, ctx_->stream()));
}

void GroupedAccumulators::add_input(const int32_t *gids, int64_t n, const DevicePage &page, int64_t group_count)
{
    if (states_.empty() || n <= 0) return;
    BufferPtr zero_gids;
    if (!gids && force_ordered()) {   // a global aggregation in Java order: one group, id 0 for every row
        zero_gids = ctx_->alloc_zero((size_t)n * 4);
        gids = zero_gids->as<int32_t>();
    }
    if (gids) decide_mode(group_count > 0 ? group_count : 1, (160 * 1024) / std::max<int64_t>(lowcard_bytes_per_group(specs()), 1));
    else if (mode_ == Mode::UNDECIDED) mode_ = Mode::EXACT;
    {
        // a low-cardinality launch leaves folded partials for every group its LDS has room for (fold_scratch): reserve those states
        // now, before the argument block below takes their addresses
        const int64_t lowcard_capacity = (160 * 1024) / std::max<int64_t>(lowcard_bytes_per_group(specs()), 1);
        const int64_t g = group_count > 0 ? group_count : 1;
        ensure(mode_ == Mode::EXACT && g <= lowcard_capacity ? lowcard_capacity : g);
    }
    AggArgs args{};
    args.n_aggs = (int32_t)states_.size();
    for (size_t k = 0; k < states_.size(); k++) {
        State &st = states_[k];
        AggView &a = args.a[k];
        a.function = st.spec.function;
        if (st.spec.function != TGPU_AGG_COUNT_ALL) {
            int32_t want = 0;
            if (bigint_valued(st.spec.function) || st.spec.function == TGPU_AGG_AVG_BIGINT) want = TGPU_BIGINT;
            if (st.spec.function == TGPU_AGG_MIN_DOUBLE || st.spec.function == TGPU_AGG_MAX_DOUBLE) want = TGPU_DOUBLE;
            if (st.spec.function == TGPU_AGG_SUM_DOUBLE || st.spec.function == TGPU_AGG_AVG_DOUBLE) want = TGPU_DOUBLE;
            check_channel(page, st.spec.input_channel, want, "aggregate input");
            a.input = page.cols[st.spec.input_channel].values;
            a.input_nulls = page.cols[st.spec.input_channel].nulls;
        }
        if (st.spec.mask_channel >= 0) {
            check_channel(page, st.spec.mask_channel, TGPU_BOOLEAN, "aggregate mask");
            a.mask = (const uint8_t *)page.cols[st.spec.mask_channel].values;
            a.mask_nulls = page.cols[st.spec.mask_channel].nulls;
        }
        a.counts = st.counts->as<long long>();
        a.limbs = st.limbs ? st.limbs->as<long long>() : nullptr;
        a.special = st.special ? st.special->as<unsigned int>() : nullptr;
        a.i128 = st.i128 ? st.i128->as<unsigned long long>() : nullptr;
        a.dsum = st.dsum ? st.dsum->as<double>() : nullptr;
    }
    if (mode_ == Mode::ORDERED) {
        TG_CHECK_STATE(gids != nullptr, "ordered accumulation needs group ids");
        // few groups with many rows each: one workgroup per group, the DOUBLE sums as chains fed from LDS
        OrdChainPlan plan{};
        int doubles = 0;
        for (int k = 0; k < kMaxAggs; k++) plan.slot[k] = (k < args.n_aggs && args.a[k].dsum) ? doubles++ : -1;
        const int64_t ids = group_count > 0 ? group_count : 1;
        const bool chained = doubles <= TG_ORD_MAX_DOUBLES && ids <= ord_chain_max_groups() && n >= ids * kOrdChainMinRows && getenv("TGPU_DISABLE_ORDERED_CHAIN") == nullptr;
        ProfileScope ps(ctx_, chained ? "agg_accumulate_ordered_chain" : "agg_accumulate_ordered");
        BufferPtr keys, rows;
        sort_rows_by_group(gids, n, group_count, keys, rows);
        if (chained) {
            BufferPtr stretches = group_stretches(keys->as<unsigned int>(), n, ids);
            agg_ordered_chain_kernel<<<(int)ids, TG_ORD_WAVES * 64, 0, ctx_->stream()>>>(args, plan, stretches->as<int>(), keys->as<unsigned int>(), rows->as<int>(), n, nullptr,
                                                                                          nullptr);
            check_launch("agg_accumulate_ordered_chain");
            return;
        }
        // one lane per group; a group of more than kOrdHandoffRows rows (one key far more frequent than the others) goes to the chained
        // kernel after that many rows
        const int64_t long_groups = n / kOrdHandoffRows;   // at most this many groups can be that long
        const bool handoff = long_groups > 0 && doubles <= TG_ORD_MAX_DOUBLES && getenv("TGPU_DISABLE_ORDERED_CHAIN") == nullptr;
        BufferPtr list = handoff ? ctx_->alloc((size_t)long_groups * 16) : nullptr, list_count = handoff ? ctx_->alloc_zero(8) : nullptr;
        agg_ordered_kernel<false><<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(args, keys->as<unsigned int>(), rows->as<int>(), n, error_->as<unsigned int>(),
                                                                                      handoff ? kOrdHandoffRows : 0, handoff ? list->as<long long>() : nullptr,
                                                                                      handoff ? list_count->as<unsigned int>() : nullptr);
        check_launch("agg_accumulate_ordered");
        if (handoff) {
            agg_ordered_chain_kernel<<<(int)std::min<int64_t>(long_groups, ctx_->cu_count()), TG_ORD_WAVES * 64, 0, ctx_->stream()>>>(
                args, plan, nullptr, keys->as<unsigned int>(), rows->as<int>(), n, list->as<long long>(), list_count->as<unsigned int>());
            check_launch("agg_accumulate_ordered_handoff");
        }
        return;
    }
    // low-cardinality path: lane-private LDS accumulators when all groups x states fit in one CU's LDS
    LowCardPlan plan{};
    plan.n_aggs = args.n_aggs;
    for (int k = 0; k < args.n_aggs; k++) {
        plan.wide_slot[k] = (is_count(args.a[k].function) || is_minmax(args.a[k].function)) ? -1 : plan.n_wide++;   // (min / max: straight to the state word)
        plan.cnt_slot[k] = k;
        plan.count_from_rows[k] = 0;
    }
    plan.n_cnt = plan.n_aggs;
    plan.rows_slot = -1;
    plan.per_group_bytes = plan.n_wide * 2 * kBlock * 8 + plan.n_cnt * kBlock * 4;
    const int64_t groups = group_count > 0 ? group_count : 1;
    const int64_t lds_bytes = groups * plan.per_group_bytes;
    const bool lowcard_enabled = getenv("TGPU_DISABLE_LOWCARD") == nullptr;
    if (lowcard_enabled && lds_bytes <= 160 * 1024 && n >= 4096) {
        plan.n_groups = (int32_t)groups;
        static bool attr_set = false;
        if (!attr_set) {
            HIP_CHECK(hipFuncSetAttribute((const void *)agg_lowcard_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        const int per_cu = (int)std::max<int64_t>(1, std::min<int64_t>(8, (160 * 1024) / lds_bytes));
        int64_t blocks = std::min<int64_t>((int64_t)ctx_->cu_count() * per_cu, ceil_div(n, kBlock));
        ProfileScope ps(ctx_, "agg_accumulate_lowcard");
        LowCardStates states{};
        for (int k = 0; k < args.n_aggs; k++) {
            states.st[k].function = args.a[k].function;
            states.st[k].counts = args.a[k].counts;
            states.st[k].limbs = args.a[k].limbs;
            states.st[k].special = args.a[k].special;
            states.st[k].i128 = args.a[k].i128;
            states.st[k].dsum = nullptr;
        }
        const FoldScratch fs = fold_scratch(blocks, (160 * 1024) / plan.per_group_bytes);   // (states reserved above: nothing moves)
        agg_lowcard_kernel<<<(int)blocks, kBlock, (size_t)lds_bytes, ctx_->stream()>>>(args, states, plan, gids, n, TgFoldScratch{fs.partials, fs.stride, nullptr});
        check_launch("agg_accumulate_lowcard");
        return;
    }
    ProfileScope ps(ctx_, "agg_accumulate");
    agg_accumulate_kernel<false><<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(args, gids, n);
    check_launch("agg_accumulate");
}

void GroupedAccumulators::add_intermediate(const int32_t *gids, int64_t n, const DevicePage &page, int64_t group_count)
{
    if (states_.empty() || n <= 0) return;
    BufferPtr zero_gids;
    if (!gids && force_ordered()) {
        zero_gids = ctx_->alloc_zero((size_t)n * 4);
        gids = zero_gids->as<int32_t>();
    }
    if (gids) decide_mode(group_count > 0 ? group_count : 1, (160 * 1024) / std::max<int64_t>(lowcard_bytes_per_group(specs()), 1));
    else if (mode_ == Mode::UNDECIDED) mode_ = Mode::EXACT;
    ensure(group_count > 0 ? group_count : 1);
    AggArgs args{};
    args.n_aggs = (int32_t)states_.size();
    for (size_t k = 0; k < states_.size(); k++) {
        State &st = states_[k];
        AggView &a = args.a[k];
        a.function = st.spec.function;
        int ch = st.spec.input_channel;
        check_channel(page, ch, TGPU_BIGINT, "intermediate count");
        a.input = page.cols[ch].values;
        ch++;
        if (!is_count(st.spec.function)) {
            check_channel(page, ch, bigint_valued(st.spec.function) ? TGPU_BIGINT : TGPU_DOUBLE, "intermediate sum");
            a.input2 = page.cols[ch].values;
            ch++;
        }
        a.counts = st.counts->as<long long>();
        a.limbs = st.limbs ? st.limbs->as<long long>() : nullptr;
        a.special = st.special ? st.special->as<unsigned int>() : nullptr;
        a.i128 = st.i128 ? st.i128->as<unsigned long long>() : nullptr;
        a.dsum = st.dsum ? st.dsum->as<double>() : nullptr;
    }
    if (mode_ == Mode::ORDERED) {
        TG_CHECK_STATE(gids != nullptr, "ordered accumulation needs group ids");
        ProfileScope ps(ctx_, "agg_combine_ordered");
        BufferPtr keys, rows;
        sort_rows_by_group(gids, n, group_count, keys, rows);
        agg_ordered_kernel<true><<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(args, keys->as<unsigned int>(), rows->as<int>(), n, error_->as<unsigned int>(), 0, nullptr,
                                                                                     nullptr);
        check_launch("agg_combine_ordered");
        return;
    }
    ProfileScope ps(ctx_, "agg_combine");
    agg_accumulate_kernel<true><<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(args, gids, n);
    check_launch("agg_combine");
}

void GroupedAccumulators::evaluate(int64_t groups, std::vector<DeviceColumn> &out)
{
    if (states_.empty()) return;
    ensure(groups > 0 ? groups : 1);
    flush_fold();
    EvalArgs args{};
    args.n_aggs = (int32_t)states_.size();
    const bool partial = step_ == TGPU_STEP_PARTIAL;
    const int64_t alloc_n = groups > 0 ? groups : 1;
    for (size_t k = 0; k < states_.size(); k++) {
        State &st = states_[k];
        EvalView &a = args.a[k];
        a.function = st.spec.function;
        a.partial = partial ? 1 : 0;
        a.counts = st.counts->as<long long>();
        a.limbs = st.limbs ? st.limbs->as<long long>() : nullptr;
        a.special = st.special ? st.special->as<unsigned int>() : nullptr;
        a.i128 = st.i128 ? st.i128->as<unsigned long long>() : nullptr;
        a.dsum = st.dsum ? st.dsum->as<double>() : nullptr;
        DeviceColumn c0;
        c0.n = groups;
        c0.values_buf = ctx_->alloc((size_t)alloc_n * 8);
        c0.values = c0.values_buf->ptr();
        a.out0 = c0.values_buf->ptr();
        if (is_count(st.spec.function) || partial) {
            c0.type = TGPU_BIGINT;
            out.push_back(c0);
            if (partial && !is_count(st.spec.function)) {
                DeviceColumn c1;
                c1.n = groups;
                c1.type = bigint_valued(st.spec.function) ? TGPU_BIGINT : TGPU_DOUBLE;
                c1.values_buf = ctx_->alloc((size_t)alloc_n * 8);
                c1.values = c1.values_buf->ptr();
                a.out1 = c1.values_buf->ptr();
                out.push_back(c1);
            }
        }
        else {
            c0.type = bigint_valued(st.spec.function) ? TGPU_BIGINT : TGPU_DOUBLE;
            c0.nulls_buf = ctx_->alloc((size_t)alloc_n);
            c0.nulls = c0.nulls_buf->as<uint8_t>();
            a.out0_nulls = c0.nulls_buf->as<uint8_t>();
            out.push_back(c0);
        }
    }
    if (groups <= 0) return;
    HIP_CHECK(hipMemsetAsync(error_->ptr(), 0, 4, ctx_->stream()));
    {
        ProfileScope ps(ctx_, "agg_evaluate");
        agg_evaluate_kernel<<<grid_for(ctx_, groups * args.n_aggs), kBlock, 0, ctx_->stream()>>>(args, groups, error_->as<unsigned int>());
        check_launch("agg_evaluate");
    }
    // only sum(bigint) can raise here (its total left int64): without one there is nothing to wait for
    bool can_raise = false;
    for (auto &st : states_) can_raise = can_raise || st.spec.function == TGPU_AGG_SUM_BIGINT;
    if (can_raise) {
        unsigned int err = ctx_->read_scalar(error_->as<unsigned int>());
        if (err) fail(TGPU_ERR_NUMERIC_VALUE_OUT_OF_RANGE, "bigint addition overflow");
    }
}

}  // namespace tgpu
