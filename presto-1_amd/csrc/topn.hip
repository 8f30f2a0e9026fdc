// topn.hip -- TopN (K11).  A page's n first rows are found without comparison-sorting the page:
//   1. every row gets a 64-bit ORDER CODE of its first sort key: the key's order-preserving bit pattern (BIGINT / INTEGER / DATE
//      biased, DOUBLE in Double.compare order, BOOLEAN, VARCHAR = first 8 bytes big-endian), complemented for DESC, shifted
//      right by one and topped with a null bit placed by the SortOrder -- code(a) < code(b) implies a sorts before b, equal
//      codes decide nothing;
//   2. a CUTOFF code that certainly admits the n first rows: the n-th smallest code of a strided sample of 64 K rows (the n-th
//      smallest of a subset is never below the n-th smallest of the whole); small pages skip this and keep every row;
//   3. the rows whose code does not exceed the cutoff are the candidates, compacted in input order (flags + scan): about
//      n x rows / 64 K of them.  If the codes are so coarse that the candidates are still many (few distinct values), they are
//      radix sorted by code (rocPRIM, stable) and cut at the ties of the n-th code;
//   4. only the candidates are sorted with the full comparator (rocPRIM merge sort over row numbers: all sort keys, then the row
//      number, so the order is total and rows that compare equal keep their input order), the first n are gathered.
// Streaming: each page contributes its winners to a small candidate store; result() runs the same selection over the store.
#include "topn.h"
#include "kernels.h"
#include "device_hash.h"
#include "device_cols.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace tgpu {

namespace {

constexpr int kBlock = 256;

// sort keys in device memory (read through a pointer: run-time column indices are then plain scalar loads)
struct TopNKeys {
    TgKeyCols cols;                       // the sort channels, in priority order
    int order[TG_MAX_KEY_CHANNELS];       // tgpu_sort_order per key
};

__device__ __forceinline__ bool asc(int order) { return order == TGPU_SORT_ASC_NULLS_FIRST || order == TGPU_SORT_ASC_NULLS_LAST; }
__device__ __forceinline__ bool nulls_first(int order) { return order == TGPU_SORT_ASC_NULLS_FIRST || order == TGPU_SORT_DESC_NULLS_FIRST; }

// Double.compare order as an unsigned key: -inf < ... < -0.0 < +0.0 < ... < +inf < NaN (all NaNs equal)
__device__ __forceinline__ unsigned long long double_order_bits(unsigned long long bits)
{
    const double v = __longlong_as_double((long long)bits);
    if (v != v) bits = 0x7ff8000000000000ULL;
    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ULL);
}

// order-preserving 64-bit pattern of a non-null cell (VARCHAR: a prefix)
__device__ __forceinline__ unsigned long long cell_order_bits(const TgColView &c, long long r)
{
    switch (c.type) {
    case TGPU_BIGINT: return (unsigned long long)((const long long *)c.values)[r] ^ 0x8000000000000000ULL;
    case TGPU_INTEGER:
    case TGPU_DATE: return (unsigned long long)(long long)((const int *)c.values)[r] ^ 0x8000000000000000ULL;
    case TGPU_DOUBLE: return double_order_bits(((const unsigned long long *)c.values)[r]);
    case TGPU_BOOLEAN: return ((const unsigned char *)c.values)[r] ? 1ULL : 0ULL;
    case TGPU_VARCHAR: {
        const int a = c.offsets[r], l = c.offsets[r + 1] - a;
        const unsigned char *p = (const unsigned char *)c.values + a;
        unsigned long long v = 0;
        for (int i = 0; i < 8; i++) v = (v << 8) | (i < l ? (unsigned long long)p[i] : 0ULL);
        return v;
    }
    default: return 0;
    }
}

// the type's COMPARISON operator on two non-null cells: <0, 0, >0 (Long.compare / Integer.compare / Double.compare /
// Boolean.compare / Slice.compareTo = unsigned bytes, then length)
__device__ __forceinline__ int compare_cells(const TgColView &c, long long a, long long b)
{
    if (c.type == TGPU_VARCHAR) {
        const int oa = c.offsets[a], la = c.offsets[a + 1] - oa, ob = c.offsets[b], lb = c.offsets[b + 1] - ob;
        const unsigned char *pa = (const unsigned char *)c.values + oa, *pb = (const unsigned char *)c.values + ob;
        const int m = la < lb ? la : lb;
        for (int i = 0; i < m; i++)
            if (pa[i] != pb[i]) return pa[i] < pb[i] ? -1 : 1;
        return la < lb ? -1 : (la > lb ? 1 : 0);
    }
    const unsigned long long x = cell_order_bits(c, a), y = cell_order_bits(c, b);
    return x < y ? -1 : (x > y ? 1 : 0);
}

// SimplePageWithPositionComparator.compareTo over the sort keys
__device__ __forceinline__ int compare_rows(const TopNKeys &k, long long a, long long b)
{
    for (int i = 0; i < k.cols.n; i++) {
        const TgColView &c = k.cols.c[i];
        const bool na = c.nulls && c.nulls[a], nb = c.nulls && c.nulls[b];
        if (na || nb) {   // TypeOperators.orderNulls
            if (na && nb) continue;
            if (na) return nulls_first(k.order[i]) ? -1 : 1;
            return nulls_first(k.order[i]) ? 1 : -1;
        }
        const int cmp = compare_cells(c, a, b);
        if (cmp) return asc(k.order[i]) ? cmp : -cmp;
    }
    return 0;
}

__global__ void __launch_bounds__(kBlock) order_codes_kernel(const TopNKeys *kp, int64_t n, unsigned long long *codes, int *rows)
{
    const TopNKeys &k = *kp;
    const TgColView &c = k.cols.c[0];
    const int order = k.order[0];
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        const bool is_null = c.nulls && c.nulls[r];
        unsigned long long v = 0;
        if (!is_null) {
            v = cell_order_bits(c, r);
            if (!asc(order)) v = ~v;
        }
        // the null bit on top (nulls first: nulls get 0 and values 1), the value's upper 63 bits below it
        const unsigned long long top = (is_null == nulls_first(order)) ? 0ULL : 1ULL;
        codes[r] = (top << 63) | (is_null ? 0ULL : (v >> 1));
        rows[r] = (int)r;
    }
}

// count[0] = number of leading entries of the sorted codes that are <= codes[want - 1] (the candidates)
__global__ void candidate_count_kernel(const unsigned long long *sorted_codes, int64_t n, int64_t want, long long *count)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long cut = sorted_codes[want - 1];
    int64_t lo = want, hi = n;   // first index in [want, n) whose code is greater than the cut
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted_codes[mid] <= cut) lo = mid + 1;
        else hi = mid;
    }
    count[0] = lo;
}

// every stride-th code, for the cutoff estimate
__global__ void __launch_bounds__(kBlock) sample_codes_kernel(const unsigned long long *codes, int64_t n, int64_t stride, int64_t samples, unsigned long long *out)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < samples; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i * stride;
        out[i] = codes[r < n ? r : n - 1];
    }
}

__global__ void __launch_bounds__(kBlock) flag_candidates_kernel(const unsigned long long *codes, int64_t n, const unsigned long long *cutoff, int *flags)
{
    const unsigned long long cut = *cutoff;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) flags[r] = codes[r] <= cut ? 1 : 0;
}

__global__ void __launch_bounds__(kBlock) compact_candidates_kernel(const int *flags, const int *offsets, const unsigned long long *codes, int64_t n, int *rows_out,
                                                                     unsigned long long *codes_out)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock)
        if (flags[r]) {
            rows_out[offsets[r]] = (int)r;
            codes_out[offsets[r]] = codes[r];
        }
}

struct RowLess {
    const TopNKeys *k;
    __device__ bool operator()(const int &a, const int &b) const
    {
        const int cmp = compare_rows(*k, a, b);
        return cmp ? cmp < 0 : a < b;
    }
};

int grid_for(Context *ctx, int64_t n)
{
    int64_t blocks = ceil_div(n, kBlock);
    const int64_t cap = (int64_t)ctx->cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

TopNGpu::TopNGpu(Context *ctx, std::vector<int32_t> types, int64_t n, std::vector<int32_t> sort_channels, std::vector<int32_t> sort_orders)
    : ctx_(ctx), types_(std::move(types)), sort_channels_(std::move(sort_channels)), sort_orders_(std::move(sort_orders)), n_(n), kept_(ctx, types_)
{
    TG_CHECK_ARG(n_ >= 0, "n must be positive");
    TG_CHECK_ARG(!sort_channels_.empty() && sort_channels_.size() == sort_orders_.size(), "sort channels and sort orders differ in length");
    TG_CHECK_ARG((int)sort_channels_.size() <= kMaxKeyChannels, "at most 8 sort channels");
    for (size_t i = 0; i < sort_channels_.size(); i++) {
        TG_CHECK_ARG(sort_channels_[i] >= 0 && sort_channels_[i] < (int)types_.size(), "sort channel out of range");
        TG_CHECK_ARG(sort_orders_[i] >= TGPU_SORT_ASC_NULLS_FIRST && sort_orders_[i] <= TGPU_SORT_DESC_NULLS_LAST, "unknown sort order");
    }
    for (int32_t t : types_) TG_CHECK_ARG(valid_type(t), "unknown type");
}

BufferPtr TopNGpu::sorted_positions(Context *ctx_, const DevicePage &page, const std::vector<int32_t> &sort_channels_, const std::vector<int32_t> &sort_orders_,
                                    int64_t limit, int64_t &count)
{
    const int64_t n = page.n;
    const int64_t want = std::min<int64_t>(limit, n);
    count = want;
    if (want == 0) return ctx_->alloc(4);
    TG_CHECK_ARG(n <= 0x7fffffffLL, "page too large");
    TopNKeys host{};
    host.cols.n = (int32_t)sort_channels_.size();
    for (size_t i = 0; i < sort_channels_.size(); i++) {
        host.cols.c[i] = view_of(page.cols[(size_t)sort_channels_[i]]);
        host.order[i] = sort_orders_[i];
    }
    BufferPtr keys = ctx_->alloc(sizeof(TopNKeys));
    ctx_->upload(keys->ptr(), &host, sizeof(TopNKeys));
    constexpr int64_t kSamples = 1 << 16;
    BufferPtr codes = ctx_->alloc((size_t)n * 8), rows = ctx_->alloc((size_t)n * 4);
    BufferPtr scalar = ctx_->alloc(16);   // [0] candidate count, [1] cutoff code
    int64_t m = n;
    {
        ProfileScope ps(ctx_, "topn_select");
        order_codes_kernel<<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(keys->as<TopNKeys>(), n, codes->as<unsigned long long>(), rows->as<int>());
        check_launch("order_codes");
        if (n > 4 * kSamples && want < kSamples / 2) {
            // cutoff from a strided sample, then the rows at or below it, in input order
            const int64_t stride = n / kSamples;
            BufferPtr sample = ctx_->alloc((size_t)kSamples * 8), sample_sorted = ctx_->alloc((size_t)kSamples * 8);
            sample_codes_kernel<<<grid_for(ctx_, kSamples), kBlock, 0, ctx_->stream()>>>(codes->as<unsigned long long>(), n, stride, kSamples, sample->as<unsigned long long>());
            check_launch("sample_codes");
            size_t temp_bytes = 0;
            HIP_CHECK(rocprim::radix_sort_keys(nullptr, temp_bytes, sample->as<unsigned long long>(), sample_sorted->as<unsigned long long>(), (size_t)kSamples, 0, 64,
                                               ctx_->stream()));
            BufferPtr temp = ctx_->alloc(temp_bytes ? temp_bytes : 1);
            HIP_CHECK(rocprim::radix_sort_keys(temp->ptr(), temp_bytes, sample->as<unsigned long long>(), sample_sorted->as<unsigned long long>(), (size_t)kSamples, 0, 64,
                                               ctx_->stream()));
            BufferPtr flags = ctx_->alloc((size_t)n * 4), offsets = ctx_->alloc((size_t)n * 4);
            flag_candidates_kernel<<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(codes->as<unsigned long long>(), n, sample_sorted->as<unsigned long long>() + (want - 1),
                                                                                     flags->as<int>());
            check_launch("flag_candidates");
            k::exclusive_scan_i32(ctx_, flags->as<int32_t>(), offsets->as<int32_t>(), n, scalar->as<int64_t>());
            BufferPtr crow = ctx_->alloc((size_t)n * 4), ccode = ctx_->alloc((size_t)n * 8);
            compact_candidates_kernel<<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(flags->as<int>(), offsets->as<int>(), codes->as<unsigned long long>(), n, crow->as<int>(),
                                                                                        ccode->as<unsigned long long>());
            check_launch("compact_candidates");
            m = ctx_->read_scalar(scalar->as<long long>());
            TG_CHECK_STATE(m >= want && m <= n, "candidate count out of range");
            rows = crow;
            codes = ccode;
        }
        if (m > std::max<int64_t>(8 * want, 1 << 18)) {
            // still many candidates (coarse codes): order them by code and keep everything up to the ties of the n-th code
            BufferPtr codes_sorted = ctx_->alloc((size_t)m * 8), rows_sorted = ctx_->alloc((size_t)m * 4);
            size_t temp_bytes = 0;
            HIP_CHECK(rocprim::radix_sort_pairs(nullptr, temp_bytes, codes->as<unsigned long long>(), codes_sorted->as<unsigned long long>(), rows->as<int>(),
                                                rows_sorted->as<int>(), (size_t)m, 0, 64, ctx_->stream()));
            BufferPtr temp = ctx_->alloc(temp_bytes ? temp_bytes : 1);
            HIP_CHECK(rocprim::radix_sort_pairs(temp->ptr(), temp_bytes, codes->as<unsigned long long>(), codes_sorted->as<unsigned long long>(), rows->as<int>(),
                                                rows_sorted->as<int>(), (size_t)m, 0, 64, ctx_->stream()));
            candidate_count_kernel<<<1, 64, 0, ctx_->stream()>>>(codes_sorted->as<unsigned long long>(), m, want, scalar->as<long long>());
            check_launch("candidate_count");
            const int64_t m2 = ctx_->read_scalar(scalar->as<long long>());
            TG_CHECK_STATE(m2 >= want && m2 <= m, "candidate count out of range");
            m = m2;
            rows = rows_sorted;
        }
    }
    // the candidates, ordered by the full comparator
    BufferPtr sorted = ctx_->alloc((size_t)m * 4);
    {
        ProfileScope ps(ctx_, "topn_sort");
        RowLess less{keys->as<TopNKeys>()};
        size_t temp_bytes = 0;
        HIP_CHECK(rocprim::merge_sort(nullptr, temp_bytes, rows->as<int>(), sorted->as<int>(), (size_t)m, less, ctx_->stream()));
        BufferPtr temp = ctx_->alloc(temp_bytes ? temp_bytes : 1);
        HIP_CHECK(rocprim::merge_sort(temp->ptr(), temp_bytes, rows->as<int>(), sorted->as<int>(), (size_t)m, less, ctx_->stream()));
    }
    return sorted;
}

void TopNGpu::add_page(const DevicePage &page)
{
    TG_CHECK_ARG(page.cols.size() == types_.size(), "page channel count does not match the operator's types");
    for (size_t i = 0; i < types_.size(); i++) TG_CHECK_ARG(page.cols[i].type == types_[i], "page channel type does not match the operator's types");
    if (page.n == 0 || n_ == 0) return;
    int64_t count = 0;
    BufferPtr pos = top_positions(page, count);
    DevicePage winners;
    winners.n = count;
    {
        ProfileScope ps(ctx_, "topn_gather");
        for (auto &c : page.cols) winners.cols.push_back(k::gather_column(ctx_, c, pos->as<int32_t>(), count, false));
    }
    sorted_ = kept_.position_count() == 0;   // a single page's winners are already the answer, in order
    kept_.add_page(winners);
    // many small pages: fold the store back to n rows now and then
    if (kept_.position_count() > std::max<int64_t>(4 * n_, 1 << 16)) {
        DevicePage folded = result();
        PagesIndexGpu fresh(ctx_, types_);
        fresh.add_page(folded);
        kept_ = std::move(fresh);
        sorted_ = true;
    }
}

DevicePage TopNGpu::result()
{
    DevicePage all;
    all.n = kept_.position_count();
    for (size_t i = 0; i < types_.size(); i++) all.cols.push_back(kept_.column((int)i));
    DevicePage out;
    int64_t count = 0;
    if (all.n == 0) {
        out.n = 0;
        for (size_t i = 0; i < types_.size(); i++) out.cols.push_back(kept_.column((int)i));
        return out;
    }
    if (sorted_) return all;
    BufferPtr pos = top_positions(all, count);
    out.n = count;
    ProfileScope ps(ctx_, "topn_gather");
    for (auto &c : all.cols) out.cols.push_back(k::gather_column(ctx_, c, pos->as<int32_t>(), count, false));
    return out;
}

}  // namespace tgpu
