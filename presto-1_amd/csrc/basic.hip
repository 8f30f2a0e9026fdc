// basic.hip -- scan, raw-hash (K3), gather (K9), partition (K10) kernels for gfx950.
// All of these are HBM-bound byte/integer kernels: coalesced row-per-lane access, 64-wide waves,
// grids capped at a few blocks per CU with grid-stride loops.
#include "kernels.h"
#include "device_hash.h"

namespace tgpu {
namespace k {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

static inline int grid_for(Context *ctx, int64_t n, int per_block = kBlock)
{
    int64_t blocks = ceil_div(n, per_block);
    int64_t cap = (int64_t)ctx->cu_count() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---------------------------------------------------------------------------------------------------------------------
// scan: three-level (tile reduce -> scan of tile sums -> tile scan).  Tile = 256 threads x 8 items.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;

template <typename T> __device__ __forceinline__ T wave_inclusive_scan(T v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across the block; returns exclusive prefix, *total = block total
template <typename T> __device__ __forceinline__ T block_exclusive_scan(T v, T *total, T *lds /* kWaves + 1 */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = wave_inclusive_scan(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) { T t = lds[w]; lds[w] = run; run += t; }
        lds[kWaves] = run;
    }
    __syncthreads();
    T prefix = lds[wave] + inc - v;
    *total = lds[kWaves];
    __syncthreads();
    return prefix;
}

// the 8 consecutive items of a thread: two 16-byte loads when the array is 16-byte aligned (it is for every buffer of the context's
// allocator; callers may pass interior pointers), else eight 4-byte ones
__device__ __forceinline__ void scan_load8(const int32_t *__restrict__ in, int64_t idx, int64_t n, bool aligned, int32_t v[kScanItems])
{
    if (aligned && idx + kScanItems <= n) {
        const int4 a = *reinterpret_cast<const int4 *>(in + idx), b = *reinterpret_cast<const int4 *>(in + idx + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        return;
    }
#pragma unroll
    for (int i = 0; i < kScanItems; i++) v[i] = idx + i < n ? in[idx + i] : 0;
}

// Every workgroup owns a contiguous RANGE of tiles (range = a multiple of the tile, at most kScanMaxBlocks ranges): the scan of the range
// sums in the middle is then a few hundred values whatever n is (one tile per workgroup made it 48 828 values for 100 M elements, scanned
// by a single workgroup in 191 rounds: 0.4 ms of the join's 0.74 ms probe scan).
constexpr int kScanMaxBlocks = 2048;

__global__ void __launch_bounds__(kBlock) scan_range_sums_i32(const int32_t *__restrict__ in, int64_t n, int64_t range, int64_t *__restrict__ range_sums)
{
    __shared__ int64_t lds[kWaves + 1];
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const int64_t a = (int64_t)blockIdx.x * range, z = a + range < n ? a + range : n;
    int64_t s = 0;
    for (int64_t base = a; base < z; base += kScanTile) {
        int32_t v[kScanItems];
        scan_load8(in, base + (int64_t)threadIdx.x * kScanItems, z, aligned, v);
#pragma unroll
        for (int i = 0; i < kScanItems; i++) s += v[i];
    }
    int64_t total;
    block_exclusive_scan<int64_t>(s, &total, lds);
    if (threadIdx.x == 0) range_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock) scan_ranges_i32(const int32_t *__restrict__ in, int32_t *__restrict__ out, int64_t n, int64_t range,
                                                           const int64_t *__restrict__ range_prefix)
{
    __shared__ int64_t lds[kWaves + 1];
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15) == 0, out_aligned = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const int64_t a = (int64_t)blockIdx.x * range, z = a + range < n ? a + range : n;
    int64_t carry = range_prefix[blockIdx.x];
    for (int64_t base = a; base < z; base += kScanTile) {
        const int64_t idx = base + (int64_t)threadIdx.x * kScanItems;
        int32_t v[kScanItems];
        scan_load8(in, idx, z, aligned, v);
        int64_t s = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; i++) s += v[i];
        int64_t total;
        int64_t prefix = carry + block_exclusive_scan<int64_t>(s, &total, lds);
        int32_t o[kScanItems];
#pragma unroll
        for (int i = 0; i < kScanItems; i++) {
            o[i] = (int32_t)prefix;
            prefix += v[i];
        }
        if (out_aligned && idx + kScanItems <= z) {
            *reinterpret_cast<int4 *>(out + idx) = make_int4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<int4 *>(out + idx + 4) = make_int4(o[4], o[5], o[6], o[7]);
        } else {
#pragma unroll
            for (int i = 0; i < kScanItems; i++)
                if (idx + i < z) out[idx + i] = o[i];
        }
        carry += total;
    }
}

// single-block exclusive scan of int64 (tile sums); n may exceed the block: loops with a carried prefix
__global__ void __launch_bounds__(kBlock) scan_small_i64(const int64_t *__restrict__ in, int64_t *__restrict__ out, int64_t n, int64_t *total_out)
{
    __shared__ int64_t lds[kWaves + 1];
    int64_t carry = 0;
    for (int64_t base = 0; base < n; base += kBlock) {
        int64_t idx = base + threadIdx.x;
        int64_t v = idx < n ? in[idx] : 0;
        int64_t total;
        int64_t p = block_exclusive_scan<int64_t>(v, &total, lds);
        if (idx < n) out[idx] = carry + p;
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// one tile: sums, prefix and scan in a single launch (the per-page scans of the join's chunk counts are a few hundred elements;
// three launches cost 11 us there, several times the kernels' work)
__global__ void __launch_bounds__(kBlock) scan_one_tile_i32(const int32_t *__restrict__ in, int32_t *__restrict__ out, int64_t n, int64_t *total_out)
{
    __shared__ int64_t lds[kWaves + 1];
    const int64_t base = (int64_t)threadIdx.x * kScanItems;
    int32_t v[kScanItems];
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    int64_t total;
    int64_t prefix = block_exclusive_scan<int64_t>(s, &total, lds);
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        if (base + i < n) out[base + i] = (int32_t)prefix;
        prefix += v[i];
    }
    if (threadIdx.x == 0 && total_out) *total_out = total;
}

void exclusive_scan_i32(Context *ctx, const int32_t *in, int32_t *out, int64_t n, int64_t *total_dev)
{
    if (n <= 0) {
        if (total_dev) HIP_CHECK(hipMemsetAsync(total_dev, 0, 8, ctx->stream()));
        return;
    }
    if (n <= kScanTile) {
        scan_one_tile_i32<<<1, kBlock, 0, ctx->stream()>>>(in, out, n, total_dev);
        check_launch("exclusive_scan_i32");
        return;
    }
    const int64_t tiles = ceil_div(n, kScanTile);
    const int64_t tiles_per_range = ceil_div(tiles, (int64_t)kScanMaxBlocks), range = tiles_per_range * kScanTile, ranges = ceil_div(n, range);
    BufferPtr sums = ctx->alloc((size_t)ranges * 8);
    BufferPtr prefix = ctx->alloc((size_t)ranges * 8);
    scan_range_sums_i32<<<(int)ranges, kBlock, 0, ctx->stream()>>>(in, n, range, sums->as<int64_t>());
    scan_small_i64<<<1, kBlock, 0, ctx->stream()>>>(sums->as<int64_t>(), prefix->as<int64_t>(), ranges, total_dev);
    scan_ranges_i32<<<(int)ranges, kBlock, 0, ctx->stream()>>>(in, out, n, range, prefix->as<int64_t>());
    check_launch("exclusive_scan_i32");
}

// ---------------------------------------------------------------------------------------------------------------------
// K3 raw hash: h = 31*h + (isNull ? 0 : typeHash(cell)), start 0 (M/operator/InterpretedHashGenerator.java:56-70)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) hash_rows_kernel(KeyCols keys, int64_t n, int64_t *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        out[r] = tg_hash_row_u(keys, r);
    }
}

void hash_rows(Context *ctx, const KeyCols &keys, int64_t n, int64_t *out)
{
    if (n <= 0) return;
    ProfileScope ps(ctx, "hash_rows");
    hash_rows_kernel<<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(keys, n, out);
    check_launch("hash_rows");
}

__global__ void __launch_bounds__(kBlock) any_null_kernel(KeyCols keys, int64_t n, uint8_t *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        uint8_t f = 0;
#pragma unroll
        for (int c = 0; c < TG_MAX_KEY_CHANNELS; c++) {
            if (c >= keys.n) break;
            f |= (keys.c[c].nulls && keys.c[c].nulls[r]) ? 1 : 0;
        }
        out[r] = f;
    }
}

void any_null(Context *ctx, const KeyCols &keys, int64_t n, uint8_t *out)
{
    if (n <= 0) return;
    any_null_kernel<<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(keys, n, out);
    check_launch("any_null");
}

// ---------------------------------------------------------------------------------------------------------------------
// fills
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> __global__ void __launch_bounds__(kBlock) fill_kernel(T *p, T v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}
__global__ void __launch_bounds__(kBlock) iota_kernel(int32_t *p, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = (int32_t)i;
}
__global__ void __launch_bounds__(kBlock) widen_kernel(const int32_t *in, int64_t *out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = in[i];
}

void fill_i32(Context *ctx, int32_t *p, int32_t v, int64_t n)
{
    if (n <= 0) return;
    fill_kernel<int32_t><<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(p, v, n);
    check_launch("fill_i32");
}
void fill_u64(Context *ctx, uint64_t *p, uint64_t v, int64_t n)
{
    if (n <= 0) return;
    fill_kernel<uint64_t><<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(p, v, n);
    check_launch("fill_u64");
}
void iota_i32(Context *ctx, int32_t *p, int64_t n)
{
    if (n <= 0) return;
    iota_kernel<<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(p, n);
    check_launch("iota_i32");
}
void widen_i32_to_i64(Context *ctx, const int32_t *in, int64_t *out, int64_t n)
{
    if (n <= 0) return;
    widen_kernel<<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(in, out, n);
    check_launch("widen");
}

// ---------------------------------------------------------------------------------------------------------------------
// K9 gather: out[i] = src[pos[i]] (pos < 0 -> null).  One row per lane; output stores are coalesced, source reads
// are as random as the positions are (sequential for a filter's ascending position list).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock) gather_fixed_kernel(const T *__restrict__ src, const uint8_t *__restrict__ src_nulls,
                                                               const int32_t *__restrict__ pos, int64_t n_out,
                                                               T *__restrict__ out, uint8_t *__restrict__ out_nulls)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * kBlock) {
        int32_t p = pos[i];
        T v = T(0);
        uint8_t isnull = 1;
        if (p >= 0) {
            isnull = src_nulls ? src_nulls[p] : 0;
            v = src[p];
        }
        out[i] = v;
        if (out_nulls) out_nulls[i] = isnull;
    }
}

__global__ void __launch_bounds__(kBlock) gather_len_kernel(const int32_t *__restrict__ offsets, const int32_t *__restrict__ pos,
                                                             int64_t n_out, int32_t *__restrict__ len)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * kBlock) {
        int32_t p = pos[i];
        len[i] = p >= 0 ? offsets[p + 1] - offsets[p] : 0;
    }
}

__global__ void __launch_bounds__(kBlock) gather_bytes_kernel(const uint8_t *__restrict__ pool, const int32_t *__restrict__ offsets,
                                                               const uint8_t *__restrict__ src_nulls, const int32_t *__restrict__ pos,
                                                               int64_t n_out, const int32_t *__restrict__ out_offsets,
                                                               const int64_t *__restrict__ total, int32_t *__restrict__ out_offsets_end,
                                                               uint8_t *__restrict__ out_pool, uint8_t *__restrict__ out_nulls)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * kBlock) {
        int32_t p = pos[i];
        uint8_t isnull = 1;
        if (p >= 0) {
            isnull = src_nulls ? src_nulls[p] : 0;
            int32_t a = offsets[p], b = offsets[p + 1], o = out_offsets[i];
            for (int32_t k = a; k < b; k++) out_pool[o + (k - a)] = pool[k];
        }
        if (out_nulls) out_nulls[i] = isnull;
        if (i == n_out - 1) *out_offsets_end = (int32_t)(*total);
    }
}

DeviceColumn gather_column(Context *ctx, const DeviceColumn &src, const int32_t *positions, int64_t n_out, bool negative_is_null)
{
    DeviceColumn out;
    out.type = src.type;
    out.n = n_out;
    const bool need_nulls = src.nulls != nullptr || negative_is_null;
    if (need_nulls) {
        out.nulls_buf = ctx->alloc((size_t)(n_out > 0 ? n_out : 1));
        out.nulls = out.nulls_buf->as<uint8_t>();
    }
    uint8_t *out_nulls = need_nulls ? out.nulls_buf->as<uint8_t>() : nullptr;
    ProfileScope ps(ctx, "gather");
    if (src.type == TGPU_VARCHAR) {
        out.offsets_buf = ctx->alloc((size_t)(n_out + 1) * 4);
        out.offsets = out.offsets_buf->as<int32_t>();
        if (n_out == 0) {
            HIP_CHECK(hipMemsetAsync(out.offsets_buf->ptr(), 0, 4, ctx->stream()));
            out.values_buf = ctx->alloc(1);
            out.values = out.values_buf->ptr();
            out.pool_bytes = 0;
            return out;
        }
        BufferPtr len = ctx->alloc((size_t)n_out * 4);
        BufferPtr total = ctx->alloc(8);
        int g = grid_for(ctx, n_out);
        gather_len_kernel<<<g, kBlock, 0, ctx->stream()>>>(src.offsets, positions, n_out, len->as<int32_t>());
        exclusive_scan_i32(ctx, len->as<int32_t>(), out.offsets_buf->as<int32_t>(), n_out, total->as<int64_t>());
        int64_t total_bytes = ctx->read_scalar(total->as<int64_t>());
        if (total_bytes > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "variable width block cannot exceed 2GB");
        out.values_buf = ctx->alloc((size_t)(total_bytes > 0 ? total_bytes : 1));
        out.values = out.values_buf->ptr();
        out.pool_bytes = total_bytes;
        gather_bytes_kernel<<<g, kBlock, 0, ctx->stream()>>>((const uint8_t *)src.values, src.offsets, src.nulls, positions, n_out,
                                                              out.offsets_buf->as<int32_t>(), total->as<int64_t>(),
                                                              out.offsets_buf->as<int32_t>() + n_out, out.values_buf->as<uint8_t>(), out_nulls);
        check_launch("gather_varchar");
        return out;
    }
    const int w = type_width(src.type);
    out.values_buf = ctx->alloc((size_t)(n_out > 0 ? n_out : 1) * w);
    out.values = out.values_buf->ptr();
    if (n_out == 0) return out;
    int g = grid_for(ctx, n_out);
    switch (w) {
    case 8:
        gather_fixed_kernel<int64_t><<<g, kBlock, 0, ctx->stream()>>>((const int64_t *)src.values, src.nulls, positions, n_out,
                                                                        out.values_buf->as<int64_t>(), out_nulls);
        break;
    case 4:
        gather_fixed_kernel<int32_t><<<g, kBlock, 0, ctx->stream()>>>((const int32_t *)src.values, src.nulls, positions, n_out,
                                                                        out.values_buf->as<int32_t>(), out_nulls);
        break;
    default:
        gather_fixed_kernel<uint8_t><<<g, kBlock, 0, ctx->stream()>>>((const uint8_t *)src.values, src.nulls, positions, n_out,
                                                                        out.values_buf->as<uint8_t>(), out_nulls);
        break;
    }
    check_launch("gather_fixed");
    return out;
}

DeviceColumn region_of(Context *ctx, const DeviceColumn &src, int64_t offset, int64_t len)
{
    (void)ctx;
    TG_CHECK_ARG(offset >= 0 && len >= 0 && offset + len <= src.n, "region out of range");
    DeviceColumn out = src;  // shares the owners
    out.n = len;
    if (src.type == TGPU_VARCHAR) {
        // offsets keep their absolute values into the shared pool (like VariableWidthBlock.getRegion)
        out.offsets = src.offsets + offset;
        out.pool_exact = src.pool_exact && offset == 0 && len == src.n;
    }
    else {
        out.values = (const uint8_t *)src.values + offset * type_width(src.type);
    }
    if (src.nulls) out.nulls = src.nulls + offset;
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// K10 partition: partition id per row, then a stable grouping of row positions by partition
// (M/operator/PartitionedOutputOperator.java:406-426 appends rows to per-partition builders in input order)
// ---------------------------------------------------------------------------------------------------------------------
template <bool LOCAL>
__global__ void __launch_bounds__(kBlock) partition_ids_kernel(const int64_t *__restrict__ raw, int64_t n, int32_t parts, int32_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = LOCAL ? tg_partition_local(raw[i], parts) : tg_partition_remote(raw[i], parts);
}

void partition_ids(Context *ctx, const int64_t *raw_hashes, int64_t n, int32_t partitions, int32_t *out, bool local)
{
    if (local) TG_CHECK_ARG((partitions & (partitions - 1)) == 0, "the local partition function needs a power-of-two partition count");
    if (n <= 0) return;
    ProfileScope ps(ctx, "partition_ids");
    if (local) partition_ids_kernel<true><<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(raw_hashes, n, partitions, out);
    else partition_ids_kernel<false><<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(raw_hashes, n, partitions, out);
    check_launch("partition_ids");
}

}  // namespace k
}  // namespace tgpu
