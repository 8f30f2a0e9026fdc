// partition.hip -- K10, the grouping half: row positions grouped by destination partition, input order kept inside each
// (M/operator/PartitionedOutputOperator.java:406-426 appends rows to per-partition page builders in input order).
//
// One stable LSD radix sort of (partition id, position) pairs over the bits the partition ids use (rocPRIM onesweep: one pass
// per 8 bits, so one pass up to 256 partitions, two up to 65536) -- the cost does not grow with the partition count, unlike a
// pass per partition.  Rows that go to every partition (replicated rows: null partitioning keys, the "replicate any row"
// flag) are expanded into one pair per partition first.
#include "kernels.h"

#include <cstring>

#include <rocprim/rocprim.hpp>

namespace tgpu {
namespace k {
namespace {

constexpr int kBlock = 256;

int grid_of(Context *ctx, int64_t n)
{
    int64_t blocks = ceil_div(n, kBlock);
    const int64_t cap = (int64_t)ctx->cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

__global__ void __launch_bounds__(kBlock) pair_keys_kernel(const int32_t *__restrict__ ids, int64_t n, unsigned int *__restrict__ keys, int32_t *__restrict__ positions)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        keys[i] = (unsigned int)ids[i];
        positions[i] = (int32_t)i;
    }
}

__global__ void __launch_bounds__(kBlock) pair_multiplicity_kernel(const uint8_t *__restrict__ replicate, int64_t n, int32_t parts, int32_t *__restrict__ mult)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) mult[i] = replicate[i] ? parts : 1;
}

__global__ void __launch_bounds__(kBlock) expand_pairs_kernel(const int32_t *__restrict__ ids, const uint8_t *__restrict__ replicate, const int32_t *__restrict__ first,
                                                               int64_t n, int32_t parts, unsigned int *__restrict__ keys, int32_t *__restrict__ positions)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t at = first[i];
        if (replicate[i]) {
            for (int32_t p = 0; p < parts; p++) {
                keys[at + p] = (unsigned int)p;
                positions[at + p] = (int32_t)i;
            }
        }
        else {
            keys[at] = (unsigned int)ids[i];
            positions[at] = (int32_t)i;
        }
    }
}

// pairs per partition: block-private LDS histogram (partitions <= 1024), one global atomic per block and partition
__global__ void __launch_bounds__(kBlock) partition_histogram_kernel(const unsigned int *__restrict__ keys, int64_t m, int32_t parts, unsigned long long *__restrict__ counts)
{
    __shared__ unsigned int h[1024];
    for (int p = threadIdx.x; p < parts; p += kBlock) h[p] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) atomicAdd(&h[keys[i]], 1u);
    __syncthreads();
    for (int p = threadIdx.x; p < parts; p += kBlock)
        if (h[p]) atomicAdd(&counts[p], (unsigned long long)h[p]);
}

// stable sort of the pairs by key + the per-partition counts
void sort_pairs(Context *ctx, BufferPtr &keys, BufferPtr &positions, int64_t m, int32_t parts, int32_t *positions_out, int64_t *counts_dev)
{
    unsigned int end_bit = 1;
    while (end_bit < 32 && (1ll << end_bit) < (long long)parts) end_bit++;
    BufferPtr keys_out = ctx->alloc((size_t)m * 4);
    size_t temp_bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys->as<unsigned int>(), keys_out->as<unsigned int>(), positions->as<int>(), positions_out, (size_t)m, 0, end_bit,
                                        ctx->stream()));
    BufferPtr temp = ctx->alloc(temp_bytes > 0 ? temp_bytes : 1);
    HIP_CHECK(rocprim::radix_sort_pairs(temp->ptr(), temp_bytes, keys->as<unsigned int>(), keys_out->as<unsigned int>(), positions->as<int>(), positions_out, (size_t)m, 0, end_bit,
                                        ctx->stream()));
    partition_histogram_kernel<<<std::min(grid_of(ctx, m), ctx->cu_count() * 2), kBlock, 0, ctx->stream()>>>(keys->as<unsigned int>(), m, parts, (unsigned long long *)counts_dev);
    check_launch("partition_histogram");
}

}  // namespace

void partition_positions(Context *ctx, const int32_t *part_ids, int64_t n, int32_t partitions, int32_t *positions_out, int64_t *counts_dev)
{
    TG_CHECK_ARG(partitions > 0 && partitions <= 1024, "partition count must be in 1..1024");
    HIP_CHECK(hipMemsetAsync(counts_dev, 0, (size_t)partitions * 8, ctx->stream()));
    if (n <= 0) return;
    ProfileScope ps(ctx, "partition_positions");
    BufferPtr keys = ctx->alloc((size_t)n * 4), positions = ctx->alloc((size_t)n * 4);
    pair_keys_kernel<<<grid_of(ctx, n), kBlock, 0, ctx->stream()>>>(part_ids, n, keys->as<unsigned int>(), positions->as<int32_t>());
    check_launch("pair_keys");
    sort_pairs(ctx, keys, positions, n, partitions, positions_out, counts_dev);
}

void partition_pairs(Context *ctx, const int32_t *part_ids, const uint8_t *replicate, int64_t n, int32_t partitions, BufferPtr &positions_out, int64_t &pairs_out,
                     int64_t *counts_dev)
{
    TG_CHECK_ARG(partitions > 0 && partitions <= 1024, "partition count must be in 1..1024");
    pairs_out = 0;
    HIP_CHECK(hipMemsetAsync(counts_dev, 0, (size_t)partitions * 8, ctx->stream()));
    if (n <= 0) {
        positions_out = ctx->alloc(4);
        return;
    }
    if (replicate == nullptr) {
        positions_out = ctx->alloc((size_t)n * 4);
        partition_positions(ctx, part_ids, n, partitions, positions_out->as<int32_t>(), counts_dev);
        pairs_out = n;
        return;
    }
    ProfileScope ps(ctx, "partition_positions");
    BufferPtr mult = ctx->alloc((size_t)n * 4), first = ctx->alloc((size_t)n * 4), total = ctx->alloc(8);
    pair_multiplicity_kernel<<<grid_of(ctx, n), kBlock, 0, ctx->stream()>>>(replicate, n, partitions, mult->as<int32_t>());
    check_launch("pair_multiplicity");
    exclusive_scan_i32(ctx, mult->as<int32_t>(), first->as<int32_t>(), n, total->as<int64_t>());
    const int64_t m = ctx->read_scalar(total->as<int64_t>());
    if (m > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "replicated rows x partitions exceed 2 billion output rows for one page");
    BufferPtr keys = ctx->alloc((size_t)m * 4), positions = ctx->alloc((size_t)m * 4);
    expand_pairs_kernel<<<grid_of(ctx, n), kBlock, 0, ctx->stream()>>>(part_ids, replicate, first->as<int32_t>(), n, partitions, keys->as<unsigned int>(),
                                                                        positions->as<int32_t>());
    check_launch("expand_pairs");
    positions_out = ctx->alloc((size_t)m * 4);
    sort_pairs(ctx, keys, positions, m, partitions, positions_out->as<int32_t>(), counts_dev);
    pairs_out = m;
}

}  // namespace k
}  // namespace tgpu
