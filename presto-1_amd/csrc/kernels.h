// kernels.h -- host-side launchers of the AOT-compiled gfx950 kernels (one .hip file per family).
#pragma once

#include "common.h"

namespace tgpu {
namespace k {

// ---- basic.hip ------------------------------------------------------------------------------------------------------
// exclusive prefix sum of int32 flags/counts; *total_dev (device int64) receives the grand total
void exclusive_scan_i32(Context *ctx, const int32_t *in, int32_t *out, int64_t n, int64_t *total_dev);
// K3: raw hash of the key columns of every row (InterpretedHashGenerator / JoinCompiler hashRow)
void hash_rows(Context *ctx, const KeyCols &keys, int64_t n, int64_t *out);
// out[i] = src[positions[i]] for every column type; positions[i] < 0 yields a null row (outer joins)
DeviceColumn gather_column(Context *ctx, const DeviceColumn &src, const int32_t *positions, int64_t n_out, bool negative_is_null);
// zero-copy region view [offset, offset+len)
DeviceColumn region_of(Context *ctx, const DeviceColumn &src, int64_t offset, int64_t len);
// appends `src` (device) to a growing column store; used by PagesIndex and the group-by key store
void fill_i32(Context *ctx, int32_t *p, int32_t v, int64_t n);
void fill_u64(Context *ctx, uint64_t *p, uint64_t v, int64_t n);
void iota_i32(Context *ctx, int32_t *p, int64_t n);
void widen_i32_to_i64(Context *ctx, const int32_t *in, int64_t *out, int64_t n);
// any-null over the key columns: out[i] = 1 if any key cell of row i is null
void any_null(Context *ctx, const KeyCols &keys, int64_t n, uint8_t *out);
// K10: partition id per row = (raw & 0x7fff...) % partitions (HashGenerator, remote exchanges), or with `local` the
// LocalPartitionGenerator function (int) xxh64(reverse(raw)) & (partitions - 1) of local exchanges
void partition_ids(Context *ctx, const int64_t *raw_hashes, int64_t n, int32_t partitions, int32_t *out, bool local = false);
// ---- partition.hip --------------------------------------------------------------------------------------------------
// stable grouping of row indices by partition id: positions (n) grouped by partition in input order, counts[partitions] (device int64)
void partition_positions(Context *ctx, const int32_t *part_ids, int64_t n, int32_t partitions, int32_t *positions_out, int64_t *counts_dev);
// the same with replicated rows (PagePartitioner.partitionPage: a row whose `replicate` byte is set goes to EVERY partition):
// (partition, position) pairs grouped by partition, positions ascending inside each; pairs_out = n + replicated rows x (partitions - 1)
void partition_pairs(Context *ctx, const int32_t *part_ids, const uint8_t *replicate, int64_t n, int32_t partitions, BufferPtr &positions_out, int64_t &pairs_out,
                     int64_t *counts_dev);

}  // namespace k
}  // namespace tgpu
