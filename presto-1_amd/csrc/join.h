// join.h -- hash join on the GPU: PagesIndex (J2), PagesHash build (J5/J6, K7) and probe (J8/J9, K8) + gather (K9).
#pragma once

#include <array>
#include <mutex>

#include "common.h"

namespace tgpu {

// M/operator/PagesIndex.java:209-240: the build side's pages appended into contiguous per-channel HBM columns.
// SyntheticAddress(pageIndex, position) collapses to the flat row number because pages are stored back to back.
class PagesIndexGpu {
public:
    PagesIndexGpu(Context *ctx, std::vector<int32_t> types);
    // `varchar_ends` (optional): {offsets[0], offsets[n]} of every channel (only read for VARCHAR channels), when the caller already
    // knows them -- otherwise they are read back (one stream synchronisation per VARCHAR channel whose pool is not known exactly)
    void add_page(const DevicePage &page, const std::vector<std::array<int32_t, 2>> *varchar_ends = nullptr);
    int64_t position_count() const { return n_; }
    int64_t estimated_size() const;
    DeviceColumn column(int ch) const;
    const std::vector<int32_t> &types() const { return types_; }

private:
    struct Store {
        int32_t type;
        BufferPtr values, nulls, offsets;
        int64_t cap = 0, pool_cap = 0, pool_used = 0;
        bool has_nulls = false;
    };
    void reserve(int64_t rows);
    Context *ctx_;
    std::vector<int32_t> types_;
    std::vector<Store> cols_;
    int64_t n_ = 0;
};

// device view of the int-key fast path table, for the JIT-fused filter+project+probe kernels
struct IntTableView {
    const void *slots = nullptr;            // TgSlot16[mask + 1]
    uint64_t mask = 0;
    const unsigned long long *bloom = nullptr;
    unsigned long long bloom_word_mask = 0;
    const unsigned long long *bitmap = nullptr;   // dense key domain: one bit per key value in [key_min, key_max]
    long long key_min = 0, key_max = -1;
    const int32_t *direct = nullptr;        // DIRECT layout: build positions in key order (slots is then nullptr)
    const int32_t *rank_base = nullptr;     //   and the number of present keys in front of every bitmap word
    const int32_t *links = nullptr;         // nullptr = no duplicate build keys
    int32_t key_type = 0;
};

// The built lookup source: PagesHash + ArrayPositionLinks behind JoinHash (M/operator/JoinHash.java:44-125).
class LookupSourceGpu {
public:
    bool int_table(IntTableView &v) const;
    LookupSourceGpu(Context *ctx, std::shared_ptr<PagesIndexGpu> index, std::vector<int32_t> key_channels, int32_t hash_channel,
                    std::vector<int32_t> output_channels);
    void build();  // PagesHash constructor (M/operator/PagesHash.java:53-125)

    int64_t position_count() const { return n_; }
    int64_t hash_size() const { return capacity_; }
    int64_t link_count() const { return link_count_; }
    int64_t estimated_size() const;
    const std::vector<int32_t> &output_channels() const { return output_channels_; }
    std::vector<int32_t> output_types() const;
    std::vector<int32_t> key_types() const;

    // probe one page: (probe position, build position) pairs in the reference's output order (probe positions ascending,
    // each probe row's matches newest build position first); build position -1 = PROBE_OUTER row without a match.
    void probe(const std::vector<const DeviceColumn *> &probe_keys, const int64_t *probe_hashes, int64_t n_probe, bool probe_outer,
               BufferPtr &out_probe_idx, BufferPtr &out_build_idx, int64_t &out_count);
    // the build side's output channel out_idx as stored in the index (all build rows)
    DeviceColumn build_column(int out_idx) const;
    DeviceColumn gather_build(int out_idx, const int32_t *build_positions, int64_t n, bool negative_is_null) const;
    // any channel of the build side (a join filter function may read channels that are not output channels)
    DeviceColumn gather_index_channel(int channel, const int32_t *build_positions, int64_t n) const;
    // OuterPositionTracker (M/operator/OuterLookupSource.java:146-190): LOOKUP_OUTER / FULL_OUTER probes record the build positions
    // they emitted; unvisited_positions = OuterPositionIterator, every build position nobody matched, ascending
    void mark_visited(const int32_t *build_positions, int64_t n);
    // DIRECT layout: the rank structure (rank_base + positions in key order) is built on first demand when the table has no output
    // channels; everything that reads build positions calls this first
    void ensure_rank() const;
    void unvisited_positions(BufferPtr &positions, int64_t &count);

private:
    bool build_direct(const KeyCols &keys);
    Context *ctx_;
    std::shared_ptr<PagesIndexGpu> index_;
    std::vector<int32_t> key_channels_, output_channels_;
    int32_t hash_channel_;
    int64_t n_ = 0, capacity_ = 0, link_count_ = 0;
    bool int_key_fast_ = false;  // single BIGINT / INTEGER / DATE key: key stored inline in the slot
    BufferPtr visited_;          // uint8[n]: build positions matched by an outer-tracking probe (allocated on first use)
    mutable std::mutex visited_mu_;   //   (several probe operators, possibly on different driver threads, share the table)
    mutable bool rank_pending_ = false;   // DIRECT layout: rank_base_ / direct_ are allocated but not filled yet (ensure_rank)
    ColView direct_key_{};
    BufferPtr heads_;            // int32[capacity], -1 empty          (PagesHash.key)
    BufferPtr slots16_;          // fast path: {int64 key, int32 head, int32 pad}[capacity]
    BufferPtr bloom_;            // fast path: blocked Bloom filter over the build keys (sparse key domains)
    int64_t bloom_words_ = 0;
    BufferPtr bitmap_;           // fast path: exact bitmap over [key_min_, key_max_] (dense key domains)
    BufferPtr direct_;           // fast path, DIRECT layout: int32 build positions in key order (no hash table)
    BufferPtr rank_base_;        //   present keys in front of every bitmap word (device_join.h tg_direct_position)
    long long key_min_ = 0, key_max_ = -1;
    BufferPtr tags_;             // uint8[n]                           (PagesHash.positionToHashes)
    BufferPtr links_;            // int32[n] or empty when no duplicates (ArrayPositionLinks)
    std::vector<DeviceColumn> key_cols_;
};

}  // namespace tgpu
