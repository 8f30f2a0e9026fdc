// groupby.h -- GroupByHash in HBM (K4/K5): open-address table probed by one lane per row, group ids assigned in
// first-seen order exactly as M/operator/BigintGroupByHash.java:213-260 and MultiChannelGroupByHash.java:281-342 do.
#pragma once

#include "common.h"

#include <functional>

namespace tgpu {

// one probe/insert launch of a sub-batch, for probe kernels compiled outside groupby.hip (jit.cpp: key-schema specialised,
// filter fused in front).  Rows are numbered from 0 inside the sub-batch; row0 is the sub-batch's offset in the page.
struct GbhProbeLaunch {
    int64_t row0, n;
    uint64_t *words;
    uint64_t mask;
    KeyCols store;
    int32_t store_groups;
    int32_t *out;
    unsigned long long *counters;   // [0] pending rows, [2] table-full error, [7] the probe kernel's expression-error word (~0 = none):
                                    // it comes back with the other counters, one read-back for all of them
    // compact mode (out == nullptr): one byte per row = group id + 1, 0 = row rejected by the fused filter, 255 = the row joined /
    // created a group that is new in this sub-batch (counted in counters[0]; the sub-batch is then re-run in int32 mode)
    uint8_t *out8 = nullptr;
};
using GbhProbeFn = std::function<void(const GbhProbeLaunch &)>;
// Speculation hook of the compact mode: called between the probe launch of a sub-batch that covers the WHOLE page and the wait for its
// counters, with the device address of those counters.  The caller enqueues the page's consumer there (the fused accumulate kernel),
// gated on the counters being clean ([0] == 0 no row met a new group, [2] == 0 no table overflow, [7] == ~0 no expression error): in the
// steady state of low-cardinality inputs the wait for the counters then overlaps the consumer instead of idling the device.
using GbhSpeculateFn = std::function<void(const unsigned long long *counters)>;

// The single-integer-key table (groupby_bigint.hip): key inline in a 16-byte slot, one random line per row, one atomic per new group.
// Used by GroupByHashGpu for one BIGINT key (as GroupByHash.createGroupByHash picks BigintGroupByHash, M/operator/GroupByHash.java:45-59)
// and for one INTEGER / DATE key without a precomputed hash channel.
class BigintGroupTable {
public:
    BigintGroupTable(Context *ctx, int32_t type);
    // room for `need_groups` groups at a fill of at most 0.75 (grows by rebuilding from the published groups)
    void ensure_table(int64_t need_groups);
    void rebuild(int64_t min_capacity);
    int64_t capacity() const { return capacity_; }
    int64_t groups() const { return groups_; }
    // group ids of one sub-batch (rows numbered from 0): false = the table overflowed (rebuild + re-run); ctr = 8 zeroed device words
    bool process(const DeviceColumn &keys, const uint8_t *row_mask, int64_t n, int32_t *out_gids, unsigned long long *ctr, int64_t *new_groups);
    void lookup(const DeviceColumn &keys, int64_t n, int32_t *out_gids, unsigned long long *ctr);
    DeviceColumn key_column();   // values_by_group (+ the null flag of the NULL group), group-id order
    int64_t estimated_size() const;

private:
    void ensure_store(int64_t need_groups);
    Context *ctx_;
    int32_t type_;
    int width_;
    int64_t groups_ = 0, capacity_ = 0, store_cap_ = 0;
    int shift_ = 64;
    BufferPtr slots_, values_, nulls_;
};

class GroupByHashGpu {
public:
    static constexpr int kCounterSets = 64;   // counter sets per initialising launch (fresh_counters)
    // allow_integer_table = false: the caller brings its own probe kernels (GbhProbeFn), which speak the generic table's layout
    GroupByHashGpu(Context *ctx, std::vector<int32_t> types, bool has_input_hash, int32_t expected_size, bool allow_integer_table = true);

    // group id (int32, device) of each of the n rows; new keys get ids in first-seen order.
    // hashes == nullptr -> raw hashes are computed from the keys (InterpretedHashGenerator).
    // row_mask (optional, one byte per row): rows with 0 take no part and get id -1 (a filter fused in front of the table).
    // inline_hash: with hashes == nullptr, compute the raw hash inside the probe kernel instead of materialising it.
    // probe: optional external probe/insert kernel (replaces the generic one; it applies its own row filter)
    // out_gids8 (optional, with an external probe kernel): while the table holds fewer than 250 groups the ids are delivered as one
    // byte per row (group id + 1, 0 = excluded row) instead of four -- on TPCH Q1's shape that is a tenth of the whole
    // pipeline's HBM traffic.  Returns true when out_gids8 holds the page's ids (out_gids untouched), false when out_gids does.
    // speculate / *speculated (optional, compact mode): see GbhSpeculateFn; *speculated = the hook ran AND the counters came back clean
    // (what it enqueued took effect); false = it did not run, or ran and -- by its gate -- did nothing.
    bool get_group_ids(const std::vector<const DeviceColumn *> &keys, const int64_t *hashes, int64_t n, int32_t *out_gids,
                       const uint8_t *row_mask = nullptr, bool inline_hash = false, const GbhProbeFn *probe = nullptr, uint8_t *out_gids8 = nullptr,
                       const GbhSpeculateFn *speculate = nullptr, bool *speculated = nullptr);
    // lookup only (GroupByHash.contains): out[i] = group id or -1
    void lookup(const std::vector<const DeviceColumn *> &keys, const int64_t *hashes, int64_t n, int32_t *out_gids);

    int64_t group_count() const { return groups_; }
    // for callers that run lookups of their own against the groups' keys (the one-pass fused aggregation): the key store by group id, and a
    // zeroed counter set ([7] = ~0) from the ring
    KeyCols key_store_view() { ensure_store(groups_ > 0 ? groups_ : 1); return store_view(); }
    unsigned long long *counter_set() { return fresh_counters(); }
    int32_t java_capacity() const { return java_capacity_; }
    int32_t java_rehash_count() const { return java_rehashes_; }
    int64_t estimated_size() const;
    const std::vector<int32_t> &types() const { return types_; }
    bool has_input_hash() const { return has_input_hash_; }

    // keys of groups [0, group_count) in group-id order (+ the raw hash column when with_hash): appendValuesTo
    DevicePage key_page(bool with_hash);
    const int64_t *raw_hash_by_group() const { return raw_hash_ ? raw_hash_->as<int64_t>() : nullptr; }

private:
    struct KeyStore {
        int32_t type;
        BufferPtr values, nulls, offsets;
        int64_t cap = 0;         // groups
        int64_t pool_cap = 0;    // varchar bytes
        int64_t pool_used = 0;
    };
    void ensure_table(int64_t need_groups);
    void ensure_store(int64_t need_groups);
    void ensure_pool(KeyStore &ks, int64_t need_bytes);
    // returns false when the table overflowed (the caller rebuilds a bigger one and re-runs the rows); *new_groups out
    bool process_sub_batch(const KeyCols &batch, const int64_t *hashes, const uint8_t *row_mask, int64_t row0, int64_t n, int32_t *out_gids,
                           const GbhProbeFn *probe, int64_t *new_groups);
    void rebuild_table(int64_t min_capacity);
    KeyCols store_view() const;
    BufferPtr device_keys(const KeyCols &k);
    void advance_java_capacity();

    void get_group_ids_integer(const DeviceColumn &key, int64_t n, int32_t *out_gids, const uint8_t *row_mask);

    Context *ctx_;
    std::vector<int32_t> types_;
    bool has_input_hash_;
    std::unique_ptr<BigintGroupTable> integer_;   // set: single integer key, the table of groupby_bigint.hip does the work
    bool optimistic_ = false;                     // the next sub-batch is an optimistic one (integer table)
    bool high_cardinality_ = false;               // the last sub-batch was mostly new groups: size the table from the page's row bound
    int64_t groups_ = 0;
    int64_t capacity_ = 0;  // slots of the device table (uint64 words)
    BufferPtr words_;
    std::vector<KeyStore> store_;
    BufferPtr raw_hash_;
    int64_t raw_hash_cap_ = 0;
    int32_t java_capacity_, java_max_fill_, java_rehashes_ = 0;
    int64_t sub_batch_;
    int64_t last_new_groups_ = 0;   // new groups of the previous sub-batch (decides whether the next one ranks eagerly)
    int64_t next_sub_ = 0;   // size of the next sub-batch (ramps up, see get_group_ids)
    BufferPtr counters_;  // ring of counter sets: [0] pending rows, [1] new groups (scan total), [2] error flag, [3] scratch total, [7] expression error
    int next_counter_set_ = 0;
    unsigned long long *fresh_counters();
};

}  // namespace tgpu
