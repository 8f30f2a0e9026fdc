// groupby.h -- GroupByHash in HBM (K4/K5): open-address table probed by one lane per row, group ids assigned in
// first-seen order exactly as M/operator/BigintGroupByHash.java:213-260 and MultiChannelGroupByHash.java:281-342 do.
#pragma once

#include "common.h"

namespace tgpu {

class GroupByHashGpu {
public:
    GroupByHashGpu(Context *ctx, std::vector<int32_t> types, bool has_input_hash, int32_t expected_size);

    // group id (int32, device) of each of the n rows; new keys get ids in first-seen order.
    // hashes == nullptr -> raw hashes are computed from the keys (InterpretedHashGenerator).
    void get_group_ids(const std::vector<const DeviceColumn *> &keys, const int64_t *hashes, int64_t n, int32_t *out_gids);
    // lookup only (GroupByHash.contains): out[i] = group id or -1
    void lookup(const std::vector<const DeviceColumn *> &keys, const int64_t *hashes, int64_t n, int32_t *out_gids);

    int64_t group_count() const { return groups_; }
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
    void process_sub_batch(const KeyCols &batch, const int64_t *hashes, int64_t n, int32_t *out_gids);
    KeyCols store_view() const;
    void advance_java_capacity();

    Context *ctx_;
    std::vector<int32_t> types_;
    bool has_input_hash_;
    int64_t groups_ = 0;
    int64_t capacity_ = 0;  // slots of the device table (uint64 words)
    BufferPtr words_;
    std::vector<KeyStore> store_;
    BufferPtr raw_hash_;
    int64_t raw_hash_cap_ = 0;
    int32_t java_capacity_, java_max_fill_, java_rehashes_ = 0;
    int64_t sub_batch_;
    BufferPtr counters_;  // [0] pending rows, [1] new groups (scan total), [2] error flag, [3] scratch total
};

}  // namespace tgpu
