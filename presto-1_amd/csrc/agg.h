// agg.h -- grouped accumulators in HBM (K6): count / sum / avg states indexed by group id.
// Reference loop shape: M/operator/aggregation/AccumulatorCompiler.java:487-566 (for each position: mask, null check,
// state.setGroupId(groupIds[pos]); input(state, value)).
//
// DOUBLE sums use an order-independent exact accumulator (a Kulisch-style fixed-point "long accumulator" of 68 x 32-bit
// limbs held in int64 words, updated with integer atomics), so the result is the correctly rounded exact sum whatever
// the interleaving of lanes -- deterministic, and equal to the Java left-to-right sum whenever that sum is itself exact.
// See DESIGN.md "DOUBLE aggregate policy".
#pragma once

#include "common.h"

namespace tgpu {

constexpr int kLimbs = 68;       // 32-bit limbs covering 2^-1074 .. 2^2101
constexpr int kMaxAggs = 16;

class GroupedAccumulators {
public:
    GroupedAccumulators(Context *ctx, std::vector<tgpu_agg_spec> specs, int32_t step);

    // raw input (SINGLE / PARTIAL): gids == nullptr means one global group 0
    void add_input(const int32_t *gids, int64_t n, const DevicePage &page, int64_t group_count);
    // intermediate input (FINAL): aggregate k reads its state from channel spec.input_channel (count) and, for sum / avg,
    // spec.input_channel + 1 (sum)
    void add_intermediate(const int32_t *gids, int64_t n, const DevicePage &page, int64_t group_count);
    // appends the output channels for groups [0, group_count)
    void evaluate(int64_t group_count, std::vector<DeviceColumn> &out);
    // device state of aggregate k for groups [0, reserve()d): used by the JIT-fused project+accumulate kernels
    struct DeviceState {
        int32_t function;
        long long *counts;
        long long *limbs;
        unsigned int *special;
        unsigned long long *i128;
    };
    void reserve(int64_t groups) { ensure(groups > 0 ? groups : 1); }
    DeviceState device_state(int k) const;
    const std::vector<tgpu_agg_spec> specs() const;
    int output_channel_count() const;
    int intermediate_channel_count() const;
    int64_t estimated_size() const;

private:
    struct State {
        tgpu_agg_spec spec;
        BufferPtr counts;   // int64[g]
        BufferPtr limbs;    // int64[g][kLimbs]  (double sums)
        BufferPtr special;  // uint32[g] NaN / +-inf flags
        BufferPtr i128;     // uint64[g][2] (bigint sums)
        int64_t cap = 0;
    };
    void ensure(int64_t groups);
    Context *ctx_;
    std::vector<State> states_;
    int32_t step_;
    BufferPtr error_;  // device uint32
};

}  // namespace tgpu
