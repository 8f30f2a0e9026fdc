// agg.h -- grouped accumulators in HBM (K6): count / sum / avg states indexed by group id.
// Reference loop shape: M/operator/aggregation/AccumulatorCompiler.java:487-566 (for each position: mask, null check,
// state.setGroupId(groupIds[pos]); input(state, value)).
//
// DOUBLE sums use an order-independent exact accumulator (a Kulisch-style fixed-point "long accumulator" of 68 x 32-bit
// limbs held in int64 words, updated with integer atomics), so the result is the correctly rounded exact sum whatever
// the interleaving of lanes -- deterministic, and equal to the Java left-to-right sum whenever that sum is itself exact.
// See DESIGN.md "DOUBLE aggregate policy".
//
// With MANY groups (more than the lane-private LDS path holds) that state would cost 544 bytes per group and per aggregate,
// and neighbouring rows of one group would serialise on the same atomics.  Accumulators that allow it then switch to the
// ORDERED mode instead: the page's rows are sorted by group id (stable radix sort, row order kept inside a group) and one
// lane per group adds its rows sequentially, in row order, into a plain double per group -- the very order of the
// reference's per-position loop (AccumulatorCompiler.java:487-566), so the sums are bit-identical to the Java operator's.
#pragma once
#include <cstdlib>

#include "common.h"

namespace tgpu {

constexpr int kLimbs = 68;       // 32-bit limbs covering 2^-1074 .. 2^2101
constexpr int kMaxAggs = 16;
// ORDERED mode: the chained kernels (one workgroup per group) are chosen up to this many groups and from this many rows per group on average
// (a workgroup costs ~14 us before its first addition, 55 ns per group with 256 CUs at work; one lane per group costs 0.2-0.3 ns per ROW:
// tools/exp_medium_groups.py, 50 M rows -- 65 536 groups 5.7 against 14.5 ms, 262 144 groups 14.5 against 10.0 ms)
constexpr int64_t kOrdChainMaxGroups = 65536, kOrdChainMinRows = 256;
// one lane per group: a lane hands its group over to the chained kernel after this many rows (0.5 ms of a lane's time)
constexpr int64_t kOrdHandoffRows = 4096;
inline int64_t ord_chain_max_groups() { const char *e = getenv("TGPU_ORD_CHAIN_MAX_GROUPS"); return e ? atoll(e) : kOrdChainMaxGroups; }   // (the variable: kernel study)

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
        double *dsum;   // ORDERED mode (limbs / special are null then)
    };
    void reserve(int64_t groups) { ensure(groups > 0 ? groups : 1); }
    // For callers that run their own accumulate kernels (the JIT-fused operator): fixes the mode with the caller's own
    // low-cardinality capacity; when the accumulators are (now) ORDERED, reserves the states and returns the page's
    // (group id + 1, row) pairs in (group, row) order -- the caller then adds the rows of every group in that order.
    bool begin_ordered(const int32_t *gids, int64_t n, int64_t groups, int64_t lowcard_max_groups, BufferPtr &keys, BufferPtr &rows);
    // the chained kernels' input: {first, end} index of every group id's stretch of the sorted keys (ids pairs of int32; {0, 0} = no rows)
    BufferPtr group_stretches(const unsigned int *keys, int64_t n, int64_t ids);
    // the JIT-fused accumulate kernels address the exact (limb) state directly: they keep the accumulators out of ORDERED mode
    void set_allow_ordered(bool on) { allow_ordered_ = on; }
    // TGPU_SUM_ORDER_JAVA: ORDERED whatever the number of groups (every group's rows are added in row order)
    void set_force_ordered(bool on) { force_ordered_ = on; }
    // merge(): every spilled run carries row-order sums -> combine them like the reference (sequential double adds in run order)
    void set_combine_ordered(bool on) { combine_ordered_ = on; }
    bool force_ordered() const { return force_ordered_ && allow_ordered_; }
    bool ordered() const { return mode_ == Mode::ORDERED; }
    // The DOUBLE mode (EXACT limbs vs ORDERED row-order sums) is a function of the operator's ROW STREAM, not of how it is cut into pages: it
    // is decided once, from the number of groups among the stream's first kModePrefixRows rows (the callers hold accumulation back until
    // they have seen that many rows, HashAggregationOperator::process_page).  lowcard_max_groups <= 0: the AOT kernels' own capacity.
    static constexpr int64_t kModePrefixRows = 65536;
    bool decided() const { return mode_ != Mode::UNDECIDED; }
    void decide(int64_t groups_in_prefix, int64_t lowcard_max_groups = 0);
    // number of groups among the first m rows of a page whose ids are first-seen ranks: max id + 1 (compact ids: byte = id + 1)
    static int64_t groups_among(Context *ctx, const int32_t *gids, const uint8_t *gids8, int64_t m);
    DeviceState device_state(int k) const;
    // folded partials of the low-cardinality launches (device_agg.h TgFoldScratch): one row per workgroup, `group_capacity` x
    // aggregates items per row; the launches add to them, flush_fold() adds them exactly into the states (evaluate does it)
    struct FoldScratch {
        unsigned long long *partials;
        int32_t stride;
        unsigned long long *pending;   // the one-pass launches' totals of a page not yet known to be clean (TgFoldScratch)
    };
    FoldScratch fold_scratch(int64_t blocks, int64_t group_capacity);
    void flush_fold();
    // the pending totals of the LAST one-pass launch (`blocks` workgroup rows): added to the partials (the page was clean) or dropped
    void resolve_pending(int64_t blocks, bool commit);
    // Spill support (SpillableHashAggregationBuilder.java:283-299): the accumulator state of groups [0, groups) in host memory -- the
    // EXACT state itself (counts, limb accumulators, NaN / inf flags, 128-bit bigint sums; ORDERED mode: the running double sums), not a
    // rounded intermediate value, so that merging runs loses nothing.
    struct HostStates {
        int64_t groups = 0;
        struct Agg {
            std::vector<long long> counts, limbs;
            std::vector<unsigned int> special;
            std::vector<unsigned long long> i128;
            std::vector<double> dsum;
        };
        std::vector<Agg> aggs;
        int64_t bytes() const;
    };
    HostStates dump(int64_t groups);
    // adds run group i's state to group gids[i] (gids == nullptr: group 0, a global aggregation); this accumulator runs in EXACT mode
    // from then on (a run's running double sums are added exactly, as one addend each)
    void merge(const int32_t *gids, const HostStates &run, int64_t live_groups);
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
        BufferPtr dsum;     // double[g]: ORDERED mode running sums (instead of limbs / special)
        int64_t cap = 0;
    };
    enum class Mode { UNDECIDED, EXACT, ORDERED };
    void ensure(int64_t groups);
    void decide_mode(int64_t groups, int64_t lowcard_max_groups);
    void sort_rows_by_group(const int32_t *gids, int64_t n, int64_t groups, BufferPtr &keys, BufferPtr &rows);
    Mode mode_ = Mode::UNDECIDED;
    bool allow_ordered_ = false, force_ordered_ = false;
    Context *ctx_;
    std::vector<State> states_;
    int32_t step_;
    BufferPtr error_;  // device uint32
    BufferPtr fold_partials_, fold_pending_;
    bool combine_ordered_ = false;
    int64_t fold_rows_ = 0, fold_stride_ = 0;
    bool fold_dirty_ = false;
};

}  // namespace tgpu
