// operators.h -- host-side mirror of the reference's operator interfaces (M/operator/Operator.java:20-102,
// OperatorFactory.java:18-50) and the four hot-path operators, as state machines over the device kernels.
#pragma once

#include "agg.h"
#include <deque>
#include <functional>

#include "common.h"
#include "groupby.h"
#include "jit.h"
#include "join.h"
#include "topn.h"

namespace tgpu {

class Operator {
public:
    explicit Operator(Context *ctx, int32_t operator_id) : ctx_(ctx), operator_id_(operator_id) {}
    virtual ~Operator() {}
    virtual bool needs_input() = 0;
    virtual void add_input(const tgpu_page *page) = 0;
    // the same with a page this library produced (an OutputPage): its buffers are shared, so an operator that keeps or forwards the
    // page does so without copying it.  Default: the borrowed-page path.
    virtual void add_input_owned(const DevicePage &page);
    virtual std::unique_ptr<OutputPage> get_output() = 0;  // nullptr = no page available
    virtual void finish() = 0;
    virtual bool is_finished() = 0;
    virtual bool is_blocked() { return false; }
    virtual int64_t memory_bytes() { return 0; }
    // Operator.startMemoryRevoke / finishMemoryRevoke (M/operator/Operator.java:53-79) and OperatorContext.getReservedRevocableBytes:
    // only a spill-enabled hash aggregation holds revocable memory
    virtual void start_memory_revoke() {}
    virtual void finish_memory_revoke() {}
    virtual int64_t revocable_memory_bytes() { return 0; }
    virtual void spill_stats(int64_t &spill_count, int64_t &spilled_bytes) { spill_count = 0; spilled_bytes = 0; }
    virtual void close() {}
    Context *context() const { return ctx_; }

protected:
    std::unique_ptr<OutputPage> wrap(DevicePage &&p)
    {
        auto o = std::make_unique<OutputPage>();
        o->ctx = ctx_;
        o->page = std::move(p);
        return o;
    }
    Context *ctx_;
    int32_t operator_id_;
};

class OperatorFactory {
public:
    virtual ~OperatorFactory() {}
    virtual std::unique_ptr<Operator> create_operator() = 0;
    virtual void no_more_operators() { closed_ = true; }
    // OperatorFactory.duplicate() (M/operator/OperatorFactory.java:49): an independent factory of the same operator for another
    // pipeline instance; a hash build cannot be duplicated (HashBuilderOperator.java:150-152 throws UnsupportedOperationException)
    virtual std::unique_ptr<OperatorFactory> duplicate() { fail(TGPU_ERR_NOT_SUPPORTED, "this operator factory cannot be duplicated"); }

protected:
    bool closed_ = false;
};

// ---- FilterAndProjectOperator (M/operator/FilterAndProjectOperator.java:37-65,117-147) ----------------------------
class FilterAndProjectOperatorFactory : public OperatorFactory {
public:
    FilterAndProjectOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    std::shared_ptr<PageProcessorGpu> processor_;
};

int64_t filter_project_dictionary_pages(Operator *op);

// ---- ScanFilterAndProjectOperator (M/operator/ScanFilterAndProjectOperator.java:66-447), page-source flavour ------------------------
class ScanFilterAndProjectOperatorFactory : public OperatorFactory {
public:
    ScanFilterAndProjectOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, const tgpu_page_processor_spec *spec);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    std::vector<int32_t> types_;
    std::shared_ptr<PageProcessorGpu> processor_;
};
void scan_add_page_source(Operator *op, const tgpu_page_source *source);
void scan_no_more_splits(Operator *op);
void scan_stats(Operator *op, int64_t *processed_positions, int64_t *lazy_loaded, int64_t *lazy_skipped);

// ---- HashAggregationOperator (M/operator/HashAggregationOperator.java:54-262,367-518) -----------------------------
struct HashAggregationConfig {
    std::vector<int32_t> group_by_types, group_by_channels;
    int32_t hash_channel = -1;
    int32_t step = TGPU_STEP_SINGLE;
    std::vector<tgpu_agg_spec> aggs;
    int32_t expected_groups = 1024;
    bool produce_default_output = false;
    int64_t max_partial_memory = 16ll << 20;  // TaskManagerConfig.java:47 (max_partial_aggregation_memory)
    bool spill_enabled = false;               // HashAggregationOperator.java:133,389-425: SINGLE / FINAL steps get the spillable builder
};

class HashAggregationOperatorFactory : public OperatorFactory {
public:
    HashAggregationOperatorFactory(Context *ctx, int32_t operator_id, HashAggregationConfig cfg);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;
    void set_spill_enabled(bool on) { cfg_.spill_enabled = on; }
    void set_max_partial_memory(int64_t bytes) { cfg_.max_partial_memory = bytes; }

private:
    Context *ctx_;
    int32_t operator_id_;
    HashAggregationConfig cfg_;
};

// ---- JoinFilterFunction (M/operator/JoinHash.java:44-47,118-130; M/sql/gen/JoinFilterFunctionCompiler.java): a predicate over
// (build row, probe row) that a join position must pass besides key equality.  Input channels [0, build types) of the expression
// are the build side's channels, the following ones the probe page's.
struct JoinFilter {
    std::vector<tgpu_expr_node> nodes;
    std::string pool;
    int32_t root = -1;
    std::vector<int32_t> build_types, probe_types;
};

// ---- join bridge: PartitionedLookupSourceFactory with one partition (M/operator/PartitionedLookupSourceFactory.java:146-205)
class LookupSourceFactory {
public:
    // The bridge is shared by the build operator, every probe operator and the outer operator, which the reference runs in
    // different drivers (threads): all of its state is guarded by one mutex.
    std::shared_ptr<LookupSourceGpu> lookup_source() const
    {
        std::lock_guard<std::mutex> lk(mu_);
        return source_;
    }
    void lend(std::shared_ptr<LookupSourceGpu> s)
    {
        std::lock_guard<std::mutex> lk(mu_);
        source_ = std::move(s);
    }
    // PartitionedLookupSourceFactory.lendPartitionLookupSource (M/operator/PartitionedLookupSourceFactory.java:110-124, 146-205): the build
    // side may run as P HashBuilderOperators (one per local-exchange partition); the probes stay blocked until every partition has been
    // lent.  On the GPU the P partitions exist to feed the build, not to be probed separately: the last lender concatenates them in
    // partition order into ONE index and builds ONE table (`make_source`) -- the chip builds and probes a single table faster than P
    // small ones.  Observable behaviour equals PartitionedLookupSource's (PartitionedLookupSource.java:87-153): all rows of a key sit in
    // one partition, so a key's chain (newest -> oldest) is that partition's; the outer iterator walks partition 0's positions, then
    // partition 1's ... (:233-262) = ascending positions of the concatenation.  Returns true for the call that completed the set.
    void set_partitioning(int partitions, std::function<std::shared_ptr<LookupSourceGpu>(std::vector<std::shared_ptr<PagesIndexGpu>> &)> make_source)
    {
        std::lock_guard<std::mutex> lk(mu_);
        partitions_.assign((size_t)partitions, nullptr);
        lent_.assign((size_t)partitions, false);
        make_source_ = std::move(make_source);
    }
    bool lend_partition(int partition, std::shared_ptr<PagesIndexGpu> index)
    {
        std::lock_guard<std::mutex> lk(mu_);
        TG_CHECK_STATE(partition >= 0 && partition < (int)partitions_.size() && !lent_[(size_t)partition], "partition already lent");
        partitions_[(size_t)partition] = std::move(index);
        lent_[(size_t)partition] = true;
        for (bool b : lent_)
            if (!b) return false;
        source_ = make_source_(partitions_);
        partitions_.clear();
        return true;
    }
    void probe_created()
    {
        std::lock_guard<std::mutex> lk(mu_);
        live_probes_++;
        any_probe_ = true;
    }
    void probe_closed()
    {
        std::lock_guard<std::mutex> lk(mu_);
        live_probes_--;
    }
    // every probe-side factory of the join (the original and its duplicate()s) registers; the probes are complete when the last
    // of them has seen noMoreOperators (JoinBridgeManager's probe factory reference counting, M/operator/JoinBridgeManager.java)
    void probe_factory_created()
    {
        std::lock_guard<std::mutex> lk(mu_);
        probe_factories_++;
    }
    void no_more_probes()
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (probe_factories_ > 0) probe_factories_--;
        if (probe_factories_ == 0) no_more_probes_ = true;
    }
    // every probe operator is done: the outer position iterator becomes available (PartitionedLookupSourceFactory.java:259-297)
    bool probes_finished() const
    {
        std::lock_guard<std::mutex> lk(mu_);
        return no_more_probes_ && live_probes_ == 0;
    }
    // LOOKUP_OUTER / FULL_OUTER joins: the LookupOuterOperator keeps the table alive until it has emitted the unmatched rows
    void outer_expected()
    {
        std::lock_guard<std::mutex> lk(mu_);
        outer_expected_ = true;
    }
    void outer_done()
    {
        std::lock_guard<std::mutex> lk(mu_);
        outer_done_ = true;
    }
    // the build operator may release the table once every probe operator is done (HashBuilderOperator.java:429-470)
    bool destroyed() const
    {
        std::lock_guard<std::mutex> lk(mu_);
        return no_more_probes_ && live_probes_ == 0 && (!outer_expected_ || outer_done_);
    }
    std::vector<int32_t> build_output_types, build_types;   // written by the build factory's constructor, read-only afterwards
    // set before the probe operators are created (the reference passes it to the build side: JoinHashSupplier.java:54-70)
    void set_join_filter(std::shared_ptr<const JoinFilter> f)
    {
        std::lock_guard<std::mutex> lk(mu_);
        filter_ = std::move(f);
    }
    std::shared_ptr<const JoinFilter> join_filter() const
    {
        std::lock_guard<std::mutex> lk(mu_);
        return filter_;
    }

private:
    mutable std::mutex mu_;
    std::vector<std::shared_ptr<PagesIndexGpu>> partitions_;
    std::vector<bool> lent_;
    std::function<std::shared_ptr<LookupSourceGpu>(std::vector<std::shared_ptr<PagesIndexGpu>> &)> make_source_;
    std::shared_ptr<const JoinFilter> filter_;
    std::shared_ptr<LookupSourceGpu> source_;
    int live_probes_ = 0, probe_factories_ = 0;
    bool any_probe_ = false, no_more_probes_ = false, outer_expected_ = false, outer_done_ = false;
};

struct HashBuilderConfig {
    std::vector<int32_t> types, output_channels, hash_channels;
    int32_t precomputed_hash_channel = -1;
    int32_t expected_positions = 0;
    int32_t partition_count = 1;   // HashBuilderOperators the factory hands out (LocalExecutionPlanner.java:2129 partitionCount = build driver instances)
};

class HashBuilderOperatorFactory : public OperatorFactory {
public:
    HashBuilderOperatorFactory(Context *ctx, int32_t operator_id, HashBuilderConfig cfg, std::shared_ptr<LookupSourceFactory> bridge);
    std::unique_ptr<Operator> create_operator() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    HashBuilderConfig cfg_;
    std::shared_ptr<LookupSourceFactory> bridge_;
    int created_ = 0;
};

struct LookupJoinConfig {
    std::vector<int32_t> probe_types, probe_join_channels, probe_output_channels;
    int32_t probe_hash_channel = -1;
    int32_t join_type = TGPU_JOIN_INNER;
};

class LookupJoinOperatorFactory : public OperatorFactory {
public:
    LookupJoinOperatorFactory(Context *ctx, int32_t operator_id, LookupJoinConfig cfg, std::shared_ptr<LookupSourceFactory> bridge);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;
    void no_more_operators() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    LookupJoinConfig cfg_;
    std::shared_ptr<LookupSourceFactory> bridge_;
};

// ---- LookupOuterOperator (M/operator/LookupOuterOperator.java:32-235): the build rows no LOOKUP_OUTER / FULL_OUTER probe matched ----
class LookupOuterOperatorFactory : public OperatorFactory {
public:
    LookupOuterOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> probe_output_types, std::shared_ptr<LookupSourceFactory> bridge);
    std::unique_ptr<Operator> create_operator() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    std::vector<int32_t> probe_output_types_;
    std::shared_ptr<LookupSourceFactory> bridge_;
    bool created_ = false;
};

// ---- FilterAndProject fused into the LookupJoin probe (operator fusion by codegen, jit.h FusedProbeGpu).  The pair behaves
// exactly like FilterAndProjectOperator feeding LookupJoinOperator; `cfg` addresses the page processor's projections. ----
class FusedFilterProjectJoinOperatorFactory : public OperatorFactory {
public:
    FusedFilterProjectJoinOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec,
                                          LookupJoinConfig cfg, std::shared_ptr<LookupSourceFactory> bridge);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;
    void no_more_operators() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    LookupJoinConfig cfg_;
    std::shared_ptr<LookupSourceFactory> bridge_;
    std::shared_ptr<PageProcessorGpu> processor_;
    std::shared_ptr<FusedProbeGpu> fused_;
};

// ---- FilterAndProject fused into the HashAggregation (jit.h FusedAggGpu); cfg's channels address the projections ----
class FusedFilterProjectAggregationOperatorFactory : public OperatorFactory {
public:
    FusedFilterProjectAggregationOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec,
                                                 HashAggregationConfig cfg);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;
    void set_spill_enabled(bool on) { cfg_.spill_enabled = on; }
    void set_max_partial_memory(int64_t bytes) { cfg_.max_partial_memory = bytes; }

private:
    Context *ctx_;
    int32_t operator_id_;
    HashAggregationConfig cfg_;
    std::shared_ptr<PageProcessorGpu> processor_;
    std::shared_ptr<FusedAggGpu> fused_;
};

// ---- TopNOperator (M/operator/TopNOperator.java:47-62,135-225) ----------------------------------------------------------
class TopNOperatorFactory : public OperatorFactory {
public:
    TopNOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, int64_t n, std::vector<int32_t> sort_channels,
                        std::vector<int32_t> sort_orders);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    std::vector<int32_t> types_, sort_channels_, sort_orders_;
    int64_t n_;
};

// ---- OrderByOperator (M/operator/OrderByOperator.java:48-131,160-300) ----------------------------------------------------
class OrderByOperatorFactory : public OperatorFactory {
public:
    OrderByOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, std::vector<int32_t> output_channels, std::vector<int32_t> sort_channels,
                           std::vector<int32_t> sort_orders);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    std::vector<int32_t> types_, output_channels_, sort_channels_, sort_orders_;
};

// ---- DynamicFilterSourceOperator (M/operator/DynamicFilterSourceOperator.java:46-425) ---------------------------------------------
// Passes the build side's pages through and collects, per join-key channel, what the probe side's scan may be narrowed to.
class DynamicFilterSourceOperatorFactory : public OperatorFactory {
public:
    DynamicFilterSourceOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, std::vector<int32_t> channels, int32_t max_distinct_values,
                                       int64_t max_filter_size_in_bytes, int32_t min_max_collection_limit);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    std::vector<int32_t> types_, channels_;
    int32_t max_distinct_, min_max_limit_;
    int64_t max_size_;
};
// the collected domain of filter channel `k` after finish(): kind 0 = ALL, 1 = VALUES (*values: one channel holding the distinct
// non-null, non-NaN values in first-seen order), 2 = RANGE [*min, *max] (BIGINT / INTEGER / DATE), 3 = NONE (only nulls were seen)
void dynamic_filter_result(Operator *op, int32_t k, int32_t *kind, std::unique_ptr<OutputPage> *values, int64_t *min, int64_t *max);

// ---- MergePages (M/operator/project/MergePages.java:40-190) as an operator: coalesces small pages in HBM -------------------------
class MergePagesOperatorFactory : public OperatorFactory {
public:
    MergePagesOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, int64_t min_page_size_in_bytes, int32_t min_row_count,
                              int64_t max_page_size_in_bytes);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    std::vector<int32_t> types_;
    int64_t min_page_size_, max_page_size_;
    int32_t min_row_count_;
};

// ---- PartitionedOutputOperator (M/operator/PartitionedOutputOperator.java:46-300; PagePartitioner :308-486) ---------------------
// A sink: input rows are grouped by destination partition and handed out as (partition, page) pairs -- what the reference
// enqueues into its OutputBuffer -- through poll(); get_output() returns nothing, as in the reference (:303-306).
class PartitionedOutputOperator;
class PartitionedOutputOperatorFactory : public OperatorFactory {
public:
    PartitionedOutputOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, std::vector<int32_t> partition_channels, int32_t hash_channel,
                                     int32_t partition_count, bool replicates_any_row, int32_t null_channel, int32_t partition_function);
    std::unique_ptr<Operator> create_operator() override;
    std::unique_ptr<OperatorFactory> duplicate() override;

private:
    Context *ctx_;
    int32_t operator_id_;
    std::vector<int32_t> types_, partition_channels_;
    int32_t hash_channel_, partition_count_, null_channel_;
    bool replicates_any_row_, local_function_;
};
// next pending (partition, page) pair of a PartitionedOutputOperator; false = nothing pending
bool partitioned_output_poll(Operator *op, int32_t *partition, std::unique_ptr<OutputPage> *out);
void partitioned_output_pending(Operator *op, size_t *max_per_partition, int32_t *partition_count);
void partitioned_output_info(Operator *op, int64_t *rows_added, int64_t *pages_added);

}  // namespace tgpu

struct tgpu_context {
    std::unique_ptr<tgpu::Context> ctx;
};
struct tgpu_operator_factory {
    std::unique_ptr<tgpu::OperatorFactory> f;
    tgpu::Context *ctx = nullptr;
};
struct tgpu_operator {
    std::unique_ptr<tgpu::Operator> op;
    tgpu::Context *ctx = nullptr;
    // tgpu_context_set_max_output_page: an output page larger than the limits is handed out as consecutive regions (shared buffers)
    std::deque<std::unique_ptr<tgpu::OutputPage>> cut_pages;
};
struct tgpu_lookup_source_factory {
    std::shared_ptr<tgpu::LookupSourceFactory> bridge;
    tgpu::Context *ctx = nullptr;
};
struct tgpu_group_by_hash {
    tgpu::Context *ctx;
    std::unique_ptr<tgpu::GroupByHashGpu> gbh;
    std::vector<int32_t> hash_channels;
    int32_t input_hash_channel;
};
