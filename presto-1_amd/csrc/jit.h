// jit.h -- the GPU counterpart of the reference's expression codegen (M/sql/gen/ExpressionCompiler.java:94-122,
// PageFunctionCompiler.java:164-212,367-404): a RowExpression filter + projections is compiled (hiprtc, gfx950) into one
// fused, row-selective kernel pair and cached on disk by source hash.
#pragma once

#include <algorithm>

#include "common.h"

namespace tgpu {

constexpr int kFpMaxCols = 24;
constexpr int kFpMaxProj = 16;

struct FpArgs {  // must match the struct declared in the generated source
    const void *col_values[kFpMaxCols];
    const uint8_t *col_nulls[kFpMaxCols];
    const int32_t *col_offsets[kFpMaxCols];
    void *out_values[kFpMaxProj];
    uint8_t *out_nulls[kFpMaxProj];
    int32_t *positions;
    const int32_t *tile_offsets;
    int32_t *tile_counts;
    unsigned long long *error;
    long long n;
    unsigned long long *sel_mask;
};

struct JitModule;

// PageProcessor (M/operator/project/PageProcessor.java:111-137): filter, then projections on the selected positions only.
class PageProcessorGpu {
public:
    PageProcessorGpu(std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec);
    ~PageProcessorGpu();
    // Process-wide cache of compiled page processors keyed by (input types, expressions): operator factories are created per
    // query, their expressions repeat -- the counterpart of the reference's compiled-class cache
    // (M/sql/gen/PageFunctionCompiler.java:101-139).  The shared objects are immutable apart from their lazily loaded modules.
    static std::shared_ptr<PageProcessorGpu> shared(const std::vector<int32_t> &input_types, const tgpu_page_processor_spec *spec);
    // generates the kernel source and compiles it into the on-disk cache; needs no GPU (used by build() to pre-warm)
    void precompile();
    // returns false when no row is selected (no output page); `out` gets one column per projection
    bool process(Context *ctx, const DevicePage &in, DevicePage &out);
    const std::vector<int32_t> &output_types() const { return output_types_; }
    const std::string &source() const { return source_; }
    // Dictionary-aware processing (M/operator/project/DictionaryAwarePageFilter.java:56-110, DictionaryAwarePageProjection.java,
    // PageFunctionCompiler.java:176-212): the input channel every computed expression (filter and non-identity projections) reads
    // when they all read the same single one, else -1.
    int single_input_channel() const { return single_input_; }
    // input channels the filter reads / the projections (computed or identity) read: what a lazy page has to load, and when
    const std::vector<int> &filter_channels() const { return filter_channels_; }
    const std::vector<int> &projection_channels() const { return projection_channels_; }
    // the same filter without projections (lazy pages: decides whether the projection-only channels are loaded at all)
    std::shared_ptr<PageProcessorGpu> filter_only();
    // The processor evaluated once per DICTIONARY ENTRY: `dictionary` stands for channel single_input_channel(); returns one row per
    // entry: [the filter's verdict as BOOLEAN (only when there is a filter), every computed projection in slot order].
    // Throws like process() when an entry raises (the caller then falls back to the flat path: only selected rows may raise).
    void process_dictionary(Context *ctx, const DeviceColumn &dictionary, DevicePage &out);
    bool has_filter() const { return filter_root_ >= 0; }
    // per projection: -1 = computed (its slot via computed_slot), else the input channel it passes through
    int identity_channel(int projection) const { return projs_[(size_t)projection].kind == ProjKind::IDENTITY ? projs_[(size_t)projection].channel : -1; }
    bool is_null_constant(int projection) const { return projs_[(size_t)projection].kind == ProjKind::CONSTANT_NULL; }
    int computed_slot(int projection) const { return projs_[(size_t)projection].slot; }
    int projection_count() const { return (int)projs_.size(); }

private:
    enum class ProjKind { COMPUTED, IDENTITY, CONSTANT_NULL };
    struct Proj {
        ProjKind kind;
        int channel = -1;   // IDENTITY
        int slot = -1;      // COMPUTED: index into FpArgs.out_*
        int32_t type = 0;
    };
    void generate();
    void ensure_loaded(Context *ctx);
    std::mutex mu_;

    std::vector<int32_t> input_types_;
    std::vector<tgpu_expr_node> nodes_;
    std::string pool_;
    int filter_root_;
    std::vector<int32_t> proj_roots_;
    std::vector<Proj> projs_;
    std::vector<int32_t> output_types_;
    int computed_count_ = 0;
    std::string source_;
    std::shared_ptr<JitModule> module_;
    hipFunction_t fn_count_ = nullptr, fn_emit_ = nullptr;
    int single_input_ = -1;
    std::vector<int> filter_channels_, projection_channels_;
    std::shared_ptr<PageProcessorGpu> filter_only_;
    std::shared_ptr<PageProcessorGpu> dict_processor_;   // lazily: the same expressions over the one-channel dictionary page
};

// ---- operator fusion by codegen ---------------------------------------------------------------------------------------
// FilterAndProject feeding a LookupJoin probe, compiled into one kernel: filter -> key projection -> Bloom filter -> table
// probe -> order-preserving compaction of the (probe row, build position) pairs (in-tile ballot compaction into workgroup-
// private regions, then a scan over the per-tile counts); the probe-side output projections are then evaluated for the
// matching rows only.  Results are identical to running the two
// reference operators back to back (M/operator/FilterAndProjectOperator.java + LookupJoinOperator.java).
constexpr int kFjCountSlots = 64, kFjMiscWords = 16 + kFjCountSlots * 16;   // misc words of a probe launch: [0] error, [2] pairs, [16 + 16 i] selected rows
struct FjArgs {  // must match the generated struct
    FpArgs fp;
    const void *slots;
    unsigned long long mask;
    const unsigned long long *bitmap;
    long long key_min, key_max;
    const unsigned long long *bloom;
    unsigned long long bloom_word_mask;
    const int32_t *direct;
    const int32_t *rank_base;
    int32_t *tile_cnt;
    int32_t *tile_src;
    const int32_t *tile_dst;
    int32_t *pair_probe;
    int32_t *pair_build;
    int32_t *out_build;
    unsigned long long *counters;   // [0] rows selected by the filter
    long long tiles;
    long long grid1;
    int32_t outer;
    int32_t chunk_shift;
    // build-side output channels gathered by pass 2 itself (fixed-width columns; one launch instead of one more per channel)
    struct BuildCol {
        const void *values;
        const uint8_t *nulls;
        void *out_values;
        uint8_t *out_nulls;
        int32_t width;
        int32_t pad;
    } bcol[4];
    int32_t n_bcol;
    int32_t pad2;
    void *carry[4];         // carry variant: block-private regions of the probe-side output values, one per output channel
    uint8_t *carry_nulls;   //                and of their null bits (nullptr: no output can be null)
    unsigned long long *host_out;   // small pages: pass 1's last workgroup scans the chunk counts and writes {error, pairs, selected} here
    unsigned int *done;             //              (host-visible signal slot); `done` counts finished workgroups
    const FpArgs *pages;            // multi-page launches (kernel variant FJ_EPILOGUE == 2): the pages' column pointers, device memory
    const int32_t *page_tile0;      //   first tile of each page, [n_pages] = all tiles
    int32_t n_pages;
    int32_t pad3;
};
constexpr int kFjMaxBuildCols = 4;
constexpr int64_t kFjMultiMaxTiles = 32768;      // tiles of one launch over a list of pages
constexpr int64_t kFjEpilogueMaxChunks = 2048;   // probe launches of up to 2048 tiles (1.5 M rows at 768-row tiles; always one-tile chunks) end pass 1 with the epilogue

class LookupSourceGpu;

class FusedProbeGpu {
public:
    // join_channel / output_channels index the page processor's projections (= the probe page the join would have seen)
    FusedProbeGpu(std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec, int32_t join_channel, std::vector<int32_t> output_channels);
    ~FusedProbeGpu();
    static std::shared_ptr<FusedProbeGpu> shared(const std::vector<int32_t> &input_types, const tgpu_page_processor_spec *spec, int32_t join_channel,
                                                 const std::vector<int32_t> &output_channels);   // see PageProcessorGpu::shared
    void precompile();
    bool supported() const { return supported_; }   // false -> the operator runs the unfused composition
    const std::vector<int32_t> &projection_types() const { return proj_types_; }
    // probes `in` against the lookup source's int-key table; returns the probe-side output columns + build positions
    // need_build_positions = false (no build output channels, no outer tracking): build_idx is left undefined and the DIRECT
    // layout skips its rank / position lookups
    // build_cols / build_out (optional): the lookup source's fixed-width output channels, gathered by the emit pass itself into
    // build_out (null vector policy of k::gather_column: only when the source has one or unmatched outer rows can occur)
    void process(Context *ctx, const DevicePage &in, const LookupSourceGpu &source, bool outer, bool need_build_positions, std::vector<DeviceColumn> &probe_out,
                 BufferPtr &build_idx, int64_t &count, int64_t &selected_rows, const std::vector<DeviceColumn> *build_cols = nullptr,
                 std::vector<DeviceColumn> *build_out = nullptr);
    // the same in two halves: begin() launches pass 1 and the read-back of its totals, finish() waits for that read and launches pass 2.  `in`
    // and the lookup source must stay alive and unchanged in between (null from begin() = empty page)
    struct Pending;
    std::shared_ptr<Pending> begin(Context *ctx, const DevicePage &in, const LookupSourceGpu &source, bool outer, bool need_build_positions, bool launch = true);
    // several (non-empty) pages probed as one sequence of rows: one launch, one output page (rows in page order).  At most
    // multi_page_row_limit() rows (counted in whole tiles per page); finish() takes any of the pages as `in`.  Returns null when the
    // context has no signal slot left for the launch: the caller then probes the pages one by one.
    std::shared_ptr<Pending> begin(Context *ctx, const std::vector<const DevicePage *> &pages, const LookupSourceGpu &source, bool outer, bool need_build_positions,
                                   bool launch = true);
    static int64_t multi_page_row_limit();
    void finish(Context *ctx, const std::shared_ptr<Pending> &pending, const DevicePage &in, std::vector<DeviceColumn> &probe_out, BufferPtr &build_idx, int64_t &count,
                int64_t &selected_rows, const std::vector<DeviceColumn> *build_cols = nullptr, std::vector<DeviceColumn> *build_out = nullptr, bool launch = true);
    // launch == false leaves the pass to the caller: launch_probe / launch_emit alone, or launch_pair = an earlier page's pass 2 and a new page's
    // pass 1 in ONE launch (page variant of the kernels only: can_pair) -- two latency-bound grids side by side instead of one after the other
    void launch_probe(Context *ctx, const std::shared_ptr<Pending> &pending);
    void launch_emit(Context *ctx, const std::shared_ptr<Pending> &pending);
    bool can_pair(const std::shared_ptr<Pending> &emit_side, const std::shared_ptr<Pending> &probe_side) const;
    void launch_pair(Context *ctx, const std::shared_ptr<Pending> &emit_side, const std::shared_ptr<Pending> &probe_side);
    void cancel(Context *ctx, const std::shared_ptr<Pending> &pending);   // a begun page nobody will finish (operator closed early)
    const std::string &source() const { return source_; }

private:
    void generate();
    struct JitModule *module_for(int prefilter_kind, bool no_nulls, bool carry, int epilogue);
    std::mutex mu_;
    std::vector<int32_t> input_types_;
    std::vector<tgpu_expr_node> nodes_;
    std::string pool_;
    int filter_root_;
    std::vector<int32_t> proj_roots_, proj_types_, output_channels_;
    int32_t join_channel_;
    bool supported_ = false;
    std::string source_;
    std::shared_ptr<JitModule> modules_[48];   // layout (4) x no-null-vectors (2) x carry (2) x {whole table, one page with the epilogue, list of pages} (3)
    bool carry_supported_ = false;
};

// FilterAndProject feeding a HashAggregation: the filter becomes a row mask in front of the group-by table (no row is
// materialised), and the projections that feed the aggregates are evaluated in registers inside the accumulate kernel.
class GroupedAccumulators;

class FusedAggGpu {
public:
    // aggs[k].input_channel / mask_channel index the page processor's projections
    // group_by_channels: projection indexes of the group-by keys (they must be plain column references for the fused path)
    FusedAggGpu(std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec, std::vector<tgpu_agg_spec> aggs,
                std::vector<int32_t> group_by_channels = {});
    static std::shared_ptr<FusedAggGpu> shared(const std::vector<int32_t> &input_types, const tgpu_page_processor_spec *spec,
                                               const std::vector<tgpu_agg_spec> &aggs, const std::vector<int32_t> &group_by_channels);   // see PageProcessorGpu::shared
    // raw input channels of the group-by keys (empty when a key is not a plain column reference)
    const std::vector<int> &key_inputs() const { return key_inputs_; }
    // one probe/insert launch of the group-by table with the filter fused in front and key code generated for the key schema
    void probe_groups(Context *ctx, const DevicePage &in, const struct GbhProbeLaunch &l);
    ~FusedAggGpu();
    void precompile();
    bool supported() const { return supported_; }
    bool has_filter() const { return filter_root_ >= 0; }
    // channel of the raw input page behind projection `ch` when that projection is a plain column reference, else -1
    int identity_channel(int ch) const;
    const std::vector<int32_t> &projection_types() const { return proj_types_; }
    // mask[row] = 1 iff the filter selects the row (PageFilter semantics); raises the filter's arithmetic errors
    void filter_mask(Context *ctx, const DevicePage &in, uint8_t *mask_out);
    // state[gid[row]] (+)= aggregate inputs of every row with gid >= 0
    // gids8 (optional, instead of gids): compact ids, one byte per row = group id + 1 (GroupByHashGpu::get_group_ids)
    // gate (optional): device counters of the group-by probe launch in front (GbhSpeculateFn): the kernels do nothing unless they are clean
    void accumulate(Context *ctx, const DevicePage &in, const int32_t *gids, const uint8_t *gids8, int64_t groups, GroupedAccumulators &accs,
                    const unsigned long long *gate = nullptr);
    // ONE launch per page for the steady state of few groups (fq_onepass, jit.cpp): filter + lookup among the `groups` (<= 16) groups whose keys
    // `store` holds + accumulate into lane-private LDS states; the page's totals stay pending (GroupedAccumulators::FoldScratch) until its
    // counters say no row met an unknown group.  prev = the counters of the previous one-pass page of the operator (its pending totals are
    // made final or dropped by this launch), or null.
    // `pages`: one page, or several that are taken as one sequence of rows (a blocking operator batching small pages: one launch for all).
    // host_out: Context::Signal::device (the kernel delivers its counters itself), or null.  keep: receives the buffer of page descriptors
    // a multi-page launch reads (alive until the launch has run).
    void onepass(Context *ctx, const std::vector<const DevicePage *> &pages, GroupedAccumulators &accs, const KeyCols &store, int64_t groups,
                 unsigned long long *counters, const unsigned long long *prev, int64_t blocks, unsigned long long *host_out = nullptr, BufferPtr *keep = nullptr);
    // groups a one-pass launch has LDS for: its key records (16 groups x keys x 32 B) sit next to the lane-private states
    int onepass_groups() const
    {
        const int room = 160 * 1024 - 64 - 16 * (int)key_inputs_.size() * 32;
        return per_group_bytes_ > 0 ? std::min(16, room / per_group_bytes_) : 0;
    }
    int lowcard_groups() const { return max_groups_; }   // groups whose lane-private states fit the LDS: the EXACT / ORDERED threshold of this operator
    bool can_onepass(int64_t groups) const { return can_speculate(groups) && !key_inputs_.empty() && groups <= onepass_groups(); }
    // the accumulate launch can be enqueued speculatively behind a probe launch (no error read-back of its own, lane-private LDS states)
    bool can_speculate(int64_t groups) const { return supported_ && !accumulate_can_raise_ && groups > 0 && groups <= max_groups_; }

private:
    void generate();
    void ensure_loaded();
    struct JitModule *module_for(const DevicePage &in, bool gid8 = false);   // the no-nulls specialisation when no column of the page has a null vector
    struct JitModule *module_variant(bool nulls, bool gid8);
    std::mutex mu_;
    void raise_if_error(Context *ctx, BufferPtr &err);
    std::vector<int32_t> input_types_;
    std::vector<tgpu_expr_node> nodes_;
    std::string pool_;
    int filter_root_;
    std::vector<int32_t> proj_roots_, proj_types_;
    std::vector<tgpu_agg_spec> aggs_;
    std::vector<int> key_inputs_;
    bool supported_ = false;
    bool accumulate_can_raise_ = true;   // false: no expression the accumulate kernels evaluate can raise (no error read-back)
    int n_wide_ = 0, n_cnt_ = 0, rows_slot_ = -1, per_group_bytes_ = 0, max_groups_ = 0;
    // ORDERED mode's chained kernel (fa_ordered_chain): DOUBLE sums of the operator, and when the kernel is chosen -- at most
    // kOrdChainMaxDoubles chains (one lane of wave 0 each, LDS for two tiles of them), one workgroup per group, >= kOrdChainMinRows rows per
    // group on average (below that the lane-per-group kernel has more lanes at work)
    int ord_doubles_ = 0;
    static constexpr int kOrdChainMaxDoubles = 8, kOrdChainWaves = 16;   // = TG_ORD_MAX_DOUBLES, TG_ORD_WAVES (device_agg.h)
    int wide_slot_[16], cnt_slot_[16];
    std::vector<std::vector<int>> cnt_inputs_;   // per count slot: the raw input channels its (mask, input) expressions read
    std::vector<bool> cnt_masked_;
    std::string source_;
    std::shared_ptr<JitModule> module_, module_nn_, module_g8_, module_nn_g8_;
};

std::string resource_dir();
void set_resource_dir(const std::string &dir);

}  // namespace tgpu
