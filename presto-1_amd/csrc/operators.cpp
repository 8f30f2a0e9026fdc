// operators.cpp -- the hot-path operators as host-side state machines (same names, call protocol and error behaviour as
// the reference's Java operators; the work inside addInput / getOutput runs in the gfx950 kernels).
#include "operators.h"

#include <algorithm>
#include <array>
#include <deque>
#include <map>
#include <set>

#include "kernels.h"

#include <cstdlib>

namespace tgpu {

// May an operator keep `in` by reference after add_input returned?  Library-owned pages (another operator's output, an ingested host
// page) hold their buffers; borrowed device blocks are the caller's unless it promised to leave them alone
// (tgpu_context_set_device_input_stable).
static bool page_is_retained(Context *ctx, const DevicePage &in)
{
    if (ctx->device_input_stable()) return true;
    for (const DeviceColumn &c : in.cols) {
        if (c.n > 0 && !c.values_buf) return false;
        if (c.nulls && !c.nulls_buf) return false;
        if (c.offsets && !c.offsets_buf) return false;
    }
    return true;
}

// =====================================================================================================================
// FilterAndProjectOperator: WorkProcessorOperatorAdapter protocol (M/operator/WorkProcessorOperatorAdapter.java:138-216)
// around PageProcessor (M/operator/FilterAndProjectOperator.java:56-64).  One output page per input page that selects
// at least one row; MergePages (M/operator/project/MergePages.java) is not applied (page boundaries are not part of the
// operator's contract: the reference's tests compare rows, T/operator/OperatorAssertion.java).
// =====================================================================================================================
namespace {
// dictionary-aware selection: row r is selected iff the verdict of its dictionary entry is a non-null true
__global__ void __launch_bounds__(256) dict_select_kernel(const uint8_t *verdict, const uint8_t *verdict_nulls, const int32_t *ids, int64_t n, int32_t *flags)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const int32_t e = ids[r];
        flags[r] = (verdict[e] != 0 && !(verdict_nulls && verdict_nulls[e])) ? 1 : 0;
    }
}
__global__ void __launch_bounds__(256) dict_compact_kernel(const int32_t *flags, const int32_t *rank, const int32_t *ids, int64_t n, int32_t *positions, int32_t *sel_ids)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256)
        if (flags[r]) {
            positions[rank[r]] = (int32_t)r;
            sel_ids[rank[r]] = ids[r];
        }
}
}  // namespace

class FilterAndProjectOperator : public Operator {
public:
    FilterAndProjectOperator(Context *ctx, int32_t id, std::shared_ptr<PageProcessorGpu> p) : Operator(ctx, id), processor_(std::move(p)) {}

    bool needs_input() override { return !finishing_ && !pending_; }

    void add_input(const tgpu_page *page) override
    {
        TG_CHECK_STATE(!finishing_, "Operator is already finishing");
        TG_CHECK_STATE(!pending_, "Operator still has pending output");
        if (try_dictionary(page)) return;
        run(ingest_page(ctx_, page));
    }

    // Dictionary-aware processing (M/operator/project/DictionaryAwarePageFilter.java:56-110, DictionaryAwarePageProjection.java:60-160,
    // selected by PageFunctionCompiler.java:176-212 for deterministic single-channel expressions): when every computed expression reads
    // the same channel and that channel arrives as a DictionaryBlock or RunLengthEncodedBlock, the filter and the projections are
    // evaluated once per dictionary entry (once in all for RLE) and mapped through the ids -- same rows, same values as the flat path.
    // An entry that raises (overflow, division by zero) may belong to no selected row: like the reference (which catches and
    // re-processes the block the normal way, DictionaryAwarePageProjection.java:139-152) the page then takes the flat path.
    bool try_dictionary(const tgpu_page *page)
    {
        const int c = processor_->single_input_channel();
        if (c < 0 || page == nullptr || c >= page->channel_count || getenv("TGPU_DISABLE_DICTIONARY_AWARE")) return false;
        const tgpu_block &blk = page->blocks[c];
        if (blk.encoding == TGPU_FLAT || blk.dictionary == nullptr || blk.dictionary->encoding != TGPU_FLAT) return false;
        const int64_t n = page->position_count;
        if (n == 0 || (int64_t)blk.dictionary->position_count * 2 > n) return false;   // a dictionary about as long as the page: nothing to gain
        for (int32_t ch = 0; ch < page->channel_count; ch++) TG_CHECK_ARG(page->blocks[ch].position_count == page->position_count, "block position count differs from the page's");
        DeviceColumn dict;
        BufferPtr ids;
        ingest_dictionary(ctx_, &blk, dict, ids);
        DevicePage per_entry;
        try {
            ProfileScope ps(ctx_, "dictionary_entries");
            processor_->process_dictionary(ctx_, dict, per_entry);
        }
        catch (const Error &e) {
            if (e.code == TGPU_ERR_NUMERIC_VALUE_OUT_OF_RANGE || e.code == TGPU_ERR_DIVISION_BY_ZERO || e.code == TGPU_ERR_INVALID_CAST_ARGUMENT) return false;
            throw;
        }
        dictionary_pages_++;
        const bool has_filter = processor_->has_filter();
        const int first_slot = has_filter ? 1 : 0;
        int64_t n_sel = n;
        BufferPtr positions, sel_ids = ids;
        if (has_filter) {
            const DeviceColumn &verdict = per_entry.cols[0];
            BufferPtr flags = ctx_->alloc((size_t)n * 4), rank = ctx_->alloc((size_t)n * 4), total = ctx_->alloc(8);
            const int g = (int)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx_->cu_count() * 8);
            ProfileScope ps(ctx_, "dictionary_select");
            dict_select_kernel<<<g, 256, 0, ctx_->stream()>>>((const uint8_t *)verdict.values, verdict.nulls, ids->as<int32_t>(), n, flags->as<int32_t>());
            k::exclusive_scan_i32(ctx_, flags->as<int32_t>(), rank->as<int32_t>(), n, total->as<int64_t>());
            n_sel = ctx_->read_scalar(total->as<int64_t>());
            if (n_sel == 0) return true;   // no output page (PageProcessor.java:122-124)
            if (n_sel < n) {
                positions = ctx_->alloc((size_t)n_sel * 4);
                sel_ids = ctx_->alloc((size_t)n_sel * 4);
                dict_compact_kernel<<<g, 256, 0, ctx_->stream()>>>(flags->as<int32_t>(), rank->as<int32_t>(), ids->as<int32_t>(), n, positions->as<int32_t>(), sel_ids->as<int32_t>());
                check_launch("dict_compact");
            }
        }
        DevicePage out;
        out.n = n_sel;
        std::vector<DeviceColumn> flat((size_t)page->channel_count);   // other channels are ingested only when a projection passes them through
        std::vector<bool> have((size_t)page->channel_count, false);
        for (int p = 0; p < processor_->projection_count(); p++) {
            const int ident = processor_->identity_channel(p);
            if (ident < 0) out.cols.push_back(k::gather_column(ctx_, per_entry.cols[(size_t)(first_slot + processor_->computed_slot(p))], sel_ids->as<int32_t>(), n_sel, false));
            else if (ident == c) out.cols.push_back(k::gather_column(ctx_, dict, sel_ids->as<int32_t>(), n_sel, false));
            else {
                if (!have[(size_t)ident]) {
                    flat[(size_t)ident] = ingest_block(ctx_, &page->blocks[ident]);
                    have[(size_t)ident] = true;
                }
                if (positions) out.cols.push_back(k::gather_column(ctx_, flat[(size_t)ident], positions->as<int32_t>(), n_sel, false));
                else out.cols.push_back(flat[(size_t)ident]);
            }
        }
        ctx_->sync();   // host blocks ingested above are only valid during the call
        own_borrowed_columns(ctx_, out);
        pending_ = wrap(std::move(out));
        return true;
    }
    int64_t dictionary_pages() const { return dictionary_pages_; }
    // a page of this library: identity projections that pass a block through share its (reference counted) buffers
    void add_input_owned(const DevicePage &page) override
    {
        TG_CHECK_STATE(!finishing_, "Operator is already finishing");
        TG_CHECK_STATE(!pending_, "Operator still has pending output");
        run(DevicePage(page));
    }

    std::unique_ptr<OutputPage> get_output() override { return std::move(pending_); }
    void finish() override { finishing_ = true; }
    bool is_finished() override { return finishing_ && !pending_; }
    int64_t memory_bytes() override { return pending_ ? pending_->page.size_in_bytes() : 0; }

private:
    void run(DevicePage in)
    {
        DevicePage out;
        if (!processor_->process(ctx_, in, out)) return;
        // An identity projection (InputPageProjection) may pass an input block through unchanged.  The output page must outlive
        // the call and the input page (tgpu.h ownership rule): a passed-through block that borrows the caller's device memory
        // is copied; one that came with owners (host ingest, add_input_owned) keeps them alive through its BufferPtrs.
        own_borrowed_columns(ctx_, out);
        pending_ = wrap(std::move(out));
    }

    std::shared_ptr<PageProcessorGpu> processor_;
    std::unique_ptr<OutputPage> pending_;
    int64_t dictionary_pages_ = 0;   // input pages that took the dictionary-aware path
    bool finishing_ = false;
};

// how many input pages of a FilterAndProjectOperator were processed once per dictionary entry (tests, stats)
int64_t filter_project_dictionary_pages(Operator *op)
{
    auto *p = dynamic_cast<FilterAndProjectOperator *>(op);
    TG_CHECK_ARG(p != nullptr, "not a FilterAndProjectOperator");
    return p->dictionary_pages();
}

FilterAndProjectOperatorFactory::FilterAndProjectOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> input_types,
                                                                 const tgpu_page_processor_spec *spec)
    : ctx_(ctx), operator_id_(operator_id), processor_(PageProcessorGpu::shared(input_types, spec))
{
}

std::unique_ptr<Operator> FilterAndProjectOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<FilterAndProjectOperator>(ctx_, operator_id_, processor_);
}

std::unique_ptr<OperatorFactory> FilterAndProjectOperatorFactory::duplicate()
{
    return std::unique_ptr<OperatorFactory>(new FilterAndProjectOperatorFactory(*this));   // shares the compiled page processor
}

// =====================================================================================================================
// ScanFilterAndProjectOperator, page-source flavour (M/operator/ScanFilterAndProjectOperator.java:232-287,354-397): a source operator.
// addSplit hands it the split's ConnectorPageSource; getOutput pulls the next page (ConnectorPageSourceToPages.process :368-397:
// finished / blocked / null page = yield) and runs the page processor over it.  LazyBlocks are loaded on demand, in the order the
// reference's PageProcessor touches them (PageProcessor.java:111-137): the channels the filter reads first; the channels only the
// projections read when -- and only when -- the filter selected a row (T/operator/project/TestPageProcessor.java:156-184,219-253).
// =====================================================================================================================
class ScanFilterAndProjectOperator : public Operator {
public:
    ScanFilterAndProjectOperator(Context *ctx, int32_t id, std::vector<int32_t> types, std::shared_ptr<PageProcessorGpu> p)
        : Operator(ctx, id), types_(std::move(types)), processor_(std::move(p))
    {
    }
    ~ScanFilterAndProjectOperator() override { close(); }

    bool needs_input() override { return false; }                                                                    // :188-191
    void add_input(const tgpu_page *) override { fail(TGPU_ERR_NOT_SUPPORTED, "ScanFilterAndProjectOperator is a source operator"); }

    void add_page_source(const tgpu_page_source *src)
    {
        TG_CHECK_ARG(src && src->get_next_page && src->is_finished, "page source callbacks are null");
        TG_CHECK_STATE(!no_more_splits_ && !closed_, "no more splits can be added");
        TG_CHECK_STATE(!have_source_, "Table scan split already set");                                               // :235
        source_ = *src;
        have_source_ = true;
    }
    void no_more_splits() { no_more_splits_ = true; }

    // waiting for a split, or for the page source (:214-230)
    bool is_blocked() override
    {
        if (closed_ || finishing_) return false;
        if (!have_source_) return !no_more_splits_;
        return source_.is_blocked && source_.is_blocked(source_.user) == 1;
    }

    std::unique_ptr<OutputPage> get_output() override
    {
        if (closed_ || finishing_ || !have_source_ || is_blocked()) return nullptr;
        if (source_.is_finished(source_.user) == 1) {
            release_source();
            return nullptr;
        }
        tgpu_page page{};
        const int32_t rc = source_.get_next_page(source_.user, &page);
        if (rc < 0) fail(TGPU_ERR_INTERNAL, "the page source failed");
        if (rc == 0) {                                                                                             // null page: finished or yield (:381-388)
            if (source_.is_finished(source_.user) == 1) release_source();
            return nullptr;
        }
        processed_positions_ += page.position_count;
        DevicePage out;
        if (!process(page, out)) return nullptr;
        own_borrowed_columns(ctx_, out);
        return wrap(std::move(out));
    }

    void finish() override
    {
        finishing_ = true;
        release_source();
    }
    bool is_finished() override { return closed_ || finishing_ || (no_more_splits_ && !have_source_); }
    void close() override
    {
        release_source();
        closed_ = true;
    }
    void stats(int64_t *positions, int64_t *loaded, int64_t *skipped) const
    {
        *positions = processed_positions_;
        *loaded = lazy_loaded_;
        *skipped = lazy_skipped_;
    }

private:
    void release_source()
    {
        if (have_source_ && source_.close) source_.close(source_.user);
        have_source_ = false;
    }

    DeviceColumn load_channel(const tgpu_page &page, int ch)
    {
        const tgpu_block &b = page.blocks[ch];
        if (b.encoding != TGPU_LAZY) return ingest_block(ctx_, &b);
        TG_CHECK_ARG(source_.load_block != nullptr, "the page has lazy blocks but the page source has no load_block");
        tgpu_block loaded{};
        if (source_.load_block(source_.user, ch, &loaded) < 0) fail(TGPU_ERR_INTERNAL, "the page source failed to load a lazy block");
        TG_CHECK_ARG(loaded.encoding != TGPU_LAZY && loaded.position_count == page.position_count && loaded.type == types_[(size_t)ch], "bad loaded block");
        lazy_loaded_++;
        return ingest_block(ctx_, &loaded);
    }

    bool process(const tgpu_page &page, DevicePage &out)
    {
        TG_CHECK_ARG(page.channel_count == (int32_t)types_.size(), "page channel count differs from the scan's types");
        const int64_t n = page.position_count;
        if (n == 0) return false;
        int lazy = 0;
        for (int32_t ch = 0; ch < page.channel_count; ch++) {
            TG_CHECK_ARG(page.blocks[ch].position_count == page.position_count && page.blocks[ch].type == types_[(size_t)ch], "bad block in the scanned page");
            lazy += page.blocks[ch].encoding == TGPU_LAZY ? 1 : 0;
        }
        DevicePage in;
        in.n = n;
        in.cols.resize(types_.size());
        std::vector<bool> have(types_.size(), false);
        for (size_t i = 0; i < types_.size(); i++) {   // placeholders: a channel no expression reads is never dereferenced
            in.cols[i].type = types_[i];
            in.cols[i].n = n;
        }
        auto need = [&](int ch) {
            if (have[(size_t)ch]) return;
            in.cols[(size_t)ch] = load_channel(page, ch);
            have[(size_t)ch] = true;
        };
        for (int ch : processor_->filter_channels()) need(ch);
        bool projection_only_lazy = false;
        for (int ch : processor_->projection_channels()) projection_only_lazy = projection_only_lazy || (!have[(size_t)ch] && page.blocks[ch].encoding == TGPU_LAZY);
        if (processor_->has_filter() && projection_only_lazy) {
            // would any row survive?  Only then are the projections' lazy blocks loaded (the filter runs again with them in place)
            DevicePage none;
            if (!processor_->filter_only()->process(ctx_, in, none)) {
                for (int ch : processor_->projection_channels())
                    if (!have[(size_t)ch] && page.blocks[ch].encoding == TGPU_LAZY) lazy_skipped_++;
                count_unread(page, have, lazy);
                return false;
            }
        }
        for (int ch : processor_->projection_channels()) need(ch);
        count_unread(page, have, lazy);
        ctx_->sync();   // the source's arrays are valid until its next call; the uploads have consumed them now
        return processor_->process(ctx_, in, out);
    }

    void count_unread(const tgpu_page &page, const std::vector<bool> &have, int lazy)
    {
        if (!lazy) return;
        const std::vector<int> &pc = processor_->projection_channels();
        for (int32_t ch = 0; ch < page.channel_count; ch++)   // lazy channels nobody reads at all
            if (page.blocks[ch].encoding == TGPU_LAZY && !have[(size_t)ch] && std::find(pc.begin(), pc.end(), (int)ch) == pc.end()) lazy_skipped_++;
    }

    std::vector<int32_t> types_;
    std::shared_ptr<PageProcessorGpu> processor_;
    tgpu_page_source source_{};
    bool have_source_ = false, no_more_splits_ = false, finishing_ = false, closed_ = false;
    int64_t processed_positions_ = 0, lazy_loaded_ = 0, lazy_skipped_ = 0;
};

ScanFilterAndProjectOperatorFactory::ScanFilterAndProjectOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, const tgpu_page_processor_spec *spec)
    : ctx_(ctx), operator_id_(operator_id), types_(std::move(types)), processor_(PageProcessorGpu::shared(types_, spec))
{
}

std::unique_ptr<Operator> ScanFilterAndProjectOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<ScanFilterAndProjectOperator>(ctx_, operator_id_, types_, processor_);
}

std::unique_ptr<OperatorFactory> ScanFilterAndProjectOperatorFactory::duplicate()
{
    return std::unique_ptr<OperatorFactory>(new ScanFilterAndProjectOperatorFactory(*this));
}

static ScanFilterAndProjectOperator *as_scan(Operator *op)
{
    auto *p = dynamic_cast<ScanFilterAndProjectOperator *>(op);
    TG_CHECK_ARG(p != nullptr, "not a ScanFilterAndProjectOperator");
    return p;
}
void scan_add_page_source(Operator *op, const tgpu_page_source *source) { as_scan(op)->add_page_source(source); }
void scan_no_more_splits(Operator *op) { as_scan(op)->no_more_splits(); }
void scan_stats(Operator *op, int64_t *processed_positions, int64_t *lazy_loaded, int64_t *lazy_skipped) { as_scan(op)->stats(processed_positions, lazy_loaded, lazy_skipped); }

// =====================================================================================================================
// HashAggregationOperator + InMemoryHashAggregationBuilder
// =====================================================================================================================
class HashAggregationOperator : public Operator {
public:
    HashAggregationOperator(Context *ctx, int32_t id, const HashAggregationConfig &cfg) : Operator(ctx, id), cfg_(cfg)
    {
        TG_CHECK_ARG(cfg_.group_by_types.size() == cfg_.group_by_channels.size(), "group-by types and channels differ in length");
        TG_CHECK_ARG(cfg_.step >= TGPU_STEP_SINGLE && cfg_.step <= TGPU_STEP_FINAL, "unknown aggregation step");
    }

    // HashAggregationOperator.java:367-378
    bool needs_input() override
    {
        if (finishing_) return false;
        if (builder_full()) return false;
        return true;
    }

    // :381-440 -> InMemoryHashAggregationBuilder.processPage (builder/InMemoryHashAggregationBuilder.java:139-155)
    void add_input(const tgpu_page *page) override
    {
        begin_input();
        DevicePage in = ingest_page(ctx_, page);
        process_page(in);
    }

protected:
    void begin_input()
    {
        TG_CHECK_STATE(!finishing_, "Operator is already finishing");
        TG_CHECK_STATE(!builder_full(), "Aggregation buffer is full");
        input_processed_ = true;
        ensure_builder();
    }

    // ---- small input pages are coalesced in HBM before they meet the group-by table ----------------------------------------------------------
    // A join's output pages are a fraction of its probe pages (TPCH Q3: 5 K rows out of a 2^20-row probe page), and every page costs the
    // group-by its insert / rank / publish protocol: launches and a read-back, whatever its size.  What the reference solves with MergePages
    // in front of its operators (M/operator/project/MergePages.java) happens inside the operator here: pages below kCoalesceBelowRows rows
    // are appended to a device column store (ONE launch per page, no read-back: fixed-width channels only) and processed as one page once
    // kCoalesceFlushRows rows are there -- or when the operator's state is needed (finish, output, revoke).  Rows keep their order, so
    // group ids and row-order sums are those of the unmerged stream.  Not for PARTIAL steps (their memory limit is checked per page).
    static constexpr int64_t kCoalesceBelowRows = 1 << 20, kCoalesceFlushRows = 1 << 22;
    bool coalesce(const DevicePage &in)
    {
        if (!gbh_ || cfg_.step == TGPU_STEP_PARTIAL || getenv("TGPU_DISABLE_COALESCE")) return false;
        const bool small = in.n < kCoalesceBelowRows;
        if (!small) {
            flush_coalesced();   // (order: what was buffered came first)
            return false;
        }
        for (const DeviceColumn &c : in.cols)
            if (c.type == TGPU_VARCHAR) {
                flush_coalesced();
                return false;
            }
        if (!pending_pages_) {
            std::vector<int32_t> types;
            for (const DeviceColumn &c : in.cols) types.push_back(c.type);
            pending_pages_ = std::make_unique<PagesIndexGpu>(ctx_, types);
        }
        TG_CHECK_ARG(pending_pages_->types().size() == in.cols.size(), "page channel count changed between pages");
        static const std::vector<std::array<int32_t, 2>> no_ends(64, {0, 0});
        pending_pages_->add_page(in, &no_ends);
        if (pending_pages_->position_count() >= kCoalesceFlushRows) flush_coalesced();
        return true;
    }
    void flush_coalesced()
    {
        if (!pending_pages_ || pending_pages_->position_count() == 0) return;
        std::unique_ptr<PagesIndexGpu> store = std::move(pending_pages_);
        DevicePage merged;
        merged.n = store->position_count();
        for (int ch = 0; ch < (int)store->types().size(); ch++) merged.cols.push_back(store->column(ch));
        process_page_now(merged);
    }

    void process_page(const DevicePage &in)
    {
        if (in.n == 0) return;
        if (coalesce(in)) return;
        process_page_now(in);
    }

    void process_page_now(const DevicePage &in)
    {
        const int32_t *gids = nullptr;
        BufferPtr gid_buf;
        if (gbh_) {
            std::vector<const DeviceColumn *> keys;
            for (int32_t ch : cfg_.group_by_channels) {
                TG_CHECK_ARG(ch >= 0 && ch < (int)in.cols.size(), "group-by channel out of range");
                keys.push_back(&in.cols[(size_t)ch]);
            }
            const int64_t *hashes = nullptr;
            if (cfg_.hash_channel >= 0) {
                TG_CHECK_ARG(cfg_.hash_channel < (int)in.cols.size() && in.cols[(size_t)cfg_.hash_channel].type == TGPU_BIGINT, "bad hash channel");
                hashes = (const int64_t *)in.cols[(size_t)cfg_.hash_channel].values;
            }
            gid_buf = ctx_->alloc((size_t)in.n * 4);
            gbh_->get_group_ids(keys, hashes, in.n, gid_buf->as<int32_t>());
            gids = gid_buf->as<int32_t>();
            if (hold_back(in, gid_buf, nullptr, /*lowcard_max_groups=*/0)) return;
        }
        accumulate_page(gids, in);
    }

    void accumulate_page(const int32_t *gids, const DevicePage &in)
    {
        const int64_t groups = gbh_ ? gbh_->group_count() : 1;
        if (cfg_.step == TGPU_STEP_FINAL) accs_->add_intermediate(gids, in.n, in, groups);
        else accs_->add_input(gids, in.n, in, groups);
    }

    // ---- the DOUBLE mode is decided from the first kModePrefixRows rows of the stream, however they are cut into pages (agg.h) ----------------
    // Pages that arrive before that many rows have been seen are kept (an owned copy + their group ids: they are small) and accumulated, in
    // order, once the mode is known; the page that crosses the mark tells how many groups its first rows hold.  Returns true when the page
    // has been kept back.  The group-by table itself always runs at once: group ids do not depend on the mode.
    struct HeldPage {
        DevicePage page;
        BufferPtr gids, gids8;
    };
    bool hold_back(const DevicePage &in, const BufferPtr &gids, const BufferPtr &gids8, int64_t lowcard_max_groups)
    {
        if (accs_->decided()) return false;
        const int64_t prefix = getenv("TGPU_MODE_PREFIX_ROWS") ? atoll(getenv("TGPU_MODE_PREFIX_ROWS")) : GroupedAccumulators::kModePrefixRows;   // (tests shorten it)
        if (rows_seen_ + in.n < prefix) {
            HeldPage h;
            h.page = in;
            resolve_varchar_ends(ctx_, h.page);   // (the fused path leaves the byte ranges of borrowed VARCHAR blocks unread)
            own_borrowed_columns(ctx_, h.page);
            h.gids = gids;
            h.gids8 = gids8;
            held_.push_back(std::move(h));
            rows_seen_ += in.n;
            groups_in_prefix_ = gbh_->group_count();
            return true;
        }
        const int64_t m = prefix - rows_seen_;
        const int64_t here = GroupedAccumulators::groups_among(ctx_, gids8 ? nullptr : gids->as<int32_t>(), gids8 ? gids8->as<uint8_t>() : nullptr, std::max<int64_t>(m, 1));
        accs_->decide(std::max(groups_in_prefix_, here), lowcard_max_groups);
        rows_seen_ += in.n;
        release_held();
        return false;
    }
    // the stream ended (or its state is needed) before the prefix was full: every row seen so far is the prefix
    void decide_now(int64_t lowcard_max_groups = 0)
    {
        if (!accs_ || accs_->decided() || held_.empty()) return;
        accs_->decide(gbh_ ? gbh_->group_count() : 1, lowcard_max_groups);
        release_held();
    }
    virtual void release_held()
    {
        std::vector<HeldPage> held = std::move(held_);
        held_.clear();
        for (HeldPage &h : held) accumulate_page(h.gids ? h.gids->as<int32_t>() : nullptr, h.page);
    }

public:
    // :470-518
    std::unique_ptr<OutputPage> get_output() override
    {
        if (finished_) return nullptr;
        if (finishing_) flush_coalesced();
        if (finishing_ || builder_full()) decide_now(held_lowcard_max_groups());
        if (finishing_) {
            if (!input_processed_ && cfg_.produce_default_output && cfg_.group_by_types.empty()) {
                // global aggregation without input: one row of default values (count 0, sum NULL) :481-485
                ensure_builder();
            }
            finished_ = true;
            producing_output_ = true;
            if (!runs_.empty()) merge_runs();   // SpillableHashAggregationBuilder.buildResult :193-240
            if (!builder_) return nullptr;
            std::unique_ptr<OutputPage> out = build_result();
            reset_builder();
            return out;
        }
        if (!builder_full()) return nullptr;  // only flush when finishing or full :494-497
        std::unique_ptr<OutputPage> out = build_result();
        reset_builder();
        return out;
    }

    void finish() override { finishing_ = true; }
    bool is_finished() override { return finished_; }
    // SpillableHashAggregationBuilder.updateMemory :117-128: until output is produced the builder's memory is revocable
    int64_t memory_bytes() override { return revocable() ? 0 : builder_bytes(); }
    int64_t revocable_memory_bytes() override { return revocable() ? builder_bytes() : 0; }

    // startMemoryRevoke -> spillToDisk (:160-168, :283-299): the builder's groups leave HBM as one run -- keys, raw hashes and the
    // EXACT accumulator state (GroupedAccumulators::dump), parked in host memory -- and an empty builder takes over.  The reference
    // sorts a run by hash for its merge of bounded memory; here the runs are merged through the group-by table when the output is
    // built (288 GB of HBM: the merge itself is not what memory pressure is about), and the output comes out in raw-hash order like
    // the reference's merged result.
    void start_memory_revoke() override
    {
        if (!revocable() || !builder_) return;
        spill_run();
    }
    void spill_stats(int64_t &spill_count, int64_t &spilled_bytes) override
    {
        spill_count = spill_count_;
        spilled_bytes = spilled_bytes_;
    }

protected:
    int64_t builder_bytes() const
    {
        return (gbh_ ? gbh_->estimated_size() : 0) + (accs_ ? accs_->estimated_size() : 0) + (pending_pages_ ? pending_pages_->estimated_size() : 0);
    }
    bool spillable() const { return cfg_.spill_enabled && (cfg_.step == TGPU_STEP_SINGLE || cfg_.step == TGPU_STEP_FINAL); }   // HashAggregationOperator.java:390
    bool revocable() const { return spillable() && !producing_output_; }
    struct SpilledRun {
        int64_t groups = 0;
        std::vector<HostColumn> keys;   // group-by channels, then the raw hash of every group
        GroupedAccumulators::HostStates states;
    };
    void spill_run()
    {
        flush_coalesced();
        decide_now(held_lowcard_max_groups());
        const int64_t groups = gbh_ ? gbh_->group_count() : (input_processed_ ? 1 : 0);
        if (groups > 0) {
            SpilledRun run;
            run.groups = groups;
            if (gbh_) {
                DevicePage keys = gbh_->key_page(true);
                for (const DeviceColumn &c : keys.cols) run.keys.push_back(download_column(ctx_, c));
            }
            run.states = accs_->dump(groups);
            spill_count_++;
            spilled_bytes_ += run.states.bytes();
            for (const HostColumn &h : run.keys) spilled_bytes_ += h.bytes();
            runs_.push_back(std::move(run));
        }
        reset_builder();   // rebuildHashAggregationBuilder :331-351
    }
    // mergeFromDisk :229-240 (after spilling what is still in memory, like the reference does when memory is short): every run's
    // groups are looked up / inserted in a fresh table, their states added exactly
    void merge_runs()
    {
        if (builder_) spill_run();
        ensure_builder();
        accs_->set_allow_ordered(false);
        // all runs in many-group (row-order) mode: combined like the reference, one double add per run in spill order; few-group runs keep
        // their exact limb state and merge exactly (bit-identical with and without spills)
        bool all_ordered = !runs_.empty();
        for (const SpilledRun &run : runs_) {
            bool ordered = false;
            for (const auto &a : run.states.aggs) ordered = ordered || !a.dsum.empty();
            bool has_double = false;
            for (const auto &sp : cfg_.aggs) has_double = has_double || sp.function == TGPU_AGG_SUM_DOUBLE || sp.function == TGPU_AGG_AVG_DOUBLE || sp.function == TGPU_AGG_AVG_BIGINT;
            all_ordered = all_ordered && (ordered || !has_double);
        }
        accs_->set_combine_ordered(all_ordered);
        for (const SpilledRun &run : runs_) {
            BufferPtr gid_buf;
            const int32_t *gids = nullptr;
            if (gbh_) {
                std::vector<DeviceColumn> cols;
                for (const HostColumn &h : run.keys) cols.push_back(upload_column(ctx_, h));
                std::vector<const DeviceColumn *> keys;
                for (size_t i = 0; i + 1 < cols.size(); i++) keys.push_back(&cols[i]);
                const int64_t *hashes = cfg_.hash_channel >= 0 ? (const int64_t *)cols.back().values : nullptr;
                gid_buf = ctx_->alloc((size_t)run.groups * 4);
                gbh_->get_group_ids(keys, hashes, run.groups, gid_buf->as<int32_t>());
                gids = gid_buf->as<int32_t>();
            }
            accs_->merge(gids, run.states, gbh_ ? gbh_->group_count() : 1);
        }
        runs_.clear();
        hash_sorted_output_ = gbh_ != nullptr;
    }

    void ensure_builder()
    {
        if (builder_) return;
        if (!cfg_.group_by_types.empty())
            gbh_ = std::make_unique<GroupByHashGpu>(ctx_, cfg_.group_by_types, cfg_.hash_channel >= 0, cfg_.expected_groups, allow_integer_table());
        accs_ = std::make_unique<GroupedAccumulators>(ctx_, cfg_.aggs, cfg_.step);
        accs_->set_allow_ordered(allow_ordered_accumulation());
        accs_->set_force_ordered(ctx_->double_sum_order() == TGPU_SUM_ORDER_JAVA);
        builder_ = true;
    }
    // many groups: rows sorted by group id, one lane per group adds them in row order (agg.h)
    virtual bool allow_ordered_accumulation() const { return true; }
    // the single-integer-key table (groupby_bigint.hip); an operator that brings its own probe kernels needs the generic layout
    virtual bool allow_integer_table() const { return true; }
    void reset_builder()
    {
        gbh_.reset();
        accs_.reset();
        builder_ = false;
        held_.clear();
        rows_seen_ = 0;
        groups_in_prefix_ = 0;
    }
    virtual int64_t held_lowcard_max_groups() const { return 0; }
    // InMemoryHashAggregationBuilder.isFull :208-215: only partial aggregations have a memory limit
    bool builder_full()
    {
        if (!builder_ || cfg_.step != TGPU_STEP_PARTIAL) return false;
        // the limit is on the logical size of the partial state (what the Java builder would hold: 13 B per slot at 0.75
        // fill + keys + accumulator state), not on this implementation's over-provisioned device buffers
        const int64_t groups = gbh_ ? gbh_->group_count() : 1;
        const int64_t per_group = 18 + 9 * (int64_t)cfg_.group_by_types.size() + 16 * (int64_t)cfg_.aggs.size();
        return groups * per_group > cfg_.max_partial_memory;
    }
    // buildResult :244-298: groups in group-id order; keys, [hash], aggregates
    std::unique_ptr<OutputPage> build_result()
    {
        DevicePage out;
        const int64_t groups = gbh_ ? gbh_->group_count() : 1;
        if (gbh_) out = gbh_->key_page(cfg_.hash_channel >= 0);
        out.n = groups;
        accs_->evaluate(groups, out.cols);
        if (groups == 0) return nullptr;
        if (hash_sorted_output_ && groups > 1) {
            // merged runs come out in raw-hash order (MergeHashSort.java:58-101: Long.compare of the hashes; equal hashes keep the
            // order in which the runs brought them)
            DevicePage hp;
            hp.n = groups;
            hp.cols.push_back(gbh_->key_page(true).cols.back());
            int64_t count = 0;
            BufferPtr order = TopNGpu::sorted_positions(ctx_, hp, {0}, {TGPU_SORT_ASC_NULLS_LAST}, groups, count);
            DevicePage sorted;
            sorted.n = groups;
            for (const DeviceColumn &c : out.cols) sorted.cols.push_back(k::gather_column(ctx_, c, order->as<int32_t>(), groups, false));
            out = std::move(sorted);
        }
        return wrap(std::move(out));
    }

    HashAggregationConfig cfg_;
    std::unique_ptr<GroupByHashGpu> gbh_;
    std::unique_ptr<GroupedAccumulators> accs_;
    std::vector<HeldPage> held_;
    std::unique_ptr<PagesIndexGpu> pending_pages_;
    int64_t rows_seen_ = 0, groups_in_prefix_ = 0;
    bool builder_ = false, finishing_ = false, finished_ = false, input_processed_ = false;
    std::vector<SpilledRun> runs_;
    bool producing_output_ = false, hash_sorted_output_ = false;
    int64_t spill_count_ = 0, spilled_bytes_ = 0;
};

HashAggregationOperatorFactory::HashAggregationOperatorFactory(Context *ctx, int32_t operator_id, HashAggregationConfig cfg)
    : ctx_(ctx), operator_id_(operator_id), cfg_(std::move(cfg))
{
    TG_CHECK_ARG(cfg_.group_by_types.size() == cfg_.group_by_channels.size(), "group-by types and channels differ in length");
    TG_CHECK_ARG((int)cfg_.group_by_types.size() <= kMaxKeyChannels, "at most 8 group-by channels");
    TG_CHECK_ARG((int)cfg_.aggs.size() <= kMaxAggs, "at most 16 aggregates");
    TG_CHECK_ARG(cfg_.expected_groups > 0, "expectedGroups must be positive");
    for (auto &a : cfg_.aggs) TG_CHECK_ARG(a.function >= TGPU_AGG_COUNT_ALL && a.function <= TGPU_AGG_MAX_DOUBLE, "unknown aggregate function");
}

std::unique_ptr<Operator> HashAggregationOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<HashAggregationOperator>(ctx_, operator_id_, cfg_);
}

std::unique_ptr<OperatorFactory> HashAggregationOperatorFactory::duplicate() { return std::make_unique<HashAggregationOperatorFactory>(ctx_, operator_id_, cfg_); }

// =====================================================================================================================
// HashBuilderOperator (M/operator/HashBuilderOperator.java:155-191 state machine, spill states omitted: the GPU path reports
// its memory as non-revocable and never spills)
// =====================================================================================================================
class HashBuilderOperator : public Operator {
public:
    enum class State { CONSUMING_INPUT, LOOKUP_SOURCE_BUILT, CLOSED };

    HashBuilderOperator(Context *ctx, int32_t id, const HashBuilderConfig &cfg, std::shared_ptr<LookupSourceFactory> bridge, int partition)
        : Operator(ctx, id), cfg_(cfg), bridge_(std::move(bridge)), index_(std::make_shared<PagesIndexGpu>(ctx, cfg.types)), partition_(partition)
    {
    }

    bool needs_input() override { return state_ == State::CONSUMING_INPUT; }

    // :318-350 -> PagesIndex.addPage
    void add_input(const tgpu_page *page) override
    {
        TG_CHECK_STATE(state_ == State::CONSUMING_INPUT, "Operator is already finishing");
        DevicePage in = ingest_page(ctx_, page);
        index_->add_page(in);  // copies: PagesIndex retains the build side for the table's lifetime (PagesIndex.java:221-228)
    }

    std::unique_ptr<OutputPage> get_output() override { return nullptr; }

    // :478-496 finishInput -> buildLookupSource -> lendPartitionLookupSource
    void finish() override
    {
        if (state_ != State::CONSUMING_INPUT) return;
        index_bytes_ = index_->estimated_size();
        bridge_->lend_partition(partition_, index_);   // the last partition to arrive builds the table (lendPartitionLookupSource)
        index_.reset();
        state_ = State::LOOKUP_SOURCE_BUILT;
    }

    // the operator stays alive (blocked) until the probes no longer need the table (:429-470)
    bool is_blocked() override { return state_ == State::LOOKUP_SOURCE_BUILT && !bridge_->destroyed(); }

    bool is_finished() override
    {
        if (state_ == State::LOOKUP_SOURCE_BUILT && bridge_->destroyed()) close();
        return state_ == State::CLOSED;
    }

    int64_t memory_bytes() override
    {
        if (index_) return index_->estimated_size();
        if (state_ == State::CLOSED) return 0;
        std::shared_ptr<LookupSourceGpu> s = bridge_->lookup_source();
        return s && partition_ == 0 ? s->estimated_size() : index_bytes_;   // the merged table is accounted once, on partition 0
    }

    void close() override { state_ = State::CLOSED; }

private:
    HashBuilderConfig cfg_;
    std::shared_ptr<LookupSourceFactory> bridge_;
    std::shared_ptr<PagesIndexGpu> index_;
    int partition_;
    int64_t index_bytes_ = 0;
    State state_ = State::CONSUMING_INPUT;
};

HashBuilderOperatorFactory::HashBuilderOperatorFactory(Context *ctx, int32_t operator_id, HashBuilderConfig cfg, std::shared_ptr<LookupSourceFactory> bridge)
    : ctx_(ctx), operator_id_(operator_id), cfg_(std::move(cfg)), bridge_(std::move(bridge))
{
    const int nt = (int)cfg_.types.size();
    for (int32_t t : cfg_.types) TG_CHECK_ARG(valid_type(t), "unknown type");
    for (int32_t ch : cfg_.output_channels) TG_CHECK_ARG(ch >= 0 && ch < nt, "output channel out of range");
    TG_CHECK_ARG(!cfg_.hash_channels.empty(), "hash join needs at least one join channel");
    for (int32_t ch : cfg_.hash_channels) TG_CHECK_ARG(ch >= 0 && ch < nt, "join channel out of range");
    TG_CHECK_ARG(cfg_.precomputed_hash_channel < nt, "hash channel out of range");
    for (int32_t ch : cfg_.output_channels) bridge_->build_output_types.push_back(cfg_.types[(size_t)ch]);
    bridge_->build_types = cfg_.types;
    TG_CHECK_ARG(cfg_.partition_count >= 1 && cfg_.partition_count <= 1024 && (cfg_.partition_count & (cfg_.partition_count - 1)) == 0,
                 "the build partition count must be a power of two (LocalPartitionGenerator.java:45-52)");
    const HashBuilderConfig config = cfg_;
    Context *c = ctx_;
    bridge_->set_partitioning(cfg_.partition_count, [config, c](std::vector<std::shared_ptr<PagesIndexGpu>> &parts) {
        const HashBuilderConfig &cfg = config;
        std::shared_ptr<PagesIndexGpu> index = parts[0];
        if (parts.size() > 1) {   // concatenation in partition order
            index = std::make_shared<PagesIndexGpu>(c, cfg.types);
            for (auto &part : parts) {
                if (!part || part->position_count() == 0) continue;
                DevicePage pg;
                pg.n = part->position_count();
                for (size_t i = 0; i < cfg.types.size(); i++) pg.cols.push_back(part->column((int)i));
                index->add_page(pg);
                part.reset();   // the partition's own copy goes back to the allocator as soon as it is merged
            }
        }
        auto source = std::make_shared<LookupSourceGpu>(c, index, cfg.hash_channels, cfg.precomputed_hash_channel, cfg.output_channels);
        source->build();
        return source;
    });
}

std::unique_ptr<Operator> HashBuilderOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    TG_CHECK_STATE(created_ < cfg_.partition_count, "one build operator per lookup source partition");
    return std::make_unique<HashBuilderOperator>(ctx_, operator_id_, cfg_, bridge_, created_++);
}

// ---- join filter function: the candidate pairs of a probe page, filtered --------------------------------------------------------
namespace {
// pairs are in (probe position, chain) order: count[row] = surviving pairs of the probe row
__global__ void __launch_bounds__(256) jf_count_kernel(const int32_t *probe_idx, int64_t pairs, int32_t *count)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * 256) atomicAdd(&count[probe_idx[i]], 1);
}
__global__ void __launch_bounds__(256) jf_emit_count_kernel(const int32_t *count, int64_t n, int32_t *emit)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) emit[r] = count[r] > 0 ? count[r] : 1;
}
// PROBE_OUTER output: every probe row's surviving pairs, or (row, -1) when it has none (LookupJoinOperator.java:354-361)
__global__ void __launch_bounds__(256) jf_outer_rows_kernel(const int32_t *count, const int32_t *offset, int64_t n, int32_t *out_probe, int32_t *out_build)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256)
        if (count[r] == 0) {
            out_probe[offset[r]] = (int32_t)r;
            out_build[offset[r]] = -1;
        }
}
__global__ void __launch_bounds__(256) jf_outer_pairs_kernel(const int32_t *probe_idx, const int32_t *build_idx, int64_t pairs, const int32_t *start, const int32_t *offset,
                                                             int32_t *out_probe, int32_t *out_build)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * 256) {
        const int32_t r = probe_idx[i];
        const int64_t at = (int64_t)offset[r] + (i - start[r]);   // the pair's rank among its row's survivors: pairs of a row are adjacent
        out_probe[at] = r;
        out_build[at] = build_idx[i];
    }
}
}  // namespace

// JoinHash.getJoinPosition / getNextJoinPosition with a filter function (M/operator/JoinHash.java:82-130): the positions of a key's
// chain are visited newest -> oldest and those the filter rejects are skipped; a PROBE_OUTER row whose every candidate is rejected comes
// out with a null build side.  Here: the key-equal candidate pairs (in that very order) are filtered by a generated kernel
// evaluating the predicate over (build channels gathered at the build position, probe channels gathered at the probe position).
static void apply_join_filter(Context *ctx, LookupSourceGpu &source, const JoinFilter &jf, const DevicePage &in, bool outer, BufferPtr &probe_idx, BufferPtr &build_idx,
                              int64_t &count)
{
    const int nb = (int)jf.build_types.size();
    TG_CHECK_ARG(jf.probe_types.size() == in.cols.size(), "the join filter's probe types differ from the probe page's channels");
    // the channels the predicate reads, compacted: [referenced build channels..., referenced probe channels..., probe index, build index]
    std::vector<tgpu_expr_node> nodes = jf.nodes;
    std::vector<int> remap((size_t)nb + in.cols.size(), -1);
    std::vector<int32_t> types;
    std::vector<int> sources;
    for (auto &nd : nodes)
        if (nd.kind == TGPU_EX_INPUT) {
            TG_CHECK_ARG(nd.op >= 0 && nd.op < (int)remap.size(), "join filter: input channel out of range");
            if (remap[(size_t)nd.op] < 0) {
                remap[(size_t)nd.op] = (int)types.size();
                types.push_back(nd.op < nb ? jf.build_types[(size_t)nd.op] : jf.probe_types[(size_t)(nd.op - nb)]);
                sources.push_back(nd.op);
            }
            nd.op = remap[(size_t)nd.op];
        }
    const int idx_ch = (int)types.size();
    types.push_back(TGPU_INTEGER);
    types.push_back(TGPU_INTEGER);
    tgpu_expr_node pi{}, bi{};
    pi.kind = bi.kind = TGPU_EX_INPUT;
    pi.type = bi.type = TGPU_INTEGER;
    pi.op = idx_ch;
    bi.op = idx_ch + 1;
    nodes.push_back(pi);
    nodes.push_back(bi);
    const int32_t roots[2] = {(int32_t)nodes.size() - 2, (int32_t)nodes.size() - 1};
    tgpu_page_processor_spec spec{nodes.data(), (int32_t)nodes.size(), jf.pool.data(), (int32_t)jf.pool.size(), jf.root, 2, roots};
    std::shared_ptr<PageProcessorGpu> pp = PageProcessorGpu::shared(types, &spec);

    // candidate pairs (inner semantics: the outer rows are added after the filter)
    DevicePage cand;
    cand.n = count;
    for (int src : sources) {
        if (src < nb) cand.cols.push_back(source.gather_index_channel(src, build_idx->as<int32_t>(), count));
        else cand.cols.push_back(k::gather_column(ctx, in.cols[(size_t)(src - nb)], probe_idx->as<int32_t>(), count, false));
    }
    auto idx_col = [&](const BufferPtr &b) {
        DeviceColumn c;
        c.type = TGPU_INTEGER;
        c.n = count;
        c.values_buf = b;
        c.values = b->ptr();
        return c;
    };
    cand.cols.push_back(idx_col(probe_idx));
    cand.cols.push_back(idx_col(build_idx));
    DevicePage kept;
    int64_t survivors = 0;
    if (count > 0 && pp->process(ctx, cand, kept)) {
        survivors = kept.n;
        probe_idx = kept.cols[0].values_buf;
        build_idx = kept.cols[1].values_buf;
    }
    if (!outer) {
        count = survivors;
        return;
    }
    // PROBE_OUTER / FULL_OUTER: a probe row without a surviving candidate is emitted once, build side null, at its place in the order
    const int64_t n = in.n;
    ProfileScope ps(ctx, "join_filter_outer");
    BufferPtr cnt = ctx->alloc_zero((size_t)n * 4), emit = ctx->alloc((size_t)n * 4), start = ctx->alloc((size_t)n * 4), offset = ctx->alloc((size_t)n * 4), total = ctx->alloc(16);
    const int g = (int)std::min<int64_t>(ceil_div(std::max<int64_t>(n, 1), 256), (int64_t)ctx->cu_count() * 8);
    if (survivors > 0) jf_count_kernel<<<g, 256, 0, ctx->stream()>>>(probe_idx->as<int32_t>(), survivors, cnt->as<int32_t>());
    jf_emit_count_kernel<<<g, 256, 0, ctx->stream()>>>(cnt->as<int32_t>(), n, emit->as<int32_t>());
    k::exclusive_scan_i32(ctx, cnt->as<int32_t>(), start->as<int32_t>(), n, total->as<int64_t>());
    k::exclusive_scan_i32(ctx, emit->as<int32_t>(), offset->as<int32_t>(), n, total->as<int64_t>() + 1);
    int64_t totals[2];
    ctx->download(totals, total->ptr(), 16);
    const int64_t out_count = totals[1];
    if (out_count > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "join output of one probe page cannot exceed 2 billion rows");
    BufferPtr op = ctx->alloc((size_t)out_count * 4), ob = ctx->alloc((size_t)out_count * 4);
    jf_outer_rows_kernel<<<g, 256, 0, ctx->stream()>>>(cnt->as<int32_t>(), offset->as<int32_t>(), n, op->as<int32_t>(), ob->as<int32_t>());
    if (survivors > 0)
        jf_outer_pairs_kernel<<<g, 256, 0, ctx->stream()>>>(probe_idx->as<int32_t>(), build_idx->as<int32_t>(), survivors, start->as<int32_t>(), offset->as<int32_t>(),
                                                           op->as<int32_t>(), ob->as<int32_t>());
    check_launch("join_filter_outer");
    probe_idx = op;
    build_idx = ob;
    count = out_count;
}

// PageJoiner.processProbe + LookupJoinPageBuilder.build for one probe page (M/operator/LookupJoinOperator.java:299-347,
// LookupJoinPageBuilder.java:101-131): probe columns by probe index, then the build side's output columns.
// Returns false when the page produces no output row.
static bool probe_page(Context *ctx, LookupSourceGpu &source, const DevicePage &in, const LookupJoinConfig &cfg, DevicePage &out, const JoinFilter *filter = nullptr)
{
    TG_CHECK_ARG(in.cols.size() == cfg.probe_types.size(), "probe page channel count differs from the operator's types");
    if (in.n == 0) return false;
    std::vector<const DeviceColumn *> keys;
    for (int32_t ch : cfg.probe_join_channels) keys.push_back(&in.cols[(size_t)ch]);
    const int64_t *hashes = nullptr;
    if (cfg.probe_hash_channel >= 0) {
        TG_CHECK_ARG(in.cols[(size_t)cfg.probe_hash_channel].type == TGPU_BIGINT, "probe hash channel must be BIGINT");
        hashes = (const int64_t *)in.cols[(size_t)cfg.probe_hash_channel].values;
    }
    const bool outer = cfg.join_type == TGPU_JOIN_PROBE_OUTER || cfg.join_type == TGPU_JOIN_FULL_OUTER;
    BufferPtr probe_idx, build_idx;
    int64_t count = 0;
    source.probe(keys, hashes, in.n, outer && !filter, probe_idx, build_idx, count);
    if (filter) apply_join_filter(ctx, source, *filter, in, outer, probe_idx, build_idx, count);
    if (count == 0) return false;  // no output page for this probe page (:276-283 pageBuilder.isEmpty)
    if (cfg.join_type == TGPU_JOIN_LOOKUP_OUTER || cfg.join_type == TGPU_JOIN_FULL_OUTER) source.mark_visited(build_idx->as<int32_t>(), count);   // OuterLookupSource.appendTo
    out.n = count;
    ProfileScope ps(ctx, "join_gather");
    for (int32_t ch : cfg.probe_output_channels) out.cols.push_back(k::gather_column(ctx, in.cols[(size_t)ch], probe_idx->as<int32_t>(), count, false));
    const int nb = (int)source.output_channels().size();
    for (int i = 0; i < nb; i++) out.cols.push_back(source.gather_build(i, build_idx->as<int32_t>(), count, outer));
    return true;
}

// =====================================================================================================================
// LookupJoinOperator / PageJoiner (M/operator/LookupJoinOperator.java:208-378)
// =====================================================================================================================
class LookupJoinOperator : public Operator {
public:
    LookupJoinOperator(Context *ctx, int32_t id, const LookupJoinConfig &cfg, std::shared_ptr<LookupSourceFactory> bridge)
        : Operator(ctx, id), cfg_(cfg), bridge_(std::move(bridge))
    {
        bridge_->probe_created();
    }
    ~LookupJoinOperator() override { close(); }

    // blocked on lookupSourceProviderFuture until the build side lends the table (:235-243) -- unless the probe side has ended: input is only
    // taken once the table is there, so an operator that is finishing without it has had no page and finishes without the build side
    // (TestHashJoinOperator.java:1241-1259 testInnerJoinWithBlockingLookupSourceAndEmptyProbe)
    bool is_blocked() override { return !closed_ && !finishing_ && !bridge_->lookup_source(); }
    bool needs_input() override { return !finishing_ && !pending_ && !is_blocked(); }

    void add_input(const tgpu_page *page) override
    {
        TG_CHECK_STATE(!finishing_, "Operator is already finishing");
        TG_CHECK_STATE(!pending_, "Operator still has pending output");
        std::shared_ptr<LookupSourceGpu> source = bridge_->lookup_source();
        TG_CHECK_STATE(source != nullptr, "Lookup source has not been built yet");
        DevicePage in = ingest_page(ctx_, page);
        DevicePage out;
        std::shared_ptr<const JoinFilter> filter = bridge_->join_filter();
        if (probe_page(ctx_, *source, in, cfg_, out, filter.get())) pending_ = wrap(std::move(out));
    }

    std::unique_ptr<OutputPage> get_output() override { return std::move(pending_); }
    void finish() override { finishing_ = true; }
    bool is_finished() override
    {
        bool done = finishing_ && !pending_;
        if (done) close();
        return done;
    }
    int64_t memory_bytes() override { return pending_ ? pending_->page.size_in_bytes() : 0; }
    void close() override
    {
        if (!closed_) {
            closed_ = true;
            bridge_->probe_closed();
        }
    }

private:
    LookupJoinConfig cfg_;
    std::shared_ptr<LookupSourceFactory> bridge_;
    std::unique_ptr<OutputPage> pending_;
    bool finishing_ = false, closed_ = false;
};

LookupJoinOperatorFactory::LookupJoinOperatorFactory(Context *ctx, int32_t operator_id, LookupJoinConfig cfg, std::shared_ptr<LookupSourceFactory> bridge)
    : ctx_(ctx), operator_id_(operator_id), cfg_(std::move(cfg)), bridge_(std::move(bridge))
{
    const int nt = (int)cfg_.probe_types.size();
    for (int32_t t : cfg_.probe_types) TG_CHECK_ARG(valid_type(t), "unknown type");
    TG_CHECK_ARG(!cfg_.probe_join_channels.empty(), "hash join needs at least one join channel");
    for (int32_t ch : cfg_.probe_join_channels) TG_CHECK_ARG(ch >= 0 && ch < nt, "probe join channel out of range");
    for (int32_t ch : cfg_.probe_output_channels) TG_CHECK_ARG(ch >= 0 && ch < nt, "probe output channel out of range");
    TG_CHECK_ARG(cfg_.probe_hash_channel < nt, "probe hash channel out of range");
    TG_CHECK_ARG(cfg_.join_type >= TGPU_JOIN_INNER && cfg_.join_type <= TGPU_JOIN_FULL_OUTER, "unknown join type");
    if (cfg_.join_type == TGPU_JOIN_LOOKUP_OUTER || cfg_.join_type == TGPU_JOIN_FULL_OUTER) bridge_->outer_expected();   // LookupJoinOperatorFactory.java:88-103
    bridge_->probe_factory_created();
}

std::unique_ptr<Operator> LookupJoinOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<LookupJoinOperator>(ctx_, operator_id_, cfg_, bridge_);
}

std::unique_ptr<OperatorFactory> LookupJoinOperatorFactory::duplicate() { return std::make_unique<LookupJoinOperatorFactory>(ctx_, operator_id_, cfg_, bridge_); }

void LookupJoinOperatorFactory::no_more_operators()
{
    if (closed_) return;
    closed_ = true;
    bridge_->no_more_probes();
}

// =====================================================================================================================
// LookupOuterOperator (M/operator/LookupOuterOperator.java:32-235): once every probe operator of a LOOKUP_OUTER / FULL_OUTER join
// is done (the outer position iterator future, PartitionedLookupSourceFactory.java:259-297), the build rows nobody matched come
// out in build-position order (OuterLookupSource.java:146-190 OuterPositionIterator -- rows with a null key included: they can never
// match), probe-side output channels null (:188-197), build output channels behind them.  One page (the reference cuts it at
// the page builder's size limit).
// =====================================================================================================================
class LookupOuterOperator : public Operator {
public:
    LookupOuterOperator(Context *ctx, int32_t id, const std::vector<int32_t> &probe_output_types, std::shared_ptr<LookupSourceFactory> bridge)
        : Operator(ctx, id), probe_output_types_(probe_output_types), bridge_(std::move(bridge))
    {
    }
    ~LookupOuterOperator() override { close(); }

    bool is_blocked() override { return !closed_ && !(bridge_->probes_finished() && bridge_->lookup_source()); }
    bool needs_input() override { return false; }                                                        // :163-167
    void add_input(const tgpu_page *) override { fail(TGPU_ERR_NOT_SUPPORTED, "LookupOuterOperator takes no input"); }   // :169-173

    std::unique_ptr<OutputPage> get_output() override
    {
        if (closed_ || is_blocked()) return nullptr;
        std::shared_ptr<LookupSourceGpu> source = bridge_->lookup_source();
        BufferPtr positions;
        int64_t count = 0;
        source->unvisited_positions(positions, count);
        std::unique_ptr<OutputPage> out;
        if (count > 0) {
            DevicePage page;
            page.n = count;
            // null probe channels: a one-position dummy column gathered at position -1 (= null) for every output row
            BufferPtr minus_one = ctx_->alloc((size_t)count * 4);
            k::fill_i32(ctx_, minus_one->as<int32_t>(), -1, count);
            for (int32_t t : probe_output_types_) {
                DeviceColumn dummy;
                dummy.type = t;
                dummy.n = 1;
                dummy.values_buf = ctx_->alloc_zero(8);
                dummy.values = dummy.values_buf->ptr();
                if (t == TGPU_VARCHAR) {
                    dummy.offsets_buf = ctx_->alloc_zero(8);
                    dummy.offsets = dummy.offsets_buf->as<int32_t>();
                }
                page.cols.push_back(k::gather_column(ctx_, dummy, minus_one->as<int32_t>(), count, true));
            }
            const int nb = (int)source->output_channels().size();
            for (int i = 0; i < nb; i++) page.cols.push_back(source->gather_build(i, positions->as<int32_t>(), count, false));
            out = wrap(std::move(page));
        }
        close();   // :219-221
        return out;
    }

    void finish() override { close(); }                    // :150-153
    bool is_finished() override { return closed_; }
    void close() override
    {
        if (!closed_) {
            closed_ = true;
            bridge_->outer_done();   // onClose: the build side may release the table now
        }
    }

private:
    std::vector<int32_t> probe_output_types_;
    std::shared_ptr<LookupSourceFactory> bridge_;
    bool closed_ = false;
};

LookupOuterOperatorFactory::LookupOuterOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> probe_output_types, std::shared_ptr<LookupSourceFactory> bridge)
    : ctx_(ctx), operator_id_(operator_id), probe_output_types_(std::move(probe_output_types)), bridge_(std::move(bridge))
{
    for (int32_t t : probe_output_types_) TG_CHECK_ARG(valid_type(t), "unknown type");
    bridge_->outer_expected();
}

std::unique_ptr<Operator> LookupOuterOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    TG_CHECK_STATE(!created_, "Only one outer operator can be created");   // LookupOuterOperator.java:86-90 (one per lifespan)
    created_ = true;
    return std::make_unique<LookupOuterOperator>(ctx_, operator_id_, probe_output_types_, bridge_);
}

// =====================================================================================================================
// FilterAndProject fused into the probe
// =====================================================================================================================
class FusedFilterProjectJoinOperator : public Operator {
public:
    FusedFilterProjectJoinOperator(Context *ctx, int32_t id, const LookupJoinConfig &cfg, std::shared_ptr<LookupSourceFactory> bridge,
                                   std::shared_ptr<PageProcessorGpu> processor, std::shared_ptr<FusedProbeGpu> fused)
        : Operator(ctx, id), cfg_(cfg), bridge_(std::move(bridge)), processor_(std::move(processor)), fused_(std::move(fused))
    {
        bridge_->probe_created();
    }
    ~FusedFilterProjectJoinOperator() override { close(); }

    bool is_blocked() override { return !closed_ && !finishing_ && !bridge_->lookup_source(); }   // (as LookupJoinOperator: a finishing probe never waits)
    bool needs_input() override { return !finishing_ && ready_.empty() && (int)inflight_.size() <= kDepth && !is_blocked(); }

    // Pages of up to kAsyncBelowRows (2^25) rows are probed asynchronously, kDepth pages deep: add_input prepares pass 1 of the new page and -- once
    // kDepth pages are in flight -- pass 2 of the OLDEST one, whose totals reached the host (through its signal slot) while its successor ran,
    // and puts both into ONE launch (FusedProbeGpu::launch_pair: two latency-bound grids side by side); get_output then hands out the oldest
    // page's output.  So a page's output appears kDepth add_inputs later -- or at finish(), or when the driver polls get_output twice without
    // bringing input (a slow source must not keep finished work back).  The Operator contract allows exactly this (getOutput() may return
    // null whenever it likes; Driver.processInternal polls).  It needs the input pages to stay alive in between: library-owned pages (another
    // operator's output, an ingested host page) are kept by reference, borrowed device blocks only qualify under
    // tgpu_context_set_device_input_stable.  An expression error of a page is raised by the call that completes it.
    static constexpr int kDepth = 2;
    static constexpr int64_t kAsyncBelowRows = 1ll << 25;

    void add_input(const tgpu_page *page) override { add_page(page, nullptr); }
    void add_input_owned(const DevicePage &page) override { add_page(nullptr, &page); }   // (keeps the page's buffers: it may stay in flight)

    // Small pages additionally SHARE launches: pages below kBatchBelowRows rows are collected (by reference) until kBatchRows rows or
    // kBatchPages pages have come together and are probed as one sequence of rows -- one launch, one output page, rows in page order (the
    // kernels' multi variant, FusedProbeGpu::begin over a page list).  A launch costs ~15 us of dependent round trips whatever it covers;
    // a 2^20-row page streams in 3.  What is collected goes out like what is in flight: when enough has come together, at finish(), or when
    // the driver polls twice without bringing a page.
    static constexpr int64_t kBatchBelowRows = 1ll << 22, kBatchRows = 1ll << 24;
    static constexpr int kBatchPages = 64;

    void add_page(const tgpu_page *page, const DevicePage *owned)
    {
        TG_CHECK_STATE(!finishing_, "Operator is already finishing");
        TG_CHECK_STATE(ready_.empty() && (int)inflight_.size() <= kDepth, "Operator still has pending output");
        std::shared_ptr<LookupSourceGpu> source = bridge_->lookup_source();
        TG_CHECK_STATE(source != nullptr, "Lookup source has not been built yet");
        DevicePage in = owned ? DevicePage(*owned) : ingest_page(ctx_, page);
        polls_since_input_ = 0;
        if (in.n == 0) return;
        IntTableView tv;
        std::shared_ptr<const JoinFilter> filter = bridge_->join_filter();   // a join filter function runs on the unfused composition
        const bool fused_ok = !filter && fused_->supported() && cfg_.probe_join_channels.size() == 1 && source->int_table(tv) && tv.links == nullptr &&
                              tv.key_type == fused_->projection_types()[(size_t)cfg_.probe_join_channels[0]] && getenv("TGPU_DISABLE_FUSION") == nullptr;
        if (fused_ok) {
            const bool async = in.n <= kAsyncBelowRows && page_is_retained(ctx_, in) && getenv("TGPU_DISABLE_ASYNC_JOIN") == nullptr;
            if (!async) {
                flush_batch();
                complete_all();
                InFlight f = describe(source);
                f.pending = fused_->begin(ctx_, in, *source, f.outer, f.need_positions);
                f.pages.push_back(std::move(in));
                inflight_.push_back(std::move(f));
                complete_all();
                return;
            }
            const bool batching = in.n < kBatchBelowRows && getenv("TGPU_DISABLE_PROBE_BATCHING") == nullptr;
            const int64_t tiles = (in.n + batch_tile_rows() - 1) / batch_tile_rows();
            if (!batch_.empty() && (!batching || batch_source_ != source || (batch_tiles_ + tiles) * batch_tile_rows() > FusedProbeGpu::multi_page_row_limit())) flush_batch();
            batch_source_ = source;
            batch_rows_ += in.n;
            batch_tiles_ += tiles;
            batch_.push_back(std::move(in));
            if (!batching || batch_rows_ >= batch_rows_target() || (int)batch_.size() >= kBatchPages) flush_batch();
            return;
        }
        flush_batch();
        complete_all();
        // unfused composition: FilterAndProject, then the probe
        DevicePage mid, out;
        if (!processor_->process(ctx_, in, mid)) return;
        probe_rows_ += mid.n;
        if (probe_page(ctx_, *source, mid, cfg_, out, filter.get())) ready_.push_back(wrap(std::move(out)));
    }

    std::unique_ptr<OutputPage> get_output() override
    {
        // collected / in flight stays that way while the driver keeps bringing pages; a second poll without input in between, or finish(), completes it
        if (ready_.empty() && (!inflight_.empty() || !batch_.empty()) && !finishing_) polls_since_input_++;
        if (ready_.empty() && (finishing_ || polls_since_input_ >= 2)) {
            flush_batch();
            while (ready_.empty() && !inflight_.empty()) complete_oldest(nullptr);
        }
        if (ready_.empty()) return nullptr;
        std::unique_ptr<OutputPage> out = std::move(ready_.front());
        ready_.pop_front();
        return out;
    }
    void finish() override { finishing_ = true; }
    bool is_finished() override
    {
        bool done = finishing_ && ready_.empty() && inflight_.empty() && batch_.empty();
        if (done) close();
        return done;
    }
    int64_t memory_bytes() override
    {
        int64_t b = 0;
        for (const auto &o : ready_) b += o->page.size_in_bytes();
        for (const InFlight &f : inflight_)
            for (const DevicePage &pg : f.pages) b += pg.size_in_bytes();
        for (const DevicePage &pg : batch_) b += pg.size_in_bytes();
        return b;
    }
    int64_t probe_rows() const { return probe_rows_; }
    void close() override
    {
        if (!closed_) {
            closed_ = true;
            for (InFlight &f : inflight_) fused_->cancel(ctx_, f.pending);
            inflight_.clear();
            batch_.clear();
            bridge_->probe_closed();
        }
    }

private:
    struct InFlight {
        std::vector<DevicePage> pages;   // one page, or the pages of a common launch
        std::shared_ptr<FusedProbeGpu::Pending> pending;
        std::shared_ptr<LookupSourceGpu> source;
        std::vector<DeviceColumn> build_cols;
        bool outer = false, track = false, need_positions = false, gather_fused = false;
    };
    static int64_t batch_tile_rows() { return FusedProbeGpu::multi_page_row_limit() / kFjMultiMaxTiles; }
    static int64_t batch_rows_target()
    {
        const char *e = getenv("TGPU_PROBE_BATCH_ROWS");   // (kernel studies)
        return e ? std::max<int64_t>(1, atoll(e)) : kBatchRows;
    }
    InFlight describe(const std::shared_ptr<LookupSourceGpu> &source) const
    {
        InFlight f;
        f.source = source;
        f.outer = cfg_.join_type == TGPU_JOIN_PROBE_OUTER || cfg_.join_type == TGPU_JOIN_FULL_OUTER;
        f.track = cfg_.join_type == TGPU_JOIN_LOOKUP_OUTER || cfg_.join_type == TGPU_JOIN_FULL_OUTER;
        f.need_positions = f.track || !source->output_channels().empty();
        // fixed-width build output channels are gathered by the probe's emit pass itself (no launch of their own)
        const int nb = (int)source->output_channels().size();
        f.gather_fused = nb > 0 && nb <= kFjMaxBuildCols;
        for (int i = 0; i < nb && f.gather_fused; i++) {
            f.build_cols.push_back(source->build_column(i));
            f.gather_fused = f.build_cols.back().type != TGPU_VARCHAR && f.build_cols.back().values != nullptr;
        }
        return f;
    }
    // what has been collected goes into the pipeline: pass 1 prepared, launched together with pass 2 of the oldest launch in flight when
    // kDepth are in flight and the two can pair, else on its own
    void flush_batch()
    {
        if (batch_.empty()) return;
        InFlight f = describe(batch_source_);
        f.pages = std::move(batch_);
        batch_.clear();
        batch_rows_ = batch_tiles_ = 0;
        std::vector<const DevicePage *> list;
        for (const DevicePage &pg : f.pages) list.push_back(&pg);
        f.pending = fused_->begin(ctx_, list, *f.source, f.outer, f.need_positions, /*launch=*/false);
        if (!f.pending) {
            // no signal slot for a launch over several pages (many operators of this context are inside one): page by page, synchronously
            complete_all();
            for (DevicePage &pg : f.pages) {
                InFlight one = describe(f.source);
                one.pending = fused_->begin(ctx_, pg, *f.source, f.outer, f.need_positions);
                one.pages.push_back(std::move(pg));
                inflight_.push_back(std::move(one));
                complete_all();
            }
            return;
        }
        if ((int)inflight_.size() >= kDepth) {
            try {
                complete_oldest(&f.pending);
            } catch (...) {
                fused_->cancel(ctx_, f.pending);
                throw;
            }
        }
        fused_->launch_probe(ctx_, f.pending);   // (no-op when the pair launch carried it)
        inflight_.push_back(std::move(f));
    }
    void complete_all()
    {
        while (!inflight_.empty()) complete_oldest(nullptr);
    }
    // pass 2 of the oldest launch in flight; with `partner` (prepared, not yet launched) in one launch with the partner's pass 1
    void complete_oldest(const std::shared_ptr<FusedProbeGpu::Pending> *partner)
    {
        InFlight f = std::move(inflight_.front());
        inflight_.pop_front();
        std::vector<DeviceColumn> probe_out, build_out;
        BufferPtr build_idx;
        int64_t count = 0, selected = 0;
        fused_->finish(ctx_, f.pending, f.pages[0], probe_out, build_idx, count, selected, f.gather_fused ? &f.build_cols : nullptr, f.gather_fused ? &build_out : nullptr, /*launch=*/false);
        probe_rows_ += selected;
        if (count == 0) return;
        if (partner && fused_->can_pair(f.pending, *partner)) fused_->launch_pair(ctx_, f.pending, *partner);
        else fused_->launch_emit(ctx_, f.pending);
        if (f.track) f.source->mark_visited(build_idx->as<int32_t>(), count);
        DevicePage out;
        out.n = count;
        out.cols = std::move(probe_out);
        if (f.gather_fused) for (DeviceColumn &c : build_out) out.cols.push_back(std::move(c));
        else {
            ProfileScope ps(ctx_, "join_gather");
            const int nb = (int)f.source->output_channels().size();
            for (int i = 0; i < nb; i++) out.cols.push_back(f.source->gather_build(i, build_idx->as<int32_t>(), count, f.outer));
        }
        ready_.push_back(wrap(std::move(out)));
    }

    LookupJoinConfig cfg_;
    std::shared_ptr<LookupSourceFactory> bridge_;
    std::shared_ptr<PageProcessorGpu> processor_;
    std::shared_ptr<FusedProbeGpu> fused_;
    std::deque<InFlight> inflight_;
    std::deque<std::unique_ptr<OutputPage>> ready_;
    std::vector<DevicePage> batch_;   // pages waiting for their common launch
    std::shared_ptr<LookupSourceGpu> batch_source_;
    int64_t batch_rows_ = 0, batch_tiles_ = 0;
    int polls_since_input_ = 0;
    int64_t probe_rows_ = 0;
    bool finishing_ = false, closed_ = false;
};

FusedFilterProjectJoinOperatorFactory::FusedFilterProjectJoinOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> input_types,
                                                                             const tgpu_page_processor_spec *spec, LookupJoinConfig cfg,
                                                                             std::shared_ptr<LookupSourceFactory> bridge)
    : ctx_(ctx), operator_id_(operator_id), cfg_(std::move(cfg)), bridge_(std::move(bridge))
{
    processor_ = PageProcessorGpu::shared(input_types, spec);
    cfg_.probe_types = processor_->output_types();
    const int nt = (int)cfg_.probe_types.size();
    TG_CHECK_ARG(!cfg_.probe_join_channels.empty(), "hash join needs at least one join channel");
    for (int32_t ch : cfg_.probe_join_channels) TG_CHECK_ARG(ch >= 0 && ch < nt, "probe join channel out of range");
    for (int32_t ch : cfg_.probe_output_channels) TG_CHECK_ARG(ch >= 0 && ch < nt, "probe output channel out of range");
    TG_CHECK_ARG(cfg_.probe_hash_channel < nt, "probe hash channel out of range");
    TG_CHECK_ARG(cfg_.join_type >= TGPU_JOIN_INNER && cfg_.join_type <= TGPU_JOIN_FULL_OUTER, "unknown join type");
    if (cfg_.join_type == TGPU_JOIN_LOOKUP_OUTER || cfg_.join_type == TGPU_JOIN_FULL_OUTER) bridge_->outer_expected();
    fused_ = FusedProbeGpu::shared(input_types, spec, cfg_.probe_join_channels[0], cfg_.probe_output_channels);
    bridge_->probe_factory_created();
}

std::unique_ptr<Operator> FusedFilterProjectJoinOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<FusedFilterProjectJoinOperator>(ctx_, operator_id_, cfg_, bridge_, processor_, fused_);
}

std::unique_ptr<OperatorFactory> FusedFilterProjectJoinOperatorFactory::duplicate()
{
    auto f = std::unique_ptr<FusedFilterProjectJoinOperatorFactory>(new FusedFilterProjectJoinOperatorFactory(*this));
    f->closed_ = false;
    bridge_->probe_factory_created();
    return f;
}

void FusedFilterProjectJoinOperatorFactory::no_more_operators()
{
    if (closed_) return;
    closed_ = true;
    bridge_->no_more_probes();
}

// =====================================================================================================================
// FilterAndProject fused into the hash aggregation: the filter becomes a row mask in front of the group-by table and the
// aggregates' input projections are evaluated inside the accumulate kernel (jit.h FusedAggGpu); no row is materialised.
// =====================================================================================================================
class FusedFilterProjectAggregationOperator : public HashAggregationOperator {
public:
    FusedFilterProjectAggregationOperator(Context *ctx, int32_t id, const HashAggregationConfig &cfg, std::shared_ptr<PageProcessorGpu> processor,
                                          std::shared_ptr<FusedAggGpu> fused)
        : HashAggregationOperator(ctx, id, cfg), processor_(std::move(processor)), fused_(std::move(fused))
    {
    }

    bool uses_fused_kernels() const { return fused_->supported() && !fused_->key_inputs().empty() && cfg_.step != TGPU_STEP_FINAL && getenv("TGPU_DISABLE_FUSION") == nullptr; }
    bool allow_integer_table() const override { return !uses_fused_kernels(); }

    void add_input(const tgpu_page *page) override { add_page(page, nullptr); }
    // a page of this library keeps its buffers through the DevicePage's owners: it may wait for a common launch or be re-run after the call
    // (the default, a borrowed view of the page, would only be safe under the caller's device_input_stable promise -- which is about the
    // CALLER's memory, not about an output page it releases right after handing it on)
    void add_input_owned(const DevicePage &page) override { add_page(nullptr, &page); }

    void add_page(const tgpu_page *page, const DevicePage *owned)
    {
        confirm_pending(kOnepassDepth - 1);   // (errors and dirty pages of earlier one-pass launches surface here at the latest)
        begin_input();
        const bool fused_ok = gbh_ && uses_fused_kernels();
        // the fused kernels address VARCHAR bytes through the offsets alone: the byte ranges of borrowed device columns stay unread
        DevicePage in = owned ? DevicePage(*owned) : ingest_page(ctx_, page, /*resolve_varchar=*/!fused_ok);
        if (in.n == 0) return;
        if (!fused_ok) {
            drain_onepass();
            // unfused composition: FilterAndProject, then the aggregation
            DevicePage mid;
            if (processor_->process(ctx_, in, mid)) process_page(mid);
            return;
        }
        // A table-sized page does not need two passes either: its first rows go through the insert protocol in three slices -- they bring
        // the groups and decide the DOUBLE mode -- and if the group set has settled by then (it has for the few-group aggregations this path
        // is for) the REST of the page is one one-pass launch: every input byte read once instead of the key columns twice.  The rest is
        // judged like any one-pass launch: a row of an unknown group makes it dirty and it is re-run through the two-launch path.
        if (in.n >= kSplitAboveRows && pending_.empty() && batch_.empty() && clean_streak_ < kOnepassAfter && retained(in) && getenv("TGPU_DISABLE_ONEPASS") == nullptr &&
            getenv("TGPU_DISABLE_SPECULATION") == nullptr && getenv("TGPU_DISABLE_PAGE_SPLIT") == nullptr) {
            static constexpr int64_t kSlices[3] = {1 << 18, 1 << 18, 1 << 20};
            int64_t at = 0;
            for (int64_t len : kSlices) {
                process_fused(slice_of(in, at, len));
                at += len;
            }
            DevicePage rest = slice_of(in, at, in.n - at);
            if (onepass_ready(rest)) {
                batch_rows_ += rest.n;
                batch_.push_back(std::move(rest));
                launch_onepass();
            } else {
                drain_onepass();
                process_fused(rest);
            }
            return;
        }
        if (onepass_ready(in)) {
            // the operator is blocking (nothing leaves it before finish): small pages are collected, by reference, into one launch
            batch_rows_ += in.n;
            batch_.push_back(std::move(in));
            if (batch_rows_ >= onepass_batch_rows() || (int)batch_.size() >= kOnepassBatchPages) launch_onepass();
            return;
        }
        drain_onepass();
        process_fused(in);
    }

    // a finishing / revoking / closing operator has no page in flight
    void finish() override
    {
        drain_onepass();
        HashAggregationOperator::finish();
    }
    std::unique_ptr<OutputPage> get_output() override
    {
        if (finishing_) drain_onepass();   // (the driver asks after every addInput: a page in flight stays in flight)
        return HashAggregationOperator::get_output();
    }
    void start_memory_revoke() override
    {
        drain_onepass();
        HashAggregationOperator::start_memory_revoke();
    }
    // (the builder's bytes + the library-owned pages this operator keeps alive: collected for a common launch, or launched and not yet confirmed)
    int64_t memory_bytes() override
    {
        int64_t b = HashAggregationOperator::memory_bytes();
        for (const DevicePage &pg : batch_) b += owned_bytes(pg);
        for (const PendingPage &p : pending_)
            for (const DevicePage &pg : p.pages) b += owned_bytes(pg);
        return b;
    }

private:
    // the two-launch path: group probe (insert protocol when a page brings new groups) + accumulate, one read-back per page
    void process_fused(const DevicePage &in)
    {
        // the filter is fused in front of the group-by table (rows it rejects get group id -1), the key code is generated
        // for the key schema, and the raw hash is computed in the kernel (a $hashvalue channel equals it by construction)
        std::vector<const DeviceColumn *> keys;
        for (int raw : fused_->key_inputs()) keys.push_back(&in.cols[(size_t)raw]);
        // few groups: the ids travel as one byte per row between the two fused kernels (int32 only if the table outgrows a byte)
        BufferPtr gids = ctx_->alloc((size_t)in.n * 4), gids8 = ctx_->alloc((size_t)in.n);
        GbhProbeFn probe = [&](const GbhProbeLaunch &l) { fused_->probe_groups(ctx_, in, l); };
        // steady state (every group of the page exists already, few groups): the accumulate launch goes out behind the probe launch,
        // gated on the probe's counters, and runs while the host waits for them (groupby.h GbhSpeculateFn)
        const int64_t groups_before = gbh_->group_count();
        // (while the DOUBLE mode is still open -- the stream's first rows, agg.h kModePrefixRows -- nothing is accumulated behind the probe)
        const bool may_speculate = fused_->can_speculate(groups_before) && !accs_->force_ordered() && accs_->decided() && getenv("TGPU_DISABLE_SPECULATION") == nullptr;
        GbhSpeculateFn speculate = [&](const unsigned long long *counters) {
            fused_->accumulate(ctx_, in, nullptr, gids8->as<uint8_t>(), groups_before, *accs_, counters);
        };
        bool speculated = false;
        const bool compact = gbh_->get_group_ids(keys, nullptr, in.n, gids->as<int32_t>(), nullptr, /*inline_hash=*/true, &probe, gids8->as<uint8_t>(),
                                                 may_speculate ? &speculate : nullptr, &speculated);
        clean_streak_ = speculated ? clean_streak_ + 1 : 0;   // a speculated page met no new group: the group set is settling
        if (speculated) return;
        if (hold_back(in, compact ? BufferPtr() : gids, compact ? gids8 : BufferPtr(), fused_->lowcard_groups())) return;
        fused_->accumulate(ctx_, in, compact ? nullptr : gids->as<int32_t>(), compact ? gids8->as<uint8_t>() : nullptr, gbh_->group_count(), *accs_);
    }
    int64_t held_lowcard_max_groups() const override { return uses_fused_kernels() ? fused_->lowcard_groups() : 0; }
    void release_held() override
    {
        if (!uses_fused_kernels()) {
            HashAggregationOperator::release_held();
            return;
        }
        std::vector<HeldPage> held = std::move(held_);
        held_.clear();
        for (HeldPage &h : held)
            fused_->accumulate(ctx_, h.page, h.gids ? h.gids->as<int32_t>() : nullptr, h.gids8 ? h.gids8->as<uint8_t>() : nullptr, gbh_->group_count(), *accs_);
    }

    // ---- one launch per page, no read-back in front of the next one (FusedAggGpu::onepass) ------------------------------------------------
    // Entered once the group set has settled (kOnepassAfter consecutive pages without a new group, at most 16 groups).  The host reads a
    // page's counters one call LATER (while its successor runs): clean = nothing to do, the successor's launch made the page's totals
    // final; dirty (a row met a group the launch did not know) = the totals were dropped on the device and the page is re-run here through
    // the insert protocol -- possible because the page is still there: library-owned pages (another operator's output, an ingested host
    // page) are kept by reference, borrowed device blocks only qualify under tgpu_context_set_device_input_stable.  Sums are exact
    // (order-independent), so re-running a page after its successors does not change a bit; new groups still get their ids in page order
    // because a successor that met one of them is dirty itself and is re-run after it.  An expression error of page i is raised by the call
    // that confirms it (the next add_input, finish or get_output).
    // Pages below kOnepassBatchRows (2^24) rows wait (by reference) until that many rows or kOnepassBatchPages pages have come together and go out as
    // ONE launch over the list of pages (fq_onepass_multi): a launch costs ~18 us whatever it covers, 1.3 M rows' worth of streaming.  A dirty
    // launch re-runs its pages one by one in page order, an expression error in a launch of several pages likewise (the re-run raises the
    // error of the first failing page, like the reference).
    static constexpr int kOnepassDepth = 2, kOnepassAfter = 2, kOnepassBatchPages = 64;
    static constexpr int64_t kSplitAboveRows = 1ll << 23;   // pages from here on are split into three leading slices + the rest (add_page)
    static int64_t owned_bytes(const DevicePage &pg)
    {
        for (const DeviceColumn &c : pg.cols)
            if (c.n > 0 && !c.values_buf) return 0;   // borrowed device blocks are the embedding's memory, not this operator's
        return pg.size_in_bytes();
    }
    DevicePage slice_of(const DevicePage &in, int64_t at, int64_t len) const
    {
        DevicePage out;
        out.n = len;
        for (const DeviceColumn &c : in.cols) out.cols.push_back(k::region_of(ctx_, c, at, len));   // views: the buffers are shared
        return out;
    }
    static constexpr int64_t kOnepassBatchRows = 1ll << 24;
    static int64_t onepass_batch_rows()
    {
        const char *e = getenv("TGPU_ONEPASS_BATCH_ROWS");
        return e ? std::max<int64_t>(1, atoll(e)) : kOnepassBatchRows;
    }
    struct PendingPage {
        std::vector<DevicePage> pages;
        BufferPtr descriptors;
        unsigned long long *counters;
        Context::AsyncRead read;
        Context::Signal signal;
    };
    bool retained(const DevicePage &in) const { return page_is_retained(ctx_, in); }
    bool onepass_ready(const DevicePage &in) const
    {
        const bool disabled = getenv("TGPU_DISABLE_ONEPASS") != nullptr || getenv("TGPU_DISABLE_SPECULATION") != nullptr;
        return !disabled && clean_streak_ >= kOnepassAfter && fused_->can_onepass(gbh_->group_count()) && !accs_->force_ordered() && accs_->decided() && !accs_->ordered() &&
               retained(in);
    }
    void launch_onepass()
    {
        if (batch_.empty()) return;
        PendingPage p;
        p.pages = std::move(batch_);
        batch_.clear();
        batch_rows_ = 0;
        p.counters = gbh_->counter_set();
        const unsigned long long *prev = last_onepass_counters_;
        onepass_blocks_ = ctx_->cu_count();   // one workgroup per CU (the lane-private states fill the LDS), one row of pending totals each
        std::vector<const DevicePage *> list;
        for (const DevicePage &pg : p.pages) list.push_back(&pg);
        // the kernel hands its counters to the host itself (a signal slot); with every slot taken they are copied behind it
        p.signal = ctx_->begin_signal();
        try {
            fused_->onepass(ctx_, list, *accs_, gbh_->key_store_view(), gbh_->group_count(), p.counters, prev, onepass_blocks_, p.signal.device, &p.descriptors);
        } catch (...) {
            ctx_->abandon_signal(p.signal);
            // (nothing was launched: the pages go the two-launch way, after whatever is in flight)
            std::vector<DevicePage> pages = std::move(p.pages);
            confirm_pending(0);
            for (const DevicePage &pg : pages) process_fused(pg);
            return;
        }
        if (p.signal.slot < 0) p.read = ctx_->begin_read(p.counters, 64);
        last_onepass_counters_ = p.counters;
        pending_.push_back(std::move(p));
        confirm_pending(kOnepassDepth);
    }
    // confirms launches, oldest first, until at most `keep` are in flight
    void confirm_pending(size_t keep)
    {
        while (pending_.size() > keep) {
            PendingPage p = std::move(pending_.front());
            pending_.pop_front();
            unsigned long long ctr[8] = {};
            if (p.signal.slot >= 0) {
                unsigned long long w[Context::kSignalWords];
                ctx_->finish_signal(p.signal, w);
                ctr[0] = w[0];
                ctr[2] = w[1];
                ctr[7] = w[2];
            } else ctx_->finish_read(p.read, ctr);
            const bool last = pending_.empty();
            const bool dirty = ctr[0] != 0 || ctr[2] != 0 || ctr[7] != ~0ull;
            if (last) {
                // nobody launched behind it: its pending totals are still on the device, and the host decides
                accs_->resolve_pending(onepass_blocks_, !dirty);
                last_onepass_counters_ = nullptr;
            }
            if (p.pages.size() == 1) raise_expression_error(ctr[7]);
            if (dirty) {
                clean_streak_ = 0;
                // (the launches behind it ran against the same group set: each is judged by its own counters)
                for (const DevicePage &pg : p.pages) process_fused(pg);
            }
        }
    }
    // nothing collected, nothing in flight
    void flush_onepass()
    {
        launch_onepass();
        confirm_pending(0);
    }
    void drain_onepass() { flush_onepass(); }

    std::vector<DevicePage> batch_;   // pages waiting for their common launch
    int64_t batch_rows_ = 0;
    std::deque<PendingPage> pending_;
    const unsigned long long *last_onepass_counters_ = nullptr;
    int64_t onepass_blocks_ = 0;
    int clean_streak_ = 0;

private:
    std::shared_ptr<PageProcessorGpu> processor_;
    std::shared_ptr<FusedAggGpu> fused_;
};

FusedFilterProjectAggregationOperatorFactory::FusedFilterProjectAggregationOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> input_types,
                                                                                           const tgpu_page_processor_spec *spec, HashAggregationConfig cfg)
    : ctx_(ctx), operator_id_(operator_id), cfg_(std::move(cfg))
{
    processor_ = PageProcessorGpu::shared(input_types, spec);
    const std::vector<int32_t> &pt = processor_->output_types();
    TG_CHECK_ARG(cfg_.group_by_types.size() == cfg_.group_by_channels.size(), "group-by types and channels differ in length");
    for (size_t i = 0; i < cfg_.group_by_channels.size(); i++) {
        const int32_t ch = cfg_.group_by_channels[i];
        TG_CHECK_ARG(ch >= 0 && ch < (int)pt.size() && pt[(size_t)ch] == cfg_.group_by_types[i], "group-by channel / type does not match the projections");
    }
    TG_CHECK_ARG(cfg_.hash_channel < (int)pt.size(), "hash channel out of range");
    TG_CHECK_ARG(cfg_.expected_groups > 0, "expectedGroups must be positive");
    fused_ = FusedAggGpu::shared(input_types, spec, cfg_.aggs, cfg_.group_by_channels);
}

std::unique_ptr<Operator> FusedFilterProjectAggregationOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<FusedFilterProjectAggregationOperator>(ctx_, operator_id_, cfg_, processor_, fused_);
}

std::unique_ptr<OperatorFactory> FusedFilterProjectAggregationOperatorFactory::duplicate()
{
    return std::unique_ptr<OperatorFactory>(new FusedFilterProjectAggregationOperatorFactory(*this));
}

// =====================================================================================================================
// TopNOperator (M/operator/TopNOperator.java:135-225 over TopNProcessor.java:45-105): consumes pages until finish(), then emits
// the n first rows in sort order as one page
// =====================================================================================================================
class TopNOperator : public Operator {
public:
    TopNOperator(Context *ctx, int32_t id, const std::vector<int32_t> &types, int64_t n, const std::vector<int32_t> &sort_channels,
                 const std::vector<int32_t> &sort_orders)
        : Operator(ctx, id), n_(n), top_(ctx, types, n, sort_channels, sort_orders)
    {
    }

    // :195-199 (n == 0: the operator is finished from the start and never wants input, :154-156)
    bool needs_input() override { return n_ > 0 && !finishing_; }

    void add_input(const tgpu_page *page) override
    {
        TG_CHECK_STATE(needs_input(), "Operator is already finishing");
        DevicePage in = ingest_page(ctx_, page);
        top_.add_page(in);
    }

    std::unique_ptr<OutputPage> get_output() override
    {
        if (n_ == 0 || !finishing_ || finished_) return nullptr;
        finished_ = true;
        DevicePage out = top_.result();
        if (out.n == 0) return nullptr;
        return wrap(std::move(out));
    }

    void finish() override { finishing_ = true; }
    bool is_finished() override { return n_ == 0 || finished_; }
    int64_t memory_bytes() override { return top_.estimated_size(); }

private:
    int64_t n_;
    TopNGpu top_;
    bool finishing_ = false, finished_ = false;
};

TopNOperatorFactory::TopNOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, int64_t n, std::vector<int32_t> sort_channels,
                                         std::vector<int32_t> sort_orders)
    : ctx_(ctx), operator_id_(operator_id), types_(std::move(types)), sort_channels_(std::move(sort_channels)), sort_orders_(std::move(sort_orders)), n_(n)
{
    TopNGpu check(ctx_, types_, n_, sort_channels_, sort_orders_);   // argument validation up front
    (void)check;
}

std::unique_ptr<Operator> TopNOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<TopNOperator>(ctx_, operator_id_, types_, n_, sort_channels_, sort_orders_);
}

std::unique_ptr<OperatorFactory> TopNOperatorFactory::duplicate() { return std::make_unique<TopNOperatorFactory>(ctx_, operator_id_, types_, n_, sort_channels_, sort_orders_); }

// =====================================================================================================================
// OrderByOperator (M/operator/OrderByOperator.java:160-300): PagesIndex.addPage per input page, one sort at finish()
// =====================================================================================================================
class OrderByOperator : public Operator {
public:
    OrderByOperator(Context *ctx, int32_t id, const std::vector<int32_t> &types, const std::vector<int32_t> &output_channels,
                    const std::vector<int32_t> &sort_channels, const std::vector<int32_t> &sort_orders)
        : Operator(ctx, id), types_(types), output_channels_(output_channels), sort_channels_(sort_channels), sort_orders_(sort_orders), index_(ctx, types)
    {
    }

    bool needs_input() override { return !finishing_; }

    void add_input(const tgpu_page *page) override
    {
        TG_CHECK_STATE(!finishing_, "Operator is already finishing");
        DevicePage in = ingest_page(ctx_, page);
        TG_CHECK_ARG(in.cols.size() == types_.size(), "page channel count does not match the operator's types");
        for (size_t i = 0; i < types_.size(); i++) TG_CHECK_ARG(in.cols[i].type == types_[i], "page channel type does not match the operator's types");
        index_.add_page(in);
    }

    std::unique_ptr<OutputPage> get_output() override
    {
        if (!finishing_ || finished_) return nullptr;
        finished_ = true;
        DevicePage all;
        all.n = index_.position_count();
        if (all.n == 0) return nullptr;
        for (size_t i = 0; i < types_.size(); i++) all.cols.push_back(index_.column((int)i));
        int64_t count = 0;
        BufferPtr pos = TopNGpu::sorted_positions(ctx_, all, sort_channels_, sort_orders_, all.n, count);
        DevicePage out;
        out.n = count;
        for (int32_t ch : output_channels_) out.cols.push_back(k::gather_column(ctx_, all.cols[(size_t)ch], pos->as<int32_t>(), count, false));
        return wrap(std::move(out));
    }

    void finish() override { finishing_ = true; }
    bool is_finished() override { return finished_ || (finishing_ && index_.position_count() == 0); }
    int64_t memory_bytes() override { return index_.estimated_size(); }

private:
    std::vector<int32_t> types_, output_channels_, sort_channels_, sort_orders_;
    PagesIndexGpu index_;
    bool finishing_ = false, finished_ = false;
};

OrderByOperatorFactory::OrderByOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, std::vector<int32_t> output_channels,
                                               std::vector<int32_t> sort_channels, std::vector<int32_t> sort_orders)
    : ctx_(ctx), operator_id_(operator_id), types_(std::move(types)), output_channels_(std::move(output_channels)), sort_channels_(std::move(sort_channels)),
      sort_orders_(std::move(sort_orders))
{
    TopNGpu check(ctx_, types_, 1, sort_channels_, sort_orders_);   // validates types / sort channels / sort orders
    (void)check;
    for (int32_t ch : output_channels_) TG_CHECK_ARG(ch >= 0 && ch < (int)types_.size(), "output channel out of range");
}

std::unique_ptr<Operator> OrderByOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<OrderByOperator>(ctx_, operator_id_, types_, output_channels_, sort_channels_, sort_orders_);
}

std::unique_ptr<OperatorFactory> OrderByOperatorFactory::duplicate()
{
    return std::make_unique<OrderByOperatorFactory>(ctx_, operator_id_, types_, output_channels_, sort_channels_, sort_orders_);
}


// =====================================================================================================================
// DynamicFilterSourceOperator (M/operator/DynamicFilterSourceOperator.java:145-425).  Pages pass through unchanged (:375-381);
// per filter channel the operator keeps the set of distinct values (TypedSet -> a GroupByHash per channel here) while no
// channel exceeds max_distinct_values and the collected blocks stay within max_filter_size_in_bytes (:238-262); beyond that it
// falls back to min / max per orderable channel (:264-289, 291-338) as long as at most min_max_collection_limit rows were seen,
// and to "all" after that (:283-289).  finish() publishes the domains (:383-424): distinct non-null non-NaN values, or a
// [min, max] range, NONE for a channel that only saw nulls, ALL otherwise.
// Deviations, both on the safe side of an advisory filter (any superset of the build values is a valid dynamic filter):
// the size test uses the reference's BLOCK accounting of the collected values ((width + 1) or (length + 5) bytes per value)
// instead of TypedSet's JVM retained size.  min / max is kept for every orderable type of the path but DOUBLE, like the reference
// (:187-190): BIGINT / INTEGER / DATE / BOOLEAN by a reduction kernel, VARCHAR by selecting the first row in ascending and in
// descending order (topn.h: the type's comparison, bytes as unsigned) among the page's values and the running pair.
// =====================================================================================================================
namespace {
__global__ void __launch_bounds__(256) df_minmax_kernel(ColView col, int64_t n, long long *minmax /* [min, max, any non-null] */)
{
    long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        if (col.nulls && col.nulls[r]) continue;
        const long long v = col.type == TGPU_BIGINT ? ((const long long *)col.values)[r]
                            : (col.type == TGPU_BOOLEAN ? (long long)(((const unsigned char *)col.values)[r] != 0) : (long long)((const int *)col.values)[r]);
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const long long l2 = __shfl_down(lo, d, 64), h2 = __shfl_down(hi, d, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0 && lo <= hi) {
        atomicMin(&minmax[0], lo);
        atomicMax(&minmax[1], hi);
        minmax[2] = 1;
    }
}
// keep[i] = the distinct value i is neither null nor NaN (convertToDomain, :402-417)
__global__ void __launch_bounds__(256) df_keep_kernel(ColView col, int64_t n, int32_t *keep)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        bool ok = !(col.nulls && col.nulls[r]);
        if (ok && col.type == TGPU_DOUBLE) {
            const double v = ((const double *)col.values)[r];
            ok = v == v;
        }
        keep[r] = ok ? 1 : 0;
    }
}
__global__ void __launch_bounds__(256) df_compact_kernel(const int32_t *keep, const int32_t *rank, int64_t n, int32_t *out)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256)
        if (keep[r]) out[rank[r]] = (int32_t)r;
}
}  // namespace

class DynamicFilterSourceOperator : public Operator {
public:
    DynamicFilterSourceOperator(Context *ctx, int32_t id, const std::vector<int32_t> &types, const std::vector<int32_t> &channels, int32_t max_distinct, int64_t max_size,
                                int32_t min_max_limit)
        : Operator(ctx, id), types_(types), channels_(channels), max_distinct_(max_distinct), max_size_(max_size), min_max_limit_(min_max_limit)
    {
        for (int32_t ch : channels_) {
            const int32_t t = types_[(size_t)ch];
            sets_.push_back(std::make_unique<GroupByHashGpu>(ctx, std::vector<int32_t>{t}, false, 1024));
            const bool orderable = min_max_limit_ > 0 && t != TGPU_DOUBLE;   // :187-190 (orderable, DOUBLE / REAL left out because of NaN)
            if (orderable) min_max_channels_.push_back((int)sets_.size() - 1);
        }
        collecting_sets_ = true;
        collecting_min_max_ = !min_max_channels_.empty();
        if (collecting_min_max_) {
            minmax_ = ctx_->alloc(channels_.size() * 24);
            std::vector<long long> init;
            for (size_t k = 0; k < channels_.size(); k++) init.insert(init.end(), {0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1, 0});
            ctx_->upload(minmax_->ptr(), init.data(), init.size() * 8);
            ctx_->sync();
        }
    }

    bool needs_input() override { return !current_ && !finished_; }   // :208-211

    void add_input(const tgpu_page *page) override { collect(ingest_page(ctx_, page), true); }
    void add_input_owned(const DevicePage &page) override { collect(DevicePage(page), false); }

    std::unique_ptr<OutputPage> get_output() override { return std::move(current_); }   // :375-381

    void finish() override { finished_ = true; }   // the domains are read through dynamic_filter_result (:383-400)
    bool is_finished() override { return finished_ && !current_; }
    int64_t memory_bytes() override
    {
        int64_t s = current_ ? current_->page.size_in_bytes() : 0;
        for (auto &g : sets_) s += g ? g->estimated_size() : 0;
        return s;
    }

    void result(int32_t k, int32_t *kind, std::unique_ptr<OutputPage> *values, int64_t *min, int64_t *max)
    {
        TG_CHECK_STATE(finished_, "the dynamic filter is available after finish()");
        TG_CHECK_ARG(k >= 0 && k < (int)channels_.size(), "filter channel out of range");
        *kind = 0;
        *min = *max = 0;
        if (collecting_sets_) {   // :393-399 convertToDomain
            DevicePage keys = sets_[(size_t)k]->key_page(false);
            const int64_t n = keys.n;
            DevicePage out;
            out.n = 0;
            if (n > 0) {
                BufferPtr keep = ctx_->alloc((size_t)n * 4), rank = ctx_->alloc((size_t)n * 4), pos = ctx_->alloc((size_t)n * 4), total = ctx_->alloc(8);
                df_keep_kernel<<<(int)std::min<int64_t>(ceil_div(n, 256), 1024), 256, 0, ctx_->stream()>>>(view_of(keys.cols[0]), n, keep->as<int32_t>());
                k::exclusive_scan_i32(ctx_, keep->as<int32_t>(), rank->as<int32_t>(), n, total->as<int64_t>());
                df_compact_kernel<<<(int)std::min<int64_t>(ceil_div(n, 256), 1024), 256, 0, ctx_->stream()>>>(keep->as<int32_t>(), rank->as<int32_t>(), n, pos->as<int32_t>());
                check_launch("df_compact");
                out.n = ctx_->read_scalar(total->as<int64_t>());
                out.cols.push_back(k::gather_column(ctx_, keys.cols[0], pos->as<int32_t>(), out.n, false));
            }
            else out.cols.push_back(keys.cols[0]);
            *kind = 1;
            *values = wrap(std::move(out));
            return;
        }
        if (!collecting_min_max_) return;   // ALL (:386-390)
        bool is_mm = false;
        for (int c : min_max_channels_) is_mm = is_mm || c == k;
        if (!is_mm) return;                 // a channel without min / max collection is left out of the tuple domain = ALL
        if (types_[(size_t)channels_[(size_t)k]] == TGPU_VARCHAR) {
            auto it = varchar_min_max_.find(k);
            if (it == varchar_min_max_.end()) {
                *kind = 3;   // no value was ever seen
                return;
            }
            DevicePage pair;   // rows: min, max (both null when every value was null: ASC / DESC NULLS LAST put a null first only then)
            pair.n = 2;
            pair.cols.push_back(it->second);
            uint8_t null_flags[2] = {0, 0};
            if (it->second.nulls) ctx_->download(null_flags, it->second.nulls, 2);
            if (null_flags[0]) {
                *kind = 3;
                return;
            }
            *kind = 2;
            *values = wrap(std::move(pair));
            return;
        }
        long long mm[3];
        ctx_->download(mm, minmax_->as<long long>() + 3 * k, 24);
        if (!mm[2]) {
            *kind = 3;   // all values were null: Domain.none (:366-369)
            return;
        }
        *kind = 2;
        *min = mm[0];
        *max = mm[1];
    }

private:
    void collect(DevicePage in, bool borrowed)
    {
        TG_CHECK_STATE(needs_input(), "DynamicFilterSourceOperator: addInput() may not be called after finish() or with a pending page");
        TG_CHECK_ARG(in.cols.size() == types_.size(), "page channel count does not match the operator's types");
        for (size_t i = 0; i < types_.size(); i++) TG_CHECK_ARG(in.cols[i].type == types_[i], "page channel type does not match the operator's types");
        const int64_t n = in.n;
        if (collecting_sets_) {   // :236-262
            min_max_limit_left_sub(n);
            int64_t size = 0, most = 0;
            BufferPtr gids = ctx_->alloc((size_t)(n > 0 ? n : 1) * 4);
            for (size_t k = 0; k < channels_.size(); k++) {
                const DeviceColumn &c = in.cols[(size_t)channels_[k]];
                if (n > 0) sets_[k]->get_group_ids({&c}, nullptr, n, gids->as<int32_t>());
                const int64_t d = sets_[k]->group_count();
                most = std::max(most, d);
                DevicePage keys = sets_[k]->key_page(false);
                size += keys.cols[0].type == TGPU_VARCHAR ? keys.cols[0].pool_bytes + 5 * d : (int64_t)(type_width(keys.cols[0].type) + 1) * d;
            }
            if (most > max_distinct_ || size > max_size_) too_large();
        }
        else if (collecting_min_max_) {   // :218-233
            min_max_limit_left_sub(n);
            if (min_max_limit_left_ < 0) collecting_min_max_ = false;   // handleMinMaxCollectionLimitExceeded: ALL
            else
                for (int k : min_max_channels_) update_min_max(k, in.cols[(size_t)channels_[(size_t)k]]);
        }
        // the page itself goes on unchanged; a borrowed (device-resident) input is copied once so that it outlives the call
        if (borrowed) {
            bool needs_copy = false;
            for (auto &c : in.cols) needs_copy = needs_copy || (c.n > 0 && !c.values_buf);
            if (needs_copy) {
                PagesIndexGpu copy(ctx_, types_);
                copy.add_page(in);
                DevicePage owned;
                owned.n = in.n;
                for (size_t i = 0; i < types_.size(); i++) owned.cols.push_back(copy.column((int)i));
                in = std::move(owned);
            }
        }
        current_ = wrap(std::move(in));
    }

    void min_max_limit_left_sub(int64_t n)
    {
        if (!limit_started_) {
            min_max_limit_left_ = min_max_limit_;
            limit_started_ = true;
        }
        min_max_limit_left_ -= n;
    }

    void update_min_max(int k, const DeviceColumn &c)
    {
        if (c.n <= 0) return;
        if (c.type == TGPU_VARCHAR) {
            // candidates = the running (min, max) pair + this page's values; the new pair = the first row in ascending and the first in
            // descending order, nulls last (updateMinMaxValues :300-345 compares with the type's comparison operator)
            PagesIndexGpu cand(ctx_, std::vector<int32_t>{TGPU_VARCHAR});
            auto it = varchar_min_max_.find(k);
            if (it != varchar_min_max_.end()) {
                DevicePage prev;
                prev.n = 2;
                prev.cols.push_back(it->second);
                cand.add_page(prev);
            }
            DevicePage pg;
            pg.n = c.n;
            pg.cols.push_back(c);
            cand.add_page(pg);
            DevicePage all;
            all.n = cand.position_count();
            all.cols.push_back(cand.column(0));
            int64_t cnt = 0;
            BufferPtr lo = TopNGpu::sorted_positions(ctx_, all, {0}, {TGPU_SORT_ASC_NULLS_LAST}, 1, cnt);
            BufferPtr hi = TopNGpu::sorted_positions(ctx_, all, {0}, {TGPU_SORT_DESC_NULLS_LAST}, 1, cnt);
            BufferPtr both = ctx_->alloc(8);
            HIP_CHECK(hipMemcpyAsync(both->ptr(), lo->ptr(), 4, hipMemcpyDeviceToDevice, ctx_->stream()));
            HIP_CHECK(hipMemcpyAsync(both->as<int32_t>() + 1, hi->ptr(), 4, hipMemcpyDeviceToDevice, ctx_->stream()));
            varchar_min_max_[k] = k::gather_column(ctx_, all.cols[0], both->as<int32_t>(), 2, false);
            return;
        }
        df_minmax_kernel<<<(int)std::min<int64_t>(ceil_div(c.n, 256), (int64_t)ctx_->cu_count() * 2), 256, 0, ctx_->stream()>>>(view_of(c), c.n, minmax_->as<long long>() + 3 * k);
        check_launch("df_minmax");
    }

    void too_large()   // handleTooLargePredicate (:264-281)
    {
        if (min_max_channels_.empty() || min_max_limit_left_ < 0) collecting_min_max_ = false;   // ALL
        else
            for (int k : min_max_channels_) {   // min / max of what was collected so far = of every row seen so far
                DevicePage keys = sets_[(size_t)k]->key_page(false);
                update_min_max(k, keys.cols[0]);
            }
        collecting_sets_ = false;
        for (auto &g : sets_) g.reset();
    }

    std::vector<int32_t> types_, channels_;
    int32_t max_distinct_;
    int64_t max_size_;
    int32_t min_max_limit_;
    int64_t min_max_limit_left_ = 0;
    bool limit_started_ = false, collecting_sets_ = false, collecting_min_max_ = false, finished_ = false;
    std::vector<std::unique_ptr<GroupByHashGpu>> sets_;
    std::vector<int> min_max_channels_;
    BufferPtr minmax_;
    std::map<int, DeviceColumn> varchar_min_max_;   // filter channel -> 2 rows (min, max)
    std::unique_ptr<OutputPage> current_;
};

DynamicFilterSourceOperatorFactory::DynamicFilterSourceOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, std::vector<int32_t> channels,
                                                                       int32_t max_distinct_values, int64_t max_filter_size_in_bytes, int32_t min_max_collection_limit)
    : ctx_(ctx), operator_id_(operator_id), types_(std::move(types)), channels_(std::move(channels)), max_distinct_(max_distinct_values),
      min_max_limit_(min_max_collection_limit), max_size_(max_filter_size_in_bytes)
{
    for (int32_t t : types_) TG_CHECK_ARG(valid_type(t), "unknown type");
    std::set<int32_t> seen;
    for (int32_t ch : channels_) {
        TG_CHECK_ARG(ch >= 0 && ch < (int)types_.size(), "filter channel out of range");
        TG_CHECK_ARG(seen.insert(ch).second, "duplicate channel indices are not allowed");   // :105-106
    }
    TG_CHECK_ARG(max_distinct_ >= 0 && max_size_ >= 0, "limits must not be negative");
}

std::unique_ptr<Operator> DynamicFilterSourceOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<DynamicFilterSourceOperator>(ctx_, operator_id_, types_, channels_, max_distinct_, max_size_, min_max_limit_);
}

std::unique_ptr<OperatorFactory> DynamicFilterSourceOperatorFactory::duplicate()
{
    // DynamicFilterSourceOperator.java:131-135: "duplicate() is not supported for DynamicFilterSourceOperatorFactory"
    fail(TGPU_ERR_NOT_SUPPORTED, "duplicate() is not supported for DynamicFilterSourceOperatorFactory");
}

void dynamic_filter_result(Operator *op, int32_t k, int32_t *kind, std::unique_ptr<OutputPage> *values, int64_t *min, int64_t *max)
{
    auto *p = dynamic_cast<DynamicFilterSourceOperator *>(op);
    TG_CHECK_ARG(p != nullptr, "not a DynamicFilterSourceOperator");
    p->result(k, kind, values, min, max);
}

void Operator::add_input_owned(const DevicePage &page)
{
    std::vector<tgpu_block> blocks(page.cols.size());
    for (size_t i = 0; i < page.cols.size(); i++) {
        const DeviceColumn &c = page.cols[i];
        blocks[i] = tgpu_block{c.type, TGPU_FLAT, TGPU_DEVICE, (int32_t)c.n, c.values, c.nulls, c.offsets, nullptr, nullptr};
    }
    tgpu_page p{(int32_t)page.n, (int32_t)blocks.size(), blocks.data()};
    add_input(&p);
}

// =====================================================================================================================
// MergePages (M/operator/project/MergePages.java:86-190, MergePagesTransformation.process): the transformation the reference
// puts behind every PageProcessor, as an operator.  A page with at least min_row_count rows or min_page_size_in_bytes bytes
// passes through untouched (after whatever is buffered); smaller pages are appended to a buffer in HBM that is flushed when it
// reaches max_page_size_in_bytes (PageBuilder.isFull, S/PageBuilder.java:126-129) or at finish.  Sizes are the reference's
// accounting: (width + 1) bytes per fixed-width cell, length + 5 per VARCHAR cell (S/block/LongArrayBlock.java:69,
// VariableWidthBlock.java:121-124 and the matching BlockBuilder.addBytes calls).  In front of the GPU operators the thresholds
// are set to tens or hundreds of MB (DESIGN.md "Page granularity"); the reference's 1 MB cap on min_page_size_in_bytes
// (MergePages.java:58,107) is therefore not enforced.
// =====================================================================================================================
class MergePagesOperator : public Operator {
public:
    MergePagesOperator(Context *ctx, int32_t id, const std::vector<int32_t> &types, int64_t min_page_size, int32_t min_row_count, int64_t max_page_size)
        : Operator(ctx, id), types_(types), min_page_size_(min_page_size), max_page_size_(max_page_size), min_row_count_(min_row_count)
    {
    }

    bool needs_input() override { return !finishing_ && output_.empty(); }

    void add_input(const tgpu_page *page) override { merge(ingest_page(ctx_, page)); }
    void add_input_owned(const DevicePage &page) override { merge(DevicePage(page)); }   // shares the page's buffers

    void merge(DevicePage in)
    {
        TG_CHECK_STATE(needs_input(), "Operator does not need input");
        TG_CHECK_ARG(in.cols.size() == types_.size(), "page channel count does not match the operator's types");
        for (size_t i = 0; i < types_.size(); i++) TG_CHECK_ARG(in.cols[i].type == types_[i], "page channel type does not match the operator's types");
        std::vector<std::array<int32_t, 2>> ends;
        const int64_t size = in.n >= min_row_count_ ? 0 : java_size_in_bytes(in, ends);   // (only needed to classify a page with few rows)
        if (in.n >= min_row_count_ || size >= min_page_size_) {   // :145-157
            flush();
            output_.push_back(owned(std::move(in)));
            return;
        }
        if (in.n == 0) return;
        if (!buffer_) buffer_ = std::make_unique<PagesIndexGpu>(ctx_, types_);
        buffer_->add_page(in, &ends);   // :159 appendPage
        buffered_size_ += size;
        if (buffered_size_ >= max_page_size_ || buffer_->position_count() == 0x7fffffffLL) flush();   // :161-163
    }

    std::unique_ptr<OutputPage> get_output() override
    {
        if (output_.empty()) return nullptr;
        DevicePage p = std::move(output_.front());
        output_.pop_front();
        return wrap(std::move(p));
    }

    void finish() override
    {
        if (!finishing_) flush();   // :134-142
        finishing_ = true;
    }
    bool is_finished() override { return finishing_ && output_.empty(); }
    int64_t memory_bytes() override
    {
        int64_t s = buffer_ ? buffer_->estimated_size() : 0;
        for (auto &p : output_) s += p.size_in_bytes();
        return s;
    }

private:
    // Page.getSizeInBytes of flat blocks.  The byte range of a VARCHAR channel is known on the host for pages that came through the
    // host ingest; for device-resident input it comes back in one batched read (the only synchronisation of the per-page path)
    int64_t java_size_in_bytes(const DevicePage &p, std::vector<std::array<int32_t, 2>> &ends)
    {
        int64_t s = 0;
        ends.assign(p.cols.size(), {0, 0});
        std::vector<Context::Transfer> reads;
        for (size_t i = 0; i < p.cols.size(); i++) {
            const DeviceColumn &c = p.cols[i];
            if (c.type != TGPU_VARCHAR) s += (int64_t)(type_width(c.type) + 1) * p.n;
            else if (c.pool_exact) ends[i] = {c.pool_first, (int32_t)c.pool_bytes};
            else if (p.n > 0) {
                reads.push_back({&ends[i][0], c.offsets, 4});
                reads.push_back({&ends[i][1], c.offsets + p.n, 4});
            }
        }
        if (!reads.empty()) ctx_->download_batch(reads);
        for (size_t i = 0; i < p.cols.size(); i++)
            if (p.cols[i].type == TGPU_VARCHAR) s += ((int64_t)ends[i][1] - ends[i][0]) + 5 * p.n;
        return s;
    }

    // a page that passes through must outlive the call: device-resident input is borrowed (common.h), so it is copied once
    DevicePage owned(DevicePage &&p)
    {
        bool borrowed = false;
        for (auto &c : p.cols) borrowed = borrowed || (c.n > 0 && !c.values_buf);
        if (!borrowed) return std::move(p);
        PagesIndexGpu copy(ctx_, types_);
        copy.add_page(p);
        DevicePage out;
        out.n = p.n;
        for (size_t i = 0; i < types_.size(); i++) out.cols.push_back(copy.column((int)i));
        return out;
    }

    void flush()
    {
        if (!buffer_ || buffer_->position_count() == 0) return;
        DevicePage out;
        out.n = buffer_->position_count();
        for (size_t i = 0; i < types_.size(); i++) out.cols.push_back(buffer_->column((int)i));
        output_.push_back(std::move(out));
        buffer_.reset();   // the flushed columns keep the buffers alive
        buffered_size_ = 0;
    }

    std::vector<int32_t> types_;
    int64_t min_page_size_, max_page_size_, buffered_size_ = 0;
    int32_t min_row_count_;
    bool finishing_ = false;
    std::unique_ptr<PagesIndexGpu> buffer_;
    std::deque<DevicePage> output_;
};

MergePagesOperatorFactory::MergePagesOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, int64_t min_page_size_in_bytes, int32_t min_row_count,
                                                     int64_t max_page_size_in_bytes)
    : ctx_(ctx), operator_id_(operator_id), types_(std::move(types)), min_page_size_(min_page_size_in_bytes), max_page_size_(max_page_size_in_bytes),
      min_row_count_(min_row_count)
{
    // MergePages.java:102-106
    TG_CHECK_ARG(min_page_size_ >= 0, "minPageSizeInBytes must be greater or equal than zero");
    TG_CHECK_ARG(min_row_count_ >= 0, "minRowCount must be greater or equal than zero");
    TG_CHECK_ARG(max_page_size_ > 0, "maxPageSizeInBytes must be greater than zero");
    TG_CHECK_ARG(max_page_size_ >= min_page_size_, "maxPageSizeInBytes must be greater or equal than minPageSizeInBytes");
    for (int32_t t : types_) TG_CHECK_ARG(valid_type(t), "unknown channel type");
}

std::unique_ptr<Operator> MergePagesOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<MergePagesOperator>(ctx_, operator_id_, types_, min_page_size_, min_row_count_, max_page_size_);
}

std::unique_ptr<OperatorFactory> MergePagesOperatorFactory::duplicate()
{
    return std::make_unique<MergePagesOperatorFactory>(ctx_, operator_id_, types_, min_page_size_, min_row_count_, max_page_size_);
}

// =====================================================================================================================
// PartitionedOutputOperator (M/operator/PartitionedOutputOperator.java; PagePartitioner.partitionPage :406-426)
//   position -> every partition when (replicatesAnyRow and no row has been replicated yet) or the null channel is null there,
//   else -> partitionFunction.getPartition = (rawHash & 0x7fff...) % partitionCount (HashGenerator.java:24-35) of the
//   precomputed hash channel or of the partition channels' InterpretedHashGenerator hash.
// Rows reach each partition in input order.  The reference appends them to per-partition PageBuilders and flushes full ones;
// here every input page is flushed as it is partitioned (page boundaries are not part of the exchange contract).
// =====================================================================================================================
namespace {
__global__ void __launch_bounds__(256) replicate_flags_kernel(const uint8_t *null_channel_nulls, int64_t n, int replicate_first, uint8_t *out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = (uint8_t)(((null_channel_nulls && null_channel_nulls[i]) || (replicate_first && i == 0)) ? 1 : 0);
}
}  // namespace

class PartitionedOutputOperator : public Operator {
public:
    PartitionedOutputOperator(Context *ctx, int32_t id, const std::vector<int32_t> &types, const std::vector<int32_t> &partition_channels, int32_t hash_channel,
                              int32_t partition_count, bool replicates_any_row, int32_t null_channel, bool local_function)
        : Operator(ctx, id), types_(types), partition_channels_(partition_channels), hash_channel_(hash_channel), partition_count_(partition_count),
          null_channel_(null_channel), replicates_any_row_(replicates_any_row), local_function_(local_function)
    {
    }

    bool needs_input() override { return !finishing_; }   // :268-271 (the output buffer's back pressure is the shim's business)

    void add_input(const tgpu_page *page) override
    {
        TG_CHECK_STATE(!finishing_, "Operator is already finishing");
        DevicePage in = ingest_page(ctx_, page);
        TG_CHECK_ARG(in.cols.size() == types_.size(), "page channel count does not match the operator's types");
        for (size_t i = 0; i < types_.size(); i++) TG_CHECK_ARG(in.cols[i].type == types_[i], "page channel type does not match the operator's types");
        const int64_t n = in.n;
        if (n == 0) return;   // :274-277
        BufferPtr own_hashes;
        const int64_t *hashes = nullptr;
        if (hash_channel_ >= 0) hashes = (const int64_t *)in.cols[(size_t)hash_channel_].values;
        else {
            std::vector<const DeviceColumn *> keys;
            for (int32_t ch : partition_channels_) keys.push_back(&in.cols[(size_t)ch]);
            own_hashes = ctx_->alloc((size_t)n * 8);
            k::hash_rows(ctx_, key_cols_of(keys), n, own_hashes->as<int64_t>());
            hashes = own_hashes->as<int64_t>();
        }
        BufferPtr ids = ctx_->alloc((size_t)n * 4), cnt = ctx_->alloc((size_t)partition_count_ * 8);
        k::partition_ids(ctx_, hashes, n, partition_count_, ids->as<int32_t>(), local_function_);
        // replicated rows of this page
        const uint8_t *null_flags = null_channel_ >= 0 ? in.cols[(size_t)null_channel_].nulls : nullptr;
        const bool replicate_first = replicates_any_row_ && !has_any_row_been_replicated_;
        BufferPtr replicate;
        if (null_flags || replicate_first) {
            replicate = ctx_->alloc((size_t)n);
            replicate_flags_kernel<<<(int)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx_->cu_count() * 8), 256, 0, ctx_->stream()>>>(null_flags, n, replicate_first ? 1 : 0,
                                                                                                                                    replicate->as<uint8_t>());
            check_launch("replicate_flags");
            has_any_row_been_replicated_ = true;   // :411-418 (row 0 if the flag was pending; null rows replicate regardless)
        }
        BufferPtr positions;
        int64_t pairs = 0;
        k::partition_pairs(ctx_, ids->as<int32_t>(), replicate ? replicate->as<uint8_t>() : nullptr, n, partition_count_, positions, pairs, cnt->as<int64_t>());
        std::vector<int64_t> counts((size_t)partition_count_);
        ctx_->download(counts.data(), cnt->ptr(), (size_t)partition_count_ * 8);
        DevicePage grouped;
        grouped.n = pairs;
        for (auto &col : in.cols) grouped.cols.push_back(k::gather_column(ctx_, col, positions->as<int32_t>(), pairs, false));
        int64_t at = 0;
        for (int32_t p = 0; p < partition_count_; p++) {
            const int64_t len = counts[(size_t)p];
            if (len > 0) {
                DevicePage part;
                part.n = len;
                for (auto &col : grouped.cols) part.cols.push_back(k::region_of(ctx_, col, at, len));   // views: the buffers are shared
                pending_.push_back({p, std::move(part)});
                pages_added_++;
                rows_added_ += len;
            }
            at += len;
        }
    }

    std::unique_ptr<OutputPage> get_output() override { return nullptr; }   // :303-306
    void finish() override { finishing_ = true; }                              // :255-259 flush(true): nothing is held back here
    bool is_finished() override { return finishing_; }

    int64_t memory_bytes() override
    {
        int64_t s = 0;
        for (auto &e : pending_) s += e.second.size_in_bytes();
        return s;
    }

    bool poll(int32_t *partition, std::unique_ptr<OutputPage> *out)
    {
        if (pending_.empty()) return false;
        *partition = pending_.front().first;
        *out = wrap(std::move(pending_.front().second));
        pending_.pop_front();
        return true;
    }
    // the largest number of pending pages any one partition has (a consumer that moves one page per partition and call checks BEFORE it polls)
    size_t max_pending_per_partition() const
    {
        std::vector<size_t> cnt((size_t)partition_count_, 0);
        size_t most = 0;
        for (auto &pp : pending_) most = std::max(most, ++cnt[(size_t)pp.first]);
        return most;
    }
    int32_t partition_count() const { return partition_count_; }
    int64_t rows_added_ = 0, pages_added_ = 0;   // PartitionedOutputInfo (:396-399)

private:
    std::vector<int32_t> types_, partition_channels_;
    int32_t hash_channel_, partition_count_, null_channel_;
    bool replicates_any_row_, local_function_, has_any_row_been_replicated_ = false, finishing_ = false;
    std::deque<std::pair<int32_t, DevicePage>> pending_;
};

PartitionedOutputOperatorFactory::PartitionedOutputOperatorFactory(Context *ctx, int32_t operator_id, std::vector<int32_t> types, std::vector<int32_t> partition_channels,
                                                                   int32_t hash_channel, int32_t partition_count, bool replicates_any_row, int32_t null_channel,
                                                                   int32_t partition_function)
    : ctx_(ctx), operator_id_(operator_id), types_(std::move(types)), partition_channels_(std::move(partition_channels)), hash_channel_(hash_channel),
      partition_count_(partition_count), null_channel_(null_channel), replicates_any_row_(replicates_any_row), local_function_(partition_function == TGPU_PARTITION_LOCAL)
{
    TG_CHECK_ARG(partition_count_ > 0 && partition_count_ <= 1024, "partition count must be in 1..1024");
    TG_CHECK_ARG(partition_function == TGPU_PARTITION_HASH_MODULO || partition_function == TGPU_PARTITION_LOCAL, "unknown partition function");
    if (local_function_) TG_CHECK_ARG((partition_count_ & (partition_count_ - 1)) == 0, "the local partition function needs a power-of-two partition count");
    TG_CHECK_ARG(null_channel_ < (int)types_.size(), "null channel out of range");
    if (hash_channel_ >= 0) TG_CHECK_ARG(hash_channel_ < (int)types_.size() && types_[(size_t)hash_channel_] == TGPU_BIGINT, "bad hash channel");
    else TG_CHECK_ARG(!partition_channels_.empty(), "partitioning needs partition channels or a hash channel");
    for (int32_t ch : partition_channels_) {
        // (a negative channel = a constant partitioning argument in the reference, :433-448)
        if (ch < 0) fail(TGPU_ERR_NOT_SUPPORTED, "constant partitioning arguments are not supported");
        TG_CHECK_ARG(ch < (int)types_.size(), "partition channel out of range");
    }
    TG_CHECK_ARG((int)partition_channels_.size() <= kMaxKeyChannels, "at most 8 partition channels are supported");
}

std::unique_ptr<Operator> PartitionedOutputOperatorFactory::create_operator()
{
    TG_CHECK_STATE(!closed_, "Factory is already closed");
    return std::make_unique<PartitionedOutputOperator>(ctx_, operator_id_, types_, partition_channels_, hash_channel_, partition_count_, replicates_any_row_, null_channel_,
                                                       local_function_);
}

std::unique_ptr<OperatorFactory> PartitionedOutputOperatorFactory::duplicate()
{
    return std::unique_ptr<OperatorFactory>(new PartitionedOutputOperatorFactory(*this));
}

bool partitioned_output_poll(Operator *op, int32_t *partition, std::unique_ptr<OutputPage> *out)
{
    auto *p = dynamic_cast<PartitionedOutputOperator *>(op);
    TG_CHECK_ARG(p != nullptr, "not a PartitionedOutputOperator");
    return p->poll(partition, out);
}

void partitioned_output_pending(Operator *op, size_t *max_per_partition, int32_t *partition_count)
{
    auto *p = dynamic_cast<PartitionedOutputOperator *>(op);
    TG_CHECK_ARG(p != nullptr, "not a PartitionedOutputOperator");
    *max_per_partition = p->max_pending_per_partition();
    *partition_count = p->partition_count();
}

void partitioned_output_info(Operator *op, int64_t *rows_added, int64_t *pages_added)
{
    auto *p = dynamic_cast<PartitionedOutputOperator *>(op);
    TG_CHECK_ARG(p != nullptr, "not a PartitionedOutputOperator");
    *rows_added = p->rows_added_;
    *pages_added = p->pages_added_;
}

}  // namespace tgpu
