// c_api.cpp -- the extern "C" boundary (include/tgpu.h): exceptions -> status codes, handles -> C++ objects.
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>

#include "exchange.h"
#include "kernels.h"
#include "operators.h"
#include "serde.h"
#include "orc.h"
#include "parquet.h"

namespace tgpu {
const std::string &last_error();
}

using namespace tgpu;

namespace {

template <typename F> int32_t guard(F &&f)
{
    try {
        f();
        return TGPU_OK;
    }
    catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    }
    catch (const std::bad_alloc &) {
        set_last_error("out of host memory");
        return TGPU_ERR_INSUFFICIENT_RESOURCES;
    }
    catch (const std::exception &e) {
        set_last_error(e.what());
        return TGPU_ERR_INTERNAL;
    }
}

// Every entry point that works on a context's device binds the calling thread to that device first: HIP's current device is
// per thread, and the handles of one context may be driven by any thread (tgpu.h threading rule; JVM driver-pool threads in the
// JNI integration start on device 0 whatever device the context lives on).
std::atomic<long long> g_bind_count{0};
inline void bind_thread(Context *c)
{
    if (!c) return;
    g_bind_count++;
    HIP_CHECK(hipSetDevice(c->device()));
}
template <typename F> int32_t guard_on(Context *c, F &&f)
{
    return guard([&] {
        bind_thread(c);
        f();
    });
}
inline Context *ctx_of(const tgpu_context *h) { return h ? h->ctx.get() : nullptr; }
inline Context *ctx_of(const tgpu_operator *h) { return h ? h->ctx : nullptr; }
inline Context *ctx_of(const tgpu_operator_factory *h) { return h ? h->ctx : nullptr; }
inline Context *ctx_of(const tgpu_lookup_source_factory *h) { return h ? h->ctx : nullptr; }
inline Context *ctx_of(const tgpu_group_by_hash *h) { return h ? h->ctx : nullptr; }
inline Context *ctx_of(const tgpu_output_page *h) { return h ? h->ctx : nullptr; }
inline Context *ctx_of(const tgpu_exchange *h) { return h ? h->ctx : nullptr; }

std::vector<int32_t> vec(const int32_t *p, int32_t n)
{
    TG_CHECK_ARG(n >= 0 && (n == 0 || p != nullptr), "null array argument");
    return std::vector<int32_t>(p, p + n);
}

std::unique_ptr<OutputPage> make_output(Context *ctx, DevicePage &&p)
{
    auto o = std::make_unique<OutputPage>();
    o->ctx = ctx;
    o->page = std::move(p);
    return o;
}

std::mutex g_handle_mu;
std::map<Context *, tgpu_context *> g_wrappers;

void retain(Context *c)
{
    std::lock_guard<std::mutex> lk(g_handle_mu);
    c->retain_handle();
}

// drops one handle; destroys the context if its destruction was requested and this was the last handle
void drop(Context *c)
{
    tgpu_context *w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_handle_mu);
        if (c->release_handle()) {
            auto it = g_wrappers.find(c);
            if (it != g_wrappers.end()) {
                w = it->second;
                g_wrappers.erase(it);
            }
        }
    }
    delete w;
}

tgpu_output_page *release_output(std::unique_ptr<OutputPage> o)
{
    retain(o->ctx);
    return static_cast<tgpu_output_page *>(o.release());
}

}  // namespace

extern "C" {

int32_t tgpu_context_create(int32_t device, void *hip_stream, tgpu_context **out)
{
    return guard([&] {
        TG_CHECK_ARG(out != nullptr, "out is null");
        auto c = std::make_unique<tgpu_context>();
        c->ctx = std::make_unique<Context>(device, (hipStream_t)hip_stream);
        {
            std::lock_guard<std::mutex> lk(g_handle_mu);
            g_wrappers[c->ctx.get()] = c.get();
        }
        *out = c.release();
    });
}

void tgpu_context_destroy(tgpu_context *ctx)
{
    if (!ctx) return;
    bool now;
    {
        std::lock_guard<std::mutex> lk(g_handle_mu);
        now = ctx->ctx->request_destroy();
        if (now) g_wrappers.erase(ctx->ctx.get());
    }
    if (now) delete ctx;  // otherwise the last handle created from it deletes it
}

int32_t tgpu_context_synchronize(tgpu_context *ctx)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx != nullptr, "context is null");
        ctx->ctx->sync();
    });
}

const char *tgpu_last_error(void) { return last_error().c_str(); }
/* diagnostics: how many times entry points bound their thread to a context's device (tests) */
long long tgpu_debug_bind_count(void) { return g_bind_count.load(); }
const char *tgpu_version(void) { return "tgpu 0.1 (gfx950)"; }

int32_t tgpu_set_resource_dir(const char *dir)
{
    return guard([&] {
        TG_CHECK_ARG(dir != nullptr, "dir is null");
        set_resource_dir(dir);
    });
}

int32_t tgpu_context_set_max_output_page(tgpu_context *ctx, int64_t max_bytes, int64_t max_rows)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx != nullptr && max_bytes >= 0 && max_rows >= 0, "bad argument");
        ctx->ctx->set_max_output_page(max_bytes, max_rows);
    });
}

int32_t tgpu_context_set_double_sum_order(tgpu_context *ctx, int32_t order)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx != nullptr, "context is null");
        TG_CHECK_ARG(order == TGPU_SUM_ORDER_EXACT || order == TGPU_SUM_ORDER_JAVA, "unknown double sum order");
        ctx->ctx->set_double_sum_order(order);
    });
}

int32_t tgpu_context_set_device_input_stable(tgpu_context *ctx, int32_t stable)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx != nullptr, "context is null");
        ctx->ctx->set_device_input_stable(stable != 0);
    });
}

int32_t tgpu_pinned_alloc(tgpu_context *ctx, int64_t bytes, void **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && bytes >= 0, "bad argument");
        *out = ctx->ctx->pinned_alloc((size_t)bytes);
    });
}

int32_t tgpu_pinned_free(tgpu_context *ctx, void *ptr)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx != nullptr, "context is null");
        ctx->ctx->pinned_free(ptr);
    });
}

int32_t tgpu_profile_enable(tgpu_context *ctx, int32_t enabled)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx != nullptr, "context is null");
        ctx->ctx->set_profiling(enabled != 0);
    });
}

int32_t tgpu_profile_reset(tgpu_context *ctx)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx != nullptr, "context is null");
        ctx->ctx->profile_reset();
    });
}

int64_t tgpu_profile_dump(tgpu_context *ctx, char *buf, int64_t buf_len)
{
    int64_t need = 0;
    int32_t rc = guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx != nullptr, "context is null");
        std::string s = ctx->ctx->profile_json();
        need = (int64_t)s.size() + 1;
        if (buf && buf_len >= need) memcpy(buf, s.c_str(), (size_t)need);
    });
    return rc == TGPU_OK ? need : rc;
}

// ---- factories ------------------------------------------------------------------------------------------------------
int32_t tgpu_filter_project_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t input_type_count, const int32_t *input_types,
                                           const tgpu_page_processor_spec *spec, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out, "null argument");
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<FilterAndProjectOperatorFactory>(ctx->ctx.get(), operator_id, vec(input_types, input_type_count), spec);
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_scan_filter_project_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types, const tgpu_page_processor_spec *spec,
                                                tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && spec, "null argument");
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<ScanFilterAndProjectOperatorFactory>(ctx->ctx.get(), operator_id, vec(types, type_count), spec);
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_scan_operator_add_page_source(tgpu_operator *op, const tgpu_page_source *source)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op && source, "null argument");
        scan_add_page_source(op->op.get(), source);
    });
}

namespace {
// RecordCursor -> pages (the library's RecordPageSource): rows are pulled through the cursor's callbacks into host columns, kBatchRows at a
// time, and handed to the scan operator as an ordinary page source; the operator closes it (which closes the cursor and frees this object)
struct CursorPageSource {
    static constexpr int32_t kBatchRows = 1 << 16;
    tgpu_record_cursor cur;
    std::vector<int32_t> types;
    bool finished = false, closed = false;
    std::vector<std::vector<uint8_t>> values, nulls;
    std::vector<std::vector<int32_t>> offsets;
    std::vector<tgpu_block> blocks;

    static int32_t get_next_page(void *u, tgpu_page *page)
    {
        CursorPageSource *s = static_cast<CursorPageSource *>(u);
        if (s->finished) return 0;
        const size_t nf = s->types.size();
        s->values.assign(nf, {});
        s->nulls.assign(nf, {});
        s->offsets.assign(nf, {});
        for (size_t f = 0; f < nf; f++)
            if (s->types[f] == TGPU_VARCHAR) s->offsets[f].push_back(0);
        std::vector<bool> any_null(nf, false);
        int32_t rows = 0;
        while (rows < kBatchRows) {
            const int32_t a = s->cur.advance_next_position(s->cur.user);
            if (a < 0) return a;
            if (a == 0) {
                s->finished = true;
                break;
            }
            for (size_t f = 0; f < nf; f++) {
                const int32_t t = s->types[f];
                const bool is_null = s->cur.is_null(s->cur.user, (int32_t)f) != 0;
                s->nulls[f].push_back(is_null ? 1 : 0);
                any_null[f] = any_null[f] || is_null;
                if (t == TGPU_VARCHAR) {
                    const void *b = nullptr;
                    int32_t len = 0;
                    if (!is_null) {
                        const int32_t rc = s->cur.get_slice(s->cur.user, (int32_t)f, &b, &len);
                        if (rc < 0) return rc;
                        if (len < 0 || (len > 0 && !b)) return TGPU_ERR_INVALID_ARGUMENT;
                    }
                    const uint8_t *bytes = static_cast<const uint8_t *>(b);
                    s->values[f].insert(s->values[f].end(), bytes, bytes + len);
                    if (s->values[f].size() > 0x7fffffffu) return TGPU_ERR_INSUFFICIENT_RESOURCES;
                    s->offsets[f].push_back((int32_t)s->values[f].size());
                    continue;
                }
                const size_t w = (size_t)type_width(t), at = s->values[f].size();
                s->values[f].resize(at + w);
                if (is_null) continue;   // (zero bytes)
                if (t == TGPU_DOUBLE) {
                    const double v = s->cur.get_double(s->cur.user, (int32_t)f);
                    memcpy(&s->values[f][at], &v, 8);
                } else if (t == TGPU_BOOLEAN) s->values[f][at] = s->cur.get_boolean(s->cur.user, (int32_t)f) ? 1 : 0;
                else {
                    const int64_t v = s->cur.get_long(s->cur.user, (int32_t)f);
                    if (w == 8) memcpy(&s->values[f][at], &v, 8);
                    else {
                        const int32_t v32 = (int32_t)v;
                        memcpy(&s->values[f][at], &v32, 4);
                    }
                }
            }
            rows++;
        }
        if (rows == 0) return 0;
        s->blocks.assign(nf, tgpu_block{});
        for (size_t f = 0; f < nf; f++) {
            tgpu_block &b = s->blocks[f];
            b.type = s->types[f];
            b.encoding = TGPU_FLAT;
            b.memory = TGPU_HOST;
            b.position_count = rows;
            if (s->types[f] == TGPU_VARCHAR && s->values[f].empty()) s->values[f].push_back(0);   // (a non-null byte pool for a column of empty strings)
            b.values = s->values[f].data();
            b.nulls = any_null[f] ? s->nulls[f].data() : nullptr;
            b.offsets = s->types[f] == TGPU_VARCHAR ? s->offsets[f].data() : nullptr;
        }
        page->position_count = rows;
        page->channel_count = (int32_t)nf;
        page->blocks = s->blocks.data();
        return 1;
    }
    static int32_t is_finished(void *u) { return static_cast<CursorPageSource *>(u)->finished ? 1 : 0; }
    static void close(void *u)
    {
        CursorPageSource *s = static_cast<CursorPageSource *>(u);
        if (s->cur.close) s->cur.close(s->cur.user);
        delete s;
    }
};
}  // namespace

int32_t tgpu_scan_operator_add_record_cursor(tgpu_operator *op, const tgpu_record_cursor *cursor, int32_t type_count, const int32_t *types)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op && cursor && type_count >= 0 && (types || type_count == 0), "null argument");
        TG_CHECK_ARG(cursor->advance_next_position && cursor->is_null && cursor->get_boolean && cursor->get_long && cursor->get_double && cursor->get_slice,
                     "a record cursor implements advance_next_position, is_null and the four getters");
        for (int32_t i = 0; i < type_count; i++) TG_CHECK_ARG(valid_type(types[i]), "unknown field type");
        std::unique_ptr<CursorPageSource> s = std::make_unique<CursorPageSource>();
        s->cur = *cursor;
        s->types.assign(types, types + type_count);
        tgpu_page_source ps{};
        ps.user = s.get();
        ps.get_next_page = &CursorPageSource::get_next_page;
        ps.is_finished = &CursorPageSource::is_finished;
        ps.close = &CursorPageSource::close;
        scan_add_page_source(op->op.get(), &ps);   // (copies the callbacks; from here on the operator owns the source and closes it)
        s.release();
    });
}

int32_t tgpu_scan_operator_no_more_splits(tgpu_operator *op)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed");
        scan_no_more_splits(op->op.get());
    });
}

int32_t tgpu_scan_operator_stats(tgpu_operator *op, int64_t *processed_positions, int64_t *lazy_blocks_loaded, int64_t *lazy_blocks_skipped)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op && processed_positions && lazy_blocks_loaded && lazy_blocks_skipped, "null argument");
        scan_stats(op->op.get(), processed_positions, lazy_blocks_loaded, lazy_blocks_skipped);
    });
}

// compile-only entry (no GPU needed): lets build() pre-warm the on-disk kernel cache
int32_t tgpu_precompile_page_processor(int32_t input_type_count, const int32_t *input_types, const tgpu_page_processor_spec *spec)
{
    return guard([&] {
        PageProcessorGpu p(vec(input_types, input_type_count), spec);
        p.precompile();
    });
}

// returns the generated kernel source (diagnostics / DESIGN.md listings); returns needed length
int64_t tgpu_page_processor_source(int32_t input_type_count, const int32_t *input_types, const tgpu_page_processor_spec *spec, char *buf, int64_t buf_len)
{
    int64_t need = 0;
    int32_t rc = guard([&] {
        PageProcessorGpu p(vec(input_types, input_type_count), spec);
        need = (int64_t)p.source().size() + 1;
        if (buf && buf_len >= need) memcpy(buf, p.source().c_str(), (size_t)need);
    });
    return rc == TGPU_OK ? need : rc;
}

int32_t tgpu_hash_aggregation_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t group_by_count, const int32_t *group_by_types,
                                             const int32_t *group_by_channels, int32_t hash_channel, int32_t step, int32_t agg_count,
                                             const tgpu_agg_spec *aggs, int32_t expected_groups, int32_t produce_default_output,
                                             tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out, "null argument");
        TG_CHECK_ARG(agg_count >= 0 && (agg_count == 0 || aggs != nullptr), "null aggregate array");
        HashAggregationConfig cfg;
        cfg.group_by_types = vec(group_by_types, group_by_count);
        cfg.group_by_channels = vec(group_by_channels, group_by_count);
        cfg.hash_channel = hash_channel;
        cfg.step = step;
        cfg.aggs.assign(aggs, aggs + agg_count);
        cfg.expected_groups = expected_groups;
        cfg.produce_default_output = produce_default_output != 0;
        if (const char *env = getenv("TGPU_MAX_PARTIAL_AGGREGATION_MEMORY")) cfg.max_partial_memory = atoll(env);
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<HashAggregationOperatorFactory>(ctx->ctx.get(), operator_id, std::move(cfg));
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_hash_builder_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types,
                                         int32_t output_channel_count, const int32_t *output_channels, int32_t hash_channel_count,
                                         const int32_t *hash_channels, int32_t precomputed_hash_channel, int32_t expected_positions,
                                         tgpu_lookup_source_factory **bridge_out, tgpu_operator_factory **out)
{
    return tgpu_partitioned_hash_builder_factory_create(ctx, operator_id, type_count, types, output_channel_count, output_channels, hash_channel_count, hash_channels,
                                                        precomputed_hash_channel, expected_positions, 1, bridge_out, out);
}

int64_t tgpu_partitioned_join_position_encode(int32_t partition, int32_t join_position, int32_t partition_count)
{
    // PartitionedLookupSource.java:101-102,222-226: shiftSize = numberOfTrailingZeros(partitions) + 1
    if (partition_count <= 0 || (partition_count & (partition_count - 1)) || partition < 0 || partition >= partition_count || join_position < 0)
        return TGPU_ERR_INVALID_ARGUMENT;   // (negative: never a valid encoded position)
    const int shift = __builtin_ctz((unsigned)partition_count) + 1;
    return ((int64_t)join_position << shift) | (int64_t)partition;
}

int32_t tgpu_partitioned_join_position_decode(int64_t partitioned_join_position, int32_t partition_count, int32_t *partition, int32_t *join_position)
{
    if (partition_count <= 0 || (partition_count & (partition_count - 1)) || !partition || !join_position) return TGPU_ERR_INVALID_ARGUMENT;
    const int shift = __builtin_ctz((unsigned)partition_count) + 1;
    *partition = (int32_t)(partitioned_join_position & (partition_count - 1));             // :212-216
    *join_position = (int32_t)((uint64_t)partitioned_join_position >> shift);               // :218-221
    return TGPU_OK;
}

int32_t tgpu_partitioned_hash_builder_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types,
                                                     int32_t output_channel_count, const int32_t *output_channels, int32_t hash_channel_count,
                                                     const int32_t *hash_channels, int32_t precomputed_hash_channel, int32_t expected_positions,
                                                     int32_t partition_count, tgpu_lookup_source_factory **bridge_out, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && bridge_out, "null argument");
        HashBuilderConfig cfg;
        cfg.types = vec(types, type_count);
        cfg.output_channels = vec(output_channels, output_channel_count);
        cfg.hash_channels = vec(hash_channels, hash_channel_count);
        cfg.precomputed_hash_channel = precomputed_hash_channel;
        cfg.expected_positions = expected_positions;
        cfg.partition_count = partition_count;
        auto bridge = std::make_unique<tgpu_lookup_source_factory>();
        bridge->bridge = std::make_shared<LookupSourceFactory>();
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<HashBuilderOperatorFactory>(ctx->ctx.get(), operator_id, std::move(cfg), bridge->bridge);
        bridge->ctx = ctx->ctx.get();
        retain(bridge->ctx);
        *bridge_out = bridge.release();
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_top_n_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types, int64_t n,
                                  int32_t sort_channel_count, const int32_t *sort_channels, const int32_t *sort_orders,
                                  tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out, "null argument");
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<TopNOperatorFactory>(ctx->ctx.get(), operator_id, vec(types, type_count), n, vec(sort_channels, sort_channel_count),
                                                     vec(sort_orders, sort_channel_count));
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_order_by_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types,
                                     int32_t output_channel_count, const int32_t *output_channels, int32_t expected_positions,
                                     int32_t sort_channel_count, const int32_t *sort_channels, const int32_t *sort_orders,
                                     tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out, "null argument");
        (void)expected_positions;   // a sizing hint of the reference's PagesIndex; the device store grows by doubling
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<OrderByOperatorFactory>(ctx->ctx.get(), operator_id, vec(types, type_count), vec(output_channels, output_channel_count),
                                                        vec(sort_channels, sort_channel_count), vec(sort_orders, sort_channel_count));
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

void tgpu_lookup_source_factory_destroy(tgpu_lookup_source_factory *bridge)
{
    if (!bridge) return;
    Context *c = bridge->ctx;
    delete bridge;
    if (c) drop(c);
}

int32_t tgpu_lookup_source_factory_set_join_filter(tgpu_lookup_source_factory *bridge, int32_t probe_type_count, const int32_t *probe_types,
                                                   const tgpu_page_processor_spec *spec)
{
    return guard_on(ctx_of(bridge), [&] {
        TG_CHECK_ARG(bridge && spec, "null argument");
        TG_CHECK_ARG(spec->filter_root >= 0 && spec->filter_root < spec->node_count, "the join filter is the spec's filter expression");
        auto f = std::make_shared<JoinFilter>();
        f->nodes.assign(spec->nodes, spec->nodes + spec->node_count);
        if (spec->string_pool && spec->string_pool_len > 0) f->pool.assign(spec->string_pool, spec->string_pool + spec->string_pool_len);
        f->root = spec->filter_root;
        f->build_types = bridge->bridge->build_types;
        f->probe_types = vec(probe_types, probe_type_count);
        TG_CHECK_ARG(f->nodes[(size_t)f->root].type == TGPU_BOOLEAN, "the join filter must be a BOOLEAN expression");
        const int limit = (int)f->build_types.size() + probe_type_count;
        for (auto &nd : f->nodes) {
            if (nd.kind != TGPU_EX_INPUT) continue;
            TG_CHECK_ARG(nd.op >= 0 && nd.op < limit, "join filter: input channel out of range");
            const int32_t t = nd.op < (int)f->build_types.size() ? f->build_types[(size_t)nd.op] : f->probe_types[(size_t)nd.op - f->build_types.size()];
            TG_CHECK_ARG(t == nd.type, "join filter: input reference type does not match the channel type");
        }
        bridge->bridge->set_join_filter(f);
    });
}

int32_t tgpu_lookup_source_stats(tgpu_lookup_source_factory *bridge, int64_t *positions, int64_t *hash_size, int64_t *link_count)
{
    return guard_on(ctx_of(bridge), [&] {
        TG_CHECK_ARG(bridge != nullptr, "bridge is null");
        auto s = bridge->bridge->lookup_source();
        TG_CHECK_STATE(s != nullptr, "Lookup source has not been built yet");
        if (positions) *positions = s->position_count();
        if (hash_size) *hash_size = s->hash_size();
        if (link_count) *link_count = s->link_count();
    });
}

int32_t tgpu_lookup_join_factory_create(tgpu_context *ctx, int32_t operator_id, tgpu_lookup_source_factory *bridge, int32_t probe_type_count,
                                        const int32_t *probe_types, int32_t probe_join_channel_count, const int32_t *probe_join_channels,
                                        int32_t probe_hash_channel, int32_t probe_output_channel_count, const int32_t *probe_output_channels,
                                        int32_t join_type, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && bridge, "null argument");
        LookupJoinConfig cfg;
        cfg.probe_types = vec(probe_types, probe_type_count);
        cfg.probe_join_channels = vec(probe_join_channels, probe_join_channel_count);
        cfg.probe_output_channels = vec(probe_output_channels, probe_output_channel_count);
        cfg.probe_hash_channel = probe_hash_channel;
        cfg.join_type = join_type;
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<LookupJoinOperatorFactory>(ctx->ctx.get(), operator_id, std::move(cfg), bridge->bridge);
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_filter_project_lookup_join_factory_create(tgpu_context *ctx, int32_t operator_id, tgpu_lookup_source_factory *bridge, int32_t input_type_count,
                                                       const int32_t *input_types, const tgpu_page_processor_spec *spec, int32_t probe_join_channel_count,
                                                       const int32_t *probe_join_channels, int32_t probe_hash_channel, int32_t probe_output_channel_count,
                                                       const int32_t *probe_output_channels, int32_t join_type, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && bridge && spec, "null argument");
        LookupJoinConfig cfg;
        cfg.probe_join_channels = vec(probe_join_channels, probe_join_channel_count);
        cfg.probe_output_channels = vec(probe_output_channels, probe_output_channel_count);
        cfg.probe_hash_channel = probe_hash_channel;
        cfg.join_type = join_type;
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<FusedFilterProjectJoinOperatorFactory>(ctx->ctx.get(), operator_id, vec(input_types, input_type_count), spec, std::move(cfg), bridge->bridge);
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_filter_project_hash_aggregation_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t input_type_count, const int32_t *input_types,
                                                            const tgpu_page_processor_spec *spec, int32_t group_by_count, const int32_t *group_by_types,
                                                            const int32_t *group_by_channels, int32_t hash_channel, int32_t step, int32_t agg_count,
                                                            const tgpu_agg_spec *aggs, int32_t expected_groups, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && spec, "null argument");
        TG_CHECK_ARG(agg_count >= 0 && (agg_count == 0 || aggs != nullptr), "null aggregate array");
        TG_CHECK_ARG(step == TGPU_STEP_SINGLE || step == TGPU_STEP_PARTIAL, "a fused filter/project feeds a SINGLE or PARTIAL aggregation");
        HashAggregationConfig cfg;
        cfg.group_by_types = vec(group_by_types, group_by_count);
        cfg.group_by_channels = vec(group_by_channels, group_by_count);
        cfg.hash_channel = hash_channel;
        cfg.step = step;
        cfg.aggs.assign(aggs, aggs + agg_count);
        cfg.expected_groups = expected_groups;
        if (const char *env = getenv("TGPU_MAX_PARTIAL_AGGREGATION_MEMORY")) cfg.max_partial_memory = atoll(env);
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<FusedFilterProjectAggregationOperatorFactory>(ctx->ctx.get(), operator_id, vec(input_types, input_type_count), spec, std::move(cfg));
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_precompile_fused_aggregation(int32_t input_type_count, const int32_t *input_types, const tgpu_page_processor_spec *spec, int32_t agg_count,
                                          const tgpu_agg_spec *aggs, int32_t group_by_count, const int32_t *group_by_channels)
{
    return guard([&] {
        FusedAggGpu f(vec(input_types, input_type_count), spec, std::vector<tgpu_agg_spec>(aggs, aggs + agg_count), vec(group_by_channels, group_by_count));
        f.precompile();
    });
}

// compile-only (no GPU): pre-warm the kernel cache of a fused filter+project+probe pipeline
int32_t tgpu_precompile_fused_probe(int32_t input_type_count, const int32_t *input_types, const tgpu_page_processor_spec *spec, int32_t join_channel,
                                    int32_t probe_output_channel_count, const int32_t *probe_output_channels)
{
    return guard([&] {
        FusedProbeGpu f(vec(input_types, input_type_count), spec, join_channel, vec(probe_output_channels, probe_output_channel_count));
        f.precompile();
    });
}

int32_t tgpu_operator_factory_create_operator(tgpu_operator_factory *factory, tgpu_operator **out)
{
    return guard_on(ctx_of(factory), [&] {
        TG_CHECK_ARG(factory && out, "null argument");
        auto o = std::make_unique<tgpu_operator>();
        o->op = factory->f->create_operator();
        o->ctx = factory->ctx;
        retain(o->ctx);
        *out = o.release();
    });
}

int32_t tgpu_operator_factory_duplicate(tgpu_operator_factory *factory, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(factory), [&] {
        TG_CHECK_ARG(factory && out, "null argument");
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = factory->f->duplicate();
        f->ctx = factory->ctx;
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_operator_factory_no_more_operators(tgpu_operator_factory *factory)
{
    return guard_on(ctx_of(factory), [&] {
        TG_CHECK_ARG(factory != nullptr, "factory is null");
        factory->f->no_more_operators();
    });
}

void tgpu_operator_factory_destroy(tgpu_operator_factory *factory)
{
    if (!factory) return;
    Context *c = factory->ctx;
    delete factory;
    if (c) drop(c);
}

// ---- Operator -------------------------------------------------------------------------------------------------------
#define OP_BOOL(expr)                                            \
    int32_t result = 0;                                          \
    int32_t rc = guard_on(ctx_of(op), [&] {                      \
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed"); \
        result = (expr) ? 1 : 0;                                 \
    });                                                          \
    return rc == TGPU_OK ? result : rc;

// (regions of a cut output page still waiting to be handed out count as pending output)
int32_t tgpu_operator_needs_input(tgpu_operator *op) { OP_BOOL(op->cut_pages.empty() && op->op->needs_input()) }
int32_t tgpu_operator_is_finished(tgpu_operator *op) { OP_BOOL(op->cut_pages.empty() && op->op->is_finished()) }
int32_t tgpu_operator_is_blocked(tgpu_operator *op) { OP_BOOL(op->op->is_blocked()) }

int32_t tgpu_operator_add_input(tgpu_operator *op, const tgpu_page *page)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed");
        TG_CHECK_ARG(page != nullptr, "page is null");
        TG_CHECK_STATE(op->op->needs_input(), "Operator does not need input");
        op->op->add_input(page);
    });
}

// PageBuilder.isFull (S/PageBuilder.java:126-129, PageBuilderStatus.java:49-60: DEFAULT_MAX_PAGE_SIZE_IN_BYTES = 1 MB) is what cuts the output
// of the reference's operators into pages (LookupJoinPageBuilder.java:51-56, HashAggregationOperator's buildResult, OrderByOperator.java:270-296).
// The GPU operators produce one page per call; with tgpu_context_set_max_output_page they hand it out as consecutive regions of at most
// `rows` rows and about `bytes` bytes (the page's Java-accounted size spread evenly over its rows) -- zero-copy views of the same buffers.
static void cut_output_page(tgpu_operator *op, std::unique_ptr<OutputPage> &p)
{
    Context *c = op->ctx;
    const int64_t max_bytes = c->max_output_page_bytes(), max_rows = c->max_output_page_rows();
    const int64_t n = p->page.n;
    if ((max_bytes <= 0 && max_rows <= 0) || n <= 1) return;
    int64_t per = n;
    if (max_rows > 0) per = std::min(per, max_rows);
    if (max_bytes > 0) {
        int64_t size = 0;   // Page.getSizeInBytes of flat blocks: (width + 1) per fixed-width cell, length + 5 per VARCHAR cell
        for (auto &col : p->page.cols) size += col.type == TGPU_VARCHAR ? col.pool_bytes + 5 * n : (int64_t)(type_width(col.type) + 1) * n;
        if (size > max_bytes) per = std::min<int64_t>(per, std::max<int64_t>(1, (int64_t)((double)n * (double)max_bytes / (double)size)));
    }
    if (per >= n) return;
    std::unique_ptr<OutputPage> whole = std::move(p);
    for (int64_t at = 0; at < n; at += per) {
        const int64_t len = std::min(per, n - at);
        DevicePage part;
        part.n = len;
        for (auto &col : whole->page.cols) part.cols.push_back(k::region_of(c, col, at, len));
        auto o = std::make_unique<OutputPage>();
        o->ctx = c;
        o->page = std::move(part);
        if (at == 0) p = std::move(o);
        else op->cut_pages.push_back(std::move(o));
    }
}

int32_t tgpu_operator_get_output(tgpu_operator *op, tgpu_output_page **out)
{
    bool would_block = false;
    int32_t rc = guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op && out, "null argument");
        *out = nullptr;
        std::unique_ptr<OutputPage> p;
        if (!op->cut_pages.empty()) {
            p = std::move(op->cut_pages.front());
            op->cut_pages.pop_front();
        }
        else {
            p = op->op->get_output();
            if (p) cut_output_page(op, p);
        }
        if (p) *out = release_output(std::move(p));
        else would_block = op->op->is_blocked();   // no page AND an unfinished isBlocked() future (Operator.java:32-35): come back later
    });
    return rc == TGPU_OK && would_block ? TGPU_WOULD_BLOCK : rc;
}

// Operator.startMemoryRevoke / finishMemoryRevoke (M/operator/Operator.java:53-79).  Only a spill-enabled SINGLE / FINAL hash aggregation
// holds revocable memory (tgpu_operator_revocable_memory_bytes): startMemoryRevoke moves its groups to host memory as one run, and the
// "future" is done when the call returns.  Every other operator reports non-revocable user memory: nothing to do.
int32_t tgpu_operator_start_memory_revoke(tgpu_operator *op)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed");
        op->op->start_memory_revoke();
    });
}

int32_t tgpu_operator_finish_memory_revoke(tgpu_operator *op)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed");
        op->op->finish_memory_revoke();
    });
}

int64_t tgpu_operator_revocable_memory_bytes(tgpu_operator *op)
{
    int64_t v = -1;
    guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed");
        v = op->op->revocable_memory_bytes();
    });
    return v;
}

int32_t tgpu_operator_spill_stats(tgpu_operator *op, int64_t *spill_count, int64_t *spilled_bytes)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed");
        int64_t c = 0, b = 0;
        op->op->spill_stats(c, b);
        if (spill_count) *spill_count = c;
        if (spilled_bytes) *spilled_bytes = b;
    });
}

int32_t tgpu_hash_aggregation_factory_set_max_partial_memory(tgpu_operator_factory *factory, int64_t bytes)
{
    return guard_on(ctx_of(factory), [&] {
        TG_CHECK_ARG(factory != nullptr && factory->f, "factory is null");
        TG_CHECK_ARG(bytes >= 0, "maxPartialMemory must not be negative");
        if (auto *f = dynamic_cast<HashAggregationOperatorFactory *>(factory->f.get())) f->set_max_partial_memory(bytes);
        else if (auto *g = dynamic_cast<FusedFilterProjectAggregationOperatorFactory *>(factory->f.get())) g->set_max_partial_memory(bytes);
        else fail(TGPU_ERR_NOT_SUPPORTED, "only hash aggregation factories have a partial-aggregation memory limit");
    });
}

int32_t tgpu_hash_aggregation_factory_set_spill_enabled(tgpu_operator_factory *factory, int32_t enabled)
{
    return guard_on(ctx_of(factory), [&] {
        TG_CHECK_ARG(factory != nullptr && factory->f, "factory is null");
        if (auto *f = dynamic_cast<HashAggregationOperatorFactory *>(factory->f.get())) f->set_spill_enabled(enabled != 0);
        else if (auto *g = dynamic_cast<FusedFilterProjectAggregationOperatorFactory *>(factory->f.get())) g->set_spill_enabled(enabled != 0);
        else fail(TGPU_ERR_NOT_SUPPORTED, "only hash aggregation factories can spill");
    });
}

/* diagnostics: input pages a FilterAndProjectOperator processed once per dictionary entry (DictionaryAwarePageFilter / -Projection path) */
int64_t tgpu_debug_dictionary_pages(tgpu_operator *op)
{
    int64_t v = 0;
    int32_t rc = guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed");
        v = filter_project_dictionary_pages(op->op.get());
    });
    return rc == TGPU_OK ? v : rc;
}

int32_t tgpu_operator_finish(tgpu_operator *op)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed");
        op->op->finish();
    });
}

int64_t tgpu_operator_memory_bytes(tgpu_operator *op)
{
    int64_t v = 0;
    int32_t rc = guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op != nullptr && op->op, "operator is null or closed");
        v = op->op->memory_bytes();
    });
    return rc == TGPU_OK ? v : rc;
}

void tgpu_operator_close(tgpu_operator *op)
{
    if (!op) return;
    guard_on(ctx_of(op), [&] {
        if (op->op) op->op->close();
    });
    Context *c = op->ctx;
    delete op;
    if (c) drop(c);
}

// ---- output pages ---------------------------------------------------------------------------------------------------
int32_t tgpu_output_page_position_count(const tgpu_output_page *page) { return page ? (int32_t)page->page.n : TGPU_ERR_INVALID_ARGUMENT; }
int32_t tgpu_output_page_channel_count(const tgpu_output_page *page) { return page ? (int32_t)page->page.cols.size() : TGPU_ERR_INVALID_ARGUMENT; }

int32_t tgpu_output_page_as_page(const tgpu_output_page *page, tgpu_page *out)
{
    return guard_on(ctx_of(page), [&] {
        TG_CHECK_ARG(page && out, "null argument");
        auto *p = const_cast<tgpu_output_page *>(page);
        p->blocks.clear();
        for (auto &c : p->page.cols) {
            tgpu_block b{};
            b.type = c.type;
            b.encoding = TGPU_FLAT;
            b.memory = TGPU_DEVICE;
            b.position_count = (int32_t)p->page.n;
            b.values = c.values;
            b.nulls = c.nulls;
            b.offsets = c.offsets;
            p->blocks.push_back(b);
        }
        out->position_count = (int32_t)p->page.n;
        out->channel_count = (int32_t)p->blocks.size();
        out->blocks = p->blocks.data();
    });
}

int32_t tgpu_output_page_block_info(const tgpu_output_page *page, int32_t ch, int32_t *type, int64_t *value_bytes, int32_t *may_have_nulls)
{
    return guard_on(ctx_of(page), [&] {
        TG_CHECK_ARG(page != nullptr && ch >= 0 && ch < (int)page->page.cols.size(), "bad page / channel");
        const DeviceColumn &c = page->page.cols[(size_t)ch];
        if (type) *type = c.type;
        if (may_have_nulls) *may_have_nulls = c.nulls ? 1 : 0;
        if (value_bytes) {
            if (c.type == TGPU_VARCHAR) {
                // region views keep absolute offsets: the bytes of this block are [offsets[0], offsets[n])
                int32_t a = 0, b = 0;
                if (page->page.n > 0) {
                    page->ctx->download(&a, c.offsets, 4);
                    page->ctx->download(&b, c.offsets + page->page.n, 4);
                }
                *value_bytes = (int64_t)b - a;
            }
            else *value_bytes = page->page.n * type_width(c.type);
        }
    });
}

int32_t tgpu_output_page_copy_block(const tgpu_output_page *page, int32_t ch, void *values, uint8_t *nulls, int32_t *offsets)
{
    return guard_on(ctx_of(page), [&] {
        TG_CHECK_ARG(page != nullptr && ch >= 0 && ch < (int)page->page.cols.size(), "bad page / channel");
        const DeviceColumn &c = page->page.cols[(size_t)ch];
        const int64_t n = page->page.n;
        Context *ctx = page->ctx;
        if (nulls) {
            if (c.nulls) ctx->download(nulls, c.nulls, (size_t)n);
            else memset(nulls, 0, (size_t)n);
        }
        if (c.type == TGPU_VARCHAR) {
            TG_CHECK_ARG(offsets != nullptr, "offsets buffer required for VARCHAR");
            ctx->download(offsets, c.offsets, (size_t)(n + 1) * 4);
            const int32_t base = n > 0 ? offsets[0] : 0;
            const int64_t bytes = n > 0 ? (int64_t)offsets[n] - base : 0;
            if (bytes && values) ctx->download(values, (const uint8_t *)c.values + base, (size_t)bytes);
            for (int64_t i = 0; i <= n; i++) offsets[i] -= base;
            if (n == 0) offsets[0] = 0;
        }
        else if (values && n > 0) {
            ctx->download(values, c.values, (size_t)n * type_width(c.type));
        }
    });
}

int32_t tgpu_output_page_copy_blocks(const tgpu_output_page *page, int32_t channel_count, void *const *values, uint8_t *const *nulls,
                                     int32_t *const *offsets)
{
    return guard_on(ctx_of(page), [&] {
        TG_CHECK_ARG(page != nullptr && values != nullptr && nulls != nullptr && offsets != nullptr, "null argument");
        TG_CHECK_ARG(channel_count == (int)page->page.cols.size(), "channel count differs from the page's");
        const int64_t n = page->page.n;
        Context *ctx = page->ctx;
        std::vector<Context::Transfer> first, second;
        for (int ch = 0; ch < channel_count; ch++) {
            const DeviceColumn &c = page->page.cols[(size_t)ch];
            if (nulls[ch]) {
                if (c.nulls) first.push_back({nulls[ch], c.nulls, (size_t)n});
                else memset(nulls[ch], 0, (size_t)n);
            }
            if (c.type == TGPU_VARCHAR) {
                TG_CHECK_ARG(offsets[ch] != nullptr, "offsets buffer required for VARCHAR");
                first.push_back({offsets[ch], c.offsets, (size_t)(n + 1) * 4});
            }
            else if (values[ch] && n > 0) first.push_back({values[ch], c.values, (size_t)n * type_width(c.type)});
        }
        ctx->download_batch(first);
        for (int ch = 0; ch < channel_count; ch++) {
            const DeviceColumn &c = page->page.cols[(size_t)ch];
            if (c.type != TGPU_VARCHAR) continue;
            int32_t *off = offsets[ch];
            const int32_t base = n > 0 ? off[0] : 0;
            const int64_t bytes = n > 0 ? (int64_t)off[n] - base : 0;
            if (bytes && values[ch]) second.push_back({values[ch], (const uint8_t *)c.values + base, (size_t)bytes});
            for (int64_t i = 0; i <= n; i++) off[i] -= base;
            if (n == 0) off[0] = 0;
        }
        ctx->download_batch(second);
    });
}

void tgpu_output_page_release(tgpu_output_page *page)
{
    if (!page) return;
    Context *c = page->ctx;
    delete static_cast<OutputPage *>(page);
    if (c) drop(c);
}

// ---- GroupByHash ----------------------------------------------------------------------------------------------------
int32_t tgpu_group_by_hash_create(tgpu_context *ctx, int32_t type_count, const int32_t *types, const int32_t *hash_channels, int32_t input_hash_channel,
                                  int32_t expected_size, tgpu_group_by_hash **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out, "null argument");
        auto g = std::make_unique<tgpu_group_by_hash>();
        g->ctx = ctx->ctx.get();
        g->hash_channels = vec(hash_channels, type_count);
        g->input_hash_channel = input_hash_channel;
        g->gbh = std::make_unique<GroupByHashGpu>(ctx->ctx.get(), vec(types, type_count), input_hash_channel >= 0, expected_size);
        retain(g->ctx);
        *out = g.release();
    });
}

void tgpu_group_by_hash_destroy(tgpu_group_by_hash *gbh)
{
    if (!gbh) return;
    Context *c = gbh->ctx;
    delete gbh;
    if (c) drop(c);
}

static void gbh_inputs(tgpu_group_by_hash *g, const DevicePage &in, std::vector<const DeviceColumn *> &keys, const int64_t *&hashes)
{
    for (int32_t ch : g->hash_channels) {
        TG_CHECK_ARG(ch >= 0 && ch < (int)in.cols.size(), "hash channel out of range");
        keys.push_back(&in.cols[(size_t)ch]);
    }
    hashes = nullptr;
    if (g->input_hash_channel >= 0) {
        TG_CHECK_ARG(g->input_hash_channel < (int)in.cols.size() && in.cols[(size_t)g->input_hash_channel].type == TGPU_BIGINT, "bad input hash channel");
        hashes = (const int64_t *)in.cols[(size_t)g->input_hash_channel].values;
    }
}

int32_t tgpu_group_by_hash_get_group_ids(tgpu_group_by_hash *gbh, const tgpu_page *page, int64_t *group_ids, int64_t *group_count)
{
    return guard_on(ctx_of(gbh), [&] {
        TG_CHECK_ARG(gbh && page, "null argument");
        DevicePage in = ingest_page(gbh->ctx, page);
        std::vector<const DeviceColumn *> keys;
        const int64_t *hashes;
        gbh_inputs(gbh, in, keys, hashes);
        BufferPtr gids = gbh->ctx->alloc((size_t)(in.n > 0 ? in.n : 1) * 4);
        gbh->gbh->get_group_ids(keys, hashes, in.n, gids->as<int32_t>());
        if (group_ids && in.n > 0) {
            BufferPtr wide = gbh->ctx->alloc((size_t)in.n * 8);
            k::widen_i32_to_i64(gbh->ctx, gids->as<int32_t>(), wide->as<int64_t>(), in.n);
            gbh->ctx->download(group_ids, wide->ptr(), (size_t)in.n * 8);
        }
        else gbh->ctx->sync();
        if (group_count) *group_count = gbh->gbh->group_count();
    });
}

int32_t tgpu_group_by_hash_add_page(tgpu_group_by_hash *gbh, const tgpu_page *page) { return tgpu_group_by_hash_get_group_ids(gbh, page, nullptr, nullptr); }

int32_t tgpu_group_by_hash_contains(tgpu_group_by_hash *gbh, int32_t position, const tgpu_page *page, int32_t *result)
{
    return guard_on(ctx_of(gbh), [&] {
        TG_CHECK_ARG(gbh && page && result, "null argument");
        DevicePage in = ingest_page(gbh->ctx, page);
        TG_CHECK_ARG(position >= 0 && position < in.n, "position out of range");
        std::vector<const DeviceColumn *> keys;
        const int64_t *hashes;
        gbh_inputs(gbh, in, keys, hashes);
        std::vector<DeviceColumn> one;
        for (auto *c : keys) one.push_back(k::region_of(gbh->ctx, *c, position, 1));
        std::vector<const DeviceColumn *> kp;
        for (auto &c : one) kp.push_back(&c);
        BufferPtr out = gbh->ctx->alloc(4);
        gbh->gbh->lookup(kp, hashes ? hashes + position : nullptr, 1, out->as<int32_t>());
        *result = gbh->ctx->read_scalar(out->as<int32_t>()) >= 0 ? 1 : 0;
    });
}

int64_t tgpu_group_by_hash_group_count(tgpu_group_by_hash *gbh) { return gbh ? gbh->gbh->group_count() : TGPU_ERR_INVALID_ARGUMENT; }
int32_t tgpu_group_by_hash_capacity(tgpu_group_by_hash *gbh) { return gbh ? gbh->gbh->java_capacity() : TGPU_ERR_INVALID_ARGUMENT; }
int64_t tgpu_group_by_hash_estimated_size(tgpu_group_by_hash *gbh) { return gbh ? gbh->gbh->estimated_size() : TGPU_ERR_INVALID_ARGUMENT; }
int32_t tgpu_group_by_hash_rehash_count(tgpu_group_by_hash *gbh) { return gbh ? gbh->gbh->java_rehash_count() : TGPU_ERR_INVALID_ARGUMENT; }

int32_t tgpu_group_by_hash_append_values(tgpu_group_by_hash *gbh, tgpu_output_page **out)
{
    return guard_on(ctx_of(gbh), [&] {
        TG_CHECK_ARG(gbh && out, "null argument");
        DevicePage p = gbh->gbh->key_page(gbh->input_hash_channel >= 0);
        *out = release_output(make_output(gbh->ctx, std::move(p)));
    });
}

// ---- hash / partition -----------------------------------------------------------------------------------------------
int32_t tgpu_hash_page(tgpu_context *ctx, const tgpu_page *page, int32_t channel_count, const int32_t *channels, int64_t *hashes)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && page && hashes, "null argument");
        Context *c = ctx->ctx.get();
        DevicePage in = ingest_page(c, page);
        std::vector<const DeviceColumn *> keys;
        for (int32_t ch : vec(channels, channel_count)) {
            TG_CHECK_ARG(ch >= 0 && ch < (int)in.cols.size(), "channel out of range");
            keys.push_back(&in.cols[(size_t)ch]);
        }
        if (in.n == 0) return;
        BufferPtr out = c->alloc((size_t)in.n * 8);
        k::hash_rows(c, key_cols_of(keys), in.n, out->as<int64_t>());
        c->download(hashes, out->ptr(), (size_t)in.n * 8);
    });
}

int32_t tgpu_partition_page(tgpu_context *ctx, const tgpu_page *page, int32_t key_channel_count, const int32_t *key_channels, int32_t hash_channel,
                            int32_t partition_count, int64_t *counts, tgpu_output_page **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && page && counts && out, "null argument");
        TG_CHECK_ARG(partition_count > 0 && partition_count <= 1024, "partition count must be in 1..1024");
        Context *c = ctx->ctx.get();
        DevicePage in = ingest_page(c, page);
        *out = nullptr;
        for (int32_t p = 0; p < partition_count; p++) counts[p] = 0;
        BufferPtr own_hashes;
        const int64_t *hashes = nullptr;
        if (hash_channel >= 0) {
            TG_CHECK_ARG(hash_channel < (int)in.cols.size() && in.cols[(size_t)hash_channel].type == TGPU_BIGINT, "bad hash channel");
            hashes = (const int64_t *)in.cols[(size_t)hash_channel].values;
        }
        else {
            std::vector<const DeviceColumn *> keys;
            for (int32_t ch : vec(key_channels, key_channel_count)) {
                TG_CHECK_ARG(ch >= 0 && ch < (int)in.cols.size(), "channel out of range");
                keys.push_back(&in.cols[(size_t)ch]);
            }
            TG_CHECK_ARG(!keys.empty(), "partitioning needs key channels or a hash channel");
            own_hashes = c->alloc((size_t)(in.n > 0 ? in.n : 1) * 8);
            k::hash_rows(c, key_cols_of(keys), in.n, own_hashes->as<int64_t>());
            hashes = own_hashes->as<int64_t>();
        }
        const int64_t n = in.n;
        BufferPtr ids = c->alloc((size_t)(n > 0 ? n : 1) * 4), pos = c->alloc((size_t)(n > 0 ? n : 1) * 4), cnt = c->alloc((size_t)partition_count * 8);
        k::partition_ids(c, hashes, n, partition_count, ids->as<int32_t>());
        k::partition_positions(c, ids->as<int32_t>(), n, partition_count, pos->as<int32_t>(), cnt->as<int64_t>());
        c->download(counts, cnt->ptr(), (size_t)partition_count * 8);
        DevicePage o;
        o.n = n;
        for (auto &col : in.cols) o.cols.push_back(k::gather_column(c, col, pos->as<int32_t>(), n, false));
        *out = release_output(make_output(c, std::move(o)));
    });
}

int32_t tgpu_operator_add_input_output_page(tgpu_operator *op, const tgpu_output_page *page)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op && page, "null argument");
        op->op->add_input_owned(page->page);
    });
}

int32_t tgpu_lookup_outer_factory_create(tgpu_context *ctx, int32_t operator_id, tgpu_lookup_source_factory *bridge, int32_t probe_output_type_count,
                                         const int32_t *probe_output_types, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && bridge && out, "null argument");
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<LookupOuterOperatorFactory>(ctx->ctx.get(), operator_id, vec(probe_output_types, probe_output_type_count), bridge->bridge);
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_dynamic_filter_source_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types, int32_t channel_count,
                                                  const int32_t *channels, int32_t max_distinct_values, int64_t max_filter_size_in_bytes,
                                                  int32_t min_max_collection_limit, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out, "null argument");
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<DynamicFilterSourceOperatorFactory>(ctx->ctx.get(), operator_id, vec(types, type_count), vec(channels, channel_count), max_distinct_values,
                                                                    max_filter_size_in_bytes, min_max_collection_limit);
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_dynamic_filter_source_result(tgpu_operator *op, int32_t filter_channel, int32_t *kind, tgpu_output_page **values, int64_t *min, int64_t *max)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op && kind && values && min && max, "null argument");
        *values = nullptr;
        std::unique_ptr<OutputPage> page;
        dynamic_filter_result(op->op.get(), filter_channel, kind, &page, min, max);
        if (page) *values = release_output(std::move(page));
    });
}

int32_t tgpu_merge_pages_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types, int64_t min_page_size_in_bytes,
                                        int32_t min_row_count, int64_t max_page_size_in_bytes, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out, "null argument");
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<MergePagesOperatorFactory>(ctx->ctx.get(), operator_id, vec(types, type_count), min_page_size_in_bytes, min_row_count, max_page_size_in_bytes);
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_partitioned_output_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types, int32_t partition_channel_count,
                                               const int32_t *partition_channels, int32_t hash_channel, int32_t partition_count, int32_t replicates_any_row,
                                               int32_t null_channel, int32_t partition_function, tgpu_operator_factory **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out, "null argument");
        auto f = std::make_unique<tgpu_operator_factory>();
        f->f = std::make_unique<PartitionedOutputOperatorFactory>(ctx->ctx.get(), operator_id, vec(types, type_count), vec(partition_channels, partition_channel_count),
                                                                  hash_channel, partition_count, replicates_any_row != 0, null_channel, partition_function);
        f->ctx = ctx->ctx.get();
        retain(f->ctx);
        *out = f.release();
    });
}

int32_t tgpu_partitioned_output_poll(tgpu_operator *op, int32_t *partition, tgpu_output_page **out)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op && partition && out, "null argument");
        *out = nullptr;
        *partition = -1;
        std::unique_ptr<OutputPage> page;
        if (partitioned_output_poll(op->op.get(), partition, &page)) *out = release_output(std::move(page));
    });
}

int32_t tgpu_partitioned_output_info(tgpu_operator *op, int64_t *rows_added, int64_t *pages_added)
{
    return guard_on(ctx_of(op), [&] {
        TG_CHECK_ARG(op && rows_added && pages_added, "null argument");
        partitioned_output_info(op->op.get(), rows_added, pages_added);
    });
}

int32_t tgpu_serialize_page(tgpu_context *ctx, const tgpu_page *page, void *out, int64_t capacity, int64_t *out_len)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && page && out_len, "null argument");
        Context *c = ctx->ctx.get();
        DevicePage in = ingest_page(c, page);
        *out_len = serde::serialize(c, in, (uint8_t *)out, capacity);
    });
}

int32_t tgpu_deserialize_page(tgpu_context *ctx, const void *bytes, int64_t len, int32_t type_count, const int32_t *types, tgpu_output_page **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && bytes && out && (types || type_count == 0), "null argument");
        Context *c = ctx->ctx.get();
        *out = nullptr;
        *out = release_output(make_output(c, serde::deserialize(c, (const uint8_t *)bytes, len, types, type_count)));
    });
}

// ---- exchange -------------------------------------------------------------------------------------------------------
int32_t tgpu_exchange_unique_id(void *id_out)
{
    return guard([&] {
        TG_CHECK_ARG(id_out != nullptr, "id_out is null");
        rccl_unique_id(id_out);
    });
}

static int32_t exchange_create(tgpu_context *ctx, int32_t rank, int32_t world, std::unique_ptr<ExchangeTransport> (*make)(Context *, const void *, int, int), const void *arg,
                               tgpu_exchange **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && arg, "null argument");
        TG_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
        auto e = std::make_unique<tgpu_exchange>();
        e->ex = std::make_unique<Exchange>(ctx->ctx.get(), rank, world, make(ctx->ctx.get(), arg, rank, world));
        e->ctx = ctx->ctx.get();
        retain(e->ctx);
        *out = e.release();
    });
}

int32_t tgpu_exchange_create(tgpu_context *ctx, const void *unique_id, int32_t rank, int32_t world, tgpu_exchange **out)
{
    return exchange_create(ctx, rank, world, [](Context *c, const void *id, int r, int w) { return make_rccl_transport(c, id, r, w); }, unique_id, out);
}

int32_t tgpu_exchange_create_with_transport(tgpu_context *ctx, int32_t rank, int32_t world, const tgpu_exchange_transport *transport, tgpu_exchange **out)
{
    return exchange_create(ctx, rank, world, [](Context *c, const void *t, int, int w) { return make_callback_transport(c, (const tgpu_exchange_transport *)t, w); },
                           transport, out);
}

void tgpu_exchange_destroy(tgpu_exchange *ex)
{
    if (!ex) return;
    Context *c = ex->ctx;
    guard_on(c, [&] { ex->ex.reset(); });
    delete ex;
    if (c) drop(c);
}

int32_t tgpu_exchange_repartition(tgpu_exchange *ex, const tgpu_page *page, int32_t key_channel_count, const int32_t *key_channels, int32_t hash_channel,
                                  tgpu_output_page **out)
{
    return guard_on(ctx_of(ex), [&] {
        TG_CHECK_ARG(ex && page && out, "null argument");
        DevicePage in = ingest_page(ex->ctx, page);
        *out = release_output(make_output(ex->ctx, ex->ex->repartition(in, vec(key_channels, key_channel_count), hash_channel)));
    });
}

int32_t tgpu_exchange_partitioned_output(tgpu_exchange *ex, tgpu_operator *op, int32_t type_count, const int32_t *types, tgpu_output_page **out)
{
    return guard_on(ctx_of(ex), [&] {
        TG_CHECK_ARG(ex && op && op->op && out, "null argument");
        TG_CHECK_ARG(op->ctx == ex->ctx, "the operator and the exchange belong to different contexts");
        const int W = ex->ex->world();
        // checked BEFORE anything is polled out of the operator: a refused call leaves its pending pages where they are
        size_t most = 0;
        int32_t partitions = 0;
        partitioned_output_pending(op->op.get(), &most, &partitions);
        TG_CHECK_ARG(partitions == W, "the operator's partition count differs from the exchange's world size");
        TG_CHECK_ARG(most <= 1, "tgpu_exchange_partitioned_output moves the pages of ONE input page per call: call it after every add_input");
        std::vector<std::vector<std::unique_ptr<OutputPage>>> by_dest((size_t)W);
        for (;;) {
            int32_t part = -1;
            std::unique_ptr<OutputPage> pg;
            if (!partitioned_output_poll(op->op.get(), &part, &pg)) break;
            TG_CHECK_ARG(part >= 0 && part < W, "the operator's partition count differs from the exchange's world size");
            by_dest[(size_t)part].push_back(std::move(pg));
        }
        std::vector<const DevicePage *> per((size_t)W, nullptr);
        for (int r = 0; r < W; r++)
            if (!by_dest[(size_t)r].empty()) per[(size_t)r] = &by_dest[(size_t)r][0]->page;
        *out = release_output(make_output(ex->ctx, ex->ex->shuffle(vec(types, type_count), per)));
    });
}

int32_t tgpu_exchange_all_gather(tgpu_exchange *ex, const tgpu_page *page, tgpu_output_page **out)
{
    return guard_on(ctx_of(ex), [&] {
        TG_CHECK_ARG(ex && page && out, "null argument");
        DevicePage in = ingest_page(ex->ctx, page);
        *out = release_output(make_output(ex->ctx, ex->ex->all_gather(in)));
    });
}

static tgpu_output_page *one_column_page(Context *c, DeviceColumn col)
{
    DevicePage p;
    p.n = col.n;
    p.cols.push_back(std::move(col));
    return release_output(make_output(c, std::move(p)));
}

int32_t tgpu_orc_decode_long_column(tgpu_context *ctx, int32_t type, int32_t encoding, int32_t position_count, const void *present, int64_t present_len, const void *data,
                                    int64_t data_len, tgpu_output_page **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && (data || data_len == 0) && present_len >= 0 && data_len >= 0, "bad argument");
        *out = one_column_page(ctx->ctx.get(), orc::decode_long_column(ctx->ctx.get(), type, encoding, position_count, (const uint8_t *)present, present_len, (const uint8_t *)data, data_len));
    });
}

int32_t tgpu_orc_decode_boolean_column(tgpu_context *ctx, int32_t position_count, const void *present, int64_t present_len, const void *data, int64_t data_len,
                                       tgpu_output_page **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && (data || data_len == 0) && present_len >= 0 && data_len >= 0, "bad argument");
        *out = one_column_page(ctx->ctx.get(), orc::decode_boolean_column(ctx->ctx.get(), position_count, (const uint8_t *)present, present_len, (const uint8_t *)data, data_len));
    });
}

int32_t tgpu_orc_decode_double_column(tgpu_context *ctx, int32_t position_count, const void *present, int64_t present_len, const void *data, int64_t data_len,
                                      tgpu_output_page **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && (data || data_len == 0) && present_len >= 0 && data_len >= 0, "bad argument");
        *out = one_column_page(ctx->ctx.get(), orc::decode_double_column(ctx->ctx.get(), position_count, (const uint8_t *)present, present_len, (const uint8_t *)data, data_len));
    });
}

int32_t tgpu_orc_decode_dictionary_string_column(tgpu_context *ctx, int32_t encoding, int32_t position_count, const void *present, int64_t present_len, const void *data,
                                                 int64_t data_len, int32_t dictionary_size, const void *length_stream, int64_t length_len, const void *dictionary_data,
                                                 int64_t dictionary_data_len, tgpu_output_page **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && present_len >= 0 && data_len >= 0 && length_len >= 0 && dictionary_data_len >= 0, "bad argument");
        *out = one_column_page(ctx->ctx.get(), orc::decode_dictionary_string_column(ctx->ctx.get(), encoding, position_count, (const uint8_t *)present, present_len,
                                                                                  (const uint8_t *)data, data_len, dictionary_size, (const uint8_t *)length_stream, length_len,
                                                                                  (const uint8_t *)dictionary_data, dictionary_data_len));
    });
}

int32_t tgpu_orc_decode_direct_string_column(tgpu_context *ctx, int32_t encoding, int32_t position_count, const void *present, int64_t present_len, const void *data,
                                             int64_t data_len, const void *length_stream, int64_t length_len, tgpu_output_page **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && present_len >= 0 && data_len >= 0 && length_len >= 0 && (data || data_len == 0), "bad argument");
        *out = one_column_page(ctx->ctx.get(), orc::decode_direct_string_column(ctx->ctx.get(), encoding, position_count, (const uint8_t *)present, present_len, (const uint8_t *)data,
                                                                              data_len, (const uint8_t *)length_stream, length_len));
    });
}

int32_t tgpu_parquet_decode_data_page(tgpu_context *ctx, int32_t type, int32_t physical, int32_t encoding, int32_t position_count, const void *definition_levels,
                                      int64_t definition_levels_len, const void *values, int64_t values_len, const void *dictionary, int64_t dictionary_len,
                                      int32_t dictionary_count, tgpu_output_page **out)
{
    return guard_on(ctx_of(ctx), [&] {
        TG_CHECK_ARG(ctx && out && (values || values_len == 0) && (dictionary || dictionary_len == 0) && (definition_levels || definition_levels_len == 0), "bad argument");
        *out = one_column_page(ctx->ctx.get(), parquet::decode_data_page(ctx->ctx.get(), type, physical, encoding, position_count, (const uint8_t *)definition_levels, definition_levels_len,
                                                                         (const uint8_t *)values, values_len, (const uint8_t *)dictionary, dictionary_len, dictionary_count));
    });
}

int64_t tgpu_exchange_bytes_sent(tgpu_exchange *ex) { return ex ? ex->ex->bytes_sent() : TGPU_ERR_INVALID_ARGUMENT; }

}  // extern "C"
