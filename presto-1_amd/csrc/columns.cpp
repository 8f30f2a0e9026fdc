// columns.cpp -- ingest of tgpu_page blocks (host or device memory; flat / dictionary / RLE) into flat HBM columns.
// This is the "Java heap -> HBM" step of the boundary (SURVEY.md hard part 3): the JNI shim pins the block's primitive
// arrays for the duration of the call and passes them here; nothing is retained on the host side.
#include "common.h"
#include "kernels.h"

#include <array>

namespace tgpu {

namespace {

bool any_set(const uint8_t *p, int64_t n)
{
    for (int64_t i = 0; i < n; i++)
        if (p[i]) return true;
    return false;
}

void check_block(const tgpu_block *b)
{
    TG_CHECK_ARG(b != nullptr, "block is null");
    TG_CHECK_ARG(valid_type(b->type), "unknown block type");
    TG_CHECK_ARG(b->position_count >= 0, "negative position count");
    TG_CHECK_ARG(b->encoding >= TGPU_FLAT && b->encoding <= TGPU_LAZY, "unknown block encoding");
    if (b->encoding == TGPU_LAZY) fail(TGPU_ERR_NOT_SUPPORTED, "a lazy block outside a page source: load it first (Page.getLoadedPage)");
}

DeviceColumn upload_flat(Context *ctx, int32_t type, int64_t n, const void *values, int64_t value_bytes, const uint8_t *nulls, const int32_t *offsets)
{
    DeviceColumn c;
    c.type = type;
    c.n = n;
    c.values_buf = ctx->alloc((size_t)(value_bytes > 0 ? value_bytes : 1));
    c.values = c.values_buf->ptr();
    ctx->upload(c.values_buf->ptr(), values, (size_t)value_bytes);
    if (type == TGPU_VARCHAR) {
        c.pool_bytes = value_bytes;
        c.pool_exact = true;
        c.offsets_buf = ctx->alloc((size_t)(n + 1) * 4);
        c.offsets = c.offsets_buf->as<int32_t>();
        ctx->upload(c.offsets_buf->ptr(), offsets, (size_t)(n + 1) * 4);
    }
    if (nulls) {
        c.nulls_buf = ctx->alloc((size_t)(n > 0 ? n : 1));
        c.nulls = c.nulls_buf->as<uint8_t>();
        ctx->upload(c.nulls_buf->ptr(), nulls, (size_t)n);
    }
    return c;
}

}  // namespace

// reads offsets[0] / offsets[n] of the borrowed device-resident VARCHAR columns in one batched transfer
static void resolve_varchar_ends(Context *ctx, std::vector<DeviceColumn *> cols)
{
    std::vector<std::array<int32_t, 2>> ends(cols.size(), {0, 0});
    std::vector<Context::Transfer> reads;
    for (size_t i = 0; i < cols.size(); i++) {
        reads.push_back({&ends[i][0], cols[i]->offsets, 4});
        reads.push_back({&ends[i][1], cols[i]->offsets + cols[i]->n, 4});
    }
    if (reads.empty()) return;
    ctx->download_batch(reads);
    for (size_t i = 0; i < cols.size(); i++) {
        cols[i]->pool_first = ends[i][0];
        cols[i]->pool_bytes = ends[i][1];
        cols[i]->pool_exact = true;
    }
}

void resolve_varchar_ends(Context *ctx, DevicePage &page)
{
    std::vector<DeviceColumn *> unresolved;
    for (DeviceColumn &c : page.cols)
        if (c.type == TGPU_VARCHAR && !c.pool_exact) unresolved.push_back(&c);
    resolve_varchar_ends(ctx, unresolved);
}

static DeviceColumn ingest_block_raw(Context *ctx, const tgpu_block *b);

DeviceColumn ingest_block(Context *ctx, const tgpu_block *b)
{
    DeviceColumn c = ingest_block_raw(ctx, b);
    if (c.type == TGPU_VARCHAR && !c.pool_exact) resolve_varchar_ends(ctx, {&c});
    return c;
}

static DeviceColumn ingest_block_raw(Context *ctx, const tgpu_block *b)
{
    check_block(b);
    const int64_t n = b->position_count;
    if (b->encoding != TGPU_FLAT) {
        // DictionaryBlock.java:40-100 / RunLengthEncodedBlock.java:30-70: the value block is ingested (recursively) and gathered
        // through the ids on the device -- device columns are always flat (common.h)
        TG_CHECK_ARG(b->dictionary != nullptr, "dictionary / RLE block without a value block");
        DeviceColumn dict = ingest_block(ctx, b->dictionary);
        TG_CHECK_ARG(dict.type == b->type, "dictionary type mismatch");
        BufferPtr ids_buf;
        const int32_t *ids = nullptr;
        if (b->encoding == TGPU_RLE) {
            TG_CHECK_ARG(dict.n == 1, "RLE value block must have exactly one position");
            ids_buf = ctx->alloc((size_t)(n > 0 ? n : 1) * 4);
            if (n > 0) k::fill_i32(ctx, ids_buf->as<int32_t>(), 0, n);
            ids = ids_buf->as<int32_t>();
        }
        else {
            TG_CHECK_ARG(b->ids != nullptr || n == 0, "dictionary block without ids");
            if (b->memory == TGPU_HOST) {
                for (int64_t i = 0; i < n; i++) TG_CHECK_ARG(b->ids[i] >= 0 && b->ids[i] < dict.n, "dictionary id out of range");
                ids_buf = ctx->alloc((size_t)(n > 0 ? n : 1) * 4);
                if (n > 0) ctx->upload(ids_buf->ptr(), b->ids, (size_t)n * 4);
                ids = ids_buf->as<int32_t>();
            }
            else ids = b->ids;   // device-resident ids are trusted (the producer was a kernel of this library)
        }
        ctx->flush_ingest();     // the dictionary and the ids must be in place before the gather reads them
        return k::gather_column(ctx, dict, ids, n, false);
    }
    if (b->memory == TGPU_DEVICE) {
        DeviceColumn c;  // borrowed: valid for the duration of the call (operators that retain input copy it)
        c.type = b->type;
        c.n = n;
        c.values = b->values;
        c.nulls = b->nulls;
        c.offsets = b->offsets;
        if (b->type == TGPU_VARCHAR) {
            TG_CHECK_ARG(b->offsets != nullptr, "varchar block without offsets");
            // offsets are absolute into the pool: pool_first / pool_bytes = offsets[0] / offsets[n], read back by the caller
            // (ingest_page batches the reads of all VARCHAR channels of a page into one round trip; see resolve_varchar_ends)
            c.pool_exact = n == 0;
        }
        return c;
    }
    const uint8_t *nulls = (b->nulls && any_set(b->nulls, n)) ? b->nulls : nullptr;
    if (b->type != TGPU_VARCHAR) {
        TG_CHECK_ARG(b->values != nullptr || n == 0, "block without values");
        return upload_flat(ctx, b->type, n, b->values, n * type_width(b->type), nulls, nullptr);
    }
    // VARCHAR: the block's bytes are [offsets[0], offsets[n]) of its slice (VariableWidthBlock.java:38-83); device offsets start at 0
    TG_CHECK_ARG(b->offsets != nullptr || n == 0, "varchar block without offsets");
    const int32_t zero = 0;
    const int32_t base = n > 0 ? b->offsets[0] : 0;
    const int64_t bytes = n > 0 ? (int64_t)b->offsets[n] - base : 0;
    TG_CHECK_ARG(bytes >= 0, "varchar offsets are not ascending");
    if (base == 0) return upload_flat(ctx, b->type, n, b->values, bytes, nulls, n > 0 ? b->offsets : &zero);
    std::vector<int32_t> rebased((size_t)n + 1);
    for (int64_t i = 0; i <= n; i++) rebased[(size_t)i] = b->offsets[i] - base;
    DeviceColumn c = upload_flat(ctx, b->type, n, (const uint8_t *)b->values + base, bytes, nulls, rebased.data());
    ctx->sync();   // `rebased` dies at return: the async copy must have consumed it
    return c;
}

// DictionaryBlock / RunLengthEncodedBlock kept as (flat dictionary, device ids) instead of being flattened: the dictionary-aware
// operators evaluate once per dictionary entry (M/operator/project/DictionaryAwarePageFilter.java:56-110).  RLE = one entry, ids all 0.
void ingest_dictionary(Context *ctx, const tgpu_block *b, DeviceColumn &dictionary, BufferPtr &ids)
{
    check_block(b);
    TG_CHECK_ARG(b->encoding != TGPU_FLAT && b->dictionary != nullptr, "not a dictionary / RLE block");
    const int64_t n = b->position_count;
    dictionary = ingest_block(ctx, b->dictionary);
    TG_CHECK_ARG(dictionary.type == b->type, "dictionary type mismatch");
    ids = ctx->alloc((size_t)(n > 0 ? n : 1) * 4);
    if (b->encoding == TGPU_RLE) {
        TG_CHECK_ARG(dictionary.n == 1, "RLE value block must have exactly one position");
        if (n > 0) k::fill_i32(ctx, ids->as<int32_t>(), 0, n);
        return;
    }
    TG_CHECK_ARG(b->ids != nullptr || n == 0, "dictionary block without ids");
    if (n == 0) return;
    if (b->memory == TGPU_HOST) {
        for (int64_t i = 0; i < n; i++) TG_CHECK_ARG(b->ids[i] >= 0 && b->ids[i] < dictionary.n, "dictionary id out of range");
        ctx->upload(ids->ptr(), b->ids, (size_t)n * 4);
        ctx->sync();   // the caller's ids array is only valid during the call
    }
    else HIP_CHECK(hipMemcpyAsync(ids->ptr(), b->ids, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream()));
}

// A column that borrows caller memory (a device-resident input block: no owning buffers) is only valid during the call; an operator
// that lets such a column out again (an identity projection, a page passing through) gives it buffers of its own first.
void own_borrowed_columns(Context *ctx, DevicePage &page)
{
    for (DeviceColumn &c : page.cols) {
        if (c.n <= 0 || c.values_buf) continue;
        const size_t vbytes = (size_t)c.value_bytes();   // VARCHAR: the pool up to offsets[n] (offsets stay absolute)
        c.values_buf = ctx->alloc(vbytes ? vbytes : 1);
        if (vbytes) HIP_CHECK(hipMemcpyAsync(c.values_buf->ptr(), c.values, vbytes, hipMemcpyDeviceToDevice, ctx->stream()));
        c.values = c.values_buf->ptr();
        if (c.nulls) {
            c.nulls_buf = ctx->alloc((size_t)c.n);
            HIP_CHECK(hipMemcpyAsync(c.nulls_buf->ptr(), c.nulls, (size_t)c.n, hipMemcpyDeviceToDevice, ctx->stream()));
            c.nulls = c.nulls_buf->as<uint8_t>();
        }
        if (c.offsets) {
            c.offsets_buf = ctx->alloc((size_t)(c.n + 1) * 4);
            HIP_CHECK(hipMemcpyAsync(c.offsets_buf->ptr(), c.offsets, (size_t)(c.n + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream()));
            c.offsets = c.offsets_buf->as<int32_t>();
        }
    }
}

DevicePage ingest_page(Context *ctx, const tgpu_page *page, bool resolve_varchar)
{
    TG_CHECK_ARG(page != nullptr, "page is null");
    TG_CHECK_ARG(page->position_count >= 0 && page->channel_count >= 0, "bad page header");
    DevicePage out;
    out.n = page->position_count;
    out.cols.reserve((size_t)page->channel_count);
    bool any_host = false;
    size_t host_bytes = 0;   // what the flat host blocks of the page upload: staged through the double-buffered ingest ring when sizeable
    for (int32_t c = 0; c < page->channel_count; c++) {
        const tgpu_block &b = page->blocks[c];
        TG_CHECK_ARG(b.position_count == page->position_count, "block position count differs from the page's");
        any_host |= b.memory == TGPU_HOST;
        if (b.memory == TGPU_HOST && b.encoding == TGPU_FLAT && valid_type(b.type) && b.position_count > 0) {
            const size_t n = (size_t)b.position_count;
            if (b.type == TGPU_VARCHAR) host_bytes += (b.offsets ? (size_t)(b.offsets[n] - b.offsets[0]) : 0) + (n + 1) * 4;
            else host_bytes += n * (size_t)type_width(b.type);
            if (b.nulls) host_bytes += n;
        }
    }
    struct IngestScope {   // end_ingest also on the error path (it releases the context's ingest lock)
        Context *c;
        bool on;
        ~IngestScope() { if (on) c->end_ingest(); }
    } scope{ctx, any_host && ctx->begin_ingest(host_bytes)};
    const bool ringed = scope.on;
    for (int32_t c = 0; c < page->channel_count; c++) out.cols.push_back(ingest_block_raw(ctx, &page->blocks[c]));
    if (scope.on) {
        scope.on = false;
        ctx->end_ingest();   // waits for the transfers only: the kernels of the previous page keep running on the compute stream
    }
    std::vector<DeviceColumn *> unresolved;
    for (DeviceColumn &c : out.cols)
        if (resolve_varchar && c.type == TGPU_VARCHAR && !c.pool_exact) unresolved.push_back(&c);
    resolve_varchar_ends(ctx, unresolved);
    // ownership rule: the caller's (Java heap) arrays are only valid during the call, so the H2D copies must have
    // consumed them before we return (the batched read above has already waited for the stream)
    if (any_host && unresolved.empty() && !ringed) ctx->sync();
    return out;
}

HostColumn download_column(Context *ctx, const DeviceColumn &c)
{
    HostColumn h;
    h.type = c.type;
    h.n = c.n;
    h.has_nulls = c.nulls != nullptr;
    if (c.n <= 0) {
        if (c.type == TGPU_VARCHAR) h.offsets.assign(1, 0);
        return h;
    }
    if (c.nulls) {
        h.nulls.resize((size_t)c.n);
        ctx->download(h.nulls.data(), c.nulls, (size_t)c.n);
    }
    if (c.type == TGPU_VARCHAR) {
        h.offsets.resize((size_t)c.n + 1);
        ctx->download(h.offsets.data(), c.offsets, ((size_t)c.n + 1) * 4);
        const int32_t first = h.offsets[0], last = h.offsets[(size_t)c.n];
        h.values.resize((size_t)(last - first));
        if (last > first) ctx->download(h.values.data(), (const uint8_t *)c.values + first, (size_t)(last - first));
        for (int32_t &o : h.offsets) o -= first;
    }
    else {
        h.values.resize((size_t)c.n * type_width(c.type));
        ctx->download(h.values.data(), c.values, h.values.size());
    }
    return h;
}

DeviceColumn upload_column(Context *ctx, const HostColumn &h)
{
    DeviceColumn c;
    c.type = h.type;
    c.n = h.n;
    c.values_buf = ctx->alloc(h.values.empty() ? 8 : h.values.size());
    c.values = c.values_buf->ptr();
    if (!h.values.empty()) ctx->upload(c.values_buf->ptr(), h.values.data(), h.values.size());
    if (h.has_nulls && h.n > 0) {
        c.nulls_buf = ctx->alloc((size_t)h.n);
        c.nulls = c.nulls_buf->as<uint8_t>();
        ctx->upload(c.nulls_buf->ptr(), h.nulls.data(), (size_t)h.n);
    }
    if (h.type == TGPU_VARCHAR) {
        c.offsets_buf = ctx->alloc(h.offsets.size() * 4);
        c.offsets = c.offsets_buf->as<int32_t>();
        ctx->upload(c.offsets_buf->ptr(), h.offsets.data(), h.offsets.size() * 4);
        c.pool_first = 0;
        c.pool_bytes = (int64_t)h.values.size();
        c.pool_exact = true;
    }
    ctx->sync();   // the uploads read the caller's vectors
    return c;
}

}  // namespace tgpu
