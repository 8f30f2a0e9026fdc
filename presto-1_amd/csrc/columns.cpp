// columns.cpp -- ingest of tgpu_page blocks (host or device memory; flat / dictionary / RLE) into flat HBM columns.
// This is the "Java heap -> HBM" step of the boundary (SURVEY.md hard part 3): the JNI shim pins the block's primitive
// arrays for the duration of the call and passes them here; nothing is retained on the host side.
#include "common.h"

namespace tgpu {

namespace {

struct HostFlat {
    int32_t type;
    int64_t n;
    std::vector<uint8_t> values;   // fixed width or byte pool
    std::vector<uint8_t> nulls;    // empty = none
    std::vector<int32_t> offsets;  // varchar
};

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
    TG_CHECK_ARG(b->encoding >= TGPU_FLAT && b->encoding <= TGPU_RLE, "unknown block encoding");
}

// host-side flatten of dictionary / RLE blocks (DictionaryBlock.java:40-100, RunLengthEncodedBlock.java:30-70)
HostFlat flatten_host(const tgpu_block *b)
{
    check_block(b);
    TG_CHECK_ARG(b->memory == TGPU_HOST, "flatten_host needs host memory");
    HostFlat out;
    out.type = b->type;
    out.n = b->position_count;
    const int w = type_width(b->type);
    if (b->encoding == TGPU_FLAT) {
        if (b->type == TGPU_VARCHAR) {
            TG_CHECK_ARG(b->offsets != nullptr || b->position_count == 0, "varchar block without offsets");
            out.offsets.assign(b->position_count + 1, 0);
            if (b->position_count) {
                int32_t base = b->offsets[0];
                for (int64_t i = 0; i <= b->position_count; i++) out.offsets[i] = b->offsets[i] - base;
                int64_t bytes = out.offsets[b->position_count];
                out.values.assign((const uint8_t *)b->values + base, (const uint8_t *)b->values + base + bytes);
            }
        }
        else {
            out.values.assign((const uint8_t *)b->values, (const uint8_t *)b->values + (size_t)b->position_count * w);
        }
        if (b->nulls && any_set(b->nulls, b->position_count)) out.nulls.assign(b->nulls, b->nulls + b->position_count);
        return out;
    }
    TG_CHECK_ARG(b->dictionary != nullptr, "dictionary / RLE block without a value block");
    HostFlat dict = flatten_host(b->dictionary);
    TG_CHECK_ARG(dict.type == b->type, "dictionary type mismatch");
    const int64_t n = b->position_count;
    std::vector<int32_t> ids((size_t)n, 0);
    if (b->encoding == TGPU_DICTIONARY) {
        TG_CHECK_ARG(b->ids != nullptr || n == 0, "dictionary block without ids");
        for (int64_t i = 0; i < n; i++) {
            TG_CHECK_ARG(b->ids[i] >= 0 && b->ids[i] < dict.n, "dictionary id out of range");
            ids[i] = b->ids[i];
        }
    }
    else {
        TG_CHECK_ARG(dict.n == 1, "RLE value block must have exactly one position");
    }
    if (!dict.nulls.empty()) {
        out.nulls.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) out.nulls[i] = dict.nulls[ids[i]];
        if (!any_set(out.nulls.data(), n)) out.nulls.clear();
    }
    if (b->type == TGPU_VARCHAR) {
        out.offsets.assign(n + 1, 0);
        int64_t total = 0;
        for (int64_t i = 0; i < n; i++) {
            total += dict.offsets[ids[i] + 1] - dict.offsets[ids[i]];
            if (total > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "variable width block cannot exceed 2GB");
            out.offsets[i + 1] = (int32_t)total;
        }
        out.values.resize((size_t)total);
        for (int64_t i = 0; i < n; i++) {
            int32_t a = dict.offsets[ids[i]], len = dict.offsets[ids[i] + 1] - a;
            if (len) memcpy(out.values.data() + out.offsets[i], dict.values.data() + a, (size_t)len);
        }
    }
    else {
        out.values.resize((size_t)n * w);
        for (int64_t i = 0; i < n; i++) memcpy(out.values.data() + (size_t)i * w, dict.values.data() + (size_t)ids[i] * w, (size_t)w);
    }
    return out;
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

DeviceColumn ingest_block(Context *ctx, const tgpu_block *b)
{
    check_block(b);
    const int64_t n = b->position_count;
    if (b->memory == TGPU_DEVICE) {
        if (b->encoding != TGPU_FLAT) fail(TGPU_ERR_NOT_SUPPORTED, "device-resident dictionary / RLE blocks are not supported yet");
        DeviceColumn c;  // borrowed: valid for the duration of the call (operators that retain input copy it)
        c.type = b->type;
        c.n = n;
        c.values = b->values;
        c.nulls = b->nulls;
        c.offsets = b->offsets;
        if (b->type == TGPU_VARCHAR) {
            TG_CHECK_ARG(b->offsets != nullptr, "varchar block without offsets");
            int32_t ends[2] = {0, 0};
            if (n > 0) {
                ctx->download(&ends[0], b->offsets, 4);
                ctx->download(&ends[1], b->offsets + n, 4);
            }
            c.pool_bytes = ends[1];  // offsets are absolute into the pool
        }
        return c;
    }
    if (b->encoding == TGPU_FLAT && b->type != TGPU_VARCHAR) {
        const uint8_t *nulls = (b->nulls && any_set(b->nulls, n)) ? b->nulls : nullptr;
        TG_CHECK_ARG(b->values != nullptr || n == 0, "block without values");
        return upload_flat(ctx, b->type, n, b->values, n * type_width(b->type), nulls, nullptr);
    }
    HostFlat f = flatten_host(b);
    DeviceColumn c = upload_flat(ctx, f.type, f.n, f.values.data(), (int64_t)f.values.size(), f.nulls.empty() ? nullptr : f.nulls.data(),
                                 f.offsets.empty() ? nullptr : f.offsets.data());
    // the temporaries above die at return: make sure the async copies have consumed them
    ctx->sync();
    return c;
}

DevicePage ingest_page(Context *ctx, const tgpu_page *page)
{
    TG_CHECK_ARG(page != nullptr, "page is null");
    TG_CHECK_ARG(page->position_count >= 0 && page->channel_count >= 0, "bad page header");
    DevicePage out;
    out.n = page->position_count;
    out.cols.reserve((size_t)page->channel_count);
    bool any_host = false;
    for (int32_t c = 0; c < page->channel_count; c++) {
        TG_CHECK_ARG(page->blocks[c].position_count == page->position_count, "block position count differs from the page's");
        any_host |= page->blocks[c].memory == TGPU_HOST;
        out.cols.push_back(ingest_block(ctx, &page->blocks[c]));
    }
    // ownership rule: the caller's (Java heap) arrays are only valid during the call, so the H2D copies must have
    // consumed them before we return
    if (any_host) ctx->sync();
    return out;
}

}  // namespace tgpu
