// serde.hip -- the reference's wire / spill page format as the GPU ingest / egress format.
//
// A serialized page (M/execution/buffer/PagesSerdeUtil.java:63-71 writeSerializedPage, little endian):
//     int32 positionCount | byte markers (PageCodecMarker.java:24-25: 1 = COMPRESSED, 2 = ENCRYPTED) | int32 uncompressedSize |
//     int32 sizeInBytes | payload
// payload (PagesSerdeUtil.java:45-51 writeRawPage): int32 channelCount, then per channel the block
// (M/metadata/InternalBlockEncodingSerde.java:56-80): int32 nameLength | encoding name | body, with the bodies
//     LONG_ARRAY / INT_ARRAY / BYTE_ARRAY (S/block/LongArrayBlockEncoding.java:37-61 and siblings):
//         int32 positionCount | null bits | values of ALL positions when the block has no null vector, else
//         int32 nonNullCount | the values of the non-null positions only
//     VARIABLE_WIDTH (S/block/VariableWidthBlockEncoding.java:37-61):
//         int32 positionCount | positionCount x int32 END offsets | null bits | int32 totalLength | bytes
//     RLE (S/block/RunLengthBlockEncoding.java:31-41):       int32 positionCount | the one-position value block
//     DICTIONARY (S/block/DictionaryBlockEncoding.java:33-55): int32 positionCount | dictionary block | ids | 24 id bytes
//     null bits (S/block/EncoderUtil.java:33-71): byte mayHaveNull | when set, ceil(n / 8) bytes, position p = bit (7 - p % 8)
//
// Decoding puts every section where it finally lives: sections that are plain arrays (values without nulls, varchar bytes,
// end offsets) go from the host buffer straight into their aligned HBM column (one DMA each, no staging copy on the device);
// only the two transformed sections -- packed null bits and values compacted around nulls -- pass through a kernel (bit
// unpack; exclusive scan of the not-null flags + expansion).  Encoding is the mirror image: plain sections are read back from
// the column buffers as they are, null vectors are packed eight positions per byte, values are compacted by the same scan.
// Compressed / encrypted pages are refused (exchange.compression-enabled defaults to false; spill encryption is a CPU path).
#include "serde.h"

#include <mutex>

#include <cstring>

#include "kernels.h"

namespace tgpu {
namespace serde {
namespace {

constexpr int kBlock = 256;
constexpr int kHeaderBytes = 13;   // positionCount + markers + uncompressedSize + sizeInBytes

int grid_for(Context *ctx, int64_t n)
{
    int64_t blocks = ceil_div(n, kBlock);
    const int64_t cap = (int64_t)ctx->cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

// ---- kernels ----------------------------------------------------------------------------------------------------------
// packed null bits -> one byte per position (Java boolean[] valueIsNull) + the int32 not-null flags the scan consumes
__global__ void __launch_bounds__(kBlock) unpack_null_bits_kernel(const uint8_t *__restrict__ packed, int64_t n, uint8_t *__restrict__ nulls,
                                                                   int32_t *__restrict__ not_null)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint8_t is_null = (uint8_t)((packed[i >> 3] >> (7 - (int)(i & 7))) & 1);
        nulls[i] = is_null;
        not_null[i] = is_null ? 0 : 1;
    }
}

// one thread per output byte: eight positions, most significant bit first (EncoderUtil.java:45-57); the tail byte holds the
// last n % 8 positions in its high bits (EncoderUtil.java:62-70)
__global__ void __launch_bounds__(kBlock) pack_null_bits_kernel(const uint8_t *__restrict__ nulls, int64_t n, uint8_t *__restrict__ packed, int32_t *__restrict__ not_null)
{
    const int64_t bytes = (n + 7) / 8;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < bytes; b += (int64_t)gridDim.x * kBlock) {
        unsigned int v = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int64_t p = b * 8 + j;
            const bool is_null = p < n && nulls[p] != 0;
            v |= is_null ? (0x80u >> j) : 0u;
            if (p < n) not_null[p] = is_null ? 0 : 1;
        }
        packed[b] = (uint8_t)v;
    }
}

// values of the non-null positions -> one value per position (null positions read as 0: the reference leaves a copy of a
// neighbouring value there, LongArrayBlockEncoding.java:82-105, which no Block accessor may observe)
template <typename T>
__global__ void __launch_bounds__(kBlock) expand_values_kernel(const T *__restrict__ compact, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n,
                                                                T *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = nulls[i] ? (T)0 : compact[rank[i]];
}

template <typename T>
__global__ void __launch_bounds__(kBlock) compact_values_kernel(const T *__restrict__ values, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n,
                                                                 T *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (!nulls[i]) out[rank[i]] = values[i];
}

// VariableWidthBlockEncoding.java:46-51: the END offset of every position, relative to the block's first byte
__global__ void __launch_bounds__(kBlock) end_offsets_kernel(const int32_t *__restrict__ offsets, int64_t n, int32_t *__restrict__ ends)
{
    const int32_t base = offsets[0];
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) ends[i] = offsets[i + 1] - base;
}

// ---- host-side stream reader ------------------------------------------------------------------------------------------------
struct Reader {
    const uint8_t *p;
    int64_t len, at = 0;
    void need(int64_t n) const
    {
        if (n < 0 || at + n > len) fail(TGPU_ERR_INVALID_ARGUMENT, "serialized page is truncated");
    }
    int32_t i32()
    {
        need(4);
        int32_t v;
        memcpy(&v, p + at, 4);
        at += 4;
        return v;
    }
    uint8_t u8()
    {
        need(1);
        return p[at++];
    }
    const uint8_t *take(int64_t n)
    {
        need(n);
        const uint8_t *r = p + at;
        at += n;
        return r;
    }
    std::string name()
    {
        const int32_t l = i32();
        if (l < 0 || l > 64) fail(TGPU_ERR_INVALID_ARGUMENT, "bad block encoding name length");
        const uint8_t *b = take(l);
        return std::string((const char *)b, (size_t)l);
    }
};

BufferPtr upload_section(Context *ctx, const uint8_t *src, int64_t bytes)
{
    BufferPtr b = ctx->alloc((size_t)(bytes > 0 ? bytes : 1));
    if (bytes > 0) ctx->upload(b->ptr(), src, (size_t)bytes);
    return b;
}

// null bits of `n` positions: fills c.nulls (device) and returns the exclusive scan of the not-null flags (nullptr = no nulls)
// *null_count (optional): the number of set null bits among the n positions, counted on the host from the wire bytes
BufferPtr read_null_bits(Context *ctx, Reader &r, int64_t n, DeviceColumn &c, int64_t *null_count = nullptr)
{
    const uint8_t may_have_null = r.u8();
    if (null_count) *null_count = 0;
    if (!may_have_null) return nullptr;
    const int64_t bytes = (n + 7) / 8;
    const uint8_t *bits = r.take(bytes);
    if (null_count) {   // EncoderUtil.java:33-71: 8 positions per byte, MSB first; the tail byte's unused low bits are ignored
        int64_t cnt = 0;
        for (int64_t i = 0; i < n / 8; i++) cnt += __builtin_popcount(bits[i]);
        if (n % 8) cnt += __builtin_popcount(bits[n / 8] >> (8 - n % 8));
        *null_count = cnt;
    }
    BufferPtr packed = upload_section(ctx, bits, bytes);
    c.nulls_buf = ctx->alloc((size_t)(n > 0 ? n : 1));
    c.nulls = c.nulls_buf->as<uint8_t>();
    BufferPtr flags = ctx->alloc((size_t)(n > 0 ? n : 1) * 4), rank = ctx->alloc((size_t)(n > 0 ? n : 1) * 4), total = ctx->alloc(8);
    if (n > 0) {
        unpack_null_bits_kernel<<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(packed->as<uint8_t>(), n, c.nulls_buf->as<uint8_t>(), flags->as<int32_t>());
        check_launch("unpack_null_bits");
        k::exclusive_scan_i32(ctx, flags->as<int32_t>(), rank->as<int32_t>(), n, total->as<int64_t>());
    }
    return rank;
}

template <typename T> void expand(Context *ctx, const BufferPtr &compact, const BufferPtr &rank, const DeviceColumn &c, int64_t n, void *out)
{
    expand_values_kernel<T><<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(compact->as<T>(), rank->as<int32_t>(), c.nulls, n, (T *)out);
    check_launch("expand_values");
}

DeviceColumn read_block(Context *ctx, Reader &r, int32_t type, int depth);

DeviceColumn read_fixed(Context *ctx, Reader &r, int32_t type, int width)
{
    TG_CHECK_ARG(type_width(type) == width, "block encoding does not match the channel type");
    const int64_t n = r.i32();
    TG_CHECK_ARG(n >= 0, "negative position count");
    DeviceColumn c;
    c.type = type;
    c.n = n;
    int64_t null_count = 0;
    BufferPtr rank = read_null_bits(ctx, r, n, c, &null_count);
    if (!rank) {
        c.values_buf = upload_section(ctx, r.take(n * width), n * width);
        c.values = c.values_buf->ptr();
        return c;
    }
    const int64_t non_null = r.i32();
    // the expansion kernel indexes the compacted values by the rank among the clear null bits: the two must agree
    // (the JVM would fail with an IndexOutOfBounds on such a page; here it would be an out-of-bounds device read)
    TG_CHECK_ARG(non_null == n - null_count, "non-null position count does not match the null bits");
    BufferPtr compact = upload_section(ctx, r.take(non_null * width), non_null * width);
    c.values_buf = ctx->alloc((size_t)(n > 0 ? n : 1) * (size_t)width);
    c.values = c.values_buf->ptr();
    if (n > 0) {
        if (width == 8) expand<unsigned long long>(ctx, compact, rank, c, n, c.values_buf->ptr());
        else if (width == 4) expand<unsigned int>(ctx, compact, rank, c, n, c.values_buf->ptr());
        else expand<unsigned char>(ctx, compact, rank, c, n, c.values_buf->ptr());
    }
    return c;
}

DeviceColumn read_variable_width(Context *ctx, Reader &r, int32_t type)
{
    TG_CHECK_ARG(type == TGPU_VARCHAR, "block encoding does not match the channel type");
    const int64_t n = r.i32();
    TG_CHECK_ARG(n >= 0, "negative position count");
    DeviceColumn c;
    c.type = type;
    c.n = n;
    const uint8_t *ends = r.take(n * 4);
    {   // every kernel that touches a VARCHAR column trusts its offsets: validate them here, like the dictionary ids below
        int32_t prev = 0;
        for (int64_t i = 0; i < n; i++) {
            int32_t e;
            memcpy(&e, ends + i * 4, 4);
            TG_CHECK_ARG(e >= prev, "variable width offsets are negative or not ascending");
            prev = e;
        }
    }
    c.offsets_buf = ctx->alloc((size_t)(n + 1) * 4);
    c.offsets = c.offsets_buf->as<int32_t>();
    HIP_CHECK(hipMemsetAsync(c.offsets_buf->ptr(), 0, 4, ctx->stream()));
    if (n > 0) ctx->upload(c.offsets_buf->as<int32_t>() + 1, ends, (size_t)n * 4);   // offsets[0] = 0, offsets[i + 1] = end of position i
    (void)read_null_bits(ctx, r, n, c);
    const int64_t total = r.i32();
    TG_CHECK_ARG(total >= 0, "negative variable width block size");
    if (n > 0) {
        int32_t last;
        memcpy(&last, ends + (n - 1) * 4, 4);
        TG_CHECK_ARG(last == total, "variable width offsets do not match the block size");
    }
    c.values_buf = upload_section(ctx, r.take(total), total);
    c.values = c.values_buf->ptr();
    c.pool_bytes = total;
    c.pool_exact = true;
    return c;
}

DeviceColumn read_block(Context *ctx, Reader &r, int32_t type, int depth)
{
    TG_CHECK_ARG(depth < 4, "block encodings nested too deep");
    const std::string enc = r.name();
    if (enc == "LONG_ARRAY") return read_fixed(ctx, r, type, 8);
    if (enc == "INT_ARRAY") return read_fixed(ctx, r, type, 4);
    if (enc == "BYTE_ARRAY") return read_fixed(ctx, r, type, 1);
    if (enc == "VARIABLE_WIDTH") return read_variable_width(ctx, r, type);
    if (enc == "RLE" || enc == "DICTIONARY") {
        // flattened on the device like every other input (common.h: device columns are always flat): decode the nested block,
        // then gather it through the ids (RLE: id 0 for every position)
        const int64_t n = r.i32();
        TG_CHECK_ARG(n >= 0, "negative position count");
        DeviceColumn inner = read_block(ctx, r, type, depth + 1);
        BufferPtr ids = ctx->alloc((size_t)(n > 0 ? n : 1) * 4);
        if (enc == "RLE") {
            TG_CHECK_ARG(inner.n == 1, "RLE value block must have exactly one position");
            if (n > 0) k::fill_i32(ctx, ids->as<int32_t>(), 0, n);
        }
        else {
            const uint8_t *src = r.take(n * 4);
            for (int64_t i = 0; i < n; i++) {
                int32_t id;
                memcpy(&id, src + i * 4, 4);
                TG_CHECK_ARG(id >= 0 && id < inner.n, "dictionary id out of range");
            }
            if (n > 0) ctx->upload(ids->ptr(), src, (size_t)n * 4);
            (void)r.take(24);   // DictionaryId: unused here
        }
        return k::gather_column(ctx, inner, ids->as<int32_t>(), n, false);
    }
    fail(TGPU_ERR_NOT_SUPPORTED, "block encoding " + enc + " is not supported");
    return DeviceColumn{};
}

const char *encoding_name(int32_t type)
{
    switch (type) {
    case TGPU_BIGINT:
    case TGPU_DOUBLE: return "LONG_ARRAY";
    case TGPU_INTEGER:
    case TGPU_DATE: return "INT_ARRAY";
    case TGPU_BOOLEAN: return "BYTE_ARRAY";
    case TGPU_VARCHAR: return "VARIABLE_WIDTH";
    default: fail(TGPU_ERR_NOT_SUPPORTED, "type cannot be serialized"); return "";
    }
}

// LZ4 block decode (the format io.airlift.compress.lz4.Lz4Decompressor reads; M/execution/buffer/PagesSerde.java:153-165): a page's
// payload is ONE block = a sequence of (token, literal run, 2-byte little-endian offset, match length) records.  The records depend on
// each other, so one wave decodes the block: the token and the length bytes are read by all lanes alike (wave-uniform control flow), the
// literal and match copies run across the 64 lanes.  A match may overlap its own output (offset < length): every 64-byte chunk of it
// reads the `offset` bytes in front of the chunk, periodically repeated, which were written before the chunk -- through a 128 KB window
// of the most recent output in LDS (offsets are < 64 KB), so no lane ever reads global memory another lane has just written.
// status: 0 ok, 1 malformed input / output overrun.
constexpr int kLz4Window = 128 * 1024;
__global__ void __launch_bounds__(64) lz4_decode_kernel(const uint8_t *__restrict__ src, long long src_len, uint8_t *__restrict__ dst, long long dst_len, int *status)
{
    extern __shared__ uint8_t window[];
    const int lane = threadIdx.x;
    long long ip = 0, op = 0;
    int bad = 0;
    while (ip < src_len && !bad) {
        const unsigned token = src[ip++];
        long long lit = token >> 4;
        if (lit == 15) {
            unsigned b;
            do {
                if (ip >= src_len) { bad = 1; break; }
                b = src[ip++];
                lit += b;
            } while (b == 255);
        }
        if (bad || ip + lit > src_len || op + lit > dst_len) { bad = 1; break; }
        for (long long i = lane; i < lit; i += 64) {
            const uint8_t v = src[ip + i];
            dst[op + i] = v;
            window[(op + i) & (kLz4Window - 1)] = v;
        }
        ip += lit;
        op += lit;
        __syncthreads();
        if (ip >= src_len) break;   // the last record of a block has no match part
        if (ip + 2 > src_len) { bad = 1; break; }
        const long long offset = (long long)src[ip] | ((long long)src[ip + 1] << 8);
        ip += 2;
        long long ml = token & 15;
        if (ml == 15) {
            unsigned b;
            do {
                if (ip >= src_len) { bad = 1; break; }
                b = src[ip++];
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if (bad || offset == 0 || offset > op || op + ml > dst_len) { bad = 1; break; }
        for (long long cs = 0; cs < ml; cs += 64) {
            const long long i = cs + lane;
            uint8_t v = 0;
            if (i < ml) v = window[(op + cs - offset + (long long)(lane % offset)) & (kLz4Window - 1)];
            __syncthreads();
            if (i < ml) {
                dst[op + i] = v;
                window[(op + i) & (kLz4Window - 1)] = v;
            }
            __syncthreads();
        }
        op += ml;
    }
    if (lane == 0) *status = (bad || op != dst_len) ? 1 : 0;
}

// decompresses an LZ4 block on the device and returns the payload in host memory (the block headers are parsed on the host)
std::vector<uint8_t> lz4_decode_on_device(Context *ctx, const uint8_t *compressed, int64_t compressed_len, int64_t uncompressed_len)
{
    TG_CHECK_ARG(uncompressed_len >= 0 && uncompressed_len <= 0x7fffffffLL, "bad uncompressed size");
    std::vector<uint8_t> out((size_t)uncompressed_len);
    if (uncompressed_len == 0) return out;
    BufferPtr src = upload_section(ctx, compressed, compressed_len), dst = ctx->alloc((size_t)uncompressed_len), status = ctx->alloc_zero(4);
    {
        // the attribute belongs to the kernel's code object on ONE device: set once per device, under a lock (operators of several
        // contexts / devices decode concurrently)
        static std::mutex mu;
        static std::vector<bool> configured;
        std::lock_guard<std::mutex> lock(mu);
        const size_t d = (size_t)ctx->device();
        if (configured.size() <= d) configured.resize(d + 1, false);
        if (!configured[d]) {
            HIP_CHECK(hipFuncSetAttribute((const void *)lz4_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLz4Window));
            configured[d] = true;
        }
    }
    {
        ProfileScope ps(ctx, "lz4_decode");
        lz4_decode_kernel<<<1, 64, kLz4Window, ctx->stream()>>>(src->as<uint8_t>(), compressed_len, dst->as<uint8_t>(), uncompressed_len, status->as<int>());
        check_launch("lz4_decode");
    }
    TG_CHECK_ARG(ctx->read_scalar(status->as<int>()) == 0, "malformed LZ4 block in a compressed serialized page");
    ctx->download(out.data(), dst->ptr(), (size_t)uncompressed_len);
    return out;
}

}  // namespace

DevicePage deserialize(Context *ctx, const uint8_t *bytes, int64_t len, const int32_t *types, int32_t type_count)
{
    TG_CHECK_ARG(bytes != nullptr && len >= kHeaderBytes, "serialized page is truncated");
    Reader h{bytes, len};
    const int32_t positions = h.i32();
    const uint8_t markers = h.u8();
    const int32_t uncompressed = h.i32(), size = h.i32();
    TG_CHECK_ARG(positions >= 0 && size >= 0 && (int64_t)kHeaderBytes + size <= len, "serialized page is truncated");
    if (markers & 2) fail(TGPU_ERR_NOT_SUPPORTED, "encrypted serialized pages are not supported");
    std::vector<uint8_t> inflated;
    const uint8_t *payload = bytes + kHeaderBytes;
    int64_t payload_len = size;
    if (markers & 1) {   // COMPRESSED (PagesSerde.java:153-165): the payload is one LZ4 block of `uncompressed` bytes
        TG_CHECK_ARG(uncompressed >= 0, "negative uncompressed size");
        inflated = lz4_decode_on_device(ctx, payload, size, uncompressed);
        payload = inflated.data();
        payload_len = uncompressed;
    }
    else TG_CHECK_ARG(uncompressed == size, "uncompressed size differs from the payload size of an uncompressed page");
    Reader r{payload, payload_len};
    const int32_t channels = r.i32();
    TG_CHECK_ARG(channels == type_count, "serialized page has a different channel count than the expected types");
    DevicePage page;
    page.n = positions;
    for (int32_t ch = 0; ch < channels; ch++) {
        DeviceColumn c = read_block(ctx, r, types[ch], 0);
        TG_CHECK_ARG(c.n == positions, "block position count differs from the page's");
        page.cols.push_back(std::move(c));
    }
    TG_CHECK_ARG(r.at == r.len, "trailing bytes after the last block");
    // the uploads read the caller's buffer asynchronously only until hipMemcpyAsync returns (pageable source); kernels may still
    // be running, but they only touch library-owned buffers: nothing to wait for here
    if (!inflated.empty()) ctx->sync();   // (the inflated payload is ours and dies with this frame)
    return page;
}

// One pass plans the layout (needs the non-null counts: one batched read-back), a second one writes the host-side scalars and
// queues the section transfers; a single synchronisation at the end.
int64_t serialize(Context *ctx, const DevicePage &page, uint8_t *out, int64_t capacity)
{
    TG_CHECK_ARG(page.n <= 0x7fffffffLL, "page too large");
    const int64_t n = page.n;
    if (out == nullptr) {
        // capacity query: an upper bound (every position non-null), no kernel runs
        int64_t bound = kHeaderBytes + 4;
        for (const DeviceColumn &c : page.cols) {
            bound += 4 + (int64_t)strlen(encoding_name(c.type)) + 4 + 1 + (c.nulls ? (n + 7) / 8 : 0);
            bound += c.type == TGPU_VARCHAR ? n * 4 + 4 + c.pool_bytes : (c.nulls ? 4 : 0) + n * type_width(c.type);
        }
        return bound;
    }
    struct Plan {
        BufferPtr packed, rank, total, compact, ends;
        int32_t first = 0, last = 0;   // VARCHAR: offsets[0], offsets[n]
    };
    std::vector<Plan> plans(page.cols.size());
    std::vector<Context::Transfer> counts;
    std::vector<int64_t> totals(page.cols.size(), 0);
    for (size_t ch = 0; ch < page.cols.size(); ch++) {
        const DeviceColumn &c = page.cols[ch];
        Plan &p = plans[ch];
        TG_CHECK_ARG(c.n == n, "block position count differs from the page's");
        if (c.nulls && n > 0) {
            BufferPtr flags = ctx->alloc((size_t)n * 4);
            p.packed = ctx->alloc((size_t)((n + 7) / 8));
            p.rank = ctx->alloc((size_t)n * 4);
            p.total = ctx->alloc(8);
            pack_null_bits_kernel<<<grid_for(ctx, (n + 7) / 8), kBlock, 0, ctx->stream()>>>(c.nulls, n, p.packed->as<uint8_t>(), flags->as<int32_t>());
            check_launch("pack_null_bits");
            k::exclusive_scan_i32(ctx, flags->as<int32_t>(), p.rank->as<int32_t>(), n, p.total->as<int64_t>());
            if (c.type != TGPU_VARCHAR) {
                const int w = type_width(c.type);
                p.compact = ctx->alloc((size_t)n * (size_t)w);
                if (w == 8)
                    compact_values_kernel<unsigned long long><<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>((const unsigned long long *)c.values, p.rank->as<int32_t>(), c.nulls, n,
                                                                                                             p.compact->as<unsigned long long>());
                else if (w == 4)
                    compact_values_kernel<unsigned int><<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>((const unsigned int *)c.values, p.rank->as<int32_t>(), c.nulls, n,
                                                                                                       p.compact->as<unsigned int>());
                else
                    compact_values_kernel<unsigned char><<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>((const unsigned char *)c.values, p.rank->as<int32_t>(), c.nulls, n,
                                                                                                        p.compact->as<unsigned char>());
                check_launch("compact_values");
                counts.push_back({&totals[ch], p.total->ptr(), 8});
            }
        }
        if (c.type == TGPU_VARCHAR && n > 0) {
            p.ends = ctx->alloc((size_t)n * 4);
            end_offsets_kernel<<<grid_for(ctx, n), kBlock, 0, ctx->stream()>>>(c.offsets, n, p.ends->as<int32_t>());
            check_launch("end_offsets");
            counts.push_back({&p.first, c.offsets, 4});   // (a region view of a larger block does not start at byte 0)
            counts.push_back({&p.last, c.offsets + n, 4});
        }
    }
    if (!counts.empty()) ctx->download_batch(counts);
    // layout
    int64_t at = kHeaderBytes + 4;
    struct Section { int64_t at; const void *src; int64_t bytes; };
    std::vector<Section> sections;
    std::vector<std::pair<int64_t, std::vector<uint8_t>>> scalars;   // host-written pieces: (offset, bytes)
    auto put = [&](const void *src, size_t bytes) {
        std::vector<uint8_t> v((const uint8_t *)src, (const uint8_t *)src + bytes);
        scalars.emplace_back(at, std::move(v));
        at += (int64_t)bytes;
    };
    auto put_i32 = [&](int64_t v) {
        const int32_t x = (int32_t)v;
        put(&x, 4);
    };
    auto section = [&](const void *src, int64_t bytes) {
        if (bytes > 0) sections.push_back({at, src, bytes});
        at += bytes;
    };
    for (size_t ch = 0; ch < page.cols.size(); ch++) {
        const DeviceColumn &c = page.cols[ch];
        const Plan &p = plans[ch];
        const char *enc = encoding_name(c.type);
        put_i32((int64_t)strlen(enc));
        put(enc, strlen(enc));
        put_i32(n);
        const uint8_t may_have_null = c.nulls ? 1 : 0;
        if (c.type == TGPU_VARCHAR) {
            section(p.ends ? p.ends->ptr() : nullptr, n * 4);
            put(&may_have_null, 1);
            if (c.nulls) section(p.packed ? p.packed->ptr() : nullptr, (n + 7) / 8);
            put_i32((int64_t)p.last - p.first);   // block bytes = [offsets[0], offsets[n]) of the pool
            section((const uint8_t *)c.values + p.first, (int64_t)p.last - p.first);
        }
        else {
            const int w = type_width(c.type);
            put(&may_have_null, 1);
            if (!c.nulls) section(c.values, n * w);
            else {
                section(p.packed ? p.packed->ptr() : nullptr, (n + 7) / 8);
                put_i32(totals[ch]);
                section(p.compact ? p.compact->ptr() : nullptr, totals[ch] * w);
            }
        }
    }
    const int64_t total_bytes = at;
    if (total_bytes > capacity) fail(TGPU_ERR_INVALID_ARGUMENT, "output buffer too small for the serialized page: " + std::to_string(total_bytes) + " bytes needed");
    if (total_bytes - kHeaderBytes > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "serialized page cannot exceed 2GB");
    const int32_t header[3] = {(int32_t)n, (int32_t)(total_bytes - kHeaderBytes), (int32_t)(total_bytes - kHeaderBytes)};
    memcpy(out, &header[0], 4);
    out[4] = 0;   // no markers
    memcpy(out + 5, &header[1], 4);
    memcpy(out + 9, &header[2], 4);
    const int32_t channels = (int32_t)page.cols.size();
    memcpy(out + kHeaderBytes, &channels, 4);
    for (auto &s : scalars) memcpy(out + s.first, s.second.data(), s.second.size());
    // small sections share one pinned staging round trip; large ones are copied straight into the caller's buffer (the runtime
    // pipelines a pageable destination through its own staging -- an extra host memcpy of the whole page would cost more)
    std::vector<Context::Transfer> small;
    bool direct = false;
    for (auto &s : sections) {
        if (s.bytes < (1 << 20)) small.push_back({out + s.at, s.src, (size_t)s.bytes});
        else {
            HIP_CHECK(hipMemcpyAsync(out + s.at, s.src, (size_t)s.bytes, hipMemcpyDeviceToHost, ctx->stream()));
            direct = true;
        }
    }
    if (!small.empty()) ctx->download_batch(small);
    else if (direct) ctx->sync();
    return total_bytes;
}

}  // namespace serde
}  // namespace tgpu
