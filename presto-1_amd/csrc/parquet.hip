// parquet.hip -- scan-side decode, the second columnar format the reference reads: the data pages of a flat Parquet column
// decoded on the device into flat HBM columns.  Reference (lib/trino-parquet/src/main/java/io/trino/parquet/): reader/PrimitiveColumnReader.java
// (readPageV1 / readPageV2 / initDataReader: levels, then the value reader of the page's encoding), reader/LevelRLEReader.java,
// ParquetEncoding.java (PLAIN / PLAIN_DICTIONARY / RLE_DICTIONARY / the DELTA encodings -> value readers and dictionaries), dictionary/*.java (PLAIN dictionary
// pages), reader/{Int,Long,Double,Boolean,Binary}ColumnReader.java.  The byte-level decoders those classes call are NOT in the reference tree:
// org.apache.parquet (parquet-mr) classes out of io.prestosql.hive:hive-apache 3.1.2-6, the shaded bundle the reference's root pom.xml:537-538 pins
// -- RunLengthBitPackingHybridDecoder, the Plain*ValuesReaders -- so their algorithm is restated from the public Parquet format specification (Encodings.md: "RLE / bit-packing
// hybrid", "PLAIN") and pinned on Apache Arrow's independent writer and reader of the same format (tests) and on the one Parquet data file the
// reference's own tests hold that these types can read (single_int_column/data.parquet).
//
// Pages arrive DECOMPRESSED in host memory (codecs and the thrift page headers stay with the file reader).  Like the ORC streams, a hybrid
// stream is a sequence of runs whose headers are variable length: the host walks the headers (a varint per run), the device decodes the values
// -- one wave per run, bit-packed values at lane-computed bit offsets (least significant bit first, little-endian bytes).
#include "parquet.h"

#include "kernels.h"

#include <rocprim/rocprim.hpp>

#include <cstring>

namespace tgpu {
namespace parquet {

namespace {

constexpr int kWave = 64;
enum Physical : int32_t { PQ_BOOLEAN = 0, PQ_INT32 = 1, PQ_INT64 = 2, PQ_DOUBLE = 5, PQ_BYTE_ARRAY = 6 };          // parquet.thrift Type
enum Encoding : int32_t { PQ_PLAIN = 0, PQ_PLAIN_DICTIONARY = 2, PQ_RLE = 3, PQ_DELTA_BINARY_PACKED = 5, PQ_DELTA_LENGTH_BYTE_ARRAY = 6, PQ_DELTA_BYTE_ARRAY = 7, PQ_RLE_DICTIONARY = 8 };                           // parquet.thrift Encoding

struct Run {
    int64_t in_off;    // bit-packed run: first byte of the packed values
    int64_t out_off;   // index of the run's first value
    int32_t count;     // values of the run (the last bit-packed run is cut at the number of values wanted: its tail is padding)
    int32_t packed;    // 1 = bit-packed, 0 = RLE
    int64_t value;     // RLE: the repeated value
};

// the run directory of an RLE / bit-packed hybrid stream (Encodings.md; RunLengthBitPackingHybridDecoder.readNext): header = ULEB128;
// bit 0 set: (header >> 1) groups of 8 bit-packed values; clear: a run of header >> 1 copies of one value of ceil(bit_width / 8) bytes
std::vector<Run> scan_hybrid(const uint8_t *bytes, int64_t len, int bit_width, int64_t want)
{
    TG_CHECK_ARG(bit_width >= 0 && bit_width <= 32, "Parquet hybrid stream: bit width out of range");
    std::vector<Run> runs;
    int64_t at = 0, total = 0;
    while (total < want) {
        if (at >= len) fail(TGPU_ERR_INVALID_ARGUMENT, "Parquet hybrid stream ends before the page's values do");
        uint64_t header = 0;
        for (int shift = 0;; shift += 7) {
            if (at >= len || shift > 35) fail(TGPU_ERR_INVALID_ARGUMENT, "Parquet hybrid stream: bad run header");
            const int b = bytes[at++];
            header |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) break;
        }
        Run r{};
        r.out_off = total;
        if (header & 1) {
            const int64_t values = (int64_t)(header >> 1) * 8, packed_bytes = (int64_t)(header >> 1) * bit_width;
            r.packed = 1;
            r.in_off = at;
            r.count = (int32_t)std::min<int64_t>(values, want - total);
            // (writers may cut the padding of the LAST group short: only the bytes the wanted values occupy must be there)
            if (at + ((int64_t)r.count * bit_width + 7) / 8 > len) fail(TGPU_ERR_INVALID_ARGUMENT, "Parquet hybrid stream: a bit-packed run is longer than the stream");
            at = std::min<int64_t>(at + packed_bytes, len);
        } else {
            const int64_t count = (int64_t)(header >> 1);
            const int width_bytes = (bit_width + 7) / 8;
            if (count == 0 || at + width_bytes > len) fail(TGPU_ERR_INVALID_ARGUMENT, "Parquet hybrid stream: bad RLE run");
            uint64_t v = 0;
            for (int i = 0; i < width_bytes; i++) v |= (uint64_t)bytes[at++] << (8 * i);
            r.value = (int64_t)v;
            r.count = (int32_t)std::min<int64_t>(count, want - total);
        }
        total += r.count;
        runs.push_back(r);
    }
    return runs;
}

__global__ void __launch_bounds__(256) hybrid_decode_kernel(const uint8_t *__restrict__ bytes, const Run *__restrict__ runs, int64_t n_runs, int bit_width, int32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long mask = bit_width >= 64 ? ~0ull : ((1ull << bit_width) - 1ull);
    for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_runs; w += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const Run r = runs[w];
        for (int i = lane; i < r.count; i += kWave) {
            long long v = r.value;
            if (r.packed) {
                const long long bit = (long long)i * bit_width;
                const uint8_t *p = bytes + r.in_off + (bit >> 3);
                unsigned long long word = 0;
#pragma unroll
                for (int k = 0; k < 5; k++) word |= (unsigned long long)p[k] << (8 * k);   // 32 bits at any bit offset span at most 5 bytes (the buffer is padded)
                v = (long long)((word >> (bit & 7)) & mask);
            }
            out[r.out_off + i] = (int32_t)v;
        }
    }
}

__global__ void __launch_bounds__(256) levels_to_flags_kernel(const int32_t *__restrict__ levels, int64_t n, uint8_t *__restrict__ nulls, int32_t *__restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int present = levels[i] != 0;
        nulls[i] = present ? 0 : 1;
        flags[i] = present;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) place_kernel(const T *__restrict__ compact, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n, T *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (nulls && nulls[i]) ? (T)0 : compact[rank ? rank[i] : i];
}

// int32 values (0 / 1) of the non-null rows -> a byte per row
__global__ void __launch_bounds__(256) place_narrow_kernel(const int32_t *__restrict__ compact, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n, uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (nulls && nulls[i]) ? (uint8_t)0 : (uint8_t)(compact[rank ? rank[i] : i] != 0);
}

// PLAIN BOOLEAN: one bit per value, least significant bit first
__global__ void __launch_bounds__(256) place_bits_kernel(const uint8_t *__restrict__ bits, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n, uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (nulls && nulls[i]) {
            out[i] = 0;
            continue;
        }
        const int64_t j = rank ? rank[i] : i;
        out[i] = (bits[j >> 3] >> (j & 7)) & 1;
    }
}

// dictionary ids of the non-null rows -> an id per row (-1 for a null row); an id outside the dictionary raises the error flag
__global__ void __launch_bounds__(256) place_ids_kernel(const int32_t *__restrict__ compact, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n,
                                                        int32_t dictionary_size, int32_t *__restrict__ out, unsigned int *error)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (nulls && nulls[i]) {
            out[i] = -1;
            continue;
        }
        const int32_t id = compact[rank ? rank[i] : i];
        if (id < 0 || id >= dictionary_size) {
            atomicOr(error, 1u);
            out[i] = -1;
        } else out[i] = id;
    }
}

// PLAIN BYTE_ARRAY: the values' bytes (behind their 4-byte length prefixes in the page) to their place in the block's byte pool
__global__ void __launch_bounds__(256) copy_strings_kernel(const uint8_t *__restrict__ page, const int32_t *__restrict__ src_off, const int32_t *__restrict__ rank,
                                                           const uint8_t *__restrict__ nulls, const int32_t *__restrict__ offsets, int64_t n, uint8_t *__restrict__ pool)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (nulls && nulls[i]) continue;
        const uint8_t *s = page + src_off[rank ? rank[i] : i];
        uint8_t *d = pool + offsets[i];
        const int len = offsets[i + 1] - offsets[i];
        for (int k = 0; k < len; k++) d[k] = s[k];
    }
}

int grid_for(Context *ctx, int64_t n)
{
    int64_t blocks = ceil_div(n, 256);
    const int64_t cap = (int64_t)ctx->cu_count() * 8;
    return (int)std::max<int64_t>(1, std::min<int64_t>(blocks, cap));
}

BufferPtr upload_padded(Context *ctx, const uint8_t *src, int64_t bytes)
{
    BufferPtr b = ctx->alloc((size_t)bytes + 16);
    if (bytes) ctx->upload(b->ptr(), src, (size_t)bytes);
    HIP_CHECK(hipMemsetAsync(static_cast<uint8_t *>(b->ptr()) + bytes, 0, 16, ctx->stream()));
    return b;
}

// `want` values of a hybrid stream as int32 on the device
BufferPtr decode_hybrid(Context *ctx, const uint8_t *bytes, int64_t len, int bit_width, int64_t want)
{
    BufferPtr out = ctx->alloc((size_t)(want > 0 ? want : 1) * 4);
    if (want == 0) return out;
    if (bit_width == 0) {   // (a dictionary of one entry: every id is 0 and the stream may be empty)
        HIP_CHECK(hipMemsetAsync(out->ptr(), 0, (size_t)want * 4, ctx->stream()));
        return out;
    }
    std::vector<Run> runs = scan_hybrid(bytes, len, bit_width, want);
    BufferPtr dbytes = upload_padded(ctx, bytes, len), druns = ctx->alloc(runs.size() * sizeof(Run));
    ctx->upload(druns->ptr(), runs.data(), runs.size() * sizeof(Run));
    ProfileScope ps(ctx, "parquet_hybrid_decode");
    hybrid_decode_kernel<<<grid_for(ctx, (int64_t)runs.size() * kWave), 256, 0, ctx->stream()>>>(dbytes->as<uint8_t>(), druns->as<Run>(), (int64_t)runs.size(), bit_width, out->as<int32_t>());
    check_launch("parquet_hybrid_decode");
    ctx->sync();   // `runs` (host) backs the upload
    return out;
}

struct Present {
    BufferPtr nulls, rank;
    int64_t non_null = 0;
};
Present decode_levels(Context *ctx, const uint8_t *def_levels, int64_t def_len, int64_t n)
{
    Present p;
    p.non_null = n;
    if (!def_levels || n == 0) return p;
    BufferPtr levels = decode_hybrid(ctx, def_levels, def_len, 1, n);
    p.nulls = ctx->alloc((size_t)n);
    p.rank = ctx->alloc((size_t)n * 4);
    BufferPtr flags = ctx->alloc((size_t)n * 4), total = ctx->alloc(8);
    levels_to_flags_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(levels->as<int32_t>(), n, p.nulls->as<uint8_t>(), flags->as<int32_t>());
    check_launch("parquet_levels");
    k::exclusive_scan_i32(ctx, flags->as<int32_t>(), p.rank->as<int32_t>(), n, total->as<int64_t>());
    p.non_null = ctx->read_scalar(total->as<int64_t>());
    return p;
}

// PLAIN BYTE_ARRAY values: where each value's bytes start in `bytes` and how long it is (a 4-byte little-endian length in front of each)
void scan_byte_arrays(const uint8_t *bytes, int64_t len, int64_t count, std::vector<int32_t> &src_off, std::vector<int32_t> &lengths)
{
    src_off.resize((size_t)count);
    lengths.resize((size_t)count);
    int64_t at = 0;
    for (int64_t i = 0; i < count; i++) {
        if (at + 4 > len) fail(TGPU_ERR_INVALID_ARGUMENT, "Parquet PLAIN BYTE_ARRAY section ends inside a length");
        uint32_t l;
        memcpy(&l, bytes + at, 4);
        at += 4;
        if (l > 0x7fffffffu || at + (int64_t)l > len) fail(TGPU_ERR_INVALID_ARGUMENT, "Parquet PLAIN BYTE_ARRAY value is longer than its section");
        src_off[(size_t)i] = (int32_t)at;
        lengths[(size_t)i] = (int32_t)l;
        at += l;
    }
}

int width_of(int32_t physical)
{
    return physical == PQ_INT32 ? 4 : (physical == PQ_INT64 || physical == PQ_DOUBLE) ? 8 : 0;
}

// `count` PLAIN values (no nulls among them) placed at the non-null rows of an n-row column
DeviceColumn plain_column(Context *ctx, int32_t type, int32_t physical, const uint8_t *bytes, int64_t len, int64_t n, const Present &p)
{
    DeviceColumn col;
    col.type = type;
    col.n = n;
    const uint8_t *nulls = p.nulls && p.non_null < n ? p.nulls->as<uint8_t>() : nullptr;
    const int32_t *rank = nulls ? p.rank->as<int32_t>() : nullptr;
    if (nulls) {
        col.nulls_buf = p.nulls;
        col.nulls = nulls;
    }
    if (physical == PQ_BYTE_ARRAY) {
        std::vector<int32_t> src_off, lengths;
        scan_byte_arrays(bytes, len, p.non_null, src_off, lengths);
        int64_t total_bytes = 0;
        for (int32_t l : lengths) total_bytes += l;
        TG_CHECK_ARG(total_bytes <= 0x7fffffffLL, "Parquet page holds more than 2 GB of string bytes");
        col.offsets_buf = ctx->alloc((size_t)(n + 1) * 4);
        col.offsets = col.offsets_buf->as<int32_t>();
        col.values_buf = ctx->alloc((size_t)(total_bytes > 0 ? total_bytes : 1));
        col.values = col.values_buf->ptr();
        col.pool_bytes = total_bytes;
        col.pool_exact = true;
        if (n == 0) {
            HIP_CHECK(hipMemsetAsync(col.offsets_buf->ptr(), 0, 4, ctx->stream()));
            return col;
        }
        BufferPtr dlen = ctx->alloc((size_t)std::max<int64_t>(p.non_null, 1) * 4), dsrc = ctx->alloc((size_t)std::max<int64_t>(p.non_null, 1) * 4), row_len = ctx->alloc((size_t)n * 4),
                  total = ctx->alloc(8), page = upload_padded(ctx, bytes, len);
        if (p.non_null > 0) {
            ctx->upload(dlen->ptr(), lengths.data(), (size_t)p.non_null * 4);
            ctx->upload(dsrc->ptr(), src_off.data(), (size_t)p.non_null * 4);
        }
        place_kernel<int32_t><<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(dlen->as<int32_t>(), rank, nulls, n, row_len->as<int32_t>());
        check_launch("parquet_place_lengths");
        k::exclusive_scan_i32(ctx, row_len->as<int32_t>(), const_cast<int32_t *>(col.offsets), n, total->as<int64_t>());
        const int32_t end = (int32_t)total_bytes;
        ctx->upload(const_cast<int32_t *>(col.offsets) + n, &end, 4);
        copy_strings_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(page->as<uint8_t>(), dsrc->as<int32_t>(), rank, nulls, col.offsets, n, col.values_buf->as<uint8_t>());
        check_launch("parquet_copy_strings");
        ctx->sync();   // the host vectors back the uploads
        return col;
    }
    if (physical == PQ_BOOLEAN) {
        TG_CHECK_ARG(len >= (p.non_null + 7) / 8, "Parquet PLAIN BOOLEAN section holds fewer bits than the page has non-null values");
        col.values_buf = ctx->alloc((size_t)(n > 0 ? n : 1));
        col.values = col.values_buf->ptr();
        if (n == 0) return col;
        BufferPtr bits = upload_padded(ctx, bytes, len);
        place_bits_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(bits->as<uint8_t>(), rank, nulls, n, col.values_buf->as<uint8_t>());
        check_launch("parquet_place_bits");
        return col;
    }
    const int w = width_of(physical);
    TG_CHECK_ARG(len >= p.non_null * w, "Parquet PLAIN section holds fewer values than the page has non-null positions");
    col.values_buf = ctx->alloc((size_t)(n > 0 ? n : 1) * (size_t)w);
    col.values = col.values_buf->ptr();
    if (n == 0) return col;
    if (!nulls) {
        ctx->upload(col.values_buf->ptr(), bytes, (size_t)n * (size_t)w);
        return col;
    }
    BufferPtr compact = ctx->alloc((size_t)std::max<int64_t>(p.non_null, 1) * (size_t)w);
    if (p.non_null > 0) ctx->upload(compact->ptr(), bytes, (size_t)p.non_null * (size_t)w);
    if (w == 8) place_kernel<long long><<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(compact->as<long long>(), rank, nulls, n, (long long *)col.values_buf->ptr());
    else place_kernel<int32_t><<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(compact->as<int32_t>(), rank, nulls, n, (int32_t *)col.values_buf->ptr());
    check_launch("parquet_place_values");
    return col;
}

// ---- DELTA_BINARY_PACKED (Encodings.md "Delta Encoding"; parquet-mr DeltaBinaryPackingValuesReader behind ParquetEncoding.java:146-154) -------
// header: block size, miniblocks per block, total count (ULEB128), first value (zigzag); blocks: min delta (zigzag), one bit-width byte per
// miniblock, the miniblocks' (delta - min delta) bit-packed least significant bit first.  The host walks the block headers (three varints per
// 128 values), the device unpacks every miniblock -- one lane per value -- and a prefix sum turns the deltas into values (wrapping 64-bit
// arithmetic; INT32 keeps the low half).
struct Mini {
    int64_t in_off;      // first byte of the miniblock's packed deltas
    int64_t out_off;     // index of the value its first delta produces (>= 1: value 0 is the header's first value)
    int64_t min_delta;
    int32_t count;       // deltas of the miniblock that are values of the page (the last one may be padded)
    int32_t width;
};

__global__ void __launch_bounds__(256) delta_unpack_kernel(const uint8_t *__restrict__ bytes, const Mini *__restrict__ minis, int64_t n_minis, unsigned long long *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < n_minis; m += (int64_t)gridDim.x * 4) {
        const Mini mb = minis[m];
        const unsigned long long mask = mb.width >= 64 ? ~0ULL : ((1ULL << mb.width) - 1ULL);
        for (int i = lane; i < mb.count; i += 64) {
            const int64_t bit = (int64_t)i * mb.width;
            const uint8_t *p = bytes + mb.in_off + (bit >> 3);
            const int sh = (int)(bit & 7);
            unsigned long long lo = 0;
            memcpy(&lo, p, 8);                                   // (the page is uploaded with 16 bytes of padding)
            unsigned long long v = lo >> sh;
            if (sh && mb.width + sh > 64) v |= (unsigned long long)p[8] << (64 - sh);
            out[mb.out_off + i] = (v & mask) + (unsigned long long)mb.min_delta;
        }
    }
}

__global__ void __launch_bounds__(256) place_low_half_kernel(const long long *__restrict__ compact, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n,
                                                             int32_t *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) out[r] = (nulls && nulls[r]) ? 0 : (int32_t)compact[rank ? rank[r] : r];
}

static bool read_uleb(const uint8_t *bytes, int64_t len, int64_t &at, uint64_t &out)
{
    uint64_t v = 0;
    for (int shift = 0;; shift += 7) {
        if (at >= len || shift > 63) return false;
        const int b = bytes[at++];
        v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) break;
    }
    out = v;
    return true;
}

// `want` DELTA_BINARY_PACKED values on the device as wrapping 64-bit integers (want >= 1); *end = the bytes the section takes
BufferPtr delta_values(Context *ctx, const uint8_t *bytes, int64_t len, int64_t want, int64_t *end)
{
    int64_t at = 0;
    uint64_t block_size, miniblocks, total, zz;
    TG_CHECK_ARG(read_uleb(bytes, len, at, block_size) && read_uleb(bytes, len, at, miniblocks) && read_uleb(bytes, len, at, total) && read_uleb(bytes, len, at, zz),
                 "Parquet DELTA_BINARY_PACKED header cut short");
    TG_CHECK_ARG(miniblocks > 0 && miniblocks <= 4096 && block_size > 0 && block_size % 128 == 0 && block_size % miniblocks == 0 && (block_size / miniblocks) % 32 == 0,
                 "Parquet DELTA_BINARY_PACKED block shape outside the specification");
    TG_CHECK_ARG((int64_t)total >= want, "Parquet DELTA_BINARY_PACKED section holds fewer values than the page has non-null positions");
    const int64_t mini = (int64_t)(block_size / miniblocks);
    const uint64_t first = (zz >> 1) ^ (~(zz & 1) + 1);
    std::vector<Mini> minis;
    for (int64_t done = 1; done < want;) {
        uint64_t zmin;
        TG_CHECK_ARG(read_uleb(bytes, len, at, zmin) && at + (int64_t)miniblocks <= len, "Parquet DELTA_BINARY_PACKED block header cut short");
        const uint64_t min_delta = (zmin >> 1) ^ (~(zmin & 1) + 1);
        const uint8_t *widths = bytes + at;
        at += (int64_t)miniblocks;
        for (uint64_t m = 0; m < miniblocks && done < want; m++) {
            const int width = widths[m];
            TG_CHECK_ARG(width <= 64 && at + mini * width / 8 <= len, "Parquet DELTA_BINARY_PACKED miniblock cut short");
            Mini mb;
            mb.in_off = at;
            mb.out_off = done;
            mb.min_delta = (int64_t)min_delta;
            mb.count = (int32_t)std::min<int64_t>(mini, want - done);
            mb.width = width;
            minis.push_back(mb);
            done += mb.count;
            at += mini * width / 8;
        }
    }
    if (end) *end = at;
    // deltas (slot 0: the first value) -> inclusive prefix sum = the values
    BufferPtr seq = ctx->alloc((size_t)want * 8), sums = ctx->alloc((size_t)want * 8);
    ctx->upload(seq->ptr(), &first, 8);
    BufferPtr page, dminis;
    if (!minis.empty()) {
        page = upload_padded(ctx, bytes, at);
        dminis = ctx->alloc(minis.size() * sizeof(Mini));
        ctx->upload(dminis->ptr(), minis.data(), minis.size() * sizeof(Mini));
        ProfileScope ps(ctx, "parquet_delta_unpack");
        delta_unpack_kernel<<<(int)std::min<int64_t>(ceil_div((int64_t)minis.size(), 4), (int64_t)ctx->cu_count() * 8), 256, 0, ctx->stream()>>>(page->as<uint8_t>(), dminis->as<Mini>(),
                                                                                                                                           (int64_t)minis.size(), seq->as<unsigned long long>());
        check_launch("parquet_delta_unpack");
    }
    {
        ProfileScope ps(ctx, "parquet_delta_scan");
        size_t temp_bytes = 0;
        HIP_CHECK(rocprim::inclusive_scan(nullptr, temp_bytes, seq->as<unsigned long long>(), sums->as<unsigned long long>(), (size_t)want, rocprim::plus<unsigned long long>(), ctx->stream()));
        BufferPtr temp = ctx->alloc(temp_bytes ? temp_bytes : 1);
        HIP_CHECK(rocprim::inclusive_scan(temp->ptr(), temp_bytes, seq->as<unsigned long long>(), sums->as<unsigned long long>(), (size_t)want, rocprim::plus<unsigned long long>(), ctx->stream()));
    }
    ctx->sync();   // the host vectors back the uploads
    return sums;
}

DeviceColumn delta_column(Context *ctx, int32_t type, int32_t physical, const uint8_t *bytes, int64_t len, int64_t n, const Present &p)
{
    if (physical != PQ_INT32 && physical != PQ_INT64) fail(TGPU_ERR_NOT_SUPPORTED, "Parquet DELTA_BINARY_PACKED is for INT32 and INT64 columns");   // ParquetEncoding.java:151
    DeviceColumn col;
    col.type = type;
    col.n = n;
    const int w = width_of(physical);
    col.values_buf = ctx->alloc((size_t)(n > 0 ? n : 1) * (size_t)w);
    col.values = col.values_buf->ptr();
    if (n == 0) return col;
    const uint8_t *nulls = p.nulls && p.non_null < n ? p.nulls->as<uint8_t>() : nullptr;
    const int32_t *rank = nulls ? p.rank->as<int32_t>() : nullptr;
    if (nulls) {
        col.nulls_buf = p.nulls;
        col.nulls = nulls;
    }
    if (p.non_null == 0) {
        HIP_CHECK(hipMemsetAsync(col.values_buf->ptr(), 0, (size_t)n * (size_t)w, ctx->stream()));
        return col;
    }
    BufferPtr sums = delta_values(ctx, bytes, len, p.non_null, nullptr);
    if (w == 8) {
        if (!nulls) HIP_CHECK(hipMemcpyAsync(col.values_buf->ptr(), sums->ptr(), (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream()));
        else place_kernel<long long><<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(sums->as<long long>(), rank, nulls, n, (long long *)col.values_buf->ptr());
    }
    else place_low_half_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(sums->as<long long>(), rank, nulls, n, (int32_t *)col.values_buf->ptr());
    check_launch("parquet_place_delta_values");
    ctx->sync();   // (sums is released on return)
    return col;
}

// DELTA_LENGTH_BYTE_ARRAY (Encodings.md "Delta-length byte array"; ParquetEncoding.java:156-163 -> parquet-mr's DeltaLengthByteArrayValuesReader): the
// non-null values' lengths as a DELTA_BINARY_PACKED section, then their bytes back to back -- which IS the column's pool (a null row has no
// bytes): one upload, and the offsets are the prefix sum of the row lengths
__global__ void __launch_bounds__(256) place_lengths_kernel(const long long *__restrict__ compact, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n,
                                                            int64_t pool_bytes, int32_t *__restrict__ out, unsigned int *__restrict__ error)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        long long l = (nulls && nulls[r]) ? 0 : compact[rank ? rank[r] : r];
        if (l < 0 || l > pool_bytes) {
            *error = 1u;
            l = 0;
        }
        out[r] = (int32_t)l;
    }
}

DeviceColumn delta_length_strings(Context *ctx, int32_t type, int32_t physical, const uint8_t *bytes, int64_t len, int64_t n, const Present &p)
{
    if (physical != PQ_BYTE_ARRAY) fail(TGPU_ERR_NOT_SUPPORTED, "Parquet DELTA_LENGTH_BYTE_ARRAY is for BYTE_ARRAY columns");   // ParquetEncoding.java:160
    DeviceColumn col;
    col.type = type;
    col.n = n;
    const uint8_t *nulls = p.nulls && p.non_null < n ? p.nulls->as<uint8_t>() : nullptr;
    const int32_t *rank = nulls ? p.rank->as<int32_t>() : nullptr;
    if (nulls) {
        col.nulls_buf = p.nulls;
        col.nulls = nulls;
    }
    col.offsets_buf = ctx->alloc((size_t)(n + 1) * 4);
    col.offsets = col.offsets_buf->as<int32_t>();
    if (n == 0 || p.non_null == 0) {
        HIP_CHECK(hipMemsetAsync(col.offsets_buf->ptr(), 0, (size_t)(n + 1) * 4, ctx->stream()));
        col.values_buf = ctx->alloc(1);
        col.values = col.values_buf->ptr();
        col.pool_bytes = 0;
        col.pool_exact = true;
        return col;
    }
    int64_t at = 0;
    BufferPtr lengths = delta_values(ctx, bytes, len, p.non_null, &at);
    const int64_t pool_bytes = len - at;
    TG_CHECK_ARG(pool_bytes <= 0x7fffffffLL, "Parquet page holds more than 2 GB of string bytes");
    col.values_buf = ctx->alloc((size_t)(pool_bytes > 0 ? pool_bytes : 1));
    col.values = col.values_buf->ptr();
    if (pool_bytes > 0) ctx->upload(col.values_buf->ptr(), bytes + at, (size_t)pool_bytes);
    BufferPtr row_len = ctx->alloc((size_t)n * 4), total = ctx->alloc(8), error = ctx->alloc_zero(4);
    place_lengths_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(lengths->as<long long>(), rank, nulls, n, pool_bytes, row_len->as<int32_t>(), error->as<unsigned int>());
    check_launch("parquet_place_lengths");
    k::exclusive_scan_i32(ctx, row_len->as<int32_t>(), const_cast<int32_t *>(col.offsets), n, total->as<int64_t>());
    // the lengths must add up to no more than the bytes that follow them (the last offset = their sum)
    const int64_t sum = ctx->read_scalar(total->as<int64_t>());
    TG_CHECK_ARG(ctx->read_scalar(error->as<unsigned int>()) == 0 && sum <= pool_bytes, "Parquet DELTA_LENGTH_BYTE_ARRAY lengths do not fit the bytes that follow them");
    const int32_t end = (int32_t)sum;
    ctx->upload(const_cast<int32_t *>(col.offsets) + n, &end, 4);
    col.pool_bytes = sum;
    col.pool_exact = true;
    ctx->sync();
    return col;
}

// DELTA_BYTE_ARRAY (Encodings.md "Delta Strings"; ParquetEncoding.java:165-173 -> parquet-mr's DeltaByteArrayReader): prefix lengths as a
// DELTA_BINARY_PACKED section, then the suffixes as a DELTA_LENGTH_BYTE_ARRAY section; value[i] = value[i - 1][0 .. prefix[i]) ++ suffix[i].
// Byte j of value i is byte j - prefix[k] of suffix k for the LAST k <= i with prefix[k] <= j (every value in between took that byte over from
// its predecessor): one workgroup per byte position j runs a max-scan of "k if prefix[k] <= j" along the values and copies the byte --
// no value waits for the one before it.
__global__ void __launch_bounds__(256) dba_prepare_kernel(const long long *__restrict__ prefix, const long long *__restrict__ slen, int64_t count, int32_t *__restrict__ len32,
                                                          int32_t *__restrict__ slen32, unsigned int *__restrict__ status /* [0] error, [1] longest value */)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const long long p = prefix[i], sl = slen[i];
        const long long before = i ? prefix[i - 1] + slen[i - 1] : 0;
        const bool bad = p < 0 || sl < 0 || p > before || p + sl > 0x7fffffffLL;
        if (bad) status[0] = 1u;
        len32[i] = bad ? 0 : (int32_t)(p + sl);
        slen32[i] = bad ? 0 : (int32_t)sl;
        if (!bad) atomicMax(&status[1], (unsigned int)(p + sl));
    }
}

__global__ void __launch_bounds__(256) dba_resolve_kernel(const long long *__restrict__ prefix, const int32_t *__restrict__ len32, const int32_t *__restrict__ soff,
                                                          const int32_t *__restrict__ voff, const uint8_t *__restrict__ suffixes, int64_t count, uint8_t *__restrict__ pool)
{
    __shared__ int scan[256];
    const int j = blockIdx.x;          // the byte position this workgroup resolves
    int carry = -1;                    // last k before this chunk with prefix[k] <= j
    for (int64_t base = 0; base < count; base += 256) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < count;
        scan[threadIdx.x] = (live && prefix[i] <= j) ? (int)i : -1;
        __syncthreads();
#pragma unroll
        for (int d = 1; d < 256; d <<= 1) {
            const int other = threadIdx.x >= d ? scan[threadIdx.x - d] : -1;
            __syncthreads();
            scan[threadIdx.x] = other > scan[threadIdx.x] ? other : scan[threadIdx.x];
            __syncthreads();
        }
        const int k = scan[threadIdx.x] > carry ? scan[threadIdx.x] : carry;
        if (live && len32[i] > j && k >= 0) pool[(size_t)voff[i] + j] = suffixes[(size_t)soff[k] + (j - (int)prefix[k])];
        const int last = scan[255];
        __syncthreads();
        carry = last > carry ? last : carry;
    }
}

DeviceColumn delta_strings(Context *ctx, int32_t type, int32_t physical, const uint8_t *bytes, int64_t len, int64_t n, const Present &p)
{
    if (physical != PQ_BYTE_ARRAY) fail(TGPU_ERR_NOT_SUPPORTED, "Parquet DELTA_BYTE_ARRAY is decoded for BYTE_ARRAY columns");   // (ParquetEncoding.java:170: and FIXED_LEN_BYTE_ARRAY, not a type here)
    DeviceColumn col;
    col.type = type;
    col.n = n;
    const uint8_t *nulls = p.nulls && p.non_null < n ? p.nulls->as<uint8_t>() : nullptr;
    const int32_t *rank = nulls ? p.rank->as<int32_t>() : nullptr;
    if (nulls) {
        col.nulls_buf = p.nulls;
        col.nulls = nulls;
    }
    col.offsets_buf = ctx->alloc((size_t)(n + 1) * 4);
    col.offsets = col.offsets_buf->as<int32_t>();
    col.pool_exact = true;
    const int64_t want = p.non_null;
    if (n == 0 || want == 0) {
        HIP_CHECK(hipMemsetAsync(col.offsets_buf->ptr(), 0, (size_t)(n + 1) * 4, ctx->stream()));
        col.values_buf = ctx->alloc(1);
        col.values = col.values_buf->ptr();
        col.pool_bytes = 0;
        return col;
    }
    int64_t a = 0, b = 0;
    BufferPtr prefix = delta_values(ctx, bytes, len, want, &a);
    BufferPtr slen = delta_values(ctx, bytes + a, len - a, want, &b);
    const int64_t sfx_bytes = len - a - b;
    BufferPtr len32 = ctx->alloc((size_t)want * 4), slen32 = ctx->alloc((size_t)want * 4), soff = ctx->alloc((size_t)want * 4), voff = ctx->alloc((size_t)want * 4),
              status = ctx->alloc_zero(8), total_s = ctx->alloc(8), total_v = ctx->alloc(8);
    dba_prepare_kernel<<<grid_for(ctx, want), 256, 0, ctx->stream()>>>(prefix->as<long long>(), slen->as<long long>(), want, len32->as<int32_t>(), slen32->as<int32_t>(),
                                                                        status->as<unsigned int>());
    check_launch("parquet_delta_strings_prepare");
    k::exclusive_scan_i32(ctx, slen32->as<int32_t>(), soff->as<int32_t>(), want, total_s->as<int64_t>());
    k::exclusive_scan_i32(ctx, len32->as<int32_t>(), voff->as<int32_t>(), want, total_v->as<int64_t>());
    unsigned int st[2];
    ctx->download(st, status->ptr(), 8);
    const int64_t suffix_total = ctx->read_scalar(total_s->as<int64_t>()), pool_bytes = ctx->read_scalar(total_v->as<int64_t>());
    TG_CHECK_ARG(st[0] == 0 && suffix_total <= sfx_bytes, "Parquet DELTA_BYTE_ARRAY prefix / suffix lengths do not fit the values before them or the bytes that follow");
    TG_CHECK_ARG(pool_bytes <= 0x7fffffffLL, "Parquet page holds more than 2 GB of string bytes");
    col.values_buf = ctx->alloc((size_t)(pool_bytes > 0 ? pool_bytes : 1));
    col.values = col.values_buf->ptr();
    col.pool_bytes = pool_bytes;
    const int64_t longest = st[1];
    if (longest > 0) {
        BufferPtr sfx = upload_padded(ctx, bytes + a + b, sfx_bytes);
        ProfileScope ps(ctx, "parquet_delta_strings_resolve");
        dba_resolve_kernel<<<(int)longest, 256, 0, ctx->stream()>>>(prefix->as<long long>(), len32->as<int32_t>(), soff->as<int32_t>(), voff->as<int32_t>(), sfx->as<uint8_t>(), want,
                                                                    col.values_buf->as<uint8_t>());
        check_launch("parquet_delta_strings_resolve");
    }
    // row offsets: a null row has no bytes, so the values' pool is the column's pool
    BufferPtr row_len = ctx->alloc((size_t)n * 4), total = ctx->alloc(8);
    place_kernel<int32_t><<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(len32->as<int32_t>(), rank, nulls, n, row_len->as<int32_t>());
    check_launch("parquet_place_lengths");
    k::exclusive_scan_i32(ctx, row_len->as<int32_t>(), const_cast<int32_t *>(col.offsets), n, total->as<int64_t>());
    const int32_t end = (int32_t)pool_bytes;
    ctx->upload(const_cast<int32_t *>(col.offsets) + n, &end, 4);
    ctx->sync();
    return col;
}

void check_types(int32_t type, int32_t physical)
{
    const bool ok = (physical == PQ_INT32 && (type == TGPU_INTEGER || type == TGPU_DATE)) || (physical == PQ_INT64 && type == TGPU_BIGINT) || (physical == PQ_DOUBLE && type == TGPU_DOUBLE) ||
                    (physical == PQ_BOOLEAN && type == TGPU_BOOLEAN) || (physical == PQ_BYTE_ARRAY && type == TGPU_VARCHAR);
    if (!ok) fail(TGPU_ERR_NOT_SUPPORTED, "Parquet physical type / column type pair not decoded on the device (INT32 -> INTEGER / DATE, INT64 -> BIGINT, DOUBLE, BOOLEAN, BYTE_ARRAY -> VARCHAR)");
}

}  // namespace

DeviceColumn decode_data_page(Context *ctx, int32_t type, int32_t physical, int32_t encoding, int64_t n, const uint8_t *def_levels, int64_t def_len, const uint8_t *values,
                              int64_t values_len, const uint8_t *dictionary, int64_t dictionary_len, int32_t dictionary_count)
{
    TG_CHECK_ARG(n >= 0 && n <= 0x7fffffffLL && def_len >= 0 && values_len >= 0 && dictionary_len >= 0 && dictionary_count >= 0, "bad argument");
    check_types(type, physical);
    Present p = decode_levels(ctx, def_levels, def_len, n);
    if (encoding == PQ_PLAIN) return plain_column(ctx, type, physical, values, values_len, n, p);
    if (encoding == PQ_DELTA_BINARY_PACKED) return delta_column(ctx, type, physical, values, values_len, n, p);
    if (encoding == PQ_DELTA_LENGTH_BYTE_ARRAY) return delta_length_strings(ctx, type, physical, values, values_len, n, p);
    if (encoding == PQ_DELTA_BYTE_ARRAY) return delta_strings(ctx, type, physical, values, values_len, n, p);
    if (encoding == PQ_RLE) {
        // ParquetEncoding.RLE as a VALUE encoding exists for BOOLEAN only (ParquetEncoding.java:105-115,198-212: bit width 1): a 4-byte length, then
        // the non-null rows' booleans as a hybrid stream
        if (physical != PQ_BOOLEAN) fail(TGPU_ERR_NOT_SUPPORTED, "Parquet RLE value encoding is for BOOLEAN columns");
        DeviceColumn col;
        col.type = type;
        col.n = n;
        col.values_buf = ctx->alloc((size_t)(n > 0 ? n : 1));
        col.values = col.values_buf->ptr();
        if (n == 0) return col;
        const uint8_t *nulls = p.nulls && p.non_null < n ? p.nulls->as<uint8_t>() : nullptr;
        if (nulls) {
            col.nulls_buf = p.nulls;
            col.nulls = nulls;
        }
        BufferPtr compact;
        if (p.non_null > 0) {
            TG_CHECK_ARG(values_len >= 4, "Parquet RLE value section without its length");
            uint32_t length;
            memcpy(&length, values, 4);
            TG_CHECK_ARG((int64_t)length <= values_len - 4, "Parquet RLE value section shorter than its length says");
            compact = decode_hybrid(ctx, values + 4, (int64_t)length, 1, p.non_null);
        } else compact = ctx->alloc(4);
        place_narrow_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(compact->as<int32_t>(), nulls ? p.rank->as<int32_t>() : nullptr, nulls, n, col.values_buf->as<uint8_t>());
        check_launch("parquet_place_booleans");
        return col;
    }
    if (encoding != PQ_PLAIN_DICTIONARY && encoding != PQ_RLE_DICTIONARY) fail(TGPU_ERR_NOT_SUPPORTED, "Parquet value encoding not decoded on the device (PLAIN, PLAIN_DICTIONARY, RLE_DICTIONARY, DELTA_BINARY_PACKED, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY; RLE for BOOLEAN)");
    TG_CHECK_ARG(physical != PQ_BOOLEAN, "BOOLEAN columns have no dictionary encoding");
    // the dictionary page: PLAIN values without nulls (dictionary/*.java); the page: one byte of bit width, then the ids as a hybrid stream
    Present all;
    all.non_null = dictionary_count;
    DeviceColumn dict = plain_column(ctx, type, physical, dictionary, dictionary_len, dictionary_count, all);
    if (n == 0) return k::region_of(ctx, dict, 0, 0);
    BufferPtr ids = ctx->alloc((size_t)n * 4), error = ctx->alloc_zero(4);
    if (p.non_null > 0) {
        TG_CHECK_ARG(values_len >= 1, "Parquet dictionary-encoded page without its bit width byte");
        const int bit_width = values[0];
        BufferPtr compact = decode_hybrid(ctx, values + 1, values_len - 1, bit_width, p.non_null);
        const uint8_t *nulls = p.nulls && p.non_null < n ? p.nulls->as<uint8_t>() : nullptr;
        place_ids_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(compact->as<int32_t>(), nulls ? p.rank->as<int32_t>() : nullptr, nulls, n, dictionary_count, ids->as<int32_t>(),
                                                                      error->as<unsigned int>());
        check_launch("parquet_place_ids");
        if (ctx->read_scalar(error->as<unsigned int>()) != 0) fail(TGPU_ERR_INVALID_ARGUMENT, "Parquet dictionary id outside the dictionary");
    } else HIP_CHECK(hipMemsetAsync(ids->ptr(), 0xff, (size_t)n * 4, ctx->stream()));
    ProfileScope ps(ctx, "parquet_dictionary_gather");
    return k::gather_column(ctx, dict, ids->as<int32_t>(), n, /*negative_is_null=*/p.non_null < n);
}

}  // namespace parquet
}  // namespace tgpu
