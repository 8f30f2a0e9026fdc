// orc.hip -- scan-side decode, first slice: the integer / boolean / dictionary-string streams of an ORC stripe decoded on the
// device into flat HBM columns.  Reference (lib/trino-orc/src/main/java/io/trino/orc/): stream/LongInputStreamV2.java:59-312 (RLEv2:
// SHORT_REPEAT / DIRECT / PATCHED_BASE / DELTA), stream/LongBitPacker.java:82-108, stream/LongDecode.java:47-158, stream/ByteInputStream.java:43-75,
// stream/BooleanInputStream.java:36-58, reader/LongColumnReader.java:100-230 (PRESENT + DATA -> a long / int block),
// reader/BooleanColumnReader.java, reader/SliceDictionaryColumnReader.java:120-330 (DATA ids + LENGTH + DICTIONARY_DATA).
//
// The streams arrive DECOMPRESSED in host memory (the ORC chunk framing and its codecs stay with the file reader).  A stream is a sequence
// of runs whose headers are variable length, so where a run starts is only known from the run before it: the host walks the HEADERS (a few
// bytes per run: control bytes, run lengths, the two vints of a DELTA run) while it stages the bytes, and uploads a run directory; every value
// is decoded on the device, one wave per run -- bit unpacking at lane-computed bit offsets, zigzag, the prefix sums of DELTA runs as a wave scan,
// the patch list of PATCHED_BASE runs as a scan of its gaps.  PRESENT bits become the null vector, the values are expanded to row positions
// through an exclusive scan of the not-null flags.  RLEv1 streams (files written before Hive 0.12): one lane per run.
#include "orc.h"

#include "kernels.h"

#include <cstring>

namespace tgpu {
namespace orc {

namespace {

constexpr int kWave = 64;

enum RunKind : int32_t { SHORT_REPEAT = 0, DIRECT = 1, PATCHED_BASE = 2, DELTA = 3, V1_RUN = 4, V1_LITERALS = 5 };

struct Run {
    int64_t in_off;       // first byte of the run's PACKED payload (behind the header fields the host has read)
    int64_t out_off;      // index of the run's first value
    int32_t kind, count;  // values of the run
    int32_t width;        // bit width of the packed values (0: DELTA with a fixed delta)
    int32_t patch_width, patch_gap_width, patch_count, patch_bits;   // PATCHED_BASE: patch list entries of patch_bits bits = gap | patch
    int64_t base;         // SHORT_REPEAT: the (zigzag-decoded) value; PATCHED_BASE: base; DELTA: first value
    int64_t delta;        // DELTA: fixed delta / delta base
    int64_t patch_off;    // PATCHED_BASE: first byte of the patch list
};

int decode_bit_width(int n)
{
    if (n >= 0 && n <= 23) return n + 1;
    static const int wide[8] = {26, 28, 30, 32, 40, 48, 56, 64};
    return wide[(n - 24) & 7];
}
int closest_fixed_bits(int w)
{
    if (w == 0) return 1;
    if (w <= 24) return w;
    if (w <= 26) return 26;
    if (w <= 28) return 28;
    if (w <= 30) return 30;
    if (w <= 32) return 32;
    if (w <= 40) return 40;
    if (w <= 48) return 48;
    if (w <= 56) return 56;
    return 64;
}

struct HostReader {
    const uint8_t *p;
    int64_t len, at = 0;
    int read()
    {
        if (at >= len) fail(TGPU_ERR_INVALID_ARGUMENT, "ORC stream: read past the end of an RLE run");
        return p[at++];
    }
    uint64_t vint()
    {
        uint64_t r = 0;
        int off = 0, b;
        do {
            b = read();
            if (off < 64) r |= (uint64_t)(b & 0x7f) << off;
            off += 7;
        } while (b & 0x80);
        return r;
    }
    void skip(int64_t n)
    {
        if (at + n > len) fail(TGPU_ERR_INVALID_ARGUMENT, "ORC stream: a run is longer than the stream");
        at += n;
    }
};
int64_t zigzag(uint64_t v) { return (int64_t)((v >> 1) ^ (uint64_t)(-(int64_t)(v & 1))); }

// the run directory of an RLEv2 stream (LongInputStreamV2.readValues :59-80 and the four readers' header parsing)
std::vector<Run> scan_rle_v2(const uint8_t *bytes, int64_t len, bool is_signed, int64_t &total)
{
    std::vector<Run> runs;
    HostReader in{bytes, len};
    total = 0;
    while (in.at < in.len) {
        Run r{};
        const int first = in.read();
        r.kind = (first >> 6) & 3;
        r.out_off = total;
        if (r.kind == SHORT_REPEAT) {   // :255-282
            const int size = ((first >> 3) & 7) + 1;
            r.count = (first & 7) + 3;
            uint64_t v = 0;
            for (int n = size; n > 0;) {
                n--;
                v |= (uint64_t)in.read() << (n * 8);
            }
            r.base = is_signed ? zigzag(v) : (int64_t)v;
            r.in_off = in.at;
        }
        else if (r.kind == DIRECT) {   // :229-252
            r.width = decode_bit_width((first >> 1) & 0x1f);
            r.count = (((first & 1) << 8) | in.read()) + 1;
            r.in_off = in.at;
            in.skip(((int64_t)r.count * r.width + 7) / 8);
        }
        else if (r.kind == PATCHED_BASE) {   // :135-226
            r.width = decode_bit_width((first >> 1) & 0x1f);
            r.count = (((first & 1) << 8) | in.read()) + 1;
            const int third = in.read();
            const int base_width = ((third >> 5) & 7) + 1;
            r.patch_width = decode_bit_width(third & 0x1f);
            const int fourth = in.read();
            r.patch_gap_width = ((fourth >> 5) & 7) + 1;
            r.patch_count = fourth & 0x1f;
            uint64_t b = 0;
            for (int n = base_width; n > 0;) {
                n--;
                b |= (uint64_t)in.read() << (n * 8);
            }
            const uint64_t mask = 1ull << (base_width * 8 - 1);
            r.base = (b & mask) ? -(int64_t)(b & ~mask) : (int64_t)b;
            if (r.patch_width + r.patch_gap_width > 64) fail(TGPU_ERR_INVALID_ARGUMENT, "Invalid RLEv2 encoded stream");
            r.patch_bits = closest_fixed_bits(r.patch_width + r.patch_gap_width);
            r.in_off = in.at;
            in.skip(((int64_t)r.count * r.width + 7) / 8);
            r.patch_off = in.at;
            in.skip(((int64_t)r.patch_count * r.patch_bits + 7) / 8);
        }
        else {   // DELTA :82-132
            int fixed_bits = (first >> 1) & 0x1f;
            if (fixed_bits != 0) fixed_bits = decode_bit_width(fixed_bits);
            const int length = ((first & 1) << 8) | in.read();
            const uint64_t fv = in.vint();
            r.base = is_signed ? zigzag(fv) : (int64_t)fv;
            r.delta = zigzag(in.vint());
            r.width = fixed_bits;
            r.count = length + 1;
            r.in_off = in.at;
            if (fixed_bits != 0) {
                if (length < 1) fail(TGPU_ERR_INVALID_ARGUMENT, "Invalid RLEv2 encoded stream");
                in.skip(((int64_t)(length - 1) * fixed_bits + 7) / 8);
            }
        }
        total += r.count;
        runs.push_back(r);
    }
    return runs;
}

// `w` bits at bit offset `bit` of a big-endian bit stream (LongBitPacker.unpackGeneric's order); the buffer has 16 bytes of slack behind it
__device__ __forceinline__ unsigned long long read_bits(const uint8_t *p, long long bit, int w)
{
    const uint8_t *q = p + (bit >> 3);
    const int sh = (int)(bit & 7);
    unsigned long long x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x = (x << 8) | q[i];
    if (sh + w <= 64) return w == 64 ? x : ((x << sh) >> (64 - w));
    return ((x << sh) | ((unsigned long long)q[8] >> (8 - sh))) >> (64 - w);
}
__device__ __forceinline__ long long dev_zigzag(unsigned long long v) { return (long long)((v >> 1) ^ (unsigned long long)(-(long long)(v & 1))); }

// one wave per run (grid-stride over the runs)
__global__ void __launch_bounds__(256) rle_v2_decode_kernel(const uint8_t *__restrict__ bytes, const Run *__restrict__ runs, int64_t n_runs, int is_signed,
                                                            long long *__restrict__ out)
{
    __shared__ long long s_stage[4][512];
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t ri = wave; ri < n_runs; ri += waves) {
        const Run r = runs[ri];
        long long *o = out + r.out_off;
        const uint8_t *p = bytes + r.in_off;
        if (r.kind == SHORT_REPEAT) {
            for (int i = lane; i < r.count; i += kWave) o[i] = r.base;
        }
        else if (r.kind == DIRECT) {
            for (int i = lane; i < r.count; i += kWave) {
                const unsigned long long v = read_bits(p, (long long)i * r.width, r.width);
                o[i] = is_signed ? dev_zigzag(v) : (long long)v;
            }
        }
        else if (r.kind == PATCHED_BASE) {
            // staged in LDS (a run has at most 512 values): the patches below land on values other lanes unpacked
            long long *stage = s_stage[threadIdx.x >> 6];
            for (int i = lane; i < r.count; i += kWave) stage[i] = (long long)((unsigned long long)r.base + read_bits(p, (long long)i * r.width, r.width));
            // the patch list: entry j sits at the running sum of the gaps up to j; an entry (gap 255, patch 0) only extends the gap
            unsigned long long gap = 0, patch = 0;
            if (lane < r.patch_count) {
                const unsigned long long e = read_bits(bytes + r.patch_off, (long long)lane * r.patch_bits, r.patch_bits);
                const unsigned long long pmask = r.patch_width >= 64 ? ~0ULL : ((1ULL << r.patch_width) - 1ULL);
                gap = r.patch_width >= 64 ? 0ULL : (e >> r.patch_width);
                patch = e & pmask;
            }
            unsigned long long pos = gap;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const unsigned long long up = __shfl_up(pos, d, 64);
                if (lane >= d) pos += up;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (lane < r.patch_count && !(gap == 255 && patch == 0) && pos < (unsigned long long)r.count)
                stage[pos] = (long long)((unsigned long long)stage[pos] + (patch << r.width));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < r.count; i += kWave) o[i] = stage[i];
            __builtin_amdgcn_wave_barrier();
        }
        else {   // DELTA
            if (lane == 0) o[0] = r.base;
            if (r.width == 0) {
                for (int i = lane; i + 1 < r.count; i += kWave) o[i + 1] = (long long)((unsigned long long)r.base + (unsigned long long)(i + 1) * (unsigned long long)r.delta);
            }
            else {
                // value[1] = first + deltaBase; value[k] = value[k - 1] +- packed[k - 2]: eight consecutive deltas per lane, then a wave scan
                const int nd = r.count - 2;                       // packed deltas
                unsigned long long local[8], sum = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i = lane * 8 + j;
                    local[j] = i < nd ? read_bits(p, (long long)i * r.width, r.width) : 0ULL;
                    sum += local[j];
                }
                unsigned long long incl = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned long long up = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += up;
                }
                unsigned long long run = incl - sum;              // exclusive prefix of this lane's block
                const unsigned long long second = (unsigned long long)r.base + (unsigned long long)r.delta;
                if (lane == 0 && r.count > 1) o[1] = (long long)second;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i = lane * 8 + j;
                    run += local[j];
                    if (i < nd) o[i + 2] = (long long)(r.delta < 0 ? second - run : second + run);
                }
            }
        }
    }
}

struct ByteRun {
    int64_t in_off, out_off;
    int32_t count, repeat;   // repeat: one value byte at in_off; else `count` literal bytes
};

std::vector<ByteRun> scan_byte_rle(const uint8_t *bytes, int64_t len, int64_t &total)   // ByteInputStream.readNextBlock :43-75
{
    std::vector<ByteRun> runs;
    HostReader in{bytes, len};
    total = 0;
    while (in.at < in.len) {
        const int control = in.read();
        ByteRun r{};
        r.out_off = total;
        if ((control & 0x80) == 0) {
            r.count = control + 3;
            r.repeat = 1;
            r.in_off = in.at;
            in.skip(1);
        }
        else {
            r.count = 0x100 - control;
            r.in_off = in.at;
            in.skip(r.count);
        }
        total += r.count;
        runs.push_back(r);
    }
    return runs;
}

__global__ void __launch_bounds__(256) byte_rle_decode_kernel(const uint8_t *__restrict__ bytes, const ByteRun *__restrict__ runs, int64_t n_runs, uint8_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t ri = wave; ri < n_runs; ri += waves) {
        const ByteRun r = runs[ri];
        for (int i = lane; i < r.count; i += kWave) out[r.out_off + i] = bytes[r.in_off + (r.repeat ? 0 : i)];
    }
}

// BooleanInputStream: bit i of the byte-RLE payload, most significant bit first.  PRESENT streams: flag = 1 means the row HAS a value.
__global__ void __launch_bounds__(256) bits_to_flags_kernel(const uint8_t *__restrict__ packed, int64_t n, int invert, uint8_t *__restrict__ flags, int32_t *__restrict__ not_null)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int bit = (packed[i >> 3] >> (7 - (i & 7))) & 1;
        flags[i] = (uint8_t)(invert ? !bit : bit);
        if (not_null) not_null[i] = bit;
    }
}

// values at their row positions: row i takes compact[rank[i]] unless it is null; INTEGER / DATE columns check the 32-bit range
// (LongInputStreamV2.next(int[]) :356-364: "Decoded value out of range for a 32bit number")
template <typename T>
__global__ void __launch_bounds__(256) place_values_kernel(const long long *__restrict__ compact, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n,
                                                           T *__restrict__ out, unsigned int *error)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (nulls && nulls[i]) {
            out[i] = 0;
            continue;
        }
        const long long v = compact[rank ? rank[i] : i];
        if (sizeof(T) == 4 && v != (long long)(int)v) atomicOr(error, 1u);
        out[i] = (T)v;
    }
}

// dictionary ids at their row positions, -1 for null rows (the gather below turns them into null cells); ids beyond the dictionary are an error
__global__ void __launch_bounds__(256) place_ids_kernel(const long long *__restrict__ compact, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n,
                                                        int32_t dictionary_size, int32_t *__restrict__ out, unsigned int *error)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (nulls && nulls[i]) {
            out[i] = -1;
            continue;
        }
        const long long v = compact[rank ? rank[i] : i];
        if (v < 0 || v >= dictionary_size) {
            atomicOr(error, 2u);
            out[i] = -1;
        }
        else out[i] = (int32_t)v;
    }
}

__global__ void __launch_bounds__(256) widen_u8_kernel(const uint8_t *__restrict__ in, int64_t n, long long *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i];
}

__global__ void __launch_bounds__(256) lengths_to_i32_kernel(const long long *__restrict__ lengths, int64_t n, int32_t *__restrict__ out, unsigned int *error)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long v = lengths[i];
        if (v < 0 || v > 0x7fffffffLL) atomicOr(error, 4u);
        out[i] = (int32_t)(v < 0 ? 0 : v);
    }
}

// lengths of the non-null rows -> a length per row (0 for a null row)
__global__ void __launch_bounds__(256) place_lengths_kernel(const int32_t *__restrict__ compact, const int32_t *__restrict__ rank, const uint8_t *__restrict__ nulls, int64_t n,
                                                            int32_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (nulls && nulls[i]) ? 0 : compact[rank ? rank[i] : i];
}

int grid_for(Context *ctx, int64_t n)
{
    int64_t blocks = ceil_div(n, 256);
    const int64_t cap = (int64_t)ctx->cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

BufferPtr upload_padded(Context *ctx, const uint8_t *src, int64_t bytes)
{
    BufferPtr b = ctx->alloc((size_t)bytes + 32);   // read_bits looks up to 9 bytes past a value's first byte
    if (bytes) ctx->upload(b->ptr(), src, (size_t)bytes);
    HIP_CHECK(hipMemsetAsync(static_cast<uint8_t *>(b->ptr()) + bytes, 0, 32, ctx->stream()));
    return b;
}

// every value of an RLEv2 stream as int64 on the device
// RLEv1 (LongInputStreamV1.java:47-103; files written before Hive 0.12): a control byte < 0x80 = a run of control + 3 values base + i * delta
// (delta a signed byte, base a varint); else 0x100 - control literal varints.  The host reads the control bytes and walks the varints'
// continuation bits (it must, to find the next control byte); the device decodes: one lane per run -- a run is at most 130 values.
std::vector<Run> scan_rle_v1(const uint8_t *bytes, int64_t len, bool is_signed, int64_t &total)
{
    std::vector<Run> runs;
    HostReader in{bytes, len};
    total = 0;
    while (in.at < in.len) {
        Run r{};
        const int control = in.read();
        r.out_off = total;
        if (control < 0x80) {
            r.kind = V1_RUN;
            r.count = control + 3;
            r.delta = (int8_t)in.read();
            const uint64_t v = in.vint();
            r.base = is_signed ? zigzag(v) : (int64_t)v;
            r.in_off = in.at;
        } else {
            r.kind = V1_LITERALS;
            r.count = 0x100 - control;
            r.in_off = in.at;
            for (int i = 0; i < r.count; i++) (void)in.vint();
        }
        total += r.count;
        if (total > 0x7fffffffLL) fail(TGPU_ERR_INVALID_ARGUMENT, "ORC stream holds more than 2^31 values");
        runs.push_back(r);
    }
    return runs;
}

__global__ void __launch_bounds__(256) rle_v1_decode_kernel(const uint8_t *__restrict__ bytes, const Run *__restrict__ runs, int64_t n_runs, int is_signed, long long *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += (int64_t)gridDim.x * blockDim.x) {
        const Run r = runs[i];
        long long *o = out + r.out_off;
        if (r.kind == V1_RUN) {
            for (int k = 0; k < r.count; k++) o[k] = (long long)((unsigned long long)r.base + (unsigned long long)((long long)k * r.delta));
            continue;
        }
        const uint8_t *p = bytes + r.in_off;
        for (int k = 0; k < r.count; k++) {   // LongDecode.readVInt :117-141
            unsigned long long v = 0;
            int off = 0, b;
            do {
                b = *p++;
                if (off < 64) v |= (unsigned long long)(b & 0x7f) << off;
                off += 7;
            } while (b & 0x80);
            o[k] = is_signed ? (long long)((v >> 1) ^ (unsigned long long)(-(long long)(v & 1))) : (long long)v;
        }
    }
}

BufferPtr decode_rle_v1(Context *ctx, const uint8_t *bytes, int64_t len, bool is_signed, int64_t &count)
{
    std::vector<Run> runs = scan_rle_v1(bytes, len, is_signed, count);
    BufferPtr out = ctx->alloc((size_t)(count > 0 ? count : 1) * 8);
    if (runs.empty()) return out;
    BufferPtr dbytes = upload_padded(ctx, bytes, len), druns = ctx->alloc(runs.size() * sizeof(Run));
    ctx->upload(druns->ptr(), runs.data(), runs.size() * sizeof(Run));
    ProfileScope ps(ctx, "orc_rle_v1_decode");
    rle_v1_decode_kernel<<<grid_for(ctx, (int64_t)runs.size()), 256, 0, ctx->stream()>>>(dbytes->as<uint8_t>(), druns->as<Run>(), (int64_t)runs.size(), is_signed ? 1 : 0,
                                                                                       out->as<long long>());
    check_launch("orc_rle_v1_decode");
    ctx->sync();   // `runs` (host) backs the upload
    return out;
}

BufferPtr decode_rle_v2(Context *ctx, const uint8_t *bytes, int64_t len, bool is_signed, int64_t &count)
{
    std::vector<Run> runs = scan_rle_v2(bytes, len, is_signed, count);
    BufferPtr out = ctx->alloc((size_t)(count > 0 ? count : 1) * 8);
    if (runs.empty()) return out;
    BufferPtr dbytes = upload_padded(ctx, bytes, len), druns = ctx->alloc(runs.size() * sizeof(Run));
    ctx->upload(druns->ptr(), runs.data(), runs.size() * sizeof(Run));
    ProfileScope ps(ctx, "orc_rle_v2_decode");
    rle_v2_decode_kernel<<<grid_for(ctx, (int64_t)runs.size() * kWave), 256, 0, ctx->stream()>>>(dbytes->as<uint8_t>(), druns->as<Run>(), (int64_t)runs.size(), is_signed ? 1 : 0,
                                                                                              out->as<long long>());
    check_launch("orc_rle_v2_decode");
    ctx->sync();   // `runs` (host) backs the upload
    return out;
}

// the first `count` bits of a boolean stream as one byte per position; PRESENT: nulls = !bit, and the not-null flags for the scan
void decode_boolean(Context *ctx, const uint8_t *bytes, int64_t len, int64_t count, bool as_nulls, uint8_t *flags_out, int32_t *not_null_out)
{
    int64_t total = 0;
    std::vector<ByteRun> runs = scan_byte_rle(bytes, len, total);
    TG_CHECK_ARG(total * 8 >= count, "ORC boolean stream is shorter than the column's positions");
    BufferPtr dbytes = upload_padded(ctx, bytes, len), druns = ctx->alloc((runs.size() + 1) * sizeof(ByteRun)), packed = ctx->alloc((size_t)(total > 0 ? total : 1));
    if (!runs.empty()) ctx->upload(druns->ptr(), runs.data(), runs.size() * sizeof(ByteRun));
    ProfileScope ps(ctx, "orc_boolean_decode");
    if (!runs.empty()) {
        byte_rle_decode_kernel<<<grid_for(ctx, (int64_t)runs.size() * kWave), 256, 0, ctx->stream()>>>(dbytes->as<uint8_t>(), druns->as<ByteRun>(), (int64_t)runs.size(),
                                                                                                    packed->as<uint8_t>());
        check_launch("orc_byte_rle_decode");
    }
    if (count > 0) {
        bits_to_flags_kernel<<<grid_for(ctx, count), 256, 0, ctx->stream()>>>(packed->as<uint8_t>(), count, as_nulls ? 1 : 0, flags_out, not_null_out);
        check_launch("orc_bits_to_flags");
    }
    ctx->sync();
}

struct Present {
    BufferPtr nulls, rank;
    int64_t non_null = 0;
};

// PRESENT stream -> null vector + the rank of every row among the non-null rows (null when the column has no PRESENT stream)
Present decode_present(Context *ctx, const uint8_t *present, int64_t present_len, int64_t n)
{
    Present p;
    p.non_null = n;
    if (!present || present_len == 0 || n == 0) return p;
    p.nulls = ctx->alloc((size_t)n);
    BufferPtr flags = ctx->alloc((size_t)n * 4), total = ctx->alloc(8);
    p.rank = ctx->alloc((size_t)n * 4);
    decode_boolean(ctx, present, present_len, n, true, p.nulls->as<uint8_t>(), flags->as<int32_t>());
    k::exclusive_scan_i32(ctx, flags->as<int32_t>(), p.rank->as<int32_t>(), n, total->as<int64_t>());
    p.non_null = ctx->read_scalar(total->as<int64_t>());
    return p;
}

void raise_if(Context *ctx, BufferPtr &error)
{
    const unsigned int e = ctx->read_scalar(error->as<unsigned int>());
    if (e & 1u) fail(TGPU_ERR_INVALID_ARGUMENT, "Decoded value out of range for a 32bit number");
    if (e & 2u) fail(TGPU_ERR_INVALID_ARGUMENT, "ORC dictionary id outside the dictionary");
    if (e & 4u) fail(TGPU_ERR_INVALID_ARGUMENT, "ORC dictionary entry length out of range");
}

// the column's integer streams: RLEv1 for the DIRECT / DICTIONARY encodings (files written before Hive 0.12), RLEv2 for the _V2 ones
bool integer_streams_are_v1(int32_t encoding, bool dictionary)
{
    if (encoding == (dictionary ? TGPU_ORC_DICTIONARY : TGPU_ORC_DIRECT)) return true;
    TG_CHECK_ARG(encoding == (dictionary ? TGPU_ORC_DICTIONARY_V2 : TGPU_ORC_DIRECT_V2), "unexpected ORC column encoding for this reader");
    return false;
}
BufferPtr decode_rle(Context *ctx, bool v1, const uint8_t *bytes, int64_t len, bool is_signed, int64_t &count)
{
    return v1 ? decode_rle_v1(ctx, bytes, len, is_signed, count) : decode_rle_v2(ctx, bytes, len, is_signed, count);
}

}  // namespace

DeviceColumn decode_long_column(Context *ctx, int32_t type, int32_t encoding, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len)
{
    TG_CHECK_ARG(type == TGPU_BIGINT || type == TGPU_INTEGER || type == TGPU_DATE, "ORC integer columns decode to BIGINT, INTEGER or DATE");
    TG_CHECK_ARG(n >= 0 && n <= 0x7fffffffLL, "bad position count");
    const bool v1 = integer_streams_are_v1(encoding, false);
    DeviceColumn col;
    col.type = type;
    col.n = n;
    const int w = type_width(type);
    col.values_buf = ctx->alloc((size_t)(n > 0 ? n : 1) * (size_t)w);
    col.values = col.values_buf->ptr();
    if (n == 0) return col;
    Present p = decode_present(ctx, present, present_len, n);
    int64_t count = 0;
    BufferPtr compact = decode_rle(ctx, v1, data, data_len, true, count);
    TG_CHECK_ARG(count >= p.non_null, "ORC DATA stream holds fewer values than the column has non-null positions");
    BufferPtr error = ctx->alloc_zero(4);
    const uint8_t *nulls = p.nulls ? p.nulls->as<uint8_t>() : nullptr;
    const int32_t *rank = p.rank ? p.rank->as<int32_t>() : nullptr;
    {
        ProfileScope ps(ctx, "orc_place_values");
        if (w == 8) place_values_kernel<long long><<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(compact->as<long long>(), rank, nulls, n, (long long *)col.values_buf->ptr(), error->as<unsigned int>());
        else place_values_kernel<int><<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(compact->as<long long>(), rank, nulls, n, (int *)col.values_buf->ptr(), error->as<unsigned int>());
        check_launch("orc_place_values");
    }
    raise_if(ctx, error);
    if (p.nulls && p.non_null < n) {
        col.nulls_buf = p.nulls;
        col.nulls = p.nulls->as<uint8_t>();
    }
    return col;
}

DeviceColumn decode_boolean_column(Context *ctx, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len)
{
    TG_CHECK_ARG(n >= 0 && n <= 0x7fffffffLL, "bad position count");
    DeviceColumn col;
    col.type = TGPU_BOOLEAN;
    col.n = n;
    col.values_buf = ctx->alloc((size_t)(n > 0 ? n : 1));
    col.values = col.values_buf->ptr();
    if (n == 0) return col;
    Present p = decode_present(ctx, present, present_len, n);
    if (!p.nulls || p.non_null == n) {
        decode_boolean(ctx, data, data_len, n, false, col.values_buf->as<uint8_t>(), nullptr);
        return col;
    }
    // values exist for the non-null rows only: decode them compact, then place them
    BufferPtr compact8 = ctx->alloc((size_t)(p.non_null > 0 ? p.non_null : 1)), compact64 = ctx->alloc((size_t)(p.non_null > 0 ? p.non_null : 1) * 8), error = ctx->alloc_zero(4);
    if (p.non_null > 0) {
        decode_boolean(ctx, data, data_len, p.non_null, false, compact8->as<uint8_t>(), nullptr);
        widen_u8_kernel<<<grid_for(ctx, p.non_null), 256, 0, ctx->stream()>>>(compact8->as<uint8_t>(), p.non_null, compact64->as<long long>());
        check_launch("orc_widen");
    }
    place_values_kernel<unsigned char><<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(compact64->as<long long>(), p.rank->as<int32_t>(), p.nulls->as<uint8_t>(), n,
                                                                                  col.values_buf->as<unsigned char>(), error->as<unsigned int>());
    check_launch("orc_place_values");
    col.nulls_buf = p.nulls;
    col.nulls = p.nulls->as<uint8_t>();
    return col;
}

// DOUBLE columns (DoubleColumnReader.java:92-175, stream/DoubleInputStream.java): DATA = the non-null rows' IEEE-754 doubles, 8 little-endian
// bytes each -- the bit patterns go to their rows as they are
DeviceColumn decode_double_column(Context *ctx, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len)
{
    TG_CHECK_ARG(n >= 0 && n <= 0x7fffffffLL && data_len >= 0, "bad argument");
    DeviceColumn col;
    col.type = TGPU_DOUBLE;
    col.n = n;
    col.values_buf = ctx->alloc((size_t)(n > 0 ? n : 1) * 8);
    col.values = col.values_buf->ptr();
    if (n == 0) return col;
    Present p = decode_present(ctx, present, present_len, n);
    TG_CHECK_ARG(data_len >= p.non_null * 8, "ORC DATA stream holds fewer doubles than the column has non-null positions");
    if (!p.nulls || p.non_null == n) {
        ctx->upload(col.values_buf->ptr(), data, (size_t)n * 8);
        return col;
    }
    BufferPtr compact = ctx->alloc((size_t)(p.non_null > 0 ? p.non_null : 1) * 8), error = ctx->alloc_zero(4);
    if (p.non_null > 0) ctx->upload(compact->ptr(), data, (size_t)p.non_null * 8);
    place_values_kernel<long long><<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(compact->as<long long>(), p.rank->as<int32_t>(), p.nulls->as<uint8_t>(), n,
                                                                              (long long *)col.values_buf->ptr(), error->as<unsigned int>());
    check_launch("orc_place_values");
    col.nulls_buf = p.nulls;
    col.nulls = p.nulls->as<uint8_t>();
    return col;
}

DeviceColumn decode_dictionary_string_column(Context *ctx, int32_t encoding, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len,
                                             int32_t dictionary_size, const uint8_t *length_stream, int64_t length_len, const uint8_t *dictionary_data, int64_t dictionary_data_len)
{
    TG_CHECK_ARG(n >= 0 && n <= 0x7fffffffLL && dictionary_size >= 0 && dictionary_data_len >= 0 && dictionary_data_len <= 0x7fffffffLL, "bad argument");
    const bool v1 = integer_streams_are_v1(encoding, true);
    // the dictionary: entry lengths (unsigned RLEv2) -> offsets by an exclusive scan; bytes as they are (SliceDictionaryColumnReader.java:260-300)
    DeviceColumn dict;
    dict.type = TGPU_VARCHAR;
    dict.n = dictionary_size;
    BufferPtr error = ctx->alloc_zero(4);
    dict.offsets_buf = ctx->alloc((size_t)(dictionary_size + 1) * 4);
    dict.offsets = dict.offsets_buf->as<int32_t>();
    dict.values_buf = ctx->alloc((size_t)(dictionary_data_len > 0 ? dictionary_data_len : 1));
    dict.values = dict.values_buf->ptr();
    if (dictionary_data_len) ctx->upload(dict.values_buf->ptr(), dictionary_data, (size_t)dictionary_data_len);
    dict.pool_bytes = dictionary_data_len;
    dict.pool_exact = true;
    if (dictionary_size > 0) {
        int64_t count = 0;
        BufferPtr lens64 = decode_rle(ctx, v1, length_stream, length_len, false, count);
        TG_CHECK_ARG(count >= dictionary_size, "ORC LENGTH stream holds fewer lengths than the dictionary has entries");
        BufferPtr lens = ctx->alloc((size_t)dictionary_size * 4), total = ctx->alloc(8);
        lengths_to_i32_kernel<<<grid_for(ctx, dictionary_size), 256, 0, ctx->stream()>>>(lens64->as<long long>(), dictionary_size, lens->as<int32_t>(), error->as<unsigned int>());
        check_launch("orc_lengths");
        k::exclusive_scan_i32(ctx, lens->as<int32_t>(), const_cast<int32_t *>(dict.offsets), dictionary_size, total->as<int64_t>());
        const int64_t bytes = ctx->read_scalar(total->as<int64_t>());
        TG_CHECK_ARG(bytes == dictionary_data_len, "ORC dictionary lengths do not add up to the DICTIONARY_DATA stream");
        const int32_t end = (int32_t)bytes;
        ctx->upload(const_cast<int32_t *>(dict.offsets) + dictionary_size, &end, 4);
        ctx->sync();
    }
    else HIP_CHECK(hipMemsetAsync(dict.offsets_buf->ptr(), 0, 4, ctx->stream()));
    if (n == 0) return k::region_of(ctx, dict, 0, 0);
    // the ids (unsigned RLEv2, one per non-null row) at their row positions, then the library's dictionary gather
    Present p = decode_present(ctx, present, present_len, n);
    int64_t count = 0;
    BufferPtr ids64 = decode_rle(ctx, v1, data, data_len, false, count);
    TG_CHECK_ARG(count >= p.non_null, "ORC DATA stream holds fewer ids than the column has non-null positions");
    BufferPtr ids = ctx->alloc((size_t)n * 4);
    place_ids_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(ids64->as<long long>(), p.rank ? p.rank->as<int32_t>() : nullptr, p.nulls ? p.nulls->as<uint8_t>() : nullptr, n,
                                                                  dictionary_size, ids->as<int32_t>(), error->as<unsigned int>());
    check_launch("orc_place_ids");
    raise_if(ctx, error);
    ProfileScope ps(ctx, "orc_dictionary_gather");
    return k::gather_column(ctx, dict, ids->as<int32_t>(), n, /*negative_is_null=*/p.nulls != nullptr && p.non_null < n);
}

// STRING / VARCHAR columns in DIRECT_V2 encoding (SliceDirectColumnReader.java:100-232): LENGTH = one unsigned RLEv2 length per NON-NULL row,
// DATA = those rows' bytes back to back.  The lengths go to their rows (a null row has none), an exclusive scan makes the offsets, the bytes
// are the block's bytes as they are.
DeviceColumn decode_direct_string_column(Context *ctx, int32_t encoding, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len,
                                         const uint8_t *length_stream, int64_t length_len)
{
    TG_CHECK_ARG(n >= 0 && n <= 0x7fffffffLL && data_len >= 0 && data_len <= 0x7fffffffLL, "bad argument");
    const bool v1 = integer_streams_are_v1(encoding, false);
    DeviceColumn col;
    col.type = TGPU_VARCHAR;
    col.n = n;
    col.offsets_buf = ctx->alloc((size_t)(n + 1) * 4);
    col.offsets = col.offsets_buf->as<int32_t>();
    col.values_buf = ctx->alloc((size_t)(data_len > 0 ? data_len : 1));
    col.values = col.values_buf->ptr();
    if (data_len) ctx->upload(col.values_buf->ptr(), data, (size_t)data_len);
    col.pool_bytes = data_len;
    col.pool_exact = true;
    if (n == 0) {
        HIP_CHECK(hipMemsetAsync(col.offsets_buf->ptr(), 0, 4, ctx->stream()));
        return col;
    }
    BufferPtr error = ctx->alloc_zero(4);
    Present p = decode_present(ctx, present, present_len, n);
    BufferPtr lens = ctx->alloc((size_t)n * 4), total = ctx->alloc(8);
    if (p.non_null > 0) {
        int64_t count = 0;
        BufferPtr lens64 = decode_rle(ctx, v1, length_stream, length_len, false, count);
        TG_CHECK_ARG(count >= p.non_null, "ORC LENGTH stream holds fewer lengths than the column has non-null positions");
        BufferPtr compact = ctx->alloc((size_t)p.non_null * 4);
        lengths_to_i32_kernel<<<grid_for(ctx, p.non_null), 256, 0, ctx->stream()>>>(lens64->as<long long>(), p.non_null, compact->as<int32_t>(), error->as<unsigned int>());
        check_launch("orc_lengths");
        place_lengths_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream()>>>(compact->as<int32_t>(), p.rank ? p.rank->as<int32_t>() : nullptr,
                                                                         p.nulls ? p.nulls->as<uint8_t>() : nullptr, n, lens->as<int32_t>());
        check_launch("orc_place_lengths");
    }
    else HIP_CHECK(hipMemsetAsync(lens->ptr(), 0, (size_t)n * 4, ctx->stream()));
    k::exclusive_scan_i32(ctx, lens->as<int32_t>(), const_cast<int32_t *>(col.offsets), n, total->as<int64_t>());
    const int64_t bytes = ctx->read_scalar(total->as<int64_t>());
    raise_if(ctx, error);
    TG_CHECK_ARG(bytes == data_len, "ORC string lengths do not add up to the DATA stream");
    const int32_t end = (int32_t)bytes;
    ctx->upload(const_cast<int32_t *>(col.offsets) + n, &end, 4);
    ctx->sync();
    if (p.nulls && p.non_null < n) {
        col.nulls_buf = p.nulls;
        col.nulls = p.nulls->as<uint8_t>();
    }
    return col;
}

}  // namespace orc
}  // namespace tgpu
