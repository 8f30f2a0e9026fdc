// common.h -- runtime plumbing shared by every translation unit of libtgpu.so:
// error propagation, the per-context stream + caching device allocator + kernel timer,
// and the device-side column / page model (flat columns in HBM).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tgpu.h"

namespace tgpu {

// ---- errors: C++ exceptions inside, int32 codes at the C ABI ------------------------------------------------------
struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string &msg) : std::runtime_error(msg), code(c) {}
};

[[noreturn]] inline void fail(int32_t code, const std::string &msg) { throw Error(code, msg); }

#define TG_CHECK_ARG(cond, msg)                                            \
    do {                                                                   \
        if (!(cond)) ::tgpu::fail(TGPU_ERR_INVALID_ARGUMENT, (msg));       \
    } while (0)

#define TG_CHECK_STATE(cond, msg)                                          \
    do {                                                                   \
        if (!(cond)) ::tgpu::fail(TGPU_ERR_INTERNAL, (msg));               \
    } while (0)

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            ::tgpu::fail(TGPU_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(_e) + " at " +   \
                                              __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")"); \
    } while (0)

void set_last_error(const std::string &msg);

inline int type_width(int32_t t)
{
    switch (t) {
    case TGPU_BIGINT: case TGPU_DOUBLE: return 8;
    case TGPU_INTEGER: case TGPU_DATE: return 4;
    case TGPU_BOOLEAN: return 1;
    default: return 0;
    }
}
inline bool valid_type(int32_t t) { return t >= TGPU_BIGINT && t <= TGPU_VARCHAR; }
const char *type_name(int32_t t);

class Context;

// ---- device memory: RAII buffer handed out by the context's caching allocator ---------------------------------------
class DeviceBuffer {
public:
    DeviceBuffer(Context *ctx, void *ptr, size_t bytes, size_t capacity) : ctx_(ctx), ptr_(ptr), bytes_(bytes), capacity_(capacity) {}
    ~DeviceBuffer();
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    void *ptr() const { return ptr_; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(ptr_); }
    size_t bytes() const { return bytes_; }
    size_t capacity() const { return capacity_; }

private:
    Context *ctx_;
    void *ptr_;
    size_t bytes_, capacity_;
};
using BufferPtr = std::shared_ptr<DeviceBuffer>;

// ---- kernel timer: hipEvent pairs recorded on the context's stream around each launch -------------------------------
struct KernelStat {
    int64_t count = 0;
    double total_ms = 0, min_ms = 1e30, max_ms = 0;
};

class Context {
public:
    Context(int device, hipStream_t stream);
    ~Context();

    int device() const { return device_; }
    hipStream_t stream() const { return stream_; }
    void sync();
    void wait_stream();   // waits for the stream by polling (low latency): the wait in front of small read-backs

    // stream-ordered caching allocator (all work of a context is on one stream, so reuse after free is ordered)
    BufferPtr alloc(size_t bytes);
    BufferPtr alloc_zero(size_t bytes);
    void release(void *ptr, size_t capacity);
    size_t bytes_in_use() const { return in_use_; }

    // pinned host staging (uploads / small readbacks)
    void *pinned(size_t bytes);
    void upload(void *dst, const void *src, size_t bytes);          // H2D; inside an ingest scope: staged through the slab ring on the copy stream
    // Double-buffered page ingest (north_star: "pinned and streamed to HBM"): between begin_ingest and end_ingest every upload() goes
    // host -> one of two device staging slabs on a SECOND stream, so that the transfer of page i + 1 runs under the kernels of page i
    // (which occupy the context's stream); end_ingest waits for the transfer (the caller's arrays are only valid during the call), lets
    // the compute stream wait on it and moves the bytes into their stream-ordered buffers with device-to-device copies.  A slab is
    // reused only after the compute stream has drained it (event).  Hosts that hand over hipHostMalloc'ed (pinned) arrays get a true
    // asynchronous DMA; pageable arrays are staged by the runtime.  Returns false (plain path) for pages larger than the slabs may grow.
    bool begin_ingest(size_t total_bytes);
    void end_ingest();
    void flush_ingest();   // inside a scope: what has been uploaded so far is in place for kernels enqueued from now on (dictionary flattening)
    void *pinned_alloc(size_t bytes);    // hipHostMalloc for the embedding host (tgpu_pinned_alloc): e.g. the exchange client's receive buffers
    void pinned_free(void *p);
    void download(void *dst, const void *src, size_t bytes);        // D2H + sync
    // A small read-back the caller keeps enqueueing work behind: begin_read copies `bytes` (<= kReadSlotBytes) into a pinned slot and marks the
    // stream; finish_read waits for THAT point only (the kernels enqueued after begin_read keep running) and copies the bytes out.
    // Slots are owned from begin_read to finish_read (handles of one context may be driven by several threads at once: a ring that
    // merely advanced would hand a slot to a second reader before the first had looked at it); with every slot taken the read is done
    // synchronously into the AsyncRead itself.
    struct AsyncRead {
        int slot = -1;          // -2: completed synchronously, bytes in `sync_bytes`
        size_t bytes = 0;
        std::vector<uint8_t> sync_bytes;
    };
    static constexpr size_t kReadSlotBytes = 16384;
    AsyncRead begin_read(const void *src, size_t bytes);
    void finish_read(const AsyncRead &r, void *dst);
    // The same without the copy engine: a result the KERNEL writes into host memory itself.  begin_signal hands out one of the read slots
    // (fine-grained pinned memory, first kSignalWords words zeroed, `device` = the pointer the kernel stores through); the kernel's last
    // workgroup writes its result words and then, with a system-scope release, a non-zero word [kSignalWords - 1]; finish_signal polls that
    // word.  No copy, no event, no stream wait: what a page-at-a-time operator pays per page.  slot == -1: every slot taken, use begin_read.
    static constexpr int kSignalWords = 8;
    struct Signal {
        int slot = -1;
        volatile unsigned long long *host = nullptr;
        unsigned long long *device = nullptr;
    };
    Signal begin_signal();
    void finish_signal(const Signal &s, unsigned long long out[kSignalWords]);
    void abandon_signal(const Signal &s);   // the kernel that would have written the slot was never launched
    // persistent device words (kZeroedScratchBytes; word [0] rests at ~0 = the expression-error word's "no error", all others at 0) for
    // kernels that count into them and put them back before they end: a launch that needs fresh counters does not need a launch that
    // resets them.  Stream-ordered: one user at a time per context.
    static constexpr size_t kZeroedScratchBytes = 16384;
    void *zeroed_scratch();
    // many small D2H copies with ONE synchronisation: staged through pinned memory, then scattered to the destinations
    struct Transfer { void *dst; const void *src; size_t bytes; };
    void download_batch(const std::vector<Transfer> &transfers);
    template <typename T> T read_scalar(const T *dptr)
    {
        T v;
        download(&v, dptr, sizeof(T));
        return v;
    }

    // profiling
    bool profiling() const { return profiling_; }
    void set_profiling(bool on);
    void profile_reset();
    void profile_begin(const char *name);
    void profile_end();
    void profile_collect();
    std::string profile_json();

    int cu_count() const { return cu_count_; }
    // tgpu_context_set_double_sum_order: read by the aggregation operators when they create their accumulators
    // tgpu_context_set_max_output_page: 0 = no limit
    int64_t max_output_page_bytes() const { return max_out_bytes_; }
    int64_t max_output_page_rows() const { return max_out_rows_; }
    void set_max_output_page(int64_t bytes, int64_t rows) { max_out_bytes_ = bytes; max_out_rows_ = rows; }
    // tgpu_context_set_device_input_stable: borrowed TGPU_DEVICE input stays valid and unchanged until the operator's NEXT call returns
    bool device_input_stable() const { return device_input_stable_; }
    void set_device_input_stable(bool on) { device_input_stable_ = on; }
    int double_sum_order() const { return double_sum_order_; }
    void set_double_sum_order(int order) { double_sum_order_ = order; }

    // C-ABI handle accounting: the context outlives every factory / operator / output page created from it, whatever the
    // order in which the caller destroys its handles (tgpu_context_destroy defers until the last handle is gone)
    void retain_handle() { handles_++; }
    bool release_handle() { return --handles_ == 0 && destroy_requested_; }
    bool request_destroy()
    {
        destroy_requested_ = true;
        return handles_ == 0;
    }

private:
    int handles_ = 0;
    bool destroy_requested_ = false;
    int device_;
    hipStream_t stream_;
    bool own_stream_ = false;
    int cu_count_ = 256;
    int double_sum_order_ = 0;
    bool device_input_stable_ = false;
    int64_t max_out_bytes_ = 0, max_out_rows_ = 0;
    std::multimap<size_t, void *> free_;
    size_t in_use_ = 0, cached_ = 0;
    void *pinned_ = nullptr;
    size_t pinned_bytes_ = 0;
    static constexpr int kReadSlots = 64;
    void *read_slots_ = nullptr;          // kReadSlots x kReadSlotBytes pinned
    void *read_events_[kReadSlots] = {};  // hipEvent_t
    bool read_busy_[kReadSlots] = {};     // owned by a begin_read whose finish_read has not run yet (under io_mu_)
    void *read_slots_device_ = nullptr;   // the slots as the device addresses them
    void *zeroed_scratch_ = nullptr;
    void ensure_read_slots();
    bool profiling_ = false;
    struct Pending { std::string name; hipEvent_t a, b; };
    std::vector<Pending> pending_;
    std::vector<hipEvent_t> event_pool_;
    hipEvent_t wait_event_ = nullptr;
    int64_t readbacks_ = 0;   // host <- device round trips (each one waits for the stream) since the last profile_reset
    std::map<std::string, KernelStat> stats_;
    const char *cur_name_ = nullptr;
    hipEvent_t cur_a_ = nullptr;
    struct Slab { void *dev = nullptr; size_t cap = 0; hipEvent_t drained = nullptr; };
    struct PendingCopy { void *dst; const void *src; size_t bytes; };
    hipStream_t copy_stream_ = nullptr;
    hipEvent_t copy_done_ = nullptr;
    Slab slabs_[2];
    int next_slab_ = 0;
    std::atomic<int> active_slab_{-1};
    std::thread::id ingest_owner_;   // the ring serves the thread that opened the ingest scope; other threads' uploads take the plain path
    size_t slab_used_ = 0;
    std::vector<PendingCopy> pending_copies_;
    std::mutex mu_;
    std::recursive_mutex io_mu_;   // the pinned staging buffer, the polling event and the profile lists: shared by every handle of the context
};

// RAII: times everything launched on the stream between construction and destruction under one name
struct ProfileScope {
    Context *ctx;
    ProfileScope(Context *c, const char *name) : ctx(c)
    {
        if (ctx->profiling()) ctx->profile_begin(name);
    }
    ~ProfileScope()
    {
        if (ctx->profiling()) ctx->profile_end();
    }
};

// an expression-error word written by a generated kernel: ~0 = none, else (row << 8) | code (7 = division by zero, 9 = invalid cast argument, else overflow)
inline void raise_expression_error(unsigned long long e)
{
    if (e == ~0ull) return;
    const long long row = (long long)(e >> 8);
    if ((int)(e & 0xff) == 7) fail(TGPU_ERR_DIVISION_BY_ZERO, "Division by zero (position " + std::to_string(row) + ")");
    if ((int)(e & 0xff) == 9) fail(TGPU_ERR_INVALID_CAST_ARGUMENT, "Cannot cast double to an integer type: NaN or out of range (position " + std::to_string(row) + ")");
    fail(TGPU_ERR_NUMERIC_VALUE_OUT_OF_RANGE, "numeric value out of range: arithmetic overflow (position " + std::to_string(row) + ")");
}

inline void check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) fail(TGPU_ERR_DEVICE, std::string("kernel launch failed (") + what + "): " + hipGetErrorString(e));
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device columns: always FLAT in HBM (dictionary / RLE inputs are flattened at ingest) ---------------------------
// Layout in HBM = the reference's block arrays: values[n] (8/4/1 bytes), nulls[n] one byte per row or absent,
// VARCHAR: byte pool + int32 offsets[n+1].
struct DeviceColumn {
    int32_t type = 0;
    int64_t n = 0;
    const void *values = nullptr;
    const uint8_t *nulls = nullptr;     // nullptr = no nulls
    const int32_t *offsets = nullptr;   // VARCHAR
    int64_t pool_bytes = 0;             // VARCHAR byte pool size
    bool pool_exact = false;            // VARCHAR: offsets[0] == pool_first and offsets[n] == pool_bytes are known on the host (no read-back needed)
    int32_t pool_first = 0;             //          offsets[0] (0 for columns the library built; a borrowed block may be a region of a larger one)
    BufferPtr values_buf, nulls_buf, offsets_buf;  // owners (may be empty for borrowed device input)

    int64_t value_bytes() const { return type == TGPU_VARCHAR ? pool_bytes : n * type_width(type); }
    int64_t size_in_bytes() const { return value_bytes() + (nulls ? n : 0) + (offsets ? (n + 1) * 4 : 0); }
};

struct DevicePage {
    int64_t n = 0;
    std::vector<DeviceColumn> cols;
    int64_t size_in_bytes() const
    {
        int64_t s = 0;
        for (auto &c : cols) s += c.size_in_bytes();
        return s;
    }
};

}  // namespace tgpu
// plain structs passed by value to kernels (device_cols.h: shared with the JIT-compiled kernels)
#include "device_hash.h"
#include "device_cols.h"
namespace tgpu {
using ColView = TgColView;
using KeyCols = TgKeyCols;
constexpr int kMaxKeyChannels = TG_MAX_KEY_CHANNELS;

inline ColView view_of(const DeviceColumn &c) { return ColView{c.values, c.nulls, c.offsets, c.type, 0}; }
inline KeyCols key_cols_of(const std::vector<const DeviceColumn *> &cols)
{
    TG_CHECK_ARG((int)cols.size() <= kMaxKeyChannels, "at most 8 key channels are supported");
    KeyCols k{};
    k.n = (int32_t)cols.size();
    for (size_t i = 0; i < cols.size(); i++) k.c[i] = view_of(*cols[i]);
    return k;
}

// ingest: tgpu_page (host or device memory, any encoding) -> flat device columns.  columns.cpp
// a device column parked in host memory (the aggregation's spilled runs) and its way back into owned device buffers
struct HostColumn {
    int32_t type = 0;
    int64_t n = 0;
    bool has_nulls = false;
    std::vector<uint8_t> values, nulls;
    std::vector<int32_t> offsets;   // VARCHAR: n + 1, rebased to 0
    int64_t bytes() const { return (int64_t)(values.size() + nulls.size() + offsets.size() * 4); }
};
HostColumn download_column(Context *ctx, const DeviceColumn &c);
DeviceColumn upload_column(Context *ctx, const HostColumn &h);

// resolve_varchar = false leaves the byte ranges of borrowed device-resident VARCHAR columns unread (pool_exact == false: no
// round trip to the device); a consumer that needs them calls resolve_varchar_ends
DevicePage ingest_page(Context *ctx, const tgpu_page *page, bool resolve_varchar = true);
void resolve_varchar_ends(Context *ctx, DevicePage &page);
DeviceColumn ingest_block(Context *ctx, const tgpu_block *block);
// a DICTIONARY / RLE block as (flat dictionary column, device ids) -- not flattened (dictionary-aware processing)
void ingest_dictionary(Context *ctx, const tgpu_block *block, DeviceColumn &dictionary, BufferPtr &ids);
// gives every column of `page` that still borrows caller memory (device-resident input) buffers of its own (a device copy)
void own_borrowed_columns(Context *ctx, DevicePage &page);

// output pages handed across the C ABI
struct OutputPage {
    Context *ctx;
    DevicePage page;
    std::vector<tgpu_block> blocks;  // filled lazily by as_page
};

}  // namespace tgpu

struct tgpu_output_page : tgpu::OutputPage {};
