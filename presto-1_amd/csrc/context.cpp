// context.cpp -- Context: device/stream binding, caching allocator, pinned staging, HIP-event kernel timer.
#include "common.h"

#include <execinfo.h>

#include <sstream>

namespace tgpu {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
const std::string &last_error() { return g_last_error; }

const char *type_name(int32_t t)
{
    switch (t) {
    case TGPU_BIGINT: return "bigint";
    case TGPU_INTEGER: return "integer";
    case TGPU_DATE: return "date";
    case TGPU_DOUBLE: return "double";
    case TGPU_BOOLEAN: return "boolean";
    case TGPU_VARCHAR: return "varchar";
    default: return "?";
    }
}

DeviceBuffer::~DeviceBuffer()
{
    if (ptr_) ctx_->release(ptr_, capacity_);
}

Context::Context(int device, hipStream_t stream) : device_(device), stream_(stream)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        fail(TGPU_ERR_DEVICE, "no HIP device available: libtgpu has no CPU fallback");
    TG_CHECK_ARG(device >= 0 && device < count, "device ordinal out of range");
    HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    cu_count_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

Context::~Context()
{
    hipSetDevice(device_);
    hipStreamSynchronize(stream_);
    for (auto &kv : free_) hipFree(kv.second);
    for (auto &p : pending_) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto ev : event_pool_) hipEventDestroy(ev);
    for (Slab &sl : slabs_) {
        if (sl.dev) hipFree(sl.dev);
        if (sl.drained) hipEventDestroy(sl.drained);
    }
    if (copy_done_) hipEventDestroy(copy_done_);
    if (copy_stream_) hipStreamDestroy(copy_stream_);
    if (pinned_) hipHostFree(pinned_);
    if (wait_event_) hipEventDestroy(wait_event_);
    if (zeroed_scratch_) hipFree(zeroed_scratch_);
    if (read_slots_) {
        hipHostFree(read_slots_);
        for (void *e : read_events_)
            if (e) hipEventDestroy(static_cast<hipEvent_t>(e));
    }
}

void Context::sync() { HIP_CHECK(hipStreamSynchronize(stream_)); }

// The wait in front of a small read-back: polls an event instead of blocking in hipStreamSynchronize (whose wake-up costs tens
// of microseconds -- with a dozen read-backs per page that is a sizeable part of a 5 ms step).  TGPU_BLOCKING_WAIT=1 restores
// the blocking wait (an embedding that must not spin a core).
void Context::wait_stream()
{
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    static const bool blocking = getenv("TGPU_BLOCKING_WAIT") != nullptr;
    if (blocking) {
        HIP_CHECK(hipStreamSynchronize(stream_));
        return;
    }
    if (!wait_event_) HIP_CHECK(hipEventCreateWithFlags(&wait_event_, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(wait_event_, stream_));
    for (;;) {
        const hipError_t e = hipEventQuery(wait_event_);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) HIP_CHECK(e);
    }
}

static size_t round_capacity(size_t bytes)
{
    if (bytes < 256) return 256;
    if (bytes <= (1u << 20)) {  // next power of two below 1 MiB
        size_t c = 256;
        while (c < bytes) c <<= 1;
        return c;
    }
    const size_t g = 2u << 20;  // 2 MiB granules above
    return (bytes + g - 1) / g * g;
}

BufferPtr Context::alloc(size_t bytes)
{
    size_t cap = round_capacity(bytes ? bytes : 1);
    void *p = nullptr;
    {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = free_.lower_bound(cap);
        if (it != free_.end() && it->first <= cap + cap / 4 + 4096) {
            p = it->second;
            cap = it->first;
            cached_ -= cap;
            free_.erase(it);
        }
    }
    if (!p) {
        HIP_CHECK(hipSetDevice(device_));
        hipError_t e = hipMalloc(&p, cap);
        if (e != hipSuccess) {
            // drop the cache and retry once
            {
                std::lock_guard<std::mutex> lk(mu_);
                hipStreamSynchronize(stream_);
                for (auto &kv : free_) hipFree(kv.second);
                free_.clear();
                cached_ = 0;
            }
            (void)hipGetLastError();
            e = hipMalloc(&p, cap);
            if (e != hipSuccess)
                fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "out of device memory allocating " + std::to_string(cap) + " bytes");
        }
    }
    {
        std::lock_guard<std::mutex> lk(mu_);
        in_use_ += cap;
    }
    // TGPU_POISON_ALLOC=<byte>: every buffer is filled with that byte when it is handed out, so that a kernel that reads memory it
    // never wrote sees the same garbage on every run (test mode: recycled buffers otherwise hold whatever their last user left)
    static const int poison = getenv("TGPU_POISON_ALLOC") ? atoi(getenv("TGPU_POISON_ALLOC")) & 0xff : -1;
    if (poison >= 0) HIP_CHECK(hipMemsetAsync(p, poison, cap, stream_));
    return std::make_shared<DeviceBuffer>(this, p, bytes, cap);
}

BufferPtr Context::alloc_zero(size_t bytes)
{
    BufferPtr b = alloc(bytes);
    if (bytes) HIP_CHECK(hipMemsetAsync(b->ptr(), 0, bytes, stream_));
    return b;
}

void Context::release(void *ptr, size_t capacity)
{
    std::lock_guard<std::mutex> lk(mu_);
    in_use_ -= capacity;
    // keep at most 64 GiB cached; beyond that give memory back to the driver
    if (cached_ + capacity > (64ull << 30)) {
        hipStreamSynchronize(stream_);
        hipFree(ptr);
        return;
    }
    free_.emplace(capacity, ptr);
    cached_ += capacity;
}

void *Context::pinned(size_t bytes)
{
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    if (bytes > pinned_bytes_) {
        if (pinned_) {
            HIP_CHECK(hipStreamSynchronize(stream_));
            HIP_CHECK(hipHostFree(pinned_));
        }
        size_t cap = 1 << 20;
        while (cap < bytes) cap <<= 1;
        HIP_CHECK(hipHostMalloc(&pinned_, cap, hipHostMallocDefault));
        pinned_bytes_ = cap;
    }
    return pinned_;
}

bool Context::begin_ingest(size_t total_bytes)
{
    static const bool disabled = getenv("TGPU_DISABLE_INGEST_RING") != nullptr;
    if (disabled || total_bytes < (256u << 10) || total_bytes > (4ull << 30)) return false;   // tiny pages: latency matters more than overlap
    io_mu_.lock();   // one ingest at a time per context (released by end_ingest)
    try {
        if (!copy_stream_) {
            HIP_CHECK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&copy_done_, hipEventDisableTiming));
        }
        Slab &sl = slabs_[next_slab_];
        const size_t need = total_bytes + 256 * 64;   // per-upload alignment slack
        if (sl.cap < need) {
            if (sl.dev) {
                HIP_CHECK(hipEventSynchronize(sl.drained));
                HIP_CHECK(hipFree(sl.dev));
                sl.dev = nullptr;
                sl.cap = 0;
            }
            size_t cap = 64u << 20;
            while (cap < need) cap <<= 1;
            HIP_CHECK(hipMalloc(&sl.dev, cap));
            sl.cap = cap;
            if (!sl.drained) HIP_CHECK(hipEventCreateWithFlags(&sl.drained, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(sl.drained, stream_));
        }
        HIP_CHECK(hipStreamWaitEvent(copy_stream_, sl.drained, 0));   // the compute stream has read the slab's previous page out
        ingest_owner_ = std::this_thread::get_id();
        active_slab_ = next_slab_;
        next_slab_ ^= 1;
        slab_used_ = 0;
        pending_copies_.clear();
    }
    catch (...) {
        io_mu_.unlock();
        throw;
    }
    return true;
}

void Context::flush_ingest()
{
    if (active_slab_ < 0 || ingest_owner_ != std::this_thread::get_id() || pending_copies_.empty()) return;
    HIP_CHECK(hipEventRecord(copy_done_, copy_stream_));
    HIP_CHECK(hipEventSynchronize(copy_done_));
    for (const PendingCopy &c : pending_copies_) HIP_CHECK(hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToDevice, stream_));
    pending_copies_.clear();
}

void Context::end_ingest()
{
    if (active_slab_ < 0) return;
    Slab &sl = slabs_[active_slab_];
    try {
        HIP_CHECK(hipEventRecord(copy_done_, copy_stream_));
        HIP_CHECK(hipEventSynchronize(copy_done_));                   // the caller's arrays have been consumed (pinned sources are real async DMA)
        for (const PendingCopy &c : pending_copies_) HIP_CHECK(hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToDevice, stream_));
        HIP_CHECK(hipEventRecord(sl.drained, stream_));
        pending_copies_.clear();
        active_slab_ = -1;
    }
    catch (...) {
        active_slab_ = -1;
        io_mu_.unlock();
        throw;
    }
    io_mu_.unlock();
}

void *Context::pinned_alloc(size_t bytes)
{
    void *p = nullptr;
    HIP_CHECK(hipSetDevice(device_));
    HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
    return p;
}

void Context::pinned_free(void *p)
{
    if (p) HIP_CHECK(hipHostFree(p));
}

void Context::upload(void *dst, const void *src, size_t bytes)
{
    if (!bytes) return;
    if (active_slab_ >= 0 && ingest_owner_ == std::this_thread::get_id()) {
        Slab &sl = slabs_[active_slab_];
        const size_t at = (slab_used_ + 255) / 256 * 256;
        if (at + bytes <= sl.cap) {
            void *stage = (uint8_t *)sl.dev + at;
            HIP_CHECK(hipMemcpyAsync(stage, src, bytes, hipMemcpyHostToDevice, copy_stream_));
            pending_copies_.push_back({dst, stage, bytes});
            slab_used_ = at + bytes;
            return;
        }
    }
    // Java heap arrays are pageable; hipMemcpyAsync from pageable memory stages internally and returns once the
    // source has been consumed, which is what the ownership rule needs (the caller may reuse src after the call).
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream_));
}

// TGPU_DEBUG_READBACKS: who waits for the device (a stack trace per host <- device round trip, to stderr)
static void trace_readback(const char *what)
{
    static const bool on = getenv("TGPU_DEBUG_READBACKS") != nullptr;
    if (!on) return;
    void *bt[10];
    const int n = backtrace(bt, 10);
    fprintf(stderr, "[tgpu] read-back (%s)\n", what);
    backtrace_symbols_fd(bt + 1, n - 1, 2);
}

void Context::download(void *dst, const void *src, size_t bytes)
{
    if (!bytes) return;
    // handles of one context may be driven by different threads (tgpu.h threading rule): they share the staging buffer
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    readbacks_++;
    trace_readback("download");
    if (bytes <= (64u << 10)) {
        // the small read-backs between kernels (counts, flags, key ranges) go through the pinned staging buffer: a copy into
        // pageable memory (a stack variable) takes the runtime's slow staged path, several times the latency of this one
        void *stage = pinned(bytes);
        HIP_CHECK(hipMemcpyAsync(stage, src, bytes, hipMemcpyDeviceToHost, stream_));
        wait_stream();
        memcpy(dst, stage, bytes);
        return;
    }
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void Context::ensure_read_slots()
{
    if (read_slots_) return;
    // fine-grained (coherent) and mapped: kernels store results straight into the slots (begin_signal), the host polls them
    HIP_CHECK(hipHostMalloc(&read_slots_, (size_t)kReadSlots * kReadSlotBytes, hipHostMallocCoherent | hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer(&read_slots_device_, read_slots_, 0));
    for (int i = 0; i < kReadSlots; i++) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        read_events_[i] = e;
    }
}

Context::Signal Context::begin_signal()
{
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    ensure_read_slots();
    Signal s;
    for (int i = 0; i < kReadSlots && s.slot < 0; i++)
        if (!read_busy_[i]) s.slot = i;
    if (s.slot < 0) return s;
    readbacks_++;
    trace_readback("signal");
    read_busy_[s.slot] = true;
    s.host = reinterpret_cast<volatile unsigned long long *>(static_cast<uint8_t *>(read_slots_) + (size_t)s.slot * kReadSlotBytes);
    s.device = reinterpret_cast<unsigned long long *>(static_cast<uint8_t *>(read_slots_device_) + (size_t)s.slot * kReadSlotBytes);
    for (int i = 0; i < kSignalWords; i++) s.host[i] = 0;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    return s;
}

void Context::finish_signal(const Signal &s, unsigned long long out[kSignalWords])
{
    TG_CHECK_STATE(s.slot >= 0 && s.slot < kReadSlots && s.host, "no signal in flight");
    struct Release {
        Context *c;
        int slot;
        ~Release()
        {
            std::lock_guard<std::recursive_mutex> io(c->io_mu_);
            c->read_busy_[slot] = false;
        }
    } release{this, s.slot};
    // the flag word arrives while the stream keeps running; a stream that has drained without it means the kernel never ran to its end.
    // TGPU_BLOCKING_WAIT=1 (an embedding that must not spin a core): wait for the stream instead of polling the word
    static const bool blocking = getenv("TGPU_BLOCKING_WAIT") != nullptr;
    if (blocking && __atomic_load_n(const_cast<unsigned long long *>(&s.host[kSignalWords - 1]), __ATOMIC_ACQUIRE) == 0) HIP_CHECK(hipStreamSynchronize(stream_));
    for (uint64_t spins = 1;; spins++) {
        if (__atomic_load_n(const_cast<unsigned long long *>(&s.host[kSignalWords - 1]), __ATOMIC_ACQUIRE) != 0) break;
        if ((spins & 0xfffff) == 0) {
            const hipError_t q = hipStreamQuery(stream_);
            if (q == hipSuccess) {
                if (__atomic_load_n(const_cast<unsigned long long *>(&s.host[kSignalWords - 1]), __ATOMIC_ACQUIRE) != 0) break;
                fail(TGPU_ERR_DEVICE, "a kernel finished without delivering its result words");
            }
            if (q != hipErrorNotReady) HIP_CHECK(q);
        }
    }
    for (int i = 0; i < kSignalWords; i++) out[i] = s.host[i];
}

void Context::abandon_signal(const Signal &s)
{
    if (s.slot < 0) return;
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    read_busy_[s.slot] = false;
}

void *Context::zeroed_scratch()
{
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    if (!zeroed_scratch_) {
        HIP_CHECK(hipMalloc(&zeroed_scratch_, kZeroedScratchBytes));
        HIP_CHECK(hipMemsetAsync(zeroed_scratch_, 0, kZeroedScratchBytes, stream_));
        HIP_CHECK(hipMemsetAsync(zeroed_scratch_, 0xff, 8, stream_));   // word [0] rests at ~0: "no error" of the expression-error word
    }
    return zeroed_scratch_;
}

Context::AsyncRead Context::begin_read(const void *src, size_t bytes)
{
    TG_CHECK_STATE(bytes > 0 && bytes <= kReadSlotBytes, "asynchronous read-backs are at most 16 KB");
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    readbacks_++;
    trace_readback("async read");
    ensure_read_slots();
    AsyncRead r;
    r.bytes = bytes;
    for (int i = 0; i < kReadSlots && r.slot < 0; i++)
        if (!read_busy_[i]) r.slot = i;
    if (r.slot < 0) {
        // every slot is owned by a reader that has not finished yet (many operators of this context inside a read at once)
        r.sync_bytes.resize(bytes);
        download(r.sync_bytes.data(), src, bytes);
        readbacks_--;   // (counted once)
        r.slot = -2;
        return r;
    }
    read_busy_[r.slot] = true;
    HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t *>(read_slots_) + (size_t)r.slot * kReadSlotBytes, src, bytes, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(read_events_[r.slot]), stream_));
    return r;
}

void Context::finish_read(const AsyncRead &r, void *dst)
{
    if (r.slot == -2) {
        memcpy(dst, r.sync_bytes.data(), r.bytes);
        return;
    }
    TG_CHECK_STATE(r.slot >= 0 && r.slot < kReadSlots, "no read-back in flight");
    struct Release {   // the slot goes back whatever happens below
        Context *c;
        int slot;
        ~Release()
        {
            std::lock_guard<std::recursive_mutex> io(c->io_mu_);
            c->read_busy_[slot] = false;
        }
    } release{this, r.slot};
    hipEvent_t e = static_cast<hipEvent_t>(read_events_[r.slot]);
    static const bool blocking = getenv("TGPU_BLOCKING_WAIT") != nullptr;
    if (blocking) HIP_CHECK(hipEventSynchronize(e));
    else
        for (;;) {
            const hipError_t q = hipEventQuery(e);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIP_CHECK(q);
        }
    memcpy(dst, static_cast<uint8_t *>(read_slots_) + (size_t)r.slot * kReadSlotBytes, r.bytes);
}

void Context::download_batch(const std::vector<Transfer> &transfers)
{
    size_t total = 0;
    for (auto &t : transfers) total += (t.bytes + 63) / 64 * 64;
    if (!total) return;
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    readbacks_++;
    trace_readback("download batch");
    uint8_t *stage = static_cast<uint8_t *>(pinned(total));
    size_t off = 0;
    for (auto &t : transfers) {
        if (t.bytes) HIP_CHECK(hipMemcpyAsync(stage + off, t.src, t.bytes, hipMemcpyDeviceToHost, stream_));
        off += (t.bytes + 63) / 64 * 64;
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    off = 0;
    for (auto &t : transfers) {
        if (t.bytes) memcpy(t.dst, stage + off, t.bytes);
        off += (t.bytes + 63) / 64 * 64;
    }
}

void Context::set_profiling(bool on) { profiling_ = on; }

void Context::profile_reset()
{
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    profile_collect();
    stats_.clear();
    readbacks_ = 0;
}

void Context::profile_begin(const char *name)
{
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    if (cur_name_) return;  // nested scopes: the outermost wins
    hipEvent_t a;
    if (!event_pool_.empty()) { a = event_pool_.back(); event_pool_.pop_back(); }
    else HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventRecord(a, stream_));
    cur_name_ = name;
    cur_a_ = a;
}

void Context::profile_end()
{
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    if (!cur_name_) return;
    hipEvent_t b;
    if (!event_pool_.empty()) { b = event_pool_.back(); event_pool_.pop_back(); }
    else HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(b, stream_));
    pending_.push_back(Pending{cur_name_, cur_a_, b});
    cur_name_ = nullptr;
    cur_a_ = nullptr;
    if (pending_.size() > 4096) profile_collect();
}

void Context::profile_collect()
{
    std::lock_guard<std::recursive_mutex> io(io_mu_);
    if (pending_.empty()) return;
    HIP_CHECK(hipStreamSynchronize(stream_));
    for (auto &p : pending_) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            KernelStat &s = stats_[p.name];
            s.count++;
            s.total_ms += ms;
            if (ms < s.min_ms) s.min_ms = ms;
            if (ms > s.max_ms) s.max_ms = ms;
        }
        event_pool_.push_back(p.a);
        event_pool_.push_back(p.b);
    }
    pending_.clear();
}

std::string Context::profile_json()
{
    profile_collect();
    std::ostringstream os;
    os << "{";
    bool first = true;
    for (auto &kv : stats_) {
        if (!first) os << ", ";
        first = false;
        os << "\"" << kv.first << "\": {\"count\": " << kv.second.count << ", \"total_ms\": " << kv.second.total_ms
           << ", \"min_ms\": " << kv.second.min_ms << ", \"max_ms\": " << kv.second.max_ms << "}";
    }
    // pseudo entry: host <- device round trips since the last reset (every one of them waits for the stream)
    if (!first) os << ", ";
    os << "\"__readbacks\": {\"count\": " << readbacks_ << ", \"total_ms\": 0, \"min_ms\": 0, \"max_ms\": 0}";
    os << "}";
    return os.str();
}

}  // namespace tgpu
