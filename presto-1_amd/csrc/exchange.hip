// exchange.hip -- repartition / broadcast exchange between the GPUs of one node (see exchange.h).
#include "exchange.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <array>

#include "kernels.h"

namespace tgpu {

namespace {

constexpr int kBlock = 256;

// ---- RCCL, resolved on first use (the library has no link-time dependency on librccl: single-GPU embeddings never load it) --------
struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

const Rccl &rccl()
{
    static const Rccl api = [] {
        // One RCCL per process: a host that already carries one (PyTorch ships its own librccl) must not get a second copy of the
        // same soname mixed in -- reuse what is loaded (RTLD_NOLOAD), else load ROCm's, with local symbol scope either way.
        void *h = nullptr;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *name : names) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (h) break;
        }
        for (const char *name : names) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!h) fail(TGPU_ERR_DEVICE, std::string("cannot load librccl: ") + dlerror());
        Rccl a;
        auto sym = [&](const char *n) {
            void *p = dlsym(h, n);
            if (!p) fail(TGPU_ERR_DEVICE, std::string("librccl lacks ") + n);
            return p;
        };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        a.Send = (decltype(a.Send))sym("ncclSend");
        a.Recv = (decltype(a.Recv))sym("ncclRecv");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        return a;
    }();
    return api;
}

#define NCCL_CHECK(expr)                                                                                              \
    do {                                                                                                              \
        ncclResult_t _r = (expr);                                                                                     \
        if (_r != ncclSuccess) fail(TGPU_ERR_DEVICE, std::string("RCCL error: ") + rccl().GetErrorString(_r) + " (" #expr ")"); \
    } while (0)

class RcclTransport : public ExchangeTransport {
public:
    RcclTransport(Context *ctx, const void *unique_id, int rank, int world) : ctx_(ctx), rank_(rank), world_(world)
    {
        static_assert(TGPU_EXCHANGE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof(id));
        NCCL_CHECK(rccl().CommInitRank(&comm_, world, id, rank));
        meta_ = ctx_->alloc(1);
    }
    ~RcclTransport() override
    {
        if (comm_) {
            hipStreamSynchronize(ctx_->stream());
            rccl().CommDestroy(comm_);
        }
    }
    const char *name() const override { return "rccl"; }

    void all_to_all_meta(const int64_t *send, int64_t *recv, int per_rank) override
    {
        const size_t bytes = (size_t)world_ * (size_t)per_rank * 8;
        if (meta_->capacity() < 2 * bytes) meta_ = ctx_->alloc(2 * bytes);
        int64_t *dsend = meta_->as<int64_t>(), *drecv = dsend + (size_t)world_ * per_rank;
        ctx_->upload(dsend, send, bytes);
        NCCL_CHECK(rccl().GroupStart());
        for (int r = 0; r < world_; r++) {
            NCCL_CHECK(rccl().Send(dsend + (size_t)r * per_rank, (size_t)per_rank, ncclInt64, r, comm_, ctx_->stream()));
            NCCL_CHECK(rccl().Recv(drecv + (size_t)r * per_rank, (size_t)per_rank, ncclInt64, r, comm_, ctx_->stream()));
        }
        NCCL_CHECK(rccl().GroupEnd());
        ctx_->download(recv, drecv, bytes);
    }

    void all_to_all_v(int transfers, const void *const *send_ptr, const int64_t *send_bytes, void *const *recv_ptr, const int64_t *recv_bytes) override
    {
        // every buffer of every channel to every peer in ONE group: RCCL fuses it into one launch that drives all xGMI links at once
        NCCL_CHECK(rccl().GroupStart());
        for (int t = 0; t < transfers; t++)
            for (int r = 0; r < world_; r++) {
                const size_t i = (size_t)t * world_ + r;
                if (send_bytes[i] > 0) NCCL_CHECK(rccl().Send(send_ptr[i], (size_t)send_bytes[i], ncclUint8, r, comm_, ctx_->stream()));
                if (recv_bytes[i] > 0) NCCL_CHECK(rccl().Recv(recv_ptr[i], (size_t)recv_bytes[i], ncclUint8, r, comm_, ctx_->stream()));
            }
        NCCL_CHECK(rccl().GroupEnd());
    }

private:
    Context *ctx_;
    int rank_, world_;
    ncclComm_t comm_ = nullptr;
    BufferPtr meta_;
};

class CallbackTransport : public ExchangeTransport {
public:
    CallbackTransport(Context *ctx, const tgpu_exchange_transport *v, int world) : ctx_(ctx), v_(*v), world_(world)
    {
        TG_CHECK_ARG(v_.all_to_all_meta && v_.all_to_all_v, "transport callbacks are null");
    }
    const char *name() const override { return "callbacks"; }
    void all_to_all_meta(const int64_t *send, int64_t *recv, int per_rank) override
    {
        if (v_.all_to_all_meta(v_.user, send, recv, per_rank) != 0) fail(TGPU_ERR_DEVICE, "exchange transport: all_to_all_meta failed");
    }
    void all_to_all_v(int transfers, const void *const *send_ptr, const int64_t *send_bytes, void *const *recv_ptr, const int64_t *recv_bytes) override
    {
        ctx_->sync();   // the callbacks move the bytes on their own: what they read must have been written
        if (v_.all_to_all_v(v_.user, transfers, send_ptr, send_bytes, recv_ptr, recv_bytes) != 0) fail(TGPU_ERR_DEVICE, "exchange transport: all_to_all_v failed");
    }

private:
    Context *ctx_;
    tgpu_exchange_transport v_;
    int world_;
};

// final offsets of a received VARCHAR channel: source r contributed rows [row_base, row_base + n) whose raw (sender-absolute) offsets sit
// at raw[seg_base .. seg_base + n] and whose bytes start at byte_base of the received pool
struct VarcharSegment {
    long long row_base, seg_base, byte_base, n;
};
__global__ void __launch_bounds__(kBlock) rebase_offsets_kernel(const int32_t *__restrict__ raw, const VarcharSegment *__restrict__ segs, int32_t *__restrict__ out)
{
    const VarcharSegment s = segs[blockIdx.y];
    const int first = s.n > 0 ? raw[s.seg_base] : 0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i <= s.n; i += (long long)gridDim.x * kBlock)
        if (i < s.n || blockIdx.y == gridDim.y - 1) out[s.row_base + i] = (int32_t)(s.byte_base + (raw[s.seg_base + i] - first));
}

}  // namespace

void rccl_unique_id(void *id_out)
{
    ncclUniqueId id;
    NCCL_CHECK(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
}

std::unique_ptr<ExchangeTransport> make_rccl_transport(Context *ctx, const void *unique_id, int rank, int world)
{
    return std::make_unique<RcclTransport>(ctx, unique_id, rank, world);
}

std::unique_ptr<ExchangeTransport> make_callback_transport(Context *ctx, const tgpu_exchange_transport *vtable, int world)
{
    return std::make_unique<CallbackTransport>(ctx, vtable, world);
}

Exchange::Exchange(Context *ctx, int rank, int world, std::unique_ptr<ExchangeTransport> transport)
    : ctx_(ctx), rank_(rank), world_(world), transport_(std::move(transport))
{
    TG_CHECK_ARG(world >= 1 && world <= 1024 && rank >= 0 && rank < world, "bad rank / world size");
}

DevicePage Exchange::shuffle(const std::vector<int32_t> &types, const std::vector<const DevicePage *> &per_destination)
{
    const int W = world_, C = (int)types.size();
    TG_CHECK_ARG((int)per_destination.size() == W, "one page per destination rank expected");
    for (int32_t t : types) TG_CHECK_ARG(valid_type(t), "unknown channel type");
    static const DevicePage kEmpty;
    auto page_of = [&](int r) -> const DevicePage & { return (per_destination[(size_t)r] && per_destination[(size_t)r]->n > 0) ? *per_destination[(size_t)r] : kEmpty; };
    for (int r = 0; r < W; r++) {
        const DevicePage &p = page_of(r);
        if (p.n == 0) continue;
        TG_CHECK_ARG((int)p.cols.size() == C, "page channel count does not match the exchange's types");
        for (int c = 0; c < C; c++) TG_CHECK_ARG(p.cols[(size_t)c].type == types[(size_t)c], "page channel type does not match the exchange's types");
    }
    // byte range of every outgoing VARCHAR region: known on the host for whole columns, else offsets[0] / offsets[n] in one batched read
    std::vector<std::array<int32_t, 2>> ends((size_t)W * C, {0, 0});
    {
        std::vector<Context::Transfer> reads;
        for (int r = 0; r < W; r++) {
            const DevicePage &p = page_of(r);
            if (p.n == 0) continue;
            for (int c = 0; c < C; c++) {
                const DeviceColumn &col = p.cols[(size_t)c];
                if (col.type != TGPU_VARCHAR) continue;
                auto &e = ends[(size_t)r * C + c];
                if (col.pool_exact) e = {col.pool_first, (int32_t)col.pool_bytes};
                else {
                    reads.push_back({&e[0], col.offsets, 4});
                    reads.push_back({&e[1], col.offsets + p.n, 4});
                }
            }
        }
        if (!reads.empty()) ctx_->download_batch(reads);
    }
    // 1. page headers: [rows, (has null vector, varchar bytes) per channel] to every destination
    const int per = 1 + 2 * C;
    std::vector<int64_t> smeta((size_t)W * per, 0), rmeta((size_t)W * per, 0);
    for (int r = 0; r < W; r++) {
        const DevicePage &p = page_of(r);
        smeta[(size_t)r * per] = p.n;
        if (p.n == 0) continue;
        for (int c = 0; c < C; c++) {
            smeta[(size_t)r * per + 1 + 2 * c] = p.cols[(size_t)c].nulls ? 1 : 0;
            smeta[(size_t)r * per + 2 + 2 * c] = types[(size_t)c] == TGPU_VARCHAR ? (int64_t)ends[(size_t)r * C + c][1] - ends[(size_t)r * C + c][0] : 0;
        }
    }
    transport_->all_to_all_meta(smeta.data(), rmeta.data(), per);
    std::vector<int64_t> rows((size_t)W), row_base((size_t)W + 1, 0);
    for (int r = 0; r < W; r++) {
        rows[(size_t)r] = rmeta[(size_t)r * per];
        TG_CHECK_STATE(rows[(size_t)r] >= 0, "exchange: negative row count received");
        row_base[(size_t)r + 1] = row_base[(size_t)r] + rows[(size_t)r];
    }
    const int64_t n_out = row_base[(size_t)W];
    TG_CHECK_STATE(n_out <= 0x7fffffffLL, "exchange: more than 2^31 rows for one rank in one page");

    // 2. receive buffers + the transfer list (one entry per buffer kind and channel, x world)
    DevicePage out;
    out.n = n_out;
    std::vector<const void *> sp;
    std::vector<void *> rp;
    std::vector<int64_t> sb, rb;
    auto add_transfer = [&]() {
        sp.resize(sp.size() + (size_t)W, nullptr);
        rp.resize(rp.size() + (size_t)W, nullptr);
        sb.resize(sb.size() + (size_t)W, 0);
        rb.resize(rb.size() + (size_t)W, 0);
        return sp.size() - (size_t)W;
    };
    struct PendingVarchar {
        int channel;
        BufferPtr raw;
        std::vector<VarcharSegment> segs;
    };
    std::vector<PendingVarchar> pending;
    for (int c = 0; c < C; c++) {
        DeviceColumn col;
        col.type = types[(size_t)c];
        col.n = n_out;
        // The null-vector transfer slot of a channel exists on EVERY rank, used or not: the sends follow what this rank's outgoing pages
        // carry, the receives what the incoming headers announce -- the two sides are independent (a rank may send a null vector and
        // receive none, e.g. when all its rows leave for a peer), and the transfer indices stay the same on all ranks.
        bool any_nulls = false;
        for (int r = 0; r < W; r++) any_nulls = any_nulls || (rows[(size_t)r] > 0 && rmeta[(size_t)r * per + 1 + 2 * c] != 0);
        if (any_nulls) {
            col.nulls_buf = ctx_->alloc_zero((size_t)(n_out > 0 ? n_out : 1));   // sources without a null vector: all false
            col.nulls = col.nulls_buf->as<uint8_t>();
        }
        {
            const size_t t = add_transfer();
            for (int r = 0; r < W; r++) {
                const DevicePage &p = page_of(r);
                if (p.n > 0 && p.cols[(size_t)c].nulls) {
                    sp[t + r] = p.cols[(size_t)c].nulls;
                    sb[t + r] = p.n;
                }
                if (rows[(size_t)r] > 0 && rmeta[(size_t)r * per + 1 + 2 * c] != 0) {
                    rp[t + r] = col.nulls_buf->as<uint8_t>() + row_base[(size_t)r];
                    rb[t + r] = rows[(size_t)r];
                }
            }
        }
        if (col.type != TGPU_VARCHAR) {
            const int w = type_width(col.type);
            col.values_buf = ctx_->alloc((size_t)(n_out > 0 ? n_out : 1) * (size_t)w);
            col.values = col.values_buf->ptr();
            const size_t t = add_transfer();
            for (int r = 0; r < W; r++) {
                const DevicePage &p = page_of(r);
                if (p.n > 0) {
                    sp[t + r] = p.cols[(size_t)c].values;
                    sb[t + r] = p.n * w;
                }
                rp[t + r] = (uint8_t *)col.values_buf->ptr() + row_base[(size_t)r] * w;
                rb[t + r] = rows[(size_t)r] * w;
            }
        }
        else {
            // bytes: each source's region of its pool, back to back; offsets: each source's n + 1 raw offsets, rebased below
            int64_t total_bytes = 0, raw_count = 0;
            PendingVarchar pv;
            pv.channel = c;
            for (int r = 0; r < W; r++) {
                const int64_t bytes = rmeta[(size_t)r * per + 2 + 2 * c];
                TG_CHECK_STATE(bytes >= 0, "exchange: negative byte count received");
                pv.segs.push_back(VarcharSegment{row_base[(size_t)r], raw_count, total_bytes, rows[(size_t)r]});
                total_bytes += bytes;
                raw_count += rows[(size_t)r] > 0 ? rows[(size_t)r] + 1 : 0;
            }
            TG_CHECK_STATE(total_bytes <= 0x7fffffffLL, "exchange: a VARCHAR channel of more than 2 GiB for one rank in one page");
            col.values_buf = ctx_->alloc((size_t)(total_bytes > 0 ? total_bytes : 1));
            col.values = col.values_buf->ptr();
            col.pool_bytes = total_bytes;
            col.pool_exact = true;
            col.offsets_buf = ctx_->alloc((size_t)(n_out + 1) * 4);
            col.offsets = col.offsets_buf->as<int32_t>();
            pv.raw = ctx_->alloc((size_t)(raw_count > 0 ? raw_count : 1) * 4);
            const size_t tv = add_transfer(), to = add_transfer();
            for (int r = 0; r < W; r++) {
                const DevicePage &p = page_of(r);
                if (p.n > 0) {
                    const auto &e = ends[(size_t)r * C + c];
                    sp[tv + r] = (const uint8_t *)p.cols[(size_t)c].values + e[0];
                    sb[tv + r] = (int64_t)e[1] - e[0];
                    sp[to + r] = p.cols[(size_t)c].offsets;
                    sb[to + r] = (p.n + 1) * 4;
                }
                rp[tv + r] = (uint8_t *)col.values_buf->ptr() + pv.segs[(size_t)r].byte_base;
                rb[tv + r] = rmeta[(size_t)r * per + 2 + 2 * c];
                rp[to + r] = pv.raw->as<int32_t>() + pv.segs[(size_t)r].seg_base;
                rb[to + r] = rows[(size_t)r] > 0 ? (rows[(size_t)r] + 1) * 4 : 0;
            }
            pending.push_back(std::move(pv));
        }
        out.cols.push_back(std::move(col));
    }
    // what stays on this rank never touches the transport: device-to-device copies on the context's stream
    const int transfers = (int)(sp.size() / (size_t)W);
    for (int t = 0; t < transfers; t++) {
        const size_t i = (size_t)t * W + rank_;
        TG_CHECK_STATE(sb[i] == rb[i], "exchange: self transfer sizes differ");
        if (sb[i] > 0) HIP_CHECK(hipMemcpyAsync(rp[i], sp[i], (size_t)sb[i], hipMemcpyDeviceToDevice, ctx_->stream()));
        sb[i] = rb[i] = 0;
    }
    for (int64_t b : sb) bytes_sent_ += b;
    {
        ProfileScope ps(ctx_, "exchange_all_to_all_v");
        if (W > 1 && transfers > 0) transport_->all_to_all_v(transfers, sp.data(), sb.data(), rp.data(), rb.data());
    }
    // 3. VARCHAR offsets of the received rows
    for (PendingVarchar &pv : pending) {
        DeviceColumn &col = out.cols[(size_t)pv.channel];
        BufferPtr segs = ctx_->alloc(pv.segs.size() * sizeof(VarcharSegment));
        ctx_->upload(segs->ptr(), pv.segs.data(), pv.segs.size() * sizeof(VarcharSegment));
        if (n_out == 0) {
            HIP_CHECK(hipMemsetAsync(col.offsets_buf->ptr(), 0, 4, ctx_->stream()));
            continue;
        }
        // the last segment also writes offsets[n_out]; trailing empty segments would leave it unwritten: give the LAST NON-EMPTY one that job
        int last = W - 1;
        while (last > 0 && pv.segs[(size_t)last].n == 0) last--;
        int64_t most = 1;
        for (auto &s : pv.segs) most = std::max<int64_t>(most, s.n + 1);
        dim3 grid((unsigned)std::min<int64_t>(ceil_div(most, kBlock), 1024), (unsigned)(last + 1));
        rebase_offsets_kernel<<<grid, kBlock, 0, ctx_->stream()>>>(pv.raw->as<int32_t>(), segs->as<VarcharSegment>(), const_cast<int32_t *>(col.offsets));
        check_launch("rebase_offsets");
        ctx_->sync();   // pv.segs (host) backs the async upload above
    }
    return out;
}

DevicePage Exchange::repartition(const DevicePage &in, const std::vector<int32_t> &key_channels, int32_t hash_channel)
{
    std::vector<int32_t> types;
    for (auto &c : in.cols) types.push_back(c.type);
    const int64_t n = in.n;
    std::vector<DevicePage> parts((size_t)world_);
    std::vector<const DevicePage *> per((size_t)world_, nullptr);
    if (n > 0) {
        BufferPtr own_hashes;
        const int64_t *hashes = nullptr;
        if (hash_channel >= 0) {
            TG_CHECK_ARG(hash_channel < (int)in.cols.size() && in.cols[(size_t)hash_channel].type == TGPU_BIGINT, "bad hash channel");
            hashes = (const int64_t *)in.cols[(size_t)hash_channel].values;
        }
        else {
            std::vector<const DeviceColumn *> keys;
            for (int32_t ch : key_channels) {
                TG_CHECK_ARG(ch >= 0 && ch < (int)in.cols.size(), "key channel out of range");
                keys.push_back(&in.cols[(size_t)ch]);
            }
            TG_CHECK_ARG(!keys.empty(), "partitioning needs key channels or a hash channel");
            own_hashes = ctx_->alloc((size_t)n * 8);
            k::hash_rows(ctx_, key_cols_of(keys), n, own_hashes->as<int64_t>());
            hashes = own_hashes->as<int64_t>();
        }
        BufferPtr ids = ctx_->alloc((size_t)n * 4), pos = ctx_->alloc((size_t)n * 4), cnt = ctx_->alloc((size_t)world_ * 8);
        k::partition_ids(ctx_, hashes, n, world_, ids->as<int32_t>());
        k::partition_positions(ctx_, ids->as<int32_t>(), n, world_, pos->as<int32_t>(), cnt->as<int64_t>());
        std::vector<int64_t> counts((size_t)world_);
        ctx_->download(counts.data(), cnt->ptr(), (size_t)world_ * 8);
        std::vector<DeviceColumn> grouped;
        for (auto &col : in.cols) grouped.push_back(k::gather_column(ctx_, col, pos->as<int32_t>(), n, false));
        int64_t at = 0;
        for (int r = 0; r < world_; r++) {
            parts[(size_t)r].n = counts[(size_t)r];
            for (auto &col : grouped) parts[(size_t)r].cols.push_back(k::region_of(ctx_, col, at, counts[(size_t)r]));
            per[(size_t)r] = &parts[(size_t)r];
            at += counts[(size_t)r];
        }
    }
    return shuffle(types, per);
}

DevicePage Exchange::all_gather(const DevicePage &in)
{
    std::vector<int32_t> types;
    for (auto &c : in.cols) types.push_back(c.type);
    std::vector<const DevicePage *> per((size_t)world_, &in);
    return shuffle(types, per);
}

}  // namespace tgpu
