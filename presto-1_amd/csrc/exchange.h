// exchange.h -- the repartition exchange between the GPUs of one node, inside the native library.
//
// Replaces the reference's PartitionedOutputOperator -> OutputBuffer -> HTTP -> ExchangeClient -> ExchangeOperator hop
// (M/operator/PartitionedOutputOperator.java:406-476, M/operator/ExchangeOperator.java) for ranks that are GPUs of one node:
// pages stay in HBM, the per-destination row groups of the K10 partition kernels travel over xGMI as one grouped
// ncclSend / ncclRecv all-to-all-v (RCCL), and the receiver gets one device-resident page.
//
//   1. one small all-to-all of a page header per destination: row count, per channel "has a null vector" and VARCHAR byte count;
//   2. ONE grouped exchange (ncclGroupStart .. ncclGroupEnd) carrying every buffer of every channel to every peer;
//   3. VARCHAR offsets are rebased on the receiver (a kernel); null vectors of senders without one are zero-filled.
//
// The transport is an interface: RCCL in production (librccl is loaded on first use), or host callbacks (tgpu_exchange_transport)
// so that several ranks can rehearse the whole exchange on ONE GPU (RCCL refuses two ranks on one device) and tests can inject a transport.
#pragma once

#include "common.h"

namespace tgpu {

class ExchangeTransport {
public:
    virtual ~ExchangeTransport() {}
    // host memory: send[r * per_rank .. (r + 1) * per_rank) goes to rank r; recv[r * per_rank ..) is what rank r sent here
    virtual void all_to_all_meta(const int64_t *send, int64_t *recv, int per_rank) = 0;
    // device memory, `transfers` x world entries, transfer-major: entry [t * world + r] sends send_bytes bytes at send_ptr to rank r and
    // receives recv_bytes bytes from rank r at recv_ptr.  Enqueued on / ordered with the context's stream.
    virtual void all_to_all_v(int transfers, const void *const *send_ptr, const int64_t *send_bytes, void *const *recv_ptr, const int64_t *recv_bytes) = 0;
    virtual const char *name() const = 0;
};

std::unique_ptr<ExchangeTransport> make_rccl_transport(Context *ctx, const void *unique_id, int rank, int world);
std::unique_ptr<ExchangeTransport> make_callback_transport(Context *ctx, const tgpu_exchange_transport *vtable, int world);
void rccl_unique_id(void *id_out /* TGPU_EXCHANGE_ID_BYTES */);

class Exchange {
public:
    Exchange(Context *ctx, int rank, int world, std::unique_ptr<ExchangeTransport> transport);
    int rank() const { return rank_; }
    int world() const { return world_; }
    int64_t bytes_sent() const { return bytes_sent_; }
    // per_destination[r]: the rows this rank sends to rank r (nullptr / zero rows = nothing); every page has the channel types `types`.
    // Returns what all ranks sent here, concatenated in source-rank order (rows of one source keep their order).
    DevicePage shuffle(const std::vector<int32_t> &types, const std::vector<const DevicePage *> &per_destination);
    // FIXED_HASH_DISTRIBUTION: rows grouped by (rawHash & 0x7fff...) % world (HashGenerator.java:24-35) of the key channels (or of the
    // precomputed hash channel), then shuffle
    DevicePage repartition(const DevicePage &in, const std::vector<int32_t> &key_channels, int32_t hash_channel);
    // FIXED_BROADCAST_DISTRIBUTION (a replicated join build side): every rank receives every rank's rows, in rank order
    DevicePage all_gather(const DevicePage &in);

private:
    Context *ctx_;
    int rank_, world_;
    std::unique_ptr<ExchangeTransport> transport_;
    int64_t bytes_sent_ = 0;
};

}  // namespace tgpu

struct tgpu_exchange {
    std::unique_ptr<tgpu::Exchange> ex;
    tgpu::Context *ctx = nullptr;
};
