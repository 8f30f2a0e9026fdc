// groupby.hip -- GroupByHash on the GPU (K4/K5).
//
// Table: one uint64 word per slot in HBM.
//     EMPTY           = all ones
//     NEW(tag, row)   = 01 | tag16 << 32 | row32   -- claimed in the sub-batch being processed by input row `row`
//     OLD(tag, gid)   = 00 | tag16 << 32 | gid32   -- key of group `gid`, stored in the key store
// tag = bits 48..63 of fmix64(rawHash) (the reference keeps a 1-byte tag: MultiChannelGroupByHash.java:441-447);
// slot = fmix64(rawHash) & mask (H6).  The table layout is not observable through the reference's API -- only the
// group ids are -- so it is free to differ from the Java arrays; the ids are not:
//
// Bit-exact first-seen ids in parallel (SURVEY.md hard part 1):
//   1. insert : every row probes; an empty slot is claimed with CAS(EMPTY -> NEW(tag,row)); a row that meets a NEW slot
//               of its own key lowers the slot's row with atomicMin, so after the kernel each new key's slot holds the
//               MINIMUM input row carrying that key.
//   2. mark   : flag[row] = 1 iff the slot of the row's key holds exactly this row (= the key's first occurrence).
//   3. scan   : rank = exclusive prefix sum of the flags = how many new keys were first seen before this row.
//   4. final  : gid = nextGroupId + rank; the slot becomes OLD(tag,gid) and the key is copied into the key store at gid.
//   5. resolve: rows that hit a NEW slot read their gid from the (now OLD) slot.
// Java does `groupId = nextGroupId++` in row order (BigintGroupByHash.java:246-253), which is exactly rank order.
#include "groupby.h"
#include "kernels.h"
#include "device_hash.h"
#include "device_groupby.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace tgpu {

namespace {

constexpr int kBlock = 256;
constexpr uint64_t kEmpty = ~0ull;

__device__ __forceinline__ uint64_t make_old(uint32_t tag, uint32_t gid) { return ((uint64_t)tag << 32) | gid; }

// generic key accessor for the shared probe protocol (device_groupby.h): run-time typed key columns
// The key sets live in device MEMORY (uploaded per launch) and are reached through references: run-time column indices are
// then plain (scalar) loads.  Passed by value as kernel parameters they would be copied to scratch on the first run-time index,
// or -- fully unrolled instead -- inflate this kernel to 35 k instructions (measured: 0.6 ms for 3 M rows from I-cache misses).
struct GenericKeys {
    const KeyCols &batch, &store;
    const int64_t *hashes;
    __device__ long long hash(long long r) const { return hashes ? hashes[r] : tg_hash_row(batch, r); }
    __device__ bool eq_store(long long r, int gid) const { return tg_rows_not_distinct(batch, r, store, gid); }
    __device__ bool eq_row(long long r, long long r2) const { return tg_rows_not_distinct(batch, r, batch, r2); }
};

// counters: [0] pending rows, [2] error
template <bool INSERT>
__global__ void __launch_bounds__(kBlock) gbh_probe_kernel(const KeyCols *batch_p, const int64_t *__restrict__ hashes, const uint8_t *__restrict__ row_mask, int64_t n,
                                                            uint64_t *words, uint64_t mask, const KeyCols *store_p, int32_t store_groups,
                                                            int32_t *__restrict__ out, unsigned long long *counters)
{
    // hashes == nullptr: the raw hash is computed from the key cells; row_mask: rows with 0 take no part (out = -1)
    GenericKeys k{*batch_p, *store_p, hashes};
    const int lane = threadIdx.x & 63;
    unsigned int my_pending = 0;
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
        const int64_t r = base + threadIdx.x;
        const bool active = r < n && (!row_mask || row_mask[r]);
        // Runs of equal keys in adjacent rows (inputs clustered by key, e.g. the lineitems of one order) would hammer one table
        // slot from neighbouring lanes: only the first row of a run inside the wave goes to the table, the others copy its
        // answer (same group, or the same pending slot -- whose first row is the leader's or an earlier one anyway).
        const long long h = active ? k.hash(r) : 0;
        const long long h_prev = __shfl_up(h, 1, 64);
        const int active_prev = __shfl_up(active ? 1 : 0, 1, 64);
        const bool follower = active && lane > 0 && active_prev && h == h_prev && k.eq_row(r, r - 1);
        const unsigned long long leaders = __ballot(active && !follower);
        bool pending = false;
        int32_t result = -1;
        if (active && !follower) result = tg_gbh_probe<INSERT>(k, r, (unsigned long long *)words, (unsigned long long)mask, store_groups, counters, pending);
        const unsigned long long at_or_below = leaders & ((2ULL << lane) - 1ULL);
        const int src = at_or_below ? 63 - __clzll((long long)at_or_below) : lane;
        const int32_t lead_result = __shfl(result, src, 64);
        const int lead_pending = __shfl(pending ? 1 : 0, src, 64);
        if (follower) {
            result = lead_result;
            pending = lead_pending != 0;
        }
        if (r < n) out[r] = result;
        if (INSERT && pending) my_pending++;
    }
    if (INSERT) {
        // one atomic per workgroup: tens of thousands of waves adding to the same word would serialise on it
        __shared__ unsigned int s_pending[kBlock / 64];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) my_pending += __shfl_down(my_pending, d, 64);
        if (lane == 0) s_pending[threadIdx.x >> 6] = my_pending;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long total = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; w++) total += s_pending[w];
            if (total) atomicAdd(&counters[0], total);
        }
    }
}

// compact ids of a sub-batch that went through the full protocol: byte = final group id + 1 (0 = excluded row)
__global__ void __launch_bounds__(kBlock) gbh_narrow_kernel(const int32_t *__restrict__ gids, int64_t n, uint8_t *__restrict__ out8)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) out8[r] = (uint8_t)(gids[r] + 1);
}

__global__ void __launch_bounds__(kBlock) gbh_mark_kernel(const int32_t *__restrict__ out, int64_t n, const uint64_t *__restrict__ words, int32_t *__restrict__ flags)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        int32_t o = out[r];
        int32_t f = 0;
        if (o < -1) {
            uint64_t w = words[(uint64_t)(-(o + 2))];
            f = ((uint32_t)w == (uint32_t)r) ? 1 : 0;
        }
        flags[r] = f;
    }
}

// step 4: first-occurrence rows publish their group: slot -> OLD, fixed-width keys + raw hash into the store,
// varchar lengths into len[ch][rank] for the byte copy that follows
struct VarLens {
    int32_t *len[kMaxKeyChannels];
};

__global__ void __launch_bounds__(kBlock) gbh_finalize_kernel(const KeyCols *batch_p, const int64_t *__restrict__ hashes, int64_t n, uint64_t *words,
                                                               const int32_t *__restrict__ out, const int32_t *__restrict__ flags,
                                                               const int32_t *__restrict__ rank, int64_t base_gid, const KeyCols *store_p,
                                                               int64_t *__restrict__ raw_hash, VarLens lens)
{
    const KeyCols &batch = *batch_p, &store = *store_p;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        if (!flags[r]) continue;
        const uint64_t pos = (uint64_t)(-(out[r] + 2));
        const int64_t gid = base_gid + rank[r];
        const uint64_t w = words[pos];
        words[pos] = make_old((uint32_t)((w >> 32) & 0xffff), (uint32_t)gid);
        raw_hash[gid] = hashes ? hashes[r] : tg_hash_row(batch, r);
        for (int c = 0; c < batch.n; c++) {
            const ColView &s = batch.c[c];
            const ColView &d = store.c[c];
            const bool isnull = s.nulls && s.nulls[r];
            ((uint8_t *)d.nulls)[gid] = isnull ? 1 : 0;
            switch (s.type) {
            case TGPU_BIGINT:
            case TGPU_DOUBLE: ((int64_t *)d.values)[gid] = isnull ? 0 : ((const int64_t *)s.values)[r]; break;
            case TGPU_INTEGER:
            case TGPU_DATE: ((int32_t *)d.values)[gid] = isnull ? 0 : ((const int32_t *)s.values)[r]; break;
            case TGPU_BOOLEAN: ((uint8_t *)d.values)[gid] = isnull ? 0 : ((const uint8_t *)s.values)[r]; break;
            case TGPU_VARCHAR: lens.len[c][rank[r]] = isnull ? 0 : s.offsets[r + 1] - s.offsets[r]; break;
            default: break;
            }
        }
    }
}

// varchar key bytes of the new groups: store.offsets[gid+1] = pool_base + off[rank] + len
__global__ void __launch_bounds__(kBlock) gbh_copy_varchar_kernel(ColView src, int64_t n, const int32_t *__restrict__ flags,
                                                                   const int32_t *__restrict__ rank, int64_t base_gid, const int32_t *__restrict__ off,
                                                                   int64_t pool_base, uint8_t *__restrict__ pool, int32_t *__restrict__ store_offsets)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        if (!flags[r]) continue;
        const int32_t k = rank[r];
        const bool isnull = src.nulls && src.nulls[r];
        const int32_t a = src.offsets[r], len = isnull ? 0 : src.offsets[r + 1] - a;
        const int64_t dst = pool_base + off[k];
        const uint8_t *p = (const uint8_t *)src.values + a;
        for (int32_t i = 0; i < len; i++) pool[dst + i] = p[i];
        store_offsets[base_gid + k + 1] = (int32_t)(dst + len);
    }
}

__global__ void __launch_bounds__(kBlock) gbh_resolve_kernel(int32_t *__restrict__ out, int64_t n, const uint64_t *__restrict__ words)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        int32_t o = out[r];
        if (o < -1) out[r] = (int32_t)(uint32_t)words[(uint64_t)(-(o + 2))];
    }
}

// grow: re-insert every group into the new table (keys are distinct: only an empty slot is needed)
__global__ void __launch_bounds__(kBlock) gbh_rehash_kernel(const int64_t *__restrict__ raw_hash, int64_t groups, uint64_t *words, uint64_t mask)
{
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock) {
        const uint64_t m = tg_fmix64((uint64_t)raw_hash[g]);
        uint64_t pos = m & mask;
        const uint64_t w = make_old((uint32_t)(m >> 48), (uint32_t)g);
        while (atomicCAS((unsigned long long *)&words[pos], (unsigned long long)kEmpty, (unsigned long long)w) != kEmpty) pos = (pos + 1) & mask;
    }
}

int grid_for(Context *ctx, int64_t n)
{
    int64_t blocks = ceil_div(n, kBlock);
    int64_t cap = (int64_t)ctx->cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

// fastutil HashCommon.arraySize + BigintGroupByHash.calculateMaxFill, to report the Java table's capacity
int32_t java_array_size(int32_t expected)
{
    float q = (float)expected / 0.75f;
    int64_t c = (int64_t)std::ceil((double)q);
    int64_t s = 1;
    while (s < c) s <<= 1;
    if (s < 2) s = 2;
    return (int32_t)s;
}
int32_t java_max_fill(int32_t hash_size)
{
    int32_t mf = (int32_t)std::ceil((double)((float)hash_size * 0.75f));
    if (mf == hash_size) mf--;
    return mf;
}

}  // namespace

GroupByHashGpu::GroupByHashGpu(Context *ctx, std::vector<int32_t> types, bool has_input_hash, int32_t expected_size, bool allow_integer_table)
    : ctx_(ctx), types_(std::move(types)), has_input_hash_(has_input_hash)
{
    TG_CHECK_ARG(!types_.empty() && (int)types_.size() <= kMaxKeyChannels, "group by needs 1..8 key channels");
    TG_CHECK_ARG(expected_size > 0, "expectedSize must be greater than zero");
    for (int32_t t : types_) TG_CHECK_ARG(valid_type(t), "unknown group-by key type");
    java_capacity_ = java_array_size(expected_size);
    java_max_fill_ = java_max_fill(java_capacity_);
    store_.resize(types_.size());
    for (size_t i = 0; i < types_.size(); i++) store_[i].type = types_[i];
    const char *env = getenv("TGPU_GBH_SUBBATCH");
    sub_batch_ = env ? atoll(env) : (1ll << 24);
    if (sub_batch_ < 1) sub_batch_ = 1;
    // first sub-batch: small when few groups are expected (see get_group_ids), a full one when the planner expects many
    const char *first = getenv("TGPU_GBH_FIRST_SUB");
    next_sub_ = std::min<int64_t>(sub_batch_, std::max<int64_t>(first ? atoll(first) : (1ll << 14), (int64_t)expected_size * 16));
    counters_ = ctx_->alloc_zero((size_t)GroupByHashGpu::kCounterSets * 8 * 8);
    // GroupByHash.createGroupByHash (M/operator/GroupByHash.java:45-59): one BIGINT key -> BigintGroupByHash, whose output hash is
    // recomputed from the value (BigintGroupByHash.java:150-157) -- a precomputed hash channel is not even read.  INTEGER / DATE keys
    // take the same table when no hash channel is given (their raw hash is then a function of the value, too).
    const char *off = getenv("TGPU_GBH_INTEGER_TABLE");
    if (allow_integer_table && types_.size() == 1 && !(off && off[0] == '0') &&
        (types_[0] == TGPU_BIGINT || ((types_[0] == TGPU_INTEGER || types_[0] == TGPU_DATE) && !has_input_hash_)))
        integer_ = std::make_unique<BigintGroupTable>(ctx_, types_[0]);
}

int64_t GroupByHashGpu::estimated_size() const
{
    if (integer_) return integer_->estimated_size();
    int64_t s = capacity_ * 8 + raw_hash_cap_ * 8;
    for (auto &k : store_) s += k.cap * (type_width(k.type) + 1) + (k.type == TGPU_VARCHAR ? k.cap * 4 + k.pool_cap : 0);
    return s;
}

void GroupByHashGpu::advance_java_capacity()
{
    // BigintGroupByHash.java:256-259,262-318: after adding a group, rehash (x2) while nextGroupId >= maxFill
    while (groups_ >= java_max_fill_) {
        if ((int64_t)java_capacity_ * 2 > 0x7fffffffLL)
            fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "Size of hash table cannot exceed 1 billion entries");
        java_capacity_ *= 2;
        java_max_fill_ = java_max_fill(java_capacity_);
        java_rehashes_++;
    }
}

void GroupByHashGpu::ensure_table(int64_t need_groups)
{
    int64_t want = 1024;
    while ((double)want * 0.75 < (double)need_groups + 1) want <<= 1;
    if (want <= capacity_) return;
    if (want > (1ll << 31)) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "Size of hash table cannot exceed 1 billion entries");
    BufferPtr nw = ctx_->alloc((size_t)want * 8);
    k::fill_u64(ctx_, nw->as<uint64_t>(), kEmpty, want);
    if (groups_ > 0) {
        ProfileScope ps(ctx_, "gbh_rehash");
        gbh_rehash_kernel<<<grid_for(ctx_, groups_), kBlock, 0, ctx_->stream()>>>(raw_hash_->as<int64_t>(), groups_, nw->as<uint64_t>(), (uint64_t)want - 1);
        check_launch("gbh_rehash");
    }
    words_ = nw;
    capacity_ = want;
}

void GroupByHashGpu::ensure_store(int64_t need)
{
    if (need > raw_hash_cap_) {
        int64_t cap = raw_hash_cap_ ? raw_hash_cap_ : 1024;
        while (cap < need) cap <<= 1;
        BufferPtr nb = ctx_->alloc((size_t)cap * 8);
        if (groups_) HIP_CHECK(hipMemcpyAsync(nb->ptr(), raw_hash_->ptr(), (size_t)groups_ * 8, hipMemcpyDeviceToDevice, ctx_->stream()));
        raw_hash_ = nb;
        raw_hash_cap_ = cap;
    }
    for (auto &ks : store_) {
        if (need <= ks.cap) continue;
        int64_t cap = ks.cap ? ks.cap : 1024;
        while (cap < need) cap <<= 1;
        BufferPtr nn = ctx_->alloc((size_t)cap);
        if (groups_) HIP_CHECK(hipMemcpyAsync(nn->ptr(), ks.nulls->ptr(), (size_t)groups_, hipMemcpyDeviceToDevice, ctx_->stream()));
        ks.nulls = nn;
        if (ks.type == TGPU_VARCHAR) {
            BufferPtr no = ctx_->alloc((size_t)(cap + 1) * 4);
            if (ks.offsets) HIP_CHECK(hipMemcpyAsync(no->ptr(), ks.offsets->ptr(), (size_t)(groups_ + 1) * 4, hipMemcpyDeviceToDevice, ctx_->stream()));
            else HIP_CHECK(hipMemsetAsync(no->ptr(), 0, 4, ctx_->stream()));
            ks.offsets = no;
            if (!ks.values) {
                ks.pool_cap = 4096;
                ks.values = ctx_->alloc((size_t)ks.pool_cap);
            }
        }
        else {
            const int w = type_width(ks.type);
            BufferPtr nv = ctx_->alloc((size_t)cap * w);
            if (groups_) HIP_CHECK(hipMemcpyAsync(nv->ptr(), ks.values->ptr(), (size_t)groups_ * w, hipMemcpyDeviceToDevice, ctx_->stream()));
            ks.values = nv;
        }
        ks.cap = cap;
    }
}

void GroupByHashGpu::ensure_pool(KeyStore &ks, int64_t need_bytes)
{
    if (need_bytes <= ks.pool_cap) return;
    if (need_bytes > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "group-by key store cannot exceed 2GB of variable width data");
    int64_t cap = ks.pool_cap ? ks.pool_cap : 4096;
    while (cap < need_bytes) cap <<= 1;
    BufferPtr nv = ctx_->alloc((size_t)cap);
    if (ks.pool_used) HIP_CHECK(hipMemcpyAsync(nv->ptr(), ks.values->ptr(), (size_t)ks.pool_used, hipMemcpyDeviceToDevice, ctx_->stream()));
    ks.values = nv;
    ks.pool_cap = cap;
}

// a key set in device memory, for the kernels that index its columns at run time
BufferPtr GroupByHashGpu::device_keys(const KeyCols &k)
{
    BufferPtr b = ctx_->alloc(sizeof(KeyCols));
    ctx_->upload(b->ptr(), &k, sizeof(KeyCols));
    return b;
}

KeyCols GroupByHashGpu::store_view() const
{
    KeyCols k{};
    k.n = (int32_t)store_.size();
    for (size_t i = 0; i < store_.size(); i++) {
        const KeyStore &s = store_[i];
        k.c[i] = ColView{s.values ? s.values->ptr() : nullptr, s.nulls ? s.nulls->as<uint8_t>() : nullptr,
                         s.offsets ? s.offsets->as<int32_t>() : nullptr, s.type, 0};
    }
    return k;
}

// the eight counter words of a sub-batch, the last one being an error word whose "none" is ~0: a ring of kCounterSets sets is
// initialised half by half by ONE launch per half and handed out set by set (a launch per sub-batch was a tenth of a 2^20-row page's
// device time).  Half by half because the set handed out LAST may still be read by the next launch (a one-pass aggregation launch gates
// the commit of its predecessor's totals on the predecessor's counters): the launch that re-initialises a half never touches the other.
static __global__ void init_counters_kernel(unsigned long long *ctr, int words)
{
    for (int i = threadIdx.x; i < words; i += blockDim.x) ctr[i] = (i & 7) == 7 ? ~0ull : 0ull;
}

unsigned long long *GroupByHashGpu::fresh_counters()
{
    constexpr int half = kCounterSets / 2;
    if (next_counter_set_ % half == 0) {
        if (next_counter_set_ == kCounterSets) next_counter_set_ = 0;
        // every earlier user of this half is in front of this launch on the stream
        init_counters_kernel<<<1, 256, 0, ctx_->stream()>>>(counters_->as<unsigned long long>() + 8 * next_counter_set_, half * 8);
        check_launch("init_counters");
    }
    return counters_->as<unsigned long long>() + 8 * (next_counter_set_++);
}

bool GroupByHashGpu::process_sub_batch(const KeyCols &batch, const int64_t *hashes, const uint8_t *row_mask, int64_t row0, int64_t n, int32_t *out,
                                       const GbhProbeFn *probe, int64_t *new_groups_out)
{
    *new_groups_out = 0;
    // room for every row of a normal sub-batch to be a new group; larger (optimistic) sub-batches rely on overflow detection
    ensure_table(groups_ + std::min<int64_t>(n, sub_batch_));
    ensure_store(groups_ > 0 ? groups_ : 1);  // the store view must be addressable for OLD slots
    unsigned long long *ctr = fresh_counters();   // [0..6] = 0, [7] (expression-error word of a fused probe kernel) = ~0
    const int g = grid_for(ctx_, n);
    if (probe) {
        GbhProbeLaunch l{row0, n, words_->as<uint64_t>(), (uint64_t)capacity_ - 1, store_view(), (int32_t)std::min<int64_t>(groups_, 1 << 20), out, ctr};
        (*probe)(l);
    }
    else {
        ProfileScope ps(ctx_, "gbh_insert");
        BufferPtr dbatch = device_keys(batch), dstore = device_keys(store_view());
        gbh_probe_kernel<true><<<g, kBlock, 0, ctx_->stream()>>>(dbatch->as<KeyCols>(), hashes, row_mask, n, words_->as<uint64_t>(), (uint64_t)capacity_ - 1,
                                                                 dstore->as<KeyCols>(), (int32_t)std::min<int64_t>(groups_, 1 << 20), out, ctr);
        check_launch("gbh_insert");
    }
    // A smallish sub-batch marks and ranks its first-occurrence rows right away, so that ONE read-back brings the pending count,
    // the number of new groups and the error flag; a big one (the steady state of low-cardinality inputs: no pending rows) first
    // looks at the counters and only then pays for the two passes over its rows.
    BufferPtr flags, rank;
    auto mark_and_rank = [&] {
        flags = ctx_->alloc((size_t)n * 4);
        rank = ctx_->alloc((size_t)n * 4);
        ProfileScope ps(ctx_, "gbh_assign");
        gbh_mark_kernel<<<g, kBlock, 0, ctx_->stream()>>>(out, n, words_->as<uint64_t>(), flags->as<int32_t>());
        k::exclusive_scan_i32(ctx_, flags->as<int32_t>(), rank->as<int32_t>(), n, (int64_t *)&ctr[1]);
    };
    const bool eager = n <= (1ll << 22) && (groups_ == 0 || last_new_groups_ > 0);   // new groups are likely: worth the two passes up front
    if (eager) mark_and_rank();
    unsigned long long host_ctr[8];
    ctx_->download(host_ctr, ctr, sizeof(host_ctr));
    raise_expression_error(host_ctr[7]);
    if (host_ctr[2] != 0) return false;  // table overflow
    if (host_ctr[0] == 0) {              // every row hit an existing group
        last_new_groups_ = 0;
        return true;
    }
    int64_t new_groups = (int64_t)host_ctr[1];
    if (!eager) {
        mark_and_rank();
        new_groups = (int64_t)ctx_->read_scalar((const unsigned long long *)&ctr[1]);
    }
    TG_CHECK_STATE(new_groups > 0, "pending rows without new groups");
    ensure_store(groups_ + new_groups);

    VarLens lens{};
    std::vector<BufferPtr> len_bufs(store_.size()), off_bufs(store_.size());
    for (size_t c = 0; c < store_.size(); c++) {
        if (store_[c].type == TGPU_VARCHAR) {
            len_bufs[c] = ctx_->alloc((size_t)new_groups * 4);
            off_bufs[c] = ctx_->alloc((size_t)new_groups * 4);
            lens.len[c] = len_bufs[c]->as<int32_t>();
        }
    }
    {
        ProfileScope ps(ctx_, "gbh_finalize");
        BufferPtr dbatch = device_keys(batch), dstore = device_keys(store_view());   // the store may have been re-allocated by ensure_store
        gbh_finalize_kernel<<<g, kBlock, 0, ctx_->stream()>>>(dbatch->as<KeyCols>(), hashes, n, words_->as<uint64_t>(), out, flags->as<int32_t>(), rank->as<int32_t>(),
                                                             groups_, dstore->as<KeyCols>(), raw_hash_->as<int64_t>(), lens);
        check_launch("gbh_finalize");
        for (size_t c = 0; c < store_.size(); c++) {
            KeyStore &ks = store_[c];
            if (ks.type != TGPU_VARCHAR) continue;
            k::exclusive_scan_i32(ctx_, len_bufs[c]->as<int32_t>(), off_bufs[c]->as<int32_t>(), new_groups, (int64_t *)&ctr[3]);
            const int64_t add_bytes = (int64_t)ctx_->read_scalar((const unsigned long long *)&ctr[3]);
            ensure_pool(ks, ks.pool_used + add_bytes);
            gbh_copy_varchar_kernel<<<g, kBlock, 0, ctx_->stream()>>>(batch.c[c], n, flags->as<int32_t>(), rank->as<int32_t>(), groups_,
                                                                     off_bufs[c]->as<int32_t>(), ks.pool_used, ks.values->as<uint8_t>(), ks.offsets->as<int32_t>());
            check_launch("gbh_copy_varchar");
            ks.pool_used += add_bytes;
        }
        gbh_resolve_kernel<<<g, kBlock, 0, ctx_->stream()>>>(out, n, words_->as<uint64_t>());
        check_launch("gbh_resolve");
    }
    groups_ += new_groups;
    last_new_groups_ = new_groups;
    *new_groups_out = new_groups;
    advance_java_capacity();
    return true;
}

// drops every NEW mark of an aborted sub-batch by re-inserting the known groups into a fresh, larger table
void GroupByHashGpu::rebuild_table(int64_t min_capacity)
{
    int64_t want = capacity_ > 0 ? capacity_ : 1024;
    while (want < min_capacity) want <<= 1;
    if (want > (1ll << 31)) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "Size of hash table cannot exceed 1 billion entries");
    BufferPtr nw = ctx_->alloc((size_t)want * 8);
    k::fill_u64(ctx_, nw->as<uint64_t>(), kEmpty, want);
    if (groups_ > 0) {
        gbh_rehash_kernel<<<grid_for(ctx_, groups_), kBlock, 0, ctx_->stream()>>>(raw_hash_->as<int64_t>(), groups_, nw->as<uint64_t>(), (uint64_t)want - 1);
        check_launch("gbh_rehash");
    }
    words_ = nw;
    capacity_ = want;
}

bool GroupByHashGpu::get_group_ids(const std::vector<const DeviceColumn *> &keys, const int64_t *hashes, int64_t n, int32_t *out_gids,
                                   const uint8_t *row_mask, bool inline_hash, const GbhProbeFn *probe, uint8_t *out_gids8, const GbhSpeculateFn *speculate,
                                   bool *speculated)
{
    if (speculated) *speculated = false;
    TG_CHECK_ARG(keys.size() == types_.size(), "wrong number of key channels");
    for (size_t i = 0; i < keys.size(); i++) TG_CHECK_ARG(keys[i]->type == types_[i], "group-by key channel type mismatch");
    if (n <= 0) return false;
    if (integer_) {
        TG_CHECK_STATE(probe == nullptr && out_gids8 == nullptr, "external probe kernels need the generic group-by table");
        get_group_ids_integer(*keys[0], n, out_gids, row_mask);
        return false;
    }
    constexpr int64_t kCompactGroups = 250;   // ids + 1 must stay below the 255 marker
    bool compact = out_gids8 != nullptr && probe != nullptr && groups_ < kCompactGroups;
    if (out_gids8 != nullptr && !compact) {   // no external probe kernel, or already too many groups for a byte
        (void)get_group_ids(keys, hashes, n, out_gids, row_mask, inline_hash, probe, nullptr);
        return false;
    }
    BufferPtr own_hashes;
    if (!hashes && !inline_hash) {
        own_hashes = ctx_->alloc((size_t)n * 8);
        k::hash_rows(ctx_, key_cols_of(keys), n, own_hashes->as<int64_t>());
        hashes = own_hashes->as<int64_t>();
    }
    // Sub-batching bounds the table growth a single launch can need.  Once a sub-batch created no new group (the steady state
    // of low-cardinality inputs: TPCH Q1 has 4 groups in 600 M rows) the next one is 64x larger: fewer, longer launches.  Such an
    // optimistic launch can overflow the table only if it meets tens of millions of new keys; the probe kernel then flags it,
    // the table is rebuilt twice as large and the rows are re-run in smaller pieces.
    // With few expected groups the very first launches ramp up from a small piece (2^14 rows): while the table is empty every
    // row takes the insert path and, with few distinct keys, they all contend for the same slots; once the first groups exist
    // the probe kernels answer from their cached copies.  next_sub_ persists across pages.
    int64_t sub = next_sub_;
    int64_t start = 0;
    while (start < n) {
        const int64_t len = std::min(sub, n - start);
        std::vector<DeviceColumn> views;
        views.reserve(keys.size());
        for (auto *c : keys) views.push_back(k::region_of(ctx_, *c, start, len));
        std::vector<const DeviceColumn *> vp;
        for (auto &v : views) vp.push_back(&v);
        int64_t new_groups = 0;
        bool ok = true;
        if (compact && groups_ == 0) {
            // nothing to look up yet: every group of this sub-batch is new, so the compact attempt would only be repeated
            BufferPtr tmp = ctx_->alloc((size_t)len * 4);
            ok = process_sub_batch(key_cols_of(vp), nullptr, nullptr, start, len, tmp->as<int32_t>(), probe, &new_groups);
            if (ok) {
                if (groups_ < kCompactGroups) {
                    gbh_narrow_kernel<<<grid_for(ctx_, len), kBlock, 0, ctx_->stream()>>>(tmp->as<int32_t>(), len, out_gids8 + start);
                    check_launch("gbh_narrow");
                }
                else compact = false;
            }
        }
        else if (compact) {
            // compact attempt: the probe kernel answers with one byte per row; rows that meet a group that is new in this
            // sub-batch are only counted
            ensure_table(groups_ + std::min<int64_t>(len, sub_batch_));
            ensure_store(groups_ > 0 ? groups_ : 1);
            unsigned long long *ctr = fresh_counters();
            GbhProbeLaunch l{start, len, words_->as<uint64_t>(), (uint64_t)capacity_ - 1, store_view(), (int32_t)std::min<int64_t>(groups_, 1 << 20), nullptr, ctr,
                             out_gids8 + start};
            (*probe)(l);
            unsigned long long host_ctr[8];
            const Context::AsyncRead rd = ctx_->begin_read(ctr, sizeof(host_ctr));
            const bool hooked = speculate != nullptr && start == 0 && len == n;
            try {
                if (hooked) (*speculate)(ctr);   // the page's consumer, gated on these counters: runs while the host waits for them
            }
            catch (...) {
                ctx_->finish_read(rd, host_ctr);   // gives the read slot back
                throw;
            }
            ctx_->finish_read(rd, host_ctr);
            if (hooked && speculated) *speculated = host_ctr[0] == 0 && host_ctr[2] == 0 && host_ctr[7] == ~0ull;
            raise_expression_error(host_ctr[7]);
            ok = host_ctr[2] == 0;
            if (ok && host_ctr[0] != 0) {
                // new groups: the full protocol on a temporary int32 buffer (the re-run finds the slots it has just claimed), then
                // the final ids are narrowed
                BufferPtr tmp = ctx_->alloc((size_t)len * 4);
                ok = process_sub_batch(key_cols_of(vp), nullptr, nullptr, start, len, tmp->as<int32_t>(), probe, &new_groups);
                if (ok) {
                    if (groups_ < kCompactGroups) {
                        gbh_narrow_kernel<<<grid_for(ctx_, len), kBlock, 0, ctx_->stream()>>>(tmp->as<int32_t>(), len, out_gids8 + start);
                        check_launch("gbh_narrow");
                    }
                    else compact = false;   // too many groups for a byte: the page is redone in int32 below
                }
            }
            else if (ok) last_new_groups_ = 0;
        }
        else
            ok = process_sub_batch(key_cols_of(vp), hashes ? hashes + start : nullptr, row_mask ? row_mask + start : nullptr, start, len, out_gids + start, probe,
                                   &new_groups);
        if (!ok) {
            rebuild_table(capacity_ * 2);
            sub = std::max<int64_t>(std::min(sub_batch_, len / 4), 1);
            continue;
        }
        start += len;
        if (new_groups == 0) sub = sub >= (1ll << 17) ? (1ll << 30) : sub * 64;   // a sizeable piece without a new group: take the rest in one launch
        else sub = sub < sub_batch_ ? std::min<int64_t>(sub * 8, sub_batch_) : sub_batch_;
        next_sub_ = sub;
        if (out_gids8 != nullptr && !compact) break;   // compact mode was abandoned: see below
    }
    if (out_gids8 != nullptr && !compact) {
        // int32 ids for the whole page: every group met so far exists, so the rows before `start` are plain lookups
        (void)get_group_ids(keys, hashes, n, out_gids, row_mask, inline_hash, probe, nullptr);
        return false;
    }
    return out_gids8 != nullptr;
}

// Sub-batches of the integer table.  A regular sub-batch is sized by bound (room for every row to be a new group: it cannot overflow);
// after a sizeable one without a new group the rest of the page goes in ONE optimistic launch with room for 2^24 new groups (an
// overflow is flagged by the kernel: rebuilt larger, re-run in pieces).  A sub-batch in which at least a quarter of the rows were new
// groups is evidence of a high-cardinality key: the table is then sized ONCE from the row bound of the rest of the page instead of
// growing by doubling (12 rehashes on the way to 40 M groups), and the rest goes in sub-batches of 2^24 rows -- rows that
// meet a group published by an earlier sub-batch are answered by the insert pass alone, only repeats of a key INSIDE its first
// sub-batch need the resolve pass (one page in one piece: 63 M of 100 M rows; in six pieces: 9 M).
void GroupByHashGpu::get_group_ids_integer(const DeviceColumn &key, int64_t n, int32_t *out_gids, const uint8_t *row_mask)
{
    int64_t sub = next_sub_, start = 0;
    while (start < n) {
        const int64_t len = std::min(sub, n - start);
        const DeviceColumn view = k::region_of(ctx_, key, start, len);
        int64_t room = optimistic_ ? std::min<int64_t>(len, sub_batch_) : len;
        if (high_cardinality_) room = std::max(room, std::min<int64_t>(n - start, 1ll << 27));
        integer_->ensure_table(integer_->groups() + room);
        int64_t new_groups = 0;
        if (!integer_->process(view, row_mask ? row_mask + start : nullptr, len, out_gids + start, fresh_counters(), &new_groups)) {
            integer_->rebuild(integer_->capacity() * 2);
            sub = std::max<int64_t>(std::min(sub_batch_, len / 4), 1);
            optimistic_ = false;
            continue;
        }
        start += len;
        groups_ = integer_->groups();
        if (new_groups > 0) advance_java_capacity();
        last_new_groups_ = new_groups;
        if (new_groups == 0) {
            optimistic_ = sub >= std::min<int64_t>(1ll << 17, sub_batch_);
            sub = optimistic_ ? (1ll << 40) : sub * 64;
            high_cardinality_ = false;
        }
        else {
            optimistic_ = false;
            high_cardinality_ = new_groups * 4 >= len;
            sub = high_cardinality_ ? sub_batch_ : std::min<int64_t>(sub * 8, sub_batch_);
        }
        next_sub_ = sub;
    }
}

void GroupByHashGpu::lookup(const std::vector<const DeviceColumn *> &keys, const int64_t *hashes, int64_t n, int32_t *out_gids)
{
    TG_CHECK_ARG(keys.size() == types_.size(), "wrong number of key channels");
    if (n <= 0) return;
    if (integer_) {
        integer_->lookup(*keys[0], n, out_gids, fresh_counters());
        return;
    }
    BufferPtr own_hashes;
    if (!hashes) {
        own_hashes = ctx_->alloc((size_t)n * 8);
        k::hash_rows(ctx_, key_cols_of(keys), n, own_hashes->as<int64_t>());
        hashes = own_hashes->as<int64_t>();
    }
    ensure_table(groups_);
    ensure_store(groups_ > 0 ? groups_ : 1);
    BufferPtr dbatch = device_keys(key_cols_of(keys)), dstore = device_keys(store_view());
    gbh_probe_kernel<false><<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(dbatch->as<KeyCols>(), hashes, nullptr, n, words_->as<uint64_t>(), (uint64_t)capacity_ - 1,
                                                                              dstore->as<KeyCols>(), (int32_t)std::min<int64_t>(groups_, 1 << 20), out_gids,
                                                                              fresh_counters());
    check_launch("gbh_lookup");
}

DevicePage GroupByHashGpu::key_page(bool with_hash)
{
    if (integer_) {
        DevicePage p;
        p.n = groups_;
        p.cols.push_back(integer_->key_column());
        if (with_hash) {
            // BigintGroupByHash.appendValuesTo (:150-157): the hash of the value, NULL_HASH_CODE (0) for the null group = H5 over one channel
            DeviceColumn c;
            c.type = TGPU_BIGINT;
            c.n = groups_;
            c.values_buf = ctx_->alloc((size_t)(groups_ > 0 ? groups_ : 1) * 8);
            c.values = c.values_buf->ptr();
            std::vector<const DeviceColumn *> kc{&p.cols[0]};
            k::hash_rows(ctx_, key_cols_of(kc), groups_, c.values_buf->as<int64_t>());
            p.cols.push_back(c);
        }
        return p;
    }
    DevicePage p;
    p.n = groups_;
    ensure_store(groups_ > 0 ? groups_ : 1);
    for (auto &ks : store_) {
        DeviceColumn c;
        c.type = ks.type;
        c.n = groups_;
        c.values_buf = ks.values;
        c.values = ks.values->ptr();
        c.nulls_buf = ks.nulls;
        c.nulls = ks.nulls->as<uint8_t>();
        if (ks.type == TGPU_VARCHAR) {
            c.offsets_buf = ks.offsets;
            c.offsets = ks.offsets->as<int32_t>();
            c.pool_bytes = ks.pool_used;
        }
        p.cols.push_back(c);
    }
    if (with_hash) {
        DeviceColumn c;
        c.type = TGPU_BIGINT;
        c.n = groups_;
        c.values_buf = raw_hash_;
        c.values = raw_hash_->ptr();
        p.cols.push_back(c);
    }
    return p;
}

}  // namespace tgpu
