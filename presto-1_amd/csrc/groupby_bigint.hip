// groupby_bigint.hip -- the single-integer-key GroupByHash (M/operator/BigintGroupByHash.java; chosen like GroupByHash.java:45-59 does).
//
// The reference keeps values[cap] + groupIds[cap] (BigintGroupByHash.java:60-67): one random cache line per probe.  So does this
// table: a 16-byte slot {int64 key; uint32 minrow; uint32 gid} holds the key INLINE (the generic table of groupby.hip keeps a tag
// word per slot and the keys in a store by group id: two random lines per row).  What the design is priced with, measured on
// this part (tools/exp_random_access.hip): a random line costs the same from the Infinity Cache and from HBM
// (55 G loads/s), random 4-byte stores into an HBM-sized region reach 22 G/s, and global atomics top out at 27 G/s even when
// they hit in L2 -- so the protocol spends ONE atomic per new group (the claiming CAS), none per row:
//
//   insert  (gbi_insert) : a row loads its slot (one 16-byte load).  Key there with a final gid -> done.  Key there, group still
//             pending in this sub-batch -> the row only lowers `minrow` (atomicMin) when it is SMALLER than the value it
//             loaded (values only fall, so a stale read errs on the safe side).  Empty -> CAS on the (minrow, gid) word from
//             EMPTY to (row, NONE), then a plain store of the key; a reader that meets a claimed slot whose key is still
//             the fill pattern re-reads it coherently (the claimer's store is on its way; the key that EQUALS the fill pattern
//             and the NULL key have slots of their own behind the table, so "fill pattern" always means "not written yet").
//             Rows that claimed or lowered a slot are *candidates* for being their key's first row (one bit per row, by ballot).
//             A row that lowers a slot tells the row it displaced (the value its atomicMin returned) that it is not the first: a bit
//             in `not_first`.  Lowering is rare -- it takes a row overtaken by a later row of its key.
//   mark    (gbi_mark)   : first occurrence = candidate & ~not_first, a streaming pass over two bit vectors (no table access); a
//             count per 1024-row tile; an exclusive scan of the tile counts gives the first-seen rank = BigintGroupByHash's
//             `nextGroupId++` order (BigintGroupByHash.java:246-253).
//   publish (gbi_publish): first rows store gid = groups + rank into their slot and the key into values_by_group[gid].
//   resolve (gbi_resolve): the other pending rows read their slot's gid.
// In the steady state (no row meets a new key) a page is ONE launch and one read-back.
#include "groupby.h"
#include "kernels.h"
#include "device_hash.h"

#include <algorithm>

namespace tgpu {

namespace {

constexpr int kBlock = 256;
constexpr int kRowsPerLane = 4;
constexpr int kTile = kBlock * kRowsPerLane;   // rows per workgroup iteration; also the granularity of the first-row counts
constexpr unsigned int kNone = 0xffffffffu;    // gid of a slot whose group is pending in the running sub-batch
constexpr unsigned long long kEmptyWord = ~0ull;
constexpr long long kFillKey = -1;             // key word of a slot nobody has written yet (the table is filled with 0xff bytes)
constexpr unsigned int kPending = 0x80000000u;
constexpr unsigned int kMasked = 0xffffffffu;  // row excluded by the row mask (= -1 as int32); never a valid pending code

struct alignas(16) Slot {
    long long key;
    unsigned int minrow;
    unsigned int gid;
};

__device__ __forceinline__ unsigned long long word_of(unsigned int minrow, unsigned int gid) { return ((unsigned long long)gid << 32) | minrow; }

// slot of a key: top bits of key * 2^64/phi (Fibonacci hashing; the layout is not observable, see groupby.hip)
__device__ __forceinline__ unsigned int slot_of(long long key, int shift) { return (unsigned int)(((unsigned long long)key * 0x9E3779B97F4A7C15ull) >> shift); }

template <typename T> __device__ __forceinline__ T coherent_load(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One row against the table.  Returns the final gid (< 2^31), or kPending | slot; `cand` = this row claimed / lowered the slot.
// `first` is the 16-byte snapshot of the row's home slot loaded by the caller (plain load).
template <bool INSERT>
__device__ __forceinline__ unsigned int probe_row(Slot *slots, unsigned int mask, unsigned int cap, long long key, bool is_null, unsigned int s, long long ks,
                                                  unsigned long long w, unsigned int r, bool &cand, unsigned long long *counters, unsigned long long *not_first)
{
    cand = false;
    const bool special = is_null || key == kFillKey;   // slots of their own: cap (NULL group), cap + 1 (the key that equals the fill pattern)
    for (unsigned int iter = 0;; iter++) {
        // a probe sequence this long means an optimistic sub-batch has flooded the table: flag it and let every lane bail out quickly -- the
        // host rebuilds a larger table and re-runs the rows (a table sized by bound, fill <= 0.75, never gets here)
        if ((iter & 63) == 63 && (iter >= 8192 || coherent_load(&counters[2]) != 0)) break;
        if (w == kEmptyWord) {
            if (!INSERT) return kMasked;
            const unsigned long long old = atomicCAS((unsigned long long *)&slots[s].minrow, kEmptyWord, word_of(r, kNone));
            if (old == kEmptyWord) {
                if (!special) __hip_atomic_store(&slots[s].key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cand = true;
                return kPending | s;
            }
            w = old;
            ks = special ? key : coherent_load(&slots[s].key);
        }
        if (!special && ks == kFillKey) {
            // claimed, key not visible yet: the claimer's store is in flight
            ks = coherent_load(&slots[s].key);
            w = coherent_load((const unsigned long long *)&slots[s].minrow);
            continue;
        }
        if (special || ks == key) {
            const unsigned int gid = (unsigned int)(w >> 32);
            if (gid != kNone) return gid;
            if (!INSERT) return kMasked;   // (lookups run between sub-batches: nothing is pending then)
            if (r < (unsigned int)w) {
                // lowering is rare (a row overtaken by a later one of its key): the row it displaces is told that it is not the first
                const unsigned int displaced = __hip_atomic_fetch_min(&slots[s].minrow, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (displaced > r) {
                    cand = true;
                    atomicOr(&not_first[displaced >> 6], 1ull << (displaced & 63));
                }
            }
            return kPending | s;
        }
        s = (s + 1) & mask;
        const ulonglong2 v = *(const ulonglong2 *)&slots[s];
        ks = (long long)v.x;
        w = v.y;
    }
    atomicExch(&counters[2], 1ULL);
    return kMasked;
}

// counters: [0] pending rows, [1] new groups (written by the scan), [2] table-overflow flag
template <typename KT, bool INSERT>
__global__ void __launch_bounds__(kBlock) gbi_insert_kernel(const KT *__restrict__ values, const uint8_t *__restrict__ nulls, const uint8_t *__restrict__ row_mask, int64_t n,
                                                             Slot *slots, unsigned int cap, int shift, unsigned int *__restrict__ out,
                                                             unsigned long long *__restrict__ cand_bits, unsigned long long *not_first, unsigned long long *counters)
{
    const unsigned int mask = cap - 1;
    const int lane = threadIdx.x & 63;
    unsigned int my_pending = 0;
    for (int64_t base = (int64_t)blockIdx.x * kTile; base < n; base += (int64_t)gridDim.x * kTile) {
        long long key[kRowsPerLane];
        bool active[kRowsPerLane], isnull[kRowsPerLane];
        unsigned int s[kRowsPerLane];
        ulonglong2 snap[kRowsPerLane];
        // all row loads, then all slot loads, before any use: four independent random lines in flight per lane
#pragma unroll
        for (int u = 0; u < kRowsPerLane; u++) {
            const int64_t r = base + u * kBlock + threadIdx.x;
            active[u] = r < n && (!row_mask || row_mask[r]);
            isnull[u] = active[u] && nulls && nulls[r];
            key[u] = active[u] ? (long long)values[r] : 0;
        }
#pragma unroll
        for (int u = 0; u < kRowsPerLane; u++) {
            s[u] = isnull[u] ? cap : (key[u] == kFillKey ? cap + 1 : slot_of(key[u], shift));
            snap[u] = *(const ulonglong2 *)&slots[active[u] ? s[u] : 0];
        }
        // runs of equal keys in adjacent rows (inputs clustered by key) go to the table once per wave: the followers copy the
        // first row's answer (same group, or the same pending slot whose first row is the leader's or an earlier one)
        bool lead[kRowsPerLane], claimed[kRowsPerLane];
        unsigned long long old[kRowsPerLane];
#pragma unroll
        for (int u = 0; u < kRowsPerLane; u++) {
            const int kind = (active[u] ? 1 : 0) | (isnull[u] ? 2 : 0);
            const long long key_prev = __shfl_up(key[u], 1, 64);
            const int kind_prev = __shfl_up(kind, 1, 64);
            lead[u] = active[u] && !(lane > 0 && kind_prev == kind && key_prev == key[u]);
        }
        // the claiming CAS of every row that found its home slot empty, all of a lane's rows in flight together
#pragma unroll
        for (int u = 0; u < kRowsPerLane; u++) {
            claimed[u] = INSERT && lead[u] && snap[u].y == kEmptyWord;
            old[u] = 0;
            if (claimed[u]) old[u] = atomicCAS((unsigned long long *)&slots[s[u]].minrow, kEmptyWord, word_of((unsigned int)(base + u * kBlock + threadIdx.x), kNone));
        }
        // the winners' keys, ALL of them before any lane may wait for one: a lane that lost a slot spins below until the slot's key is
        // visible, and the winner may be another row of the same wave (a later u) -- which only gets there once the spinner has moved on
#pragma unroll
        for (int u = 0; u < kRowsPerLane; u++)
            if (claimed[u] && old[u] == kEmptyWord && !(isnull[u] || key[u] == kFillKey))
                __hip_atomic_store(&slots[s[u]].key, key[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < kRowsPerLane; u++) {
            const int64_t r = base + u * kBlock + threadIdx.x;
            const bool special = isnull[u] || key[u] == kFillKey;
            bool cand = false;
            unsigned int result = kMasked;
            if (claimed[u] && old[u] == kEmptyWord) {
                cand = true;
                result = kPending | s[u];
            }
            else if (lead[u]) {
                long long ks = (long long)snap[u].x;
                unsigned long long w = snap[u].y;
                if (claimed[u]) {   // lost the slot to another row: carry on with what is there now
                    w = old[u];
                    ks = special ? key[u] : coherent_load(&slots[s[u]].key);
                }
                result = probe_row<INSERT>(slots, mask, cap, key[u], isnull[u], s[u], ks, w, (unsigned int)r, cand, counters, not_first);
            }
            const unsigned long long leaders = __ballot(lead[u]);
            const unsigned long long at_or_below = leaders & ((2ULL << lane) - 1ULL);
            const int src = at_or_below ? 63 - __clzll((long long)at_or_below) : lane;
            const unsigned int lead_result = __shfl(result, src, 64);
            if (active[u] && !lead[u]) result = lead_result;
            if (r < n) out[r] = result;
            if (INSERT) {
                const unsigned long long cb = __ballot(cand);
                if (lane == 0 && base + u * kBlock + (threadIdx.x & ~63) < n) cand_bits[(base + u * kBlock + threadIdx.x) >> 6] = cb;
                if (result != kMasked && (result & kPending)) my_pending++;
            }
        }
    }
    if (INSERT) {
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

// first-occurrence rows of the sub-batch = candidates that nobody displaced: one thread per 64-row word, 16 words per 1024-row tile
__global__ void __launch_bounds__(kBlock) gbi_mark_kernel(int64_t words, const unsigned long long *__restrict__ cand_bits, const unsigned long long *__restrict__ not_first,
                                                           unsigned long long *__restrict__ first_bits, int32_t *__restrict__ tile_counts,
                                                           const unsigned long long *__restrict__ counters)
{
    if (counters[0] == 0) return;   // nothing pending: the host will not look at the counts
    const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    unsigned long long fb = 0;
    if (w < words) {
        fb = cand_bits[w] & ~not_first[w];
        first_bits[w] = fb;
    }
    unsigned int c = (unsigned int)__popcll(fb);
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 15) == 0 && w < words) tile_counts[w >> 4] = (int32_t)c;
}

// exclusive scan of the tile counts by ONE workgroup (a tile is 1024 rows: 600 M rows are 586 K counts): every thread sums a contiguous
// chunk, the chunk sums are scanned through LDS, the chunk is rewritten as prefixes.  counters[1] = the total = number of new groups.
constexpr int kScanThreads = 1024;
__global__ void __launch_bounds__(kScanThreads) gbi_scan_kernel(const int32_t *__restrict__ counts, int32_t *__restrict__ prefix, int64_t tiles, unsigned long long *counters)
{
    if (counters[0] == 0) return;
    const int64_t chunk = (tiles + kScanThreads - 1) / kScanThreads;
    const int64_t a = (int64_t)threadIdx.x * chunk, z = a + chunk < tiles ? a + chunk : tiles;
    long long sum = 0;
    for (int64_t i = a; i < z; i++) sum += counts[i];
    __shared__ long long s_sum[kScanThreads];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < kScanThreads; d <<= 1) {
        const long long v = threadIdx.x >= d ? s_sum[threadIdx.x - d] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    long long run = s_sum[threadIdx.x] - sum;
    for (int64_t i = a; i < z; i++) {
        prefix[i] = (int32_t)run;
        run += counts[i];
    }
    if (threadIdx.x == kScanThreads - 1) counters[1] = (unsigned long long)s_sum[kScanThreads - 1];
}

template <typename KT>
__global__ void __launch_bounds__(kBlock) gbi_publish_kernel(const KT *__restrict__ values, const uint8_t *__restrict__ nulls, unsigned int *__restrict__ out, int64_t n,
                                                              Slot *slots, const unsigned long long *__restrict__ first_bits, const int32_t *__restrict__ tile_prefix,
                                                              int64_t base_gid, KT *__restrict__ values_by_group, uint8_t *__restrict__ nulls_by_group)
{
    const int lane = threadIdx.x & 63;
    const int64_t base = (int64_t)blockIdx.x * kTile;
    // popcounts of the tile's 16 first-row words, in row order: word (u, wave) covers rows base + u * 256 + wave * 64 ..
    __shared__ unsigned int s_pop[kTile / 64];
    if (threadIdx.x < kTile / 64) {
        const int64_t row0 = base + (int64_t)threadIdx.x * 64;
        s_pop[threadIdx.x] = row0 < n ? (unsigned int)__popcll(first_bits[row0 >> 6]) : 0u;
    }
    __syncthreads();
    const int64_t tile_base = base_gid + tile_prefix[blockIdx.x];
#pragma unroll
    for (int u = 0; u < kRowsPerLane; u++) {
        const int64_t r = base + u * kBlock + threadIdx.x;
        const int word_in_tile = u * (kBlock / 64) + (threadIdx.x >> 6);
        if (base + (int64_t)word_in_tile * 64 >= n) continue;
        const unsigned long long fb = first_bits[(base >> 6) + word_in_tile];
        if (!((fb >> lane) & 1ull)) continue;
        unsigned int before = 0;
        for (int w = 0; w < word_in_tile; w++) before += s_pop[w];
        const int64_t gid = tile_base + before + __popcll(fb & ((1ull << lane) - 1ull));
        const unsigned int s = out[r] & ~kPending;
        slots[s].gid = (unsigned int)gid;
        const bool isnull = nulls && nulls[r];
        values_by_group[gid] = isnull ? (KT)0 : values[r];
        nulls_by_group[gid] = isnull ? 1 : 0;
        out[r] = (unsigned int)gid;
    }
}

__global__ void __launch_bounds__(kBlock) gbi_resolve_kernel(unsigned int *__restrict__ out, int64_t n, const Slot *__restrict__ slots)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        const unsigned int o = out[r];
        if (o != kMasked && (o & kPending)) out[r] = slots[o & ~kPending].gid;
    }
}

// grow: every group into the new table (keys are distinct: an empty slot is all a group needs)
template <typename KT>
__global__ void __launch_bounds__(kBlock) gbi_rehash_kernel(const KT *__restrict__ values_by_group, const uint8_t *__restrict__ nulls_by_group, int64_t groups, Slot *slots,
                                                             unsigned int cap, int shift)
{
    const unsigned int mask = cap - 1;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock) {
        const long long key = (long long)values_by_group[g];
        const unsigned long long w = word_of(0, (unsigned int)g);
        if (nulls_by_group[g]) {
            *(unsigned long long *)&slots[cap].minrow = w;
            continue;
        }
        if (key == kFillKey) {
            *(unsigned long long *)&slots[cap + 1].minrow = w;
            continue;
        }
        unsigned int s = slot_of(key, shift);
        while (atomicCAS((unsigned long long *)&slots[s].minrow, kEmptyWord, w) != kEmptyWord) s = (s + 1) & mask;
        slots[s].key = key;
    }
}

int grid_for(Context *ctx, int64_t n, int per_block)
{
    int64_t blocks = ceil_div(n, per_block);
    const int64_t cap = (int64_t)ctx->cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

BigintGroupTable::BigintGroupTable(Context *ctx, int32_t type) : ctx_(ctx), type_(type), width_(type_width(type)) {}

int64_t BigintGroupTable::estimated_size() const { return (capacity_ ? (capacity_ + 2) * 16 : 0) + store_cap_ * (width_ + 1); }

void BigintGroupTable::ensure_store(int64_t need)
{
    if (need <= store_cap_) return;
    int64_t cap = store_cap_ ? store_cap_ : 1024;
    while (cap < need) cap <<= 1;
    BufferPtr nv = ctx_->alloc((size_t)cap * (size_t)width_), nn = ctx_->alloc((size_t)cap);
    if (groups_) {
        HIP_CHECK(hipMemcpyAsync(nv->ptr(), values_->ptr(), (size_t)groups_ * (size_t)width_, hipMemcpyDeviceToDevice, ctx_->stream()));
        HIP_CHECK(hipMemcpyAsync(nn->ptr(), nulls_->ptr(), (size_t)groups_, hipMemcpyDeviceToDevice, ctx_->stream()));
    }
    values_ = nv;
    nulls_ = nn;
    store_cap_ = cap;
}

// a table of at least `min_capacity` slots holding the published groups (pending marks of an aborted sub-batch are dropped)
void BigintGroupTable::rebuild(int64_t min_capacity)
{
    int64_t want = 1024;
    while (want < min_capacity) want <<= 1;
    if (want > (1ll << 30)) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "Size of hash table cannot exceed 1 billion entries");
    BufferPtr ns = ctx_->alloc((size_t)(want + 2) * 16);
    HIP_CHECK(hipMemsetAsync(ns->ptr(), 0xff, (size_t)(want + 2) * 16, ctx_->stream()));
    int log2 = 0;
    while ((1ll << log2) < want) log2++;
    if (groups_ > 0) {
        ProfileScope ps(ctx_, "gbh_rehash");
        if (width_ == 8)
            gbi_rehash_kernel<int64_t><<<grid_for(ctx_, groups_, kBlock), kBlock, 0, ctx_->stream()>>>(values_->as<int64_t>(), nulls_->as<uint8_t>(), groups_, ns->as<Slot>(),
                                                                                                     (unsigned int)want, 64 - log2);
        else
            gbi_rehash_kernel<int32_t><<<grid_for(ctx_, groups_, kBlock), kBlock, 0, ctx_->stream()>>>(values_->as<int32_t>(), nulls_->as<uint8_t>(), groups_, ns->as<Slot>(),
                                                                                                     (unsigned int)want, 64 - log2);
        check_launch("gbh_rehash");
    }
    slots_ = ns;
    capacity_ = want;
    shift_ = 64 - log2;
}

void BigintGroupTable::ensure_table(int64_t need_groups)
{
    // fill <= 0.75 like the reference's table (BigintGroupByHash.java:325-334)
    int64_t want = 1024;
    while ((double)want * 0.75 < (double)need_groups + 1) want <<= 1;
    if (want <= capacity_) return;
    rebuild(std::max(want, capacity_ * 2));
}

bool BigintGroupTable::process(const DeviceColumn &keys, const uint8_t *row_mask, int64_t n, int32_t *out_gids, unsigned long long *ctr, int64_t *new_groups_out)
{
    *new_groups_out = 0;
    ensure_store(groups_ > 0 ? groups_ : 1);
    unsigned int *out = reinterpret_cast<unsigned int *>(out_gids);
    const int64_t tiles = ceil_div(n, kTile), words = ceil_div(n, 64);
    BufferPtr cand = ctx_->alloc((size_t)words * 8), displaced = ctx_->alloc_zero((size_t)words * 8), first = ctx_->alloc((size_t)words * 8),
              counts = ctx_->alloc((size_t)tiles * 4), prefix = ctx_->alloc((size_t)tiles * 4);
    Slot *slots = slots_->as<Slot>();
    {
        ProfileScope ps(ctx_, "gbh_insert");
        const int g = grid_for(ctx_, n, kTile);
        if (width_ == 8)
            gbi_insert_kernel<int64_t, true><<<g, kBlock, 0, ctx_->stream()>>>((const int64_t *)keys.values, keys.nulls, row_mask, n, slots, (unsigned int)capacity_, shift_, out,
                                                                              cand->as<unsigned long long>(), displaced->as<unsigned long long>(), ctr);
        else
            gbi_insert_kernel<int32_t, true><<<g, kBlock, 0, ctx_->stream()>>>((const int32_t *)keys.values, keys.nulls, row_mask, n, slots, (unsigned int)capacity_, shift_, out,
                                                                              cand->as<unsigned long long>(), displaced->as<unsigned long long>(), ctr);
        check_launch("gbh_insert");
    }
    {
        // gated on the pending count on the device: with nothing pending (the steady state) both launches return at once, and the
        // ONE read-back below brings the pending count, the number of new groups and the overflow flag
        ProfileScope ps(ctx_, "gbh_assign");
        gbi_mark_kernel<<<(unsigned)ceil_div(words, kBlock), kBlock, 0, ctx_->stream()>>>(words, cand->as<unsigned long long>(), displaced->as<unsigned long long>(),
                                                                                        first->as<unsigned long long>(), counts->as<int32_t>(), ctr);
        check_launch("gbh_mark");
        gbi_scan_kernel<<<1, kScanThreads, 0, ctx_->stream()>>>(counts->as<int32_t>(), prefix->as<int32_t>(), tiles, ctr);
        check_launch("gbh_scan");
    }
    unsigned long long host_ctr[8];
    ctx_->download(host_ctr, ctr, sizeof(host_ctr));
    if (host_ctr[2] != 0) return false;   // table overflow (an optimistic sub-batch met more new keys than the table had room for)
    if (host_ctr[0] == 0) return true;    // every row hit a published group
    const int64_t new_groups = (int64_t)host_ctr[1];
    TG_CHECK_STATE(new_groups > 0, "pending rows without new groups");
    ensure_store(groups_ + new_groups);
    {
        ProfileScope ps(ctx_, "gbh_finalize");
        if (width_ == 8)
            gbi_publish_kernel<int64_t><<<(unsigned)tiles, kBlock, 0, ctx_->stream()>>>((const int64_t *)keys.values, keys.nulls, out, n, slots, first->as<unsigned long long>(),
                                                                                       prefix->as<int32_t>(), groups_, values_->as<int64_t>(), nulls_->as<uint8_t>());
        else
            gbi_publish_kernel<int32_t><<<(unsigned)tiles, kBlock, 0, ctx_->stream()>>>((const int32_t *)keys.values, keys.nulls, out, n, slots, first->as<unsigned long long>(),
                                                                                       prefix->as<int32_t>(), groups_, values_->as<int32_t>(), nulls_->as<uint8_t>());
        check_launch("gbh_publish");
        gbi_resolve_kernel<<<grid_for(ctx_, n, kBlock), kBlock, 0, ctx_->stream()>>>(out, n, slots);
        check_launch("gbh_resolve");
    }
    groups_ += new_groups;
    *new_groups_out = new_groups;
    return true;
}

void BigintGroupTable::lookup(const DeviceColumn &keys, int64_t n, int32_t *out_gids, unsigned long long *ctr)
{
    ensure_table(groups_);
    const int g = grid_for(ctx_, n, kTile);
    if (width_ == 8)
        gbi_insert_kernel<int64_t, false><<<g, kBlock, 0, ctx_->stream()>>>((const int64_t *)keys.values, keys.nulls, nullptr, n, slots_->as<Slot>(), (unsigned int)capacity_, shift_,
                                                                           reinterpret_cast<unsigned int *>(out_gids), nullptr, nullptr, ctr);
    else
        gbi_insert_kernel<int32_t, false><<<g, kBlock, 0, ctx_->stream()>>>((const int32_t *)keys.values, keys.nulls, nullptr, n, slots_->as<Slot>(), (unsigned int)capacity_, shift_,
                                                                           reinterpret_cast<unsigned int *>(out_gids), nullptr, nullptr, ctr);
    check_launch("gbh_lookup");
}

DeviceColumn BigintGroupTable::key_column()
{
    ensure_store(groups_ > 0 ? groups_ : 1);
    DeviceColumn c;
    c.type = type_;
    c.n = groups_;
    c.values_buf = values_;
    c.values = values_->ptr();
    c.nulls_buf = nulls_;
    c.nulls = nulls_->as<uint8_t>();
    return c;
}

}  // namespace tgpu
