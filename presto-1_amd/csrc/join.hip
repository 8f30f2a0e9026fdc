// join.hip -- hash-join build (K7) and probe (K8) kernels + PagesIndex storage.
//
// Build (PagesHash constructor, M/operator/PagesHash.java:53-125): the Java loop inserts rows in position order; equal keys
// chain newest -> oldest (ArrayPositionLinks.link(new, existing), M/operator/ArrayPositionLinks.java:45-50), so the slot of a
// key holds its MAXIMUM build position and links[p] is the next smaller position with the same key.  That end state is
// reproduced in parallel: every non-null row claims / joins its key's slot with CAS + atomicMax (slot = max position), and --
// only when duplicates exist -- (slot, position) pairs are radix-sorted so that each position's predecessor is its link.
// The table layout itself (which slot a key lands in) is unobservable through JoinHash and is free to differ.
//
// Probe (PagesHash.getAddressIndex :157-169 + JoinHash.getNextJoinPosition + PageJoiner.joinCurrentPosition,
// M/operator/LookupJoinOperator.java:299-347): one lane per probe row; pass 1 counts the row's matches, a scan turns counts
// into output offsets (probe positions stay ascending as LookupJoinPageBuilder.java:144-153 asserts), pass 2 writes the pairs.
#include "join.h"
#include "kernels.h"
#include "device_hash.h"
#include "device_join.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>

namespace tgpu {

namespace {

constexpr int kBlock = 256;

int grid_for(Context *ctx, int64_t n)
{
    int64_t blocks = ceil_div(n, kBlock);
    int64_t cap = (int64_t)ctx->cu_count() * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

using Slot16 = TgSlot16;

// EQUAL operators on non-null cells (S/type/AbstractLongType.java:132-136, DoubleType.java:157-161: NaN != NaN, -0 == +0)
__device__ __forceinline__ bool rows_equal_nonnull(const KeyCols &a, int64_t ra, const KeyCols &b, int64_t rb)
{
#pragma unroll
    for (int c = 0; c < TG_MAX_KEY_CHANNELS; c++) {
        if (c >= a.n) break;
        const ColView &x = a.c[c], &y = b.c[c];
        switch (x.type) {
        case TGPU_BIGINT:
            if (((const int64_t *)x.values)[ra] != ((const int64_t *)y.values)[rb]) return false;
            break;
        case TGPU_INTEGER:
        case TGPU_DATE:
            if (((const int32_t *)x.values)[ra] != ((const int32_t *)y.values)[rb]) return false;
            break;
        case TGPU_DOUBLE:
            if (!(((const double *)x.values)[ra] == ((const double *)y.values)[rb])) return false;
            break;
        case TGPU_BOOLEAN:
            if ((((const uint8_t *)x.values)[ra] != 0) != (((const uint8_t *)y.values)[rb] != 0)) return false;
            break;
        case TGPU_VARCHAR: {
            int32_t ax = x.offsets[ra], lx = x.offsets[ra + 1] - ax;
            int32_t ay = y.offsets[rb], ly = y.offsets[rb + 1] - ay;
            if (lx != ly) return false;
            const uint8_t *px = (const uint8_t *)x.values + ax, *py = (const uint8_t *)y.values + ay;
            for (int32_t i = 0; i < lx; i++)
                if (px[i] != py[i]) return false;
            break;
        }
        default: return false;
        }
    }
    return true;
}

__device__ __forceinline__ bool row_has_null(const KeyCols &k, int64_t r)
{
#pragma unroll
    for (int c = 0; c < TG_MAX_KEY_CHANNELS; c++) {
        if (c >= k.n) break;
        if (k.c[c].nulls && k.c[c].nulls[r]) return true;
    }
    return false;
}

__device__ __forceinline__ long long int_key_at(const ColView &c, int64_t r)
{
    return c.type == TGPU_BIGINT ? ((const long long *)c.values)[r] : (long long)((const int *)c.values)[r];
}

// PagesIndex.addPage / MergePages' appendPage as ONE launch: every buffer of the page (values and null vectors of all channels, VARCHAR
// byte ranges) and every VARCHAR channel's rebased offsets.  A per-buffer hipMemcpyAsync costs ~5 us of host time each -- nine of them per
// 7-channel page were most of MergePages' per-page cost (DESIGN.md "Page granularity").
struct AppendJob {
    const uint8_t *src;
    uint8_t *dst;
    long long bytes;        // plain copy; for an offsets job: the row count
    int src_base, dst_base; // offsets job (kind 1): dst[i + 1] = dst_base + (src[i + 1] - src_base)
    int kind, pad;
};
constexpr int kMaxAppendJobs = 40;
struct AppendJobs {
    AppendJob j[kMaxAppendJobs];
};
__global__ void __launch_bounds__(kBlock) append_page_kernel(AppendJobs jobs)
{
    // the job of this blockIdx.y, selected without indexing the by-value argument at run time (that would copy it to scratch)
    AppendJob job = jobs.j[0];
#pragma unroll
    for (int k = 1; k < kMaxAppendJobs; k++)
        if ((int)blockIdx.y == k) job = jobs.j[k];
    const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x, stride = (long long)gridDim.x * kBlock;
    if (job.kind == 1) {
        const int *src = (const int *)job.src;
        int *dst = (int *)job.dst;
        for (long long i = tid; i < job.bytes; i += stride) dst[i + 1] = job.dst_base + (src[i + 1] - job.src_base);
        return;
    }
    if ((((unsigned long long)job.src | (unsigned long long)job.dst) & 15ull) == 0) {
        const long long n16 = job.bytes >> 4;
        const uint4 *s = (const uint4 *)job.src;
        uint4 *d = (uint4 *)job.dst;
        for (long long i = tid; i < n16; i += stride) d[i] = s[i];
        for (long long i = (n16 << 4) + tid; i < job.bytes; i += stride) job.dst[i] = job.src[i];
    }
    else
        for (long long i = tid; i < job.bytes; i += stride) job.dst[i] = job.src[i];
}

__global__ void __launch_bounds__(kBlock) append_offsets_kernel(const int32_t *__restrict__ src, int64_t n, int32_t src_base, int32_t dst_base,
                                                                 int32_t *__restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) dst[i + 1] = dst_base + (src[i + 1] - src_base);
}

__global__ void __launch_bounds__(kBlock) tags_kernel(const int64_t *__restrict__ hashes, int64_t n, uint8_t *__restrict__ tags)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) tags[i] = (uint8_t)hashes[i];
}

// folds the lanes' (lo, hi) into minmax[0] / [1] with ONE pair of atomics per workgroup (wave shuffles, then LDS): thousands of
// waves hitting the same two words with atomics would serialise for longer than the scan itself
__device__ __forceinline__ void block_minmax(long long lo, long long hi, long long *minmax)
{
    __shared__ long long s_lo[kBlock / 64], s_hi[kBlock / 64];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const long long l2 = __shfl_down(lo, d, 64), h2 = __shfl_down(hi, d, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBlock / 64; w++) {
            lo = s_lo[w] < lo ? s_lo[w] : lo;
            hi = s_hi[w] > hi ? s_hi[w] : hi;
        }
        if (lo <= hi) {
            atomicMin(&minmax[0], lo);
            atomicMax(&minmax[1], hi);
        }
    }
}

// counters[0] = rows that joined an existing key (ArrayPositionLinks.FactoryBuilder.size()), counters[1] = error
__global__ void __launch_bounds__(kBlock) build_insert_kernel(KeyCols keys, const int64_t *__restrict__ hashes, const uint8_t *__restrict__ tags, int64_t n,
                                                               int *heads, int stride, uint64_t mask, int32_t *__restrict__ row_slot,
                                                               unsigned long long *counters)
{
    // `heads` is either the plain int32 slot array (stride 1) or the head field of the TgSlot16 array (stride 4 words)
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        bool dup = false;
        int32_t slot = -1;
        if (!row_has_null(keys, r)) {  // PagesHash.java:94-96: rows with a null key are not indexed
            const int64_t h = hashes[r];
            uint64_t pos = tg_fmix64((uint64_t)h) & mask;
            for (uint64_t iter = 0; iter <= mask; iter++) {
                int *hp = heads + pos * stride;
                int cur = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == -1) {
                    int old = atomicCAS(hp, -1, (int)r);
                    if (old == -1) { slot = (int32_t)pos; break; }
                    cur = old;
                }
                if (tags[cur] == (uint8_t)h && rows_equal_nonnull(keys, cur, keys, r)) {
                    atomicMax(hp, (int)r);
                    slot = (int32_t)pos;
                    dup = true;
                    break;
                }
                pos = (pos + 1) & mask;
                if (iter == mask) atomicExch(&counters[1], 1ull);
            }
        }
        row_slot[r] = slot;
        unsigned long long b = __ballot(dup);
        if (dup && (threadIdx.x & 63) == (__ffsll((long long)b) - 1)) atomicAdd(&counters[0], (unsigned long long)__popcll(b));
    }
}

// int-key fast path: insert + key publication + key range in one pass, four rows per lane: all row loads, then all slot loads (one
// 16-byte load brings key and head), then the claiming CAS of every row that found its slot empty -- independent requests in flight
// together (tools/exp_random_access.hip: the memory system answers 55 G random loads and 27 G atomics per second, whatever the table size;
// one row per lane with its load -> CAS -> store chain left most of that unused).  The row that claims an empty slot stores the key into
// it; a row that meets an occupied slot compares against the slot's key when it is published, and against the KEY COLUMN at the slot's
// current head while the slot still shows the unwritten pattern (the claimer's store may be on its way -- or the key may really be
// that value: the column decides either way).  A row with the slot's key raises the head to the newest position (PagesHash.java:104-119
// keeps the last inserted position as the head of the key's chain).
// counters[0] = rows that joined an existing key, counters[1] = error; minmax[0] / [1] = smallest / largest indexed key
constexpr long long kUnwrittenKey = (long long)0x8000000000000001ull;
constexpr int kBuildRows = 4;

__device__ __forceinline__ int build_insert_row(const ColView &key, long long k, int r, Slot16 *slots, uint64_t mask, uint64_t pos, long long ks, int head, bool &dup,
                                                unsigned long long *counters)
{
    for (uint64_t iter = 0; iter <= mask; iter++) {
        if (head == -1) {
            const int old = atomicCAS(&slots[pos].head, -1, r);
            if (old == -1) {
                __hip_atomic_store(&slots[pos].key, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return (int)pos;
            }
            head = old;
            ks = __hip_atomic_load(&slots[pos].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool same = ks == kUnwrittenKey ? int_key_at(key, head) == k : ks == k;
        if (same) {
            (void)__hip_atomic_fetch_max(&slots[pos].head, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dup = true;
            return (int)pos;
        }
        pos = (pos + 1) & mask;
        const ulonglong2 v = *(const ulonglong2 *)&slots[pos];
        ks = (long long)v.x;
        head = (int)(unsigned int)v.y;
    }
    atomicExch(&counters[1], 1ull);
    return -1;
}

__global__ void __launch_bounds__(kBlock) build_insert_int_kernel(ColView key, int64_t n, Slot16 *slots, uint64_t mask, int32_t *__restrict__ row_slot,
                                                                   unsigned long long *counters, long long *minmax)
{
    long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
    unsigned int my_dups = 0;
    for (int64_t base = (int64_t)blockIdx.x * (kBlock * kBuildRows); base < n; base += (int64_t)gridDim.x * (kBlock * kBuildRows)) {
        long long k[kBuildRows];
        bool live[kBuildRows], claimed[kBuildRows];
        uint64_t pos[kBuildRows];
        ulonglong2 snap[kBuildRows];
        int old[kBuildRows];
#pragma unroll
        for (int u = 0; u < kBuildRows; u++) {
            const int64_t r = base + u * kBlock + threadIdx.x;
            live[u] = r < n && !(key.nulls && key.nulls[r]);   // PagesHash.java:94-96: rows with a null key are not indexed
            k[u] = live[u] ? int_key_at(key, r) : 0;
            if (live[u]) {
                lo = k[u] < lo ? k[u] : lo;
                hi = k[u] > hi ? k[u] : hi;
            }
        }
#pragma unroll
        for (int u = 0; u < kBuildRows; u++) {
            pos[u] = tg_slot_of(k[u], mask);
            snap[u] = *(const ulonglong2 *)&slots[live[u] ? pos[u] : 0];
        }
#pragma unroll
        for (int u = 0; u < kBuildRows; u++) {
            claimed[u] = live[u] && (int)(unsigned int)snap[u].y == -1;
            old[u] = 0;
            if (claimed[u]) old[u] = atomicCAS(&slots[pos[u]].head, -1, (int)(base + u * kBlock + threadIdx.x));
        }
#pragma unroll
        for (int u = 0; u < kBuildRows; u++)
            if (claimed[u] && old[u] == -1) __hip_atomic_store(&slots[pos[u]].key, k[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < kBuildRows; u++) {
            const int64_t r = base + u * kBlock + threadIdx.x;
            int32_t slot = -1;
            bool dup = false;
            if (claimed[u] && old[u] == -1) slot = (int32_t)pos[u];
            else if (live[u]) {
                long long ks = (long long)snap[u].x;
                int head = (int)(unsigned int)snap[u].y;
                if (claimed[u]) {   // lost the slot to another row
                    head = old[u];
                    ks = __hip_atomic_load(&slots[pos[u]].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                slot = build_insert_row(key, k[u], (int)r, slots, mask, pos[u], ks, head, dup, counters);
            }
            if (r < n) row_slot[r] = slot;
            my_dups += dup ? 1u : 0u;
        }
    }
    // one atomic per workgroup
    __shared__ unsigned int s_dups[kBlock / 64];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) my_dups += __shfl_down(my_dups, d, 64);
    if ((threadIdx.x & 63) == 0) s_dups[threadIdx.x >> 6] = my_dups;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) total += s_dups[w];
        if (total) atomicAdd(&counters[0], total);
    }
    block_minmax(lo, hi, minmax);
}

__global__ void __launch_bounds__(kBlock) sort_keys_kernel(const int32_t *__restrict__ row_slot, int64_t n, unsigned long long *__restrict__ keys)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        int32_t s = row_slot[r];
        keys[r] = s < 0 ? ~0ull : (((unsigned long long)(uint32_t)s << 32) | (unsigned long long)(uint32_t)r);
    }
}

// sorted = (slot << 32 | position), ascending: a slot's rows sit next to each other, oldest first.  links[p] = the next older row of p's key; the
// newest row of a run (the slot's head) also counts the run and publishes the count in the slot (int-key table: TgSlot16.count)
__global__ void __launch_bounds__(kBlock) links_kernel(const unsigned long long *__restrict__ sorted, int64_t n, int32_t *__restrict__ links, Slot16 *slots)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        unsigned long long k = sorted[i];
        if (k == ~0ull) continue;  // null-key rows: links stay -1
        int32_t link = -1;
        if (i > 0) {
            unsigned long long p = sorted[i - 1];
            if ((p >> 32) == (k >> 32)) link = (int32_t)(uint32_t)p;
        }
        links[(uint32_t)k] = link;
        if (slots && (i + 1 == n || (sorted[i + 1] >> 32) != (k >> 32))) {
            int64_t first = i;
            while (first > 0 && (sorted[first - 1] >> 32) == (k >> 32)) first--;
            slots[k >> 32].count = (int)(i - first + 1);
        }
    }
}

// DIRECT layout, step 1: smallest / largest non-null key; minmax[2] is set when some key is not larger than its predecessor
// (it stays 0 for a build side in strictly ascending key order: no repeated key, and a key's rank is its build position)
__global__ void __launch_bounds__(kBlock) key_range_kernel(ColView key, int64_t n, long long *minmax)
{
    long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
    bool descent = false;
    // four independent loads per lane and iteration; few, long-running blocks: every block ends in two atomics on the same
    // two words, which serialise in L2
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int lane = threadIdx.x & 63;
    for (int64_t r0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; r0 < n; r0 += 4 * stride) {
        long long k[4], before[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {   // loads only: the keys, and (first lane of the wave) the key in front of the wave's rows
            const int64_t r = r0 + u * stride;
            live[u] = r < n;
            const int64_t rc = live[u] ? r : r0;
            live[u] = live[u] && !(key.nulls && key.nulls[rc]);
            k[u] = int_key_at(key, rc);
            before[u] = (lane == 0 && r < n && r > 0) ? int_key_at(key, r - 1) : (-0x7fffffffffffffffLL - 1);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t r = r0 + u * stride;
            lo = (live[u] && k[u] < lo) ? k[u] : lo;
            hi = (live[u] && k[u] > hi) ? k[u] : hi;
            // the predecessor of a lane's row is the previous lane's row (the rows of a wave are consecutive)
            const long long up = __shfl_up(k[u], 1, 64);
            const long long pred = lane == 0 ? before[u] : up;
            descent = descent || (r < n && r > 0 && pred >= k[u]);
        }
    }
    block_minmax(lo, hi, minmax);
    if (__any(descent) && (threadIdx.x & 63) == 0) minmax[2] = 1;   // idempotent plain store
}

// DIRECT layout, step 2: one pass sets the key's bit in the (zeroed) bitmap.  A bit that was already set means a repeated build
// key: counted in counters[0]; the caller then drops this layout (position links need the hash table).
__global__ void __launch_bounds__(kBlock) build_direct_kernel(ColView key, int64_t n, long long key_min, unsigned long long *bitmap,
                                                               unsigned long long *counters)
{
    // Neighbouring rows often fall into the same bitmap word (build sides clustered by key: consecutive customer keys share one
    // word, TPCH order keys every other one): the lanes of a contiguous run with the same word OR their bits together first and
    // only the run's first lane touches memory -- atomics on one cache line serialise in L2.
    const int lane = threadIdx.x & 63;
    unsigned int my_dups = 0;
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
        const int64_t r = base + threadIdx.x;
        const bool active = r < n && !(key.nulls && key.nulls[r]);   // PagesHash.java:94-96: rows with a null key are not indexed
        unsigned long long d = 0;
        if (active) {
            d = (unsigned long long)int_key_at(key, r) - (unsigned long long)key_min;
        }
        const unsigned int word = active ? (unsigned int)(d >> 6) : 0xffffffffu;
        unsigned long long bits = active ? (1ULL << (d & 63)) : 0ULL;
        unsigned int rows = active ? 1u : 0u;
        const unsigned int word_prev = __shfl_up(word, 1, 64);
        const bool head = lane == 0 || word_prev != word;
        const unsigned long long heads = __ballot(head);
        const int run = __popcll(heads & ((2ULL << lane) - 1ULL));   // id of the contiguous run this lane belongs to
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {   // segmented suffix reduction: the run's first lane ends up with the whole run
            const unsigned long long b2 = __shfl_down(bits, k, 64);
            const unsigned int c2 = __shfl_down(rows, k, 64);
            const int run2 = __shfl_down(run, k, 64);
            if (lane + k < 64 && run2 == run) {
                bits |= b2;
                rows += c2;
            }
        }
        if (head && active) {
            const unsigned long long old = atomicOr(&bitmap[word], bits);
            // repeated build keys: a bit that was already set, or two rows of the run with the same bit
            my_dups += (unsigned int)__popcll(old & bits) + (rows - (unsigned int)__popcll(bits));
        }
    }
    __shared__ unsigned int s_dups[kBlock / 64];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) my_dups += __shfl_down(my_dups, d, 64);
    if (lane == 0) s_dups[threadIdx.x >> 6] = my_dups;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) total += s_dups[w];
        if (total) atomicAdd(&counters[0], total);
    }
}

// int-key fast path: empty table = {key = the unwritten pattern, head -1, count 1}
__global__ void __launch_bounds__(kBlock) init_slots_kernel(Slot16 *__restrict__ slots, int64_t capacity)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < capacity; i += (int64_t)gridDim.x * kBlock) {
        Slot16 s;
        s.key = kUnwrittenKey;
        s.head = -1;
        s.count = 1;
        slots[i] = s;
    }
}

// DIRECT layout, step 3: present keys per bitmap word (its exclusive scan is rank_base)
__global__ void __launch_bounds__(kBlock) bitmap_popcount_kernel(const unsigned long long *__restrict__ bitmap, int64_t words, int32_t *__restrict__ counts)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < words; i += (int64_t)gridDim.x * kBlock) counts[i] = (int32_t)__popcll(bitmap[i]);
}

// DIRECT layout, step 4: every build row stores its position at the rank of its key: positions in key order, written densely
// (a build side that arrives clustered by key writes them front to back)
__global__ void __launch_bounds__(kBlock) build_rank_kernel(ColView key, int64_t n, long long key_min, const unsigned long long *__restrict__ bitmap,
                                                             const int32_t *__restrict__ rank_base, int32_t *__restrict__ direct)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        if (key.nulls && key.nulls[r]) continue;
        const unsigned long long d = (unsigned long long)int_key_at(key, r) - (unsigned long long)key_min;
        direct[rank_base[d >> 6] + __popcll(bitmap[d >> 6] & ((1ULL << (d & 63)) - 1ULL))] = (int32_t)r;
    }
}

__global__ void __launch_bounds__(kBlock) bitmap_build_kernel(const int32_t *__restrict__ row_slot, int64_t n, ColView key, long long key_min,
                                                               unsigned long long *bitmap)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        if (row_slot[r] < 0) continue;
        const unsigned long long d = (unsigned long long)(int_key_at(key, r) - key_min);
        atomicOr(&bitmap[d >> 6], 1ULL << (d & 63));
    }
}

// blocked Bloom filter over the build keys of the int-key fast path (device_join.h), for sparse key domains
__global__ void __launch_bounds__(kBlock) bloom_build_kernel(const int32_t *__restrict__ row_slot, int64_t n, ColView key, unsigned long long *bloom,
                                                              unsigned long long word_mask)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        if (row_slot[r] < 0) continue;
        const unsigned long long m = tg_fmix64((unsigned long long)tg_hash_long(int_key_at(key, r)));
        atomicOr(&bloom[tg_bloom_word(m, word_mask)], tg_bloom_mask(m));
    }
}

struct ProbeTable {
    const int *heads;
    const Slot16 *slots;
    const uint8_t *tags;
    const int32_t *links;
    uint64_t mask;
    TgPrefilter pf;
};

// head build position of the probe row's key, or -1 (PagesHash.getAddressIndex)
template <bool FAST>
__device__ __forceinline__ int find_head(const ProbeTable &t, const KeyCols &build, const KeyCols &probe, int64_t r, int64_t h)
{
    if (FAST) return tg_find_head_int(t.slots, t.mask, t.pf, int_key_at(probe.c[0], r));   // slot position from the key alone
    uint64_t pos = tg_fmix64((uint64_t)h) & t.mask;
    for (uint64_t iter = 0; iter <= t.mask; iter++) {
        const int b = t.heads[pos];
        if (b < 0) return -1;
        if (t.tags[b] == (uint8_t)h && rows_equal_nonnull(build, b, probe, r)) return b;  // positionEqualsCurrentRowIgnoreNulls :198-209
        pos = (pos + 1) & t.mask;
    }
    return -1;
}

template <bool FAST>
__global__ void __launch_bounds__(kBlock) probe_count_kernel(ProbeTable t, KeyCols build, KeyCols probe, const int64_t *__restrict__ hashes, int64_t n,
                                                              int probe_outer, int32_t *__restrict__ heads_out, int32_t *__restrict__ counts)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        int head = -1, chain = 0;
        if (!row_has_null(probe, r)) {  // JoinProbe.java:87-97
            if (FAST) head = tg_find_head_int(t.slots, t.mask, t.pf, int_key_at(probe.c[0], r), &chain);   // the slot knows its chain's length
            else head = find_head<FAST>(t, build, probe, r, hashes[r]);
        }
        int32_t c = 0;
        if (head >= 0) {
            c = 1;
            if (t.links) {
                if (FAST && !t.pf.rank_base) c = chain;
                else
                    for (int p = t.links[head]; p >= 0; p = t.links[p]) c++;
            }
        }
        else if (probe_outer) c = 1;
        heads_out[r] = head;
        counts[r] = c;
    }
}

__global__ void __launch_bounds__(kBlock) probe_write_kernel(const int32_t *__restrict__ links, const int32_t *__restrict__ heads, const int32_t *__restrict__ offsets,
                                                              const int32_t *__restrict__ counts, int64_t n, int probe_outer, int32_t *__restrict__ out_probe,
                                                              int32_t *__restrict__ out_build)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        int head = heads[r];
        int64_t o = offsets[r];
        if (head >= 0) {
            // the row's match count is known: the last link (the -1 that ends the chain) is never fetched
            int p = head;
            for (int32_t left = counts[r]; left > 0; left--) {
                out_probe[o] = (int32_t)r;
                out_build[o] = p;
                o++;
                if (left > 1) p = links[p];
            }
        }
        else if (probe_outer) {
            out_probe[o] = (int32_t)r;
            out_build[o] = -1;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
PagesIndexGpu::PagesIndexGpu(Context *ctx, std::vector<int32_t> types) : ctx_(ctx), types_(std::move(types))
{
    cols_.resize(types_.size());
    for (size_t i = 0; i < types_.size(); i++) {
        TG_CHECK_ARG(valid_type(types_[i]), "unknown type");
        cols_[i].type = types_[i];
    }
}

int64_t PagesIndexGpu::estimated_size() const
{
    int64_t s = 0;
    for (auto &c : cols_) s += c.cap * type_width(c.type) + (c.nulls ? c.cap : 0) + (c.offsets ? (c.cap + 1) * 4 : 0) + c.pool_cap;
    return s;
}

void PagesIndexGpu::reserve(int64_t rows)
{
    for (auto &c : cols_) {
        if (rows <= c.cap) continue;
        int64_t cap = c.cap ? c.cap : 1024;
        while (cap < rows) cap <<= 1;
        if (c.cap == 0) cap = std::max<int64_t>(rows, 1024);  // first page: exact fit (a single big build page needs no slack)
        if (c.type == TGPU_VARCHAR) {
            BufferPtr no = ctx_->alloc((size_t)(cap + 1) * 4);
            if (c.offsets) HIP_CHECK(hipMemcpyAsync(no->ptr(), c.offsets->ptr(), (size_t)(n_ + 1) * 4, hipMemcpyDeviceToDevice, ctx_->stream()));
            else HIP_CHECK(hipMemsetAsync(no->ptr(), 0, 4, ctx_->stream()));
            c.offsets = no;
        }
        else {
            const int w = type_width(c.type);
            BufferPtr nv = ctx_->alloc((size_t)cap * w);
            if (n_) HIP_CHECK(hipMemcpyAsync(nv->ptr(), c.values->ptr(), (size_t)n_ * w, hipMemcpyDeviceToDevice, ctx_->stream()));
            c.values = nv;
        }
        if (c.nulls) {
            BufferPtr nn = ctx_->alloc_zero((size_t)cap);
            if (n_) HIP_CHECK(hipMemcpyAsync(nn->ptr(), c.nulls->ptr(), (size_t)n_, hipMemcpyDeviceToDevice, ctx_->stream()));
            c.nulls = nn;
        }
        c.cap = cap;
    }
}

void PagesIndexGpu::add_page(const DevicePage &page, const std::vector<std::array<int32_t, 2>> *varchar_ends)
{
    TG_CHECK_ARG(page.cols.size() == types_.size(), "page channel count does not match the index");
    if (page.n == 0) return;
    if (n_ + page.n > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "Size of pages index cannot exceed 2 billion entries");  // PagesIndex.java:234-236
    reserve(n_ + page.n);
    AppendJobs jobs{};
    int n_jobs = 0;
    long long most = 0;
    auto launch_jobs = [&]() {
        if (n_jobs == 0) return;
        dim3 grid((unsigned)std::min<long long>(std::max<long long>(ceil_div(most / 16 + 1, kBlock), 1), (long long)ctx_->cu_count() * 4), (unsigned)n_jobs);
        append_page_kernel<<<grid, kBlock, 0, ctx_->stream()>>>(jobs);
        check_launch("append_page");
        n_jobs = 0;
        most = 0;
    };
    auto add_job = [&](const void *src, void *dst, long long bytes, int kind = 0, int src_base = 0, int dst_base = 0) {
        if (bytes <= 0) return;
        if (n_jobs == kMaxAppendJobs) launch_jobs();
        jobs.j[n_jobs++] = AppendJob{(const uint8_t *)src, (uint8_t *)dst, bytes, src_base, dst_base, kind, 0};
        most = std::max(most, kind == 1 ? bytes * 16 : bytes);
    };
    for (size_t i = 0; i < cols_.size(); i++) {
        Store &c = cols_[i];
        const DeviceColumn &src = page.cols[i];
        TG_CHECK_ARG(src.type == c.type, "page channel type does not match the index");
        if (src.nulls) {
            if (!c.nulls) c.nulls = ctx_->alloc_zero((size_t)c.cap);
            add_job(src.nulls, c.nulls->as<uint8_t>() + n_, page.n);
            c.has_nulls = true;
        }
        if (c.type == TGPU_VARCHAR) {
            int32_t a = 0, b = 0;
            if (varchar_ends) {
                a = (*varchar_ends)[i][0];
                b = (*varchar_ends)[i][1];
            }
            else if (src.pool_exact) {
                a = src.pool_first;
                b = (int32_t)src.pool_bytes;
            }
            else {
                ctx_->download(&a, src.offsets, 4);
                ctx_->download(&b, src.offsets + page.n, 4);
            }
            const int64_t bytes = (int64_t)b - a;
            if (c.pool_used + bytes > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "variable width channel of the pages index cannot exceed 2GB");
            if (c.pool_used + bytes > c.pool_cap) {
                int64_t cap = c.pool_cap ? c.pool_cap : 4096;
                while (cap < c.pool_used + bytes) cap <<= 1;
                BufferPtr nv = ctx_->alloc((size_t)cap);
                if (c.pool_used) HIP_CHECK(hipMemcpyAsync(nv->ptr(), c.values->ptr(), (size_t)c.pool_used, hipMemcpyDeviceToDevice, ctx_->stream()));
                c.values = nv;
                c.pool_cap = cap;
            }
            add_job((const uint8_t *)src.values + a, c.values->as<uint8_t>() + c.pool_used, bytes);
            add_job(src.offsets, c.offsets->as<int32_t>() + n_, page.n, 1, a, (int32_t)c.pool_used);
            c.pool_used += bytes;
        }
        else {
            const int w = type_width(c.type);
            add_job(src.values, c.values->as<uint8_t>() + n_ * w, (long long)page.n * w);
        }
    }
    launch_jobs();
    n_ += page.n;
}

DeviceColumn PagesIndexGpu::column(int ch) const
{
    TG_CHECK_ARG(ch >= 0 && ch < (int)cols_.size(), "channel out of range");
    const Store &s = cols_[ch];
    DeviceColumn c;
    c.type = s.type;
    c.n = n_;
    c.values_buf = s.values;
    c.values = s.values ? s.values->ptr() : nullptr;
    if (s.has_nulls) {
        c.nulls_buf = s.nulls;
        c.nulls = s.nulls->as<uint8_t>();
    }
    if (s.type == TGPU_VARCHAR) {
        c.offsets_buf = s.offsets;
        c.offsets = s.offsets ? s.offsets->as<int32_t>() : nullptr;
        c.pool_bytes = s.pool_used;
        c.pool_exact = true;
    }
    return c;
}

// ---------------------------------------------------------------------------------------------------------------------
LookupSourceGpu::LookupSourceGpu(Context *ctx, std::shared_ptr<PagesIndexGpu> index, std::vector<int32_t> key_channels, int32_t hash_channel,
                                 std::vector<int32_t> output_channels)
    : ctx_(ctx), index_(std::move(index)), key_channels_(std::move(key_channels)), output_channels_(std::move(output_channels)), hash_channel_(hash_channel)
{
    TG_CHECK_ARG(!key_channels_.empty() && (int)key_channels_.size() <= kMaxKeyChannels, "join needs 1..8 key channels");
}

std::vector<int32_t> LookupSourceGpu::output_types() const
{
    std::vector<int32_t> t;
    for (int32_t ch : output_channels_) t.push_back(index_->types()[ch]);
    return t;
}

std::vector<int32_t> LookupSourceGpu::key_types() const
{
    std::vector<int32_t> t;
    for (int32_t ch : key_channels_) t.push_back(index_->types()[ch]);
    return t;
}

int64_t LookupSourceGpu::estimated_size() const
{
    return index_->estimated_size() + (rank_base_ ? (int64_t)((direct_ ? direct_->bytes() : 0) + rank_base_->bytes()) : capacity_ * (int_key_fast_ ? 16 : 4)) + (tags_ ? n_ : 0) + (links_ ? n_ * 4 : 0) +
           (bitmap_ ? (int64_t)bitmap_->bytes() : 0) + (bloom_ ? bloom_words_ * 8 : 0);
}

void LookupSourceGpu::build()
{
    n_ = index_->position_count();
    key_cols_.clear();
    for (int32_t ch : key_channels_) key_cols_.push_back(index_->column(ch));
    int_key_fast_ = key_cols_.size() == 1 && (key_cols_[0].type == TGPU_BIGINT || key_cols_[0].type == TGPU_INTEGER || key_cols_[0].type == TGPU_DATE);
    // table size: the reference uses arraySize(n, 0.75) (PagesHash.java:63); the layout is unobservable, so the int-key fast
    // path runs at load <= 0.5 (HBM is plentiful and almost every probe then resolves in its first slot)
    const double max_load = int_key_fast_ ? 0.5 : 0.75;
    capacity_ = 1024;
    while ((double)capacity_ * max_load < (double)n_) capacity_ <<= 1;
    if (capacity_ > (1ll << 31)) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "hash table size cannot exceed 2 billion slots");
    link_count_ = 0;
    direct_.reset();
    rank_base_.reset();
    links_.reset();
    tags_.reset();
    bitmap_.reset();
    bloom_.reset();
    heads_.reset();
    slots16_.reset();
    key_min_ = 0;
    key_max_ = -1;
    int *heads = nullptr;
    int stride = 1;
    if (int_key_fast_ && n_ > 0 && getenv("TGPU_DISABLE_DIRECT") == nullptr) {
        std::vector<const DeviceColumn *> kp0;
        for (auto &c : key_cols_) kp0.push_back(&c);
        if (build_direct(key_cols_of(kp0))) return;
    }
    if (int_key_fast_) {
        slots16_ = ctx_->alloc((size_t)capacity_ * 16);
        init_slots_kernel<<<grid_for(ctx_, capacity_), kBlock, 0, ctx_->stream()>>>(slots16_->as<Slot16>(), capacity_);
        check_launch("init_slots");
        heads = &slots16_->as<Slot16>()->head;
        stride = 4;
    }
    else {
        heads_ = ctx_->alloc((size_t)capacity_ * 4);
        k::fill_i32(ctx_, heads_->as<int32_t>(), -1, capacity_);
        heads = heads_->as<int>();
    }
    if (n_ == 0) return;
    std::vector<const DeviceColumn *> kp;
    for (auto &c : key_cols_) kp.push_back(&c);
    const KeyCols keys = key_cols_of(kp);

    BufferPtr row_slot = ctx_->alloc((size_t)n_ * 4);
    BufferPtr counters = ctx_->alloc(32);   // [0] duplicate rows, [1] error, [2] smallest key, [3] largest key
    {
        const long long init[4] = {0, 0, 0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1};
        ctx_->upload(counters->ptr(), init, 32);
    }
    const int g = grid_for(ctx_, n_);
    if (int_key_fast_) {
        // slot positions come from the key alone (tg_slot_of), so that probes -- and the JIT-fused probe kernels -- need no hash
        // channel; no row hashes, tags or separate key publication pass on this path
        ProfileScope ps(ctx_, "join_build_insert");
        build_insert_int_kernel<<<g, kBlock, 0, ctx_->stream()>>>(keys.c[0], n_, slots16_->as<Slot16>(), (uint64_t)capacity_ - 1, row_slot->as<int32_t>(),
                                                                 counters->as<unsigned long long>(), counters->as<long long>() + 2);
        check_launch("build_insert_int");
    }
    else {
        BufferPtr own_hashes;
        const int64_t *hashes;
        DeviceColumn hash_col;
        if (hash_channel_ >= 0) {  // precomputed $hashvalue channel: JoinCompiler.java:405-428
            hash_col = index_->column(hash_channel_);
            TG_CHECK_ARG(hash_col.type == TGPU_BIGINT, "hash channel must be BIGINT");
            hashes = (const int64_t *)hash_col.values;
        }
        else {
            own_hashes = ctx_->alloc((size_t)n_ * 8);
            k::hash_rows(ctx_, keys, n_, own_hashes->as<int64_t>());
            hashes = own_hashes->as<int64_t>();
        }
        tags_ = ctx_->alloc((size_t)n_);
        ProfileScope ps(ctx_, "join_build_insert");
        tags_kernel<<<g, kBlock, 0, ctx_->stream()>>>(hashes, n_, tags_->as<uint8_t>());
        build_insert_kernel<<<g, kBlock, 0, ctx_->stream()>>>(keys, hashes, tags_->as<uint8_t>(), n_, heads, stride, (uint64_t)capacity_ - 1,
                                                             row_slot->as<int32_t>(), counters->as<unsigned long long>());
        check_launch("build_insert");
    }
    unsigned long long host_ctr[4];
    ctx_->download(host_ctr, counters->ptr(), 32);
    TG_CHECK_STATE(host_ctr[1] == 0, "join table overflow");
    link_count_ = (int64_t)host_ctr[0];
    if (link_count_ > 0) {
        // duplicates: links[p] = next smaller position with the same key (= same slot)
        ProfileScope ps(ctx_, "join_build_links");
        links_ = ctx_->alloc((size_t)n_ * 4);
        k::fill_i32(ctx_, links_->as<int32_t>(), -1, n_);
        BufferPtr keys_in = ctx_->alloc((size_t)n_ * 8), keys_out = ctx_->alloc((size_t)n_ * 8);
        sort_keys_kernel<<<g, kBlock, 0, ctx_->stream()>>>(row_slot->as<int32_t>(), n_, keys_in->as<unsigned long long>());
        size_t temp_bytes = 0;
        HIP_CHECK(rocprim::radix_sort_keys(nullptr, temp_bytes, keys_in->as<unsigned long long>(), keys_out->as<unsigned long long>(), (size_t)n_, 0, 64, ctx_->stream()));
        BufferPtr temp = ctx_->alloc(temp_bytes ? temp_bytes : 1);
        HIP_CHECK(rocprim::radix_sort_keys(temp->ptr(), temp_bytes, keys_in->as<unsigned long long>(), keys_out->as<unsigned long long>(), (size_t)n_, 0, 64, ctx_->stream()));
        links_kernel<<<g, kBlock, 0, ctx_->stream()>>>(keys_out->as<unsigned long long>(), n_, links_->as<int32_t>(), int_key_fast_ ? slots16_->as<Slot16>() : nullptr);
        check_launch("links");
    }
    if (int_key_fast_) {
        ProfileScope ps(ctx_, "join_build_prefilter");
        key_min_ = (long long)host_ctr[2];
        key_max_ = (long long)host_ctr[3];
        // pre-filter: exact bitmap when the key domain is dense enough, else a blocked Bloom filter (16 bits per build row)
        const bool has_keys = key_max_ >= key_min_;
        const unsigned long long range = has_keys ? (unsigned long long)key_max_ - (unsigned long long)key_min_ + 1ULL : 0ULL;
        // dense: at most 4096 candidate key values per build row and a bitmap of at most 2 GiB
        const bool dense = has_keys && range != 0 && range <= (1ULL << 34) && range / 4096ULL <= (unsigned long long)std::max<int64_t>(n_, 1) &&
                           getenv("TGPU_DISABLE_BITMAP") == nullptr;
        if (dense) {
            const int64_t words = (int64_t)((range + 63) / 64);
            bitmap_ = ctx_->alloc_zero((size_t)words * 8);
            bitmap_build_kernel<<<g, kBlock, 0, ctx_->stream()>>>(row_slot->as<int32_t>(), n_, keys.c[0], key_min_, bitmap_->as<unsigned long long>());
            check_launch("bitmap_build");
        }
        else if (has_keys) {
            static const int bits_per_key = getenv("TGPU_BLOOM_BITS_PER_KEY") ? std::max(1, atoi(getenv("TGPU_BLOOM_BITS_PER_KEY"))) : 16;   // kernel study
            int64_t words = 1024;
            while (words * 64 < n_ * bits_per_key) words <<= 1;
            // (Kernel study, round 3: a probe of the filter and a probe of the table are one random request each, so the filter cannot lower
            // the request count -- yet without it the 324 M-row probe of 14.6 M sparse keys takes 13.5 ms instead of 6.95, the 100 M-row
            // duplicate-key probe 2.60 instead of 2.42: a probe that MISSES walks 2.5 slots on average at a fill of 0.5 before it meets an
            // empty one, each step a dependent load, and the filter spares 7 of 8 rows that walk.  TGPU_BLOOM_MAX_BYTES bounds the filter.)
            static const int64_t max_bloom_bytes = getenv("TGPU_BLOOM_MAX_BYTES") ? atoll(getenv("TGPU_BLOOM_MAX_BYTES")) : (1ll << 40);
            if (words * 8 <= max_bloom_bytes) {
                bloom_words_ = words;
                bloom_ = ctx_->alloc_zero((size_t)words * 8);
                bloom_build_kernel<<<g, kBlock, 0, ctx_->stream()>>>(row_slot->as<int32_t>(), n_, keys.c[0], bloom_->as<unsigned long long>(), (unsigned long long)words - 1);
                check_launch("bloom_build");
            }
        }
    }
}

// DIRECT layout of the int-key fast path: when the key domain is dense (at least one key value in 64 is present, at most 2^31
// values) and no build key repeats, the "table" is an exact bitmap over [key_min, key_max] plus the build positions in key
// order, addressed by the key's rank among the present keys (rank_base[word] + popcount of the lower bits of the word): the
// build is one atomicOr per row, a scan over the bitmap words and one dense store per row (no probing, no key comparison, no
// write amplification from scattering 4-byte entries over the key range); a probe is a bit test, and only a MATCH pays the two
// further loads; inputs clustered by key walk all three arrays front to back.  Returns false (nothing kept) when the keys do
// not qualify; the hash table is built then.
bool LookupSourceGpu::build_direct(const KeyCols &keys)
{
    const int g = grid_for(ctx_, n_);
    BufferPtr mm = ctx_->alloc(24);
    const long long init[3] = {0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1, 0};
    ctx_->upload(mm->ptr(), init, 24);
    long long host_mm[3];
    {
        ProfileScope ps(ctx_, "join_build_range");
        key_range_kernel<<<std::min(g, ctx_->cu_count() * 2), kBlock, 0, ctx_->stream()>>>(keys.c[0], n_, mm->as<long long>());
        check_launch("key_range");
    }
    ctx_->download(host_mm, mm->ptr(), 24);
    if (host_mm[1] < host_mm[0]) return false;   // only null keys
    // strictly ascending keys without a null vector: no key repeats (no duplicate check, no second read-back) and a key's rank is
    // its build position (no direct[] array, no scatter pass) -- the shape of a build side that comes out of an ordered scan
    const bool ascending = host_mm[2] == 0 && keys.c[0].nulls == nullptr && getenv("TGPU_DISABLE_ASCENDING_BUILD") == nullptr;
    const unsigned long long range = (unsigned long long)host_mm[1] - (unsigned long long)host_mm[0] + 1ULL;
    if (range == 0 || range > (1ULL << 31) || range / 64ULL > (unsigned long long)n_) return false;
    const int64_t words = (int64_t)((range + 63) / 64);
    BufferPtr bitmap = ctx_->alloc_zero((size_t)words * 8);
    BufferPtr direct = ascending ? nullptr : ctx_->alloc((size_t)n_ * 4);
    BufferPtr rank_base = ctx_->alloc((size_t)words * 4);
    BufferPtr counters = ctx_->alloc_zero(16);
    {
        ProfileScope ps(ctx_, "join_build_insert");
        build_direct_kernel<<<g, kBlock, 0, ctx_->stream()>>>(keys.c[0], n_, host_mm[0], bitmap->as<unsigned long long>(), counters->as<unsigned long long>());
        check_launch("build_direct");
    }
    if (!ascending && ctx_->read_scalar(counters->as<unsigned long long>()) != 0) return false;   // repeated build keys: position links needed
    key_min_ = host_mm[0];
    key_max_ = host_mm[1];
    bitmap_ = bitmap;
    direct_ = direct;
    rank_base_ = rank_base;
    direct_key_ = keys.c[0];
    rank_pending_ = true;
    // a table without output channels is usually probed for membership only (the build side of a semi-join-like filter, Q3's
    // customer table): the rank structure is then built on first demand (ensure_rank), i.e. normally never
    if (!output_channels_.empty()) ensure_rank();
    return true;
}

// DIRECT layout: steps 3 and 4 (rank_base, positions in key order); idempotent, callable from any probe operator's thread
void LookupSourceGpu::ensure_rank() const
{
    if (!rank_base_) return;
    std::lock_guard<std::mutex> lk(visited_mu_);
    if (!rank_pending_) return;
    const int64_t range = (int64_t)((unsigned long long)key_max_ - (unsigned long long)key_min_ + 1ULL);
    const int64_t words = (range + 63) / 64;
    ProfileScope ps(ctx_, "join_build_rank");
    BufferPtr counts = ctx_->alloc((size_t)words * 4), total = ctx_->alloc(8);
    bitmap_popcount_kernel<<<grid_for(ctx_, words), kBlock, 0, ctx_->stream()>>>(bitmap_->as<unsigned long long>(), words, counts->as<int32_t>());
    check_launch("bitmap_popcount");
    k::exclusive_scan_i32(ctx_, counts->as<int32_t>(), rank_base_->as<int32_t>(), words, total->as<int64_t>());
    if (direct_) {   // (not for a build side in ascending key order: position = rank)
        build_rank_kernel<<<grid_for(ctx_, n_), kBlock, 0, ctx_->stream()>>>(direct_key_, n_, key_min_, bitmap_->as<unsigned long long>(), rank_base_->as<int32_t>(),
                                                                            direct_->as<int32_t>());
        check_launch("build_rank");
    }
    rank_pending_ = false;
}

bool LookupSourceGpu::int_table(IntTableView &v) const
{
    if (!int_key_fast_ || (!slots16_ && !rank_base_)) return false;
    v.slots = slots16_ ? slots16_->ptr() : nullptr;
    v.direct = direct_ ? direct_->as<int32_t>() : nullptr;
    v.rank_base = rank_base_ ? rank_base_->as<int32_t>() : nullptr;
    v.mask = (uint64_t)capacity_ - 1;
    v.bloom = bloom_ ? bloom_->as<unsigned long long>() : nullptr;
    v.bloom_word_mask = bloom_ ? (unsigned long long)bloom_words_ - 1 : 0;
    v.bitmap = bitmap_ ? bitmap_->as<unsigned long long>() : nullptr;
    v.key_min = key_min_;
    v.key_max = key_max_;
    v.links = links_ ? links_->as<int32_t>() : nullptr;
    v.key_type = key_cols_.empty() ? index_->types()[(size_t)key_channels_[0]] : key_cols_[0].type;
    return true;
}

void LookupSourceGpu::probe(const std::vector<const DeviceColumn *> &probe_keys, const int64_t *probe_hashes, int64_t n, bool probe_outer,
                            BufferPtr &out_probe_idx, BufferPtr &out_build_idx, int64_t &out_count)
{
    TG_CHECK_ARG(probe_keys.size() == key_channels_.size(), "probe key channel count differs from the build side's");
    for (size_t i = 0; i < probe_keys.size(); i++) {
        const int32_t bt = index_->types()[key_channels_[i]];
        const bool int_like = (bt == TGPU_BIGINT || bt == TGPU_INTEGER || bt == TGPU_DATE);
        TG_CHECK_ARG(probe_keys[i]->type == bt || (int_key_fast_ && int_like && probe_keys[i]->type == bt), "probe key type differs from the build key type");
    }
    out_count = 0;
    if (n <= 0) {
        out_probe_idx = ctx_->alloc(4);
        out_build_idx = ctx_->alloc(4);
        return;
    }
    ensure_rank();   // this path always produces build positions
    const KeyCols probe = key_cols_of(probe_keys);
    std::vector<const DeviceColumn *> kp;
    for (auto &c : key_cols_) kp.push_back(&c);
    const KeyCols build = key_cols_.empty() ? KeyCols{} : key_cols_of(kp);
    BufferPtr own_hashes;
    if (!probe_hashes && !int_key_fast_) {  // hashRow: JoinCompiler.java:449-477 (the int-key table derives its slots from the key)
        own_hashes = ctx_->alloc((size_t)n * 8);
        k::hash_rows(ctx_, probe, n, own_hashes->as<int64_t>());
        probe_hashes = own_hashes->as<int64_t>();
    }
    ProbeTable t{};
    t.heads = heads_ ? heads_->as<int>() : nullptr;
    t.slots = slots16_ ? slots16_->as<Slot16>() : nullptr;
    t.tags = tags_ ? tags_->as<uint8_t>() : nullptr;
    t.links = links_ ? links_->as<int32_t>() : nullptr;
    t.mask = (uint64_t)capacity_ - 1;
    t.pf.bitmap = bitmap_ ? bitmap_->as<unsigned long long>() : nullptr;
    t.pf.key_min = key_min_;
    t.pf.key_max = key_max_;
    t.pf.bloom = bloom_ ? bloom_->as<unsigned long long>() : nullptr;
    t.pf.bloom_word_mask = bloom_ ? (unsigned long long)bloom_words_ - 1 : 0;
    t.pf.direct = direct_ ? direct_->as<int>() : nullptr;
    t.pf.rank_base = rank_base_ ? rank_base_->as<int>() : nullptr;
    BufferPtr heads = ctx_->alloc((size_t)n * 4), counts = ctx_->alloc((size_t)n * 4), offsets = ctx_->alloc((size_t)n * 4), total = ctx_->alloc(8);
    const int g = grid_for(ctx_, n);
    {
        ProfileScope ps(ctx_, "join_probe_count");
        if (int_key_fast_)
            probe_count_kernel<true><<<g, kBlock, 0, ctx_->stream()>>>(t, build, probe, probe_hashes, n, probe_outer ? 1 : 0, heads->as<int32_t>(), counts->as<int32_t>());
        else
            probe_count_kernel<false><<<g, kBlock, 0, ctx_->stream()>>>(t, build, probe, probe_hashes, n, probe_outer ? 1 : 0, heads->as<int32_t>(), counts->as<int32_t>());
        check_launch("probe_count");
    }
    {
        ProfileScope ps(ctx_, "join_probe_scan");
        k::exclusive_scan_i32(ctx_, counts->as<int32_t>(), offsets->as<int32_t>(), n, total->as<int64_t>());
    }
    out_count = ctx_->read_scalar(total->as<int64_t>());
    if (out_count > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "join output of one probe page cannot exceed 2 billion rows");
    out_probe_idx = ctx_->alloc((size_t)(out_count > 0 ? out_count : 1) * 4);
    out_build_idx = ctx_->alloc((size_t)(out_count > 0 ? out_count : 1) * 4);
    if (out_count > 0) {
        ProfileScope ps(ctx_, "join_probe_write");
        probe_write_kernel<<<g, kBlock, 0, ctx_->stream()>>>(t.links, heads->as<int32_t>(), offsets->as<int32_t>(), counts->as<int32_t>(), n, probe_outer ? 1 : 0,
                                                            out_probe_idx->as<int32_t>(), out_build_idx->as<int32_t>());
        check_launch("probe_write");
    }
}

DeviceColumn LookupSourceGpu::gather_index_channel(int channel, const int32_t *build_positions, int64_t n) const
{
    TG_CHECK_ARG(channel >= 0 && channel < (int)index_->types().size(), "build channel out of range");
    return k::gather_column(ctx_, index_->column(channel), build_positions, n, false);
}

DeviceColumn LookupSourceGpu::build_column(int out_idx) const { return index_->column(output_channels_[out_idx]); }

DeviceColumn LookupSourceGpu::gather_build(int out_idx, const int32_t *build_positions, int64_t n, bool negative_is_null) const
{
    DeviceColumn src = index_->column(output_channels_[out_idx]);
    if (src.type != TGPU_VARCHAR && src.values == nullptr) {  // empty build side
        BufferPtr dummy = ctx_->alloc(8);
        src.values_buf = dummy;
        src.values = dummy->ptr();
    }
    if (src.type == TGPU_VARCHAR && src.offsets == nullptr) {
        BufferPtr off = ctx_->alloc_zero(8), pool = ctx_->alloc(8);
        src.offsets_buf = off;
        src.offsets = off->as<int32_t>();
        src.values_buf = pool;
        src.values = pool->ptr();
    }
    return k::gather_column(ctx_, src, build_positions, n, negative_is_null);
}


// ---- outer position tracking ----------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(kBlock) mark_visited_kernel(const int32_t *__restrict__ positions, int64_t n, uint8_t *__restrict__ visited)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int32_t p = positions[i];
        if (p >= 0) visited[p] = 1;   // plain idempotent stores: several probes may mark the same position
    }
}
__global__ void __launch_bounds__(kBlock) unvisited_flags_kernel(const uint8_t *__restrict__ visited, int64_t n, int32_t *__restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) flags[i] = (visited && visited[i]) ? 0 : 1;
}
__global__ void __launch_bounds__(kBlock) compact_positions_kernel(const int32_t *__restrict__ flags, const int32_t *__restrict__ rank, int64_t n, int32_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (flags[i]) out[rank[i]] = (int32_t)i;
}
}  // namespace

void LookupSourceGpu::mark_visited(const int32_t *build_positions, int64_t n)
{
    if (n <= 0 || n_ <= 0) return;
    {
        std::lock_guard<std::mutex> lk(visited_mu_);
        if (!visited_) visited_ = ctx_->alloc_zero((size_t)n_);
    }
    mark_visited_kernel<<<grid_for(ctx_, n), kBlock, 0, ctx_->stream()>>>(build_positions, n, visited_->as<uint8_t>());
    check_launch("mark_visited");
}

void LookupSourceGpu::unvisited_positions(BufferPtr &positions, int64_t &count)
{
    count = 0;
    positions = ctx_->alloc((size_t)(n_ > 0 ? n_ : 1) * 4);
    if (n_ <= 0) return;
    BufferPtr flags = ctx_->alloc((size_t)n_ * 4), rank = ctx_->alloc((size_t)n_ * 4), total = ctx_->alloc(8);
    unvisited_flags_kernel<<<grid_for(ctx_, n_), kBlock, 0, ctx_->stream()>>>(visited_ ? visited_->as<uint8_t>() : nullptr, n_, flags->as<int32_t>());
    check_launch("unvisited_flags");
    k::exclusive_scan_i32(ctx_, flags->as<int32_t>(), rank->as<int32_t>(), n_, total->as<int64_t>());
    compact_positions_kernel<<<grid_for(ctx_, n_), kBlock, 0, ctx_->stream()>>>(flags->as<int32_t>(), rank->as<int32_t>(), n_, positions->as<int32_t>());
    check_launch("compact_positions");
    count = ctx_->read_scalar(total->as<int64_t>());
}

}  // namespace tgpu
