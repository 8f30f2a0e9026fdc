// device_groupby.h -- the group-by table's probe/insert step (K4/K5), templated on a key accessor so that the AOT kernel
// (generic, run-time typed keys) and the JIT kernels (code generated for one key schema, with the filter fused in front --
// the GPU counterpart of the reference's JoinCompiler-generated PagesHashStrategy) share one protocol.
// Self-contained; needs device_hash.h (tg_fmix64) before it.
//
// Slot word (uint64):  EMPTY = ~0 | NEW(tag,row) = 01|tag16<<32|row32 | OLD(tag,gid) = 00|tag16<<32|gid32   (see groupby.hip)
#pragma once

#define TG_GBH_EMPTY (~0ULL)

__device__ inline unsigned long long tg_gbh_new(unsigned int tag, unsigned int row) { return (1ULL << 62) | ((unsigned long long)tag << 32) | row; }
__device__ inline unsigned long long tg_gbh_old(unsigned int tag, unsigned int gid) { return ((unsigned long long)tag << 32) | gid; }

// K must provide:  long long hash(long long r)            raw hash of row r's key (H5)
//                  bool eq_store(long long r, int gid)    key(r) IS NOT DISTINCT FROM the stored key of group gid
//                  bool eq_row(long long r, long long r2) key(r) IS NOT DISTINCT FROM key(r2), both rows of this batch
// Returns the group id (>= 0), or -(slot + 2) when the row joined / created a slot that is NEW in this batch (pending), or -1
// (INSERT == false and the key is absent).  counters[2] is set if the table is full (cannot happen by construction).
template <bool INSERT, typename K>
__device__ inline int tg_gbh_probe(const K &k, long long r, unsigned long long *words, unsigned long long mask, int store_groups,
                                   unsigned long long *counters, bool &pending)
{
    pending = false;
    // a handful of groups (TPCH Q1: 4): compare against the key store directly -- no hashing, no table access
    if (store_groups > 0 && store_groups <= 8) {
        for (int g = 0; g < store_groups; g++)
            if (k.eq_store(r, g)) return g;
    }
    const unsigned long long m = tg_fmix64((unsigned long long)k.hash(r));
    unsigned long long pos = m & mask;
    const unsigned int tag = (unsigned int)(m >> 48);
    for (unsigned long long iter = 0; iter <= mask; iter++) {
        // a probe sequence this long means the table is (over)full: flag it and let every lane bail out quickly -- the host
        // rebuilds a bigger table and re-runs the rows (groupby.hip get_group_ids)
        if ((iter & 63) == 63 && (iter >= 8192 || counters[2] != 0)) break;
        // plain (cacheable) load: a stale value is harmless, every decision taken on it is re-validated by the CAS / atomicMin
        unsigned long long w = words[pos];
        if (w == TG_GBH_EMPTY) {
            if (!INSERT) return -1;
            const unsigned long long old = atomicCAS(&words[pos], TG_GBH_EMPTY, tg_gbh_new(tag, (unsigned int)r));
            if (old == TG_GBH_EMPTY) {
                pending = true;
                return -(int)(pos + 2);
            }
            w = old;
        }
        if ((unsigned int)((w >> 32) & 0xffff) == tag) {
            if ((w >> 62) == 0) {
                const unsigned int gid = (unsigned int)w;
                if (k.eq_store(r, (int)gid)) return (int)gid;
            }
            else if (INSERT) {
                const unsigned int r2 = (unsigned int)w;
                if (r2 == (unsigned int)r || k.eq_row(r, (long long)r2)) {
                    if ((unsigned int)r < r2) atomicMin(&words[pos], tg_gbh_new(tag, (unsigned int)r));
                    pending = true;
                    return -(int)(pos + 2);
                }
            }
        }
        pos = (pos + 1) & mask;
    }
    atomicExch(&counters[2], 1ULL);
    return -1;
}
