// device_join.h -- probe-side device code shared by the AOT join kernels and the JIT-fused filter+project+probe kernels.
// Self-contained (embedded verbatim into generated sources after device_hash.h).
#pragma once

// fast path table for a single BIGINT / INTEGER / DATE key: key stored inline, one 16-byte load per probe step
struct TgSlot16 {
    long long key;
    int head;   // build position of the newest row with this key (PagesHash.key[]), -1 = empty
    int count;  // tables with repeated build keys (position links): rows that carry this key = the length of the chain behind `head`, so that
                // counting a probe row's matches needs no walk over the links (one random line per probe row instead of 1 + chain length)
};

// Blocked Bloom filter in front of the table: one 64-bit word per key, 4 bits set.  Sized to stay cache resident
// (16 bits per build key), it answers most probes of keys that are NOT in the build side without touching the table.
__device__ inline unsigned long long tg_bloom_mask(unsigned long long m)
{
    return (1ULL << (m & 63)) | (1ULL << ((m >> 6) & 63)) | (1ULL << ((m >> 12) & 63)) | (1ULL << ((m >> 18) & 63));
}
__device__ inline unsigned long long tg_bloom_word(unsigned long long m, unsigned long long word_mask) { return (m >> 24) & word_mask; }

// Pre-filter in front of the table (answers "key certainly not in the build side" without touching the table):
//   * dense key domains (TPCH keys): an exact direct-address bitmap, one bit per key value in [key_min, key_max] -- its access
//     pattern follows the probe keys, so clustered probe keys stream through it and small domains stay L2 resident;
//   * otherwise the blocked Bloom filter above.
struct TgPrefilter {
    const unsigned long long *bitmap;   // nullptr = not available
    long long key_min, key_max;
    const unsigned long long *bloom;    // nullptr = not available
    unsigned long long bloom_word_mask;
    // DIRECT layout (dense key domain without duplicate build keys): no hash table at all -- the bitmap says whether a key is
    // in the build side, and the key's RANK among the present keys (rank_base[word] = set bits in front of the bitmap word,
    // plus the set bits below the key's own bit) indexes direct[], the build positions in key order.  Both side arrays are
    // dense: 4 bytes per 64 key values + 4 bytes per build row, whatever the spread of the keys.
    const int *direct;      // nullptr with rank_base set: build positions ARE the ranks (build side in strictly ascending key order)
    const int *rank_base;   // non-null = DIRECT layout
};

// DIRECT layout: build position of the present key at offset d = key - key_min, `word` = its bitmap word
__device__ inline int tg_direct_position(const TgPrefilter &pf, unsigned long long d, unsigned long long word)
{
    const int rank = pf.rank_base[d >> 6] + __popcll(word & ((1ULL << (d & 63)) - 1ULL));
    return pf.direct ? pf.direct[rank] : rank;   // no direct[]: the build side arrived in strictly ascending key order, position = rank
}

// Slot of a key in the int-key table: Fibonacci hashing, the top log2(capacity) bits of key * 2^64 / phi (one 64-bit multiply
// per probe instead of the four of the reference's hash + a finaliser; the table layout is not observable, only build
// positions are).  mask = capacity - 1, capacity a power of two >= 1024.
__device__ inline unsigned long long tg_slot_of(long long key, unsigned long long mask)
{
    return ((unsigned long long)key * 0x9E3779B97F4A7C15ULL) >> (64 - __popcll(mask));
}

// head build position of `key`, or -1 (PagesHash.getAddressIndex, M/operator/PagesHash.java:157-169, specialised); *count = the slot's chain length
__device__ inline int tg_find_head_int(const TgSlot16 *slots, unsigned long long mask, const TgPrefilter &pf, long long key, int *count = nullptr)
{
    if (pf.bitmap) {
        if (key < pf.key_min || key > pf.key_max) return -1;
        const unsigned long long d = (unsigned long long)(key - pf.key_min);
        const unsigned long long word = pf.bitmap[d >> 6];
        if (!((word >> (d & 63)) & 1ULL)) return -1;
        if (pf.rank_base) return tg_direct_position(pf, d, word);
    }
    else if (pf.bloom) {
        const unsigned long long m = tg_fmix64((unsigned long long)tg_hash_long(key));
        const unsigned long long *bloom = pf.bloom;
        const unsigned long long bloom_word_mask = pf.bloom_word_mask;
        const unsigned long long bits = tg_bloom_mask(m);
        if ((bloom[tg_bloom_word(m, bloom_word_mask)] & bits) != bits) return -1;
    }
    unsigned long long pos = tg_slot_of(key, mask);
    for (unsigned long long iter = 0; iter <= mask; iter++) {
        const TgSlot16 s = slots[pos];
        if (s.head < 0) return -1;
        if (s.key == key) {
            if (count) *count = s.count;
            return s.head;
        }
        pos = (pos + 1) & mask;
    }
    return -1;
}
