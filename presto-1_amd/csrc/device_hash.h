// device_hash.h -- the reference's hash family as host+device inline functions (bit-exact).
// Also embedded verbatim into JIT-compiled kernels (jit.cpp reads this file), so it must stay
// self-contained: no includes beyond <stdint.h>-style builtin types.
//
// S/ = core/trino-spi/src/main/java/io/trino/spi/ , M/ = core/trino-main/src/main/java/io/trino/
#pragma once

#ifndef TGPU_HD
#define TGPU_HD __host__ __device__ inline __attribute__((always_inline))
#endif

typedef unsigned long long tg_u64;
typedef long long tg_i64;
typedef unsigned int tg_u32;
typedef unsigned char tg_u8;

TGPU_HD tg_u64 tg_rotl64(tg_u64 x, int r) { return (x << r) | (x >> (64 - r)); }

// H1: S/type/AbstractLongType.java:126-130
TGPU_HD tg_i64 tg_hash_long(tg_i64 v)
{
    return (tg_i64)(tg_rotl64((tg_u64)v * 0xC2B2AE3D27D4EB4FULL, 31) * 0x9E3779B185EBCA87ULL);
}
// H2: S/type/AbstractIntType.java:141-145 (sign-extended)
TGPU_HD tg_i64 tg_hash_int(int v) { return tg_hash_long((tg_i64)v); }
// H3: S/type/DoubleType.java:163-170 ; doubleToLongBits canonicalises NaN, -0.0 -> +0.0
TGPU_HD tg_i64 tg_hash_double_bits(tg_u64 bits)
{
    if ((bits << 1) == 0) bits = 0;                                   // +-0.0 -> +0.0
    if ((bits & 0x7fffffffffffffffULL) > 0x7ff0000000000000ULL) bits = 0x7ff8000000000000ULL; // NaN
    return tg_hash_long((tg_i64)bits);
}

#define TG_P1 0x9E3779B185EBCA87ULL
#define TG_P2 0xC2B2AE3D27D4EB4FULL
#define TG_P3 0x165667B19E3779F9ULL
#define TG_P4 0x85EBCA77C2B2AE63ULL
#define TG_P5 0x27D4EB2F165667C5ULL

TGPU_HD tg_u64 tg_xxh_round(tg_u64 acc, tg_u64 in) { return tg_rotl64(acc + in * TG_P2, 31) * TG_P1; }
TGPU_HD tg_u64 tg_xxh_merge(tg_u64 h, tg_u64 v) { return (h ^ tg_xxh_round(0, v)) * TG_P1 + TG_P4; }
TGPU_HD tg_u64 tg_xxh_avalanche(tg_u64 h)
{
    h ^= h >> 33; h *= TG_P2; h ^= h >> 29; h *= TG_P3; h ^= h >> 32;
    return h;
}
TGPU_HD tg_u64 tg_rd64(const tg_u8 *p)
{
    tg_u64 v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}
TGPU_HD tg_u32 tg_rd32(const tg_u8 *p) { return (tg_u32)p[0] | ((tg_u32)p[1] << 8) | ((tg_u32)p[2] << 16) | ((tg_u32)p[3] << 24); }

// H4: XxHash64.hash(Slice, offset, length), seed 0 (S/block/AbstractVariableWidthBlock.java:92-95)
TGPU_HD tg_u64 tg_xxh64(const tg_u8 *p, tg_i64 len)
{
    const tg_u8 *end = p + len;
    tg_u64 h;
    if (len >= 32) {
        tg_u64 v1 = TG_P1 + TG_P2, v2 = TG_P2, v3 = 0, v4 = 0 - TG_P1;
        const tg_u8 *limit = end - 32;
        do {
            v1 = tg_xxh_round(v1, tg_rd64(p)); v2 = tg_xxh_round(v2, tg_rd64(p + 8));
            v3 = tg_xxh_round(v3, tg_rd64(p + 16)); v4 = tg_xxh_round(v4, tg_rd64(p + 24));
            p += 32;
        } while (p <= limit);
        h = tg_rotl64(v1, 1) + tg_rotl64(v2, 7) + tg_rotl64(v3, 12) + tg_rotl64(v4, 18);
        h = tg_xxh_merge(h, v1); h = tg_xxh_merge(h, v2); h = tg_xxh_merge(h, v3); h = tg_xxh_merge(h, v4);
    }
    else {
        h = TG_P5;
    }
    h += (tg_u64)len;
    while (p + 8 <= end) { h ^= tg_xxh_round(0, tg_rd64(p)); h = tg_rotl64(h, 27) * TG_P1 + TG_P4; p += 8; }
    if (p + 4 <= end) { h ^= (tg_u64)tg_rd32(p) * TG_P1; h = tg_rotl64(h, 23) * TG_P2 + TG_P3; p += 4; }
    while (p < end) { h ^= (tg_u64)(*p) * TG_P5; h = tg_rotl64(h, 11) * TG_P1; p++; }
    return tg_xxh_avalanche(h);
}

// XxHash64.hash(long)
TGPU_HD tg_i64 tg_xxh64_long(tg_i64 v)
{
    tg_u64 h = TG_P5 + 8;
    h ^= tg_xxh_round(0, (tg_u64)v);
    h = tg_rotl64(h, 27) * TG_P1 + TG_P4;
    return (tg_i64)tg_xxh_avalanche(h);
}
// BOOLEAN: S/type/BooleanType.java:39-40,151-155
TGPU_HD tg_i64 tg_hash_boolean(tg_u8 v) { return tg_xxh64_long(v ? 1 : 0); }

// H5: M/operator/scalar/CombineHashFunction.java:24-29
TGPU_HD tg_i64 tg_combine_hash(tg_i64 prev, tg_i64 v) { return (tg_i64)(31ULL * (tg_u64)prev + (tg_u64)v); }

// H6: M/operator/PagesHash.java:224-240 (== fastutil HashCommon.murmurHash3)
TGPU_HD tg_u64 tg_fmix64(tg_u64 x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// H8 remote: M/operator/HashGenerator.java:24-35
TGPU_HD int tg_partition_remote(tg_i64 raw, int n) { return (int)((raw & 0x7fffffffffffffffLL) % n); }
// M/operator/exchange/LocalPartitionGenerator.java:45-65: (int) XxHash64.hash(Long.reverse(rawHash)) & (n - 1), n a power of two
TGPU_HD int tg_partition_local(tg_i64 raw, int n)
{
    tg_u64 x = (tg_u64)raw;
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = (x >> 32) | (x << 32);
    return (int)tg_xxh64_long((tg_i64)x) & (n - 1);
}
