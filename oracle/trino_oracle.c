/*
 * trino_oracle.c -- CPU restatement of the reference's operator hot path.  TEST INFRASTRUCTURE ONLY
 * (see trino_oracle.h).  Row-at-a-time, same data structures and probe sequences as the Java code, so it
 * doubles as the "port" CPU baseline in bench.py.
 *
 * M/ = core/trino-main/src/main/java/io/trino/ , S/ = core/trino-spi/src/main/java/io/trino/spi/
 */
#include "trino_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * hash family
 * ------------------------------------------------------------------------------------------------ */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

/* S/type/AbstractLongType.java:126-130: rotateLeft(value * 0xC2B2AE3D27D4EB4FL, 31) * 0x9E3779B185EBCA87L */
int64_t o_hash_long(int64_t v)
{
    return (int64_t)(rotl64((uint64_t)v * 0xC2B2AE3D27D4EB4FULL, 31) * 0x9E3779B185EBCA87ULL);
}

/* S/type/AbstractIntType.java:141-145: AbstractLongType.hash((int) value) -- sign extended */
int64_t o_hash_int(int32_t v) { return o_hash_long((int64_t)v); }

/* S/type/DoubleType.java:163-170: -0.0 -> +0.0, then hash(doubleToLongBits(v)); doubleToLongBits canonicalises NaN */
int64_t o_hash_double(double v)
{
    if (v == 0) {
        v = 0;
    }
    uint64_t bits;
    if (v != v) {
        bits = 0x7ff8000000000000ULL;
    }
    else {
        memcpy(&bits, &v, 8);
    }
    return o_hash_long((int64_t)bits);
}

/* io.airlift.slice.XxHash64 = standard XXH64 (published algorithm, Yann Collet) */
#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL

static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xxh_round(uint64_t acc, uint64_t in) { return rotl64(acc + in * P2, 31) * P1; }
static inline uint64_t xxh_merge(uint64_t h, uint64_t v) { return (h ^ xxh_round(0, v)) * P1 + P4; }
static inline uint64_t xxh_avalanche(uint64_t h)
{
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

uint64_t o_xxh64(const uint8_t *p, size_t len, uint64_t seed)
{
    const uint8_t *end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t *limit = end - 32;
        do {
            v1 = xxh_round(v1, rd64(p)); v2 = xxh_round(v2, rd64(p + 8));
            v3 = xxh_round(v3, rd64(p + 16)); v4 = xxh_round(v4, rd64(p + 24));
            p += 32;
        } while (p <= limit);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
    }
    else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xxh_round(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p) * P5; h = rotl64(h, 11) * P1; p++; }
    return xxh_avalanche(h);
}

/* XxHash64.hash(long): XXH64 of the 8 little-endian bytes, seed 0 (parity unpinned by reference literals) */
int64_t o_xxh64_long(int64_t v)
{
    uint64_t h = P5 + 8;
    h ^= xxh_round(0, (uint64_t)v);
    h = rotl64(h, 27) * P1 + P4;
    return (int64_t)xxh_avalanche(h);
}

/* S/type/BooleanType.java:39-40,151-155: boolean has no HASH_CODE -> XX_HASH_64 fallback = XxHash64.hash(1 / 0) */
int64_t o_hash_boolean(uint8_t v) { return o_xxh64_long(v ? 1 : 0); }

/* M/operator/scalar/CombineHashFunction.java:24-29 */
int64_t o_combine_hash(int64_t prev, int64_t v) { return (int64_t)(31ULL * (uint64_t)prev + (uint64_t)v); }

/* M/operator/PagesHash.java:224-240 */
uint64_t o_murmur3_fmix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

/* fastutil 8.3.0 HashCommon.arraySize(expected, f) = max(2, nextPowerOfTwo((long) Math.ceil(expected / f)));
 * note `expected / f` is FLOAT division in Java (int / float). */
int32_t o_array_size(int32_t expected, float f)
{
    float q = (float)expected / f;
    int64_t c = (int64_t)ceil((double)q);
    int64_t s = 1;
    while (s < c) s <<= 1;
    if (c == 0) s = 1; /* nextPowerOfTwo(0) == 1 in fastutil */
    if (s < 2) s = 2;
    if (s > (1LL << 30)) return -1;
    return (int32_t)s;
}

/* M/operator/BigintGroupByHash.java:325-334 (FILL_RATIO = 0.75f) */
int32_t o_calculate_max_fill(int32_t hash_size)
{
    int32_t max_fill = (int32_t)ceil((double)((float)hash_size * 0.75f));
    if (max_fill == hash_size) max_fill--;
    return max_fill;
}

static inline int cell_is_null(const o_column *c, int32_t pos) { return c->nulls != NULL && c->nulls[pos] != 0; }

/* M/type/BlockTypeOperators.java:102-108 hashCodeNullSafe, NULL_HASH_CODE = 0 (M/type/TypeUtils.java:42) */
int64_t o_hash_cell(const o_column *c, int32_t pos)
{
    if (cell_is_null(c, pos)) return 0;
    switch (c->type) {
    case O_BIGINT: return o_hash_long(((const int64_t *)c->values)[pos]);
    case O_INTEGER:
    case O_DATE: return o_hash_int(((const int32_t *)c->values)[pos]);
    case O_DOUBLE: return o_hash_double(((const double *)c->values)[pos]);
    case O_BOOLEAN: return o_hash_boolean(((const uint8_t *)c->values)[pos]);
    case O_VARCHAR: {
        /* S/block/AbstractVariableWidthBlock.java:92-95 (VARCHAR has no HASH_CODE: S/type/TypeOperators.java:228-233) */
        int32_t a = c->offsets[pos], b = c->offsets[pos + 1];
        return (int64_t)o_xxh64((const uint8_t *)c->values + a, (size_t)(b - a), 0);
    }
    default: return 0;
    }
}

/* M/operator/InterpretedHashGenerator.java:56-70 ; INITIAL_HASH_VALUE = 0 (HashGenerationOptimizer.java:98) */
void o_hash_rows(const o_column *cols, int32_t ncols, int32_t n, int64_t *out)
{
    for (int32_t r = 0; r < n; r++) {
        int64_t h = 0;
        for (int32_t c = 0; c < ncols; c++) h = o_combine_hash(h, o_hash_cell(&cols[c], r));
        out[r] = h;
    }
}

/* M/operator/HashGenerator.java:24-35 */
int32_t o_partition_remote(int64_t raw_hash, int32_t partition_count)
{
    raw_hash &= 0x7fffffffffffffffLL;
    return (int32_t)(raw_hash % partition_count);
}

static uint64_t bit_reverse64(uint64_t x)
{
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x);
}

/* M/operator/exchange/LocalPartitionGenerator.java:45-65: (int) XxHash64.hash(Long.reverse(rawHash)) & (n - 1) */
int32_t o_partition_local(int64_t raw_hash, int32_t partition_count_pow2)
{
    int32_t h = (int32_t)o_xxh64_long((int64_t)bit_reverse64((uint64_t)raw_hash));
    return h & (partition_count_pow2 - 1);
}

/* value equality used by key comparisons.
 * EQUAL operators: S/type/AbstractLongType.java:132-136, AbstractIntType.java, DoubleType.java:157-161 (left == right),
 * VARCHAR Slice.equals.  "not distinct" adds null == null (JoinCompiler.java positionNotDistinctFromRow). */
static int cells_equal_nonnull(const o_column *a, int32_t pa, const o_column *b, int32_t pb)
{
    switch (a->type) {
    case O_BIGINT: return ((const int64_t *)a->values)[pa] == ((const int64_t *)b->values)[pb];
    case O_INTEGER:
    case O_DATE: return ((const int32_t *)a->values)[pa] == ((const int32_t *)b->values)[pb];
    case O_DOUBLE: return ((const double *)a->values)[pa] == ((const double *)b->values)[pb];
    case O_BOOLEAN: return (((const uint8_t *)a->values)[pa] != 0) == (((const uint8_t *)b->values)[pb] != 0);
    case O_VARCHAR: {
        int32_t la = a->offsets[pa + 1] - a->offsets[pa], lb = b->offsets[pb + 1] - b->offsets[pb];
        if (la != lb) return 0;
        return memcmp((const uint8_t *)a->values + a->offsets[pa], (const uint8_t *)b->values + b->offsets[pb], (size_t)la) == 0;
    }
    default: return 0;
    }
}

/* IS DISTINCT FROM for doubles treats NaN as not distinct from NaN (S/type/DoubleType.java distinctFromOperator);
 * for the other types it is null-aware equality. */
static int cells_not_distinct(const o_column *a, int32_t pa, const o_column *b, int32_t pb)
{
    int na = cell_is_null(a, pa), nb = cell_is_null(b, pb);
    if (na || nb) return na && nb;
    if (a->type == O_DOUBLE) {
        double x = ((const double *)a->values)[pa], y = ((const double *)b->values)[pb];
        if (x != x && y != y) return 1;
        return x == y;
    }
    return cells_equal_nonnull(a, pa, b, pb);
}

/* ------------------------------------------------------------------------------------------------
 * BigintGroupByHash  (M/operator/BigintGroupByHash.java)
 * ------------------------------------------------------------------------------------------------ */
struct o_bigint_gbh {
    int32_t hash_capacity, max_fill, mask;
    int64_t *values;            /* values[hashCapacity]        :60 */
    int32_t *group_ids;         /* groupIds[hashCapacity], -1  :61 */
    int32_t null_group_id;      /* :64 */
    int64_t *values_by_group;   /* valuesByGroupId             :67 */
    int32_t values_by_group_cap;
    int32_t next_group_id;
    int64_t hash_collisions;
    int32_t rehash_count;
};

o_bigint_gbh *o_bigint_gbh_new(int32_t expected_size)
{
    /* constructor :78-101 */
    o_bigint_gbh *g = (o_bigint_gbh *)calloc(1, sizeof(*g));
    g->hash_capacity = o_array_size(expected_size, 0.75f);
    g->max_fill = o_calculate_max_fill(g->hash_capacity);
    g->mask = g->hash_capacity - 1;
    g->values = (int64_t *)calloc((size_t)g->hash_capacity, 8);
    g->group_ids = (int32_t *)malloc((size_t)g->hash_capacity * 4);
    for (int32_t i = 0; i < g->hash_capacity; i++) g->group_ids[i] = -1;
    g->values_by_group_cap = g->hash_capacity;
    g->values_by_group = (int64_t *)calloc((size_t)g->values_by_group_cap, 8);
    g->null_group_id = -1;
    return g;
}

void o_bigint_gbh_free(o_bigint_gbh *g)
{
    if (!g) return;
    free(g->values); free(g->group_ids); free(g->values_by_group); free(g);
}

static void bigint_ensure_vbg(o_bigint_gbh *g, int32_t cap)
{
    if (cap > g->values_by_group_cap) {
        g->values_by_group = (int64_t *)realloc(g->values_by_group, (size_t)cap * 8);
        memset(g->values_by_group + g->values_by_group_cap, 0, (size_t)(cap - g->values_by_group_cap) * 8);
        g->values_by_group_cap = cap;
    }
}

/* tryRehash :262-313 -- re-inserts in GROUP-ID order (:287-303) */
static int32_t bigint_try_rehash(o_bigint_gbh *g)
{
    int64_t new_cap_l = (int64_t)g->hash_capacity * 2;
    if (new_cap_l > 0x7fffffffLL) return O_ERR_INSUFFICIENT_RESOURCES; /* "Size of hash table cannot exceed 1 billion entries" :264-267 */
    int32_t new_cap = (int32_t)new_cap_l, new_mask = new_cap - 1;
    int64_t *nv = (int64_t *)calloc((size_t)new_cap, 8);
    int32_t *ng = (int32_t *)malloc((size_t)new_cap * 4);
    for (int32_t i = 0; i < new_cap; i++) ng[i] = -1;
    for (int32_t gid = 0; gid < g->next_group_id; gid++) {
        if (gid == g->null_group_id) continue;
        int64_t value = g->values_by_group[gid];
        int64_t pos = (int64_t)(o_murmur3_fmix((uint64_t)value) & (uint64_t)new_mask);
        while (ng[pos] != -1) { pos = (pos + 1) & new_mask; g->hash_collisions++; }
        nv[pos] = value; ng[pos] = gid;
    }
    free(g->values); free(g->group_ids);
    g->values = nv; g->group_ids = ng;
    g->mask = new_mask; g->hash_capacity = new_cap; g->max_fill = o_calculate_max_fill(new_cap);
    bigint_ensure_vbg(g, g->max_fill);
    g->rehash_count++;
    return O_OK;
}

/* putIfAbsent :213-242 + addNewGroup :244-260 */
static int32_t bigint_put_if_absent(o_bigint_gbh *g, const o_column *col, int32_t pos, int32_t *err)
{
    if (cell_is_null(col, pos)) {
        if (g->null_group_id < 0) g->null_group_id = g->next_group_id++;
        return g->null_group_id;
    }
    int64_t value = ((const int64_t *)col->values)[pos];
    /* NB: position = murmurHash3(VALUE) & mask, not of H1(value) (:224-225,320-323) */
    int64_t hp = (int64_t)(o_murmur3_fmix((uint64_t)value) & (uint64_t)g->mask);
    while (1) {
        int32_t gid = g->group_ids[hp];
        if (gid == -1) break;
        if (value == g->values[hp]) return gid;
        hp = (hp + 1) & g->mask;
        g->hash_collisions++;
    }
    int32_t gid = g->next_group_id++;
    g->values[hp] = value;
    bigint_ensure_vbg(g, gid + 1);
    g->values_by_group[gid] = value;
    g->group_ids[hp] = gid;
    if (g->next_group_id >= g->max_fill) {
        int32_t rc = bigint_try_rehash(g);
        if (rc != O_OK) *err = rc;
    }
    return gid;
}

int32_t o_bigint_gbh_get_group_ids(o_bigint_gbh *g, const o_column *col, int64_t *out)
{
    int32_t err = O_OK;
    /* GetGroupIdsWork.process :382-402 */
    if (g->next_group_id >= g->max_fill) {
        int32_t rc = bigint_try_rehash(g);
        if (rc != O_OK) return rc;
    }
    for (int32_t i = 0; i < col->n; i++) {
        int32_t gid = bigint_put_if_absent(g, col, i, &err);
        if (out) out[i] = gid;
        if (err != O_OK) return err;
    }
    return O_OK;
}

int32_t o_bigint_gbh_contains(const o_bigint_gbh *g, const o_column *col, int32_t pos)
{
    if (cell_is_null(col, pos)) return g->null_group_id >= 0;
    int64_t value = ((const int64_t *)col->values)[pos];
    int64_t hp = (int64_t)(o_murmur3_fmix((uint64_t)value) & (uint64_t)g->mask);
    while (1) {
        int32_t gid = g->group_ids[hp];
        if (gid == -1) return 0;
        if (value == g->values[hp]) return 1;
        hp = (hp + 1) & g->mask;
    }
}

int32_t o_bigint_gbh_group_count(const o_bigint_gbh *g) { return g->next_group_id; }
int32_t o_bigint_gbh_capacity(const o_bigint_gbh *g) { return g->hash_capacity; }
int64_t o_bigint_gbh_hash_collisions(const o_bigint_gbh *g) { return g->hash_collisions; }
int32_t o_bigint_gbh_rehash_count(const o_bigint_gbh *g) { return g->rehash_count; }

/* appendValuesTo :137-158 */
void o_bigint_gbh_values(const o_bigint_gbh *g, int64_t *values, uint8_t *nulls, int64_t *raw_hashes)
{
    for (int32_t gid = 0; gid < g->next_group_id; gid++) {
        if (gid == g->null_group_id) {
            values[gid] = 0; nulls[gid] = 1;
            if (raw_hashes) raw_hashes[gid] = 0; /* NULL_HASH_CODE */
        }
        else {
            values[gid] = g->values_by_group[gid]; nulls[gid] = 0;
            if (raw_hashes) raw_hashes[gid] = o_hash_long(g->values_by_group[gid]);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * MultiChannelGroupByHash  (M/operator/MultiChannelGroupByHash.java)
 * The Java class copies new keys into internal PageBuilder pages and stores a SyntheticAddress per
 * slot; here the key store is a set of growing flat columns and the "address" is the group's row in it
 * (== its group id, because one key row is appended per new group).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t type;
    int32_t n, cap;
    uint8_t *values;   /* fixed width store, or byte pool */
    int64_t values_cap_bytes;
    uint8_t *nulls;
    int32_t *offsets;  /* VARCHAR: n+1 */
} key_store_col;

struct o_multi_gbh {
    int32_t nchannels;
    int32_t hash_capacity, max_fill, mask;
    int64_t *group_address_by_hash;  /* :75 (-1 empty) */
    int32_t *group_ids_by_hash;      /* :76 */
    uint8_t *raw_hash_by_hash_pos;   /* :77 1-byte tag */
    key_store_col *keys;
    int64_t *raw_hash_by_group;      /* precomputedHashChannel copy (:311-313) / recomputed hash */
    int64_t *first_row_by_group;
    int32_t by_group_cap;
    int32_t next_group_id;
    int64_t hash_collisions;
    int32_t rehash_count;
    int64_t rows_seen;
};

static int32_t type_width(int32_t t)
{
    switch (t) {
    case O_BIGINT: case O_DOUBLE: return 8;
    case O_INTEGER: case O_DATE: return 4;
    case O_BOOLEAN: return 1;
    default: return 0;
    }
}

o_multi_gbh *o_multi_gbh_new(int32_t nchannels, const int32_t *types, int32_t expected_size)
{
    /* constructor :93-157 */
    o_multi_gbh *g = (o_multi_gbh *)calloc(1, sizeof(*g));
    g->nchannels = nchannels;
    g->hash_capacity = o_array_size(expected_size, 0.75f);
    g->max_fill = o_calculate_max_fill(g->hash_capacity);
    g->mask = g->hash_capacity - 1;
    g->group_address_by_hash = (int64_t *)malloc((size_t)g->hash_capacity * 8);
    for (int32_t i = 0; i < g->hash_capacity; i++) g->group_address_by_hash[i] = -1;
    g->group_ids_by_hash = (int32_t *)calloc((size_t)g->hash_capacity, 4);
    g->raw_hash_by_hash_pos = (uint8_t *)calloc((size_t)g->hash_capacity, 1);
    g->keys = (key_store_col *)calloc((size_t)nchannels, sizeof(key_store_col));
    for (int32_t c = 0; c < nchannels; c++) g->keys[c].type = types[c];
    return g;
}

void o_multi_gbh_free(o_multi_gbh *g)
{
    if (!g) return;
    for (int32_t c = 0; c < g->nchannels; c++) { free(g->keys[c].values); free(g->keys[c].nulls); free(g->keys[c].offsets); }
    free(g->keys); free(g->group_address_by_hash); free(g->group_ids_by_hash); free(g->raw_hash_by_hash_pos);
    free(g->raw_hash_by_group); free(g->first_row_by_group); free(g);
}

static void key_store_as_column(const key_store_col *k, o_column *out)
{
    out->type = k->type; out->n = k->n; out->values = k->values; out->nulls = k->nulls; out->offsets = k->offsets;
}

static void key_store_append(key_store_col *k, const o_column *src, int32_t pos)
{
    if (k->n == k->cap) {
        int32_t ncap = k->cap ? k->cap * 2 : 64;
        k->nulls = (uint8_t *)realloc(k->nulls, (size_t)ncap);
        if (k->type == O_VARCHAR) {
            k->offsets = (int32_t *)realloc(k->offsets, ((size_t)ncap + 1) * 4);
            if (k->cap == 0) k->offsets[0] = 0;
        }
        else {
            k->values = (uint8_t *)realloc(k->values, (size_t)ncap * (size_t)type_width(k->type));
        }
        k->cap = ncap;
    }
    int isnull = cell_is_null(src, pos);
    k->nulls[k->n] = (uint8_t)isnull;
    if (k->type == O_VARCHAR) {
        int32_t len = isnull ? 0 : src->offsets[pos + 1] - src->offsets[pos];
        int32_t at = k->offsets[k->n];
        if ((int64_t)at + len > k->values_cap_bytes) {
            int64_t nb = k->values_cap_bytes ? k->values_cap_bytes * 2 : 1024;
            while (nb < (int64_t)at + len) nb *= 2;
            k->values = (uint8_t *)realloc(k->values, (size_t)nb);
            k->values_cap_bytes = nb;
        }
        if (len) memcpy(k->values + at, (const uint8_t *)src->values + src->offsets[pos], (size_t)len);
        k->offsets[k->n + 1] = at + len;
    }
    else {
        int32_t w = type_width(k->type);
        if (isnull) memset(k->values + (size_t)k->n * w, 0, (size_t)w);
        else memcpy(k->values + (size_t)k->n * w, (const uint8_t *)src->values + (size_t)pos * w, (size_t)w);
    }
    k->n++;
}

/* tryRehash :364-424 -- walks the OLD table in slot order (:391-413) */
static int32_t multi_try_rehash(o_multi_gbh *g)
{
    int64_t new_cap_l = (int64_t)g->hash_capacity * 2;
    if (new_cap_l > 0x7fffffffLL) return O_ERR_INSUFFICIENT_RESOURCES;
    int32_t new_cap = (int32_t)new_cap_l, new_mask = new_cap - 1;
    int64_t *nk = (int64_t *)malloc((size_t)new_cap * 8);
    for (int32_t i = 0; i < new_cap; i++) nk[i] = -1;
    uint8_t *nh = (uint8_t *)calloc((size_t)new_cap, 1);
    int32_t *nv = (int32_t *)calloc((size_t)new_cap, 4);
    int32_t old_index = 0;
    for (int32_t gid = 0; gid < g->next_group_id; gid++) {
        while (g->group_address_by_hash[old_index] == -1) old_index++;
        int64_t address = g->group_address_by_hash[old_index];
        int64_t raw_hash = g->raw_hash_by_group[address]; /* hashPosition(address) :426-434 */
        int32_t pos = (int32_t)(o_murmur3_fmix((uint64_t)raw_hash) & (uint64_t)new_mask);
        while (nk[pos] != -1) { pos = (pos + 1) & new_mask; g->hash_collisions++; }
        nk[pos] = address; nh[pos] = (uint8_t)raw_hash; nv[pos] = g->group_ids_by_hash[old_index];
        old_index++;
    }
    free(g->group_address_by_hash); free(g->raw_hash_by_hash_pos); free(g->group_ids_by_hash);
    g->group_address_by_hash = nk; g->raw_hash_by_hash_pos = nh; g->group_ids_by_hash = nv;
    g->mask = new_mask; g->hash_capacity = new_cap; g->max_fill = o_calculate_max_fill(new_cap);
    g->rehash_count++;
    return O_OK;
}

/* positionNotDistinctFromCurrentRow :441-447 : 1-byte tag prefilter, then key compare (null == null) */
static int multi_row_matches(const o_multi_gbh *g, int64_t address, int32_t hash_pos, const o_column *cols, int32_t pos, uint8_t tag)
{
    if (g->raw_hash_by_hash_pos[hash_pos] != tag) return 0;
    for (int32_t c = 0; c < g->nchannels; c++) {
        o_column kc;
        key_store_as_column(&g->keys[c], &kc);
        if (!cells_not_distinct(&kc, (int32_t)address, &cols[c], pos)) return 0;
    }
    return 1;
}

/* putIfAbsent :281-304 + addNewGroup :306-342 */
static int32_t multi_put_if_absent(o_multi_gbh *g, const o_column *cols, int32_t pos, int64_t raw_hash, int32_t *err)
{
    int32_t hp = (int32_t)(o_murmur3_fmix((uint64_t)raw_hash) & (uint64_t)g->mask);
    while (g->group_address_by_hash[hp] != -1) {
        if (multi_row_matches(g, g->group_address_by_hash[hp], hp, cols, pos, (uint8_t)raw_hash)) {
            return g->group_ids_by_hash[hp];
        }
        hp = (hp + 1) & g->mask;
        g->hash_collisions++;
    }
    for (int32_t c = 0; c < g->nchannels; c++) key_store_append(&g->keys[c], &cols[c], pos);
    int32_t gid = g->next_group_id++;
    if (gid >= g->by_group_cap) {
        int32_t ncap = g->by_group_cap ? g->by_group_cap * 2 : 64;
        g->raw_hash_by_group = (int64_t *)realloc(g->raw_hash_by_group, (size_t)ncap * 8);
        g->first_row_by_group = (int64_t *)realloc(g->first_row_by_group, (size_t)ncap * 8);
        g->by_group_cap = ncap;
    }
    g->raw_hash_by_group[gid] = raw_hash;
    g->first_row_by_group[gid] = g->rows_seen + pos;
    g->group_address_by_hash[hp] = gid; /* address == row in the key store == group id */
    g->raw_hash_by_hash_pos[hp] = (uint8_t)raw_hash;
    g->group_ids_by_hash[hp] = gid;
    if (g->next_group_id >= g->max_fill) {
        int32_t rc = multi_try_rehash(g);
        if (rc != O_OK) *err = rc;
    }
    return gid;
}

int32_t o_multi_gbh_get_group_ids(o_multi_gbh *g, const o_column *cols, const int64_t *hashes, int32_t n, int64_t *out)
{
    int32_t err = O_OK;
    if (g->next_group_id >= g->max_fill) {
        int32_t rc = multi_try_rehash(g);
        if (rc != O_OK) return rc;
    }
    for (int32_t i = 0; i < n; i++) {
        int64_t raw_hash;
        if (hashes) raw_hash = hashes[i];                     /* PrecomputedHashGenerator.java:31-35 */
        else {
            raw_hash = 0;
            for (int32_t c = 0; c < g->nchannels; c++) raw_hash = o_combine_hash(raw_hash, o_hash_cell(&cols[c], i));
        }
        int32_t gid = multi_put_if_absent(g, cols, i, raw_hash, &err);
        if (out) out[i] = gid;
        if (err != O_OK) return err;
    }
    g->rows_seen += n;
    return O_OK;
}

int32_t o_multi_gbh_contains(const o_multi_gbh *g, const o_column *cols, int32_t pos, int64_t raw_hash)
{
    /* :258-272 */
    int32_t hp = (int32_t)(o_murmur3_fmix((uint64_t)raw_hash) & (uint64_t)g->mask);
    while (g->group_address_by_hash[hp] != -1) {
        if (multi_row_matches(g, g->group_address_by_hash[hp], hp, cols, pos, (uint8_t)raw_hash)) return 1;
        hp = (hp + 1) & g->mask;
    }
    return 0;
}

int32_t o_multi_gbh_group_count(const o_multi_gbh *g) { return g->next_group_id; }
int32_t o_multi_gbh_capacity(const o_multi_gbh *g) { return g->hash_capacity; }
int32_t o_multi_gbh_rehash_count(const o_multi_gbh *g) { return g->rehash_count; }
void o_multi_gbh_group_rows(const o_multi_gbh *g, int64_t *first_rows, int64_t *raw_hashes)
{
    for (int32_t i = 0; i < g->next_group_id; i++) {
        if (first_rows) first_rows[i] = g->first_row_by_group[i];
        if (raw_hashes) raw_hashes[i] = g->raw_hash_by_group[i];
    }
}

/* ------------------------------------------------------------------------------------------------
 * accumulators, Java order
 * ------------------------------------------------------------------------------------------------ */
void o_agg_double_sum(const int64_t *gids, const double *v, const uint8_t *nulls, const uint8_t *mask,
                      int32_t n, int64_t *counts, double *sums)
{
    for (int32_t i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        if (nulls && nulls[i]) continue;
        int64_t g = gids ? gids[i] : 0;
        counts[g] += 1;
        sums[g] += v[i];
    }
}

void o_agg_long_avg(const int64_t *gids, const int64_t *v, const uint8_t *nulls, const uint8_t *mask,
                    int32_t n, int64_t *counts, double *sums)
{
    for (int32_t i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        if (nulls && nulls[i]) continue;
        int64_t g = gids ? gids[i] : 0;
        counts[g] += 1;
        sums[g] += (double)v[i];
    }
}

int32_t o_agg_long_sum(const int64_t *gids, const int64_t *v, const uint8_t *nulls, const uint8_t *mask,
                       int32_t n, int64_t *counts, int64_t *sums)
{
    for (int32_t i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        if (nulls && nulls[i]) continue;
        int64_t g = gids ? gids[i] : 0;
        counts[g] += 1;
        int64_t r;
        if (__builtin_add_overflow(sums[g], v[i], &r)) return O_ERR_NUMERIC_VALUE_OUT_OF_RANGE; /* M/type/BigintOperators.java:47-57 */
        sums[g] = r;
    }
    return O_OK;
}

/* min(bigint) / max(bigint): M/operator/aggregation/AbstractMinMaxAggregationFunction.java:233-236 (input) -> :274-289 compareAndUpdateState on a
 * NullableLongState: the first value is taken, then every value the comparison prefers (MinAggregationFunction: value < state,
 * MaxAggregationFunction: value > state).  counts[g] > 0 <=> the state is not null; values[g] is meaningful only then. */
void o_agg_long_minmax(const int64_t *gids, const int64_t *v, const uint8_t *nulls, const uint8_t *mask, int32_t n, int32_t is_min,
                       int64_t *counts, int64_t *values)
{
    for (int32_t i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        if (nulls && nulls[i]) continue;
        int64_t g = gids ? gids[i] : 0;
        if (counts[g] == 0) values[g] = v[i];                                   /* state.isNull(): setNull(false), setLong(value) */
        else if (is_min ? v[i] < values[g] : v[i] > values[g]) values[g] = v[i];
        counts[g] += 1;
    }
}

/* min(double) / max(double): AbstractMinMaxAggregationFunction.java:227-230 -> :291-306 on a NullableDoubleState.  min: the comparison is
 * Double.compare(value, state) < 0 (MinMaxCompare.getMinMaxCompare over DoubleType.java:194-198 comparisonOperator): a total order with
 * -0.0 < +0.0 and NaN above everything; max: M/util/MinMaxCompare.java maxDouble = (value > state) || isNaN(state) -- plain IEEE >, a NaN
 * state gives way to whatever comes next. */
static int o_double_compare(double a, double b)   /* java.lang.Double.compare */
{
    if (a < b) return -1;
    if (a > b) return 1;
    int64_t x, y;
    memcpy(&x, &a, 8);
    memcpy(&y, &b, 8);
    if (a != a) x = 0x7ff8000000000000LL;   /* doubleToLongBits: every NaN is the canonical one */
    if (b != b) y = 0x7ff8000000000000LL;
    return x == y ? 0 : (x < y ? -1 : 1);
}

void o_agg_double_minmax(const int64_t *gids, const double *v, const uint8_t *nulls, const uint8_t *mask, int32_t n, int32_t is_min,
                         int64_t *counts, double *values)
{
    for (int32_t i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        if (nulls && nulls[i]) continue;
        int64_t g = gids ? gids[i] : 0;
        if (counts[g] == 0) values[g] = v[i];
        else if (is_min ? o_double_compare(v[i], values[g]) < 0 : (v[i] > values[g] || values[g] != values[g])) values[g] = v[i];
        counts[g] += 1;
    }
}

void o_agg_count(const int64_t *gids, const uint8_t *nulls, const uint8_t *mask, int32_t n, int64_t *counts)
{
    for (int32_t i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        if (nulls && nulls[i]) continue;
        counts[gids ? gids[i] : 0] += 1;
    }
}

/* Shewchuk's exact partials sum (as in CPython's math.fsum): result is the correctly rounded exact sum.
 * Not part of the reference; it is the yard-stick for the product's "exact sum" DOUBLE policy. */
typedef struct { double *p; int32_t n, cap; double special; int has_special; } msum_t;

static void msum_add(msum_t *m, double x)
{
    if (!isfinite(x)) { m->special += x; m->has_special = 1; return; }
    int32_t i = 0;
    for (int32_t j = 0; j < m->n; j++) {
        double y = m->p[j];
        if (fabs(x) < fabs(y)) { double t = x; x = y; y = t; }
        double hi = x + y;
        double lo = y - (hi - x);
        if (lo != 0.0) m->p[i++] = lo;
        x = hi;
    }
    if (i >= m->cap) { m->cap = m->cap ? m->cap * 2 : 32; m->p = (double *)realloc(m->p, (size_t)m->cap * 8); }
    m->p[i] = x;
    m->n = i + 1;
}

static double msum_result(msum_t *m)
{
    if (m->has_special) return m->special;
    int32_t n = m->n;
    double hi = 0.0, lo = 0.0;
    if (n > 0) {
        hi = m->p[--n];
        while (n > 0) {
            double x = hi, y = m->p[--n];
            hi = x + y;
            double yr = hi - x;
            lo = y - yr;
            if (lo != 0.0) break;
        }
        /* round-half-even correction (CPython fsum) */
        if (n > 0 && ((lo < 0.0 && m->p[n - 1] < 0.0) || (lo > 0.0 && m->p[n - 1] > 0.0))) {
            double y = lo * 2.0, x = hi + y, yr = x - hi;
            if (y == yr) hi = x;
        }
    }
    return hi;
}

double o_exact_sum(const double *v, int64_t n)
{
    msum_t m = {0};
    for (int64_t i = 0; i < n; i++) msum_add(&m, v[i]);
    double r = msum_result(&m);
    free(m.p);
    return r;
}

void o_agg_double_sum_exact(const int64_t *gids, const double *v, const uint8_t *nulls, const uint8_t *mask,
                            int64_t n, int32_t ngroups, int64_t *counts, double *sums)
{
    msum_t *ms = (msum_t *)calloc((size_t)ngroups, sizeof(msum_t));
    for (int64_t i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        if (nulls && nulls[i]) continue;
        int64_t g = gids ? gids[i] : 0;
        counts[g] += 1;
        msum_add(&ms[g], v[i]);
    }
    for (int32_t g = 0; g < ngroups; g++) { sums[g] = msum_result(&ms[g]); free(ms[g].p); }
    free(ms);
}

/* ------------------------------------------------------------------------------------------------
 * join: PagesHash / ArrayPositionLinks / JoinHash / PageJoiner
 * ------------------------------------------------------------------------------------------------ */
struct o_pages_hash {
    int32_t ncols, n;
    o_column *cols;            /* borrowed key columns of the build side (PagesIndex channels) */
    int32_t hash_size, mask;
    int32_t *key;              /* :42 */
    uint8_t *position_to_hashes; /* :48 */
    int32_t *links;            /* ArrayPositionLinks.positionLinks :34-41 */
    int32_t link_count;
    int64_t hash_collisions;
    int precomputed;
};

static int build_row_has_null(const o_pages_hash *h, int32_t pos)
{
    /* isPositionNull :180-187 -> PagesHashStrategy.isPositionNull: any join channel null */
    for (int32_t c = 0; c < h->ncols; c++) if (cell_is_null(&h->cols[c], pos)) return 1;
    return 0;
}

o_pages_hash *o_pages_hash_new(const o_column *key_cols, int32_t ncols, int32_t n, const int64_t *hashes)
{
    o_pages_hash *h = (o_pages_hash *)calloc(1, sizeof(*h));
    h->ncols = ncols; h->n = n;
    h->cols = (o_column *)malloc(sizeof(o_column) * (size_t)ncols);
    memcpy(h->cols, key_cols, sizeof(o_column) * (size_t)ncols);
    h->hash_size = o_array_size(n, 0.75f);          /* :63 */
    h->mask = h->hash_size - 1;
    h->key = (int32_t *)malloc((size_t)h->hash_size * 4);
    for (int32_t i = 0; i < h->hash_size; i++) h->key[i] = -1;
    h->position_to_hashes = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    h->links = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * 4);
    for (int32_t i = 0; i < n; i++) h->links[i] = -1;
    h->precomputed = hashes != NULL;
    /* the 4096-row stepping (:72-84) only batches the hash extraction; order of insertion is position order */
    for (int32_t pos = 0; pos < n; pos++) {
        int64_t hash;
        if (hashes) hash = hashes[pos];
        else {
            hash = 0;
            for (int32_t c = 0; c < ncols; c++) hash = o_combine_hash(hash, o_hash_cell(&h->cols[c], pos));
        }
        h->position_to_hashes[pos] = (uint8_t)hash;
        if (build_row_has_null(h, pos)) continue;   /* :94-96 */
        int32_t real_position = pos;
        int32_t p = (int32_t)(o_murmur3_fmix((uint64_t)hash) & (uint64_t)h->mask);
        while (h->key[p] != -1) {
            int32_t current_key = h->key[p];
            int eq = ((uint8_t)hash) == h->position_to_hashes[current_key];
            if (eq) {
                for (int32_t c = 0; c < ncols && eq; c++) eq = cells_equal_nonnull(&h->cols[c], current_key, &h->cols[c], real_position);
            }
            if (eq) {
                /* ArrayPositionLinks.link(left=new, right=existing): links[left] = right; return left (:45-50) */
                h->links[real_position] = current_key;
                h->link_count++;
                break;
            }
            p = (p + 1) & h->mask;
            h->hash_collisions++;
        }
        h->key[p] = real_position;
    }
    return h;
}

void o_pages_hash_free(o_pages_hash *h)
{
    if (!h) return;
    free(h->cols); free(h->key); free(h->position_to_hashes); free(h->links); free(h);
}

int32_t o_pages_hash_size(const o_pages_hash *h) { return h->hash_size; }
int32_t o_pages_hash_link_count(const o_pages_hash *h) { return h->link_count; }
const int32_t *o_pages_hash_links(const o_pages_hash *h) { return h->links; }
const int32_t *o_pages_hash_keys(const o_pages_hash *h) { return h->key; }
int64_t o_pages_hash_collisions(const o_pages_hash *h) { return h->hash_collisions; }

int32_t o_pages_hash_get_address_index(const o_pages_hash *h, const o_column *probe_cols, int32_t pos, int64_t raw_hash)
{
    int32_t p = (int32_t)(o_murmur3_fmix((uint64_t)raw_hash) & (uint64_t)h->mask);
    while (h->key[p] != -1) {
        int32_t k = h->key[p];
        /* positionEqualsCurrentRowIgnoreNulls :198-209 */
        if (h->position_to_hashes[k] == (uint8_t)raw_hash) {
            int eq = 1;
            for (int32_t c = 0; c < h->ncols && eq; c++) eq = cells_equal_nonnull(&h->cols[c], k, &probe_cols[c], pos);
            if (eq) return k;
        }
        p = (p + 1) & h->mask;
    }
    return -1;
}

int64_t o_join_probe(const o_pages_hash *h, const o_column *probe_cols, int32_t n_probe, const int64_t *hashes,
                     int32_t probe_outer, int32_t *out_probe, int32_t *out_build, int64_t cap)
{
    int64_t out = 0;
    for (int32_t pos = 0; pos < n_probe; pos++) {
        /* JoinProbe.getCurrentJoinPosition :87-97 : any null probe key -> -1 */
        int has_null = 0;
        for (int32_t c = 0; c < h->ncols; c++) if (cell_is_null(&probe_cols[c], pos)) has_null = 1;
        int32_t jp = -1;
        if (!has_null) {
            int64_t raw_hash;
            if (hashes) raw_hash = hashes[pos];
            else {
                raw_hash = 0; /* hashRow: JoinCompiler.java:449-477 */
                for (int32_t c = 0; c < h->ncols; c++) raw_hash = o_combine_hash(raw_hash, o_hash_cell(&probe_cols[c], pos));
            }
            jp = o_pages_hash_get_address_index(h, probe_cols, pos, raw_hash);
        }
        int produced = 0;
        /* joinCurrentPosition :328-347 */
        while (jp >= 0) {
            if (out < cap) { out_probe[out] = pos; out_build[out] = jp; }
            out++;
            produced = 1;
            jp = h->links[jp];  /* JoinHash.getNextJoinPosition :107-113 */
        }
        if (!produced && probe_outer) { /* outerJoinCurrentPosition :354-361 */
            if (out < cap) { out_probe[out] = pos; out_build[out] = -1; }
            out++;
        }
    }
    return out <= cap ? out : -out;
}

/* ------------------------------------------------------------------------------------------------
 * RowExpression interpreter (filter / project)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int is_null;
    int64_t i;       /* bigint / integer / date / boolean */
    double d;
    const uint8_t *s; int32_t slen;
} o_val;

typedef struct {
    const o_expr_node *nodes;
    const char *pool;
    const o_column *cols;
    int32_t err;
} o_eval_ctx;

static o_val eval_node(o_eval_ctx *cx, int32_t idx, int32_t row);

static int is_int_type(int32_t t) { return t == O_BIGINT || t == O_INTEGER || t == O_DATE; }

static int cmp_vals(int32_t type, const o_val *a, const o_val *b, int op)
{
    /* comparison operators: S/type/AbstractLongType.java:156-166, DoubleType (plain IEEE), VARCHAR Slice.compareTo (unsigned bytes) */
    if (type == O_DOUBLE) {
        switch (op) {
        case O_OP_EQUAL: return a->d == b->d;
        case O_OP_NOT_EQUAL: return a->d != b->d;
        case O_OP_LESS_THAN: return a->d < b->d;
        case O_OP_LESS_THAN_OR_EQUAL: return a->d <= b->d;
        case O_OP_GREATER_THAN: return a->d > b->d;
        default: return a->d >= b->d;
        }
    }
    int64_t c;
    if (type == O_VARCHAR) {
        int32_t m = a->slen < b->slen ? a->slen : b->slen;
        int r = m ? memcmp(a->s, b->s, (size_t)m) : 0;
        c = r != 0 ? r : (a->slen - b->slen);
    }
    else {
        c = (a->i > b->i) - (a->i < b->i);
    }
    switch (op) {
    case O_OP_EQUAL: return c == 0;
    case O_OP_NOT_EQUAL: return c != 0;
    case O_OP_LESS_THAN: return c < 0;
    case O_OP_LESS_THAN_OR_EQUAL: return c <= 0;
    case O_OP_GREATER_THAN: return c > 0;
    default: return c >= 0;
    }
}

/* java.lang.Math.round(double) as a double: floor(x + 1/2) computed without the rounding error of the addition (values of
 * magnitude >= 2^52 are integers already); the saturation of the long result is left to the caller's range check */
static double java_math_round(double x)
{
    if (!(fabs(x) < 4503599627370496.0)) return x;
    double f = floor(x);
    return (x - f >= 0.5) ? f + 1.0 : f;
}

/* guava DoubleMath.roundToLong(x, HALF_UP) as a double: nearest integer, ties away from zero */
static double java_half_up(double x)
{
    if (!(fabs(x) < 4503599627370496.0)) return x;
    double t = trunc(x);
    return (fabs(x - t) >= 0.5) ? t + (x < 0 ? -1.0 : 1.0) : t;
}

static o_val eval_call(o_eval_ctx *cx, const o_expr_node *nd, int32_t row)
{
    o_val r; memset(&r, 0, sizeof(r));
    o_val a[3];
    /* BytecodeUtils.generateFullInvocation: arguments are evaluated in order; the first null argument
     * jumps to the end (later arguments are NOT evaluated) and the call yields null. */
    for (int32_t k = 0; k < nd->n_args; k++) {
        a[k] = eval_node(cx, nd->args[k], row);
        if (cx->err) return r;
        if (a[k].is_null) { r.is_null = 1; return r; }
    }
    int32_t at = cx->nodes[nd->args[0]].type;
    switch (nd->op) {
    case O_OP_ADD: case O_OP_SUBTRACT: case O_OP_MULTIPLY: case O_OP_DIVIDE: case O_OP_MODULUS:
        if (nd->type == O_DOUBLE) {
            /* M/type/DoubleOperators.java:59-95 plain IEEE */
            switch (nd->op) {
            case O_OP_ADD: r.d = a[0].d + a[1].d; break;
            case O_OP_SUBTRACT: r.d = a[0].d - a[1].d; break;
            case O_OP_MULTIPLY: r.d = a[0].d * a[1].d; break;
            case O_OP_DIVIDE: r.d = a[0].d / a[1].d; break;
            default: r.d = fmod(a[0].d, a[1].d); break;
            }
        }
        else if (nd->type == O_BIGINT) {
            /* M/type/BigintOperators.java:47-113 checked arithmetic */
            int64_t x = a[0].i, y = a[1].i, z = 0; int ov = 0;
            switch (nd->op) {
            case O_OP_ADD: ov = __builtin_add_overflow(x, y, &z); break;
            case O_OP_SUBTRACT: ov = __builtin_sub_overflow(x, y, &z); break;
            case O_OP_MULTIPLY: ov = __builtin_mul_overflow(x, y, &z); break;
            case O_OP_DIVIDE:
                if (y == 0) { cx->err = O_ERR_DIVISION_BY_ZERO; return r; }
                if (x == INT64_MIN && y == -1) { cx->err = O_ERR_NUMERIC_VALUE_OUT_OF_RANGE; return r; }
                z = x / y; break;
            default:
                if (y == 0) { cx->err = O_ERR_DIVISION_BY_ZERO; return r; }
                z = (y == -1) ? 0 : x % y; break;
            }
            if (ov) { cx->err = O_ERR_NUMERIC_VALUE_OUT_OF_RANGE; return r; }
            r.i = z;
        }
        else { /* INTEGER: M/type/IntegerOperators.java checked 32-bit arithmetic */
            int32_t x = (int32_t)a[0].i, y = (int32_t)a[1].i, z = 0; int ov = 0;
            switch (nd->op) {
            case O_OP_ADD: ov = __builtin_add_overflow(x, y, &z); break;
            case O_OP_SUBTRACT: ov = __builtin_sub_overflow(x, y, &z); break;
            case O_OP_MULTIPLY: ov = __builtin_mul_overflow(x, y, &z); break;
            case O_OP_DIVIDE:
                if (y == 0) { cx->err = O_ERR_DIVISION_BY_ZERO; return r; }
                if (x == INT32_MIN && y == -1) { cx->err = O_ERR_NUMERIC_VALUE_OUT_OF_RANGE; return r; }
                z = x / y; break;
            default:
                if (y == 0) { cx->err = O_ERR_DIVISION_BY_ZERO; return r; }
                z = (y == -1) ? 0 : x % y; break;
            }
            if (ov) { cx->err = O_ERR_NUMERIC_VALUE_OUT_OF_RANGE; return r; }
            r.i = z;
        }
        return r;
    case O_OP_NEGATE:
        if (nd->type == O_DOUBLE) r.d = -a[0].d;
        else if (nd->type == O_BIGINT) {
            if (a[0].i == INT64_MIN) { cx->err = O_ERR_NUMERIC_VALUE_OUT_OF_RANGE; return r; }
            r.i = -a[0].i;
        }
        else {
            if ((int32_t)a[0].i == INT32_MIN) { cx->err = O_ERR_NUMERIC_VALUE_OUT_OF_RANGE; return r; }
            r.i = -(int32_t)a[0].i;
        }
        return r;
    case O_OP_EQUAL: case O_OP_NOT_EQUAL: case O_OP_LESS_THAN: case O_OP_LESS_THAN_OR_EQUAL:
    case O_OP_GREATER_THAN: case O_OP_GREATER_THAN_OR_EQUAL:
        if (at == O_BOOLEAN) { a[0].i = a[0].i != 0; a[1].i = a[1].i != 0; }
        r.i = cmp_vals(at, &a[0], &a[1], nd->op);
        return r;
    case O_OP_NOT:
        r.i = !a[0].i;
        return r;
    case O_OP_CAST:
        if (nd->type == at) return a[0];
        if (nd->type == O_DOUBLE && is_int_type(at)) { r.d = (double)a[0].i; return r; }
        if (nd->type == O_BIGINT && (at == O_INTEGER)) { r.i = a[0].i; return r; }
        if (nd->type == O_INTEGER && at == O_BIGINT) {
            if (a[0].i > INT32_MAX || a[0].i < INT32_MIN) { cx->err = O_ERR_NUMERIC_VALUE_OUT_OF_RANGE; return r; }
            r.i = a[0].i; return r;
        }
        /* castToBoolean: M/type/BigintOperators.java:115-120, IntegerOperators.java:146-151, DoubleOperators.java:101-106 (NaN != 0 is true) */
        if (nd->type == O_BOOLEAN && (at == O_BIGINT || at == O_INTEGER)) { r.i = a[0].i != 0; return r; }
        if (nd->type == O_BOOLEAN && at == O_DOUBLE) { r.i = a[0].d != 0; return r; }
        /* M/type/BooleanOperators.java:37-63 */
        if (at == O_BOOLEAN && (nd->type == O_BIGINT || nd->type == O_INTEGER)) { r.i = a[0].i ? 1 : 0; return r; }
        if (at == O_BOOLEAN && nd->type == O_DOUBLE) { r.d = a[0].i ? 1.0 : 0.0; return r; }
        if (nd->type == O_BIGINT && at == O_DOUBLE) {
            /* DoubleOperators.castToLong :153-163 = guava DoubleMath.roundToLong(value, HALF_UP): nearest, ties away from zero;
             * NaN, infinities and results outside [-2^63, 2^63) -> INVALID_CAST_ARGUMENT */
            double x = a[0].d;
            if (x != x || x - x != 0) { cx->err = O_ERR_INVALID_CAST_ARGUMENT; return r; }
            double z = java_half_up(x);
            if (!(z >= -9223372036854775808.0 && z < 9223372036854775808.0)) { cx->err = O_ERR_INVALID_CAST_ARGUMENT; return r; }
            r.i = (int64_t)z; return r;
        }
        if (nd->type == O_INTEGER && at == O_DOUBLE) {
            /* DoubleOperators.castToInteger :108-121: NaN -> INVALID_CAST_ARGUMENT; toIntExact((long) MathFunctions.round(value))
             * (MathFunctions.java:821-833: -(Math.round(-x)) for x < 0, infinities pass through; the (long) conversion saturates) */
            double x = a[0].d;
            if (x != x) { cx->err = O_ERR_INVALID_CAST_ARGUMENT; return r; }
            double z = (x - x != 0) ? x : (x < 0 ? -java_math_round(-x) : java_math_round(x));
            if (!(z >= -2147483648.0 && z <= 2147483647.0)) { cx->err = O_ERR_NUMERIC_VALUE_OUT_OF_RANGE; return r; }
            r.i = (int64_t)z; return r;
        }
        cx->err = O_ERR_INVALID;
        return r;
    default:
        cx->err = O_ERR_INVALID;
        return r;
    }
}

static o_val eval_special(o_eval_ctx *cx, const o_expr_node *nd, int32_t row)
{
    o_val r; memset(&r, 0, sizeof(r));
    switch (nd->op) {
    case O_SF_AND: {
        /* AndCodeGenerator.java:44-105: left false -> false (right NOT evaluated); else evaluate right */
        o_val l = eval_node(cx, nd->args[0], row);
        if (cx->err) return r;
        if (!l.is_null && !l.i) { r.i = 0; return r; }
        o_val rt = eval_node(cx, nd->args[1], row);
        if (cx->err) return r;
        if (rt.is_null) { r.is_null = 1; return r; }
        if (!rt.i) { r.i = 0; return r; }
        r.is_null = l.is_null; r.i = 1;
        return r;
    }
    case O_SF_OR: {
        o_val l = eval_node(cx, nd->args[0], row);
        if (cx->err) return r;
        if (!l.is_null && l.i) { r.i = 1; return r; }
        o_val rt = eval_node(cx, nd->args[1], row);
        if (cx->err) return r;
        if (rt.is_null) { r.is_null = 1; return r; }
        if (rt.i) { r.i = 1; return r; }
        r.is_null = l.is_null; r.i = 0;
        return r;
    }
    case O_SF_IF: {
        /* IfCodeGenerator: condition null or false -> else branch */
        o_val c = eval_node(cx, nd->args[0], row);
        if (cx->err) return r;
        if (!c.is_null && c.i) return eval_node(cx, nd->args[1], row);
        return eval_node(cx, nd->args[2], row);
    }
    case O_SF_IS_NULL: {
        o_val v = eval_node(cx, nd->args[0], row);
        if (cx->err) return r;
        r.i = v.is_null;
        return r;
    }
    case O_SF_COALESCE: {
        for (int32_t k = 0; k < nd->n_args; k++) {
            o_val v = eval_node(cx, nd->args[k], row);
            if (cx->err) return r;
            if (!v.is_null) return v;
        }
        r.is_null = 1;
        return r;
    }
    case O_SF_BETWEEN: {
        /* BetweenCodeGenerator: value >= min AND value <= max with AND's three-valued logic */
        o_val v = eval_node(cx, nd->args[0], row);
        if (cx->err) return r;
        o_val lo = eval_node(cx, nd->args[1], row);
        if (cx->err) return r;
        int32_t t = cx->nodes[nd->args[0]].type;
        int l_null = v.is_null || lo.is_null, l_val = 0;
        if (!l_null) l_val = cmp_vals(t, &v, &lo, O_OP_GREATER_THAN_OR_EQUAL);
        if (!l_null && !l_val) { r.i = 0; return r; }
        o_val hi = eval_node(cx, nd->args[2], row);
        if (cx->err) return r;
        int r_null = v.is_null || hi.is_null, r_val = 0;
        if (!r_null) r_val = cmp_vals(t, &v, &hi, O_OP_LESS_THAN_OR_EQUAL);
        if (r_null) { r.is_null = 1; return r; }
        if (!r_val) { r.i = 0; return r; }
        r.is_null = l_null; r.i = 1;
        return r;
    }
    default:
        cx->err = O_ERR_INVALID;
        return r;
    }
}

static o_val eval_node(o_eval_ctx *cx, int32_t idx, int32_t row)
{
    const o_expr_node *nd = &cx->nodes[idx];
    o_val r; memset(&r, 0, sizeof(r));
    switch (nd->kind) {
    case O_EX_INPUT: {
        const o_column *c = &cx->cols[nd->op];
        if (cell_is_null(c, row)) { r.is_null = 1; return r; }
        switch (c->type) {
        case O_BIGINT: r.i = ((const int64_t *)c->values)[row]; break;
        case O_INTEGER: case O_DATE: r.i = ((const int32_t *)c->values)[row]; break;
        case O_DOUBLE: r.d = ((const double *)c->values)[row]; break;
        case O_BOOLEAN: r.i = ((const uint8_t *)c->values)[row] != 0; break;
        case O_VARCHAR: r.s = (const uint8_t *)c->values + c->offsets[row]; r.slen = c->offsets[row + 1] - c->offsets[row]; break;
        default: cx->err = O_ERR_INVALID;
        }
        return r;
    }
    case O_EX_CONST:
        if (nd->is_null) { r.is_null = 1; return r; }
        if (nd->type == O_DOUBLE) r.d = nd->dval;
        else if (nd->type == O_VARCHAR) { r.s = (const uint8_t *)cx->pool + nd->ival; r.slen = nd->slen; }
        else r.i = nd->ival;
        return r;
    case O_EX_CALL: return eval_call(cx, nd, row);
    case O_EX_SPECIAL: return eval_special(cx, nd, row);
    default: cx->err = O_ERR_INVALID; return r;
    }
}

int32_t o_filter(const o_expr_node *nodes, int32_t root, const char *pool, const o_column *cols, int32_t n,
                 int32_t *positions, int32_t *err_row)
{
    o_eval_ctx cx = { nodes, pool, cols, 0 };
    int32_t k = 0;
    for (int32_t r = 0; r < n; r++) {
        /* generated filter: selected = !wasNull && value (PageFunctionCompiler.java:502-544) */
        o_val v = eval_node(&cx, root, r);
        if (cx.err) { if (err_row) *err_row = r; return cx.err; }
        if (!v.is_null && v.i) positions[k++] = r;
    }
    return k;
}

int32_t o_project(const o_expr_node *nodes, int32_t root, const char *pool, const o_column *cols,
                  const int32_t *positions, int32_t n_sel, void *out_values, uint8_t *out_nulls, int32_t *err_row)
{
    o_eval_ctx cx = { nodes, pool, cols, 0 };
    int32_t t = nodes[root].type;
    for (int32_t k = 0; k < n_sel; k++) {
        int32_t r = positions ? positions[k] : k;
        o_val v = eval_node(&cx, root, r);
        if (cx.err) { if (err_row) *err_row = r; return cx.err; }
        out_nulls[k] = (uint8_t)v.is_null;
        switch (t) {
        case O_BIGINT: ((int64_t *)out_values)[k] = v.is_null ? 0 : v.i; break;
        case O_INTEGER: case O_DATE: ((int32_t *)out_values)[k] = v.is_null ? 0 : (int32_t)v.i; break;
        case O_DOUBLE: ((double *)out_values)[k] = v.is_null ? 0.0 : v.d; break;
        case O_BOOLEAN: ((uint8_t *)out_values)[k] = v.is_null ? 0 : (uint8_t)(v.i != 0); break;
        default: return O_ERR_INVALID;
        }
    }
    return O_OK;
}

/* ================================================================================================================== */
/* TopN                                                                                                                */
/* ================================================================================================================== */

/* the type's COMPARISON operator on two non-null cells (Long.compare: S/type/AbstractLongType.java; Integer.compare:
 * AbstractIntType.java; Double.compare: S/type/DoubleType.java:194-197; Boolean.compare; Slice.compareTo for VARCHAR) */
static int compare_cells(const o_column *c, int32_t a, int32_t b)
{
    switch (c->type) {
    case O_BIGINT: { int64_t x = ((const int64_t *)c->values)[a], y = ((const int64_t *)c->values)[b]; return x < y ? -1 : (x > y ? 1 : 0); }
    case O_INTEGER: case O_DATE: { int32_t x = ((const int32_t *)c->values)[a], y = ((const int32_t *)c->values)[b]; return x < y ? -1 : (x > y ? 1 : 0); }
    case O_BOOLEAN: { int x = ((const uint8_t *)c->values)[a] != 0, y = ((const uint8_t *)c->values)[b] != 0; return x - y; }
    case O_DOUBLE: {
        /* Double.compare: numeric order, then -0.0 < 0.0, NaN equal to itself and greater than everything else */
        double x = ((const double *)c->values)[a], y = ((const double *)c->values)[b];
        if (x < y) return -1;
        if (x > y) return 1;
        int64_t bx, by;
        if (x != x) bx = 0x7ff8000000000000LL; else memcpy(&bx, &x, 8);
        if (y != y) by = 0x7ff8000000000000LL; else memcpy(&by, &y, 8);
        return bx == by ? 0 : (bx < by ? -1 : 1);
    }
    case O_VARCHAR: {
        int32_t oa = c->offsets[a], la = c->offsets[a + 1] - oa, ob = c->offsets[b], lb = c->offsets[b + 1] - ob;
        const uint8_t *pa = (const uint8_t *)c->values + oa, *pb = (const uint8_t *)c->values + ob;
        int32_t m = la < lb ? la : lb;
        int r = memcmp(pa, pb, (size_t)m);
        if (r) return r < 0 ? -1 : 1;
        return la < lb ? -1 : (la > lb ? 1 : 0);
    }
    default: return 0;
    }
}

int32_t o_compare_rows(const o_column *cols, const int32_t *sort_channels, const int32_t *sort_orders, int32_t n_sort, int32_t a, int32_t b)
{
    for (int32_t i = 0; i < n_sort; i++) {
        const o_column *c = &cols[sort_channels[i]];
        const int asc = sort_orders[i] == 0 || sort_orders[i] == 1, nulls_first = sort_orders[i] == 0 || sort_orders[i] == 2;
        const int na = c->nulls && c->nulls[a], nb = c->nulls && c->nulls[b];
        if (na || nb) {   /* TypeOperators.orderNulls */
            if (na && nb) continue;
            if (na) return nulls_first ? -1 : 1;
            return nulls_first ? 1 : -1;
        }
        int cmp = compare_cells(c, a, b);
        if (cmp) return asc ? cmp : -cmp;
    }
    return 0;
}

/* every row in sort order: bottom-up merge sort, stable (the left run wins ties), what PagesIndex.sort (M/operator/PagesIndex.java
 * :386-394) computes up to the order of rows that compare equal */
static void sort_all_rows(const o_column *cols, int32_t rows, const int32_t *sort_channels, const int32_t *sort_orders, int32_t n_sort, int32_t *pos)
{
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rows > 0 ? rows : 1));
    for (int32_t i = 0; i < rows; i++) pos[i] = i;
    for (int32_t width = 1; width < rows; width *= 2) {
        for (int32_t lo = 0; lo < rows; lo += 2 * width) {
            int32_t mid = lo + width < rows ? lo + width : rows, hi = lo + 2 * width < rows ? lo + 2 * width : rows;
            int32_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = o_compare_rows(cols, sort_channels, sort_orders, n_sort, pos[j], pos[i]) < 0 ? pos[j++] : pos[i++];
            while (i < mid) tmp[k++] = pos[i++];
            while (j < hi) tmp[k++] = pos[j++];
        }
        memcpy(pos, tmp, sizeof(int32_t) * (size_t)rows);
    }
    free(tmp);
}

int32_t o_top_n(const o_column *cols, int32_t rows, int32_t n, const int32_t *sort_channels, const int32_t *sort_orders, int32_t n_sort,
                int32_t *positions_out)
{
    if (n > 64) {   /* large n (OrderBy: n = rows): sort everything, keep the first n -- the same rows in the same order */
        int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rows > 0 ? rows : 1));
        sort_all_rows(cols, rows, sort_channels, sort_orders, n_sort, all);
        int32_t k = n < rows ? n : rows;
        memcpy(positions_out, all, sizeof(int32_t) * (size_t)k);
        free(all);
        return k;
    }
    /* row-at-a-time, like the reference's per-row heap insert: a sorted buffer of at most n rows; a new row goes behind every kept
     * row that does not sort after it (input order among equals) and pushes the last one out */
    int32_t kept = 0;
    if (n <= 0) return 0;
    for (int32_t r = 0; r < rows; r++) {
        int32_t at = kept;
        while (at > 0 && o_compare_rows(cols, sort_channels, sort_orders, n_sort, r, positions_out[at - 1]) < 0) at--;
        if (at >= n) continue;
        int32_t last = kept < n ? kept : n - 1;
        for (int32_t j = last; j > at; j--) positions_out[j] = positions_out[j - 1];
        positions_out[at] = r;
        if (kept < n) kept++;
    }
    return kept;
}
