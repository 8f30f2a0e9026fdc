// device_cols.h -- column views passed by value to kernels + generic per-row key operations over them.
// Shared by the AOT kernels and (embedded verbatim, after device_hash.h) the JIT-compiled ones: self-contained.
#pragma once

// One flat column in HBM: the reference's block arrays (values[n]; one null byte per row or absent; VARCHAR = byte pool +
// int32 offsets[n+1]).  type uses the tgpu_type codes (1 BIGINT, 2 INTEGER, 3 DATE, 4 DOUBLE, 5 BOOLEAN, 6 VARCHAR).
struct TgColView {
    const void *values;
    const unsigned char *nulls;
    const int *offsets;
    int type;
    int pad;
};
#define TG_MAX_KEY_CHANNELS 8
struct TgKeyCols {
    int n;
    int pad;
    TgColView c[TG_MAX_KEY_CHANNELS];
};

// type hash of one cell, null -> 0 (M/type/BlockTypeOperators.java:102-108, M/type/TypeUtils.java:42)
__device__ inline tg_i64 tg_hash_cell(const TgColView &c, long long r)
{
    if (c.nulls && c.nulls[r]) return 0;
    switch (c.type) {
    case 1: return tg_hash_long(((const tg_i64 *)c.values)[r]);
    case 2:
    case 3: return tg_hash_int(((const int *)c.values)[r]);
    case 4: return tg_hash_double_bits(((const tg_u64 *)c.values)[r]);
    case 5: return tg_hash_boolean(((const tg_u8 *)c.values)[r]);
    case 6: {
        const int a = c.offsets[r], b = c.offsets[r + 1];
        return (tg_i64)tg_xxh64((const tg_u8 *)c.values + a, b - a);
    }
    default: return 0;
    }
}

// Two flavours of the per-row key functions.  The plain ones index the column array at run time: use them on key sets that live
// in MEMORY (kernels that take `const TgKeyCols *`; the loads are scalar and the code stays small).  The `_u` ones are fully
// unrolled over the 8 possible channels with constant indices: use them on key sets passed BY VALUE as kernel parameters (a
// run-time index would make the compiler copy the whole parameter block to scratch) -- at the price of 8x the code.
// raw hash of the key columns of one row (M/operator/InterpretedHashGenerator.java:56-70)
__device__ inline tg_i64 tg_hash_row(const TgKeyCols &k, long long r)
{
    tg_i64 h = 0;
    for (int c = 0; c < k.n; c++) h = tg_combine_hash(h, tg_hash_cell(k.c[c], r));
    return h;
}

// IS NOT DISTINCT FROM per channel (JoinCompiler.java positionNotDistinctFromRow; DoubleType.java:181-192 NaN rule)
__device__ inline bool tg_rows_not_distinct(const TgKeyCols &a, long long ra, const TgKeyCols &b, long long rb)
{
    for (int c = 0; c < a.n; c++) {
        const TgColView &x = a.c[c], &y = b.c[c];
        const bool nx = x.nulls && x.nulls[ra], ny = y.nulls && y.nulls[rb];
        if (nx || ny) {
            if (nx != ny) return false;
            continue;
        }
        switch (x.type) {
        case 1:
            if (((const tg_i64 *)x.values)[ra] != ((const tg_i64 *)y.values)[rb]) return false;
            break;
        case 2:
        case 3:
            if (((const int *)x.values)[ra] != ((const int *)y.values)[rb]) return false;
            break;
        case 4: {
            const double u = ((const double *)x.values)[ra], v = ((const double *)y.values)[rb];
            if (!((u != u && v != v) || u == v)) return false;
            break;
        }
        case 5:
            if ((((const tg_u8 *)x.values)[ra] != 0) != (((const tg_u8 *)y.values)[rb] != 0)) return false;
            break;
        case 6: {
            const int ax = x.offsets[ra], lx = x.offsets[ra + 1] - ax;
            const int ay = y.offsets[rb], ly = y.offsets[rb + 1] - ay;
            if (lx != ly) return false;
            const tg_u8 *px = (const tg_u8 *)x.values + ax, *py = (const tg_u8 *)y.values + ay;
            for (int i = 0; i < lx; i++)
                if (px[i] != py[i]) return false;
            break;
        }
        default: return false;
        }
    }
    return true;
}
// the same for a key set passed BY VALUE as a kernel parameter
__device__ inline tg_i64 tg_hash_row_u(const TgKeyCols &k, long long r)
{
    // full unroll with constant column indices: a run-time index into the by-value column array would move it to scratch
    tg_i64 h = 0;
#pragma unroll
    for (int c = 0; c < TG_MAX_KEY_CHANNELS; c++) {
        if (c >= k.n) break;
        h = tg_combine_hash(h, tg_hash_cell(k.c[c], r));
    }
    return h;
}

// the same for key sets passed BY VALUE as kernel parameters; IS NOT DISTINCT FROM per channel (JoinCompiler.java positionNotDistinctFromRow; DoubleType.java:181-192 NaN rule)
__device__ inline bool tg_rows_not_distinct_u(const TgKeyCols &a, long long ra, const TgKeyCols &b, long long rb)
{
#pragma unroll
    for (int c = 0; c < TG_MAX_KEY_CHANNELS; c++) {
        if (c >= a.n) break;
        const TgColView &x = a.c[c], &y = b.c[c];
        const bool nx = x.nulls && x.nulls[ra], ny = y.nulls && y.nulls[rb];
        if (nx || ny) {
            if (nx != ny) return false;
            continue;
        }
        switch (x.type) {
        case 1:
            if (((const tg_i64 *)x.values)[ra] != ((const tg_i64 *)y.values)[rb]) return false;
            break;
        case 2:
        case 3:
            if (((const int *)x.values)[ra] != ((const int *)y.values)[rb]) return false;
            break;
        case 4: {
            const double u = ((const double *)x.values)[ra], v = ((const double *)y.values)[rb];
            if (!((u != u && v != v) || u == v)) return false;
            break;
        }
        case 5:
            if ((((const tg_u8 *)x.values)[ra] != 0) != (((const tg_u8 *)y.values)[rb] != 0)) return false;
            break;
        case 6: {
            const int ax = x.offsets[ra], lx = x.offsets[ra + 1] - ax;
            const int ay = y.offsets[rb], ly = y.offsets[rb + 1] - ay;
            if (lx != ly) return false;
            const tg_u8 *px = (const tg_u8 *)x.values + ax, *py = (const tg_u8 *)y.values + ay;
            for (int i = 0; i < lx; i++)
                if (px[i] != py[i]) return false;
            break;
        }
        default: return false;
        }
    }
    return true;
}
