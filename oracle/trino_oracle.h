/*
 * trino_oracle.h -- CPU restatement (plain C) of the reference's page-at-a-time operator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker
 * (or as the timed CPU baseline).  The product library (presto-1_amd/csrc) never links, imports
 * or calls it.
 *
 * Every function cites the reference file:line it follows.  Paths are relative to /root/reference,
 * with M/ = core/trino-main/src/main/java/io/trino/ and S/ = core/trino-spi/src/main/java/io/trino/spi/.
 *
 * Parity pinning: the reference is Java and no JVM exists in the build image, so the oracle is
 * pinned by the reference's own test literals (tests/test_oracle_golden.py, tests/golden/): the XXH64 known answers of
 * T/operator/scalar/TestVarbinaryFunctions.java:334-335, the group-id vectors of
 * T/operator/TestGroupByHash.java, the aggregation rows of T/operator/TestHashAggregationOperator.java:161-220,
 * the join rows of T/operator/TestHashJoinOperator.java:164-199 (+ null variants) and the chain
 * order of T/operator/TestPositionLinks.java.  XxHash64.hash(long) has no literal in the reference
 * tests ("parity unpinned" for that one function; cross-checked against the python xxhash package).
 */
#ifndef TRINO_ORACLE_H
#define TRINO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data model (S/block/LongArrayBlock.java:38-75, IntArrayBlock.java, ByteArrayBlock.java,
 *      VariableWidthBlock.java:38-83): flat values + one null byte per position (+ offsets) ---- */
enum {
    O_BIGINT = 1,   /* int64 */
    O_INTEGER = 2,  /* int32 */
    O_DATE = 3,     /* int32 days */
    O_DOUBLE = 4,   /* IEEE double */
    O_BOOLEAN = 5,  /* 1 byte */
    O_VARCHAR = 6   /* bytes + int32 offsets[n+1] */
};

typedef struct o_column {
    int32_t type;
    int32_t n;
    const void *values;      /* fixed width: n elements; VARCHAR: byte pool */
    const uint8_t *nulls;    /* may be NULL (= no nulls) */
    const int32_t *offsets;  /* VARCHAR only */
} o_column;

/* ---- hash family (bit-exact) ---- */
int64_t o_hash_long(int64_t v);                 /* S/type/AbstractLongType.java:126-130 */
int64_t o_hash_int(int32_t v);                  /* S/type/AbstractIntType.java:141-145 */
int64_t o_hash_double(double v);                /* S/type/DoubleType.java:163-170 */
int64_t o_hash_boolean(uint8_t v);              /* S/type/BooleanType.java hashCodeOperator: value ? 1231 : 1237 */
uint64_t o_xxh64(const uint8_t *p, size_t len, uint64_t seed); /* io.airlift.slice.XxHash64 (standard XXH64) */
int64_t o_xxh64_long(int64_t v);                /* XxHash64.hash(long) */
int64_t o_combine_hash(int64_t prev, int64_t v);/* M/operator/scalar/CombineHashFunction.java:24-29 */
uint64_t o_murmur3_fmix(uint64_t x);            /* M/operator/PagesHash.java:224-240 (== fastutil HashCommon.murmurHash3) */
int32_t o_array_size(int32_t expected, float f);/* fastutil 8.3.0 HashCommon.arraySize */
int32_t o_calculate_max_fill(int32_t hash_size);/* M/operator/BigintGroupByHash.java:325-334 */

/* type hash of one cell, null -> 0: M/type/BlockTypeOperators.java:102-108, M/type/TypeUtils.java:42 */
int64_t o_hash_cell(const o_column *c, int32_t pos);
/* raw hash of the key columns of every row: M/operator/InterpretedHashGenerator.java:56-70 */
void o_hash_rows(const o_column *cols, int32_t ncols, int32_t n, int64_t *out);
/* remote partition: M/operator/HashGenerator.java:24-35 */
int32_t o_partition_remote(int64_t raw_hash, int32_t partition_count);
/* local partition: M/operator/exchange/LocalPartitionGenerator.java:45-65 */
int32_t o_partition_local(int64_t raw_hash, int32_t partition_count_pow2);

/* ---- BigintGroupByHash: M/operator/BigintGroupByHash.java ---- */
typedef struct o_bigint_gbh o_bigint_gbh;
o_bigint_gbh *o_bigint_gbh_new(int32_t expected_size);
void o_bigint_gbh_free(o_bigint_gbh *g);
/* getGroupIds (:168-172, GetGroupIdsWork :366-413): out[i] = group id of row i; returns -3 on the 1-billion-entry limit */
int32_t o_bigint_gbh_get_group_ids(o_bigint_gbh *g, const o_column *col, int64_t *out);
int32_t o_bigint_gbh_contains(const o_bigint_gbh *g, const o_column *col, int32_t pos); /* :181-203 */
int32_t o_bigint_gbh_group_count(const o_bigint_gbh *g);
int32_t o_bigint_gbh_capacity(const o_bigint_gbh *g);
int64_t o_bigint_gbh_hash_collisions(const o_bigint_gbh *g);
int32_t o_bigint_gbh_rehash_count(const o_bigint_gbh *g);
/* appendValuesTo (:137-158) for group ids 0..count-1: values, null flags and raw hashes */
void o_bigint_gbh_values(const o_bigint_gbh *g, int64_t *values, uint8_t *nulls, int64_t *raw_hashes);

/* ---- MultiChannelGroupByHash: M/operator/MultiChannelGroupByHash.java:275-463 ---- */
typedef struct o_multi_gbh o_multi_gbh;
o_multi_gbh *o_multi_gbh_new(int32_t nchannels, const int32_t *types, int32_t expected_size);
void o_multi_gbh_free(o_multi_gbh *g);
/* getGroupIds over a page.  hashes == NULL -> InterpretedHashGenerator, else the precomputed channel (:114) */
int32_t o_multi_gbh_get_group_ids(o_multi_gbh *g, const o_column *cols, const int64_t *hashes, int32_t n, int64_t *out);
int32_t o_multi_gbh_contains(const o_multi_gbh *g, const o_column *cols, int32_t pos, int64_t raw_hash);
int32_t o_multi_gbh_group_count(const o_multi_gbh *g);
int32_t o_multi_gbh_capacity(const o_multi_gbh *g);
int32_t o_multi_gbh_rehash_count(const o_multi_gbh *g);
/* group g's first-seen input row (global row counter across pages) and raw hash, for key round trips */
void o_multi_gbh_group_rows(const o_multi_gbh *g, int64_t *first_rows, int64_t *raw_hashes);

/* ---- aggregation accumulators in Java (sequential) order:
 *      M/operator/aggregation/AccumulatorCompiler.java:487-566 loop shape ---- */
/* sum(double)/avg(double): DoubleSumAggregation.java:34-38, AverageAggregations.java:42-47 */
void o_agg_double_sum(const int64_t *gids, const double *v, const uint8_t *nulls, const uint8_t *mask,
                      int32_t n, int64_t *counts, double *sums);
/* avg(bigint): AverageAggregations.java:35-40  (state double += (double) long) */
void o_agg_long_avg(const int64_t *gids, const int64_t *v, const uint8_t *nulls, const uint8_t *mask,
                    int32_t n, int64_t *counts, double *sums);
/* sum(bigint): LongSumAggregation.java:34-39 ; returns -2 (NUMERIC_VALUE_OUT_OF_RANGE) on overflow */
int32_t o_agg_long_sum(const int64_t *gids, const int64_t *v, const uint8_t *nulls, const uint8_t *mask,
                       int32_t n, int64_t *counts, int64_t *sums);
void o_agg_double_minmax(const int64_t *gids, const double *v, const uint8_t *nulls, const uint8_t *mask, int32_t n, int32_t is_min,
                         int64_t *counts, double *values);
void o_agg_long_minmax(const int64_t *gids, const int64_t *v, const uint8_t *nulls, const uint8_t *mask, int32_t n, int32_t is_min,
                       int64_t *counts, int64_t *values);
/* count(*) / count(col): CountAggregation.java:34-38, CountColumn.java */
void o_agg_count(const int64_t *gids, const uint8_t *nulls, const uint8_t *mask, int32_t n, int64_t *counts);
/* exactly-rounded sum of doubles (Shewchuk / msum): the scale reference for the GPU's exact accumulation policy */
double o_exact_sum(const double *v, int64_t n);
void o_agg_double_sum_exact(const int64_t *gids, const double *v, const uint8_t *nulls, const uint8_t *mask,
                            int64_t n, int32_t ngroups, int64_t *counts, double *sums);

/* ---- join: PagesHash + ArrayPositionLinks + JoinHash + PageJoiner ---- */
typedef struct o_pages_hash o_pages_hash;
/* M/operator/PagesHash.java:53-125 ; hashes == NULL -> hashPosition via H5 (JoinCompiler.java:405-447) */
o_pages_hash *o_pages_hash_new(const o_column *key_cols, int32_t ncols, int32_t n, const int64_t *hashes);
void o_pages_hash_free(o_pages_hash *h);
int32_t o_pages_hash_size(const o_pages_hash *h);
int32_t o_pages_hash_link_count(const o_pages_hash *h);        /* ArrayPositionLinks.FactoryBuilder.size() */
const int32_t *o_pages_hash_links(const o_pages_hash *h);      /* links[n], -1 terminated chains */
const int32_t *o_pages_hash_keys(const o_pages_hash *h);       /* key[hashSize] */
int64_t o_pages_hash_collisions(const o_pages_hash *h);
/* PagesHash.getAddressIndex :157-169 */
int32_t o_pages_hash_get_address_index(const o_pages_hash *h, const o_column *probe_cols, int32_t pos, int64_t raw_hash);
/* INNER / PROBE_OUTER probe of one page: M/operator/LookupJoinOperator.java:299-378, JoinProbe.java:87-117.
 * Writes (probe idx, build idx) pairs in output order; build idx -1 = outer row.  Returns the number of pairs,
 * or -(needed) if cap is too small. */
int64_t o_join_probe(const o_pages_hash *h, const o_column *probe_cols, int32_t n_probe, const int64_t *hashes,
                     int32_t probe_outer, int32_t *out_probe, int32_t *out_build, int64_t cap);

/* ---- filter / project: RowExpression interpreter with the generated code's null / short-circuit
 *      protocol (M/sql/gen/BytecodeUtils.java:189-356, AndCodeGenerator.java:44-105, OrCodeGenerator.java) ---- */
enum { O_EX_INPUT = 0, O_EX_CONST = 1, O_EX_CALL = 2, O_EX_SPECIAL = 3 };
enum { /* CALL ops */
    O_OP_ADD = 1, O_OP_SUBTRACT, O_OP_MULTIPLY, O_OP_DIVIDE, O_OP_MODULUS, O_OP_NEGATE,
    O_OP_EQUAL, O_OP_NOT_EQUAL, O_OP_LESS_THAN, O_OP_LESS_THAN_OR_EQUAL, O_OP_GREATER_THAN, O_OP_GREATER_THAN_OR_EQUAL,
    O_OP_NOT, O_OP_CAST
};
enum { /* special forms: M/sql/relational/SpecialForm.java:137-152 */
    O_SF_AND = 1, O_SF_OR, O_SF_IF, O_SF_IS_NULL, O_SF_COALESCE, O_SF_BETWEEN
};
typedef struct o_expr_node {
    int32_t kind;
    int32_t type;      /* result type */
    int32_t op;        /* CALL op / special form / input channel */
    int32_t n_args;
    int32_t args[3];   /* node indices */
    int32_t is_null;   /* CONST: null literal */
    int64_t ival;      /* CONST bigint/integer/date/boolean; VARCHAR const: offset into the string pool */
    double dval;       /* CONST double */
    int32_t slen;      /* VARCHAR const length */
    int32_t pad;
} o_expr_node;

/* error codes shared with the product C ABI (include/tgpu.h) */
enum {
    O_OK = 0,
    O_ERR_INVALID = -1,
    O_ERR_NUMERIC_VALUE_OUT_OF_RANGE = -2,
    O_ERR_INSUFFICIENT_RESOURCES = -3,
    O_ERR_DIVISION_BY_ZERO = -7,
    O_ERR_INVALID_CAST_ARGUMENT = -9
};

/* PageFilter.filter + positionsArrayToSelectedPositions (M/operator/project/PageFilter.java:27-50):
 * writes the selected positions ascending, returns their count (or <0 error, *err_row = first failing row) */
int32_t o_filter(const o_expr_node *nodes, int32_t root, const char *pool, const o_column *cols, int32_t n,
                 int32_t *positions, int32_t *err_row);
/* PageProjection over selected positions (M/sql/gen/PageFunctionCompiler.java:274-320).  Output arrays sized
 * n_sel: fixed width into out_values (8/4/1 bytes by type) + out_nulls.  VARCHAR results are not supported here. */
int32_t o_project(const o_expr_node *nodes, int32_t root, const char *pool, const o_column *cols,
                  const int32_t *positions, int32_t n_sel, void *out_values, uint8_t *out_nulls, int32_t *err_row);

#ifdef __cplusplus
}
#endif
/* ---- TopN: M/operator/TopNOperator.java:47-62, TopNProcessor.java:45-66; row order = SimplePageWithPositionComparator.java:58-79
 * with the null placement and DESC negation of S/type/TypeOperators.java:578-596.  sort_orders: 0 ASC_NULLS_FIRST, 1 ASC_NULLS_LAST,
 * 2 DESC_NULLS_FIRST, 3 DESC_NULLS_LAST (S/connector/SortOrder.java:18-21).  Writes the row numbers of the min(n, rows) first rows in
 * sort order; rows equal on every sort channel keep their input order (the reference's heap leaves that order unspecified).
 * Returns the number of rows written. */
int32_t o_compare_rows(const o_column *cols, const int32_t *sort_channels, const int32_t *sort_orders, int32_t n_sort, int32_t a, int32_t b);
int32_t o_top_n(const o_column *cols, int32_t rows, int32_t n, const int32_t *sort_channels, const int32_t *sort_orders, int32_t n_sort,
                int32_t *positions_out);

#endif
