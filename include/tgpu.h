/*
 * tgpu.h -- C ABI of the MI355X-native operator hot path (libtgpu.so).
 *
 * This is the drop-in boundary: exactly what a JNI shim for core/trino-main would bind (INTEGRATION.md shows
 * that shim).  Plain C: opaque handles, plain pointers and sizes, int32 status codes.  No torch / HIP types.
 *
 * Reference interfaces replaced (paths relative to the reference root;
 * M/ = core/trino-main/src/main/java/io/trino/ , S/ = core/trino-spi/src/main/java/io/trino/spi/):
 *   - tgpu_operator_*            <-> M/operator/Operator.java:20-102
 *   - tgpu_operator_factory_*    <-> M/operator/OperatorFactory.java:18-50
 *   - tgpu_block / tgpu_page     <-> S/Page.java:33-73, S/block/LongArrayBlock.java:38-75, IntArrayBlock.java,
 *                                    ByteArrayBlock.java, VariableWidthBlock.java:38-83, DictionaryBlock.java:40-100,
 *                                    RunLengthEncodedBlock.java:30-70
 *   - tgpu_filter_project_*      <-> M/operator/FilterAndProjectOperator.java:73-88 + M/sql/gen/ExpressionCompiler.java:94-122
 *   - tgpu_hash_aggregation_*    <-> M/operator/HashAggregationOperator.java:54-262
 *   - tgpu_hash_builder_* / tgpu_lookup_join_* <-> M/operator/HashBuilderOperator.java:54-152,
 *                                    M/operator/LookupJoinOperatorFactory.java:40-113, LookupJoinOperators.java:30-63
 *   - tgpu_group_by_hash_*       <-> M/operator/GroupByHash.java:45-99
 *   - tgpu_hash_page             <-> M/operator/InterpretedHashGenerator.java:56-70
 *   - tgpu_partition_page        <-> M/operator/PartitionedOutputOperator.java:406-426 + HashGenerator.java:24-35
 *   - tgpu_partitioned_output_*  <-> M/operator/PartitionedOutputOperator.java:46-486 (operator + PagePartitioner)
 *   - tgpu_top_n_* / tgpu_order_by_* <-> M/operator/TopNOperator.java:47-225, OrderByOperator.java:48-300
 *   - tgpu_lookup_outer_*        <-> M/operator/LookupOuterOperator.java:32-235, OuterLookupSource.java:146-190
 *   - tgpu_merge_pages_*         <-> M/operator/project/MergePages.java:64-190
 *   - tgpu_dynamic_filter_source_* <-> M/operator/DynamicFilterSourceOperator.java:74-425
 *   - tgpu_serialize_page / tgpu_deserialize_page <-> M/execution/buffer/PagesSerde.java:64-160, PagesSerdeUtil.java:45-71,
 *                                    S/block/{LongArray,IntArray,ByteArray,VariableWidth,RunLength,Dictionary}BlockEncoding.java, EncoderUtil.java:33-118
 *   - tgpu_exchange_*            <-> M/operator/PartitionedOutputOperator.java:406-476 -> M/operator/ExchangeOperator.java (the hop between
 *                                    the stages of a FIXED_HASH / FIXED_BROADCAST distribution), over RCCL / xGMI instead of HTTP
 *   - tgpu_operator_add_input_output_page: Operator.addInput with a page that never left the device (no reference counterpart:
 *                                    on the JVM the Page object itself is what travels between operators)
 *
 * Threading rule = the reference's (M/operator/Driver.java:55-62): one caller at a time per handle; distinct
 * handles are independent, also when they belong to one context (their kernels then share its stream and execute in issue
 * order; the context serialises its staging buffers; the per-kernel profile assumes a single driver thread); a built lookup
 * source is immutable and shared read-only by probe operators (outer joins mark visited build positions with idempotent stores).
 * Ownership rule: input buffers are only read during the call (the library copies/uploads what it keeps);
 * output pages are owned by the library until tgpu_output_page_release.
 */
#ifndef TGPU_H
#define TGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: 0 ok, >0 informational, <0 error mirroring io.trino.spi.StandardErrorCode ---- */
#define TGPU_OK 0
#define TGPU_WOULD_BLOCK 1                          /* tgpu_operator_get_output: no page yet and the operator's isBlocked() future is not done */
#define TGPU_ERR_INVALID_ARGUMENT (-1)              /* GENERIC_INTERNAL_ERROR / IllegalArgumentException */
#define TGPU_ERR_NUMERIC_VALUE_OUT_OF_RANGE (-2)    /* M/type/BigintOperators.java:47-79 */
#define TGPU_ERR_INSUFFICIENT_RESOURCES (-3)        /* GENERIC_INSUFFICIENT_RESOURCES: BigintGroupByHash.java:264-267, PagesIndex.java:234-236 */
#define TGPU_ERR_COMPILER (-4)                      /* COMPILER_ERROR: M/sql/gen/PageFunctionCompiler.java:199-205 */
#define TGPU_ERR_INTERNAL (-5)                      /* GENERIC_INTERNAL_ERROR (illegal operator state etc.) */
#define TGPU_ERR_DEVICE (-6)                        /* no HIP device / HIP runtime failure: the library never falls back to CPU */
#define TGPU_ERR_DIVISION_BY_ZERO (-7)              /* DIVISION_BY_ZERO */
#define TGPU_ERR_NOT_SUPPORTED (-8)                 /* NOT_SUPPORTED */
#define TGPU_ERR_INVALID_CAST_ARGUMENT (-9)         /* INVALID_CAST_ARGUMENT: M/type/DoubleOperators.java:108-163 */

/* ---- types (S/type): storage is what the reference's flat blocks hold ---- */
typedef enum tgpu_type {
    TGPU_BIGINT = 1,   /* int64   LongArrayBlock */
    TGPU_INTEGER = 2,  /* int32   IntArrayBlock */
    TGPU_DATE = 3,     /* int32   IntArrayBlock (days) */
    TGPU_DOUBLE = 4,   /* IEEE double, LongArrayBlock bits */
    TGPU_BOOLEAN = 5,  /* 1 byte  ByteArrayBlock */
    TGPU_VARCHAR = 6   /* VariableWidthBlock: byte pool + int32 offsets[n+1] */
} tgpu_type;

typedef enum tgpu_encoding {
    TGPU_FLAT = 0, TGPU_DICTIONARY = 1, TGPU_RLE = 2,
    TGPU_LAZY = 3   /* S/block/LazyBlock.java: not loaded yet; only in pages a tgpu_page_source hands out (loaded through its load_block) */
} tgpu_encoding;
typedef enum tgpu_memory { TGPU_HOST = 0, TGPU_DEVICE = 1 } tgpu_memory;

/* One Block.  All pointers of one block live in the same memory space (`memory`). */
typedef struct tgpu_block {
    int32_t type;            /* tgpu_type */
    int32_t encoding;        /* tgpu_encoding */
    int32_t memory;          /* tgpu_memory */
    int32_t position_count;
    const void *values;      /* FLAT fixed width: position_count elements (arrayOffset already applied); VARCHAR: byte pool */
    const uint8_t *nulls;    /* FLAT: one byte per position (Java boolean[] valueIsNull), NULL = no nulls */
    const int32_t *offsets;  /* FLAT VARCHAR: position_count + 1 */
    const int32_t *ids;      /* DICTIONARY: position_count ids into `dictionary` */
    const struct tgpu_block *dictionary; /* DICTIONARY: the dictionary; RLE: the single-position value block */
} tgpu_block;

typedef struct tgpu_page {
    int32_t position_count;
    int32_t channel_count;
    const tgpu_block *blocks;
} tgpu_page;

typedef struct tgpu_context tgpu_context;
typedef struct tgpu_operator_factory tgpu_operator_factory;
typedef struct tgpu_operator tgpu_operator;
typedef struct tgpu_lookup_source_factory tgpu_lookup_source_factory;
typedef struct tgpu_group_by_hash tgpu_group_by_hash;
typedef struct tgpu_output_page tgpu_output_page;

/* ---- context: one HIP device + one stream; all handles created from it launch on that stream ---- */
int32_t tgpu_context_create(int32_t device, void *hip_stream /* hipStream_t or NULL = null stream */, tgpu_context **out);
void tgpu_context_destroy(tgpu_context *ctx);
int32_t tgpu_context_synchronize(tgpu_context *ctx);
/* last error message of the calling thread (messages mirror the reference's, e.g. "bigint multiplication overflow: 3 * 4") */
const char *tgpu_last_error(void);
const char *tgpu_version(void);
/* directory holding the JIT kernel sources/cache (defaults to the directory of libtgpu.so) */
int32_t tgpu_set_resource_dir(const char *dir);

/* Order in which sum(double) / avg(double) add their rows (DESIGN.md "DOUBLE aggregate policy"), for operators created afterwards:
 *   EXACT (default): few groups -> the correctly rounded exact sum (order independent; equals the Java result whenever that is itself
 *                    exact, else differs from it by the Java order's own rounding error); many groups -> rows added in row order;
 *   JAVA:            always in row order, one group's rows after another -- the loop of DoubleSumAggregation.java:34-38 /
 *                    AccumulatorCompiler.java:487-566, bit-identical to the Java operator for any input, at the price of a
 *                    sequential chain per group (meant for page-sized inputs and strict-parity runs). */
/* Output page size.  The reference's operators cut their output at PageBuilder.isFull (1 MB, S/block/PageBuilderStatus.java:49-60;
 * LookupJoinPageBuilder.java:51-56, OrderByOperator.java:270-296); the GPU operators produce one page per call, which is what downstream
 * GPU operators want.  In front of Java operators set limits: tgpu_operator_get_output then hands a larger page out as consecutive
 * zero-copy regions of at most max_rows rows and about max_bytes bytes (Java block accounting); 0 = no limit (the default). */
int32_t tgpu_context_set_max_output_page(tgpu_context *ctx, int64_t max_bytes, int64_t max_rows);

/* A promise about borrowed TGPU_DEVICE input (hosts that keep their own HBM buffers; library-owned pages and host pages need none): the
 * blocks of a page stay valid and UNCHANGED until the operator they were handed to is finished (tgpu_operator_is_finished) or closed,
 * instead of "until overwritten in stream order".  It lets the operators that keep input BY REFERENCE do so with such pages: the fused
 * aggregation collects small pages for one launch and re-runs a page whose launch met a new group, the fused join keeps two probe pages in
 * flight (DESIGN.md "Page granularity").  Java pages are immutable and referenced by the operator for as long as it needs them
 * (Operator.addInput transfers a reference, SURVEY.md 8b "Ownership"): this is the device-memory counterpart of that rule.  Default off:
 * borrowed device pages then take the synchronous protocol. */
int32_t tgpu_context_set_device_input_stable(tgpu_context *ctx, int32_t stable);

typedef enum tgpu_double_sum_order { TGPU_SUM_ORDER_EXACT = 0, TGPU_SUM_ORDER_JAVA = 1 } tgpu_double_sum_order;
int32_t tgpu_context_set_double_sum_order(tgpu_context *ctx, int32_t order);

/* Pinned (page-locked) host memory for hosts that fill their own receive buffers (an exchange client, a file reader): blocks whose arrays
 * live in such memory are transferred by asynchronous DMA.  Every tgpu_operator_add_input of a host page stages its arrays through a
 * double-buffered ring on a second HIP stream, so the transfer of page i + 1 runs under the kernels of page i (DESIGN.md "Ingest"). */
int32_t tgpu_pinned_alloc(tgpu_context *ctx, int64_t bytes, void **out);
int32_t tgpu_pinned_free(tgpu_context *ctx, void *ptr);

/* per-kernel HIP-event timing on the context's stream (bench.py's roofline leg) */
int32_t tgpu_profile_enable(tgpu_context *ctx, int32_t enabled);
int32_t tgpu_profile_reset(tgpu_context *ctx);
/* writes a JSON object {"kernel": {"count": n, "total_ms": t, "min_ms": a, "max_ms": b}, ...}; returns needed length */
int64_t tgpu_profile_dump(tgpu_context *ctx, char *buf, int64_t buf_len);

/* ---- RowExpression IR (M/sql/relational/{CallExpression,ConstantExpression,InputReferenceExpression,SpecialForm}.java) ---- */
typedef enum tgpu_expr_kind { TGPU_EX_INPUT = 0, TGPU_EX_CONST = 1, TGPU_EX_CALL = 2, TGPU_EX_SPECIAL = 3 } tgpu_expr_kind;
typedef enum tgpu_expr_op {
    TGPU_OP_ADD = 1, TGPU_OP_SUBTRACT, TGPU_OP_MULTIPLY, TGPU_OP_DIVIDE, TGPU_OP_MODULUS, TGPU_OP_NEGATE,
    TGPU_OP_EQUAL, TGPU_OP_NOT_EQUAL, TGPU_OP_LESS_THAN, TGPU_OP_LESS_THAN_OR_EQUAL, TGPU_OP_GREATER_THAN,
    TGPU_OP_GREATER_THAN_OR_EQUAL, TGPU_OP_NOT, TGPU_OP_CAST
} tgpu_expr_op;
typedef enum tgpu_special_form { /* M/sql/relational/SpecialForm.java:137-152 */
    TGPU_SF_AND = 1, TGPU_SF_OR, TGPU_SF_IF, TGPU_SF_IS_NULL, TGPU_SF_COALESCE, TGPU_SF_BETWEEN
} tgpu_special_form;

typedef struct tgpu_expr_node {
    int32_t kind;      /* tgpu_expr_kind */
    int32_t type;      /* result tgpu_type */
    int32_t op;        /* CALL: tgpu_expr_op; SPECIAL: tgpu_special_form; INPUT: channel */
    int32_t n_args;
    int32_t args[3];   /* indices into the node array */
    int32_t is_null;   /* CONST: null literal */
    int64_t ival;      /* CONST BIGINT/INTEGER/DATE/BOOLEAN value; VARCHAR: offset into string_pool */
    double dval;       /* CONST DOUBLE */
    int32_t slen;      /* CONST VARCHAR length */
    int32_t pad;
} tgpu_expr_node;

/* A page processor = optional filter + projections over one shared node array (ExpressionCompiler.compilePageProcessor) */
typedef struct tgpu_page_processor_spec {
    const tgpu_expr_node *nodes;
    int32_t node_count;
    const char *string_pool;
    int32_t string_pool_len;
    int32_t filter_root;            /* -1 = no filter */
    int32_t projection_count;
    const int32_t *projection_roots;
} tgpu_page_processor_spec;

/* ---- operator factories ---- */
/* FilterAndProjectOperator.createOperatorFactory (M/operator/FilterAndProjectOperator.java:73-88).  The expressions are
 * compiled to one fused gfx950 kernel (the GPU counterpart of M/sql/gen/PageFunctionCompiler.java). */
int32_t tgpu_filter_project_factory_create(tgpu_context *ctx, int32_t operator_id,
                                           int32_t input_type_count, const int32_t *input_types,
                                           const tgpu_page_processor_spec *spec,
                                           tgpu_operator_factory **out);

typedef enum tgpu_agg_function {
    TGPU_AGG_COUNT_ALL = 1,     /* count(*)        M/operator/aggregation/CountAggregation.java:34-56 */
    TGPU_AGG_COUNT_COLUMN = 2,  /* count(col)      CountColumn.java */
    TGPU_AGG_SUM_BIGINT = 3,    /* sum(bigint)     LongSumAggregation.java:34-63 */
    TGPU_AGG_SUM_DOUBLE = 4,    /* sum(double)     DoubleSumAggregation.java:34-63 */
    TGPU_AGG_AVG_BIGINT = 5,    /* avg(bigint)     AverageAggregations.java:35-80 */
    TGPU_AGG_AVG_DOUBLE = 6,    /* avg(double)     AverageAggregations.java:42-80 */
    TGPU_AGG_MIN_BIGINT = 7,    /* min(bigint)     AbstractMinMaxAggregationFunction.java:233-289 (LONG_INPUT / LONG_COMBINE, NullableLongState) */
    TGPU_AGG_MAX_BIGINT = 8,    /* max(bigint)     the same with the comparison turned round (MaxAggregationFunction.java) */
    TGPU_AGG_MIN_DOUBLE = 9,    /* min(double)     :227-230,291-306 with Double.compare (DoubleType.java:194-198): -0.0 < +0.0, NaN above +inf */
    TGPU_AGG_MAX_DOUBLE = 10    /* max(double)     M/util/MinMaxCompare.java maxDouble: value > state || isNaN(state).  A NaN result is the canonical NaN; among
                                 *                 zeros of both signs as the maximum +0.0 is returned (the reference: the one that came first) */
} tgpu_agg_function;

typedef struct tgpu_agg_spec {
    int32_t function;       /* tgpu_agg_function */
    int32_t input_channel;  /* -1 for count(*) */
    int32_t mask_channel;   /* BOOLEAN channel or -1 (AccumulatorCompiler.java:487-566 mask handling) */
} tgpu_agg_spec;

typedef enum tgpu_agg_step { TGPU_STEP_SINGLE = 0, TGPU_STEP_PARTIAL = 1, TGPU_STEP_FINAL = 2 } tgpu_agg_step;

/* HashAggregationOperatorFactory (M/operator/HashAggregationOperator.java:54-262).  Output channels: group-by keys,
 * [hash channel if hash_channel >= 0], then one channel per aggregate (PARTIAL: two channels per sum / avg / min / max = count BIGINT,
 * sum DOUBLE|BIGINT or the extreme BIGINT -- the flattened LongDoubleState / LongLongState / NullableLongState, count 0 = null state;
 * FINAL consumes that layout). */
int32_t tgpu_hash_aggregation_factory_create(tgpu_context *ctx, int32_t operator_id,
                                             int32_t group_by_count, const int32_t *group_by_types, const int32_t *group_by_channels,
                                             int32_t hash_channel /* -1 = none */, int32_t step,
                                             int32_t agg_count, const tgpu_agg_spec *aggs,
                                             int32_t expected_groups, int32_t produce_default_output,
                                             tgpu_operator_factory **out);

/* HashBuilderOperatorFactory + its JoinBridge (M/operator/HashBuilderOperator.java:54-152;
 * PartitionedLookupSourceFactory.java:110-124 with one partition per GPU). */
int32_t tgpu_hash_builder_factory_create(tgpu_context *ctx, int32_t operator_id,
                                         int32_t type_count, const int32_t *types,
                                         int32_t output_channel_count, const int32_t *output_channels,
                                         int32_t hash_channel_count, const int32_t *hash_channels /* join key channels */,
                                         int32_t precomputed_hash_channel /* -1 = none */, int32_t expected_positions,
                                         tgpu_lookup_source_factory **bridge_out, tgpu_operator_factory **out);
/* The build side as `partition_count` HashBuilderOperators (PartitionedLookupSourceFactory.java:110-124: one per build driver; each receives
 * the rows a LocalExchange with the LocalPartitionGenerator function -- TGPU_PARTITION_LOCAL below -- routes to its partition).
 * tgpu_operator_factory_create_operator may then be called partition_count times; the probes stay blocked until every build operator has
 * finished (lendPartitionLookupSource).  The partitions are concatenated in partition order into ONE table in HBM (the reference keeps P
 * tables to parallelise a CPU build); results equal PartitionedLookupSource's (PartitionedLookupSource.java:87-153): a key's matches
 * newest -> oldest within its partition, unmatched outer rows partition by partition (:233-262).  partition_count: a power of two. */
int32_t tgpu_partitioned_hash_builder_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types,
                                                     int32_t output_channel_count, const int32_t *output_channels, int32_t hash_channel_count,
                                                     const int32_t *hash_channels, int32_t precomputed_hash_channel, int32_t expected_positions,
                                                     int32_t partition_count, tgpu_lookup_source_factory **bridge_out, tgpu_operator_factory **out);
/* PartitionedLookupSource's join-position encoding (PartitionedLookupSource.java:212-226): (joinPosition << shiftSize) | partition with
 * shiftSize = numberOfTrailingZeros(partitionCount) + 1; for shims that must hand such positions to Java code.  encode returns
 * TGPU_ERR_INVALID_ARGUMENT (negative, never a position) unless partition_count is a power of two, 0 <= partition < partition_count and
 * join_position >= 0 */
int64_t tgpu_partitioned_join_position_encode(int32_t partition, int32_t join_position, int32_t partition_count);
int32_t tgpu_partitioned_join_position_decode(int64_t partitioned_join_position, int32_t partition_count, int32_t *partition, int32_t *join_position);
void tgpu_lookup_source_factory_destroy(tgpu_lookup_source_factory *bridge);
/* JoinFilterFunction (M/operator/JoinHash.java:44-47,82-130; M/sql/gen/JoinFilterFunctionCompiler.java; handed to the build side like
 * JoinHashSupplier.java:54-70): a predicate over (build row, probe row) that a join position must pass besides key equality.  `spec`'s
 * filter expression is that predicate (its projections are ignored); its input channels [0, build type count) are the build side's
 * channels (the hash builder's `types`), channel build-type-count + k is probe channel k.  A key's chain is walked newest -> oldest and
 * rejected positions are skipped; a PROBE_OUTER / FULL_OUTER row all of whose candidates are rejected comes out with a null build side.
 * Call before the probe operators are created. */
int32_t tgpu_lookup_source_factory_set_join_filter(tgpu_lookup_source_factory *bridge, int32_t probe_type_count, const int32_t *probe_types,
                                                   const tgpu_page_processor_spec *spec);
/* statistics of the built table (valid once the build operator finished): positions, table slots, position links */
int32_t tgpu_lookup_source_stats(tgpu_lookup_source_factory *bridge, int64_t *positions, int64_t *hash_size, int64_t *link_count);

/* LookupJoinOperators.JoinType ordinals (M/operator/LookupJoinOperators.java:30-36).  LOOKUP_OUTER / FULL_OUTER probes also record the
 * build positions they matched; the unmatched build rows come out of the LookupOuterOperator (tgpu_lookup_outer_factory_create) */
typedef enum tgpu_join_type { TGPU_JOIN_INNER = 0, TGPU_JOIN_PROBE_OUTER = 1, TGPU_JOIN_LOOKUP_OUTER = 2, TGPU_JOIN_FULL_OUTER = 3 } tgpu_join_type;

/* LookupJoinOperators.innerJoin / probeOuterJoin (M/operator/LookupJoinOperators.java:30-63).  Output page = probe output
 * channels then the build side's output channels (M/operator/LookupJoinPageBuilder.java:101-131). */
int32_t tgpu_lookup_join_factory_create(tgpu_context *ctx, int32_t operator_id, tgpu_lookup_source_factory *bridge,
                                        int32_t probe_type_count, const int32_t *probe_types,
                                        int32_t probe_join_channel_count, const int32_t *probe_join_channels,
                                        int32_t probe_hash_channel /* -1 = none */,
                                        int32_t probe_output_channel_count, const int32_t *probe_output_channels,
                                        int32_t join_type, tgpu_operator_factory **out);

/* Operator fusion by codegen: FilterAndProjectOperator feeding LookupJoinOperator compiled into one kernel (what
 * LocalExecutionPlanner.visitJoin would construct when the probe source is a filter/project node, M/sql/planner/
 * LocalExecutionPlanner.java:1742,2284-2311).  Behaves exactly like the two reference operators back to back:
 * `spec`'s projections form the probe page, and probe_join_channels / probe_hash_channel / probe_output_channels index
 * those projections.  Configurations the fused kernel does not cover run the two steps unfused inside the operator.
 * Pages the operator may keep (library-owned pages, host pages, borrowed device pages under tgpu_context_set_device_input_stable) of up to
 * 2^25 rows are probed asynchronously: pages below 2^22 rows are collected until 2^24 rows or 64 pages share one launch (one output page,
 * rows in input order), and two launches are kept in flight.  tgpu_operator_get_output returns no page for an input page until that has
 * happened, tgpu_operator_finish was called or get_output is polled twice without input in between (Operator.getOutput may return null,
 * M/operator/Operator.java:53-79); an expression error of such a page is returned by the call that completes its launch. */
int32_t tgpu_filter_project_lookup_join_factory_create(tgpu_context *ctx, int32_t operator_id, tgpu_lookup_source_factory *bridge,
                                                       int32_t input_type_count, const int32_t *input_types,
                                                       const tgpu_page_processor_spec *spec,
                                                       int32_t probe_join_channel_count, const int32_t *probe_join_channels,
                                                       int32_t probe_hash_channel /* -1 = none */,
                                                       int32_t probe_output_channel_count, const int32_t *probe_output_channels,
                                                       int32_t join_type, tgpu_operator_factory **out);

/* LookupOuterOperator.LookupOuterOperatorFactory (M/operator/LookupOuterOperator.java:35-110; created by LookupJoinOperatorFactory for
 * LOOKUP_OUTER / FULL_OUTER joins, LookupJoinOperatorFactory.java:88-103): a source operator that, once every probe operator of the join
 * has finished (is_blocked until then), emits the build rows no probe matched, in build-position order: `probe_output_types` channels of
 * nulls followed by the build output channels. */
int32_t tgpu_lookup_outer_factory_create(tgpu_context *ctx, int32_t operator_id, tgpu_lookup_source_factory *bridge, int32_t probe_output_type_count,
                                         const int32_t *probe_output_types, tgpu_operator_factory **out);

/* FilterAndProjectOperator feeding HashAggregationOperator as one fused pipeline (what LocalExecutionPlanner.visitAggregation,
 * M/sql/planner/LocalExecutionPlanner.java:1198,2965-3056, would construct over a filter/project source; the shape of
 * testing/trino-benchmark HandTpchQuery1.java:60-133).  Same results as the two reference operators back to back.  `spec`'s
 * projections form the aggregation's input page: group_by_channels, hash_channel and the aggregates' input / mask channels
 * index those projections.  step is SINGLE or PARTIAL. */
int32_t tgpu_filter_project_hash_aggregation_factory_create(tgpu_context *ctx, int32_t operator_id,
                                                            int32_t input_type_count, const int32_t *input_types,
                                                            const tgpu_page_processor_spec *spec,
                                                            int32_t group_by_count, const int32_t *group_by_types, const int32_t *group_by_channels,
                                                            int32_t hash_channel /* -1 = none */, int32_t step,
                                                            int32_t agg_count, const tgpu_agg_spec *aggs,
                                                            int32_t expected_groups, tgpu_operator_factory **out);

/* TopNOperator.createOperatorFactory (M/operator/TopNOperator.java:47-62; TopNProcessor.java:45-66): the n first rows of the input
 * in the order of the sort channels (S/connector/SortOrder.java:18-21; row order = SimplePageWithPositionComparator.java:58-79:
 * nulls placed by the sort order, values by the type's COMPARISON operator, negated for DESC).  Rows that compare equal on every
 * sort channel come out in input order (the reference's heap leaves their order unspecified).  n == 0: no output, finished at
 * once (TopNOperator.java:154-156). */
typedef enum tgpu_sort_order {
    TGPU_SORT_ASC_NULLS_FIRST = 0, TGPU_SORT_ASC_NULLS_LAST = 1, TGPU_SORT_DESC_NULLS_FIRST = 2, TGPU_SORT_DESC_NULLS_LAST = 3
} tgpu_sort_order;
int32_t tgpu_top_n_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types, int64_t n,
                                  int32_t sort_channel_count, const int32_t *sort_channels, const int32_t *sort_orders,
                                  tgpu_operator_factory **out);

/* OrderByOperator.OrderByOperatorFactory (M/operator/OrderByOperator.java:48-131): every input row, in the order of the sort channels
 * (PagesIndex.sort, M/operator/PagesIndex.java:386-394, with the same row order as TopN above); output_channels selects the channels
 * of the output page.  Rows that compare equal come out in input order (the reference's quicksort leaves their order unspecified).
 * The sorted rows are emitted as one page (the reference cuts them into <= 1 MB pages, :270-296). */
int32_t tgpu_order_by_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types,
                                     int32_t output_channel_count, const int32_t *output_channels, int32_t expected_positions,
                                     int32_t sort_channel_count, const int32_t *sort_channels, const int32_t *sort_orders,
                                     tgpu_operator_factory **out);

/* ---- ScanFilterAndProjectOperator (M/operator/ScanFilterAndProjectOperator.java:66-447): a SOURCE operator that pulls pages from the split's
 * ConnectorPageSource (S/connector/ConnectorPageSource.java; here callbacks) and runs the page processor over them ---- */
typedef struct tgpu_page_source {
    void *user;
    /* ConnectorPageSource.getNextPage: 1 = *page filled (its arrays stay valid until the next call on this source), 0 = no page right now, < 0 error.
     * Channels may be TGPU_LAZY blocks (only type and position_count set). */
    int32_t (*get_next_page)(void *user, tgpu_page *page);
    int32_t (*is_finished)(void *user);                                        /* ConnectorPageSource.isFinished */
    int32_t (*is_blocked)(void *user);                                         /* isBlocked() future not done; NULL = never blocked */
    /* LazyBlock.getLoadedBlock for channel `channel` of the page get_next_page returned last: fills *block with a FLAT / DICTIONARY / RLE block.
     * Called at most once per channel and page, only for channels the filter reads and -- when the filter selected at least one row -- the
     * channels the projections read (PageProcessor.java:111-137; T/operator/project/TestPageProcessor.java:156-184,219-253). */
    int32_t (*load_block)(void *user, int32_t channel, tgpu_block *block);
    void (*close)(void *user);                                                 /* ConnectorPageSource.close; NULL = nothing to do */
} tgpu_page_source;
/* ScanFilterAndProjectOperatorFactory (:449-560) over the page-source flavour (processPageSource :275-287); `types` = the channels the
 * page source produces.  The operator takes no input: tgpu_operator_add_input fails. */
int32_t tgpu_scan_filter_project_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types,
                                                const tgpu_page_processor_spec *spec, tgpu_operator_factory **out);
/* SourceOperator.addSplit (:232-263: the split's page source; one at a time, the next one after the current is finished) / noMoreSplits */
int32_t tgpu_scan_operator_add_page_source(tgpu_operator *op, const tgpu_page_source *source);
int32_t tgpu_scan_operator_no_more_splits(tgpu_operator *op);
/* The record-cursor flavour (processColumnSource / RecordCursorToPages, :267-273,290-351; S/connector/RecordCursor.java): the split's source
 * is a row cursor.  The reference runs a compiled CursorProcessor over it row by row into a PageBuilder; here the rows are read through the
 * callbacks into host columns, up to 65 536 rows at a time (what RecordPageSource does, S/connector/RecordPageSource.java:70-135), and take
 * the page path from there -- the same rows in the same order, cut into other pages (page cuts are not part of the contract).  `types` =
 * the cursor's fields: BIGINT / INTEGER / DATE fields are read with get_long, DOUBLE with get_double, BOOLEAN with get_boolean, VARCHAR
 * with get_slice (bytes valid until the next advance).  close may be NULL. */
typedef struct tgpu_record_cursor {
    void *user;
    int32_t (*advance_next_position)(void *user);      /* RecordCursor.advanceNextPosition: 1 = positioned on a row, 0 = no more rows, < 0 error */
    int32_t (*is_null)(void *user, int32_t field);
    int32_t (*get_boolean)(void *user, int32_t field);
    int64_t (*get_long)(void *user, int32_t field);
    double (*get_double)(void *user, int32_t field);
    int32_t (*get_slice)(void *user, int32_t field, const void **bytes, int32_t *length);   /* 0 = ok, < 0 error */
    int64_t (*completed_bytes)(void *user);            /* RecordCursor.getCompletedBytes; NULL = not tracked */
    void (*close)(void *user);
} tgpu_record_cursor;
int32_t tgpu_scan_operator_add_record_cursor(tgpu_operator *op, const tgpu_record_cursor *cursor, int32_t type_count, const int32_t *types);
/* OperatorStats the scan side feeds (:354-397): positions pulled from the page sources, lazy blocks loaded / skipped */
int32_t tgpu_scan_operator_stats(tgpu_operator *op, int64_t *processed_positions, int64_t *lazy_blocks_loaded, int64_t *lazy_blocks_skipped);

/* OperatorFactory.createOperator / noMoreOperators (M/operator/OperatorFactory.java:18-50) */
int32_t tgpu_operator_factory_create_operator(tgpu_operator_factory *factory, tgpu_operator **out);
int32_t tgpu_operator_factory_no_more_operators(tgpu_operator_factory *factory);
/* OperatorFactory.duplicate() (M/operator/OperatorFactory.java:49): another factory of the same operator; probe-side join factories share
 * the join bridge (the probes are complete once EVERY duplicate has seen noMoreOperators); a hash builder cannot be duplicated
 * (TGPU_ERR_NOT_SUPPORTED, HashBuilderOperator.java:150-152) */
int32_t tgpu_operator_factory_duplicate(tgpu_operator_factory *factory, tgpu_operator_factory **out);
void tgpu_operator_factory_destroy(tgpu_operator_factory *factory);

/* ---- Operator (M/operator/Operator.java:20-102).  Boolean queries return 1/0, or <0 on error. ---- */
int32_t tgpu_operator_needs_input(tgpu_operator *op);
/* Lifetime of `page`: TGPU_HOST arrays have been consumed when the call returns (Java heap arrays are only pinned for the JNI call).
 * TGPU_DEVICE arrays are read by kernels on the context's stream, some of which may still be queued when the call returns (the
 * aggregation's accumulate launch is enqueued behind its group-by probe and not waited for): the caller may overwrite or free them in
 * STREAM ORDER -- by work on that stream, hipFreeAsync on it, hipFree (which synchronises), or after tgpu_context_synchronize -- the
 * rule of any stream-ordered device buffer.  An operator that keeps rows beyond the call copies them or shares the owner
 * (tgpu_operator_add_input_output_page). */
int32_t tgpu_operator_add_input(tgpu_operator *op, const tgpu_page *page);
/* *out = NULL when no page is available (Operator.getOutput() == null); returns TGPU_WOULD_BLOCK instead of TGPU_OK when, in addition,
 * the operator is blocked (a probe waiting for its build side, the outer operator waiting for the probes): the driver should park the
 * pipeline on tgpu_operator_is_blocked instead of spinning (Operator.java:32-35, Driver.java:367-400) */
int32_t tgpu_operator_get_output(tgpu_operator *op, tgpu_output_page **out);
/* Operator.startMemoryRevoke() / finishMemoryRevoke() (M/operator/Operator.java:53-79).  Only a spill-enabled SINGLE / FINAL hash
 * aggregation holds revocable memory: startMemoryRevoke moves its groups out of HBM (below) and has finished when it returns (the
 * reference's future is done); for every other operator state is user memory (tgpu_operator_memory_bytes) and both calls do nothing. */
int32_t tgpu_operator_start_memory_revoke(tgpu_operator *op);
int32_t tgpu_operator_finish_memory_revoke(tgpu_operator *op);
/* OperatorContext.getReservedRevocableBytes(): what startMemoryRevoke would free (the driver revokes while this is > 0,
 * T/operator/OperatorAssertion.java:150-156) */
int64_t tgpu_operator_revocable_memory_bytes(tgpu_operator *op);
/* HashAggregationOperatorFactory(..., spillEnabled, ...) (M/operator/HashAggregationOperator.java:133-154,389-425;
 * M/operator/aggregation/builder/SpillableHashAggregationBuilder.java:47-351), for operators created afterwards, SINGLE and FINAL steps:
 * startMemoryRevoke parks the builder's groups -- keys, raw hashes and the exact accumulator state -- in host memory as one run and
 * starts an empty builder (spillToDisk :283-299); when the output is built the runs are merged through a fresh group-by table, their
 * states added exactly (mergeFromDisk :229-240), and the groups come out in raw-hash order like the reference's merged result
 * (M/operator/MergeHashSort.java).  Accepted by tgpu_hash_aggregation_factory_create and
 * tgpu_filter_project_hash_aggregation_factory_create factories, TGPU_ERR_NOT_SUPPORTED otherwise. */
int32_t tgpu_hash_aggregation_factory_set_spill_enabled(tgpu_operator_factory *factory, int32_t enabled);
/* maxPartialMemory of HashAggregationOperatorFactory (M/operator/HashAggregationOperator.java:128-131; default 16 MB =
 * TaskManagerConfig max_partial_aggregation_memory), for operators created afterwards: a PARTIAL aggregation whose builder is full stops
 * taking input and flushes its groups as an intermediate page (:367-378, :494-497; InMemoryHashAggregationBuilder.isFull :208-215) */
int32_t tgpu_hash_aggregation_factory_set_max_partial_memory(tgpu_operator_factory *factory, int64_t bytes);
/* spills so far (DummySpillerFactory.getSpillsCount in the reference's tests) and the host bytes they hold or held */
int32_t tgpu_operator_spill_stats(tgpu_operator *op, int64_t *spill_count, int64_t *spilled_bytes);
int32_t tgpu_operator_finish(tgpu_operator *op);
int32_t tgpu_operator_is_finished(tgpu_operator *op);
/* addInput with a page another operator of this library produced: the buffers are shared (reference counted), so an operator that keeps
 * or forwards the page (MergePages passing a big page through) does not copy it; `page` may be released right after the call */
int32_t tgpu_operator_add_input_output_page(tgpu_operator *op, const tgpu_output_page *page);
int32_t tgpu_operator_is_blocked(tgpu_operator *op);    /* 1 = isBlocked() future not done (probe waiting for the build) */
int64_t tgpu_operator_memory_bytes(tgpu_operator *op);  /* what the shim reports to LocalMemoryContext.setBytes */
void tgpu_operator_close(tgpu_operator *op);            /* Operator.close(); also frees the handle */

/* ---- output pages (device resident, library owned) ---- */
int32_t tgpu_output_page_position_count(const tgpu_output_page *page);
int32_t tgpu_output_page_channel_count(const tgpu_output_page *page);
/* the page as device-memory blocks (valid until release); lets a downstream GPU operator consume it without a copy */
int32_t tgpu_output_page_as_page(const tgpu_output_page *page, tgpu_page *out);
/* sizes needed to host-materialise channel `ch`: value bytes (VARCHAR: byte-pool size) and whether it may hold nulls */
int32_t tgpu_output_page_block_info(const tgpu_output_page *page, int32_t ch, int32_t *type, int64_t *value_bytes, int32_t *may_have_nulls);
/* D2H copy of one channel into caller buffers: values (value_bytes), nulls (position_count bytes, may be NULL),
 * offsets ((position_count+1) int32, VARCHAR only) */
int32_t tgpu_output_page_copy_block(const tgpu_output_page *page, int32_t ch, void *values, uint8_t *nulls, int32_t *offsets);
/* the same for ALL channels at once (arrays indexed by channel, entries as for copy_block): every transfer is queued through
 * pinned staging and the stream is synchronised once (twice with VARCHAR channels: offsets first, then the bytes) instead of once
 * per buffer -- what the JNI shim uses to turn a GPU output page into heap blocks (S/Page.java:33-73) */
int32_t tgpu_output_page_copy_blocks(const tgpu_output_page *page, int32_t channel_count, void *const *values, uint8_t *const *nulls,
                                     int32_t *const *offsets);
void tgpu_output_page_release(tgpu_output_page *page);

/* ---- GroupByHash (M/operator/GroupByHash.java:45-99; BigintGroupByHash / MultiChannelGroupByHash semantics) ---- */
int32_t tgpu_group_by_hash_create(tgpu_context *ctx, int32_t type_count, const int32_t *types, const int32_t *hash_channels,
                                  int32_t input_hash_channel /* -1 = none */, int32_t expected_size, tgpu_group_by_hash **out);
void tgpu_group_by_hash_destroy(tgpu_group_by_hash *gbh);
int32_t tgpu_group_by_hash_add_page(tgpu_group_by_hash *gbh, const tgpu_page *page);
/* group id of every position (first-seen order, bit-exact with the Java classes) into group_ids[position_count] (host) */
int32_t tgpu_group_by_hash_get_group_ids(tgpu_group_by_hash *gbh, const tgpu_page *page, int64_t *group_ids, int64_t *group_count);
int32_t tgpu_group_by_hash_contains(tgpu_group_by_hash *gbh, int32_t position, const tgpu_page *page, int32_t *result);
int64_t tgpu_group_by_hash_group_count(tgpu_group_by_hash *gbh);
int32_t tgpu_group_by_hash_capacity(tgpu_group_by_hash *gbh);   /* the Java table's capacity for this many groups */
int64_t tgpu_group_by_hash_estimated_size(tgpu_group_by_hash *gbh);
/* appendValuesTo for group ids [0, group_count): key channels (+ raw hash channel when input_hash_channel >= 0) */
int32_t tgpu_group_by_hash_append_values(tgpu_group_by_hash *gbh, tgpu_output_page **out);

/* ---- hash / partition kernels exposed for parity tests and the exchange ---- */
/* InterpretedHashGenerator.hashPosition for every row: hashes[position_count] (host) */
int32_t tgpu_hash_page(tgpu_context *ctx, const tgpu_page *page, int32_t channel_count, const int32_t *channels, int64_t *hashes);
/* PagePartitioner.partitionPage: rows of `page` scattered into `partition_count` contiguous segments, ordered by partition
 * then by input position.  hash_channel >= 0 uses the precomputed raw hash, else the key channels are hashed.
 * partition = (rawHash & 0x7fff...) % partition_count (HashGenerator.java:24-35).  counts[partition_count] (host). */
int32_t tgpu_partition_page(tgpu_context *ctx, const tgpu_page *page, int32_t key_channel_count, const int32_t *key_channels,
                            int32_t hash_channel, int32_t partition_count, int64_t *counts, tgpu_output_page **out);

/* ---- DynamicFilterSourceOperator ---- */
/* M/operator/DynamicFilterSourceOperator.java:74-143 (factory), :145-425 (operator): a pass-through operator on the build side of a join
 * that collects, for every `channels[k]`, what the probe side's scan may be narrowed to: the distinct values while no channel has more
 * than max_distinct_values of them and the collected blocks stay within max_filter_size_in_bytes, else a [min, max] range per orderable
 * channel other than DOUBLE (BIGINT / INTEGER / DATE / BOOLEAN / VARCHAR, :187-190) while at most min_max_collection_limit rows have been
 * seen, else nothing ("all").  Sizes are the reference's block accounting, not TypedSet's JVM retained size (a superset at worst, which an
 * advisory filter allows). */
int32_t tgpu_dynamic_filter_source_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types, int32_t channel_count,
                                                  const int32_t *channels, int32_t max_distinct_values, int64_t max_filter_size_in_bytes,
                                                  int32_t min_max_collection_limit, tgpu_operator_factory **out);
typedef enum tgpu_dynamic_filter_kind { TGPU_DF_ALL = 0, TGPU_DF_VALUES = 1, TGPU_DF_RANGE = 2, TGPU_DF_NONE = 3 } tgpu_dynamic_filter_kind;
/* after finish(): the Domain of filter channel k that DynamicFilterSourceOperator.finish() hands to its consumer (:383-424):
 * VALUES -> *values = a one-channel page of the distinct non-null, non-NaN values (first-seen order; release it as usual);
 * RANGE -> [*min, *max] (BOOLEAN as 0 / 1); a VARCHAR channel's range comes as *values = a one-channel page of two rows, min then max;
 * NONE -> the channel only saw nulls; ALL -> no constraint */
int32_t tgpu_dynamic_filter_source_result(tgpu_operator *op, int32_t filter_channel, int32_t *kind, tgpu_output_page **values, int64_t *min, int64_t *max);

/* ---- MergePages as an operator: page coalescing in HBM ---- */
/* M/operator/project/MergePages.java:64-190 (MergePagesTransformation.process): a page with >= min_row_count rows or >= min_page_size_in_bytes
 * bytes passes through as it is, after the buffered rows; smaller pages are appended to a device buffer that is flushed when it reaches
 * max_page_size_in_bytes (PageBuilder.isFull) or at finish().  Sizes follow the reference's accounting ((width + 1) bytes per fixed-width
 * cell, length + 5 per VARCHAR cell).  The GPU operators want pages of tens of MB and more (DESIGN.md "Page granularity"): put this
 * operator in front of them with large thresholds; the reference's 1 MB cap on min_page_size_in_bytes (MergePages.java:58) is not enforced. */
int32_t tgpu_merge_pages_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types, int64_t min_page_size_in_bytes,
                                        int32_t min_row_count, int64_t max_page_size_in_bytes, tgpu_operator_factory **out);

/* ---- PartitionedOutputOperator: the shuffle producer behind the Operator API ---- */
/* M/operator/PartitionedOutputOperator.java:46-300, PagePartitioner :308-486.  A sink operator (getOutput() returns nothing, :303-306):
 * addInput groups the page's rows by partition = (rawHash & 0x7fff...) % partition_count (HashGenerator.java:24-35) of `hash_channel`
 * (>= 0: the precomputed raw hash) or of the `partition_channels` (InterpretedHashGenerator), rows in input order inside each
 * partition; a row goes to EVERY partition when `null_channel` >= 0 is null there, and so does the first row ever seen when
 * `replicates_any_row` is set (:411-418).  Constant partitioning arguments (:433-448): TGPU_ERR_NOT_SUPPORTED. */
typedef enum tgpu_partition_function {
    TGPU_PARTITION_HASH_MODULO = 0, /* remote exchanges: (rawHash & 0x7fff...) % partition_count, M/operator/HashGenerator.java:24-35 */
    TGPU_PARTITION_LOCAL = 1        /* LocalExchange (M/operator/exchange/PartitioningExchanger.java): (int) XxHash64.hash(Long.reverse(rawHash)) &
                                     * (partition_count - 1), partition_count a power of two, M/operator/exchange/LocalPartitionGenerator.java:45-65 */
} tgpu_partition_function;
int32_t tgpu_partitioned_output_factory_create(tgpu_context *ctx, int32_t operator_id, int32_t type_count, const int32_t *types, int32_t partition_channel_count,
                                               const int32_t *partition_channels, int32_t hash_channel /* -1 = hash the channels */, int32_t partition_count,
                                               int32_t replicates_any_row, int32_t null_channel /* -1 = none */, int32_t partition_function,
                                               tgpu_operator_factory **out);
/* what the reference enqueues into its OutputBuffer (PagePartitioner.flush :451-470: outputBuffer.enqueue(partition, pages)): the next
 * pending (partition, page) pair in enqueue order, device resident (serialize it with tgpu_serialize_page or hand it to a GPU
 * exchange); *out = NULL when nothing is pending */
int32_t tgpu_partitioned_output_poll(tgpu_operator *op, int32_t *partition, tgpu_output_page **out);
/* PartitionedOutputInfo (:396-399) */
int32_t tgpu_partitioned_output_info(tgpu_operator *op, int64_t *rows_added, int64_t *pages_added);

/* ---- SerializedPage <-> HBM: the reference's exchange / spill page format as the ingest / egress format ---- */
/* PagesSerde.serialize + PagesSerdeUtil.writeSerializedPage (M/execution/buffer/PagesSerde.java:64-115, PagesSerdeUtil.java:45-71;
 * block bodies: S/block/LongArrayBlockEncoding.java:37-61, IntArrayBlockEncoding, ByteArrayBlockEncoding, VariableWidthBlockEncoding.java:37-61,
 * null bits S/block/EncoderUtil.java:33-71) of `page` into `out` (host memory, `capacity` bytes): positionCount | markers (none) |
 * uncompressedSize | sizeInBytes | payload, byte for byte what the Java serde writes for the same flat blocks (one representation
 * detail aside: a null vector that holds no null is dropped at ingest, so such a block is written with mayHaveNull = 0 where Java
 * keeps the flag of an all-false valueIsNull array -- both decode to equal blocks).  *out_len = bytes
 * written.  out == NULL: *out_len = an upper bound of the size (nothing is computed), for sizing the buffer. */
int32_t tgpu_serialize_page(tgpu_context *ctx, const tgpu_page *page, void *out, int64_t capacity, int64_t *out_len);
/* PagesSerde.deserialize (PagesSerde.java:117-160) of one uncompressed, unencrypted SerializedPage in host memory, straight into a
 * device-resident page: LONG_ARRAY / INT_ARRAY / BYTE_ARRAY / VARIABLE_WIDTH blocks, RLE and DICTIONARY (RunLengthBlockEncoding.java:31-53,
 * DictionaryBlockEncoding.java:33-80) flattened on the device.  `types` = the tgpu_type of every channel (the encodings do not tell
 * BIGINT from DOUBLE, INTEGER from DATE).  A COMPRESSED page (exchange.compression-enabled: one LZ4 block, PagesSerde.java:73-93,153-165) is
 * inflated on the device first; ENCRYPTED: TGPU_ERR_NOT_SUPPORTED.  (tgpu_serialize_page always writes uncompressed pages, which every reader accepts.) */
int32_t tgpu_deserialize_page(tgpu_context *ctx, const void *bytes, int64_t len, int32_t type_count, const int32_t *types, tgpu_output_page **out);

/* ---- scan-side decode, first slice: ORC stripe streams -> device-resident blocks ---- */
/* What an ORC page source does per column and stripe / row group (lib/trino-orc/src/main/java/io/trino/orc/reader/LongColumnReader.java:100-230,
 * BooleanColumnReader.java, SliceDictionaryColumnReader.java:120-330 over stream/LongInputStreamV2.java:59-312, LongBitPacker.java:82-108,
 * ByteInputStream.java:43-75, BooleanInputStream.java:36-58), on the device: the streams are handed over DECOMPRESSED in host memory (the
 * chunk framing and its codecs stay with the file reader), `present` = the PRESENT stream or NULL (no nulls); *out = a one-channel page.
 * `encoding` = the column's ColumnEncoding kind: the integer streams of DIRECT / DICTIONARY columns (files written before Hive 0.12) are RLEv1
 * (stream/LongInputStreamV1.java:47-103), those of DIRECT_V2 / DICTIONARY_V2 columns RLEv2. */
typedef enum tgpu_orc_encoding { TGPU_ORC_DIRECT = 0, TGPU_ORC_DICTIONARY = 1, TGPU_ORC_DIRECT_V2 = 2, TGPU_ORC_DICTIONARY_V2 = 3 } tgpu_orc_encoding;
/* SHORT / INT / LONG / DATE columns: DATA = signed RLEv2; type = TGPU_BIGINT, TGPU_INTEGER or TGPU_DATE (32-bit types check the range like
 * LongInputStreamV2.next(int[]) :356-364) */
int32_t tgpu_orc_decode_long_column(tgpu_context *ctx, int32_t type, int32_t encoding, int32_t position_count, const void *present, int64_t present_len,
                                    const void *data, int64_t data_len, tgpu_output_page **out);
/* BOOLEAN columns: DATA = a boolean stream */
int32_t tgpu_orc_decode_boolean_column(tgpu_context *ctx, int32_t position_count, const void *present, int64_t present_len, const void *data, int64_t data_len,
                                       tgpu_output_page **out);
/* DOUBLE columns (reader/DoubleColumnReader.java:92-175): DATA = the non-null rows' doubles, 8 little-endian bytes each */
int32_t tgpu_orc_decode_double_column(tgpu_context *ctx, int32_t position_count, const void *present, int64_t present_len, const void *data, int64_t data_len,
                                      tgpu_output_page **out);
/* STRING / VARCHAR / CHAR columns in DICTIONARY_V2 encoding: DATA = unsigned RLEv2 ids, LENGTH = unsigned RLEv2 lengths of the dictionary_size
 * entries, DICTIONARY_DATA = their bytes; the result is a flat VARCHAR block */
int32_t tgpu_orc_decode_dictionary_string_column(tgpu_context *ctx, int32_t encoding, int32_t position_count, const void *present, int64_t present_len,
                                                 const void *data, int64_t data_len, int32_t dictionary_size, const void *length_stream, int64_t length_len,
                                                 const void *dictionary_data, int64_t dictionary_data_len, tgpu_output_page **out);

/* STRING / VARCHAR / CHAR columns in DIRECT_V2 encoding (reader/SliceDirectColumnReader.java:100-232): LENGTH = unsigned RLEv2 lengths of the
 * non-null rows, DATA = their bytes back to back; the result is a flat VARCHAR block */
int32_t tgpu_orc_decode_direct_string_column(tgpu_context *ctx, int32_t encoding, int32_t position_count, const void *present, int64_t present_len,
                                             const void *data, int64_t data_len, const void *length_stream, int64_t length_len, tgpu_output_page **out);

/* ---- scan-side decode, the second columnar format: Parquet data pages -> device-resident blocks ---- */
/* What a Parquet page source does per data page of a FLAT column (lib/trino-parquet/src/main/java/io/trino/parquet/reader/PrimitiveColumnReader.java
 * readPageV1 / readPageV2 / initDataReader, LevelRLEReader.java, ParquetEncoding.java, the dictionary package, reader/{Int,Long,Double,Boolean,Binary}ColumnReader.java;
 * the byte-level decoders are parquet-mr's, restated from the Parquet format specification), on the device.  The page arrives DECOMPRESSED and
 * taken apart by the file reader (thrift page headers and codecs stay there): `definition_levels` = the levels' RLE / bit-packed hybrid of bit
 * width 1 WITHOUT the 4-byte length a V1 page carries in front of it, NULL for a required column; `values` = the value section;
 * `dictionary` / `dictionary_count` = the column chunk's PLAIN dictionary page for the dictionary encodings, else NULL / 0.
 * `physical` = parquet.thrift Type (0 BOOLEAN, 1 INT32, 2 INT64, 5 DOUBLE, 6 BYTE_ARRAY) with `type` INTEGER / DATE, BIGINT, DOUBLE, BOOLEAN, VARCHAR;
 * `encoding` = parquet.thrift Encoding (0 PLAIN, 2 PLAIN_DICTIONARY, 8 RLE_DICTIONARY; 3 RLE for BOOLEAN values; 5 DELTA_BINARY_PACKED for INT32 and INT64, 6 DELTA_LENGTH_BYTE_ARRAY and 7 DELTA_BYTE_ARRAY for BYTE_ARRAY); anything else: TGPU_ERR_NOT_SUPPORTED.  *out = a one-channel page. */
int32_t tgpu_parquet_decode_data_page(tgpu_context *ctx, int32_t type, int32_t physical, int32_t encoding, int32_t position_count, const void *definition_levels,
                                      int64_t definition_levels_len, const void *values, int64_t values_len, const void *dictionary, int64_t dictionary_len,
                                      int32_t dictionary_count, tgpu_output_page **out);

/* ---- exchange between the GPUs of one node ---- */
/* What replaces PartitionedOutputOperator -> OutputBuffer -> HTTP -> ExchangeOperator (M/operator/PartitionedOutputOperator.java:406-476,
 * M/operator/ExchangeOperator.java) when the consumers of a FIXED_HASH_DISTRIBUTION / FIXED_BROADCAST_DISTRIBUTION stage
 * (M/sql/planner/SystemPartitioningHandle.java:59-60) are the GPUs of one node, one rank (process, context) per GPU: pages stay in HBM
 * and travel over xGMI.  One header all-to-all per page (row counts, null-vector flags, VARCHAR byte counts), then ONE grouped RCCL
 * ncclSend / ncclRecv exchange carrying every buffer of every channel to every peer.  Every call is collective: all ranks of the
 * exchange make the same calls in the same order (as the stages of one query do). */
typedef struct tgpu_exchange tgpu_exchange;
#define TGPU_EXCHANGE_ID_BYTES 128
/* rank 0 creates the id and hands it to the other ranks out of band (in the engine: with the task's exchange locations) */
int32_t tgpu_exchange_unique_id(void *id_out /* TGPU_EXCHANGE_ID_BYTES */);
int32_t tgpu_exchange_create(tgpu_context *ctx, const void *unique_id, int32_t rank, int32_t world, tgpu_exchange **out);
/* The same exchange over a transport of the caller's (rehearsals of several ranks on one GPU -- RCCL refuses two ranks on one device --
 * and tests).  all_to_all_meta: host int64 values, per_rank of them per destination / source.  all_to_all_v: device buffers; entry
 * [t * world + r] of the four arrays describes what goes to / comes from rank r in transfer t; the library has synchronised its stream
 * before the call and expects the bytes in place when it returns.  Entries for the caller's own rank are zero (handled inside). */
typedef struct tgpu_exchange_transport {
    void *user;
    int32_t (*all_to_all_meta)(void *user, const int64_t *send, int64_t *recv, int32_t per_rank);
    int32_t (*all_to_all_v)(void *user, int32_t transfers, const void *const *send_ptr, const int64_t *send_bytes, void *const *recv_ptr, const int64_t *recv_bytes);
} tgpu_exchange_transport;
int32_t tgpu_exchange_create_with_transport(tgpu_context *ctx, int32_t rank, int32_t world, const tgpu_exchange_transport *transport, tgpu_exchange **out);
void tgpu_exchange_destroy(tgpu_exchange *ex);
/* hash repartition of `page`: rows go to rank (rawHash & 0x7fff...) % world (M/operator/HashGenerator.java:24-35) of the key channels
 * (hash_channel >= 0: of that precomputed raw hash); *out = the rows this rank owns, grouped by source rank, input order kept */
int32_t tgpu_exchange_repartition(tgpu_exchange *ex, const tgpu_page *page, int32_t key_channel_count, const int32_t *key_channels, int32_t hash_channel,
                                  tgpu_output_page **out);
/* the pages a PartitionedOutputOperator with partition_count == world has pending (tgpu_partitioned_output_poll) shuffled to their
 * ranks: the shuffle producer and the exchange back to back without leaving the library */
int32_t tgpu_exchange_partitioned_output(tgpu_exchange *ex, tgpu_operator *partitioned_output_operator, int32_t type_count, const int32_t *types,
                                         tgpu_output_page **out);
/* broadcast (replicated join build side): every rank receives the pages of all ranks, concatenated in rank order */
int32_t tgpu_exchange_all_gather(tgpu_exchange *ex, const tgpu_page *page, tgpu_output_page **out);
/* bytes this rank has sent to OTHER ranks so far (xGMI traffic; bench.py's exchange_bytes_sent_per_step) */
int64_t tgpu_exchange_bytes_sent(tgpu_exchange *ex);

#ifdef __cplusplus
}
#endif
#endif /* TGPU_H */
