"""MI355X-native operator hot path for Trino (filter/project, hash aggregation, hash join) behind the C ABI in
include/tgpu.h.  The package directory name is not a Python identifier: import it with
importlib.import_module("presto-1_amd").  There is no CPU fallback: importing works anywhere, but every compute call
needs libtgpu.so (built by __graft_entry__.build()) and a HIP device.
"""
from . import _lib, expressions, operators, spi
from ._lib import SO_PATH, TgpuError, build
from .expressions import (and_, between, call, cast, coalesce, constant, field, if_, is_null, not_, or_)
from .operators import (ASC_NULLS_FIRST, ASC_NULLS_LAST, DESC_NULLS_FIRST, DESC_NULLS_LAST, DynamicFilterSourceOperatorFactory, MergePagesOperatorFactory, OrderByOperatorFactory, PartitionedOutputOperator, PartitionedOutputOperatorFactory, TopNOperatorFactory, AVG_BIGINT, AVG_DOUBLE, MIN_BIGINT, MAX_BIGINT, MIN_DOUBLE, MAX_DOUBLE, SUM_ORDER_EXACT, SUM_ORDER_JAVA, COUNT_ALL, COUNT_COLUMN, FINAL, FULL_OUTER, INNER, LOOKUP_OUTER, PARTIAL, PROBE_OUTER, SINGLE, SUM_BIGINT, SUM_DOUBLE,
                        Context, FilterAndProjectOperatorFactory, ScanFilterAndProjectOperatorFactory, PageSource, RecordCursor, FilterProjectHashAggregationOperatorFactory, FilterProjectLookupJoinOperatorFactory, GroupByHash, HashAggregationOperatorFactory, HashBuilderOperatorFactory,
                        LookupJoinOperatorFactory, LookupOuterOperatorFactory, Operator, OperatorFactory, page_processor_source, precompile_fused_aggregation, precompile_fused_probe, precompile_page_processor, to_pages)
from .spi import (BIGINT, BOOLEAN, DATE, DOUBLE, INTEGER, VARCHAR, Block, DeviceBlock, DictionaryBlock, LazyBlock, OutputPage, Page, RunLengthEncodedBlock)
