"""Host-side mirror of the reference's operator API over the C ABI (include/tgpu.h).

Same names and call protocol as io.trino.operator.Operator / OperatorFactory (M/operator/Operator.java:20-102,
OperatorFactory.java:18-50) so parity tests read like the reference's own operator tests
(T/operator/TestHashAggregationOperator.java, TestHashJoinOperator.java, TestFilterAndProjectOperator.java).
Everything here is a thin wrapper: the state machines live in csrc/operators.cpp, the work in the HIP kernels.
"""
import ctypes as C
import json

import numpy as np

from . import _lib
from .expressions import FlatProgram
from .spi import OutputPage, Page

COUNT_ALL, COUNT_COLUMN, SUM_BIGINT, SUM_DOUBLE, AVG_BIGINT, AVG_DOUBLE, MIN_BIGINT, MAX_BIGINT, MIN_DOUBLE, MAX_DOUBLE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
SINGLE, PARTIAL, FINAL = 0, 1, 2
# S/connector/SortOrder.java:18-21
ASC_NULLS_FIRST, ASC_NULLS_LAST, DESC_NULLS_FIRST, DESC_NULLS_LAST = 0, 1, 2, 3
INNER, PROBE_OUTER, LOOKUP_OUTER, FULL_OUTER = 0, 1, 2, 3   # LookupJoinOperators.JoinType ordinals
SUM_ORDER_EXACT, SUM_ORDER_JAVA = 0, 1                      # tgpu_double_sum_order


def _i32(seq):
    seq = list(seq)
    return (C.c_int32 * max(1, len(seq)))(*seq), len(seq)


class Context:
    """One HIP device + stream (tgpu_context).  stream: a hipStream_t address (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_context_create(device, stream, C.byref(h)))
        self.handle = h

    def synchronize(self):
        _lib.check(_lib.lib().tgpu_context_synchronize(self.handle))

    def pinned_array(self, dtype, count):
        """a numpy array over page-locked host memory (tgpu_pinned_alloc): blocks built on it are transferred by asynchronous DMA;
        the array keeps the allocation alive and frees it with the context still open (call .base.free() or let it be collected)"""
        dt = np.dtype(dtype)
        p = C.c_void_p()
        _lib.check(_lib.lib().tgpu_pinned_alloc(self.handle, int(count) * dt.itemsize, C.byref(p)))
        ctx_handle = self.handle

        class _Pinned:
            def __init__(self):
                self.ptr = p.value
                self.__array_interface__ = {"shape": (int(count),), "typestr": dt.str, "data": (p.value, False), "version": 3}

            def free(self):
                if self.ptr:
                    _lib.lib().tgpu_pinned_free(ctx_handle, self.ptr)
                    self.ptr = None

        holder = _Pinned()
        return np.asarray(holder), holder

    def set_max_output_page(self, max_bytes=0, max_rows=0):
        """operators hand pages larger than this out as consecutive regions (PageBuilder.isFull granularity for Java consumers); 0 = no limit"""
        _lib.check(_lib.lib().tgpu_context_set_max_output_page(self.handle, int(max_bytes), int(max_rows)))

    def set_double_sum_order(self, order):
        """SUM_ORDER_EXACT (default) or SUM_ORDER_JAVA for the aggregation operators created from now on (tgpu.h)"""
        _lib.check(_lib.lib().tgpu_context_set_double_sum_order(self.handle, order))

    def set_device_input_stable(self, stable=True):
        """the promise of tgpu_context_set_device_input_stable: borrowed device blocks stay valid and unchanged until the operator they were given to has finished"""
        _lib.check(_lib.lib().tgpu_context_set_device_input_stable(self.handle, int(bool(stable))))

    def profile_enable(self, on=True):
        _lib.check(_lib.lib().tgpu_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        _lib.check(_lib.lib().tgpu_profile_reset(self.handle))

    def profile(self):
        L = _lib.lib()
        need = L.tgpu_profile_dump(self.handle, None, 0)
        buf = C.create_string_buffer(int(need))
        L.tgpu_profile_dump(self.handle, buf, need)
        return json.loads(buf.value.decode())

    def hash_page(self, page: Page, channels):
        ch, n = _i32(channels)
        out = np.zeros(max(page.position_count, 1), dtype=np.int64)
        cp, keep = page.to_c()
        _lib.check(_lib.lib().tgpu_hash_page(self.handle, C.byref(cp), n, ch, out.ctypes.data))
        return out[: page.position_count]

    def partition_page(self, page: Page, key_channels, partition_count, hash_channel=-1):
        ch, n = _i32(key_channels)
        counts = np.zeros(partition_count, dtype=np.int64)
        out = C.c_void_p()
        cp, keep = page.to_c()
        _lib.check(_lib.lib().tgpu_partition_page(self.handle, C.byref(cp), n, ch, hash_channel, partition_count, counts.ctypes.data, C.byref(out)))
        return counts, OutputPage(out)

    def serialize_page(self, page: Page, into=None):
        """PagesSerde.serialize + writeSerializedPage (M/execution/buffer/PagesSerde.java:64-115): the page as the reference's wire
        bytes.  `into` (a uint8 numpy buffer, e.g. a reused exchange buffer): written in place, returns the byte count."""
        cp, keep = page.to_c()
        need = C.c_int64()
        if into is not None:
            _lib.check(_lib.lib().tgpu_serialize_page(self.handle, C.byref(cp), into.ctypes.data, into.size, C.byref(need)))
            return int(need.value)
        _lib.check(_lib.lib().tgpu_serialize_page(self.handle, C.byref(cp), None, 0, C.byref(need)))
        buf = np.empty(max(int(need.value), 1), dtype=np.uint8)
        _lib.check(_lib.lib().tgpu_serialize_page(self.handle, C.byref(cp), buf.ctypes.data, buf.size, C.byref(need)))
        return buf[: need.value].tobytes()

    def deserialize_page(self, data: bytes, types):
        """PagesSerde.deserialize (PagesSerde.java:117-160) into a device-resident OutputPage; `types` = channel types"""
        t, n = _i32(types)
        raw = np.frombuffer(data, dtype=np.uint8)
        out = C.c_void_p()
        _lib.check(_lib.lib().tgpu_deserialize_page(self.handle, raw.ctypes.data, raw.size, n, t, C.byref(out)))
        return OutputPage(out)

    def close(self):
        if self.handle:
            _lib.lib().tgpu_context_destroy(self.handle)
            self.handle = None


class Operator:
    def __init__(self, handle):
        self.handle = handle

    def needsInput(self):
        return bool(_lib.check(_lib.lib().tgpu_operator_needs_input(self.handle)))

    def addInput(self, page):
        if isinstance(page, OutputPage):   # a page of this library: buffers shared, nothing copied (tgpu_operator_add_input_output_page)
            _lib.check(_lib.lib().tgpu_operator_add_input_output_page(self.handle, page.handle))
            return
        cp, keep = page.to_c()
        _lib.check(_lib.lib().tgpu_operator_add_input(self.handle, C.byref(cp)))

    def getOutput(self):
        out = C.c_void_p()
        self.last_get_output_status = _lib.check(_lib.lib().tgpu_operator_get_output(self.handle, C.byref(out)))   # 1 = TGPU_WOULD_BLOCK
        return OutputPage(out) if out.value else None

    def startMemoryRevoke(self):
        _lib.check(_lib.lib().tgpu_operator_start_memory_revoke(self.handle))

    def finishMemoryRevoke(self):
        _lib.check(_lib.lib().tgpu_operator_finish_memory_revoke(self.handle))

    def revocableMemoryBytes(self):
        """OperatorContext.getReservedRevocableBytes()"""
        return int(_lib.lib().tgpu_operator_revocable_memory_bytes(self.handle))

    def spillStats(self):
        """(spills so far, host bytes spilled)"""
        n, b = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().tgpu_operator_spill_stats(self.handle, C.byref(n), C.byref(b)))
        return n.value, b.value

    def finish(self):
        _lib.check(_lib.lib().tgpu_operator_finish(self.handle))

    def isFinished(self):
        return bool(_lib.check(_lib.lib().tgpu_operator_is_finished(self.handle)))

    def isBlocked(self):
        return bool(_lib.check(_lib.lib().tgpu_operator_is_blocked(self.handle)))

    def memoryBytes(self):
        return _lib.lib().tgpu_operator_memory_bytes(self.handle)

    def close(self):
        if self.handle:
            _lib.lib().tgpu_operator_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # a constructor that failed before `handle` existed, or interpreter shutdown
            pass


class OperatorFactory:
    def __init__(self, handle, keep=None):
        self.handle = handle
        self._keep = keep

    def createOperator(self):
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_operator_factory_create_operator(self.handle, C.byref(h)))
        return Operator(h)

    def noMoreOperators(self):
        _lib.check(_lib.lib().tgpu_operator_factory_no_more_operators(self.handle))

    def setMaxPartialMemory(self, nbytes):
        """maxPartialMemory of HashAggregationOperatorFactory (M/operator/HashAggregationOperator.java:128-131), for operators created afterwards"""
        _lib.check(_lib.lib().tgpu_hash_aggregation_factory_set_max_partial_memory(self.handle, int(nbytes)))

    def setSpillEnabled(self, on=True):
        """spillEnabled of HashAggregationOperatorFactory (M/operator/HashAggregationOperator.java:133), for operators created afterwards;
        hash aggregation factories only (NOT_SUPPORTED otherwise)"""
        _lib.check(_lib.lib().tgpu_hash_aggregation_factory_set_spill_enabled(self.handle, int(on)))

    def duplicate(self):
        """OperatorFactory.duplicate() (M/operator/OperatorFactory.java:49)"""
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_operator_factory_duplicate(self.handle, C.byref(h)))
        dup = type(self).__new__(type(self))
        OperatorFactory.__init__(dup, h, self._keep)
        for k, v in self.__dict__.items():
            if k not in ("handle", "_keep"):
                setattr(dup, k, v)
        return dup

    def close(self):
        if self.handle:
            _lib.lib().tgpu_operator_factory_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # a constructor that failed before `handle` existed, or interpreter shutdown
            pass


class FilterAndProjectOperatorFactory(OperatorFactory):
    """FilterAndProjectOperator.createOperatorFactory (M/operator/FilterAndProjectOperator.java:73-88): the PageProcessor
    (filter + projections as RowExpressions) is compiled into one fused gfx950 kernel pair."""

    def __init__(self, ctx: Context, operator_id, input_types, filter_expr, projections):
        self.program = FlatProgram(filter_expr, projections)
        spec, keep = self.program.to_c()
        types, n = _i32(input_types)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_filter_project_factory_create(ctx.handle, operator_id, n, types, C.byref(spec), C.byref(h)))
        super().__init__(h, keep)


class PageSource:
    """a ConnectorPageSource (S/connector/ConnectorPageSource.java) over Python pages, as the callbacks of tgpu_page_source"""

    def __init__(self, pages, blocked_polls=0):
        self.pages = list(pages)
        self.at = 0
        self.blocked_polls = blocked_polls      # isBlocked() reports "not done" this many times first
        self.current = None
        self.closed = False
        self._keep = None
        self._load_keep = []
        self.error = None
        self.struct = _lib.PageSource(None, _lib.SOURCE_NEXT_FN(self._next), _lib.SOURCE_FLAG_FN(self._finished), _lib.SOURCE_FLAG_FN(self._blocked),
                                      _lib.SOURCE_LOAD_FN(self._load), _lib.SOURCE_CLOSE_FN(self._close))

    def _next(self, user, out):
        try:
            if self.at >= len(self.pages):
                return 0
            self.current = self.pages[self.at]
            self.at += 1
            self._load_keep = []
            cp, self._keep = self.current.to_c()   # valid until the next call
            out[0].position_count, out[0].channel_count, out[0].blocks = cp.position_count, cp.channel_count, cp.blocks
            return 1
        except Exception as e:   # must not unwind through the C frames
            self.error = e
            return -1

    def _finished(self, user):
        return 1 if self.at >= len(self.pages) else 0

    def _blocked(self, user):
        if self.blocked_polls > 0:
            self.blocked_polls -= 1
            return 1
        return 0

    def _load(self, user, channel, out):
        try:
            blk = self.current.blocks[channel].load()
            blk._fill(out[0], self._load_keep)
            return 0
        except Exception as e:
            self.error = e
            return -1

    def _close(self, user):
        self.closed = True


class RecordCursor:
    """a RecordCursor (S/connector/RecordCursor.java) over Python rows (tuples; None = null), as the callbacks of tgpu_record_cursor"""

    def __init__(self, types, rows):
        self.types = list(types)
        self.rows = list(rows)
        self.at = -1
        self.closed = False
        self.error = None
        self._slice_keep = None
        self.struct = _lib.RecordCursor(None, _lib.SOURCE_FLAG_FN(self._advance), _lib.CURSOR_FIELD_FN(self._is_null), _lib.CURSOR_FIELD_FN(self._boolean),
                                        _lib.CURSOR_LONG_FN(self._long), _lib.CURSOR_DOUBLE_FN(self._double), _lib.CURSOR_SLICE_FN(self._slice),
                                        _lib.CURSOR_BYTES_FN(self._bytes), _lib.SOURCE_CLOSE_FN(self._close))

    def _advance(self, user):
        self.at += 1
        return 1 if self.at < len(self.rows) else 0

    def _is_null(self, user, field):
        return 1 if self.rows[self.at][field] is None else 0

    def _boolean(self, user, field):
        return 1 if self.rows[self.at][field] else 0

    def _long(self, user, field):
        return int(self.rows[self.at][field])

    def _double(self, user, field):
        return float(self.rows[self.at][field])

    def _slice(self, user, field, out_bytes, out_len):
        try:
            v = self.rows[self.at][field]
            b = v.encode("utf-8") if isinstance(v, str) else bytes(v)
            self._slice_keep = C.create_string_buffer(b, len(b)) if b else None   # valid until the next advance
            out_bytes[0] = C.cast(self._slice_keep, C.c_void_p).value if b else None
            out_len[0] = len(b)
            return 0
        except Exception as e:   # must not unwind through the C frames
            self.error = e
            return -1

    def _bytes(self, user):
        return 0

    def _close(self, user):
        self.closed = True


class ScanOperator(Operator):
    def addSplit(self, source):
        """SourceOperator.addSplit: the split's page source (PageSource) or record cursor (RecordCursor)"""
        self._sources = getattr(self, "_sources", []) + [source]
        if isinstance(source, RecordCursor):
            t, n = _i32(source.types)
            _lib.check(_lib.lib().tgpu_scan_operator_add_record_cursor(self.handle, C.byref(source.struct), n, t))
            return
        _lib.check(_lib.lib().tgpu_scan_operator_add_page_source(self.handle, C.byref(source.struct)))

    def noMoreSplits(self):
        _lib.check(_lib.lib().tgpu_scan_operator_no_more_splits(self.handle))

    def stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().tgpu_scan_operator_stats(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return {"processedPositions": a.value, "lazyBlocksLoaded": b.value, "lazyBlocksSkipped": c.value}


class ScanFilterAndProjectOperatorFactory(OperatorFactory):
    """ScanFilterAndProjectOperator.ScanFilterAndProjectOperatorFactory (M/operator/ScanFilterAndProjectOperator.java:449-560), page-source flavour"""

    def __init__(self, ctx: Context, operator_id, types, filter_expr, projections):
        self.program = FlatProgram(filter_expr, projections)
        spec, keep = self.program.to_c()
        t, n = _i32(types)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_scan_filter_project_factory_create(ctx.handle, operator_id, n, t, C.byref(spec), C.byref(h)))
        super().__init__(h, keep)

    def createOperator(self):
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_operator_factory_create_operator(self.handle, C.byref(h)))
        return ScanOperator(h)


def precompile_page_processor(input_types, filter_expr, projections):
    """Compile the kernels of a page processor into the on-disk cache (no GPU needed)."""
    prog = FlatProgram(filter_expr, projections)
    spec, keep = prog.to_c()
    types, n = _i32(input_types)
    _lib.check(_lib.lib().tgpu_precompile_page_processor(n, types, C.byref(spec)))


def page_processor_source(input_types, filter_expr, projections):
    prog = FlatProgram(filter_expr, projections)
    spec, keep = prog.to_c()
    types, n = _i32(input_types)
    L = _lib.lib()
    need = _lib.check(L.tgpu_page_processor_source(n, types, C.byref(spec), None, 0))
    buf = C.create_string_buffer(int(need))
    L.tgpu_page_processor_source(n, types, C.byref(spec), buf, need)
    return buf.value.decode()


class HashAggregationOperatorFactory(OperatorFactory):
    """M/operator/HashAggregationOperator.java:54-262.  aggs: list of (function, input_channel, mask_channel)."""

    def __init__(self, ctx: Context, operator_id, group_by_types, group_by_channels, aggs, step=SINGLE, hash_channel=-1,
                 expected_groups=10_000, produce_default_output=False, spill_enabled=False):
        gt, ng = _i32(group_by_types)
        gc, _ = _i32(group_by_channels)
        arr = (_lib.AggSpec * max(1, len(aggs)))()
        for i, a in enumerate(aggs):
            f, ch = a[0], a[1]
            arr[i] = _lib.AggSpec(f, ch, a[2] if len(a) > 2 else -1)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_hash_aggregation_factory_create(ctx.handle, operator_id, ng, gt, gc, hash_channel, step, len(aggs), arr,
                                                                   expected_groups, int(produce_default_output), C.byref(h)))
        super().__init__(h)
        if spill_enabled:
            self.setSpillEnabled(True)


class LookupSourceFactory:
    """The JoinBridge between the build and probe pipelines (M/operator/PartitionedLookupSourceFactory.java)."""

    def __init__(self, handle):
        self.handle = handle

    def setJoinFilter(self, probe_types, filter_expr):
        """JoinFilterFunction (M/operator/JoinHash.java:44-47,118-130): `filter_expr` over (build channels..., probe channels...)"""
        prog = FlatProgram(filter_expr, [])
        spec, keep = prog.to_c()
        t, nt = _i32(probe_types)
        _lib.check(_lib.lib().tgpu_lookup_source_factory_set_join_filter(self.handle, nt, t, C.byref(spec)))

    def stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().tgpu_lookup_source_stats(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return dict(positions=a.value, hash_size=b.value, link_count=c.value)

    def close(self):
        if self.handle:
            _lib.lib().tgpu_lookup_source_factory_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # a constructor that failed before `handle` existed, or interpreter shutdown
            pass


class HashBuilderOperatorFactory(OperatorFactory):
    """M/operator/HashBuilderOperator.java:54-152.  `lookup_source_factory` is the bridge to hand to the probe factory."""

    def __init__(self, ctx: Context, operator_id, types, output_channels, hash_channels, precomputed_hash_channel=-1, expected_positions=100, partition_count=1):
        """partition_count > 1: createOperator() hands out that many build operators, one per local-exchange partition
        (PartitionedLookupSourceFactory.java:110-124)"""
        t, nt = _i32(types)
        oc, no = _i32(output_channels)
        hc, nh = _i32(hash_channels)
        bridge, h = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().tgpu_partitioned_hash_builder_factory_create(ctx.handle, operator_id, nt, t, no, oc, nh, hc, precomputed_hash_channel,
                                                                           expected_positions, partition_count, C.byref(bridge), C.byref(h)))
        super().__init__(h)
        self.lookup_source_factory = LookupSourceFactory(bridge)


class LookupJoinOperatorFactory(OperatorFactory):
    """LookupJoinOperators.innerJoin / probeOuterJoin (M/operator/LookupJoinOperators.java:30-63)."""

    def __init__(self, ctx: Context, operator_id, lookup_source_factory: LookupSourceFactory, probe_types, probe_join_channels,
                 probe_hash_channel=-1, probe_output_channels=None, join_type=INNER):
        if probe_output_channels is None:
            probe_output_channels = list(range(len(probe_types)))
        t, nt = _i32(probe_types)
        jc, nj = _i32(probe_join_channels)
        oc, no = _i32(probe_output_channels)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_lookup_join_factory_create(ctx.handle, operator_id, lookup_source_factory.handle, nt, t, nj, jc,
                                                              probe_hash_channel, no, oc, join_type, C.byref(h)))
        super().__init__(h)
        self._bridge = lookup_source_factory


class LookupOuterOperatorFactory(OperatorFactory):
    """LookupOuterOperator.LookupOuterOperatorFactory (M/operator/LookupOuterOperator.java:35-110): the unmatched build rows of a
    LOOKUP_OUTER / FULL_OUTER join, after every probe operator has finished"""

    def __init__(self, ctx: Context, operator_id, lookup_source_factory: LookupSourceFactory, probe_output_types):
        t, nt = _i32(probe_output_types)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_lookup_outer_factory_create(ctx.handle, operator_id, lookup_source_factory.handle, nt, t, C.byref(h)))
        super().__init__(h)
        self._bridge = lookup_source_factory


class FilterProjectLookupJoinOperatorFactory(OperatorFactory):
    """FilterAndProjectOperator fused into the LookupJoin probe (one generated kernel); same results as the two reference
    operators back to back.  probe_join_channels / probe_output_channels index the page processor's projections."""

    def __init__(self, ctx: Context, operator_id, lookup_source_factory: LookupSourceFactory, input_types, filter_expr, projections,
                 probe_join_channels, probe_hash_channel=-1, probe_output_channels=None, join_type=INNER):
        if probe_output_channels is None:
            probe_output_channels = list(range(len(projections)))
        self.program = FlatProgram(filter_expr, projections)
        spec, keep = self.program.to_c()
        t, nt = _i32(input_types)
        jc, nj = _i32(probe_join_channels)
        oc, no = _i32(probe_output_channels)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_filter_project_lookup_join_factory_create(ctx.handle, operator_id, lookup_source_factory.handle, nt, t, C.byref(spec),
                                                                             nj, jc, probe_hash_channel, no, oc, join_type, C.byref(h)))
        super().__init__(h, keep)
        self._bridge = lookup_source_factory


def _agg_array(aggs):
    arr = (_lib.AggSpec * max(1, len(aggs)))()
    for i, a in enumerate(aggs):
        arr[i] = _lib.AggSpec(a[0], a[1], a[2] if len(a) > 2 else -1)
    return arr


class TopNOperatorFactory(OperatorFactory):
    """TopNOperator.createOperatorFactory (M/operator/TopNOperator.java:47-62): the n first rows in the order of the sort channels."""

    def __init__(self, ctx: Context, operator_id, types, n, sort_channels, sort_orders):
        t, nt = _i32(types)
        sc, ns = _i32(sort_channels)
        so, no = _i32(sort_orders)
        assert ns == no, "sort channels and sort orders differ in length"
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_top_n_factory_create(ctx.handle, operator_id, nt, t, int(n), ns, sc, so, C.byref(h)))
        super().__init__(h)


class OrderByOperatorFactory(OperatorFactory):
    """OrderByOperator.OrderByOperatorFactory (M/operator/OrderByOperator.java:48-131): all rows in the order of the sort channels."""

    def __init__(self, ctx: Context, operator_id, types, output_channels, expected_positions, sort_channels, sort_orders):
        t, nt = _i32(types)
        oc, no = _i32(output_channels)
        sc, ns = _i32(sort_channels)
        so, nso = _i32(sort_orders)
        assert ns == nso, "sort channels and sort orders differ in length"
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_order_by_factory_create(ctx.handle, operator_id, nt, t, no, oc, expected_positions, ns, sc, so, C.byref(h)))
        super().__init__(h)


class FilterProjectHashAggregationOperatorFactory(OperatorFactory):
    """FilterAndProjectOperator fused into HashAggregationOperator (HandTpchQuery1's pipeline shape): group_by_channels and the
    aggregates' channels index the page processor's projections; same results as the two reference operators back to back."""

    def __init__(self, ctx: Context, operator_id, input_types, filter_expr, projections, group_by_types, group_by_channels, aggs, step=SINGLE,
                 hash_channel=-1, expected_groups=10_000):
        self.program = FlatProgram(filter_expr, projections)
        spec, keep = self.program.to_c()
        t, nt = _i32(input_types)
        gt, ng = _i32(group_by_types)
        gc, _ = _i32(group_by_channels)
        arr = _agg_array(aggs)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_filter_project_hash_aggregation_factory_create(ctx.handle, operator_id, nt, t, C.byref(spec), ng, gt, gc, hash_channel, step,
                                                                                  len(aggs), arr, expected_groups, C.byref(h)))
        super().__init__(h, keep)


def precompile_fused_aggregation(input_types, filter_expr, projections, aggs, group_by_channels=()):
    prog = FlatProgram(filter_expr, projections)
    spec, keep = prog.to_c()
    t, nt = _i32(input_types)
    arr = _agg_array(aggs)
    gc, ng = _i32(group_by_channels)
    _lib.check(_lib.lib().tgpu_precompile_fused_aggregation(nt, t, C.byref(spec), len(aggs), arr, ng, gc))


def precompile_fused_probe(input_types, filter_expr, projections, join_channel, probe_output_channels):
    prog = FlatProgram(filter_expr, projections)
    spec, keep = prog.to_c()
    t, nt = _i32(input_types)
    oc, no = _i32(probe_output_channels)
    _lib.check(_lib.lib().tgpu_precompile_fused_probe(nt, t, C.byref(spec), join_channel, no, oc))


class GroupByHash:
    """GroupByHash.createGroupByHash (M/operator/GroupByHash.java:45-59) over the GPU table."""

    def __init__(self, ctx: Context, types, hash_channels, input_hash_channel=None, expected_size=100):
        t, n = _i32(types)
        hc, _ = _i32(hash_channels)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_group_by_hash_create(ctx.handle, n, t, hc, -1 if input_hash_channel is None else input_hash_channel,
                                                        expected_size, C.byref(h)))
        self.handle = h

    def addPage(self, page: Page):
        cp, keep = page.to_c()
        _lib.check(_lib.lib().tgpu_group_by_hash_add_page(self.handle, C.byref(cp)))

    def getGroupIds(self, page: Page):
        out = np.zeros(max(page.position_count, 1), dtype=np.int64)
        gc = C.c_int64()
        cp, keep = page.to_c()
        _lib.check(_lib.lib().tgpu_group_by_hash_get_group_ids(self.handle, C.byref(cp), out.ctypes.data, C.byref(gc)))
        return out[: page.position_count]

    def contains(self, position, page: Page):
        r = C.c_int32()
        cp, keep = page.to_c()
        _lib.check(_lib.lib().tgpu_group_by_hash_contains(self.handle, position, C.byref(cp), C.byref(r)))
        return bool(r.value)

    def getGroupCount(self):
        return _lib.lib().tgpu_group_by_hash_group_count(self.handle)

    def getCapacity(self):
        return _lib.lib().tgpu_group_by_hash_capacity(self.handle)

    def getRehashCount(self):
        return _lib.lib().tgpu_group_by_hash_rehash_count(self.handle)

    def getEstimatedSize(self):
        return _lib.lib().tgpu_group_by_hash_estimated_size(self.handle)

    def appendValues(self) -> Page:
        out = C.c_void_p()
        _lib.check(_lib.lib().tgpu_group_by_hash_append_values(self.handle, C.byref(out)))
        op = OutputPage(out)
        pg = op.to_host()
        op.release()
        return pg

    def appendValuesDevice(self) -> OutputPage:
        """appendValuesTo for every group, left on the device (the caller releases the page)"""
        out = C.c_void_p()
        _lib.check(_lib.lib().tgpu_group_by_hash_append_values(self.handle, C.byref(out)))
        return OutputPage(out)

    def close(self):
        if self.handle:
            _lib.lib().tgpu_group_by_hash_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # a constructor that failed before `handle` existed, or interpreter shutdown
            pass


def to_pages(operator: Operator, input_pages, to_host=True):
    """The reference's mini driver loop, T/operator/OperatorAssertion.java:82-137 (toPages)."""
    outputs = []
    it = iter(input_pages)
    pending = next(it, None)
    loops = 0
    while pending is not None and loops < 10_000_000:
        loops += 1
        if operator.isBlocked():
            raise RuntimeError("operator is blocked")
        if operator.needsInput():
            operator.addInput(pending)
            pending = next(it, None)
        out = operator.getOutput()
        if out is not None and out.position_count > 0:
            outputs.append(out)
    operator.finish()
    loops = 0
    while not operator.isFinished() and loops < 1_000_000:
        loops += 1
        out = operator.getOutput()
        if out is not None and out.position_count > 0:
            outputs.append(out)
    assert operator.isFinished() and not operator.needsInput() and not operator.isBlocked()
    if not to_host:
        return outputs
    host = [o.to_host() for o in outputs]
    for o in outputs:
        o.release()
    return host


DF_ALL, DF_VALUES, DF_RANGE, DF_NONE = 0, 1, 2, 3


class DynamicFilterSourceOperator(Operator):
    def domain(self, filter_channel):
        """after finish(): ("all",) | ("values", [..]) | ("range", lo, hi) | ("none",) -- DynamicFilterSourceOperator.finish() (:383-424)"""
        kind, out, lo, hi = C.c_int32(), C.c_void_p(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().tgpu_dynamic_filter_source_result(self.handle, filter_channel, C.byref(kind), C.byref(out), C.byref(lo), C.byref(hi)))
        if kind.value == DF_VALUES:
            page = OutputPage(out)
            vals = page.to_host().blocks[0].to_list() if page.position_count else []
            page.release()
            return ("values", vals)
        if kind.value == DF_RANGE and out.value:      # a VARCHAR channel: two rows, min then max
            page = OutputPage(out)
            vals = page.to_host().blocks[0].to_list()
            page.release()
            return ("range", vals[0], vals[1])
        return {DF_ALL: ("all",), DF_RANGE: ("range", int(lo.value), int(hi.value)), DF_NONE: ("none",)}[kind.value]


class DynamicFilterSourceOperatorFactory(OperatorFactory):
    """DynamicFilterSourceOperator.DynamicFilterSourceOperatorFactory (M/operator/DynamicFilterSourceOperator.java:74-143)"""

    def __init__(self, ctx: Context, operator_id, types, channels, max_distinct_values, max_filter_size_in_bytes, min_max_collection_limit):
        t, nt = _i32(types)
        ch, nc = _i32(channels)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_dynamic_filter_source_factory_create(ctx.handle, operator_id, nt, t, nc, ch, int(max_distinct_values), int(max_filter_size_in_bytes),
                                                                        int(min_max_collection_limit), C.byref(h)))
        super().__init__(h)

    def createOperator(self):
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_operator_factory_create_operator(self.handle, C.byref(h)))
        return DynamicFilterSourceOperator(h)


class MergePagesOperatorFactory(OperatorFactory):
    """MergePages.mergePages (M/operator/project/MergePages.java:64-96) as an operator: small pages are coalesced in HBM, big ones pass through"""

    def __init__(self, ctx: Context, operator_id, types, min_page_size_in_bytes, min_row_count, max_page_size_in_bytes=1024 * 1024):
        t, nt = _i32(types)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_merge_pages_factory_create(ctx.handle, operator_id, nt, t, int(min_page_size_in_bytes), int(min_row_count), int(max_page_size_in_bytes),
                                                              C.byref(h)))
        super().__init__(h)


class PartitionedOutputOperator(Operator):
    """the sink side of PartitionedOutputOperator (M/operator/PartitionedOutputOperator.java:46-300): poll() hands out what the
    reference enqueues into its OutputBuffer"""

    def poll(self):
        """next pending (partition, OutputPage) pair, or None"""
        part, out = C.c_int32(-1), C.c_void_p()
        _lib.check(_lib.lib().tgpu_partitioned_output_poll(self.handle, C.byref(part), C.byref(out)))
        return (int(part.value), OutputPage(out)) if out.value else None

    def info(self):
        rows, pages = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().tgpu_partitioned_output_info(self.handle, C.byref(rows), C.byref(pages)))
        return {"rowsAdded": int(rows.value), "pagesAdded": int(pages.value)}


class PartitionedOutputOperatorFactory(OperatorFactory):
    """PartitionedOutputOperator.PartitionedOutputFactory (M/operator/PartitionedOutputOperator.java:52-130): hash partitioning on
    `partition_channels` (or the precomputed `hash_channel`), `null_channel` / `replicates_any_row` replication (:411-418)."""

    def __init__(self, ctx: Context, operator_id, types, partition_channels, partition_count, hash_channel=-1, replicates_any_row=False, null_channel=-1, local=False):
        """local=True: the LocalExchange partition function (LocalPartitionGenerator.java:45-65; power-of-two partition count)"""
        t, nt = _i32(types)
        pc, npc = _i32(partition_channels)
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_partitioned_output_factory_create(ctx.handle, operator_id, nt, t, npc, pc, hash_channel, partition_count,
                                                                     1 if replicates_any_row else 0, null_channel, 1 if local else 0, C.byref(h)))
        super().__init__(h)

    def createOperator(self):
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_operator_factory_create_operator(self.handle, C.byref(h)))
        return PartitionedOutputOperator(h)
