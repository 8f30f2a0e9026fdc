"""CPU tests of the page-level restatements in oracle/oracle.py that the GPU parity tests of the widened operators lean on
(PagePartitioner, MergePages, DynamicFilterSource): hand-checked cases, including the reference's own TestMergePages scenarios."""
import numpy as np

from seqpages import sequence_page


def _cols(oracle, types, n, start=0):
    return [oracle.Col(t, v) for t, v in zip(types, sequence_page(types, n, *([start] * len(types))))]


def test_page_partitioner_replication_rules(oracle):
    O = oracle
    keys = O.Col(O.BIGINT, [5, 6, 7, 8], nulls=[0, 1, 0, 0])
    raw = O.hash_rows([keys])
    pid = O.partition_remote(raw, 3)
    # null channel: the row with a null key goes to every partition (PartitionedOutputOperator.java:411-418)
    p = O.PagePartitioner(3, False, 0)
    out = p.partition_page([keys], raw)
    for part in range(3):
        assert out[part] == sorted([1] + [i for i in (0, 2, 3) if pid[i] == part])
    # replicate-any-row: only the first row ever seen is replicated, then never again
    p = O.PagePartitioner(3, True, -1)
    first = p.partition_page([keys], raw)
    assert all(0 in first[part] for part in range(3))
    second = p.partition_page([keys], raw)
    assert sorted(sum(second, [])) == [0, 1, 2, 3]
    # the LocalExchange function masks with count - 1
    p = O.PagePartitioner(4, False, -1, local=True)
    out = p.partition_page([keys], raw)
    loc = O.partition_local(raw, 4)
    assert [sorted(x) for x in out] == [[i for i in range(4) if loc[i] == part] for part in range(4)]


def test_merge_pages_reference_scenarios(oracle):
    O = oracle
    types = [O.BIGINT, O.INTEGER, O.DOUBLE]   # TestMergePages.java:42 (REAL has INTEGER's layout)
    page = _cols(O, types, 10)
    size = O.page_size_in_bytes(page)
    assert size == 10 * (9 + 5 + 9)
    big = 2**31 - 1
    # testMinPageSizeThreshold / testMinRowCountThreshold: pass through
    m = O.MergePages(size, big, big)
    assert m.add(page) == [[(0, r) for r in range(10)]] and m.finish() == []
    m = O.MergePages(1024 * 1024, 10, big)
    assert m.add(page) == [[(0, r) for r in range(10)]]
    # testBufferSmallPages: two halves buffered, one page at the end
    whole = O.page_size_in_bytes(_cols(O, types, 20))
    m = O.MergePages(whole + 1, 21, big)
    assert m.add(_cols(O, types, 10)) == [] and m.add(_cols(O, types, 10, 10)) == []
    assert m.finish() == [[(0, r) for r in range(10)] + [(1, r) for r in range(10)]]
    # testFlushOnBigPage: buffered small page first, then the big one
    m = O.MergePages(O.page_size_in_bytes(_cols(O, types, 100)), 100, big)
    assert m.add(_cols(O, types, 10)) == []
    assert m.add(_cols(O, types, 100)) == [[(0, r) for r in range(10)], [(1, r) for r in range(100)]]
    # testFlushOnFullPage: the buffer is flushed when it reaches the max page size
    t1 = [O.BIGINT]
    whole = O.page_size_in_bytes(_cols(O, t1, 20))
    m = O.MergePages(whole // 2 + 1, 11, whole)
    outs = []
    for i in range(4):
        outs += m.add(_cols(O, t1, 10, 10 * (i % 2)))
    assert [len(o) for o in outs] == [20, 20] and m.finish() == []


def test_dynamic_filter_source_state_machine(oracle):
    O = oracle
    types = [O.BIGINT, O.VARCHAR]
    page = [O.Col(O.BIGINT, [3, 1, 0, 3, 9], nulls=[0, 0, 1, 0, 0]), O.Col(O.VARCHAR, ["b", None, "a", "b", "c"])]
    # everything fits: distinct non-null values in first-seen order
    d = O.DynamicFilterSource(types, [0, 1], 100, 1 << 20, 1000)
    d.add(page)
    assert d.domain(0) == ("values", [3, 1, 9]) and d.domain(1) == ("values", ["b", "a", "c"])
    # too many distinct values: min / max for the BIGINT channel, nothing for VARCHAR
    d = O.DynamicFilterSource(types, [0, 1], 2, 1 << 20, 1000)
    d.add(page)
    d.add([O.Col(O.BIGINT, [-4, 20]), O.Col(O.VARCHAR, ["x", "y"])])
    assert d.domain(0) == ("range", -4, 20) and d.domain(1) == ("range", "a", "y")
    # ... but only while the row limit holds (DynamicFilterSourceOperator.java:283-289)
    d = O.DynamicFilterSource(types, [0, 1], 2, 1 << 20, 6)
    d.add(page)
    d.add([O.Col(O.BIGINT, [-4, 20]), O.Col(O.VARCHAR, ["x", "y"])])
    assert d.domain(0) == ("all",)
    # an orderable channel that only saw nulls
    d = O.DynamicFilterSource([O.BIGINT, O.BIGINT], [0, 1], 1, 1 << 20, 100)
    d.add([O.Col(O.BIGINT, [0, 0], nulls=[1, 1]), O.Col(O.BIGINT, [1, 2])])
    assert d.domain(0) == ("none",) and d.domain(1) == ("range", 1, 2)
