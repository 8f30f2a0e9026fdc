"""The native exchange (csrc/exchange.hip behind tgpu_exchange_*) on the GPU:
  * over RCCL at world size 1 (the single-GPU box): partition kernels -> page header -> grouped ncclSend / ncclRecv -> device page;
  * a 2-rank rehearsal on ONE GPU (two processes, RCCL would refuse two ranks on one device): the same library code over the
    callback transport (GlooTransport, host staging) -- rows land on the rank that owns their hash partition
    ((rawHash & 0x7fff...) % world, M/operator/HashGenerator.java:24-35, checked with the oracle), nothing lost or duplicated, source
    order kept, VARCHAR / null vectors (present on one rank only) / empty contributions survive."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from gpu_common import rand_block

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")   # first: the library then reuses the RCCL PyTorch has loaded instead of adding a second copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def test_exchange_world1_rccl(pkg, ctx, oracle):
    ex_mod = importlib.import_module("presto-1_amd.exchange")
    ex = ex_mod.Exchange.over_rccl(ctx, 0, 1, lambda payload: payload)
    rng = np.random.default_rng(37)
    n = 50_000
    T = [pkg.BIGINT, pkg.VARCHAR, pkg.DOUBLE]
    page = pkg.Page(rand_block(pkg, rng, pkg.BIGINT, n, 0.02, (0, 10**6)), rand_block(pkg, rng, pkg.VARCHAR, n, 0.05, (0, 40)), rand_block(pkg, rng, pkg.DOUBLE, n, 0.1))
    out = ex.repartition(page, [0])
    assert out.to_host().rows() == page.rows()          # one partition: input order preserved
    out.release()
    # the PartitionedOutputOperator (shuffle producer) feeding the exchange inside the library
    pf = pkg.PartitionedOutputOperatorFactory(ctx, 0, T, [0], 1)
    pop = pf.createOperator()
    pop.addInput(page)
    out = ex.partitioned_output(pop, T)
    assert out.to_host().rows() == page.rows()
    out.release()
    pop.finish()
    pop.close()
    # broadcast of an operator's device output page, then into a join build
    keys = rng.permutation(200_000)[:60_000].astype(np.int64)
    f = pkg.field
    fp = pkg.FilterAndProjectOperatorFactory(ctx, 1, [pkg.BIGINT], None, [f(0, pkg.BIGINT)])
    op = fp.createOperator()
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, keys)))
    o = op.getOutput()
    rep = ex.all_gather(o.as_device_page())
    o.release()
    assert rep.position_count == len(keys)
    bf = pkg.HashBuilderOperatorFactory(ctx, 2, [pkg.BIGINT], [0], [0])
    b = bf.createOperator()
    b.addInput(rep)
    b.finish()
    rep.release()
    jf = pkg.LookupJoinOperatorFactory(ctx, 3, bf.lookup_source_factory, [pkg.BIGINT], [0], probe_output_channels=[0])
    probe = rng.integers(0, 200_000, 100_000).astype(np.int64)
    joined = pkg.to_pages(jf.createOperator(), [pkg.Page(pkg.Block(pkg.BIGINT, probe))])
    want_p, want_b = oracle.PagesHash([oracle.Col(pkg.BIGINT, keys)]).probe([oracle.Col(pkg.BIGINT, probe)])
    assert [r for pg in joined for r in pg.rows()] == [(int(probe[i]), int(keys[j])) for i, j in zip(want_p, want_b)]
    assert ex.bytes_sent == 0                           # nothing leaves a single rank
    # an empty page takes part in the collectives like any other
    empty = pkg.Page(pkg.Block(pkg.BIGINT, np.zeros(0, dtype=np.int64)), pkg.Block(pkg.VARCHAR, []), pkg.Block(pkg.DOUBLE, np.zeros(0)))
    out = ex.repartition(empty, [0])
    assert out.position_count == 0
    out.release()
    ex.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(rank, n):
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    keys = rng.integers(0, 5000, n).astype(np.int64)
    pay = np.arange(n, dtype=np.int64) + rank * 1_000_000
    strs = [None if k % 11 == 0 else "s%d" % (k % 37) * (1 + k % 3) for k in keys]
    dbl = rng.standard_normal(n)
    dnull = (rng.random(n) < 0.1).astype(np.uint8) if rank == 0 else None     # a null vector on one rank only
    return keys, pay, strs, dbl, dnull


def _rehearsal_worker(rank, world, port, n, out_q):
    try:
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pkg = importlib.import_module("presto-1_amd")
        ex_mod = importlib.import_module("presto-1_amd.exchange")
        ctx = pkg.Context(0)
        dev = torch.device("cuda", 0)
        ex = ex_mod.Exchange.over_transport(ctx, rank, world, ex_mod.GlooTransport(dist, ex_mod.TorchDeviceMemory(dev)))
        keys, pay, strs, dbl, dnull = _rows(rank, n if rank == 0 else n // 3)
        page = pkg.Page(pkg.Block(pkg.BIGINT, keys), pkg.Block(pkg.BIGINT, pay), pkg.Block(pkg.VARCHAR, strs), pkg.Block(pkg.DOUBLE, dbl, dnull))
        got = ex.repartition(page, [0]).to_host()
        rep = ex.all_gather(pkg.Page(pkg.Block(pkg.BIGINT, pay[: 5 + 3 * rank]), pkg.Block(pkg.VARCHAR, strs[: 5 + 3 * rank]))).to_host()
        empty = pkg.Page(pkg.Block(pkg.BIGINT, np.zeros(0, dtype=np.int64)), pkg.Block(pkg.BIGINT, np.zeros(0, dtype=np.int64)), pkg.Block(pkg.VARCHAR, []),
                         pkg.Block(pkg.DOUBLE, np.zeros(0)))
        lonely = ex.repartition(page if rank == 1 else empty, [0]).to_host()       # one rank contributes nothing
        # one way: every row of both ranks belongs to rank 1, and only rank 0's page carries a null vector -- rank 0 sends a null vector
        # and receives nothing at all (the send side of a channel's null transfer must not depend on what the rank receives)
        from oracle import oracle
        cand = np.arange(4000, dtype=np.int64)
        to1 = cand[oracle.partition_remote(oracle.hash_rows([oracle.Col(oracle.BIGINT, cand)]), world) == 1]
        k1 = to1[(np.arange(400) * (3 + rank)) % len(to1)]
        d1 = np.arange(400, dtype=np.float64) + 1000.0 * rank
        n1 = (np.arange(400) % 7 == 0).astype(np.uint8) if rank == 0 else None
        one_way = ex.repartition(pkg.Page(pkg.Block(pkg.BIGINT, k1), pkg.Block(pkg.DOUBLE, d1, n1)), [0]).to_host()
        out_q.put((rank, got.rows(), rep.rows(), lonely.rows(), ex.bytes_sent, None, one_way.rows()))
        ex.close()
        ctx.close()
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:   # surfaced by the parent
        import traceback
        out_q.put((rank, None, None, None, 0, traceback.format_exc(), None))


def test_exchange_two_ranks_on_one_gpu_over_the_callback_transport(pkg, oracle):
    import torch.multiprocessing as mp
    world, n = 2, 3000
    port = _free_port()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_rehearsal_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    for r in results:
        assert r[5] is None, r[5]
    sent = {}
    for r in range(world):
        keys, pay, strs, dbl, dnull = _rows(r, n if r == 0 else n // 3)
        sent[r] = [(int(k), int(p), s, None if (dnull is not None and dnull[i]) else float(d)) for i, (k, p, s, d) in enumerate(zip(keys, pay, strs, dbl))]

    def owner(key):
        return int(oracle.partition_remote(oracle.hash_rows([oracle.Col(oracle.BIGINT, np.array([key], dtype=np.int64))]), world)[0])

    cand = np.arange(4000, dtype=np.int64)
    to1 = cand[oracle.partition_remote(oracle.hash_rows([oracle.Col(oracle.BIGINT, cand)]), world) == 1]
    one_way_sent = []
    for r in range(world):
        k1 = to1[(np.arange(400) * (3 + r)) % len(to1)]
        one_way_sent += [(int(k), None if (r == 0 and i % 7 == 0) else float(i + 1000.0 * r)) for i, k in enumerate(k1)]
    for rank, got, rep, lonely, nbytes, _, one_way in results:
        assert one_way == (one_way_sent if rank == 1 else [])
        want = [row for src in range(world) for row in sent[src] if owner(row[0]) == rank]     # grouped by source rank, source order kept
        assert got == want
        assert nbytes > 0
        want_rep = [(row[1], row[2]) for src in range(world) for row in sent[src][: 5 + 3 * src]]
        assert rep == want_rep
        assert lonely == [row for row in sent[1] if owner(row[0]) == rank]


def _agg_rows(rank, n):
    rng = np.random.Generator(np.random.PCG64(500 + rank))
    keys = [None if k == 7 else "g%d" % k for k in rng.integers(0, 9, n)]
    vals = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, n)
    ints = rng.integers(-10**9, 10**9, n).astype(np.int64)
    return keys, vals, ints


def _agg_worker(rank, world, port, n, out_q):
    try:
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pkg = importlib.import_module("presto-1_amd")
        ex_mod = importlib.import_module("presto-1_amd.exchange")
        ctx = pkg.Context(0)
        ex = ex_mod.Exchange.over_transport(ctx, rank, world, ex_mod.GlooTransport(dist, ex_mod.TorchDeviceMemory(torch.device("cuda", 0))))
        keys, vals, ints = _agg_rows(rank, n)
        page = pkg.Page(pkg.Block(pkg.VARCHAR, keys), pkg.Block(pkg.DOUBLE, vals), pkg.Block(pkg.BIGINT, ints))
        aggs = [(pkg.SUM_DOUBLE, 1), (pkg.COUNT_ALL, -1), (pkg.SUM_BIGINT, 2), (pkg.AVG_DOUBLE, 1)]
        part = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.VARCHAR], [0], aggs, step=pkg.PARTIAL).createOperator()
        part.addInput(page)
        part.finish()
        fin = pkg.HashAggregationOperatorFactory(ctx, 1, [pkg.VARCHAR], [0], [(pkg.SUM_DOUBLE, 1), (pkg.COUNT_ALL, 3), (pkg.SUM_BIGINT, 4), (pkg.AVG_DOUBLE, 6)], step=pkg.FINAL).createOperator()
        while not part.isFinished():
            o = part.getOutput()
            if o is not None:
                g = ex.all_gather(o.as_device_page())     # every rank's partial page, in rank order
                fin.addInput(g)
                g.release()
                o.release()
        fin.finish()
        rows = []
        while not fin.isFinished():
            o = fin.getOutput()
            if o is not None:
                rows += o.to_host().rows()
        out_q.put((rank, rows, None))
        ex.close()
        ctx.close()
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        out_q.put((rank, None, traceback.format_exc()))


def test_partial_all_gather_final_aggregation_on_two_ranks(pkg, oracle):
    """SURVEY.md 8e step 3 (what bench.py --gpus N runs for Q1): PARTIAL aggregation per rank -> all-gather of the partial pages ->
    FINAL combine on every rank.  Counts and bigint sums exact; a double sum = the exact sum of the ranks' exactly rounded partial sums
    (restated with the oracle); every rank ends with the same rows, groups in first-seen order of the gathered pages."""
    import torch.multiprocessing as mp
    from gpu_common import ulp_diff
    world, n = 2, 20_000
    port = _free_port()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_agg_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    for r in results:
        assert r[2] is None, r[2]
    assert results[0][1] == results[1][1]
    rows = results[0][1]
    per_rank = []
    for r in range(world):
        keys, vals, ints = _agg_rows(r, n)
        d = {}
        for k, v, i in zip(keys, vals, ints):
            e = d.setdefault(k, [[], 0, 0])
            e[0].append(v)
            e[1] += 1
            e[2] += int(i)
        per_rank.append(d)
    order = []
    for d in per_rank:          # first-seen order of the gathered partial pages: rank 0's groups in ITS first-seen order, then new ones of rank 1
        for k in d:
            if k not in order:
                order.append(k)
    assert [r[0] for r in rows] == order
    for row in rows:
        k = row[0]
        addends = []
        for d in per_rank:
            if k in d:
                _, s = oracle.agg_double_sum_exact(np.zeros(len(d[k][0]), dtype=np.int64), np.array(d[k][0]), 1)
                addends.append(float(s[0]))
        _, total = oracle.agg_double_sum_exact(np.zeros(len(addends), dtype=np.int64), np.array(addends), 1)
        cnt = sum(d[k][1] for d in per_rank if k in d)
        assert row[2] == cnt and row[3] == sum(d[k][2] for d in per_rank if k in d)
        assert ulp_diff(np.array([row[1]]), np.array([float(total[0])])).max() == 0
        assert ulp_diff(np.array([row[4]]), np.array([float(total[0]) / cnt])).max() == 0


# ---- Q3 on two ranks: repartition -> build -> probe -> aggregate against the single-instance oracle (VERDICT r2 "next" 4) ----------------
def _q3_worker(rank, world, port, sf, out_q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), TGPU_BENCH_BACKEND="gloo", TGPU_BENCH_DEVICE="0",
                          LOCAL_RANK="0")
        import argparse
        import bench as bench_mod
        b = bench_mod.Bench(argparse.Namespace())
        b.setup_q3(sf)
        tables = {k: v.cpu().numpy() for k, v in b.q3.items()}
        result = {}
        for plan, step in (("co_partitioned", b.step_q3_dist), ("repartition", b.step_q3_dist_repartition)):
            b.capture = {}
            step()
            cap, b.capture = b.capture, None
            rows = [r for o in b.q3_result for r in o.to_host().rows()]
            check = b.check_q3_dist() if plan == "co_partitioned" else b.check_q3_dist_repartition()
            for o in b.q3_result:
                o.release()
            b.q3_result = None
            joins = {name: [r for pg in pages for r in pg.rows()] for name, pages in cap.items()}
            result[plan] = {"rows": rows, "joins": joins, "stats": dict(b.q3_stats), "check_ok": check["ok"]}
        out_q.put((rank, tables, result, None))
        b.exchange.close()
        b.ctx.close()
        b.dist.barrier()
        b.dist.destroy_process_group()
    except Exception:
        import traceback
        out_q.put((rank, None, None, traceback.format_exc()))


def test_q3_join_chain_on_two_ranks_equals_the_single_instance_oracle(pkg, oracle):
    """bench.py's two distributed Q3 plans (what `bench.py --gpus N` times) on two processes over the callback transport, SF 0.2 per rank:
    the co-partitioned plan (replicated customer build via all-gather, fused probes, single-step aggregation per rank) and the repartition
    plan (every join input through tgpu_exchange_repartition, fused probes behind the exchange).  The union of the ranks' join outputs and of
    their final (orderkey, orderdate, shippriority) -> sum(revenue) rows equals the single-instance oracle over the union of the ranks'
    inputs: rows as multisets per join, groups as a map, sums bit for bit (a group's rows all sit on one rank, in their input order)."""
    import torch.multiprocessing as mp
    from gpu_common import ulp_diff
    world, sf = 2, 0.2
    port = _free_port()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_q3_worker, args=(r, world, port, sf, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    for r in results:
        assert r[3] is None, r[3]
    h = {k: np.concatenate([results[r][1][k] for r in range(world)]) for k in results[0][1] if k not in ("c_seg_off",)}
    # the varchar column's offsets restart per rank: rebase
    offs, base = [], 0
    for r in range(world):
        o = results[r][1]["c_seg_off"]
        offs.append(o[:-1] + base)
        base += int(o[-1])
    seg_off = np.concatenate(offs + [np.array([base], dtype=offs[0].dtype)])
    seg_first = h["c_seg_bytes"][seg_off[:-1]]
    seg_len = np.diff(seg_off)
    ck = h["c_custkey"][(seg_first == ord("B")) & (seg_len == 8)]
    cust = oracle.PagesHash([oracle.Col(oracle.BIGINT, ck)])
    om = np.nonzero(h["o_orderdate"] < 9204)[0]
    op, _ = cust.probe([oracle.Col(oracle.BIGINT, h["o_custkey"][om])])
    o_rows = om[op]
    okeys, odate, oprio = h["o_orderkey"][o_rows], h["o_orderdate"][o_rows], h["o_shippriority"][o_rows]
    want_orders_join = sorted(zip(okeys.tolist(), odate.tolist(), oprio.tolist()))
    orders = oracle.PagesHash([oracle.Col(oracle.BIGINT, okeys)])
    lm = np.nonzero(h["l_shipdate"] > 9204)[0]
    rev = h["l_extendedprice"][lm] * (1.0 - h["l_discount"][lm])
    lp, lb = orders.probe([oracle.Col(oracle.BIGINT, h["l_orderkey"][lm])])
    want_lineitem_join = sorted(zip(h["l_orderkey"][lm][lp].tolist(), rev[lp].view(np.int64).tolist(), odate[lb].tolist(), oprio[lb].tolist()))
    kcols = [oracle.Col(oracle.BIGINT, h["l_orderkey"][lm][lp]), oracle.Col(oracle.DATE, odate[lb]), oracle.Col(oracle.INTEGER, oprio[lb])]
    gbh = oracle.MultiChannelGroupByHash([oracle.BIGINT, oracle.DATE, oracle.INTEGER], 1 << 16)
    gids = gbh.get_group_ids(kcols, oracle.hash_rows(kcols))
    first_rows, _ = gbh.group_rows()
    _, sums = oracle.agg_double_sum(gids, rev[lp], gbh.group_count)
    want_groups = {(int(h["l_orderkey"][lm][lp][fr]), int(odate[lb][fr]), int(oprio[lb][fr])): float(s) for fr, s in zip(first_rows, sums)}
    assert len(want_groups) > 3_000
    for plan in ("co_partitioned", "repartition"):
        per_rank = [results[r][2][plan] for r in range(world)]
        assert all(x["check_ok"] for x in per_rank), plan
        got_oj = sorted(tuple(row) for x in per_rank for row in x["joins"].get("orders_join", []))
        assert got_oj == want_orders_join, plan
        got_lj = sorted((row[0], int(np.float64(row[1]).view(np.int64)), row[2], row[3]) for x in per_rank for row in x["joins"].get("lineitem_join", []))
        assert got_lj == want_lineitem_join, plan
        got_groups = {}
        for x in per_rank:
            for k, d, pr, s in x["rows"]:
                assert (k, d, pr) not in got_groups, "a group came out of two ranks"
                got_groups[(k, d, pr)] = s
        assert got_groups.keys() == want_groups.keys(), plan
        keys = sorted(want_groups)
        assert ulp_diff(np.array([got_groups[k] for k in keys]), np.array([want_groups[k] for k in keys])).max() == 0, plan
        assert all(x["stats"]["exchange_bytes_sent"] > 0 for x in per_rank), plan
