#!/usr/bin/env python3
"""bench.py -- headline benchmark of the operator hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload at N=1 (BASELINE.json configs[3], the one the metric is quoted on): the TPCH-SF100 Q3 operator pipeline on
synthetic, seeded, device-resident TPCH-shaped columns --
    customer : FilterAndProject(mktsegment = 'BUILDING')                      -> HashBuilder(custkey)
    orders   : FilterAndProject(orderdate < 1995-03-15) -> LookupJoin(custkey) -> HashBuilder(orderkey)
    lineitem : FilterAndProject(shipdate > 1995-03-15, ep*(1-disc))           -> LookupJoin(orderkey)
               -> HashAggregation(orderkey, orderdate, shippriority; sum(revenue))
A "step" is one pass of that whole pipeline (both builds + both probes + filters + the final aggregation).
`value` = lineitem rows entering the lineitem join probe / step time (probe rows/s of the whole job, inputs resident in HBM).
The same JSON line carries `q1` (BASELINE configs[2]: SF100 Q1 filter/project + hash aggregation, input rows/s) and `cfg2`
(configs[1]: 100 M-row BIGINT filter+project), a `roofline` object for the dominant kernel (HIP-event timed, algorithmic bytes
per DESIGN.md) and a `cpu_baseline` object (the C oracle = a row-at-a-time port of the Java operators, 1 core, bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
SEED = 42
M64 = (1 << 64) - 1


def s64(x):
    x &= M64
    return x - (1 << 64) if x >> 63 else x


def lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64(x):
    """counter-based generator: identical on CPU and GPU, int64 wrap-around arithmetic"""
    z = x + s64(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def rnd(col, idx, mod):
    """uniform integer in [0, mod) for (column id, row index tensor)"""
    z = splitmix64(idx ^ s64(SEED ^ ((col * 0x9E3779B97F4A7C15) & M64)))
    return lsr(z, 1) % mod


D_1995_03_15 = 9204
SEGMENTS = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"HOUSEHOLD", b"MACHINERY"]


def varchar_from_ids(ids, table, dev):
    lens = torch.tensor([len(s) for s in table], dtype=torch.int64, device=dev)
    width = max(len(s) for s in table)
    tab = torch.zeros((len(table), width), dtype=torch.uint8, device=dev)
    for i, s in enumerate(table):
        tab[i, : len(s)] = torch.tensor(list(s), dtype=torch.uint8)
    rl = lens[ids]
    offsets = torch.zeros(ids.numel() + 1, dtype=torch.int32, device=dev)
    offsets[1:] = torch.cumsum(rl, 0).to(torch.int32)
    row = torch.repeat_interleave(torch.arange(ids.numel(), device=dev), rl)
    pos = torch.arange(row.numel(), device=dev) - offsets[:-1].to(torch.int64)[row]
    data = tab[ids[row], pos].contiguous()
    return data, offsets


def gen_q3(dev, sf, rank=0, world=1):
    """rank's input split of a TPCH SF(sf * world) database: its slice of customer and orders (global keys) and the lineitems of
    those orders.  world == 1 is the plain SF(sf) database."""
    n_c = int(150_000 * sf)
    n_o = int(1_500_000 * sf)
    base = rank * n_o * 8      # distinct counter ranges per rank
    t = {}
    ci = torch.arange(n_c, device=dev, dtype=torch.int64)
    t["c_custkey"] = rank * n_c + ci + 1
    t["c_seg_bytes"], t["c_seg_off"] = varchar_from_ids(rnd(1, ci + base, 5), SEGMENTS, dev)
    oi = torch.arange(n_o, device=dev, dtype=torch.int64)
    og = rank * n_o + oi
    t["o_orderkey"] = (og // 8) * 32 + (og % 8) + 1
    t["o_custkey"] = rnd(2, oi + base, n_c * world) + 1
    t["o_orderdate"] = (8035 + rnd(3, oi + base, 2406)).to(torch.int32)
    t["o_shippriority"] = torch.zeros(n_o, dtype=torch.int32, device=dev)
    cnt = 1 + rnd(4, oi + base, 7)
    lo = torch.repeat_interleave(oi, cnt)
    li = torch.arange(lo.numel(), device=dev, dtype=torch.int64)
    t["l_orderkey"] = t["o_orderkey"][lo]
    t["l_shipdate"] = (t["o_orderdate"][lo].to(torch.int64) + 1 + rnd(5, li + base, 121)).to(torch.int32)
    qty = (1 + rnd(6, li + base, 50)).to(torch.float64)
    t["l_extendedprice"] = qty * ((90000 + rnd(7, li + base, 120001)).to(torch.float64) / 100.0)
    t["l_discount"] = rnd(8, li + base, 11).to(torch.float64) / 100.0
    del lo, li, qty, cnt
    return t


def gen_q1(dev, n, rank=0):
    base = rank * 1_000_003
    i = torch.arange(n, device=dev, dtype=torch.int64)
    r = rnd(11, i + base, 10000)
    rf = torch.where(r < 2460, 65, torch.where(r < 7535, 78, 82)).to(torch.uint8)      # A / N / R
    ls = torch.where((r >= 2525) & (r < 7535), 79, 70).to(torch.uint8)                 # O / F
    t = {"returnflag": rf, "linestatus": ls, "off": torch.arange(n + 1, device=dev, dtype=torch.int32)}
    qty = (1 + rnd(12, i + base, 50)).to(torch.float64)
    t["quantity"] = qty
    t["extendedprice"] = qty * ((90000 + rnd(13, i + base, 120001)).to(torch.float64) / 100.0)
    t["discount"] = rnd(14, i + base, 11).to(torch.float64) / 100.0
    t["tax"] = rnd(15, i + base, 9).to(torch.float64) / 100.0
    t["shipdate"] = (8036 + rnd(16, i + base, 2526)).to(torch.int32)
    return t


class Bench:
    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("TGPU_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # TGPU_BENCH_DEVICE: rehearsal on one GPU
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.backend = os.environ.get("TGPU_BENCH_BACKEND", "nccl")   # "gloo": rehearsal of several ranks on one GPU
        if self.world > 1 or os.environ.get("TGPU_BENCH_FORCE_DIST"):
            import torch.distributed as dist
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
            self.dist = dist
        else:
            self.dist = None
        self.coll_dev = self.dev if self.backend == "nccl" else torch.device("cpu")
        self.pkg = importlib.import_module("presto-1_amd")
        self.entry = importlib.import_module("__graft_entry__")
        self.ctx = self.pkg.Context(self.local_rank, stream=torch.cuda.current_stream().cuda_stream)
        # the bench's device-resident tables outlive every operator: the promise of tgpu_context_set_device_input_stable holds (it lets the fused
        # aggregation run one launch per page without waiting for a page's counters; `device_input_stable` in the line's config says so)
        self.ctx.set_device_input_stable(True)
        self.capture = None
        self.exchange = None

    # -- helpers ------------------------------------------------------------------------------------------------------
    def dblock(self, type_id, values, offsets=None):
        n = values.numel() if offsets is None else offsets.numel() - 1
        return self.pkg.DeviceBlock(type_id, n, values, None, offsets)

    def barrier_sync(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, steps, warmup, profile_apart=False):
        """(seconds per step, per-kernel profile); the profile's pseudo entry "__readbacks" (host <- device round trips of the timed
        region) is moved to self.last_readbacks_per_step.  profile_apart (page-at-a-time runs: thousands of launches per step, and the
        library's per-kernel profile costs two event records per launch): the K steps are timed with the profile switched off, the
        per-kernel breakdown comes from one more, untimed, step"""
        for _ in range(warmup):
            fn()
        if profile_apart:
            self.ctx.profile_enable(False)
            self.barrier_sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            self.barrier_sync()
            dt = time.perf_counter() - t0
            self.ctx.profile_enable(True)
            self.ctx.profile_reset()
            fn()
            self.barrier_sync()
            prof = self.ctx.profile()
            self.last_readbacks_per_step = prof.pop("__readbacks", {"count": 0})["count"]
            for v in prof.values():   # callers divide by `steps`
                v["total_ms"] *= steps
                v["count"] *= steps
            return dt / steps, prof
        self.ctx.profile_reset()
        self.barrier_sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.barrier_sync()
        dt = time.perf_counter() - t0
        if self.dist is not None:
            tt = torch.tensor([dt], device=self.coll_dev, dtype=torch.float64)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        prof = self.ctx.profile()
        self.last_readbacks_per_step = prof.pop("__readbacks", {"count": 0})["count"] / steps
        return dt / steps, prof

    def step_stats(self, fn, steps):
        """median / min of individually timed steps (a device synchronisation around each: SURVEY.md 8d "median + min"); the contract's
        ms_per_step stays the mean over the back-to-back steps of timed()"""
        ts = []
        for _ in range(max(steps, 1)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return {"median_ms": ts[len(ts) // 2], "min_ms": ts[0], "steps": len(ts)}

    def drive(self, op, page):
        """the operator's only page through it (Driver.processInternal for a single hop: addInput, finish, getOutput until isFinished -- an
        operator may hold its output back until then, the fused join does for small pages); returns device output pages"""
        outs = []
        assert op.needsInput()
        op.addInput(page)
        op.finish()
        while not op.isFinished():
            o = op.getOutput()
            if o is not None:
                outs.append(o)
            elif op.isBlocked():
                break
        return outs

    def finish(self, op):
        outs = []
        op.finish()
        while not op.isFinished():
            o = op.getOutput()
            if o is not None:
                outs.append(o)
            elif op.isBlocked():
                break
        return outs

    # -- Q3 -----------------------------------------------------------------------------------------------------------
    def setup_q3(self, sf):
        p = self.pkg
        self.q3 = gen_q3(self.dev, sf, self.rank, self.world)
        self.ensure_exchange()
        pp = self.entry.bench_page_processors(p)
        B, D, DT, V, I = p.BIGINT, p.DOUBLE, p.DATE, p.VARCHAR, p.INTEGER
        self.q3_fac = {
            "cust_fp": p.FilterAndProjectOperatorFactory(self.ctx, 0, *pp["q3_customer"]),
            "ord_fp": p.FilterAndProjectOperatorFactory(self.ctx, 1, *pp["q3_orders"]),
            "li_fp": p.FilterAndProjectOperatorFactory(self.ctx, 2, *pp["q3_lineitem"]),
        }
        t = self.q3
        self.q3_pages = {
            "customer": p.Page(self.dblock(B, t["c_custkey"]), self.dblock(V, t["c_seg_bytes"], t["c_seg_off"])),
            "orders": p.Page(self.dblock(B, t["o_orderkey"]), self.dblock(B, t["o_custkey"]), self.dblock(DT, t["o_orderdate"]), self.dblock(I, t["o_shippriority"])),
            "lineitem": p.Page(self.dblock(B, t["l_orderkey"]), self.dblock(D, t["l_extendedprice"]), self.dblock(D, t["l_discount"]), self.dblock(DT, t["l_shipdate"])),
        }
        self.q3_stats = {}

    def ensure_exchange(self):
        if self.dist is not None and getattr(self, "exchange", None) is None:
            # the exchange is the native library's (tgpu_exchange_*: RCCL send / recv groups over xGMI); torch.distributed only hands
            # rank 0's RCCL unique id to the other ranks.  Rehearsals of several ranks on one GPU (TGPU_BENCH_BACKEND=gloo) run the same
            # library code over its callback transport with host staging.
            ex_mod = importlib.import_module("presto-1_amd.exchange")
            if self.backend == "nccl":
                def broadcast(payload):
                    t = torch.zeros(ex_mod.ID_BYTES, dtype=torch.uint8, device=self.dev)
                    if payload is not None:
                        t.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
                    self.dist.broadcast(t, src=0)
                    return bytes(t.cpu().numpy().tobytes())
                self.exchange = ex_mod.Exchange.over_rccl(self.ctx, self.rank, self.world, broadcast)
            else:
                self.exchange = ex_mod.Exchange.over_transport(self.ctx, self.rank, self.world, ex_mod.GlooTransport(self.dist, ex_mod.TorchDeviceMemory(self.dev)))

    def step_q3(self):
        p, ctx, f, pages = self.pkg, self.ctx, self.q3_fac, self.q3_pages
        B, D, DT, I = p.BIGINT, p.DOUBLE, p.DATE, p.INTEGER
        st = self.q3_stats
        # customer: filter -> build
        cb = p.HashBuilderOperatorFactory(ctx, 10, [B], [], [0])
        cbuild = cb.createOperator()
        op = f["cust_fp"].createOperator()
        for o in self.drive(op, pages["customer"]):
            st["customer_build_rows"] = o.position_count
            self.captured("customer_filter", o)
            cbuild.addInput(o.as_device_page())
            o.release()
        cbuild.finish()
        # orders: filter/project fused into the probe of the customer table -> build on orderkey (output orderdate, shippriority)
        pp = self.entry.bench_page_processors(p)
        oj = p.FilterProjectLookupJoinOperatorFactory(ctx, 11, cb.lookup_source_factory, *pp["q3_orders"], [1], probe_output_channels=[0, 2, 3])
        ob = p.HashBuilderOperatorFactory(ctx, 12, [B, DT, I], [1, 2], [0])
        obuild = ob.createOperator()
        ojoin = oj.createOperator()
        for j in self.drive(ojoin, pages["orders"]):
            st["orders_build_rows"] = j.position_count
            self.captured("orders_join", j)
            obuild.addInput(j.as_device_page())
            j.release()
        obuild.finish()
        ojoin.close()
        # lineitem: filter/project fused into the probe of the orders table -> aggregate
        lj = p.FilterProjectLookupJoinOperatorFactory(ctx, 13, ob.lookup_source_factory, *pp["q3_lineitem"], [0], probe_output_channels=[0, 1])
        agg = p.HashAggregationOperatorFactory(ctx, 14, [B, DT, I], [0, 2, 3], [(p.SUM_DOUBLE, 1)], expected_groups=1 << 20)
        ljoin = lj.createOperator()
        aop = agg.createOperator()
        for j in self.drive(ljoin, pages["lineitem"]):
            st["lineitem_join_rows"] = j.position_count
            self.captured("lineitem_join", j)
            aop.addInput(j.as_device_page())
            j.release()
        outs = self.finish(aop)
        st["groups"] = sum(o.position_count for o in outs)
        self.q3_result = outs
        ljoin.close()
        cbuild.close()
        obuild.close()
        aop.close()

    def q3_paged(self, steps, warmup, page_rows, merge_mb=None):
        """Q3 with every table fed as pages of `page_rows` rows (the engine hands pages, not tables): each probe page's join output goes
        straight into the next hash build / the aggregation as a library-owned page (buffers shared, nothing copied); with `merge_mb`
        the (small: 0.5 - 10 % of the probe page) join output pages are coalesced by a MergePagesOperator in front of the aggregation,
        the reference's remedy for small pages (M/operator/project/MergePages.java)"""
        p, ctx, f, t = self.pkg, self.ctx, self.q3_fac, self.q3
        B, D, DT, V, I = p.BIGINT, p.DOUBLE, p.DATE, p.VARCHAR, p.INTEGER
        import ctypes as C
        L = p._lib.lib()

        def slices(n):
            return [(a, min(a + page_rows, n)) for a in range(0, n, page_rows)]

        def fixed(ty, key, a, z):
            return p.DeviceBlock(ty, z - a, t[key][a:z])
        cust = [p.Page(fixed(B, "c_custkey", a, z), p.DeviceBlock(V, z - a, t["c_seg_bytes"], None, t["c_seg_off"][a:z + 1]), position_count=z - a)
                for a, z in slices(int(t["c_custkey"].numel()))]
        orders = [p.Page(fixed(B, "o_orderkey", a, z), fixed(B, "o_custkey", a, z), fixed(DT, "o_orderdate", a, z), fixed(I, "o_shippriority", a, z), position_count=z - a)
                  for a, z in slices(int(t["o_orderkey"].numel()))]
        lineitem = [p.Page(fixed(B, "l_orderkey", a, z), fixed(D, "l_extendedprice", a, z), fixed(D, "l_discount", a, z), fixed(DT, "l_shipdate", a, z), position_count=z - a)
                    for a, z in slices(int(t["l_orderkey"].numel()))]
        marshalled = {k: [pg.to_c() for pg in v] for k, v in (("customer", cust), ("orders", orders), ("lineitem", lineitem))}   # once: the timed loop is the library's work
        pp = self.entry.bench_page_processors(p)
        st = {}

        def pump(op, name, sink):
            rows = 0

            def move():
                nonlocal rows
                o = op.getOutput()
                if o is not None:
                    rows += o.position_count
                    sink.addInput(o)
                    o.release()
            for cp, keep in marshalled[name]:
                p._lib.check(L.tgpu_operator_add_input(op.handle, C.byref(cp)))
                move()
            op.finish()
            while not op.isFinished():
                move()
            return rows

        def step():
            cb = p.HashBuilderOperatorFactory(ctx, 10, [B], [], [0])
            cbuild = cb.createOperator()
            cfp = f["cust_fp"].createOperator()
            st["customer_build_rows"] = pump(cfp, "customer", cbuild)
            cbuild.finish()
            oj = p.FilterProjectLookupJoinOperatorFactory(ctx, 11, cb.lookup_source_factory, *pp["q3_orders"], [1], probe_output_channels=[0, 2, 3])
            ob = p.HashBuilderOperatorFactory(ctx, 12, [B, DT, I], [1, 2], [0])
            obuild, ojoin = ob.createOperator(), oj.createOperator()
            st["orders_build_rows"] = pump(ojoin, "orders", obuild)
            obuild.finish()
            ojoin.close()
            lj = p.FilterProjectLookupJoinOperatorFactory(ctx, 13, ob.lookup_source_factory, *pp["q3_lineitem"], [0], probe_output_channels=[0, 1])
            agg = p.HashAggregationOperatorFactory(ctx, 14, [B, DT, I], [0, 2, 3], [(p.SUM_DOUBLE, 1)], expected_groups=1 << 20)
            ljoin, aop = lj.createOperator(), agg.createOperator()
            if merge_mb:
                class Coalesced:   # MergePages in front of the aggregation
                    def __init__(self, ctx):
                        self.m = p.MergePagesOperatorFactory(ctx, 15, [B, D, DT, I], merge_mb << 20, 1 << 22, (merge_mb << 20) * 2).createOperator()

                    def drain(self):
                        while True:
                            o = self.m.getOutput()
                            if o is None:
                                return
                            aop.addInput(o)
                            o.release()

                    def addInput(self, page):
                        self.m.addInput(page)
                        self.drain()
                sink = Coalesced(ctx)
                st["lineitem_join_rows"] = pump(ljoin, "lineitem", sink)
                sink.m.finish()
                sink.drain()
                sink.m.close()
            else:
                st["lineitem_join_rows"] = pump(ljoin, "lineitem", aop)
            outs = self.finish(aop)
            st["groups"] = sum(o.position_count for o in outs)
            for o in (self.q3_result or []):
                o.release()
            self.q3_result = outs
            for op in (ljoin, cfp, cbuild, obuild, aop):
                op.close()

        step_s, prof = self.timed(step, steps, warmup, profile_apart=True)
        keep = self.q3_stats
        self.q3_stats = dict(st)
        chk = self.check_q3()
        probe_rows = self.q3_stats["lineitem_probe_rows"]
        self.q3_stats = keep
        n_pages = len(cust) + len(orders) + len(lineitem)
        return {"page_rows": page_rows, "pages": {"customer": len(cust), "orders": len(orders), "lineitem": len(lineitem)}, "merge_pages_before_aggregation_mb": merge_mb,
                "ms_per_step": step_s * 1e3, "per_kernel_profile": "one more, untimed, step (the timed steps run with the library's profile off)",
                "probe_rows_per_sec": probe_rows / step_s, "readbacks_per_page": self.last_readbacks_per_step / n_pages,
                "kernel_launches_per_page": sum(v["count"] for v in prof.values()) / steps / n_pages,
                "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}, "ok": chk["ok"]}

    def captured(self, name, out_page):
        """tests/test_gpu_bench_programs.py sets self.capture = {}: host copies of the intermediate pages of one step, so that the very
        pipeline this file times is compared with the oracle pair by pair (not part of a timed run: capture is None there)"""
        if self.capture is not None:
            self.capture.setdefault(name, []).append(out_page.to_host())

    def step_q3_dist(self):
        """Q3 as the distributed plan Trino's optimizer picks for these inputs (N ranks, one per GPU):
          * customer (small after its filter) is the REPLICATED build side of a broadcast join (join_distribution_type
            BROADCAST, M/sql/planner/optimizations/AddExchanges.java: replicated build -> FIXED_BROADCAST_DISTRIBUTION): one
            all-gather of the filtered keys over xGMI, every rank builds the full customer table;
          * orders and lineitem are partitioned on orderkey by their connector (the tpch connector exposes that partitioning,
            P/trino-tpch TpchNodePartitioningProvider), so orders |x| customer and lineitem |x| orders are co-located: the same
            fused filter+probe operators as on one GPU, no exchange of the big probe sides;
          * the aggregation groups by (l_orderkey, ...), i.e. by a superset of the partitioning column: single-step per rank, no
            exchange.  TGPU_BENCH_PLAN=partial_final runs it as PARTIAL -> FIXED_HASH_DISTRIBUTION exchange on the group keys (K10
            partition kernels + RCCL all-to-all-v) -> FINAL instead (HashAggregationOperator.java Step.PARTIAL / FINAL).
        TGPU_BENCH_PLAN=repartition runs the plan with hash-repartitioned join inputs (step_q3_dist_repartition)."""
        p, ctx, f, pages, ex = self.pkg, self.ctx, self.q3_fac, self.q3_pages, self.exchange
        B, D, DT, I = p.BIGINT, p.DOUBLE, p.DATE, p.INTEGER
        st = self.q3_stats
        sent0 = ex.bytes_sent
        # customer: filter -> broadcast -> build on every rank
        cb = p.HashBuilderOperatorFactory(ctx, 10, [B], [], [0])
        cbuild = cb.createOperator()
        outs = self.drive(f["cust_fp"].createOperator(), pages["customer"])
        local = outs[0].as_device_page() if outs else p.Page(self.dblock(B, torch.zeros(0, dtype=torch.int64, device=self.dev)), position_count=0)
        allc = ex.all_gather(local)
        st["customer_build_rows"] = allc.position_count
        if allc.position_count:
            cbuild.addInput(allc)
        allc.release()
        cbuild.finish()
        for o in outs:
            o.release()
        st["exchange_bytes_sent"] = ex.bytes_sent - sent0
        # orders: filter/project fused into the probe of the replicated customer table -> local build on orderkey
        pp = self.entry.bench_page_processors(p)
        oj = p.FilterProjectLookupJoinOperatorFactory(ctx, 11, cb.lookup_source_factory, *pp["q3_orders"], [1], probe_output_channels=[0, 2, 3])
        ob = p.HashBuilderOperatorFactory(ctx, 12, [B, DT, I], [1, 2], [0])
        obuild = ob.createOperator()
        ojoin = oj.createOperator()
        st["orders_build_rows"] = 0
        for j in self.drive(ojoin, pages["orders"]):
            self.captured("orders_join", j)
            st["orders_build_rows"] = j.position_count
            obuild.addInput(j.as_device_page())
            j.release()
        obuild.finish()
        ojoin.close()
        # lineitem: filter/project fused into the probe of the local orders table -> aggregation
        lj = p.FilterProjectLookupJoinOperatorFactory(ctx, 13, ob.lookup_source_factory, *pp["q3_lineitem"], [0], probe_output_channels=[0, 1])
        if os.environ.get("TGPU_BENCH_PLAN") != "partial_final":
            # The join output is partitioned on l_orderkey (co-located join on the connector's orderkey partitioning) and the grouping
            # keys contain it, so no group spans two ranks: AddExchanges keeps the aggregation SINGLE-step on each node, without an
            # exchange (M/sql/planner/optimizations/AddExchanges.java visitAggregation: the child's partitioning satisfies the
            # grouping keys).  check_q3_dist verifies exactly that (the per-rank group counts add up to the global count).
            agg = p.HashAggregationOperatorFactory(ctx, 14, [B, DT, I], [0, 2, 3], [(p.SUM_DOUBLE, 1)], expected_groups=1 << 20)
            ljoin = lj.createOperator()
            aop = agg.createOperator()
            st["lineitem_join_rows"] = 0
            for j in self.drive(ljoin, pages["lineitem"]):
                self.captured("lineitem_join", j)
                st["lineitem_join_rows"] = j.position_count
                aop.addInput(j.as_device_page())
                j.release()
            outs = self.finish(aop)
            st["groups"] = sum(o.position_count for o in outs)
            st["partial_groups"] = st["groups"]
            self.q3_result = outs
            ljoin.close()
            cbuild.close()
            obuild.close()
            aop.close()
            return
        # TGPU_BENCH_PLAN=partial_final: the aggregation as PARTIAL -> FIXED_HASH exchange on the group keys -> FINAL (what the plan
        # would be if the join output were not known to be partitioned on the grouping keys); exercises the all-to-all-v path
        pagg = p.HashAggregationOperatorFactory(ctx, 14, [B, DT, I], [0, 2, 3], [(p.SUM_DOUBLE, 1)], step=p.PARTIAL, expected_groups=1 << 20)
        ljoin = lj.createOperator()
        paop = pagg.createOperator()
        st["lineitem_join_rows"] = 0
        for j in self.drive(ljoin, pages["lineitem"]):
            st["lineitem_join_rows"] = j.position_count
            paop.addInput(j.as_device_page())
            j.release()
        partial = self.finish(paop)
        # exchange on the group keys -> FINAL aggregation (intermediate channels: count at 3, sum at 4)
        fagg = p.HashAggregationOperatorFactory(ctx, 15, [B, DT, I], [0, 1, 2], [(p.SUM_DOUBLE, 3)], step=p.FINAL, expected_groups=1 << 20)
        faop = fagg.createOperator()
        st["partial_groups"] = 0
        sent1 = ex.bytes_sent
        empty = None
        for o in partial:
            st["partial_groups"] += o.position_count
            pg = ex.repartition(o.as_device_page(), [0, 1, 2])
            o.release()
            if pg.position_count:
                faop.addInput(pg)
            pg.release()
        if not partial:   # a rank without groups still takes part in the collectives
            empty = p.Page(self.dblock(B, torch.zeros(0, dtype=torch.int64, device=self.dev)), self.dblock(DT, torch.zeros(0, dtype=torch.int32, device=self.dev)),
                           self.dblock(I, torch.zeros(0, dtype=torch.int32, device=self.dev)), self.dblock(B, torch.zeros(0, dtype=torch.int64, device=self.dev)),
                           self.dblock(D, torch.zeros(0, dtype=torch.float64, device=self.dev)), position_count=0)
            pg = ex.repartition(empty, [0, 1, 2])
            if pg.position_count:
                faop.addInput(pg)
            pg.release()
        st["exchange_bytes_sent"] += ex.bytes_sent - sent1
        outs = self.finish(faop)
        st["groups"] = sum(o.position_count for o in outs)
        self.q3_result = outs
        ljoin.close()
        cbuild.close()
        obuild.close()
        paop.close()
        faop.close()

    def step_q3_dist_repartition(self):
        """the same Q3 pipeline as a distributed plan (N ranks = N stages of FIXED_HASH_DISTRIBUTION,
        M/sql/planner/SystemPartitioningHandle.java:60): every exchange is the K10 partition kernel + an RCCL all-to-all-v.
        The filter/project runs in the producing fragment, so probes behind an exchange use the plain LookupJoinOperator."""
        p, ctx, f, pages, ex = self.pkg, self.ctx, self.q3_fac, self.q3_pages, self.exchange
        B, D, DT, I = p.BIGINT, p.DOUBLE, p.DATE, p.INTEGER
        st = self.q3_stats

        def filtered(fac, page):
            outs = self.drive(fac.createOperator(), page)
            return outs[0] if outs else None

        sent0 = ex.bytes_sent

        def repartition(out_page, keys, empty_types=None):
            """all ranks take part in every exchange, also a rank whose upstream produced no page"""
            if out_page is None:
                blocks = [self.dblock(t, torch.zeros(0, dtype={B: torch.int64, D: torch.float64}.get(t, torch.int32), device=self.dev)) for t in empty_types]
                return ex.repartition(p.Page(*blocks, position_count=0), keys)
            pg = ex.repartition(out_page.as_device_page(), keys)
            out_page.release()
            return pg

        # customer: filter -> exchange(custkey) -> build
        cb = p.HashBuilderOperatorFactory(ctx, 10, [B], [], [0])
        cbuild = cb.createOperator()
        cpage = repartition(filtered(f["cust_fp"], pages["customer"]), [0], [B])
        st["customer_build_rows"] = cpage.position_count
        if cpage.position_count:
            cbuild.addInput(cpage)
        cpage.release()
        cbuild.finish()
        # orders: filter -> exchange(custkey) -> probe customers -> exchange(orderkey) -> build
        opage = repartition(filtered(f["ord_fp"], pages["orders"]), [1], [B, B, DT, I])
        st["orders_probe_rows"] = opage.position_count
        # behind the exchange the filter has already run: the probe is still the FUSED operator (identity projections, no filter), i.e. the
        # pipelined fj_probe_* / fj_emit_* kernels and the DIRECT table layout, not the three-pass LookupJoinOperator
        fld = p.field
        oj = p.FilterProjectLookupJoinOperatorFactory(ctx, 11, cb.lookup_source_factory, [B, B, DT, I], None, [fld(0, B), fld(1, B), fld(2, DT), fld(3, I)], [1],
                                                      probe_output_channels=[0, 2, 3])
        ojoin = oj.createOperator()
        joined = self.drive(ojoin, opage) if opage.position_count else []
        for j in joined:
            self.captured("orders_join", j)
        opage.release()
        ob = p.HashBuilderOperatorFactory(ctx, 12, [B, DT, I], [1, 2], [0])
        obuild = ob.createOperator()
        bpage = repartition(joined[0] if joined else None, [0], [B, DT, I])
        st["orders_build_rows"] = bpage.position_count
        if bpage.position_count:
            obuild.addInput(bpage)
        bpage.release()
        obuild.finish()
        ojoin.close()
        # lineitem: filter/project -> exchange(orderkey) -> probe orders -> aggregate (groups are co-located: no second exchange)
        lpage = repartition(filtered(f["li_fp"], pages["lineitem"]), [0], [B, D])
        st["lineitem_probe_rows"] = lpage.position_count
        lj = p.FilterProjectLookupJoinOperatorFactory(ctx, 13, ob.lookup_source_factory, [B, D], None, [fld(0, B), fld(1, D)], [0], probe_output_channels=[0, 1])
        agg = p.HashAggregationOperatorFactory(ctx, 14, [B, DT, I], [0, 2, 3], [(p.SUM_DOUBLE, 1)], expected_groups=1 << 20)
        ljoin = lj.createOperator()
        aop = agg.createOperator()
        st["lineitem_join_rows"] = 0
        for j in (self.drive(ljoin, lpage) if lpage.position_count else []):
            self.captured("lineitem_join", j)
            st["lineitem_join_rows"] = j.position_count
            aop.addInput(j.as_device_page())
            j.release()
        lpage.release()
        outs = self.finish(aop)
        st["groups"] = sum(o.position_count for o in outs)
        st["exchange_bytes_sent"] = ex.bytes_sent - sent0
        self.q3_result = outs
        ljoin.close()
        cbuild.close()
        obuild.close()
        aop.close()

    def q3_top10(self, steps, warmup):
        """the tail of Q3 (q03.sql: ORDER BY revenue DESC, o_orderdate LIMIT 10) as a TopNOperator over the aggregation's output page;
        timed on its own, not part of `value`.  Checked against torch.topk of the revenue column."""
        p, ctx = self.pkg, self.ctx
        B, D, DT, I = p.BIGINT, p.DOUBLE, p.DATE, p.INTEGER
        pages = [o.as_device_page() for o in self.q3_result]
        rows = sum(pg.position_count for pg in pages)
        fac = p.TopNOperatorFactory(ctx, 16, [B, DT, I, D], 10, [3, 1], [p.DESC_NULLS_LAST, p.ASC_NULLS_LAST])
        result = {}

        def step():
            op = fac.createOperator()
            for pg in pages:
                op.addInput(pg)
            outs = self.finish(op)
            result["rows"] = [r for o in outs for r in o.to_host().rows()]
            op.close()

        step_s, prof = self.timed(step, steps, warmup)
        rev = np.concatenate([o.to_host().getBlock(3).values for o in self.q3_result])
        want = np.sort(rev)[::-1][:10].tolist()
        got = [r[3] for r in result["rows"]]
        return {"workload": "TopN(10) ORDER BY revenue DESC, o_orderdate over the Q3 groups", "input_rows": rows, "ms_per_step": step_s * 1e3,
                "rows_per_sec": rows / step_s, "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in prof.items()}, "ok": got == want}

    def q3_top10_dist(self, steps, warmup):
        """the tail of Q3 on N ranks ("final top-10 is a gather of N x 10 rows"): TopN(10) over each rank's groups ->
        all-gather of the N x 10 candidate rows (tgpu_exchange_all_gather) -> TopN(10) again on every rank.  Checked against the top 10 of
        the all-gathered per-rank torch top-10s."""
        p, ctx = self.pkg, self.ctx
        B, D, DT, I = p.BIGINT, p.DOUBLE, p.DATE, p.INTEGER
        pages = [o.as_device_page() for o in self.q3_result]
        fac = p.TopNOperatorFactory(ctx, 16, [B, DT, I, D], 10, [3, 1], [p.DESC_NULLS_LAST, p.ASC_NULLS_LAST])
        result = {}

        def step():
            op = fac.createOperator()
            for pg in pages:
                op.addInput(pg)
            local = self.finish(op)
            op.close()
            fin = fac.createOperator()
            for o in local:
                g = self.exchange.all_gather(o.as_device_page())
                fin.addInput(g)
                g.release()
                o.release()
            outs = self.finish(fin)
            result["rows"] = [r for o in outs for r in o.to_host().rows()]
            fin.close()

        step_s, prof = self.timed(step, steps, warmup)
        rev = torch.from_numpy(np.concatenate([o.to_host().getBlock(3).values for o in self.q3_result]))
        mine = torch.sort(rev, descending=True).values[:10]
        mine = torch.cat([mine, torch.full((10 - mine.numel(),), -1.0, dtype=torch.float64)]).to(self.coll_dev)
        parts = [torch.empty(10, dtype=torch.float64, device=self.coll_dev) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        want = torch.sort(torch.cat(parts), descending=True).values[:10].cpu().tolist()
        got = [r[3] for r in result["rows"]]
        return {"workload": f"TopN(10) per rank -> all-gather of {self.world} x 10 rows -> TopN(10)", "ms_per_step": step_s * 1e3, "ok": got == [w for w in want if w >= 0.0][:len(got)] and len(got) == 10}

    def check_q3_dist(self):
        """N > 1, replicated-customer plan: every count and the revenue total against a reference computed independently with torch
        (each rank evaluates its own split against the all-gathered customer segment flags; totals are all-reduced)."""
        st, t, dist = self.q3_stats, self.q3, self.dist
        n_c = t["c_custkey"].numel()
        seg_ok = (t["c_seg_bytes"][t["c_seg_off"][:-1].to(torch.int64)] == ord("B")).to(torch.uint8)
        send = seg_ok.to(self.coll_dev)
        parts = [torch.empty(n_c, dtype=torch.uint8, device=self.coll_dev) for _ in range(self.world)]
        dist.all_gather(parts, send)
        cust_ok = torch.cat([torch.zeros(1, dtype=torch.bool, device=self.dev)] + [x.to(self.dev).to(torch.bool) for x in parts])   # indexed by custkey (1-based)
        o_ok = (t["o_orderdate"] < D_1995_03_15) & cust_ok[t["o_custkey"]]
        base = int(t["o_orderkey"].min().item())
        ok_by_key = torch.zeros(int(t["o_orderkey"].max().item()) - base + 2, dtype=torch.bool, device=self.dev)
        ok_by_key[t["o_orderkey"] - base] = o_ok
        l_probe = t["l_shipdate"] > D_1995_03_15
        l_ok = l_probe & ok_by_key[t["l_orderkey"] - base]
        rev = (t["l_extendedprice"] * (1.0 - t["l_discount"]))[l_ok]
        local_groups = int(torch.unique(t["l_orderkey"][l_ok]).numel())
        total = 0.0
        for o in self.q3_result:
            total += float(np.sum(o.to_host().getBlock(3).values))
        want = torch.tensor([float(seg_ok.sum().item()), float(o_ok.sum().item()), float(l_ok.sum().item()), float(local_groups), float(rev.sum().item())],
                            dtype=torch.float64, device=self.coll_dev)
        got = torch.tensor([float(st["customer_build_rows"]) / self.world, float(st["orders_build_rows"]), float(st["lineitem_join_rows"]), float(st["groups"]), total],
                           dtype=torch.float64, device=self.coll_dev)
        dist.all_reduce(want)
        dist.all_reduce(got)
        st["orders_probe_rows"] = int((t["o_orderdate"] < D_1995_03_15).sum().item())
        st["lineitem_probe_rows"] = int(l_probe.sum().item())
        counts_ok = bool(torch.equal(want[:4], got[:4]))
        rel = abs(float(got[4].item()) - float(want[4].item())) / max(abs(float(want[4].item())), 1.0)
        names = ["customer_build_rows", "orders_build_rows", "lineitem_join_rows", "groups", "revenue"]
        return {"counts_match": counts_ok, "got": dict(zip(names, got.tolist())), "want": dict(zip(names, want.tolist())), "revenue_rel_err": rel,
                "partial_groups_local": st.get("partial_groups"), "ok": bool(counts_ok and rel < 1e-9)}

    def check_q3_dist_repartition(self):
        """N > 1: conservation checks across ranks (each rank only sees its own split, so the per-rank torch reference of the
        single-GPU check does not apply): rows in == rows out of every exchange, revenue total equals the all-reduced reference of
        the rows that survive both joins -- computed from the exchanged pages' owners via all-reduce of local partial sums."""
        st = self.q3_stats
        t = self.q3
        sent = torch.tensor([int((t["l_shipdate"] > D_1995_03_15).sum().item()), int((t["o_orderdate"] < D_1995_03_15).sum().item())], device=self.coll_dev, dtype=torch.int64)
        recv = torch.tensor([st["lineitem_probe_rows"], st["orders_probe_rows"]], device=self.coll_dev, dtype=torch.int64)
        self.dist.all_reduce(sent)
        self.dist.all_reduce(recv)
        ok = bool(torch.equal(sent, recv))
        return {"exchange_rows_conserved": ok, "sent": sent.tolist(), "received": recv.tolist(), "ok": ok}

    def check_q3(self):
        """size-independent checks at full size, computed independently with torch on the same device data"""
        t = self.q3
        cust_ok = torch.zeros(t["c_custkey"].numel() + 2, dtype=torch.bool, device=self.dev)
        seg = t["c_seg_bytes"][t["c_seg_off"][:-1].to(torch.int64)]                       # first byte: 'B' only for BUILDING
        cust_ok[t["c_custkey"]] = seg == ord("B")
        o_ok = (t["o_orderdate"] < D_1995_03_15) & cust_ok[t["o_custkey"]]
        ok_by_key = torch.zeros(int(t["o_orderkey"].max().item()) + 2, dtype=torch.bool, device=self.dev)
        ok_by_key[t["o_orderkey"]] = o_ok
        l_probe = t["l_shipdate"] > D_1995_03_15
        l_ok = l_probe & ok_by_key[t["l_orderkey"]]
        rev = (t["l_extendedprice"] * (1.0 - t["l_discount"]))[l_ok]
        want = {
            "customer_build_rows": int((seg == ord("B")).sum().item()),
            "orders_probe_rows": int((t["o_orderdate"] < D_1995_03_15).sum().item()),
            "orders_build_rows": int(o_ok.sum().item()),
            "lineitem_probe_rows": int(l_probe.sum().item()),
            "lineitem_join_rows": int(l_ok.sum().item()),
            "groups": int(torch.unique(t["l_orderkey"][l_ok]).numel()),
        }
        got = {k: self.q3_stats.get(k, want[k] if k in ("orders_probe_rows", "lineitem_probe_rows") else None) for k in want}
        ok = got == want
        self.q3_stats.update({"orders_probe_rows": want["orders_probe_rows"], "lineitem_probe_rows": want["lineitem_probe_rows"]})
        # checksum of checksums: the sum over all groups of sum(revenue) (exact per group on the GPU) vs torch's float64 sum
        total = 0.0
        for o in self.q3_result:
            host = o.to_host()
            total += float(np.sum(host.getBlock(3).values))
        ref = float(rev.sum().item())
        rel = abs(total - ref) / max(abs(ref), 1.0)
        return {"counts_match": ok, "got": got, "want": want, "revenue_rel_err": rel, "ok": bool(ok and rel < 1e-9)}

    # -- Q1 -----------------------------------------------------------------------------------------------------------
    def setup_q1(self, n):
        p = self.pkg
        t = self.q1 = gen_q1(self.dev, n, self.rank)
        pp = self.entry.bench_page_processors(p)
        V, D, DT = p.VARCHAR, p.DOUBLE, p.DATE
        self.q1_page_ = p.Page(self.dblock(V, t["returnflag"], t["off"]), self.dblock(V, t["linestatus"], t["off"]), self.dblock(D, t["quantity"]),
                              self.dblock(D, t["extendedprice"]), self.dblock(D, t["discount"]), self.dblock(D, t["tax"]), self.dblock(DT, t["shipdate"]))
        # HandTpchQuery1.java:60-133: scan filter/project fused into the hash aggregation (group by returnflag, linestatus)
        self.q1_agg = p.FilterProjectHashAggregationOperatorFactory(self.ctx, 21, *pp["q1"], [V, V], [0, 1], self.entry.q1_aggregates(p), expected_groups=16)

    def step_q1(self):
        aop = self.q1_agg.createOperator()
        aop.addInput(self.q1_page_)
        outs = self.finish(aop)
        self.q1_result = [o.to_host().rows() for o in outs]
        aop.close()

    def q1_double_sum_modes(self, steps, warmup, sf_java=1.0):
        """What `north_star` asks of DOUBLE aggregates (<= 1 ULP against the Java operators) and what the timed mode delivers, measured:
          * the line's Q1 number is timed in the EXACT order (correctly rounded exact sums: 0 ULP against the exact sum, and as far from the
            Java left-to-right sum as that sum's own rounding error);
          * TGPU_SUM_ORDER_JAVA adds every group's rows in row order -- bit-identical to DoubleSumAggregation.java:34-38 -- at the price of a
            sequential chain per group: timed here on an SF`sf_java` shard (a size that finishes in seconds);
          * the ULP distance between the two modes' sums (= EXACT vs the Java order) at page-sized inputs of 8 K and 64 K rows."""
        p = self.pkg
        keep = (getattr(self, "q1", None), getattr(self, "q1_page_", None), getattr(self, "q1_agg", None))
        out = {"timed_mode": "EXACT (correctly rounded exact sums; order independent)", "strict_mode": "TGPU_SUM_ORDER_JAVA (row order, bit-identical to the Java operators)"}

        def sums(order, n):
            self.ctx.set_double_sum_order(order)
            try:
                self.setup_q1(n)
                self.step_q1()
            finally:
                self.ctx.set_double_sum_order(p.SUM_ORDER_EXACT)
            return {(r[0], r[1]): np.array(r[2:6], dtype=np.float64) for pg in self.q1_result for r in pg}

        def ulps(a, b):
            ia, ib = a.view(np.int64).copy(), b.view(np.int64).copy()
            ia = np.where(ia < 0, np.int64(-2**63) - ia, ia)
            ib = np.where(ib < 0, np.int64(-2**63) - ib, ib)
            return int(np.abs(ia - ib).max())

        dist = {}
        for n in (8192, 65536):
            ex, ja = sums(p.SUM_ORDER_EXACT, n), sums(p.SUM_ORDER_JAVA, n)
            dist[str(n)] = max(ulps(ex[k], ja[k]) for k in ex)
        out["ulp_distance_exact_vs_java_order"] = dist
        n = int(6_000_379.02 * sf_java)
        self.ctx.set_double_sum_order(p.SUM_ORDER_JAVA)
        try:
            self.setup_q1(n)
            s, prof = self.timed(self.step_q1, max(2, steps // 4), 1)
        finally:
            self.ctx.set_double_sum_order(p.SUM_ORDER_EXACT)
        out["java_order_timing"] = {"rows": n, "ms_per_step": s * 1e3, "rows_per_sec": n / s,
                                    "kernels_ms_per_step": {k: v["total_ms"] / max(2, steps // 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:4]}}
        self.q1, self.q1_page_, self.q1_agg = keep
        self.q1_result = None
        return out

    def q1_java_order_at_line_size(self, n):
        """the strict mode on the line's own table: Q1 under TGPU_SUM_ORDER_JAVA -- every group's rows sorted out of the page and added in row
        order, one workgroup per group (`fa_ordered_chain`: the additions of one sum are a dependent chain, ~7.5 ns per row of the largest group)
        -- timed, and sum(extendedprice) compared BIT FOR BIT with the sequential sums check_q1 computed on the host (numpy cumsum)"""
        p = self.pkg
        self.ctx.set_double_sum_order(p.SUM_ORDER_JAVA)
        try:
            s, prof = self.timed(self.step_q1, 2, 1)
            rows = [r for pg in self.q1_result for r in pg]
        finally:
            self.ctx.set_double_sum_order(p.SUM_ORDER_EXACT)
        out = {"rows": n, "ms_per_step": s * 1e3, "rows_per_sec": n / s,
               "kernels_ms_per_step": {k: v["total_ms"] / 2 for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:3]}}
        java = getattr(self, "q1_java_price", None)
        if java:
            out["sum_extendedprice_bit_identical_to_java_order"] = bool(len(rows) == len(java) and all(
                np.float64(r[3]).view(np.int64) == np.float64(java.get(r[0] + r[1], np.nan)).view(np.int64) for r in rows))
        self.q1_result = None
        return out

    def setup_q1_dist(self):
        """Q1 on N ranks: every rank aggregates its own row-range shard with the fused PARTIAL operator, the 4-row
        partial pages are all-gathered in rank order (tgpu_exchange_all_gather) and a FINAL HashAggregationOperator combines them on every
        rank -- the rank order is fixed, so the DOUBLE results are reproducible"""
        p = self.pkg
        self.ensure_exchange()
        pp = self.entry.bench_page_processors(p)
        V = p.VARCHAR
        aggs = self.entry.q1_aggregates(p)
        self.q1_partial = p.FilterProjectHashAggregationOperatorFactory(self.ctx, 22, *pp["q1"], [V, V], [0, 1], aggs, step=p.PARTIAL, expected_groups=16)
        final_aggs, ch = [], 2
        for a in aggs:                      # intermediate layout: keys, then per aggregate its count [and its sum]
            final_aggs.append((a[0], ch))
            ch += 1 if a[0] in (p.COUNT_ALL, p.COUNT_COLUMN) else 2
        self.q1_final = p.HashAggregationOperatorFactory(self.ctx, 23, [V, V], [0, 1], final_aggs, step=p.FINAL, expected_groups=16)

    def step_q1_dist(self):
        aop = self.q1_partial.createOperator()
        aop.addInput(self.q1_page_)
        parts = self.finish(aop)
        fop = self.q1_final.createOperator()
        for o in parts:
            g = self.exchange.all_gather(o.as_device_page())
            fop.addInput(g)
            g.release()
            o.release()
        outs = self.finish(fop)
        self.q1_result = [o.to_host().rows() for o in outs]
        aop.close()
        fop.close()

    def check_q1_dist(self):
        """counts exact, sums against the all-reduced torch sums of every rank's shard"""
        t = self.q1
        sel = t["shipdate"] <= 10471
        rows = [r for pg in self.q1_result for r in pg]
        combos = ((65, 70), (78, 70), (78, 79), (82, 70))
        local = torch.zeros(len(combos), 5, dtype=torch.float64, device=self.dev)
        for i, (rf, ls) in enumerate(combos):
            m = sel & (t["returnflag"] == rf) & (t["linestatus"] == ls)
            q, e, d, x = t["quantity"][m], t["extendedprice"][m], t["discount"][m], t["tax"][m]
            local[i] = torch.stack([q.sum(), e.sum(), (e * (1 - d)).sum(), (e * (1 - d) * (1 + x)).sum(), m.sum().to(torch.float64)])
        tot = local.to(self.coll_dev)
        self.dist.all_reduce(tot)
        want = {(chr(rf), chr(ls)): tot[i].tolist() for i, (rf, ls) in enumerate(combos)}
        ok, worst = len(rows) == 4, 0.0
        for r in rows:
            w = want.get((r[0], r[1]))
            if w is None:
                ok = False
                continue
            ok = ok and r[9] == int(w[4])
            for g, v in zip(r[2:6], w[:4]):
                worst = max(worst, abs(g - v) / max(abs(v), 1.0))
        return {"groups": len(rows), "sum_rel_err_vs_all_reduced_torch": worst, "ok": bool(ok and worst < 1e-9)}

    def q1_page(self, a, z):
        """rows [a, z) of the Q1 input as a device page (zero-copy views of the generated columns)"""
        p, t = self.pkg, self.q1
        V, D, DT = p.VARCHAR, p.DOUBLE, p.DATE
        m = z - a
        vb = lambda key: p.DeviceBlock(V, m, t[key], None, t["off"][a:z + 1])
        db = lambda ty, key: p.DeviceBlock(ty, m, t[key][a:z])
        return p.Page(vb("returnflag"), vb("linestatus"), db(D, "quantity"), db(D, "extendedprice"), db(D, "discount"), db(D, "tax"), db(DT, "shipdate"), position_count=m)

    def q1_paged(self, steps, warmup, page_rows, merge_mb=None):
        """Q1 fed as pages of `page_rows` rows (the engine hands pages, not tables): directly, or through a MergePagesOperator with
        `merge_mb` MB thresholds in front (DESIGN.md "Page granularity")"""
        p, ctx = self.pkg, self.ctx
        n = int(self.q1["quantity"].numel())
        pages = [self.q1_page(a, min(a + page_rows, n)) for a in range(0, n, page_rows)]
        cpages = [pg.to_c() for pg in pages]     # marshalled once: the timed loop is the library's work, not ctypes struct building
        types = [p.VARCHAR, p.VARCHAR, p.DOUBLE, p.DOUBLE, p.DOUBLE, p.DOUBLE, p.DATE]
        mfac = p.MergePagesOperatorFactory(ctx, 20, types, merge_mb << 20, 1 << 27, (merge_mb << 20) * 2) if merge_mb else None
        L = p._lib.lib()
        import ctypes as C
        res = {}

        def step():
            aop = self.q1_agg.createOperator()
            if mfac is None:
                for cp, keep in cpages:
                    p._lib.check(L.tgpu_operator_add_input(aop.handle, C.byref(cp)))
            else:
                m = mfac.createOperator()

                def drain():
                    while True:
                        o = m.getOutput()
                        if o is None:
                            return
                        aop.addInput(o)        # a library-owned page: buffers shared, nothing copied
                        o.release()
                for cp, keep in cpages:
                    p._lib.check(L.tgpu_operator_add_input(m.handle, C.byref(cp)))
                    drain()
                m.finish()
                drain()
                m.close()
            outs = self.finish(aop)
            res["rows"] = [o.to_host().rows() for o in outs]
            aop.close()

        step_s, prof = self.timed(step, steps, warmup, profile_apart=True)
        self.q1_result = res["rows"]
        ok = self.check_q1()["ok"]
        return {"page_rows": page_rows, "pages": len(pages), "through_merge_pages_mb": merge_mb, "ms_per_step": step_s * 1e3, "rows_per_sec": n / step_s,
                "readbacks_per_page": self.last_readbacks_per_step / max(len(pages), 1),
                "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:6]},
                "kernel_launches_per_page": sum(v["count"] for v in prof.values()) / steps / max(len(pages), 1), "ok": ok}

    def q1_pcie_inclusive(self, steps, warmup, n, page_rows, pinned):
        """Q1 with every input byte crossing PCIe inside the timed region: the columns live in host memory (pinned = hipHostMalloc'ed
        through tgpu_pinned_alloc, else pageable numpy arrays), fed as pages of `page_rows` rows; every add_input stages its arrays
        through the double-buffered ingest ring on the copy stream while the kernels of the previous page run"""
        p, ctx = self.pkg, self.ctx
        t = gen_q1(self.dev, n, self.rank)
        V, D, DT = p.VARCHAR, p.DOUBLE, p.DATE
        host, holders = {}, []
        for key in ("returnflag", "linestatus", "quantity", "extendedprice", "discount", "tax", "shipdate"):
            src = t[key].cpu().numpy()
            if pinned:
                arr, holder = ctx.pinned_array(src.dtype, src.size)
                np.copyto(arr, src)
                holders.append(holder)
                host[key] = arr
            else:
                host[key] = src
        del t
        torch.cuda.empty_cache()
        off0 = np.arange(page_rows + 1, dtype=np.int32)
        pages = []
        for a in range(0, n, page_rows):
            z = min(a + page_rows, n)
            m = z - a
            vb = lambda key: p.Block(V, host[key][a:z], None, off0[:m + 1])
            fb = lambda ty, key: p.Block(ty, host[key][a:z])
            pages.append(p.Page(vb("returnflag"), vb("linestatus"), fb(D, "quantity"), fb(D, "extendedprice"), fb(D, "discount"), fb(D, "tax"), fb(DT, "shipdate")))
        cpages = [pg.to_c() for pg in pages]
        L = p._lib.lib()
        import ctypes as C
        res = {}

        def step():
            aop = self.q1_agg.createOperator()
            for cp, keep in cpages:
                p._lib.check(L.tgpu_operator_add_input(aop.handle, C.byref(cp)))
            outs = self.finish(aop)
            res["rows"] = [o.to_host().rows() for o in outs]
            aop.close()

        step_s, prof = self.timed(step, steps, warmup)
        groups = sum(len(r) for r in res["rows"])
        for h in holders:
            h.free()
        return {"rows": n, "pages": len(pages), "page_rows": page_rows, "host_memory": "pinned (tgpu_pinned_alloc)" if pinned else "pageable", "ms_per_step": step_s * 1e3,
                "rows_per_sec": n / step_s, "host_to_hbm_GBps": 46.0 * n / step_s / 1e9, "groups": groups, "ok": groups == 4}

    def check_q1(self, java_order_distance=False):
        t = self.q1
        sel = t["shipdate"] <= 10471
        rows = [r for pg in self.q1_result for r in pg]
        ok = len(rows) == 4
        want = {}
        for rf, ls in ((65, 70), (78, 70), (78, 79), (82, 70)):
            m = sel & (t["returnflag"] == rf) & (t["linestatus"] == ls)
            q, e, d, x = t["quantity"][m], t["extendedprice"][m], t["discount"][m], t["tax"][m]
            want[(chr(rf), chr(ls))] = (float(q.sum()), float(e.sum()), float((e * (1 - d)).sum()), float((e * (1 - d) * (1 + x)).sum()), int(m.sum()))
        worst = 0.0
        for r in rows:
            w = want.get((r[0], r[1]))
            if w is None:
                ok = False
                continue
            ok = ok and r[9] == w[4]
            for g, v in zip(r[2:6], w[:4]):
                worst = max(worst, abs(g - v) / max(abs(v), 1.0))
        # first-seen group order must equal the order in which the four combos first appear among the selected rows
        first = {}
        idx = torch.nonzero(sel)[:4096, 0]
        for i in idx.tolist():
            k = (chr(int(t["returnflag"][i])), chr(int(t["linestatus"][i])))
            first.setdefault(k, len(first))
            if len(first) == 4:
                break
        order_ok = [(r[0], r[1]) for r in rows] == sorted(first, key=first.get) if len(first) == 4 else True
        out = {"groups": len(rows), "sum_rel_err_vs_torch": worst, "group_order_ok": bool(order_ok), "ok": bool(ok and order_ok and worst < 1e-9)}
        if java_order_distance and rows:
            # DoubleSumAggregation.java:34-38 adds the rows of a group left to right; numpy's cumsum is that very loop (a sequential ufunc
            # accumulate).  Distance of the GPU's exactly rounded sum(extendedprice) to it, in ULP -- the Java order's own rounding error at
            # this many rows per group (DESIGN.md "DOUBLE aggregate policy"); SUM_ORDER_JAVA gives 0 here at the price of a sequential chain
            t0 = time.perf_counter()
            price = t["extendedprice"].cpu().numpy()
            rf, ls, sd = t["returnflag"].cpu().numpy(), t["linestatus"].cpu().numpy(), t["shipdate"].cpu().numpy()
            dist = {}
            self.q1_java_price = {}
            for r in rows:
                m = (sd <= 10471) & (rf == ord(r[0])) & (ls == ord(r[1]))
                java = float(np.cumsum(price[m])[-1])
                self.q1_java_price[r[0] + r[1]] = java
                a, b = np.float64(r[3]).view(np.int64), np.float64(java).view(np.int64)
                dist[r[0] + r[1]] = int(abs(int(a) - int(b)))
            out["sum_extendedprice_ulp_distance_to_java_order"] = dist
            out["java_order_check_seconds"] = time.perf_counter() - t0
        return out

    # -- cfg2 ---------------------------------------------------------------------------------------------------------
    def setup_cfg2(self, n):
        p = self.pkg
        i = torch.arange(n, device=self.dev, dtype=torch.int64)
        self.c2 = [rnd(21, i, 1000), rnd(22, i, 1 << 20), rnd(23, i, 1 << 20)]
        self.c2_fp = p.FilterAndProjectOperatorFactory(self.ctx, 30, *self.entry.bench_page_processors(p)["cfg2"])
        self.c2_page = p.Page(*[self.dblock(p.BIGINT, c) for c in self.c2])

    def step_cfg2(self):
        op = self.c2_fp.createOperator()
        outs = self.drive(op, self.c2_page)
        self.c2_out = outs

    def check_cfg2(self):
        o = self.c2_out[0]
        want = (self.c2[1] * self.c2[2])[self.c2[0] > 899]
        host = o.to_host().getBlock(0).values
        return {"rows": int(want.numel()), "ok": bool(np.array_equal(host, want.cpu().numpy()))}


# ---------------------------------------------------------------------------------------------------------------------
# Sub-benchmarks (N = 1 only; each with its own `roofline`): the table layouts and operator shapes the TPCH headline never reaches
# -- both TPCH build sides get the DIRECT bitmap + rank layout --, so that the open-address tables north_star names are measured too:
#   join_hash_layout : a Q3-lineitem-shaped fused filter + probe whose build keys are sparse random 64-bit values
#                      (TgSlot16 open-address table + blocked Bloom pre-filter), one DOUBLE build output channel
#   join_duplicate_keys : every build key twice (position links, PagesHash.java:102-117), unfused LookupJoinOperator
#   group_by_hash    : T/operator/BenchmarkGroupByHash.java:65-74,119-137 bigintGroupByHash: addPage of BIGINT keys uniform in
#                      [0, groups) + appendValuesTo of every group (10 M rows / 3 M groups as in the reference, and 100 M / 40 M)
# ---------------------------------------------------------------------------------------------------------------------
# What bounds the open-address tables is not HBM bytes but the rate at which this part serves RANDOM accesses to tables beyond its caches,
# measured in isolation by tools/exp_random_access.hip (G accesses per second, whole chip): loads 56 / 55 / 54 and returning
# atomics (CAS / atomicMin) 27 / 22 / 18 for tables of <= 128 MB / 512 MB / >= 2 GB.  The byte roofline of these kernels stays in `roofline`;
# this object prices them in requests.
def random_access_rates(table_bytes):
    """(G random loads/s, G returning atomics/s) of tools/exp_random_access.hip for a table of this size: Infinity-Cache-sized, 512 MB, DRAM-sized"""
    if table_bytes <= (128 << 20):
        return 56.0, 27.0
    if table_bytes <= (512 << 20):
        return 55.0, 22.0
    return 54.0, 18.0


def request_roofline(profile, steps, kernel, rows, loads_per_row, atomics_per_row, table_bytes, note):
    st = profile.get(kernel)
    if not st or st["count"] == 0:
        return None
    ms = st["total_ms"] / steps
    achieved = rows / (ms * 1e-3) / 1e9
    RANDOM_LOADS_G_PER_S, RANDOM_ATOMICS_G_PER_S = random_access_rates(table_bytes)
    # (loads are served by the cache hierarchy, returning atomics at the memory side: the two streams overlap, the slower one bounds)
    peak = 1.0 / max(loads_per_row / RANDOM_LOADS_G_PER_S, atomics_per_row / RANDOM_ATOMICS_G_PER_S)
    return {"bound": "random_access", "kernel": kernel, "achieved": achieved, "peak": peak, "unit": "G rows/s", "frac": achieved / peak, "kernel_ms_per_step": ms,
            "rows_per_step": rows, "random_loads_per_row": loads_per_row, "returning_atomics_per_row": atomics_per_row, "table_bytes": table_bytes,
            "peak_source": f"tools/exp_random_access.hip on MI355X (DESIGN.md 4a): {RANDOM_LOADS_G_PER_S:g} G random loads/s, {RANDOM_ATOMICS_G_PER_S:g} G returning atomics/s "
                           "for a table of this size; a measured ceiling of the isolated access pattern, not a datasheet number", "note": note}


def sub_join_hash_layout(b, steps, warmup, sf):
    p, ctx, dev = b.pkg, b.ctx, b.dev
    B, D, I = p.BIGINT, p.DOUBLE, p.INTEGER
    nb, n = int(146_000 * sf), int(6_000_000 * sf)
    bi = torch.arange(nb, device=dev, dtype=torch.int64)
    bkeys = splitmix64(bi + 7_777_777)                        # a bijection: distinct, spread over all 64 bits
    payload = bi.to(torch.float64)
    i = torch.arange(n, device=dev, dtype=torch.int64)
    r = rnd(31, i, 8 * nb)                                    # 1 probe row in 8 hits the build side
    pkeys = splitmix64(r + 7_777_777)
    fcol = rnd(32, i, 100).to(torch.int32)
    f = p.field
    filt, projs = f(1, I) > 45, [f(0, B)]                    # 54 % pass, like Q3's shipdate filter
    bpage = p.Page(b.dblock(B, bkeys), b.dblock(D, payload))
    ppage = p.Page(b.dblock(B, pkeys), b.dblock(I, fcol))
    res = {}

    def step():
        bf = p.HashBuilderOperatorFactory(ctx, 40, [B, D], [1], [0])
        bop = bf.createOperator()
        bop.addInput(bpage)
        bop.finish()
        jf = p.FilterProjectLookupJoinOperatorFactory(ctx, 41, bf.lookup_source_factory, [B, I], filt, projs, [0], probe_output_channels=[0])
        jop = jf.createOperator()
        for o in res.pop("outs", []):
            o.release()
        outs = b.drive(jop, ppage)
        res["stats"] = bf.lookup_source_factory.stats()
        res["pairs"] = sum(o.position_count for o in outs)
        res["outs"] = outs            # checked after the timed region
        jop.close()
        bop.close()

    step_s, prof = b.timed(step, steps, warmup)
    passing = fcol > 45
    hit = passing & (r < nb)
    want_pairs, want_sum = int(hit.sum().item()), float(r[hit].to(torch.float64).sum().item())
    n_pass = int(passing.sum().item())
    got_sum = 0.0
    for o in res.pop("outs", []):
        got_sum += float(np.sum(o.to_host().getBlock(1).values))
        o.release()
    ok = res["pairs"] == want_pairs and abs(got_sum - want_sum) <= 1e-9 * max(abs(want_sum), 1.0) and res["stats"]["link_count"] == 0
    # algorithmic bytes of the probe launch, hash layout: filter column 4 B x input rows + (key 8 B + one 16-byte TgSlot16) x rows
    # passing the filter + 8 B per emitted pair
    alg = 4.0 * n + 24.0 * n_pass + 8.0 * want_pairs
    roof = dominant(prof, steps, {"fused_filter_probe": n}, {"fused_filter_probe": alg / n}, pmc_prefix="sub_join_hash_layout:")
    return {"workload": "fused filter + probe, sparse random 64-bit build keys (open-address TgSlot16 table + Bloom pre-filter)", "build_rows": nb, "input_rows": n,
            "probe_rows": n_pass, "pairs": want_pairs, "table_slots": res["stats"]["hash_size"], "ms_per_step": step_s * 1e3, "probe_rows_per_sec": n_pass / step_s,
            "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}, "roofline": roof,
            "request_roofline": request_roofline(prof, steps, "fused_filter_probe", n_pass, 1.0 + want_pairs / max(n_pass, 1), 0.0, res["stats"]["hash_size"] * 16,
                                                 "one Bloom word per row that passes the filter (29 MB filter: beyond the L2) + one table slot per survivor"), "ok": bool(ok)}


def sub_join_duplicate_keys(b, steps, warmup, sf):
    p, ctx, dev = b.pkg, b.ctx, b.dev
    B = p.BIGINT
    nb, n = int(146_000 * sf), int(1_000_000 * sf)
    bi = torch.arange(nb, device=dev, dtype=torch.int64)
    bkeys = splitmix64((bi // 2) + 99_999)                    # every key twice -> position links
    i = torch.arange(n, device=dev, dtype=torch.int64)
    r = rnd(33, i, 2 * nb)                                    # distinct keys: nb / 2 -> 1 probe row in 4 matches, two pairs each
    pkeys = splitmix64(r + 99_999)
    bpage, ppage = p.Page(b.dblock(B, bkeys)), p.Page(b.dblock(B, pkeys))
    res = {}

    def step():
        bf = p.HashBuilderOperatorFactory(ctx, 42, [B], [0], [0])
        bop = bf.createOperator()
        bop.addInput(bpage)
        bop.finish()
        jf = p.LookupJoinOperatorFactory(ctx, 43, bf.lookup_source_factory, [B], [0])
        jop = jf.createOperator()
        outs = b.drive(jop, ppage)
        res["stats"] = bf.lookup_source_factory.stats()
        res["pairs"] = sum(o.position_count for o in outs)
        for o in outs:
            o.release()
        jop.close()
        bop.close()

    step_s, prof = b.timed(step, steps, warmup)
    want_pairs = 2 * int((r < (nb + 1) // 2).sum().item())
    ok = res["pairs"] == want_pairs and res["stats"]["link_count"] > 0
    # unfused probe: key 8 B + one table slot 16 B + head / count out 8 B per probe row
    roof = dominant(prof, steps, {"join_probe_count": n}, {"join_probe_count": 32.0}, pmc_prefix="sub_join_duplicate_keys:")
    return {"workload": "LookupJoinOperator over a table with every build key twice (position links, newest -> oldest chains)", "build_rows": nb, "probe_rows": n,
            "pairs": want_pairs, "link_count": res["stats"]["link_count"], "ms_per_step": step_s * 1e3, "probe_rows_per_sec": n / step_s,
            "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}, "roofline": roof,
            "request_roofline": request_roofline(prof, steps, "join_probe_count", n, 1.0, 0.0, res["stats"]["hash_size"] * 16,
                                                 "one table slot per probe row (the chain length sits in the slot)"), "ok": bool(ok)}


def sub_group_by_hash(b, steps, warmup, rows, groups):
    p, ctx, dev = b.pkg, b.ctx, b.dev
    B = p.BIGINT
    i = torch.arange(rows, device=dev, dtype=torch.int64)
    keys = rnd(34, i, groups)
    page = p.Page(b.dblock(B, keys))
    res = {}

    def step():
        g = p.GroupByHash(ctx, [B], [0], expected_size=10_000)   # EXPECTED_SIZE of the reference benchmark: the table grows by rehashing
        g.addPage(page)
        out = g.appendValuesDevice()
        res["groups"] = out.position_count
        res["capacity"] = g.getCapacity()
        out.release()
        g.close()

    step_s, prof = b.timed(step, steps, warmup)
    want = int(torch.unique(keys).numel())
    # rows that pay the returning atomic: those whose key has not been published by an EARLIER sub-batch of 2^24 rows
    batch = i >> 24
    first_batch = torch.full((groups,), 1 << 40, device=dev, dtype=torch.int64).scatter_reduce_(0, keys, batch, "amin")
    atomic_rows = int((first_batch[keys] == batch).sum().item())
    # gbh_insert per row: key 8 B + one 8-byte table word + the 4-byte group id it answers with
    shape = f"{rows // 1_000_000}M_{groups // 1_000_000}M"
    roof = dominant(prof, steps, {"gbh_insert": rows}, {"gbh_insert": 20.0}, pmc_prefix=f"sub_group_by_hash_{shape}:")
    return {"workload": f"BenchmarkGroupByHash.bigintGroupByHash shape: addPage of {rows} BIGINT keys uniform in [0, {groups}) + appendValuesTo of every group",
            "rows": rows, "groups": want, "ms_per_step": step_s * 1e3, "rows_per_sec": rows / step_s, "ns_per_row": step_s * 1e9 / rows,
            "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}, "roofline": roof,
            "request_roofline": request_roofline(prof, steps, "gbh_insert", rows, 1.0, atomic_rows / rows, res["capacity"] * 16,
                                                 "one slot load per row; one returning atomicMin per row whose group is new in its 2^24-row sub-batch (counted from the keys: "
                                                 f"{atomic_rows} of {rows} rows), rows of groups published by an earlier sub-batch only read"),
            "ok": bool(res["groups"] == want)}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle = a row-at-a-time C port of the Java operators, timed on this box's host cores on a
# bounded sample of the Q3 workload -- on 1 thread and on T = all cores of this process's CPU share, one independent operator
# instance per thread over a row-range shard (Trino's task.concurrency drivers, M/execution/TaskManagerConfig.java:69).
# Every step is the row-at-a-time port: filters (o_filter), projections (o_project), PagesHash build + JoinProbe, MultiChannelGroupByHash
# over the three grouping keys, DoubleSum in row order; numpy only moves columns between the steps (what LookupJoinPageBuilder /
# PageBuilder do in the reference).
# ---------------------------------------------------------------------------------------------------------------------
def cpu_q3_pipeline(oracle, pkg, entry, h, threads):
    from concurrent.futures import ThreadPoolExecutor
    E = pkg.expressions
    pp = entry.bench_page_processors(pkg)

    def program(name):
        types, filt, projs = pp[name]
        return E.FlatProgram(filt, projs)

    def shards(n):
        step = (n + threads - 1) // threads
        return [(a, min(a + step, n)) for a in range(0, n, step)]

    pool = ThreadPoolExecutor(max_workers=threads)
    # customer: mktsegment = 'BUILDING' -> build on custkey
    prog = program("q3_customer")
    ccols = [oracle.Col(oracle.BIGINT, h["c_custkey"]), oracle.Col(oracle.VARCHAR, h["c_seg_bytes"], None, h["c_seg_off"])]
    cpos = oracle.filter_positions(prog.nodes, prog.filter_root, bytes(prog.pool), ccols)
    ck = h["c_custkey"][cpos]
    cust = oracle.PagesHash([oracle.Col(oracle.BIGINT, ck)])

    # orders: orderdate < 1995-03-15, probe customers on custkey -> (orderkey, orderdate, shippriority)
    oprog = program("q3_orders")

    def orders_shard(rng):
        a, b = rng
        cols = [oracle.Col(oracle.BIGINT, h["o_orderkey"][a:b]), oracle.Col(oracle.BIGINT, h["o_custkey"][a:b]), oracle.Col(oracle.DATE, h["o_orderdate"][a:b]),
                oracle.Col(oracle.INTEGER, h["o_shippriority"][a:b])]
        pos = oracle.filter_positions(oprog.nodes, oprog.filter_root, b"", cols)
        op, _ = cust.probe([oracle.Col(oracle.BIGINT, h["o_custkey"][a:b][pos])])
        rows = pos[op] + a
        return h["o_orderkey"][rows], h["o_orderdate"][rows], h["o_shippriority"][rows]

    parts = list(pool.map(orders_shard, shards(len(h["o_orderkey"]))))
    okeys, odate, oprio = (np.concatenate([p[k] for p in parts]) for k in range(3))
    orders = oracle.PagesHash([oracle.Col(oracle.BIGINT, okeys)])

    # lineitem: shipdate > 1995-03-15, revenue = extendedprice * (1 - discount), probe orders, group by 3 keys, sum(revenue)
    lprog = program("q3_lineitem")
    probe_rows = [0]

    def lineitem_shard(rng):
        a, b = rng
        cols = [oracle.Col(oracle.BIGINT, h["l_orderkey"][a:b]), oracle.Col(oracle.DOUBLE, h["l_extendedprice"][a:b]), oracle.Col(oracle.DOUBLE, h["l_discount"][a:b]),
                oracle.Col(oracle.DATE, h["l_shipdate"][a:b])]
        pos = oracle.filter_positions(lprog.nodes, lprog.filter_root, b"", cols)
        rev, _ = oracle.project(lprog.nodes, lprog.projection_roots[1], b"", cols, pos)
        lk = h["l_orderkey"][a:b][pos]
        lp, lb = orders.probe([oracle.Col(oracle.BIGINT, lk)])
        kcols = [oracle.Col(oracle.BIGINT, lk[lp]), oracle.Col(oracle.DATE, odate[lb]), oracle.Col(oracle.INTEGER, oprio[lb])]
        g = oracle.MultiChannelGroupByHash([oracle.BIGINT, oracle.DATE, oracle.INTEGER], 1 << 16)
        gids = g.get_group_ids(kcols, oracle.hash_rows(kcols))
        cnt, sums = oracle.agg_double_sum(gids, rev[lp], g.group_count)
        first, _ = g.group_rows()
        return len(pos), lk[lp][first], odate[lb][first], oprio[lb][first], cnt, sums

    parts = list(pool.map(lineitem_shard, shards(len(h["l_orderkey"]))))
    probe_rows = sum(p[0] for p in parts)
    if threads > 1:   # FINAL step over the per-thread partial aggregates (HashAggregationOperator Step.PARTIAL -> FINAL)
        kcols = [oracle.Col(oracle.BIGINT, np.concatenate([p[1] for p in parts])), oracle.Col(oracle.DATE, np.concatenate([p[2] for p in parts])),
                 oracle.Col(oracle.INTEGER, np.concatenate([p[3] for p in parts]))]
        g = oracle.MultiChannelGroupByHash([oracle.BIGINT, oracle.DATE, oracle.INTEGER], 1 << 16)
        gids = g.get_group_ids(kcols, oracle.hash_rows(kcols))
        oracle.agg_double_sum(gids, np.concatenate([p[5] for p in parts]), g.group_count)
        groups = g.group_count
    else:
        groups = len(parts[0][4])
    pool.shutdown()
    return probe_rows, len(okeys), groups


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(bench, sample_sf):
    from oracle import oracle
    t = gen_q3(bench.dev, sample_sf, 0)
    h = {k: v.cpu().numpy() for k, v in t.items()}
    del t
    torch.cuda.empty_cache()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    probe_rows, build_rows, groups = cpu_q3_pipeline(oracle, bench.pkg, bench.entry, h, 1)
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    probe_rows_t, _, groups_t = cpu_q3_pipeline(oracle, bench.pkg, bench.entry, h, cores)
    dtt = time.perf_counter() - t0
    assert probe_rows_t == probe_rows and groups_t == groups
    return {"value": float(probe_rows / dtt), "unit": "probe rows/s", "cores": cores, "kind": "port",
            "value_1_thread": float(probe_rows / dt1), "seconds_1_thread": dt1, "seconds_all_threads": dtt, "cpu": cpu_model(),
            "sample": f"same Q3 pipeline at SF{sample_sf:g} ({probe_rows} lineitem probe rows, {build_rows} build rows, {groups} groups): C oracle = row-at-a-time port of "
                      f"the generated filter / projection loops, PagesHash / JoinHash, MultiChannelGroupByHash and DoubleSum; 1 thread {dt1:.1f} s, {cores} threads "
                      f"(one operator instance per row-range shard, partial -> final aggregation) {dtt:.1f} s; reference Java operators not runnable: no JVM"}


PMC_PROFILES = ["r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_v5_pmc_traffic.json"]   # newest first


def dominant(profile, steps, rows_per_step, bytes_per_row, pmc_prefix=""):
    """Roofline of the kernel that takes most of the step's device time among those priced in `bytes_per_row`.
    achieved = ALGORITHMIC bytes of ONE STEP (bytes_per_row x the rows the kernel processes per step, whatever number of launches they are
    spread over) / the kernel's TOTAL time per step, HIP-event timed on the context's stream.  (Round 2 divided the whole step's bytes by the
    AVERAGE launch: right for one launch per step, 3-9x too high for the group-by table's 3 / 9 sub-batch launches.)"""
    best = None
    for name, st in profile.items():
        if name not in bytes_per_row or st["count"] == 0:
            continue
        if best is None or st["total_ms"] > profile[best]["total_ms"]:
            best = name
    if best is None:
        return None
    st = profile[best]
    launches = st["count"] / steps
    ms_per_step = st["total_ms"] / steps
    alg_bytes_step = bytes_per_row[best] * rows_per_step[best]
    achieved = alg_bytes_step / (ms_per_step * 1e-3) / 1e9
    # HBM bytes per launch: NOT measured by this process (PMC counters need their own rocprofv3 --pmc passes); the figure is read from
    # the committed summary of those passes over this very command and labelled as such (`traffic_source`); null when absent
    traffic, traffic_source = None, None
    for name in PMC_PROFILES:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"].get(pmc_prefix + best)   # sub-benchmarks: "sub_<name>:<kernel>"
        except (OSError, ValueError, KeyError):
            continue
        if pmc:
            traffic, traffic_source = pmc["traffic_bytes_per_launch_avg"], f"profiles/{name} (committed rocprofv3 --pmc passes of `python bench.py`, not this run)"
            break
    return {"bound": "hbm", "kernel": best, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
            "avg_launch_ms": ms_per_step / launches, "launches": st["count"], "launches_per_step": launches, "kernel_ms_per_step": ms_per_step,
            "algorithmic_bytes_per_step": alg_bytes_step, "algorithmic_bytes_per_launch": alg_bytes_step / launches,
            "algorithmic_bytes_per_row": bytes_per_row[best], "rows_per_step": rows_per_step[best], "rows_per_launch": rows_per_step[best] / launches}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sf", type=float, default=100.0, help="TPCH scale factor per GPU")
    ap.add_argument("--only", default="", help="comma list of q3,q1,cfg2,sub,paged (default at N=1: all of them; at N>1: q3,q1 as distributed plans); paged = the same programs fed as 2^20- / 2^24-row pages and the PCIe-inclusive Q1 line (profiler passes name the other four, so that their per-kernel averages are those of the headline launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-sf", type=float, default=20.0, help="scale factor of the bounded sample the CPU baseline runs (~10-25 s of CPU work)")
    args = ap.parse_args()
    b = Bench(args)
    assert b.world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={b.world}"
    only = set(args.only.split(",")) if args.only else ({"q3", "q1", "cfg2", "sub", "paged"} if b.world == 1 else {"q3", "q1"})
    b.ctx.profile_enable(os.environ.get("TGPU_BENCH_NOPROFILE") is None)   # (kernel study: cost of the event timers)
    out = {}
    extra = {}

    # ---- Q3 (headline) ----
    b.setup_q3(args.sf)
    distributed = b.dist is not None
    repartition = distributed and os.environ.get("TGPU_BENCH_PLAN") == "repartition"
    step_fn = b.step_q3 if not distributed else (b.step_q3_dist_repartition if repartition else b.step_q3_dist)
    step_s, prof = b.timed(step_fn, args.steps, args.warmup)
    q3_readbacks = b.last_readbacks_per_step
    q3_check = (b.check_q3_dist_repartition() if repartition else b.check_q3_dist()) if distributed else b.check_q3()
    st = dict(b.q3_stats)
    probe_rows = st["lineitem_probe_rows"]
    total_probe = probe_rows
    if b.dist is not None:
        tt = torch.tensor([probe_rows], device=b.coll_dev, dtype=torch.int64)
        b.dist.all_reduce(tt)
        total_probe = int(tt.item())
    # per-kernel algorithmic bytes per row (DESIGN.md "kernels and their rooflines")
    # fused filter+probe kernel, launched twice per step (orders, lineitem).  Algorithmic bytes per launch, for the
    # DIRECT table layout both TPCH build sides get: filter column 4 B x input rows + (key 8 B + one 8-byte bitmap word) x rows
    # passing the filter + (rank word 4 B + pair 8 B) x emitted pairs.  (The hash-table layout reads a 12-byte slot instead of the
    # bitmap word + rank word.)
    n_o, n_l = int(b.q3["o_orderkey"].numel()), int(b.q3["l_orderkey"].numel())
    # (the customer table has no build output channels, so the orders launch skips the rank word: 8 B per pair there)
    alg = ((4.0 * n_o + 16.0 * st["orders_probe_rows"] + 8.0 * st["orders_build_rows"]) + (4.0 * n_l + 16.0 * st["lineitem_probe_rows"] + 12.0 * st["lineitem_join_rows"])) / 2.0
    rows_avg = (n_o + n_l) / 2.0
    if repartition:
        # behind the exchange the probe is the unfused kernel: key 8 B + one table slot 12 B + head/count out 8 B per probe row
        rows_avg = (st["orders_probe_rows"] + st["lineitem_probe_rows"]) / 2.0
        # behind the exchange the probes are the fused kernels without a filter: key 8 B + bitmap word 8 B per probe row + 12 B per pair (DIRECT layout)
        roof = dominant(prof, args.steps, {"fused_filter_probe": 2.0 * rows_avg, "join_probe_count": 2.0 * rows_avg}, {"fused_filter_probe": 16.0 + 12.0 * (st["orders_build_rows"] + st["lineitem_join_rows"]) / max(2.0 * rows_avg, 1.0), "join_probe_count": 28.0})
    else:
        roof = dominant(prof, args.steps, {"fused_filter_probe": 2.0 * rows_avg}, {"fused_filter_probe": alg / rows_avg})   # two launches per step (orders, lineitem)
        if roof and roof["kernel"] == "fused_filter_probe" and "fused_filter_probe" in prof:
            # the two launches of a step are different animals: the orders launch probes random keys (bound by
            # divergent L2 lookups), the lineitem launch streams.  Their own figures, from the shortest (orders) and the longest
            # (lineitem) launch of the timed region -- conservative for the lineitem launch:
            fp = prof["fused_filter_probe"]
            alg_o = 4.0 * n_o + 16.0 * st["orders_probe_rows"] + 8.0 * st["orders_build_rows"]
            alg_l = 4.0 * n_l + 16.0 * st["lineitem_probe_rows"] + 12.0 * st["lineitem_join_rows"]
            roof["per_launch"] = {
                "orders": {"algorithmic_bytes": alg_o, "launch_ms": fp["min_ms"], "achieved": alg_o / (fp["min_ms"] * 1e-3) / 1e9, "frac": alg_o / (fp["min_ms"] * 1e-3) / 8e12},
                "lineitem": {"algorithmic_bytes": alg_l, "launch_ms": fp["max_ms"], "achieved": alg_l / (fp["max_ms"] * 1e-3) / 1e9, "frac": alg_l / (fp["max_ms"] * 1e-3) / 8e12}}
    out.update({
        "metric": "probe_rows_per_sec", "value": total_probe / step_s, "unit": "rows/s", "n_gpus": b.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
        "config": {"workload": "tpch_q3_hash_join_build_probe_agg (BASELINE configs[3])", "scale_factor_per_gpu": args.sf, "seed": SEED, "device_input_stable": True,
                   "lineitem_rows": int(b.q3["l_orderkey"].numel()), "orders_rows": int(b.q3["o_orderkey"].numel()), "customer_rows": int(b.q3["c_custkey"].numel()),
                   "lineitem_probe_rows": probe_rows, "orders_build_rows": st["orders_build_rows"], "join_output_rows": st["lineitem_join_rows"],
                   "groups": st["groups"], "parallelism": ("single GPU" if not distributed else f"hash-repartitioned joins x{b.world} (K10 partition + RCCL all-to-all-v)" if repartition else
                                   f"x{b.world}: customer replicated (RCCL all-gather), orders/lineitem co-partitioned on orderkey, partial->final aggregation over a hash exchange (K10 partition + RCCL all-to-all-v)"
                                   if os.environ.get("TGPU_BENCH_PLAN") == "partial_final" else
                                   f"x{b.world}: customer replicated (RCCL all-gather), orders/lineitem and the aggregation co-partitioned on orderkey (no further exchange)"),
                   "exchange_bytes_sent_per_step": st.get("exchange_bytes_sent", 0)},
        "roofline": roof, "checks": {"q3": q3_check},
    })
    if distributed:
        # Extras of the N-rank line (Q3's top-10 tail, the repartition plan): the headline fields above are complete; should an extra raise,
        # the line still goes out with what is there; a collective in an extra that never returns ends the process with exit code 3 after the
        # watchdog has printed the line marked `extras_timed_out` (a hang is a failure, not a result).
        import threading

        class ExtrasWatchdog:
            """armed around EACH extra: a collective that never returns is a failure -- the line goes out with what is there, marked, and the
            process exits NON-ZERO (a hang must not look like a clean run)"""

            def __init__(self):
                self.timer, self.what = None, None
                self.seconds = float(os.environ.get("TGPU_BENCH_EXTRAS_TIMEOUT", "300"))

            def fire(self):
                if b.rank == 0:
                    print(json.dumps({**out, **extra, "extras_timed_out": True, "extras_timed_out_in": self.what}), flush=True)
                os._exit(3)

            def arm(self, what):
                self.cancel()
                self.what = what
                self.timer = threading.Timer(self.seconds, self.fire)
                self.timer.daemon = True
                self.timer.start()

            def cancel(self):
                if self.timer is not None:
                    self.timer.cancel()
                    self.timer = None

        watchdog = ExtrasWatchdog()
        b.extras_watchdog = watchdog
        try:
            if distributed and b.q3_result:
                watchdog.arm("q3_top10")
                extra["q3_top10"] = b.q3_top10_dist(args.steps, args.warmup)
                watchdog.cancel()
            if distributed and not repartition and os.environ.get("TGPU_BENCH_PLAN") is None:
                # the alternative plan in the same line: every join input hash-repartitioned over xGMI (the all-to-all BASELINE.json's metric
                # names); `value` stays the plan the optimizer picks (co-partitioned joins), this object says what the exchange-heavy plan costs
                for o in (b.q3_result or []):
                    o.release()
                b.q3_result = None
                keep = dict(b.q3_stats)
                watchdog.arm("repartition_plan")
                s_rp, prof_rp = b.timed(b.step_q3_dist_repartition, args.steps, args.warmup)
                chk_rp = b.check_q3_dist_repartition()
                st_rp = dict(b.q3_stats)
                tt = torch.tensor([st_rp["lineitem_probe_rows"], st_rp.get("exchange_bytes_sent", 0)], device=b.coll_dev, dtype=torch.int64)
                b.dist.all_reduce(tt)
                ex_ms = prof_rp.get("exchange_all_to_all_v", {"total_ms": 0.0})["total_ms"] / args.steps
                extra["repartition_plan"] = {
                    "plan": f"hash-repartitioned joins x{b.world}: K10 partition kernels + grouped RCCL send/recv all-to-all-v of every join input (tgpu_exchange_repartition)",
                    "ms_per_step": s_rp * 1e3, "probe_rows_per_sec": int(tt[0].item()) / s_rp, "exchange_bytes_sent_per_step_all_ranks": int(tt[1].item()),
                    "exchange_bytes_sent_per_step_this_rank": st_rp.get("exchange_bytes_sent", 0), "exchange_ms_per_step_this_rank": ex_ms,
                    "achieved_xgmi_GBps_this_rank": (st_rp.get("exchange_bytes_sent", 0) / (ex_ms * 1e-3) / 1e9) if ex_ms > 0 else None,
                    "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in sorted(prof_rp.items(), key=lambda kv: -kv[1]["total_ms"])[:10]}, "check": chk_rp}
                for o in (b.q3_result or []):
                    o.release()
                b.q3_result = None
                b.q3_stats = keep
                watchdog.cancel()
        except Exception as e:   # (every rank runs the same code on the same schedule: they fail together)
            extra["extras_error"] = repr(e)
        watchdog.cancel()
    if not distributed and b.q3_result:
        extra["q3_top10"] = b.q3_top10(args.steps, args.warmup)

    if not distributed:
        for o in (b.q3_result or []):
            o.release()
        b.q3_result = None
        extra["q3_step_stats"] = b.step_stats(b.step_q3, min(args.steps, 20))
    extra["q3_readbacks_per_step"] = q3_readbacks   # host <- device round trips (stream waits) of one Q3 step
    extra["q3_kernels_ms_per_step"] = {k: v["total_ms"] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
    extra["q3_kernel_launch_min_max_ms"] = {k: [v["min_ms"], v["max_ms"], v["count"]] for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}
    if b.world == 1 and "paged" in only:
        # the engine hands over pages, not tables: the same three-table pipeline fed as 2^20- and 2^24-row pages
        extra["q3_paged"] = {"direct_2^20": b.q3_paged(args.steps, args.warmup, 1 << 20), "merged_2^20_16MB": b.q3_paged(args.steps, args.warmup, 1 << 20, merge_mb=16),
                             "direct_2^24": b.q3_paged(args.steps, args.warmup, 1 << 24)}
        out["checks"]["q3_paged"] = all(v["ok"] for v in extra["q3_paged"].values())
    for o in (b.q3_result or []):
        o.release()
    b.q3_result = None
    del b.q3, b.q3_pages
    torch.cuda.empty_cache()

    if "q1" in only and b.world > 1:
        try:
            b.extras_watchdog.arm("q1_distributed")
            # Q1 on N ranks: partial aggregation of every rank's shard -> all-gather of the 4-row partial pages -> final combine (8e step 3)
            n = int(6_000_379.02 * args.sf)
            b.setup_q1(n)
            b.setup_q1_dist()
            s1, p1 = b.timed(b.step_q1_dist, args.steps, args.warmup)
            out["q1"] = {"metric": "input_rows_per_sec", "value": n * b.world / s1, "unit": "rows/s", "ms_per_step": s1 * 1e3, "rows_per_rank": n, "scaling": "weak",
                         "workload": "tpch_q1_filter_project_hash_aggregation (BASELINE configs[2]), one SF%g shard per rank" % args.sf,
                         "plan": f"x{b.world}: fused PARTIAL aggregation per rank -> all-gather of the partial pages (RCCL, rank order) -> FINAL combine on every rank",
                         "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in sorted(p1.items(), key=lambda kv: -kv[1]["total_ms"])[:6]}}
            out["checks"]["q1"] = b.check_q1_dist()
            del b.q1, b.q1_page_
            b.q1_result = None
            torch.cuda.empty_cache()
        except Exception as e:   # an extra of the N-rank line: the headline goes out whatever happens here
            out["q1"] = {"error": repr(e)}
        b.extras_watchdog.cancel()
    elif "q1" in only:
        n = int(6_000_379.02 * args.sf)
        b.setup_q1(n)
        s1, p1 = b.timed(b.step_q1, args.steps, args.warmup)
        out["q1"] = {"metric": "input_rows_per_sec", "value": n / s1, "unit": "rows/s", "ms_per_step": s1 * 1e3, "rows": n,
                     "workload": "tpch_q1_filter_project_hash_aggregation (BASELINE configs[2])",
                     "algorithmic_bytes_per_row": 46.0, "achieved_gbps_whole_step": 46.0 * n / s1 / 1e9, "frac_of_8TBps": 46.0 * n / s1 / 8e12,
                     "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in sorted(p1.items(), key=lambda kv: -kv[1]["total_ms"])}}
        # roofline of Q1's dominant kernel: the fused project+accumulate pass reads the group id (one byte per row in the
        # low-cardinality mode Q1 runs in, int32 otherwise) + four 8-byte inputs per row
        # (the table-sized page: three leading slices through probe + accumulate, the rest -- n - 1.5 M rows -- in ONE one-pass launch that reads
        # every input byte once: 4 + 2 x (4 + 1) + 32 = 46 B per row; TGPU_DISABLE_PAGE_SPLIT: the two-launch path, 1 + 32 B per row in the accumulate)
        out["q1"]["roofline"] = dominant(p1, args.steps, {"fused_filter_group_accumulate_onepass": n - 1_572_864, "fused_project_accumulate_lowcard": n, "fused_project_accumulate": n},
                                         {"fused_filter_group_accumulate_onepass": 46.0, "fused_project_accumulate_lowcard": 33.0, "fused_project_accumulate": 36.0})
        out["checks"]["q1"] = b.check_q1(java_order_distance=(b.world == 1 and not args.no_cpu_baseline))
        out["q1"]["step_stats"] = b.step_stats(b.step_q1, min(args.steps, 20))
        out["q1"]["double_sum_order"] = "EXACT"
        if b.world == 1:
            strict = b.q1_java_order_at_line_size(n)
            out["q1"]["double_sum_modes"] = b.q1_double_sum_modes(args.steps, args.warmup)
            out["q1"]["double_sum_modes"]["java_order_at_line_size"] = strict
            if strict.get("sum_extendedprice_bit_identical_to_java_order") is False:
                out["checks"]["q1"]["ok"] = False
        if b.world == 1 and "paged" in only:
            # the engine hands over pages, not tables: the same program fed as 2^20-row pages (573 of them at SF100) directly and through MergePages
            out["q1"]["paged"] = {"direct_2^20": b.q1_paged(args.steps, args.warmup, 1 << 20), "merged_2^20_1GB": b.q1_paged(args.steps, args.warmup, 1 << 20, merge_mb=1024),
                                  "direct_2^24": b.q1_paged(args.steps, args.warmup, 1 << 24)}
            out["checks"]["q1_paged"] = all(v["ok"] for v in out["q1"]["paged"].values())
        del b.q1, b.q1_page_
        b.q1_result = None
        torch.cuda.empty_cache()

    if "cfg2" in only:
        n2 = int(1_000_000 * args.sf)
        b.setup_cfg2(n2)
        s2, p2 = b.timed(b.step_cfg2, args.steps, args.warmup)
        out["cfg2"] = {"metric": "input_rows_per_sec", "value": n2 / s2, "unit": "rows/s", "ms_per_step": s2 * 1e3, "rows": n2,
                       "workload": "bigint_filter_project_sel10 (BASELINE configs[1])", "algorithmic_bytes_per_row": 10.4,
                       "achieved_gbps_whole_step": 10.4 * n2 / s2 / 1e9, "frac_of_8TBps": 10.4 * n2 / s2 / 8e12,
                       # SURVEY.md 8d's second bound: at 10 % selectivity nearly every 128-byte line of col1 / col2 holds a selected row, so
                       # the memory system moves the full columns (8 + 16 read + 0.8 written = 24.8 B/row)
                       "full_column_bytes_per_row": 24.8, "achieved_gbps_full_column_bound": 24.8 * n2 / s2 / 1e9, "frac_of_8TBps_full_column_bound": 24.8 * n2 / s2 / 8e12,
                       "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in sorted(p2.items(), key=lambda kv: -kv[1]["total_ms"])}}
        # pass 1 streams the 8-byte filter column of every row
        out["cfg2"]["roofline"] = dominant(p2, args.steps, {"filter_count": n2}, {"filter_count": 8.0})
        out["checks"]["cfg2"] = b.check_cfg2()
        del b.c2, b.c2_page
        b.c2_out = None
        torch.cuda.empty_cache()

    if b.world == 1 and "paged" in only and "q1" in only:
        # PCIe-inclusive Q1 on a bounded sample (every input byte host -> HBM inside the timed region; never `value`)
        n_pcie = int(6_000_379.02 * min(args.sf, 20.0))
        b.setup_q1(1024)   # (re)creates the factory
        out["q1"]["pcie_inclusive"] = {"pinned": b.q1_pcie_inclusive(args.steps, args.warmup, n_pcie, 1 << 24, True),
                                       "pageable": b.q1_pcie_inclusive(args.steps, args.warmup, n_pcie, 1 << 24, False)}
        del b.q1, b.q1_page_
    if b.world == 1 and "sub" in only:
        torch.cuda.empty_cache()
        sub = {}
        sub["join_hash_layout"] = sub_join_hash_layout(b, args.steps, args.warmup, args.sf)
        torch.cuda.empty_cache()
        sub["join_duplicate_keys"] = sub_join_duplicate_keys(b, args.steps, args.warmup, args.sf)
        torch.cuda.empty_cache()
        sub["group_by_hash_10M_3M"] = sub_group_by_hash(b, args.steps, args.warmup, int(100_000 * args.sf), int(30_000 * args.sf))
        sub["group_by_hash_100M_40M"] = sub_group_by_hash(b, args.steps, args.warmup, int(1_000_000 * args.sf), int(400_000 * args.sf))
        torch.cuda.empty_cache()
        out["sub_benchmarks"] = sub
        out["checks"]["sub_benchmarks"] = {k: v["ok"] for k, v in sub.items()}
    if b.rank == 0 and b.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(b, args.cpu_sample_sf)
    out.update(extra)
    if getattr(b, "extras_watchdog", None) is not None:
        b.extras_watchdog.cancel()
    if b.rank == 0:
        print(json.dumps(out))
    if b.dist is not None:
        b.dist.barrier()
        b.dist.destroy_process_group()


if __name__ == "__main__":
    main()
