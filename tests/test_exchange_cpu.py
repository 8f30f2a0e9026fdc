"""World-size-2 `gloo` test of the repartition exchange (presto-1_amd/exchange.py) on CPU.  The exchange code is the product's;
the partitioner is injected (the CPU oracle's restatement of HashGenerator.getPartition), because the product's own partitioner
is the HIP kernel.  Checks: every row lands on the rank that owns its hash partition, nothing is lost or duplicated, rows of one
source keep their order, VARCHAR / null columns survive the all-to-all-v."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rng(seed):
    """the same stream in the parent and in the spawned ranks (conftest's TGPU_TEST_SEED_OFFSET patch lives in the parent only)"""
    return np.random.Generator(np.random.PCG64(seed))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_rows(rank, n):
    rng = _rng(100 + rank)
    keys = rng.integers(0, 5000, n).astype(np.int64)
    pay = (np.arange(n, dtype=np.int64) + rank * 1_000_000)
    strs = [None if k % 11 == 0 else "s%d" % (k % 37) for k in keys]
    dbl = rng.standard_normal(n)
    dnull = (rng.random(n) < 0.1).astype(np.uint8) if rank == 0 else np.zeros(n, dtype=np.uint8)   # rank 1 sends this channel WITHOUT a null vector
    return keys, pay, strs, dbl, dnull


def _worker(rank, world, port, n, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("presto-1_amd")
    from oracle import oracle
    ex_mod = importlib.import_module("presto-1_amd.exchange")
    keys, pay, strs, dbl, dnull = _make_rows(rank, n)

    def oracle_partitioner(page, key_channels, w):
        kcol = oracle.Col(oracle.BIGINT, keys)
        pid = oracle.partition_remote(oracle.hash_rows([kcol]), w)
        order = np.argsort(pid, kind="stable")
        counts = np.bincount(pid, minlength=w).astype(np.int64)
        sb = pkg.Block(pkg.VARCHAR, [strs[i] for i in order])
        cols = [
            {"type": pkg.BIGINT, "values": torch.from_numpy(keys[order]), "nulls": None, "offsets": None},
            {"type": pkg.BIGINT, "values": torch.from_numpy(pay[order]), "nulls": None, "offsets": None},
            {"type": pkg.VARCHAR, "values": torch.from_numpy(sb.values.copy()), "nulls": torch.from_numpy(sb.nulls.copy()) if sb.nulls is not None else None,
             "offsets": torch.from_numpy(sb.offsets.copy())},
            # (a channel travels with nulls when ANY rank has a null vector for it: the page header all-to-all carries the flags)
            {"type": pkg.DOUBLE, "values": torch.from_numpy(dbl[order]), "nulls": torch.from_numpy(dnull[order]) if rank == 0 else None, "offsets": None},
        ]
        return counts, cols

    ex = ex_mod.HashExchange(dist, torch.device("cpu"), oracle_partitioner)
    out = ex.exchange(None, [0])
    b = out.blocks
    n_out = out.position_count
    got_keys = b[0].values[:n_out].numpy().copy()
    got_pay = b[1].values[:n_out].numpy().copy()
    off = b[2].offsets.numpy()
    raw = b[2].values.numpy().tobytes()
    snull = b[2].nulls.numpy() if b[2].nulls is not None else np.zeros(n_out, dtype=np.uint8)
    got_strs = [None if snull[i] else raw[off[i]:off[i + 1]].decode() for i in range(n_out)]
    got_dbl = b[3].values[:n_out].numpy().copy()
    got_dnull = b[3].nulls[:n_out].numpy().copy()
    pid = oracle.partition_remote(oracle.hash_rows([oracle.Col(oracle.BIGINT, got_keys)]), world) if n_out else np.zeros(0, dtype=np.int32)
    out_q.put((rank, got_keys, got_pay, got_strs, got_dbl, got_dnull, bool((pid == rank).all()), ex.bytes_sent))
    dist.barrier()
    dist.destroy_process_group()


def test_hash_exchange_world2_gloo(oracle):
    world, n = 2, 3000
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sent = []
    for r in range(world):
        keys, pay, strs, dbl, dnull = _make_rows(r, n)
        sent += list(zip(keys.tolist(), pay.tolist(), strs, dbl.tolist(), dnull.tolist()))
    recv = []
    for rank, k, p, s, d, dn, owned, nbytes in results:
        assert owned, "a row landed on a rank that does not own its partition"
        assert nbytes > 0
        recv += list(zip(k.tolist(), p.tolist(), s, d.tolist(), dn.tolist()))
        # rows from one source rank keep their input order (payload is monotone per source)
        for src in range(world):
            mine = [x for x in p.tolist() if x // 1_000_000 == src]
            assert mine == sorted(mine)
    assert sorted(recv, key=lambda t: t[1]) == sorted(sent, key=lambda t: t[1])


def _gather_rows(rank):
    n = [0, 2500, 700][rank]          # ragged, and one rank contributes nothing
    rng = _rng(200 + rank)
    keys = rng.integers(0, 1 << 40, n).astype(np.int64)
    dates = rng.integers(8000, 10000, n).astype(np.int32)
    dnull = (rng.random(n) < 0.2).astype(np.uint8) if rank == 1 else None   # nulls on one rank only
    return keys, dates, dnull


def _gather_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("presto-1_amd")
    ex_mod = importlib.import_module("presto-1_amd.exchange")
    keys, dates, dnull = _gather_rows(rank)
    page = pkg.Page(pkg.DeviceBlock(pkg.BIGINT, len(keys), torch.from_numpy(keys)),
                    pkg.DeviceBlock(pkg.DATE, len(dates), torch.from_numpy(dates), torch.from_numpy(dnull) if dnull is not None else None),
                    position_count=len(keys))
    out = ex_mod.all_gather_page(dist, torch.device("cpu"), page)
    n = out.position_count
    b = out.blocks
    out_q.put((rank, n, b[0].values[:n].numpy().copy(), b[1].values[:n].numpy().copy(), None if b[1].nulls is None else b[1].nulls[:n].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_page_world3_gloo():
    """the replicated (broadcast-join) distribution: every rank ends up with all rows in rank order; ragged and empty contributions,
    a null vector present on one rank only"""
    world = 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = [_gather_rows(r) for r in range(world)]
    want_keys = np.concatenate([p[0] for p in parts])
    want_dates = np.concatenate([p[1] for p in parts])
    want_nulls = np.concatenate([p[2] if p[2] is not None else np.zeros(len(p[0]), dtype=np.uint8) for p in parts])
    for rank, n, keys, dates, nulls in results:
        assert n == len(want_keys)
        assert np.array_equal(keys, want_keys)
        assert nulls is not None and np.array_equal(nulls, want_nulls)
        assert np.array_equal(dates[want_nulls == 0], want_dates[want_nulls == 0])
