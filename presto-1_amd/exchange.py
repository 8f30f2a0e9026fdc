"""Hash repartition exchange across GPUs: the reference's PartitionedOutputOperator -> HTTP page exchange ->
ExchangeOperator hop (M/operator/PartitionedOutputOperator.java:406-476, HttpPageBufferClient.java, ExchangeOperator.java)
collapsed into

    1. K10 partition kernel (tgpu_partition_page): rows grouped by destination = (rawHash & 0x7fff...) % world
       (M/operator/HashGenerator.java:24-35), input order kept inside each destination;
    2. one all-to-all of the row counts, then one all-to-all-v per column buffer (RCCL over xGMI through torch.distributed's
       `nccl` backend; every rank talks to its 7 peers at once, one xGMI link per pair).

Rows never leave HBM.  One process per GPU; `torch.distributed` must be initialised by the caller.  The partitioner is
injectable so that the CPU-only test-suite can drive the same exchange code over `gloo` (tests inject a reference
partitioner); the default -- and the only one the product ever uses -- is the HIP kernel behind the C ABI.
"""
import numpy as np
import torch

from .spi import BIGINT, BOOLEAN, DATE, DOUBLE, INTEGER, VARCHAR, DeviceBlock, Page

TORCH_DTYPE = {BIGINT: torch.int64, INTEGER: torch.int32, DATE: torch.int32, DOUBLE: torch.float64, BOOLEAN: torch.uint8}
_TYPESTR = {torch.int64: "<i8", torch.int32: "<i4", torch.float64: "<f8", torch.uint8: "|u1"}


class _DevArray:
    """a raw device pointer exposed through __cuda_array_interface__ (zero-copy view for torch); keeps its owner alive"""

    def __init__(self, ptr, n, dtype, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": _TYPESTR[dtype], "data": (int(ptr), False), "version": 3}


def device_view(ptr, n, dtype, owner, device):
    if n == 0 or not ptr:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DevArray(ptr, n, dtype, owner), device=device)


def hip_partitioner(ctx, device):
    """the product partitioner: K10 on the GPU.  Returns (counts[world] numpy, columns) where each column is a dict of torch
    tensors {type, values, nulls|None, offsets|None} holding the rows grouped by destination."""

    def run(page: Page, key_channels, world):
        counts, out = ctx.partition_page(page, key_channels, world)
        dp = out.as_device_page()
        cols = []
        n = dp.position_count
        for b in dp.blocks:
            if b.type == VARCHAR:
                offsets = device_view(b.offsets, n + 1, torch.int32, out, device)
                nbytes = int(offsets[-1].item()) if n else 0
                values = device_view(b.values, nbytes, torch.uint8, out, device)
            else:
                offsets = None
                values = device_view(b.values, n, TORCH_DTYPE[b.type], out, device)
            nulls = device_view(b.nulls, n, torch.uint8, out, device) if b.nulls else None
            cols.append({"type": b.type, "values": values, "nulls": nulls, "offsets": offsets})
        return counts, cols

    return run


def page_columns(page: Page, device):
    """torch views of a device Page's fixed-width columns: [{type, values, nulls|None}] (VARCHAR is not handled here)"""
    cols = []
    n = page.position_count
    for b in page.blocks:
        if b.type == VARCHAR:
            raise NotImplementedError("all_gather_page replicates fixed-width columns only")
        values = b.values if hasattr(b.values, "data_ptr") else device_view(b.values, n, TORCH_DTYPE[b.type], page, device)
        nulls = None
        if hasattr(b.nulls, "data_ptr"):
            nulls = b.nulls
        elif b.nulls:   # a raw device address (None / 0 = no null vector)
            nulls = device_view(b.nulls, n, torch.uint8, page, device)
        cols.append({"type": b.type, "values": values, "nulls": nulls})
    return cols


def all_gather_page(dist, device, page: Page) -> Page:
    """REPLICATED distribution of a (small) build side: every rank receives the rows of all ranks, concatenated in rank order --
    the broadcast exchange in front of a replicated join (M/sql/planner/SystemPartitioningHandle.java:59 FIXED_BROADCAST_DISTRIBUTION).
    One all-gather of the row counts, then one all-gather per column buffer (RCCL over xGMI), padded to the largest rank."""
    w = dist.get_world_size()
    host_staging = dist.get_backend() == "gloo" and torch.device(device).type == "cuda"
    coll = torch.device("cpu") if host_staging else torch.device(device)
    n = int(page.position_count)
    counts = [torch.zeros(1, dtype=torch.int64, device=coll) for _ in range(w)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=coll))
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    total = sum(counts)

    def gather(t, dtype):
        pad = torch.zeros(m, dtype=dtype, device=coll)
        if n:
            pad[:n] = t.to(coll) if host_staging else t
        outs = [torch.empty(m, dtype=dtype, device=coll) for _ in range(w)]
        dist.all_gather(outs, pad)
        parts = [outs[r][: counts[r]] for r in range(w)]
        res = torch.cat(parts) if total else torch.zeros(1, dtype=dtype, device=coll)
        return res.to(device) if host_staging else res

    cols = page_columns(page, device)
    blocks, keep = [], []
    # which channels carry a null vector on ANY rank: one collective for all channels (not one per channel)
    any_nulls = torch.tensor([1 if c["nulls"] is not None else 0 for c in cols] or [0], device=coll)
    dist.all_reduce(any_nulls, op=dist.ReduceOp.MAX)
    any_nulls = any_nulls.tolist()
    for i, c in enumerate(cols):
        nulls = None
        if any_nulls[i]:
            nulls = gather(c["nulls"] if c["nulls"] is not None else torch.zeros(n, dtype=torch.uint8, device=device), torch.uint8)
        values = gather(c["values"], TORCH_DTYPE[c["type"]])
        blocks.append(DeviceBlock(c["type"], total, values, nulls))
        keep += [values, nulls]
    out = Page(*blocks, position_count=total)
    out._keep = keep
    return out


class HashExchange:
    def __init__(self, dist, device, partitioner):
        self.dist = dist
        self.device = device
        self.partitioner = partitioner
        self.world = dist.get_world_size()
        self.bytes_sent = 0
        # rehearsal only (several ranks sharing one GPU over `gloo`): gloo moves host tensors, so stage through the host
        self.host_staging = dist.get_backend() == "gloo" and torch.device(device).type == "cuda"
        self.coll_device = torch.device("cpu") if self.host_staging else torch.device(device)

    def _a2a(self, send, send_splits, recv_splits):
        if self.host_staging:
            send = send.cpu()
        recv = torch.empty(int(sum(recv_splits)), dtype=send.dtype, device=send.device)
        self.dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=[int(x) for x in recv_splits], input_split_sizes=[int(x) for x in send_splits])
        self.bytes_sent += send.numel() * send.element_size()
        return recv.to(self.device) if self.host_staging else recv

    def exchange(self, page: Page, key_channels):
        """repartition `page` by the hash of `key_channels`; returns the rows this rank owns as a device (or host-tensor) Page"""
        w = self.world
        counts, cols = self.partitioner(page, key_channels, w)
        # one small all-to-all carries, per destination, the row count AND this rank's per-channel "has a null vector" flags (one
        # collective + one read-back for the whole page header instead of one all-reduce per channel)
        flags = [1 if c["nulls"] is not None else 0 for c in cols]
        header = torch.as_tensor(np.asarray([[int(x)] + flags for x in counts], dtype=np.int64).reshape(-1), device=self.coll_device)
        recv_header = torch.empty_like(header)
        self.dist.all_to_all_single(recv_header, header)
        recv_header = recv_header.reshape(w, 1 + len(cols)).tolist()
        sc = [int(x) for x in counts]
        rc = [int(r[0]) for r in recv_header]
        # a channel travels with nulls when any SENDER has them; every rank must agree, so reduce over what all ranks told everyone:
        # each rank told every destination its own flags, hence each rank now knows the flags of all ranks
        any_nulls = [max(int(r[1 + i]) for r in recv_header) for i in range(len(cols))]
        n_out = sum(rc)
        blocks, keep = [], []
        row_starts = np.concatenate([[0], np.cumsum(sc)])
        for ci, c in enumerate(cols):
            nulls = None
            if any_nulls[ci]:
                send_nulls = c["nulls"] if c["nulls"] is not None else torch.zeros(sum(sc), dtype=torch.uint8, device=self.device)
                nulls = self._a2a(send_nulls, sc, rc)
            if c["type"] == VARCHAR:
                off = c["offsets"].to(torch.int64)
                lens = (off[1:] - off[:-1]).to(torch.int32)
                recv_lens = self._a2a(lens, sc, rc)
                seg = off[torch.as_tensor(row_starts, device=off.device)]
                send_bytes = [int(x) for x in (seg[1:] - seg[:-1]).tolist()]
                sb = torch.as_tensor(np.asarray(send_bytes, dtype=np.int64), device=self.coll_device)
                rb = torch.empty(w, dtype=torch.int64, device=self.coll_device)
                self.dist.all_to_all_single(rb, sb)
                values = self._a2a(c["values"][: sum(send_bytes)], send_bytes, [int(x) for x in rb.tolist()])
                offsets = torch.zeros(n_out + 1, dtype=torch.int32, device=self.device)
                if n_out:
                    offsets[1:] = torch.cumsum(recv_lens.to(torch.int64), 0).to(torch.int32)
                if values.numel() == 0:
                    values = torch.zeros(1, dtype=torch.uint8, device=self.device)
                blocks.append(DeviceBlock(VARCHAR, n_out, values, nulls, offsets))
                keep += [values, offsets, nulls]
            else:
                values = self._a2a(c["values"], sc, rc)
                if values.numel() == 0:
                    values = torch.zeros(1, dtype=values.dtype, device=self.device)
                blocks.append(DeviceBlock(c["type"], n_out, values, nulls))
                keep += [values, nulls]
        out = Page(*blocks, position_count=n_out)
        out._keep = keep
        return out
