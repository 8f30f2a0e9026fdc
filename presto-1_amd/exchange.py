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
        send_counts = torch.as_tensor(np.asarray(counts, dtype=np.int64), device=self.coll_device)
        recv_counts = torch.empty(w, dtype=torch.int64, device=self.coll_device)
        self.dist.all_to_all_single(recv_counts, send_counts)
        sc = [int(x) for x in counts]
        rc = [int(x) for x in recv_counts.tolist()]
        n_out = sum(rc)
        blocks, keep = [], []
        row_starts = np.concatenate([[0], np.cumsum(sc)])
        for c in cols:
            any_nulls = torch.tensor([1 if c["nulls"] is not None else 0], device=self.coll_device)
            self.dist.all_reduce(any_nulls, op=self.dist.ReduceOp.MAX)
            nulls = None
            if int(any_nulls.item()):
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
