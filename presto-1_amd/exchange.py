"""Exchange between the GPUs of one node: ctypes mirror of tgpu_exchange_* (include/tgpu.h).

The exchange itself -- K10 partition kernels, page-header all-to-all, one grouped RCCL ncclSend / ncclRecv all-to-all-v per page over
xGMI, VARCHAR offset rebasing -- lives in libtgpu.so (csrc/exchange.hip); nothing here touches the rows.  What this module adds is
bootstrap plumbing for Python hosts:

  * `Exchange.over_rccl(ctx, rank, world, broadcast)`: rank 0 makes the RCCL unique id, `broadcast(bytes) -> bytes` hands it to the
    other ranks (bench.py uses torch.distributed for that; a JVM host would ship it with the task's exchange locations);
  * `GlooTransport`: the library's transport callbacks (tgpu_exchange_transport) over torch.distributed's `gloo` backend with host
    staging -- for rehearsing several ranks on ONE GPU (RCCL refuses two ranks on one device) and for CPU tests of the transport.
"""
import ctypes as C

import numpy as np

from . import _lib
from .spi import OutputPage, Page

ID_BYTES = 128


class Exchange:
    def __init__(self, handle, keep=None):
        self.handle = handle
        self._keep = keep

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(ID_BYTES)
        _lib.check(_lib.lib().tgpu_exchange_unique_id(buf))
        return buf.raw

    @classmethod
    def over_rccl(cls, ctx, rank, world, broadcast):
        """collective: every rank calls it; `broadcast(payload_or_None) -> bytes` returns rank 0's payload on every rank"""
        uid = broadcast(cls.unique_id() if rank == 0 else None)
        assert len(uid) == ID_BYTES
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_exchange_create(ctx.handle, uid, rank, world, C.byref(h)))
        return cls(h)

    @classmethod
    def over_transport(cls, ctx, rank, world, transport):
        h = C.c_void_p()
        _lib.check(_lib.lib().tgpu_exchange_create_with_transport(ctx.handle, rank, world, C.byref(transport.struct), C.byref(h)))
        return cls(h, keep=transport)

    def repartition(self, page: Page, key_channels, hash_channel=-1) -> OutputPage:
        """FIXED_HASH_DISTRIBUTION: the rows of `page` go to rank (rawHash & 0x7fff...) % world; returns this rank's rows"""
        ch = (C.c_int32 * max(1, len(key_channels)))(*key_channels)
        cp, keep = page.to_c()
        out = C.c_void_p()
        _lib.check(_lib.lib().tgpu_exchange_repartition(self.handle, C.byref(cp), len(key_channels), ch, hash_channel, C.byref(out)))
        return OutputPage(out)

    def partitioned_output(self, operator, types) -> OutputPage:
        """shuffles what a PartitionedOutputOperator (partition_count == world) has pending after one addInput"""
        t = (C.c_int32 * max(1, len(types)))(*types)
        out = C.c_void_p()
        _lib.check(_lib.lib().tgpu_exchange_partitioned_output(self.handle, operator.handle, len(types), t, C.byref(out)))
        return OutputPage(out)

    def all_gather(self, page: Page) -> OutputPage:
        """FIXED_BROADCAST_DISTRIBUTION: every rank's rows on every rank, in rank order"""
        cp, keep = page.to_c()
        out = C.c_void_p()
        _lib.check(_lib.lib().tgpu_exchange_all_gather(self.handle, C.byref(cp), C.byref(out)))
        return OutputPage(out)

    @property
    def bytes_sent(self):
        return int(_lib.lib().tgpu_exchange_bytes_sent(self.handle))

    def close(self):
        if self.handle:
            _lib.lib().tgpu_exchange_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown
            pass


class HostMemory:
    """buffers addressed by the callbacks are host memory (CPU tests of the transport)"""

    @staticmethod
    def read(ptr, nbytes):
        return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)).copy()

    @staticmethod
    def write(ptr, data):
        C.memmove(ptr, data.ctypes.data, data.nbytes)


class TorchDeviceMemory:
    """buffers addressed by the callbacks are device memory: staged through the host with torch (rehearsal on one GPU)"""

    def __init__(self, device):
        import torch
        self.torch, self.device = torch, device

    def _view(self, ptr, nbytes):
        class _Dev:
            __cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}
        return self.torch.as_tensor(_Dev(), device=self.device)

    def read(self, ptr, nbytes):
        return self._view(ptr, nbytes).cpu().numpy()

    def write(self, ptr, data):
        self._view(ptr, data.nbytes).copy_(self.torch.from_numpy(data))
        self.torch.cuda.synchronize(self.device)


class GlooTransport:
    """tgpu_exchange_transport over torch.distributed (any backend that moves host tensors, i.e. gloo)"""

    def __init__(self, dist, memory):
        import torch
        self.torch, self.dist, self.memory = torch, dist, memory
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.error = None
        self._meta = _lib.TRANSPORT_META_FN(self._all_to_all_meta)
        self._v = _lib.TRANSPORT_V_FN(self._all_to_all_v)
        self.struct = _lib.ExchangeTransport(None, self._meta, self._v)

    def _all_to_all_meta(self, user, send, recv, per_rank):
        try:
            n = self.world * per_rank
            s = self.torch.from_numpy(np.ctypeslib.as_array(send, shape=(n,)).copy())
            r = self.torch.empty_like(s)
            self.dist.all_to_all_single(r, s)
            np.ctypeslib.as_array(recv, shape=(n,))[:] = r.numpy()
            return 0
        except Exception as e:   # an exception must not unwind through the C frames
            self.error = e
            return -1

    def _all_to_all_v(self, user, transfers, send_ptr, send_bytes, recv_ptr, recv_bytes):
        try:
            w = self.world
            for t in range(transfers):
                sb = [int(send_bytes[t * w + r]) for r in range(w)]
                rb = [int(recv_bytes[t * w + r]) for r in range(w)]
                parts = [self.memory.read(send_ptr[t * w + r], sb[r]) if sb[r] else np.zeros(0, dtype=np.uint8) for r in range(w)]
                send = self.torch.from_numpy(np.concatenate(parts)) if sum(sb) else self.torch.zeros(0, dtype=self.torch.uint8)
                recv = self.torch.empty(sum(rb), dtype=self.torch.uint8)
                self.dist.all_to_all_single(recv, send, output_split_sizes=rb, input_split_sizes=sb)
                at = 0
                got = recv.numpy()
                for r in range(w):
                    if rb[r]:
                        self.memory.write(recv_ptr[t * w + r], np.ascontiguousarray(got[at:at + rb[r]]))
                    at += rb[r]
            return 0
        except Exception as e:
            self.error = e
            return -1
