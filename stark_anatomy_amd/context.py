"""
context.py — sh_ctx wrapper and device buffers.  One Context per GPU / process (one process per GPU
under torch.distributed); all work is enqueued on the context's HIP stream.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def elems_to_array(vals):
    """python ints (canonical field values) or an (n,2) uint64 array -> contiguous (n,2) uint64 {lo,hi}"""
    if isinstance(vals, np.ndarray):
        a = np.ascontiguousarray(vals, dtype=np.uint64)
        return a.reshape(-1, 2)
    vals = list(vals)
    a = np.empty((len(vals), 2), dtype=np.uint64)
    for i, v in enumerate(vals):
        v = int(v)
        a[i, 0] = v & 0xFFFFFFFFFFFFFFFF
        a[i, 1] = v >> 64
    return a


def array_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 2)
    return [int(lo) | (int(hi) << 64) for lo, hi in arr]


def _u64p(a): return a.ctypes.data_as(_lib.u64p)
def _u8p(a): return a.ctypes.data_as(_lib.u8p)
def scalar(v): return elems_to_array([v])


class DeviceBuffer:
    """`nbytes` of device memory owned by a Context (hipMalloc through sh_dev_alloc)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        check(lib.sh_dev_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, host_array):
        a = np.ascontiguousarray(host_array)
        assert a.nbytes <= self.nbytes
        check(lib.sh_dev_upload(self.ctx.handle, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib.sh_dev_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr and self.ctx.handle:
            check(lib.sh_dev_free(self.ctx.handle, self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_ptr(x):
    """DeviceBuffer | int | torch CUDA tensor -> raw device address"""
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda"):
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError("expected a contiguous CUDA/HIP tensor")
        return x.data_ptr()
    raise TypeError(f"not a device buffer: {type(x)}")


class Context:
    def __init__(self, device=0, stream=None):
        h = _lib.ctx_t()
        check(lib.sh_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle = h
        self.device = int(device)

    def sync(self): check(lib.sh_ctx_sync(self.handle))

    @property
    def chain_retries(self):
        """rounds the FRI commit driver had to enqueue a second time on this context (stark_hip.h: sh_ctx_chain_retries)"""
        return int(lib.sh_ctx_chain_retries(self.handle))

    def alloc(self, nbytes): return DeviceBuffer(self, nbytes)

    def upload_elems(self, vals):
        a = elems_to_array(vals)
        return DeviceBuffer(self, max(a.nbytes, 16)).upload(a)

    def close(self):
        if self.handle:
            lib.sh_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = None


def default_context():
    """lazily created Context on device LOCAL_RANK (or 0)."""
    global _default
    if _default is None:
        import os
        _default = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default
