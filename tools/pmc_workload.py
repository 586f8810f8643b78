#!/usr/bin/env python3
"""One pass of the hot path at 2^log_n for counter collection: coset LDE (scale + NTT passes) then a full FRI commit.
No torch: plain ctypes on libstark_hip.so, so `rocprofv3 --pmc ... -- python3 tools/pmc_workload.py 24` starts no other runtime."""
import ctypes as C, os, sys
import numpy as np
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stark_anatomy_amd", "libstark_hip.so")
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # B > 1: no LDE, one lock-step batched commit of B x 2^log_n
lib = C.CDLL(os.path.abspath(so))
ctx = C.c_void_p()
assert lib.sh_ctx_create(0, None, C.byref(ctx)) == 0, lib.sh_last_error()
N = 1 << log_n
n = N >> 2


class Params(C.Structure):
    _fields_ = [("expansion", C.c_uint32), ("t", C.c_uint32), ("offset", C.c_uint64 * 2)]


G = 85408008396924667383611388730472331217
off = (C.c_uint64 * 2)(G & (2**64 - 1), G >> 64)
if B > 1:
    d_b = C.c_void_p()
    assert lib.sh_dev_alloc(ctx, C.c_size_t(B * N * 16), C.byref(d_b)) == 0
    xb = np.frombuffer(np.random.default_rng(2).bytes(B * N * 16), dtype=np.uint64).copy()
    xb[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    assert lib.sh_dev_upload(ctx, d_b, C.c_void_p(xb.ctypes.data), C.c_size_t(B * N * 16)) == 0
    p = Params(4, 17, off)
    streams = (C.c_void_p * B)()
    for b in range(B):
        assert lib.sh_proofstream_new(C.byref(streams, b * C.sizeof(C.c_void_p))) == 0
    roots = (C.c_uint8 * (64 * 32 * B))()
    last = (C.c_uint64 * (2 * 1024 * B))()
    ll = C.c_size_t(0)
    lib.sh_dev_fri_commit_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for rep in range(1):
        assert lib.sh_dev_fri_commit_batch(ctx, d_b, C.c_size_t(B), C.c_size_t(N), C.byref(p), streams, roots, last, C.byref(ll)) == 0, lib.sh_last_error()
    lib.sh_ctx_sync(ctx)
    print("ok batch", B, log_n, ll.value, flush=True)
    sys.exit(0)
d_c, d_x = C.c_void_p(), C.c_void_p()
assert lib.sh_dev_alloc(ctx, C.c_size_t(n * 16), C.byref(d_c)) == 0 and lib.sh_dev_alloc(ctx, C.c_size_t(N * 16), C.byref(d_x)) == 0
x = np.frombuffer(np.random.default_rng(1).bytes(n * 16), dtype=np.uint64).copy()
x[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF)          # < 2^127 < p: canonical
assert lib.sh_dev_upload(ctx, d_c, C.c_void_p(x.ctypes.data), C.c_size_t(n * 16)) == 0
lib.sh_dev_coset_lde.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
assert lib.sh_dev_coset_lde(ctx, d_c, log_n - 2, 2, off, d_x) == 0, lib.sh_last_error()


p = Params(4, 17, off)
ps = C.c_void_p()
assert lib.sh_proofstream_new(C.byref(ps)) == 0
roots = (C.c_uint8 * (64 * 64))()
last = (C.c_uint64 * (2 * 1024))()
ll = C.c_size_t(0)
lib.sh_dev_fri_commit_proofstream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
assert lib.sh_dev_fri_commit_proofstream(ctx, d_x, C.c_size_t(N), C.byref(p), ps, roots, last, C.byref(ll)) == 0, lib.sh_last_error()
lib.sh_ctx_sync(ctx)
print("ok", bytes(roots[:8]).hex(), ll.value, flush=True)
