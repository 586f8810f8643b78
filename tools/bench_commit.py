#!/usr/bin/env python3
"""Time sh_dev_fri_commit_proofstream on a resident codeword at 2^log_n; optional .so path to A/B builds.
Prints the first root so two builds can be compared for equality."""
import ctypes as C, sys, time, os
import numpy as np
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "stark_anatomy_amd", "libstark_hip.so")
logs = [int(a) for a in sys.argv[2:]] or [24, 20, 16]
lib = C.CDLL(os.path.abspath(so))
ctx = C.c_void_p()
assert lib.sh_ctx_create(0, None, C.byref(ctx)) == 0
G = 85408008396924667383611388730472331217


class Params(C.Structure):
    _fields_ = [("expansion", C.c_uint32), ("t", C.c_uint32), ("offset", C.c_uint64 * 2)]


p = Params(4, 17, (C.c_uint64 * 2)(G & (2**64 - 1), G >> 64))
lib.sh_dev_fri_commit_proofstream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for log_n in logs:
    n = 1 << log_n
    d = C.c_void_p()
    lib.sh_dev_alloc(ctx, C.c_size_t(n * 16), C.byref(d))
    x = np.frombuffer(np.random.default_rng(log_n).bytes(n * 16), dtype=np.uint64).copy()
    x[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    lib.sh_dev_upload(ctx, d, C.c_void_p(x.ctypes.data), C.c_size_t(n * 16))
    roots = (C.c_uint8 * (64 * 64))()
    last = (C.c_uint64 * (2 * 4096))()
    ll = C.c_size_t(0)

    def run():
        ps = C.c_void_p()
        lib.sh_proofstream_new(C.byref(ps))
        rc = lib.sh_dev_fri_commit_proofstream(ctx, d, C.c_size_t(n), C.byref(p), ps, roots, last, C.byref(ll))
        lib.sh_proofstream_free(ps)
        assert rc == 0
    run(); run()
    reps = 20 if log_n <= 20 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    ms = (time.perf_counter() - t0) / reps * 1e3
    R = lib.sh_fri_num_rounds(C.c_size_t(n), 4, 17)
    print(f"2^{log_n}: {ms:8.3f} ms  {n / ms / 1e6:7.3f} G elems/s  last_root={bytes(roots[64*(R-1):64*(R-1)+8]).hex()}  ({os.path.basename(so)})", flush=True)
    lib.sh_dev_free(ctx, d)
