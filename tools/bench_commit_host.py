#!/usr/bin/env python3
"""Time sh_fri_commit_proofstream with the codeword in HOST memory (the literal drop-in call: H2D copy included)."""
import ctypes as C, os, sys, time
import numpy as np
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stark_anatomy_amd", "libstark_hip.so")
lib = C.CDLL(os.path.abspath(so))
ctx = C.c_void_p()
assert lib.sh_ctx_create(0, None, C.byref(ctx)) == 0
G = 85408008396924667383611388730472331217


class Params(C.Structure):
    _fields_ = [("expansion", C.c_uint32), ("t", C.c_uint32), ("offset", C.c_uint64 * 2)]


p = Params(4, 17, (C.c_uint64 * 2)(G & (2**64 - 1), G >> 64))
lib.sh_fri_commit_proofstream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for log_n in [int(a) for a in sys.argv[1:]] or [24, 20]:
    n = 1 << log_n
    x = np.frombuffer(np.random.default_rng(log_n).bytes(n * 16), dtype=np.uint64).copy()
    x[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    roots = (C.c_uint8 * (64 * 64))(); last = (C.c_uint64 * (2 * 4096))(); ll = C.c_size_t(0)

    def run():
        ps = C.c_void_p(); lib.sh_proofstream_new(C.byref(ps))
        rc = lib.sh_fri_commit_proofstream(ctx, C.c_void_p(x.ctypes.data), C.c_size_t(n), C.byref(p), ps, roots, last, C.byref(ll))
        lib.sh_proofstream_free(ps); assert rc == 0
    run(); run()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps): run()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"2^{log_n} from host memory: {ms:8.3f} ms  {n / ms / 1e6:6.3f} G elems/s  ({16 * n / ms / 1e6:5.1f} GB/s if it were all copy)", flush=True)
