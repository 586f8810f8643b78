import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, '.')
lib = C.CDLL(os.path.abspath("stark_anatomy_amd/libstark_hip.so"))
ctx = C.c_void_p(); assert lib.sh_ctx_create(0, None, C.byref(ctx)) == 0
G = 85408008396924667383611388730472331217
class Params(C.Structure):
    _fields_ = [("expansion", C.c_uint32), ("t", C.c_uint32), ("offset", C.c_uint64 * 2)]
p = Params(4, 17, (C.c_uint64 * 2)(G & (2**64 - 1), G >> 64))
lib.sh_dev_fri_commit_proofstream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.sh_ctx_chain_retries.restype = C.c_uint64
for log_n, reps in ((12, 1500), (16, 1500), (20, 600), (22, 100)):
    n = 1 << log_n
    d = C.c_void_p(); lib.sh_dev_alloc(ctx, C.c_size_t(n * 16), C.byref(d))
    x = np.frombuffer(np.random.default_rng(log_n).bytes(n * 16), dtype=np.uint64).copy(); x[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    lib.sh_dev_upload(ctx, d, C.c_void_p(x.ctypes.data), C.c_size_t(n * 16))
    roots = (C.c_uint8 * (64 * 64))(); last = (C.c_uint64 * (2 * 4096))(); ll = C.c_size_t(0)
    ref = None; t0 = time.perf_counter(); worst = 0
    for i in range(reps):
        ps = C.c_void_p(); lib.sh_proofstream_new(C.byref(ps))
        t1 = time.perf_counter()
        rc = lib.sh_dev_fri_commit_proofstream(ctx, d, C.c_size_t(n), C.byref(p), ps, roots, last, C.byref(ll))
        worst = max(worst, time.perf_counter() - t1)
        lib.sh_proofstream_free(ps)
        assert rc == 0, (log_n, i, rc)
        b = bytes(roots)
        if ref is None: ref = b
        assert b == ref, (log_n, i)
    dt = time.perf_counter() - t0
    print(f"2^{log_n}: {reps} commits, {1e3*dt/reps:.3f} ms avg incl. python, worst call {1e3*worst:.3f} ms, chain retries so far {lib.sh_ctx_chain_retries(ctx)}", flush=True)
    lib.sh_dev_free(ctx, d)
