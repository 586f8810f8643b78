import ctypes as C, sys, time, os
import numpy as np
lib = C.CDLL(os.path.abspath("stark_anatomy_amd/libstark_hip.so"))
ctx = C.c_void_p(); assert lib.sh_ctx_create(0, None, C.byref(ctx)) == 0
for log_n in (24, 20):
    n = 1 << log_n; d = C.c_void_p(); lib.sh_dev_alloc(ctx, C.c_size_t(n * 16), C.byref(d))
    x = np.frombuffer(np.random.default_rng(1).bytes(n * 16), dtype=np.uint64).copy(); x[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    lib.sh_dev_upload(ctx, d, C.c_void_p(x.ctypes.data), C.c_size_t(n * 16))
    for inv in (0, 1):
        lib.sh_dev_ntt(ctx, d, C.c_uint32(log_n), inv); lib.sh_ctx_sync(ctx)
        t0 = time.perf_counter()
        for _ in range(10): lib.sh_dev_ntt(ctx, d, C.c_uint32(log_n), inv)
        lib.sh_ctx_sync(ctx)
        print(f"2^{log_n} inverse={inv}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
    lib.sh_dev_free(ctx, d)
