#!/usr/bin/env python3
"""Time sh_dev_fri_prove (commit + query phase) and the host verifier on an LDE codeword at 2^log_n."""
import sys, time
import numpy as np
import torch  # noqa: F401
import stark_anatomy_amd as sa
from stark_anatomy_amd.proof_stream import ProofStream
logs = [int(a) for a in sys.argv[1:]] or [20, 22, 24]
ctx = sa.Context(0)
for log_n in logs:
    N = 1 << log_n
    n = N >> 2
    raw = np.frombuffer(np.random.default_rng(log_n).bytes(n * 16), dtype=np.uint64).copy().reshape(n, 2)
    raw[:, 1] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    d_c = ctx.alloc(n * 16).upload(raw)
    d_cw = ctx.alloc(N * 16)
    sa.dev_coset_lde(ctx, d_c, log_n - 2, 2, sa.GENERATOR, d_cw)
    fri = sa.Fri(sa.GENERATOR, None, N, 4, 17, ctx=ctx)
    for fn, name in ((fri.commit, "commit"), (fri.prove, "prove")):
        fn(d_cw, ProofStream())
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ps = ProofStream(); fn(d_cw, ps)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"2^{log_n} {name:6s} {ms:8.3f} ms", flush=True)
    ser = ps.serialize()
    t0 = time.perf_counter()
    ok, why = fri.verify(ProofStream.deserialize(ser))
    print(f"2^{log_n} verify {1e3 * (time.perf_counter() - t0):8.3f} ms  accepted={ok}  proof {len(ser) / 1e6:.2f} MB, {len(ps)} items", flush=True)
    d_c.free(); d_cw.free()
