#!/usr/bin/env python3
"""One rank's share of the multi-GPU path (P = 1 on one card), for `rocprofv3 --kernel-trace --stats`: four-step NTT at
2^log_n natural -> natural (three exchange-transposes; the four-step code is forced, one rank would take the direct transform),
distributed LDE 2^(log_n-2) -> 2^log_n, sharded FRI commit with every round taken through the sharded code (min_half = 2^8) and
with the default hand-over to the local driver (min_half = 2^16).  The kernel list must hold no torch copy/permute kernel.
STARK_DIST_PYTHON=1 runs dist_ntt.py's own orchestration instead of the C++ one (sh_dist_*) for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import stark_anatomy_amd as sa
from stark_anatomy_amd import dist_ntt as dn
from stark_anatomy_amd.field import GENERATOR
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1 << log_n
eng = dn.GpuEngine(0)
raw = np.frombuffer(np.random.default_rng(7).bytes(n * 16), dtype=np.uint64).copy().reshape(n, 2)
raw[:, 1] &= np.uint64(0x7FFFFFFFFFFFFFFF)
x = eng.from_host(raw)
mode = "python orchestration" if os.environ.get("STARK_DIST_PYTHON") == "1" else "sh_dist_* (C++)"
for rep in range(reps):
    os.environ["STARK_HIP_DIST_FORCE_FOURSTEP"] = "1"
    t0 = time.perf_counter()
    X = dn.dist_ntt(x, log_n, eng)
    eng.ctx.sync(); t1 = time.perf_counter()
    del os.environ["STARK_HIP_DIST_FORCE_FOURSTEP"]
    X1 = dn.dist_ntt(x, log_n, eng)
    eng.ctx.sync(); t1b = time.perf_counter()
    cw = dn.dist_coset_lde(x[: n // 4], log_n - 2, 2, GENERATOR, eng)
    eng.ctx.sync(); t2 = time.perf_counter()
    roots, last, ps = dn.dist_fri_commit(cw, n, eng, GENERATOR, 4, 17, min_half=1 << 8)
    eng.ctx.sync(); t3 = time.perf_counter()
    roots2, last2, ps2 = dn.dist_fri_commit(cw, n, eng, GENERATOR, 4, 17)
    eng.ctx.sync(); t4 = time.perf_counter()
    ps3 = sa.ProofStream()
    roots3, last3 = sa.Fri(GENERATOR, None, n, 4, 17, ctx=eng.ctx).commit(cw, ps3)
    t5 = time.perf_counter()
    assert [bytes(a) for a in roots] == [h.as_bytes() for h in roots3] == [bytes(a) for a in roots2]
    print(f"2^{log_n} P=1 [{mode}]: four-step NTT {1e3*(t1-t0):.3f} ms (one rank's default, the direct transform: {1e3*(t1b-t1):.3f}), LDE {1e3*(t2-t1b):.3f} ms, "
          f"sharded commit {1e3*(t3-t2):.3f} ms (min_half 2^8, {len(roots)} rounds) / {1e3*(t4-t3):.3f} ms (default min_half 2^16); "
          f"single-GPU entry point {1e3*(t5-t4):.3f} ms; chain_retries {eng.ctx.chain_retries}", flush=True)
