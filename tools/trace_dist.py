#!/usr/bin/env python3
"""One rank's share of the multi-GPU path (P = 1 on one card), for `rocprofv3 --kernel-trace --stats`: four-step NTT at
2^log_n natural -> natural (three exchange-transposes), distributed LDE 2^(log_n-2) -> 2^log_n, sharded FRI commit with
every round taken through the sharded code (min_half = 2^8).  The kernel list must hold no torch copy/permute kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from stark_anatomy_amd import dist_ntt as dn
from stark_anatomy_amd.field import GENERATOR
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log_n
eng = dn.GpuEngine(0)
raw = np.frombuffer(np.random.default_rng(7).bytes(n * 16), dtype=np.uint64).copy().reshape(n, 2)
raw[:, 1] &= np.uint64(0x7FFFFFFFFFFFFFFF)
x = eng.from_host(raw)
for rep in range(2):
    t0 = time.perf_counter()
    X = dn.dist_ntt(x, log_n, eng)
    eng.ctx.sync(); t1 = time.perf_counter()
    cw = dn.dist_coset_lde(x[: n // 4], log_n - 2, 2, GENERATOR, eng)
    eng.ctx.sync(); t2 = time.perf_counter()
    roots, last, ps = dn.dist_fri_commit(cw, n, eng, GENERATOR, 4, 17, min_half=1 << 8)
    eng.ctx.sync(); t3 = time.perf_counter()
    print(f"2^{log_n} P=1: four-step NTT {1e3*(t1-t0):.3f} ms, LDE {1e3*(t2-t1):.3f} ms, sharded-path commit {1e3*(t3-t2):.3f} ms ({len(roots)} rounds)", flush=True)
