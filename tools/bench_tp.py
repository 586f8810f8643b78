#!/usr/bin/env python3
"""Throughput leg of bench.py in isolation: sh_dev_fri_commit_batch for B x 2^log_n (two lock-step groups when B·N >= 2^23), every call
timed.  TORCH_STREAM=1: the context rides on a torch side stream as in bench.py (default: the context's own stream)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stark_anatomy_amd as sa
from stark_anatomy_amd import _lib
from stark_anatomy_amd.context import _u8p, _u64p
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n = 1 << log_n
if os.environ.get("TORCH_STREAM") == "1":
    import torch
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(device=0)
    ctx = sa.Context(0, stream=stream.cuda_stream)
else:
    ctx = sa.Context(0)
H = ctx.handle
raw = np.frombuffer(np.random.default_rng(1).bytes(16 * B * n), dtype=np.uint64).reshape(-1, 2)
red = np.empty_like(raw); _lib.check(_lib.lib.sh_fp_new(H, _u64p(np.ascontiguousarray(raw)), _u64p(red), B * n))
d = ctx.alloc(B * n * 16).upload(red)
p = _lib.FriParams(); p.expansion, p.num_colinearity_tests = 4, 17
p.offset[0], p.offset[1] = sa.GENERATOR & 0xFFFFFFFFFFFFFFFF, sa.GENERATOR >> 64
R = int(_lib.lib.sh_fri_num_rounds(n, 4, 17))
roots = np.zeros((B, R, 64), dtype=np.uint8); last = np.empty((B, n >> (R - 1), 2), dtype=np.uint64); ll = C.c_size_t(0)
all_pss = [[sa.ProofStream() for _ in range(B)] for _ in range(calls)]
all_handles = [(_lib.ps_t * B)(*[q.handle for q in pss]) for pss in all_pss]
ts = []
for it in range(calls):
    t0 = time.perf_counter()
    _lib.check(_lib.lib.sh_dev_fri_commit_batch(H, d.ptr, B, n, C.byref(p), all_handles[it], _u8p(roots), _u64p(last), C.byref(ll)))
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"B={B} N=2^{log_n} torch_stream={os.environ.get('TORCH_STREAM','0')}: ms per call:", " ".join(f"{t:.2f}" for t in ts), flush=True)
