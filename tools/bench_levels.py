"""Wall time of sh_dev_merkle_commit (digests -> root, host-visible) for L = 2 .. 2^17: the slope is the cost of one
more Merkle level on the serial chain; the intercept is launch + hand-off to the host."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stark_anatomy_amd as sa
from stark_anatomy_amd import _lib
from stark_anatomy_amd.context import _u8p

ctx = sa.Context(0)
H = ctx.handle
maxlg = 17
buf = ctx.alloc((1 << maxlg) * 64).upload(np.random.default_rng(1).integers(0, 256, (1 << maxlg) * 64, dtype=np.uint8))
root = np.zeros(64, dtype=np.uint8)
prev = None
for lg in range(1, maxlg + 1):
    L = 1 << lg
    for _ in range(20):
        _lib.check(_lib.lib.sh_dev_merkle_commit(H, buf.ptr, L, _u8p(root)))
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.check(_lib.lib.sh_dev_merkle_commit(H, buf.ptr, L, _u8p(root)))
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"L=2^{lg:<2d}  {us:7.2f} us" + (f"   (+{us - prev:5.2f})" if prev is not None else ""), flush=True)
    prev = us
