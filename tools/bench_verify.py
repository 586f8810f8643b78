import sys, time, numpy as np
sys.path.insert(0, '.')
import stark_anatomy_amd as sa
from oracle import coracle as co
ctx = sa.Context(0)
lp = 20
c = co.splitmix_elems(4000, 1 << (lp - 2))
cw = sa.coset_lde(c, 2, sa.GENERATOR, ctx=ctx)
d = ctx.alloc((1 << lp) * 16).upload(cw)
f = sa.Fri(sa.GENERATOR, None, 1 << lp, 4, 17, ctx=ctx)
ps = sa.ProofStream(); f.prove(d, ps)
ser = ps.serialize()
for _ in range(3):
    t0 = time.perf_counter(); q = sa.ProofStream.deserialize(ser); t1 = time.perf_counter()
    ok, why = f.verify(q); t2 = time.perf_counter()
    print(f"deserialize {1e3*(t1-t0):.2f} ms  verify {1e3*(t2-t1):.2f} ms  ok={ok}", flush=True)
