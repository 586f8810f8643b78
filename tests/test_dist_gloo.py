"""
world_size-2 rehearsal (gloo, CPU) of the multi-GPU path: whole commits are sharded over ranks with no data-path
collective, results are gathered to rank 0 in the original order.  The local engine here is the CPU oracle
(tests may use it; the product's engine is the GPU) — what is under test is sharding, ordering and gathering.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
import numpy as np
import torch  # before stark_anatomy_amd (two HIP runtimes otherwise, DESIGN.md §1)
import torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"])
from stark_anatomy_amd.dist import commit_batch_sharded, shard_range, max_over_ranks
from oracle import coracle as co
from oracle import pyoracle as po

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B, N = int(os.environ["TEST_B"]), int(os.environ["TEST_N"])

def oracle_engine(cws):
    roots, last, streams = [], [], []
    for cw in cws:
        r, _, l, ps = co.fri_commit(cw, 4, 17, po.G)
        roots.append(np.frombuffer(b"".join(r), dtype=np.uint8).reshape(-1, 64)); last.append(l); streams.append(ps.serialize())
    R = co.fri_num_rounds(N, 4, 17)
    if not roots:
        return np.zeros((0, R, 64), np.uint8), np.zeros((0, N >> (R - 1), 2), np.uint64), []
    return np.stack(roots), np.stack(last), streams

def make(a, b):                       # each rank materialises only its own slice
    if a is None:
        return B
    return [co.splitmix_elems(1000 + i, N) for i in range(a, b)]

out = commit_batch_sharded(make, oracle_engine)
t = max_over_ranks(1.0 + rank)
assert t == float(world), t
if rank == 0:
    roots, last, streams = out
    pickle.dump((roots, last, streams), open(os.environ["TEST_OUT"], "wb"))
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("B,world", [(5, 2), (1, 2), (8, 3)])
def test_sharded_batch_commit_gloo(tmp_path, B, world):
    from oracle import coracle as co
    from oracle import pyoracle as po
    from stark_anatomy_amd.dist import shard_range
    N = 1 << 8
    # shard arithmetic: contiguous, balanced, covers everything
    cuts = [shard_range(B, world, r) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == B and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "out.pkl"
    env = dict(os.environ, REPO_ROOT=ROOT, TEST_B=str(B), TEST_N=str(N), TEST_OUT=str(out), MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() + B * 7 + world) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    import pickle
    roots, last, streams = pickle.load(open(out, "rb"))
    assert roots.shape[0] == B and len(streams) == B
    for i in range(B):
        r, _, l, ps = co.fri_commit(co.splitmix_elems(1000 + i, N), 4, 17, po.G)
        assert roots[i].tobytes() == b"".join(r) and (last[i] == l).all() and streams[i] == ps.serialize()
