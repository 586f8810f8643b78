"""
Rehearsal (gloo, CPU ranks) of the single-transform / single-commit multi-GPU path in stark_anatomy_amd/dist_ntt.py:
four-step NTT with all-to-all transposes, distributed coset LDE, and the FRI commit of one codeword sharded over
ranks.  Local compute is the CPU oracle (tests/cpu_engine.py); what is under test is the exchange logic and that the
distributed result is bit-identical to the single-process oracle.
"""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
import numpy as np
import torch  # before stark_anatomy_amd
import torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
from stark_anatomy_amd import dist_ntt as dn
from cpu_engine import OracleEngine
from oracle import coracle as co
from oracle import pyoracle as po

multi = "RANK" in os.environ
if multi:
    dist.init_process_group("gloo")
P, g = (dist.get_world_size(), dist.get_rank()) if multi else (1, 0)
eng = OracleEngine()
log_n = int(os.environ["TEST_LOG_N"])
n = 1 << log_n
x = co.splitmix_elems(77, n)
sl = lambda a, m: a[g * (m // P):(g + 1) * (m // P)]
res = {}

# forward, natural -> natural
X = dn.dist_ntt(eng.from_host(sl(x, n)), log_n, eng)
res["ntt"] = eng.to_host(X)
# inverse of it
res["intt"] = eng.to_host(dn.dist_ntt(X.clone(), log_n, eng, inverse=True))
# rows layout out, columns layout in
l1, l2 = dn.split_log(log_n); n1, n2 = 1 << l1, 1 << l2
Z = dn.dist_ntt(eng.from_host(sl(x, n)), log_n, eng, output_layout="rows")
res["rows"] = eng.to_host(Z)
xc = x.reshape(n1, n2, 2)[:, g * (n2 // P):(g + 1) * (n2 // P)].transpose(1, 0, 2).copy()
res["from_cols"] = eng.to_host(dn.dist_ntt(eng.from_host(xc), log_n, eng, input_layout="columns"))
# coset LDE of the first n/4 coefficients
m = n // 4
res["lde"] = eng.to_host(dn.dist_coset_lde(eng.from_host(sl(x[:m], m)), log_n - 2, 2, po.G, eng))
# sharded FRI commit of the codeword x
roots, last, ps = dn.dist_fri_commit(eng.from_host(sl(x, n)), n, eng, po.G, 4, int(os.environ["TEST_T"]), min_half=int(os.environ["TEST_MIN_HALF"]))
res["fri"] = (roots, np.asarray(last), ps.serialize())

if multi:
    out = [None] * P if g == 0 else None
    dist.gather_object(res, out, dst=0)
    dist.barrier(); dist.destroy_process_group()
else:
    out = [res]
if g == 0:
    pickle.dump(out, open(os.environ["TEST_OUT"], "wb"))
'''


def _run(tmp_path, world, log_n, t, min_half):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "out.pkl"
    env = dict(os.environ, REPO_ROOT=ROOT, TEST_LOG_N=str(log_n), TEST_T=str(t), TEST_MIN_HALF=str(min_half), TEST_OUT=str(out),
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    if world == 1:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, str(script)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(29650 + world + log_n), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return pickle.load(open(out, "rb"))


@pytest.mark.parametrize("world,log_n,t,min_half", [(1, 10, 4, 16), (2, 11, 4, 16), (4, 12, 17, 32)])
def test_dist_ntt_lde_fri_match_oracle(tmp_path, world, log_n, t, min_half):
    from oracle import coracle as co
    from oracle import pyoracle as po
    parts = _run(tmp_path, world, log_n, t, min_half)
    assert len(parts) == world
    n = 1 << log_n
    x = co.splitmix_elems(77, n)
    X = co.ntt(x)
    cat = lambda key: np.concatenate([p[key] for p in parts])
    assert np.array_equal(cat("ntt"), X)
    assert np.array_equal(cat("intt"), x)
    assert np.array_equal(cat("from_cols"), X)
    l1 = log_n // 2
    n1, n2 = 1 << l1, 1 << (log_n - l1)
    rows = cat("rows").reshape(n1, n2, 2)                     # [k1][k2] = X[k1 + n1·k2]
    assert np.array_equal(rows, X.reshape(n2, n1, 2).transpose(1, 0, 2))
    assert np.array_equal(cat("lde"), co.coset_lde(x[: n // 4], 2, po.G))
    roots, _, last, ps = co.fri_commit(x, 4, t, po.G)
    assert len(roots) >= 2
    for p in parts:                                           # every rank ends with the same transcript
        r, l, s = p["fri"]
        assert [bytes(a) for a in r] == roots
        assert np.array_equal(l, last)
        assert s == ps.serialize()


def test_exchange_transpose_single_process():
    import torch
    from stark_anatomy_amd.dist_ntt import exchange_transpose
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cpu_engine import OracleEngine
    x = torch.arange(3 * 8 * 2, dtype=torch.int64).reshape(3, 8, 2)
    assert torch.equal(exchange_transpose(x, OracleEngine()), x.transpose(0, 1).contiguous())
