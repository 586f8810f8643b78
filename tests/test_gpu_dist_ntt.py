"""
GPU parity tests for the local steps of the multi-GPU path (batched NTT, four-step twiddle, ranged coset scale,
ranged fold) and for dist_ntt.GpuEngine: in one process (P = 1: the four-step decomposition itself), and with two
ranks sharing cuda:0 over gloo (tensors staged through host memory) so that the all-to-all transposes, the regroup
and the sub-root gather run with the real kernels.  Bit-exact against the CPU oracle.
"""
import ctypes as C
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401  (before stark_anatomy_amd)

import stark_anatomy_amd as sa  # noqa: E402
from oracle import coracle as co  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from stark_anatomy_amd import _lib  # noqa: E402
from stark_anatomy_amd.context import _u64p, scalar  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, G = po.P, po.G


@pytest.fixture(scope="module")
def ctx():
    c = sa.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("batch,log_len", [(1, 5), (7, 0), (3, 1), (5, 8), (64, 9), (16, 12), (3, 13), (4, 17)])
def test_ntt_batch_vs_oracle(ctx, batch, log_len):
    n = 1 << log_len
    x = co.splitmix_elems(500 + batch + log_len, batch * n)
    for inverse in (False, True):
        a = x.copy()
        _lib.check(_lib.lib.sh_ntt_batch(ctx.handle, _u64p(a), batch, log_len, int(inverse)))
        exp = np.concatenate([co.ntt(x[b * n:(b + 1) * n], inverse) for b in range(batch)])
        assert np.array_equal(a, exp)


def test_ntt_batch_rejects_oversize(ctx):
    a = np.zeros((2, 2), dtype=np.uint64)
    assert _lib.lib.sh_ntt_batch(ctx.handle, _u64p(a), 1, 29, 0) == _lib.SH_EINVAL
    assert _lib.lib.sh_ntt_batch(ctx.handle, _u64p(a), 1 << 40, 10, 0) == _lib.SH_EINVAL
    assert _lib.lib.sh_ntt_batch(ctx.handle, None, 0, 10, 0) == _lib.SH_OK


@pytest.mark.parametrize("rows,log_cols,row0,col0,log_n", [(5, 3, 0, 0, 6), (8, 4, 24, 0, 9), (3, 2, 1000, 7, 14), (16, 6, 4080, 0, 20)])
def test_twiddle_scale_vs_python(ctx, rows, log_cols, row0, col0, log_n):
    cols = 1 << log_cols
    x = co.splitmix_elems(9, rows * cols)
    vals = co.to_ints(x)
    for inverse in (False, True):
        a = x.copy()
        _lib.check(_lib.lib.sh_twiddle_scale(ctx.handle, _u64p(a), rows, log_cols, row0, col0, log_n, int(inverse)))
        w = po.root_of_unity(log_n)
        if inverse:
            w = po.fp_inv(w)
        exp = [v * pow(w, ((row0 + i // cols) * (col0 + i % cols)) % (1 << log_n), P) % P for i, v in enumerate(vals)]
        assert co.to_ints(a) == exp


def test_scale_powers_and_fold_range_vs_oracle(ctx):
    n = 1 << 12
    x = co.splitmix_elems(21, n)
    d = ctx.alloc(n * 16).upload(x)
    j0 = 123456
    _lib.check(_lib.lib.sh_dev_scale_powers(ctx.handle, d.ptr, n, j0, _u64p(scalar(G))))
    got = co.to_ints(d.download(np.uint64, (n, 2)))
    assert got == [v * pow(G, j0 + i, P) % P for i, v in enumerate(co.to_ints(x))]
    # ranged fold == the matching slice of the oracle's whole-codeword fold
    L, alpha, off = n, 0x1234567890abcdef1234567890abcdef % P, pow(G, 4, P)
    omega = po.root_of_unity(12)
    exp = co.fri_fold(x, alpha, off, omega)
    d.upload(x)
    for i0, cnt in [(0, L // 2), (512, 512), (2047, 1), (100, 37)]:
        out = ctx.alloc(cnt * 16)
        _lib.check(_lib.lib.sh_dev_fri_fold_range(ctx.handle, d.ptr + 16 * i0, d.ptr + 16 * (L // 2 + i0), cnt, i0, _u64p(scalar(alpha)),
                                                  _u64p(scalar(off)), _u64p(scalar(omega)), out.ptr))
        assert np.array_equal(out.download(np.uint64, (cnt, 2)), exp[i0:i0 + cnt])
    assert _lib.lib.sh_dev_fri_fold_range(ctx.handle, d.ptr, d.ptr, 4, 0, _u64p(scalar(alpha)), _u64p(scalar(0)), _u64p(scalar(omega)), d.ptr) == _lib.SH_EDIVZERO


@pytest.fixture(scope="module")
def engine():
    from stark_anatomy_amd.dist_ntt import GpuEngine
    return GpuEngine(0)


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 8), (32, 32), (33, 65), (100, 7), (1024, 4096), (4096, 513)])
def test_transpose_and_interleave_kernels(ctx, rows, cols):
    """the pack / unpack kernels of an all-to-all exchange (sh_dev_transpose, sh_dev_interleave_blocks) against numpy"""
    x = co.splitmix_elems(rows * 31 + cols, rows * cols)
    d_in, d_out = ctx.alloc(rows * cols * 16).upload(x), ctx.alloc(rows * cols * 16)
    _lib.check(_lib.lib.sh_dev_transpose(ctx.handle, d_in.ptr, rows, cols, d_out.ptr))
    assert np.array_equal(d_out.download(np.uint64, (cols, rows, 2)), x.reshape(rows, cols, 2).transpose(1, 0, 2))
    for Pn in (1, 2, 4):
        if cols % Pn:
            continue
        b, a = cols // Pn, rows                                        # in = [s][c][r] -> out = [c][s][r]
        _lib.check(_lib.lib.sh_dev_interleave_blocks(ctx.handle, d_in.ptr, Pn, b, a, d_out.ptr))
        assert np.array_equal(d_out.download(np.uint64, (b, Pn, a, 2)), x.reshape(Pn, b, a, 2).transpose(1, 0, 2, 3))
    assert _lib.lib.sh_dev_transpose(ctx.handle, d_in.ptr, rows, cols, d_in.ptr) == _lib.SH_EINVAL      # no aliasing
    d_in.free(); d_out.free()


def test_pair_roots_stay_on_device(ctx):
    """sh_dev_merkle_root_elems + sh_dev_merkle_commit_pairs (a sharded round's sub-roots and top tree) == one tree over
    the whole codeword: P = 4 ranks' half-slices s and P + s."""
    Pn, h = 4, 1 << 10
    cw = co.splitmix_elems(77, 2 * Pn * h)
    d_cw = ctx.alloc(cw.nbytes).upload(cw)
    d_pairs = ctx.alloc(128 * Pn)
    for s in range(Pn):
        _lib.check(_lib.lib.sh_dev_merkle_root_elems(ctx.handle, d_cw.ptr + 16 * s * h, h, d_pairs.ptr + 128 * s))
        _lib.check(_lib.lib.sh_dev_merkle_root_elems(ctx.handle, d_cw.ptr + 16 * (Pn + s) * h, h, d_pairs.ptr + 128 * s + 64))
    root = np.zeros(64, dtype=np.uint8)
    from stark_anatomy_amd.context import _u8p
    _lib.check(_lib.lib.sh_dev_merkle_commit_pairs(ctx.handle, d_pairs.ptr, Pn, _u8p(root)))
    assert root.tobytes() == co.merkle_commit_elems(cw)
    assert _lib.lib.sh_dev_merkle_commit_pairs(ctx.handle, d_pairs.ptr, 3, _u8p(root)) == _lib.SH_EINVAL
    d_cw.free(); d_pairs.free()


def test_native_dist_rccl_single_rank(engine):
    """sh_dist_create on one rank: librccl is dlopen()ed, a one-rank communicator is made, and sh_dist_selftest drives all three
    transport operations (grouped ncclSend/ncclRecv as all-to-all and as a point-to-point pair, ncclAllGather) on the card."""
    from stark_anatomy_amd import dist_ntt as dn
    nat = dn.NativeDist(engine)
    assert nat.transport == "rccl" and (nat.P, nat.g) == (1, 0)
    nat.selftest()
    nat.close()


@pytest.mark.parametrize("mode", ["native-direct", "native-four-step", "python"])
@pytest.mark.parametrize("log_n", [2, 9, 16, 20, 24])
def test_four_step_single_rank(engine, log_n, mode, monkeypatch):
    """P = 1: the four-step decomposition (batched column NTTs, twiddle, batched row NTTs, transposes by k_transpose16)
    equals the direct transform; from 2^20 against the library's own one-shot NTT, below that against the oracle.
    native-*: sh_dist_ntt (one rank takes the direct transform unless the four-step code is forced); python: dist_ntt.py's own
    orchestration over the same kernels (what the gloo rehearsals with a CPU engine run)."""
    from stark_anatomy_amd import dist_ntt as dn
    if mode == "native-four-step":
        monkeypatch.setenv("STARK_HIP_DIST_FORCE_FOURSTEP", "1")
    if mode == "python":
        if log_n == 24:
            pytest.skip("covered by the native path at this size")
        monkeypatch.setenv("STARK_DIST_PYTHON", "1")
    n = 1 << log_n
    x = co.splitmix_elems(31 + log_n, n)
    X = engine.to_host(dn.dist_ntt(engine.from_host(x), log_n, engine))
    exp = co.ntt(x) if log_n <= 16 else sa.ntt(x, ctx=engine.ctx)
    assert np.array_equal(X, exp)
    back = engine.to_host(dn.dist_ntt(engine.from_host(X), log_n, engine, inverse=True))
    assert np.array_equal(back, x)


@pytest.mark.parametrize("layouts", [("columns", "natural"), ("natural", "rows"), ("columns", "rows")])
def test_four_step_layouts_single_rank(engine, layouts, monkeypatch):
    """sh_dist_ntt's relaxed layouts (one exchange instead of three) against the natural-order transform, P = 1."""
    from stark_anatomy_amd import dist_ntt as dn
    monkeypatch.setenv("STARK_HIP_DIST_FORCE_FOURSTEP", "1")
    log_n = 13
    l1, l2 = dn.split_log(log_n)
    n1, n2 = 1 << l1, 1 << l2
    x = co.splitmix_elems(71, 1 << log_n)
    exp = co.ntt(x)
    inp = x if layouts[0] == "natural" else np.ascontiguousarray(x.reshape(n1, n2, 2).transpose(1, 0, 2)).reshape(-1, 2)     # [j2][j1]
    got = engine.to_host(dn.dist_ntt(engine.from_host(inp), log_n, engine, input_layout=layouts[0], output_layout=layouts[1]))
    want = exp if layouts[1] == "natural" else np.ascontiguousarray(exp.reshape(n2, n1, 2).transpose(1, 0, 2)).reshape(-1, 2)  # [k1][k2]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("log_N,min_half", [(16, 1 << 8), (16, 1), (20, 1 << 16), (22, 1 << 16), (22, 1 << 10)])
def test_native_sharded_commit_single_rank(engine, log_N, min_half, monkeypatch):
    """sh_dist_fri_commit on one rank == the single-GPU commit (oracle-checked at these sizes), byte for byte: the sharded rounds
    (fused fold + leaf subtree with a slice offset, chain records, enqueue thread) followed by the locally committed tail."""
    from stark_anatomy_amd import dist_ntt as dn
    monkeypatch.setenv("STARK_HIP_CHAIN_SPIN_US", "50000")        # parity test: a descheduled host thread is not a re-launch cycle (tests/test_gpu_parity.py)
    N = 1 << log_N
    cw = co.splitmix_elems(900 + log_N, N)
    roots, last, ps = dn.dist_fri_commit(engine.from_host(cw), N, engine, G, 4, 17, min_half=min_half)
    if log_N <= 16:
        r, _, l, ops = co.fri_commit(cw, 4, 17, G)
        ser = ops.serialize()
    else:
        ps1 = sa.ProofStream()
        roots1, l = sa.Fri(G, None, N, 4, 17, ctx=engine.ctx).commit(cw, ps1)
        r, ser = [h.as_bytes() for h in roots1], ps1.serialize()
    assert [bytes(a) for a in roots] == r and np.array_equal(last, l) and ps.serialize() == ser
    assert engine.ctx.chain_retries == 0


@pytest.mark.parametrize("mode", ["native", "python"])
def test_dist_lde_and_commit_single_rank(engine, mode, monkeypatch):
    from stark_anatomy_amd import dist_ntt as dn
    if mode == "python":
        monkeypatch.setenv("STARK_DIST_PYTHON", "1")
    log_n = 14
    c = co.splitmix_elems(5, 1 << log_n)
    cw = dn.dist_coset_lde(engine.from_host(c), log_n, 2, G, engine)
    exp_cw = co.coset_lde(c, 2, G)
    assert np.array_equal(engine.to_host(cw), exp_cw)
    roots, last, ps = dn.dist_fri_commit(cw, 1 << (log_n + 2), engine, G, 4, 17, min_half=1 << 8)
    r, _, l, ops = co.fri_commit(exp_cw, 4, 17, G)
    assert [bytes(a) for a in roots] == r and np.array_equal(last, l) and ps.serialize() == ops.serialize()


WORKER = r'''
import os, sys, pickle
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"])
from stark_anatomy_amd import dist_ntt as dn
from stark_anatomy_amd.field import GENERATOR

dist.init_process_group("gloo")
P, g = dist.get_world_size(), dist.get_rank()
eng = dn.GpuEngine(0)                         # both ranks share cuda:0; gloo stages tensors through the host
log_n = int(os.environ["TEST_LOG_N"]); n = 1 << log_n
x = np.load(os.environ["TEST_IN"])
sl = lambda a, m: a[g * (m // P):(g + 1) * (m // P)]
res = {}
nat = dn._native(eng, None)
if nat is not None:
    assert nat.transport == "callbacks"
    nat.selftest()
X = dn.dist_ntt(eng.from_host(sl(x, n)), log_n, eng)
res["ntt"] = eng.to_host(X)
m = n // 4
cw = dn.dist_coset_lde(eng.from_host(sl(x[:m], m)), log_n - 2, 2, GENERATOR, eng)
res["lde"] = eng.to_host(cw)
roots, last, ps = dn.dist_fri_commit(cw, n, eng, GENERATOR, 4, 17, min_half=256)
res["fri"] = (roots, np.asarray(last), ps.serialize())
out = [None] * P if g == 0 else None
dist.gather_object(res, out, dst=0)
dist.barrier(); dist.destroy_process_group()
if g == 0:
    pickle.dump(out, open(os.environ["TEST_OUT"], "wb"))
'''


@pytest.mark.parametrize("log_n,mode,ranks", [(14, "native", 2), (14, "python", 2), (22, "native", 2), (18, "native", 4), (20, "native", 8)])
def test_ranks_on_one_gpu_gloo(tmp_path, log_n, mode, ranks):
    """14: against the oracle.  22 (config 5's shape at a size two ranks on one card finish in seconds): against the
    single-GPU NTT / LDE / commit of the product, which are themselves oracle-checked at that size.
    native: the C++ orchestration (sh_dist_*) with its transport operations served by gloo-backed callbacks that stage through
    host memory — every exchange, regroup and sub-root gather of the product code with two real ranks; python: dist_ntt.py's
    own orchestration over the same kernels.  4 and 8 ranks: the rank counts of the scaling runs, which no RCCL run on a one-GPU
    box can rehearse — slice sizes, sharded round counts and the regroup pattern differ with P."""
    n = 1 << log_n
    x = co.splitmix_elems(99, n)
    np.save(tmp_path / "in.npy", x)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "out.pkl"
    env = dict(os.environ, REPO_ROOT=ROOT, TEST_LOG_N=str(log_n), TEST_IN=str(tmp_path / "in.npy"), TEST_OUT=str(out),
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    if mode == "python":
        env["STARK_DIST_PYTHON"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(29731 + ranks), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    parts = pickle.load(open(out, "rb"))
    if log_n <= 16:
        exp_ntt, exp_cw = co.ntt(x), co.coset_lde(x[: n // 4], 2, G)
        r0, _, l0, ps0 = co.fri_commit(exp_cw, 4, 17, G)
        ser0 = ps0.serialize()
    else:
        c1 = sa.Context(0)
        exp_ntt, exp_cw = sa.ntt(x, ctx=c1), sa.coset_lde(x[: n // 4], 2, G, ctx=c1)
        ps1 = sa.ProofStream()
        roots1, l0 = sa.Fri(G, None, n, 4, 17, ctx=c1).commit(exp_cw, ps1)
        r0, ser0 = [h.as_bytes() for h in roots1], ps1.serialize()
        c1.close()
    assert np.array_equal(np.concatenate([p["ntt"] for p in parts]), exp_ntt)
    assert np.array_equal(np.concatenate([p["lde"] for p in parts]), exp_cw)
    for p in parts:
        roots, last, s = p["fri"]
        assert [bytes(a) for a in roots] == r0 and np.array_equal(last, l0) and s == ser0
