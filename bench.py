#!/usr/bin/env python3
"""
bench.py — FRI-commit throughput on MI355X (BASELINE.json metric: "FRI commit field-elems/s at 2^20 and
2^24 domains, 1/2/4/8 GPU; NTT GB/s").

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one full FRI commit (all fold rounds, all Merkle roots, host Fiat-Shamir transcript, last
codeword back on the host) of one synthetic random codeword of N_dom = 2^--log-n elements that is already
resident in HBM.  Each rank (one process per GPU) commits its own independent codeword — the path shards
by whole commits with no data-path collective ("weak" scaling); `value` = all ranks' elements / max-over-
ranks time.  Rank 0 prints ONE JSON line; the other figures the metric names (2^20 commit, NTT GB/s,
NTT+LDE+commit end to end) ride in `extra`.

roofline   : dominant kernel = k_leaf_subtree (one 2^K-leaf Merkle subtree per lane: Blake2b-512 leaf +
             node compressions).  It is integer-VALU bound, not HBM bound, so `roofline`
             prices it against the VALU issue peak and `roofline_hbm` gives the same launch against HBM.
cpu_baseline: the reference-faithful CPU port (oracle mode 1: I320 schoolbook multiply + Knuth division,
             ext-Euclid per fold element, sequential Blake2b Merkle — SURVEY.md §8d B1), 1 thread, on a
             bounded sample.  The Rust reference itself cannot be built (no cargo/rustc on this image).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import stark_anatomy_amd as sa  # noqa: E402  (fails loudly without the HIP extension)
from stark_anatomy_amd import _lib  # noqa: E402
from stark_anatomy_amd.context import _u8p, _u64p, scalar  # noqa: E402

# VALU issue peak derived from MI355X_MICROARCH.md: 157.3 TFLOP/s FP32 vector = 2 flop x 78.6e12 lane-ops/s
# (256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz); simple 32-bit integer VALU ops issue at that rate.
VALU_PEAK_OPS = 78.6e12
HBM_PEAK_BPS = 8.0e12
# Algorithmic 32-bit VALU ops per Blake2b-512 compression on CDNA4 (DESIGN.md "Roofline"):
# 96 G x (6 add64 x 2 + 4 xor64 x 2 + rotations 0+2+2+2) = 2496, + 24 for init/feed-forward.
OPS_NODE = 2520
# leaf: 14 of the 16 message words are zero, so 14 of the 16 message additions per round fold away.
OPS_LEAF = 2520 - 12 * 14 * 2


def random_elems(n, seed):
    """n canonical field elements: random u128 reduced mod p on the GPU (FieldElement::new, field.rs:30-35)."""
    rng = np.random.default_rng(seed)
    raw = np.frombuffer(rng.bytes(16 * n), dtype=np.uint64).reshape(n, 2)
    return raw


def dist_leg_main(args):
    """Child process of one rank of `bench.py --gpus N` (N > 1): one 2^log_n transform and one commit spread over all ranks."""
    import torch.distributed as dist
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("STARK_BENCH_DIST_LOCAL_RANK", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("STARK_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    log_n = args.log_n
    N = 1 << log_n
    try:
        from stark_anatomy_amd import dist_ntt as dn
        eng = dn.GpuEngine(local_rank)
        H = eng.ctx.handle
        m = N // world
        seed = np.frombuffer(np.random.default_rng(4242 + rank).bytes(16 * m), dtype=np.uint64).reshape(m, 2)
        red = np.empty_like(seed)
        _lib.check(_lib.lib.sh_fp_new(H, _u64p(np.ascontiguousarray(seed)), _u64p(red), m))
        x = eng.from_host(red)
        res = {}

        def t_max(fn, reps):
            fn()
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize(); dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()) / reps, r
        t_ntt, X = t_max(lambda: dn.dist_ntt(x, log_n, eng), 3)
        res[f"ntt_2^{log_n}_ms"] = 1e3 * t_ntt
        res[f"ntt_2^{log_n}_GBps"] = 32.0 * N / t_ntt / 1e9
        t_rows, _ = t_max(lambda: dn.dist_ntt(x, log_n, eng, output_layout="rows"), 3)
        res[f"ntt_2^{log_n}_rows_layout_ms"] = 1e3 * t_rows
        back = dn.dist_ntt(X, log_n, eng, inverse=True)
        with eng.scope():
            ok = bool(torch.equal(back, x))
        t_fri, (roots, last, _ps) = t_max(lambda: dn.dist_fri_commit(X, N, eng, sa.GENERATOR, 4, 17), 2)
        res[f"fri_commit_2^{log_n}_ms"] = 1e3 * t_fri
        res[f"fri_commit_2^{log_n}_elems_per_s"] = N / t_fri
        full = dn.all_gather_slices(X)
        torch.cuda.synchronize()
        r1, _l1 = sa.Fri(sa.GENERATOR, None, N, 4, 17, ctx=eng.ctx).commit(full, sa.ProofStream())
        same = [h.as_bytes() for h in r1] == [bytes(a) for a in roots]
        flag = torch.tensor([int(ok and same)], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res["round_trip_and_single_gpu_roots_match"] = bool(flag.item())
        res["ranks"] = world
    except Exception as e:  # reported, never fatal for the headline
        res = {"error": f"{type(e).__name__}: {e}"[:300]}
    print(json.dumps(res), flush=True)
    os._exit(0)                                                                # no teardown of a communicator that may be wedged


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=24, help="log2 of the FRI domain (codeword length)")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dist-leg", action="store_true", help="internal: the child process of the N > 1 strong-scaling leg")
    args = ap.parse_args()
    if args.dist_leg:
        return dist_leg_main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
            sys.exit(2)
    backend = os.environ.get("STARK_BENCH_BACKEND", "nccl")       # "gloo" + STARK_BENCH_SHARE_GPU=1: rehearsal of N ranks on one card
    if os.environ.get("STARK_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stream = torch.cuda.Stream(device=local_rank)
    ctx = sa.Context(local_rank, stream=stream.cuda_stream)
    H = ctx.handle
    log_n = args.log_n
    N = 1 << log_n

    # ---- synthetic codeword, resident in HBM before the timed region ----
    def resident_codeword(n, seed):
        raw = random_elems(n, seed)
        red = np.empty_like(raw)
        _lib.check(_lib.lib.sh_fp_new(H, _u64p(np.ascontiguousarray(raw)), _u64p(red), n))
        return ctx.alloc(n * 16).upload(red)

    params = _lib.FriParams()
    params.expansion, params.num_colinearity_tests = 4, 17
    params.offset[0], params.offset[1] = sa.GENERATOR & 0xFFFFFFFFFFFFFFFF, sa.GENERATOR >> 64

    import ctypes as C

    out_bufs = {}

    def commit(dbuf, n):
        if n not in out_bufs:                                  # output buffers once per size; a fresh (empty) transcript per commit
            R = int(_lib.lib.sh_fri_num_rounds(n, 4, 17))
            roots = np.zeros((max(R, 1), 64), dtype=np.uint8)
            last = np.empty((n >> max(R - 1, 0), 2), dtype=np.uint64)
            out_bufs[n] = (roots, last, _u8p(roots), _u64p(last), C.c_size_t(0))
        roots, last, p_roots, p_last, last_len = out_bufs[n]
        ps = sa.ProofStream()
        _lib.check(_lib.lib.sh_dev_fri_commit_proofstream(H, dbuf.ptr, n, C.byref(params), ps.handle, p_roots, p_last, C.byref(last_len)))
        return roots, ps

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    d_cw = resident_codeword(N, 1000 + rank)
    dt = timed(lambda: commit(d_cw, N), args.steps, args.warmup)
    value = world * N * args.steps / dt
    ms_per_step = 1e3 * dt / args.steps

    out = {
        "metric": "FRI commit field-elems/s", "value": value, "unit": "field-elems/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u128 (4x u32 Montgomery limbs) + u64 Blake2b",
        "data": "synthetic",
        "config": {"workload": f"full FRI commit of one resident 2^{log_n}-element codeword per GPU "
                               f"(expansion 4, 17 colinearity tests, offset g: {int(_lib.lib.sh_fri_num_rounds(N, 4, 17))} rounds, "
                               "Blake2b-512 Merkle roots, SHA-256/JSON transcript on host)",
                   "domain": N, "per_gpu_codewords": 1, "sharding": "independent commits per rank, no collective"},
    }

    extra = {}
    if not args.no_extra:
        # ---- dominant kernel, timed with events on the stream it is launched on ----
        K = max(0, min(3, log_n - 18))
        n_out = N >> K
        d_dig = ctx.alloc(n_out * 64)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        with torch.cuda.stream(stream):
            _lib.check(_lib.lib.sh_dev_merkle_subtrees(H, d_cw.ptr, n_out, K, d_dig.ptr))
            ev0.record(stream)
            for _ in range(reps):
                _lib.check(_lib.lib.sh_dev_merkle_subtrees(H, d_cw.ptr, n_out, K, d_dig.ptr))
            ev1.record(stream)
        torch.cuda.synchronize()
        k_ms = ev0.elapsed_time(ev1) / reps
        comp_leaf, comp_node = N, N - n_out
        ops = comp_leaf * OPS_LEAF + comp_node * OPS_NODE
        achieved = ops / (k_ms * 1e-3)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"k_leaf_subtree_log{log_n}")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "valu", "kernel": "k_leaf_subtree_staged", "achieved": achieved / 1e12, "peak": VALU_PEAK_OPS / 1e12,
                           "unit": "Tops/s (32-bit integer VALU lane-ops)", "frac": achieved / VALU_PEAK_OPS, "traffic": traffic,
                           "launch_ms": k_ms, "units_per_launch": {"leaves": N, "leaf_compressions": comp_leaf, "node_compressions": comp_node},
                           "ops_per_leaf_compression": OPS_LEAF, "ops_per_node_compression": OPS_NODE,
                           "compressions_per_s": (comp_leaf + comp_node) / (k_ms * 1e-3)}
        hbm_bytes = N * 16 + n_out * 64
        out["roofline_hbm"] = {"bound": "hbm", "kernel": "k_leaf_subtree_staged", "achieved": hbm_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_BPS / 1e9,
                               "unit": "GB/s", "frac": hbm_bytes / (k_ms * 1e-3) / HBM_PEAK_BPS, "traffic": traffic,
                               "algorithmic_bytes_per_launch": hbm_bytes}
        d_dig.free()

        # ---- NTT GB/s at the same size (32·n bytes = one read + one write per point) ----
        host_cw = d_cw.download(np.uint64, (N, 2))
        d_x = ctx.alloc(N * 16).upload(host_cw)
        ntt_steps = max(3, args.steps)
        dt_ntt = timed(lambda: (_lib.check(_lib.lib.sh_dev_ntt(H, d_x.ptr, log_n, 0)), ctx.sync()), ntt_steps, 1)
        extra[f"ntt_2^{log_n}_GBps"] = world * 32.0 * N * ntt_steps / dt_ntt / 1e9
        extra[f"ntt_2^{log_n}_ms"] = 1e3 * dt_ntt / ntt_steps

        # ---- north-star end to end: n = N/4 coefficients -> coset LDE (NTT) -> full commit ----
        d_coef = ctx.alloc((N // 4) * 16).upload(host_cw[:N // 4])

        def e2e():
            _lib.check(_lib.lib.sh_dev_coset_lde(H, d_coef.ptr, log_n - 2, 2, _u64p(scalar(sa.GENERATOR)), d_x.ptr))
            commit(d_x, N)
        e2e_steps = max(5, args.steps)
        dt_e2e = timed(e2e, e2e_steps, 2)
        extra[f"lde_plus_commit_2^{log_n}_elems_per_s"] = world * N * e2e_steps / dt_e2e
        extra[f"lde_plus_commit_2^{log_n}_ms"] = 1e3 * dt_e2e / e2e_steps
        d_x.free(); d_coef.free()

        # ---- BASELINE config 4 (per-GPU share): 128 independent 2^14 commits in lock step ----
        Bb, nb = 128, 1 << 14
        d_b = resident_codeword(Bb * nb, 3000 + rank)
        Rb = int(_lib.lib.sh_fri_num_rounds(nb, 4, 17))
        roots_b = np.zeros((Bb, Rb, 64), dtype=np.uint8)
        last_b = np.empty((Bb, nb >> (Rb - 1), 2), dtype=np.uint64)

        # fresh (empty) transcripts for every call, made before the clock starts: creating 128 Python objects is not the library's time.
        # The whole leg runs twice and the second pass is the one timed: the first pass's transcripts are freed before the second's
        # are made, so their items grow into heap memory that has been touched before, as in a process that commits batch after
        # batch (on never-touched heap the page faults of ~0.6 MB of transcript items add ~0.2 ms per call).
        n_calls = max(5, args.steps) + 2
        calls = [0]
        fresh_handles = []

        def batch_commit():
            ll = C.c_size_t(0)
            handles = fresh_handles[calls[0]]; calls[0] += 1
            _lib.check(_lib.lib.sh_dev_fri_commit_batch(H, d_b.ptr, Bb, nb, C.byref(params), handles, _u8p(roots_b), _u64p(last_b), C.byref(ll)))
        for timed_pass in (False, True):
            fresh = [[sa.ProofStream() for _ in range(Bb)] for _ in range(n_calls)]
            fresh_handles = [(_lib.ps_t * Bb)(*[p.handle for p in pss]) for pss in fresh]
            calls[0] = 0
            dt_b = timed(batch_commit, max(5, args.steps), 2)
            del fresh, fresh_handles
        # the front end of that batch: 128 polynomials of 2^12 coefficients -> 128 codewords of 2^14, one launch per NTT pass
        d_bc = resident_codeword(Bb * (nb >> 2), 3500 + rank)
        dt_l = timed(lambda: (_lib.check(_lib.lib.sh_dev_coset_lde_batch(H, d_bc.ptr, Bb, 12, 2, _u64p(scalar(sa.GENERATOR)), d_b.ptr)), ctx.sync()), 5, 2)
        extra["lde_batch_128x2^12_to_2^14_ms"] = 1e3 * dt_l / 5
        d_bc.free()
        extra["batch_128x2^14_elems_per_s"] = world * Bb * nb * max(5, args.steps) / dt_b
        extra["batch_128x2^14_ms"] = 1e3 * dt_b / max(5, args.steps)
        d_b.free()

        # ---- throughput at the two sizes the metric names: several resident codewords per call (lock-step groups on two streams,
        # one group's latency-bound tree tops and hand-offs under the other's wide kernels) ----
        for Bt, lt in ((8, 20), (2, 24)):
            if lt > log_n:
                continue
            nt = 1 << lt
            d_t = resident_codeword(Bt * nt, 6000 + lt + rank)
            Rt = int(_lib.lib.sh_fri_num_rounds(nt, 4, 17))
            roots_t = np.zeros((Bt, Rt, 64), dtype=np.uint8)
            last_t = np.empty((Bt, nt >> (Rt - 1), 2), dtype=np.uint64)
            st = max(4, args.steps // 2)
            pss = [[sa.ProofStream() for _ in range(Bt)] for _ in range(st + 2)]
            hs = [(_lib.ps_t * Bt)(*[p.handle for p in row]) for row in pss]
            it = [0]

            def tp_commit():
                ll = C.c_size_t(0)
                h = hs[it[0]]; it[0] += 1
                _lib.check(_lib.lib.sh_dev_fri_commit_batch(H, d_t.ptr, Bt, nt, C.byref(params), h, _u8p(roots_t), _u64p(last_t), C.byref(ll)))
            dt_t = timed(tp_commit, st, 2)
            extra[f"fri_commit_{Bt}x2^{lt}_elems_per_s"] = world * Bt * nt * st / dt_t
            extra[f"fri_commit_{Bt}x2^{lt}_ms"] = 1e3 * dt_t / st
            del pss, hs
            d_t.free()

        # ---- the other domain the metric names ----
        other = 20 if log_n != 20 else 24
        d_o = resident_codeword(1 << other, 2000 + rank)
        so = max(args.steps, 10) if other == 20 else max(3, args.steps // 2)
        dt_o = timed(lambda: commit(d_o, 1 << other), so, 2)
        extra[f"fri_commit_2^{other}_elems_per_s"] = world * (1 << other) * so / dt_o
        extra[f"fri_commit_2^{other}_ms"] = 1e3 * dt_o / so
        d_o.free()

        # ---- the literal drop-in call: codeword in pageable HOST memory (PCIe-inclusive; never `value`) ----
        for lh in (24, 20):
            if lh > log_n:
                continue
            nh = 1 << lh
            raw_h = random_elems(nh, 7000 + lh + rank).copy()
            raw_h[:, 1] &= np.uint64(0x7FFFFFFFFFFFFFFF)                 # canonical (< p) without a device pass
            Rh = int(_lib.lib.sh_fri_num_rounds(nh, 4, 17))
            roots_h = np.zeros((Rh, 64), dtype=np.uint8)
            last_h = np.empty((nh >> (Rh - 1), 2), dtype=np.uint64)

            def host_commit():
                ll = C.c_size_t(0)
                ps = sa.ProofStream()
                _lib.check(_lib.lib.sh_fri_commit_proofstream(H, _u64p(raw_h), nh, C.byref(params), ps.handle, _u8p(roots_h), _u64p(last_h), C.byref(ll)))
            dt_h = timed(host_commit, 4, 2)
            extra[f"fri_commit_host_2^{lh}_ms"] = 1e3 * dt_h / 4
            del raw_h

        # ---- config 1 widened: full prove (commit + query phase) and host verify at 2^20, on a low-degree codeword ----
        lp = 20
        d_pc = resident_codeword(1 << (lp - 2), 4000 + rank)
        d_pw = ctx.alloc((1 << lp) * 16)
        _lib.check(_lib.lib.sh_dev_coset_lde(H, d_pc.ptr, lp - 2, 2, _u64p(scalar(sa.GENERATOR)), d_pw.ptr))
        fri_p = sa.Fri(sa.GENERATOR, None, 1 << lp, 4, 17, ctx=ctx)
        box_ps = []

        def prove():
            box_ps[:] = [sa.ProofStream()]
            fri_p.prove(d_pw, box_ps[0])
        dt_p = timed(prove, 5, 1)
        extra[f"fri_prove_2^{lp}_ms"] = 1e3 * dt_p / 5
        ser = box_ps[0].serialize()
        t0 = time.perf_counter()
        okv, _why = fri_p.verify(sa.ProofStream.deserialize(ser))
        extra[f"fri_verify_host_2^{lp}_ms"] = 1e3 * (time.perf_counter() - t0)
        extra[f"fri_proof_2^{lp}_bytes"] = len(ser)
        extra[f"fri_proof_2^{lp}_accepted"] = bool(okv)
        d_pc.free(); d_pw.free()

        # ---- polynomial layer: batched inversion and a product through the NTT, device resident ----
        li = 24 if log_n >= 24 else log_n
        d_iv = resident_codeword(1 << li, 5000 + rank)
        d_io = ctx.alloc((1 << li) * 16)
        dt_i = timed(lambda: (_lib.check(_lib.lib.sh_dev_fp_inv(H, d_iv.ptr, d_io.ptr, 1 << li))), 5, 1)
        extra[f"fp_inv_2^{li}_ms"] = 1e3 * dt_i / 5
        extra[f"fp_inv_2^{li}_elems_per_s"] = world * (1 << li) * 5 / dt_i
        hp = 1 << (li - 1)
        d_pm = ctx.alloc(2 * hp * 16)
        dt_m = timed(lambda: (_lib.check(_lib.lib.sh_dev_poly_mul(H, d_iv.ptr, hp, d_iv.ptr + hp * 16, hp, d_pm.ptr)), ctx.sync()), 5, 1)
        extra[f"poly_mul_2^{li - 1}x2^{li - 1}_ms"] = 1e3 * dt_m / 5
        d_iv.free(); d_io.free(); d_pm.free()
    out["extra"] = extra

    # ---- N > 1 only: ONE 2^log_n transform / commit spread over all ranks (BASELINE config 5; strong scaling, RCCL
    # all-to-all over xGMI).  Reported beside the headline, never as `value`.  Runs in a CHILD process per rank (its own process
    # group on its own port): a collective that stalls, or a crash inside the communication library, costs this leg only — the
    # parent kills its own child at the time limit, reports the error in the line and still prints the headline.
    if world > 1 and not args.no_extra and os.environ.get("STARK_BENCH_DIST", "1") != "0":
        import socket
        import subprocess
        port_t = torch.zeros(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        if rank == 0:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port_t[0] = sk.getsockname()[1]
        dist.broadcast(port_t, 0)
        env = dict(os.environ)
        env["MASTER_PORT"] = str(int(port_t.item()))
        env["STARK_BENCH_DIST_LOCAL_RANK"] = str(local_rank)
        for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE"):      # the child makes a plain env:// rendezvous on its own port
            env.pop(k, None)
        limit = float(os.environ.get("STARK_BENCH_DIST_TIMEOUT", "150"))
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--dist-leg", "--log-n", str(log_n), "--gpus", str(world)],
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            c_out, c_err = child.communicate(timeout=limit)
            res = None
            for line in reversed(c_out.splitlines()):
                if line.startswith("{"):
                    res = json.loads(line)
                    break
            if res is None:
                res = {"error": f"distributed leg left with status {child.returncode}: {c_err.strip()[-300:]}"}
        except subprocess.TimeoutExpired:
            child.kill()                                                        # this rank's own child, by handle
            child.communicate()
            res = {"error": f"watchdog: distributed leg did not finish within {limit:.0f} s"}
        out["extra"]["dist_one_codeword_over_all_gpus"] = res
        if "error" in res:
            out["dist_error"] = res["error"]
        elif res.get("round_trip_and_single_gpu_roots_match") is False:
            out["dist_error"] = "distributed NTT round trip or distributed commit roots differ from the single-GPU result"

    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import coracle as co      # checker used as the CPU baseline leg only
        log_s = 18
        cw = co.splitmix_elems(3, 1 << log_s)
        t0 = time.perf_counter()
        co.fri_commit(cw, 4, 17, sa.GENERATOR, mode=1)
        t1 = time.perf_counter() - t0
        log_o = 20                                                   # optimized port (B2): larger sample, 1 thread and all of this box's share
        cwo = co.splitmix_elems(4, 1 << log_o)
        t0 = time.perf_counter()
        co.fri_commit(cwo, 4, 17, sa.GENERATOR, mode=0)
        t2 = time.perf_counter() - t0
        threads = max(1, min(16, os.cpu_count() or 1))
        co.set_threads(threads)
        t0 = time.perf_counter()
        co.fri_commit(cwo, 4, 17, sa.GENERATOR, mode=0)
        t3 = time.perf_counter() - t0
        co.set_threads(1)
        out["cpu_baseline"] = {"value": (1 << log_s) / t1, "unit": "field-elems/s", "cores": 1, "kind": "port",
                               "sample": f"one full FRI commit at N=2^{log_s} (same parameters), reference-faithful port: I320 5-limb "
                                         "multiply + Knuth division per field op, ext-Euclid per fold element, sequential Blake2b Merkle "
                                         f"({t1:.1f} s); the Rust reference cannot be built here (no cargo/rustc)",
                               "optimized_port_1core_elems_per_s": (1 << log_o) / t2,
                               "optimized_port_threads": threads, "optimized_port_threaded_elems_per_s": (1 << log_o) / t3,
                               "optimized_port_sample": f"one full FRI commit at N=2^{log_o}: u128 field with 407*2^119 = -1 reduction, one inverse per "
                                                        "round, leaf hashing / Merkle levels / fold split over host threads",
                               "host_cpus": os.cpu_count()}

    if rank == 0:
        print(json.dumps(out), flush=True)
    d_cw.free()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
