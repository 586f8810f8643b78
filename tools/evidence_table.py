#!/usr/bin/env python3
"""Turn gpurun_out/evidence/ (tools/refresh_evidence.sh) into the committed tables under profiles/:
   <prefix>_bench_kernel_stats.csv            rocprofv3 --stats of bench.py (roofline.kernel must appear in it)
   <prefix>_pmc_<workload>.csv                per-kernel counter sums, the four --pmc passes merged
   <prefix>_kernel_table_<workload>.md        SURVEY.md §8d: kernel, calls, avg us, algorithmic bytes, counter bytes
                                              (2·FETCH_SIZE + WRITE_SIZE, KiB -> bytes), GB/s, % of 8 TB/s, VALUBusy %, % of VALU peak
   <prefix>_timeline_<workload>.txt           kernel-by-kernel timeline of the workload
usage: evidence_table.py [prefix=r03] [evidence dir]"""
import collections, csv, glob, os, shutil, sys

prefix = sys.argv[1] if len(sys.argv) > 1 else "r03"
EV = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/evidence"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
HBM_PEAK, VALU_PEAK, SIMDS, XCDS = 8.0e12, 78.6e12, 1024, 8


def one(pattern):
    g = glob.glob(os.path.join(EV, pattern), recursive=True)
    return g[0] if g else None


def ilog2(x): return x.bit_length() - 1
def leaf_k(log_total, lg): return min(max(0, min(3, log_total - 18)), lg)


def forest_step(lg, B):
    total = B << lg
    if total > 1 << 17:
        return min(min(ilog2(total) - 17, 3), lg), "k_node_subtree"
    return (lg if lg <= 9 else (9 if total >= 1 << 14 else (lg + 1) // 2)), "k_quad_reduce"


def commit_plan(log_N, B, alg):
    """algorithmic bytes per kernel name for one commit of B x 2^log_N, following api_fri.hip's launch plan"""
    N = 1 << log_N
    R, L = 0, N
    while L > 4 and 4 * 17 < L:
        L //= 2; R += 1
    for r in range(R):
        lg = log_N - r
        total = B << lg
        small = total <= 1 << 15
        fold = r > 0
        if small:
            K = min(lg, 7)
            alg["k_round_wg"] += (32 * total if fold else 16 * total) + (16 * total if fold else 0) + (64 * (total >> 7) if lg > 7 else 64 * B)
        else:
            K = leaf_k(ilog2(total), lg)
            staged = K >= 1 and ((total >> K) & 255) == 0 and (not fold or lg >= 6 + K)      # hash_kernels.hip: launch_*_subtree
            name = ("k_fold_leaf_subtree" if fold else "k_leaf_subtree") + ("_staged" if staged else "")
            alg[name] += (32 * total + 16 * total if fold else 16 * total) + 64 * (total >> K)
        rem = lg - K
        while rem > 0:
            st, name = forest_step(rem, B)
            alg[name] += 64 * (B << rem) + 64 * (B << (rem - st))
            rem -= st
    alg["k_copy_to_host"] += 32 * B * (N >> (R - 1))


def workload_alg(log_n, B):
    alg = collections.defaultdict(int)
    commit_plan(log_n, B, alg)
    if B == 1:                                         # LDE in front: scale + pad, then the NTT passes (32·N bytes each)
        N = 1 << log_n
        alg["k_coset_scale_pad"] += 16 * (N >> 2) + 16 * N
        alg["k_ntt_pass"] = -32 * N                    # per call: multiplied by the call count below
        alg["k_ntt_pass_reg"] = -32 * N                # the register-exchange form of the same pass (all template instances under one name)
    return alg


def agg_pmc(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for sub in ("fetch", "write", "sq", "mix"):
        f = one(f"pmc_{tag}_{sub}/**/*counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


def table(tag, log_n, B, title):
    tr = one(f"trace_{tag}/**/*kernel_trace.csv")
    if not tr:
        return
    rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
    dur, calls = collections.defaultdict(float), collections.defaultdict(int)
    with open(os.path.join(PROF, f"{prefix}_timeline_{tag}.txt"), "w") as f:
        f.write(f"# {title}: kernel timeline (rocprofv3 --kernel-trace of tools/pmc_workload.py {log_n} {B}); gap = idle time before the kernel\n")
        prev = None
        for r in rows:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            dur[name] += e - s; calls[name] += 1
            f.write(f"{name:28s} grid={int(r['Grid_Size_X']):9d} wg={int(r['Workgroup_Size_X']):5d} dur={(e - s) / 1e3:9.1f} us gap={((s - prev) / 1e3 if prev else 0):8.1f} us\n")
            prev = e
        f.write(f"sum of kernel time {sum(dur.values()) / 1e3:.1f} us, span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3:.1f} us\n")
    pmc = agg_pmc(tag)
    names = sorted(pmc_names := {c for k in pmc for c in pmc[k]})
    with open(os.path.join(PROF, f"{prefix}_pmc_{tag}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls"] + names)
        for k in sorted(pmc):
            w.writerow([k, calls.get(k, 0)] + [int(pmc[k].get(c, 0)) for c in names])
    alg = workload_alg(log_n, B)
    with open(os.path.join(PROF, f"{prefix}_kernel_table_{tag}.md"), "w") as f:
        f.write(f"# {title}\n\nrocprofv3 on MI355X: durations from `--kernel-trace`, counters from four separate `--pmc` passes of the same command "
                f"(`python3 tools/pmc_workload.py {log_n} {B}`).  Counter bytes = 2·FETCH_SIZE + WRITE_SIZE (KiB; the gfx950 FETCH correction of "
                "MI355X_MICROARCH.md).  VALUBusy = 100·SQ_ACTIVE_INST_VALU·4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) (rocprof's formula).  "
                "% of VALU peak = SQ_INSTS_VALU·64 lane-ops / time / 78.6 T lane-ops/s.  Algorithmic bytes follow the launch plan of api_fri.hip "
                "(tools/evidence_table.py); '-' where a kernel moves no bulk data.  STARK_HIP_CHAIN_SPIN_US is raised for these runs so that no wait kernel gives up "
                "and re-launches a round while the profiler slows the host.\n\n")
        f.write("| kernel | calls | total us | avg us | algorithmic MB | counter MB | alg GB/s | % of 8 TB/s | VALUBusy % | % of VALU peak | wait-for-issue % | parked % |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k in sorted(dur, key=lambda k: -dur[k]):
            t = dur[k] * 1e-9
            a = alg.get(k, 0)
            if a < 0:
                a = -a * calls[k]
            c = pmc.get(k, {})
            cb = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
            cyc = c.get("GRBM_GUI_ACTIVE", 0) / XCDS
            vb = 100 * c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / SIMDS / cyc if cyc else 0
            vp = 100 * c.get("SQ_INSTS_VALU", 0) * 64 / t / VALU_PEAK if t else 0
            wc = c.get("SQ_WAVE_CYCLES", 0)
            f.write(f"| {k} | {calls[k]} | {dur[k] / 1e3:.1f} | {dur[k] / 1e3 / calls[k]:.1f} | {(f'{a / 1e6:.1f}' if a else '-')} | {cb / 1e6:.1f} | "
                    f"{(f'{a / t / 1e9:.0f}' if a else '-')} | {(f'{100 * a / t / HBM_PEAK:.1f}' if a else '-')} | {vb:.0f} | {vp:.0f} | "
                    f"{(100 * c.get('SQ_WAIT_INST_ANY', 0) / wc if wc else 0):.0f} | {(100 * c.get('SQ_WAIT_ANY', 0) / wc if wc else 0):.0f} |\n")
        f.write(f"\nsum of kernel time without k_wait_c0: {sum(v for k, v in dur.items() if k != 'k_wait_c0') / 1e3:.1f} us.  A single codeword's rounds have no k_wait_c0 "
                "any more (late round 3): the first kernel of a round (k_fold_leaf_subtree*, k_round_wg) waits for the host's challenge itself, so its duration here includes "
                "the host's turn — which under the profiler, in the first call of a process, is many times slower than in a timed run (there: ~4 us per round).  Batches "
                "(w14x128) keep k_wait_c0.\n")


b = one("bench/**/*kernel_stats.csv")
if b:
    shutil.copy(b, os.path.join(PROF, f"{prefix}_bench_kernel_stats.csv"))
bt = one("bench/**/*kernel_trace.csv")
if bt:
    # the roofline kernel of bench.py split by grid size, and the roofline leg's own back-to-back launches
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])) for r in csv.DictReader(open(bt))), key=lambda r: r[0])
    by = {}
    for s0, e0, name, grid in rows:
        if "k_leaf_subtree_staged<3>" in name:
            by.setdefault(grid, []).append((e0 - s0) / 1e3)
    best, run = [], []
    for s0, e0, name, grid in rows:
        if "k_leaf_subtree_staged<3>" in name and grid == 2097152:
            run.append((e0 - s0) / 1e3)
        else:
            if len(run) > len(best):
                best = run
            run = []
    if len(run) > len(best):
        best = run
    with open(os.path.join(PROF, f"{prefix}_bench_roofline_kernel_by_grid.csv"), "w") as f:
        f.write(f"# {prefix}: k_leaf_subtree_staged<3> in the rocprofv3 --kernel-trace of `python3 bench.py --steps 5 --warmup 2 --no-cpu` (the run behind {prefix}_bench_kernel_stats.csv),\n"
                "# split by grid size.  The --stats summary averages over every launch of the kernel name; bench.py launches it at several sizes (the 2^24 headline and\n"
                "# roofline launches: 2^21 lanes; the 2 x 2^24 batch: 2^22 lanes; 2^20..2^22 commits; and one launch per 8 MiB chunk of the host-buffer commits), so the\n"
                "# roofline launch (grid 2097152 threads = 2^24 leaves, K = 3) is listed on its own here: its average is what roofline.launch_ms is compared with\n"
                "# (profiled runs clock lower than the event-timed launches of a plain run, MI355X_MICROARCH.md on DVFS).\n"
                "grid_threads,calls,avg_us,min_us,max_us\n")
        for grid in sorted(by):
            v = by[grid]
            f.write(f"{grid},{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}\n")
        if best:
            f.write(f"# the {len(best)} back-to-back launches of bench.py's roofline leg (sh_dev_merkle_subtrees, nothing else on the GPU): us each\n")
            f.write("roofline_leg," + ",".join(f"{x:.1f}" for x in best) + "\n")
            f.write(f"roofline_leg_avg_us,{sum(best) / len(best):.1f}\n")
table("w24x1", 24, 1, "coset LDE 2^22 -> 2^24 + FRI commit 2^24")
table("w20x1", 20, 1, "coset LDE 2^18 -> 2^20 + FRI commit 2^20")
table("w14x128", 14, 128, "batched FRI commit 128 x 2^14")
print(open(os.path.join(PROF, f"{prefix}_kernel_table_w24x1.md")).read() if os.path.exists(os.path.join(PROF, f"{prefix}_kernel_table_w24x1.md")) else "no tables")
