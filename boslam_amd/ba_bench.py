"""Secondary metric of BASELINE.json: local-BA Levenberg iterations per second on configs[2]
(50 keyframes x 10 k landmarks x ~80 k observations, 10 LM iterations, Huber sqrt(7.815), f64)."""
import os
import time
import numpy as np
from . import ba, synth


_PMC_REL = "profiles/r05_ba_pmc.json"


def _pmc(workload):
    """Per-launch hardware counters of the BA kernels (rocprofv3 --pmc, tools/prof_r05.sh -> tools/ba_pmc.py; the committed summary).
    bench.py cannot run the profiler inside its timed run: the counters per launch are a property of the kernel and the workload, the
    durations next to them are measured live.  A summary collected on another build of ba.hip is not used (see _profiles): the blocks
    derived from it say `stale` instead."""
    from . import _profiles
    d, why = _profiles.load(_PMC_REL, ["ba.hip"])
    if d is None:
        return {}
    if why:
        return {"_stale": why}
    return d.get(workload, {})


def _counter_block(pm, names, live_ms):
    """Counter flops / bytes of the named kernels (per launch, summed over `names` = (kernel, launches per iteration)) against the live time."""
    if "_stale" in pm:
        from . import _profiles
        return _profiles.stale_block(_PMC_REL, pm["_stale"])
    fl = rd = rdu = wr = busy = 0.0
    have = False
    for k, n in names:
        e = pm.get(k)
        if not e:
            continue
        have = True
        d = e["derived"]
        fl += n * d.get("mfma_f64_flops", 0.0); rd += n * d.get("hbm_read_bytes", 0.0); rdu += n * d.get("hbm_read_bytes_uncorrected", 0.0)
        wr += n * d.get("hbm_write_bytes", 0.0); busy += n * e["counters"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if not have or live_ms <= 0:
        return None
    t = live_ms * 1e-3
    return {"source": _PMC_REL + " (rocprofv3 --pmc, counters per launch) over the live kernel time",
            "mfma_f64_flops": int(fl), "mfma_tflops": round(fl / t / 1e12, 5), "mfma_frac_of_peak": round(fl / t / 78.6e12, 6),
            "mfma_busy_simds": round(busy / (t * 2.4e9), 3),
            "hbm_bytes": int(rd + wr), "hbm_bytes_fetch_uncorrected": int(rdu + wr), "hbm_gbps": round((rd + wr) / t / 1e9, 1),
            "hbm_frac_of_peak": round((rd + wr) / t / 8e12, 4)}


def run(ctx, dev=None, repeats=5):
    pr = synth.ba_problem()
    s = ba.LocalBA(ctx, pr["K"])
    times, setups, iters = [], [], 0
    for r in range(repeats + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])      # graph structure built on the device: timed
        ctx.synchronize()
        t1 = time.perf_counter()
        iters = s.optimize(10)
        ctx.synchronize()
        if r:
            times.append(time.perf_counter() - t0); setups.append(t1 - t0)
    ctx.timing(True)                                   # one more pass with per-kernel hipEvent timing
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    s.optimize(10)
    kt = ctx.timing_read()
    ctx.timing(False)
    lg = s.log()[:iters]
    _, t = s.poses()
    out = {"metric": "local_ba_lm_iters_per_sec", "value": round(iters / float(np.median(times)), 1), "unit": "iters/s",
           "iters": int(iters), "ms_per_iter": round(float(np.median(times)) / iters * 1e3, 3), "dtype": "f64",
           "set_problem_ms": round(float(np.median(setups)) * 1e3, 3),
           "timed_region": "bh_ba_set_problem (structure built on the device, host arrays in) + 10 LM iterations",
           "workload": f"configs[2]: {len(pr['R'])} keyframes ({int(pr['fixed'].sum())} fixed) x {len(pr['pts'])} landmarks x "
                       f"{len(pr['ecam'])} observations, 10 LM iterations",
           "final_robust_chi2": float(lg[-1, 0]), "trials": int(lg[:, 2].sum()),
           "kernel_ms_per_iter": {k: round(v[0] / iters, 4) for k, v in sorted(kt.items()) if k.startswith("k_ba")},
           "pose_err_before_m": float(np.abs(pr["t"] - pr["tgt"]).max()), "pose_err_after_m": float(np.abs(t - pr["tgt"]).max())}
    # matrix-core roofline of the reduced-camera solve, for the solver bh_ba_set_problem actually selected (bh_ba_solver):
    # v_mfma_f64_16x16x4_f64 = 2 048 flop.  "dense" (k_ba_chol_dense<false>, n <= 288, the whole factor on one CU): per block
    # column J of NT = ceil(n/16): the NT-1-J tiles below the diagonal take 4 MFMAs each for L(i,J) = T(i,J) Linv_J', and the
    # (NT-1-J)(NT-J)/2 lower tiles of the trailing matrix (diagonal ones included) 4 MFMAs each.  "panel" (k_ba_chol<NB>): per
    # NB-column panel nt(nt+1)/2 trailing tiles x NB/4 MFMAs.
    n6 = 6 * int((1 - pr["fixed"]).sum())
    solver = s.solver()
    flops = 0
    if solver == "dense":
        NT = (n6 + 15) // 16
        for J in range(NT):
            below = NT - 1 - J
            flops += (below * 4 + below * (below + 1) // 2 * 4) * 2048
        kname = "k_ba_chol_dense<false>"
        note = ("a %d x %d dense system factored on ONE CU (1024-thread workgroup, 16 x 16 f64 tiles in MFMA accumulator layout): %d dependent block "
                "columns, latency-bound; the iteration as a whole is launch/latency-bound (2.4 MB of algorithmic traffic)" % (n6, n6, NT))
    else:
        NB = 16
        for j0 in range(0, n6, NB):
            nt = (max(n6 - j0 - NB, 0) + 15) // 16
            flops += nt * (nt + 1) // 2 * (NB // 4) * 2048
        kname = "k_ba_chol<16>" if solver == "panel" else solver
        note = "a %d x %d system, solver '%s': dependent panels on one CU, latency-bound" % (n6, n6, solver)
    ms = out["kernel_ms_per_iter"].get("k_ba_solve", 0.0)
    out["solver"] = solver
    if ms > 0:
        out["solve_roofline"] = {"bound": "mfma", "kernel": kname, "achieved": round(flops / (ms * 1e-3) / 1e12, 6), "unit": "TFLOP/s",
                                 "peak": 78.6, "frac": round(flops / (ms * 1e-3) / 78.6e12, 7), "cus_used": 1,
                                 "frac_of_cus_used": round(flops / (ms * 1e-3) / (78.6e12 / 256), 4), "mfma_flops_per_solve": int(flops),
                                 "note": note}
        pm = _pmc("local")
        # (a local window runs as a batch of one: the profiler sees the batched kernels' names)
        cb = _counter_block(pm, [("k_bb_chol" if "k_bb_chol" in pm else kname, 1)], ms)
        if cb:
            out["solve_roofline"]["counters"] = cb
    # the Schur complement (S = Hpp - sum Hpl Hll^-1 Hpl', b~): the iteration's HBM-side stage.  Algorithmic bytes: per edge the 6 x 3 Hpl
    # block (144 B), per landmark Hll^-1 and bl (96 B), S out once; the counter bytes say how much of the gathered re-reads reach the fabric.
    sms = out["kernel_ms_per_iter"].get("k_ba_schur", 0.0)
    if sms > 0:
        E, P = len(pr["ecam"]), len(pr["pts"])
        alg = E * 144 + P * 96 + n6 * n6 * 8
        out["schur_roofline"] = {"bound": "hbm", "kernel": "k_ba_dinv + k_ba_schur_Sb", "algorithmic_bytes": int(alg), "ms": sms,
                                 "achieved": round(alg / (sms * 1e-3) / 1e9, 1), "unit": "GB/s", "peak": 8000, "frac": round(alg / (sms * 1e-3) / 8e12, 5)}
        pml = _pmc("local")
        cb = _counter_block(pml, [("k_bb_dinv", 1), ("k_bb_schur", 1)] if "k_bb_schur" in pml else [("k_ba_dinv", 1), ("k_ba_schur_Sb", 1)], sms)
        if cb:
            out["schur_roofline"]["counters"] = cb
            if not cb.get("stale"):
                out["schur_roofline"]["traffic_over_algorithmic"] = round(cb["hbm_bytes"] / alg, 2)
    s.close()
    return out


def run_batch(ctx, W=64, n_kf=50, n_pts=10000, iters=10, repeats=3, n_distinct=8, label="configs[2]"):
    """W local windows optimised together (bh_ba_batch_optimize): aggregate LM iterations/s, and the same windows one after the other
    through bh_ba_optimize for the ratio.  The windows cycle through `n_distinct` different problems of the given size."""
    probs = [synth.ba_problem(n_kf=n_kf, n_pts=n_pts, seed=7 + k) for k in range(n_distinct)]
    wins = [ba.LocalBA(ctx, probs[w % n_distinct]["K"]) for w in range(W)]

    args = [tuple(probs[w % n_distinct][k] for k in ("R", "t", "fixed", "pts", "ecam", "ept", "meas")) for w in range(W)]

    def set_all():                                        # bh_ba_batch_set_problems: the local windows are built by one launch, a workgroup each
        ba.set_problems(ctx, wins, args)
    batch = ba.LocalBABatch(ctx, wins)
    t_set, t_opt, done = [], [], None
    for r in range(repeats + 1):
        ctx.synchronize(); t0 = time.perf_counter()
        set_all()
        ctx.synchronize(); t1 = time.perf_counter()
        done = batch.optimize(iters)
        ctx.synchronize(); t2 = time.perf_counter()
        if r:
            t_set.append(t1 - t0); t_opt.append(t2 - t1)
    # one window at a time, same problems (bh_ba_set_problem + bh_ba_optimize per window)
    t_one, t_set1 = [], []
    for r in range(2):
        ctx.synchronize(); t0 = time.perf_counter()
        for s, a in zip(wins[:8], args[:8]):
            s.set_problem(*a)
        ctx.synchronize(); t1 = time.perf_counter()
        for s in wins[:8]:
            s.optimize(iters)
        ctx.synchronize()
        if r:
            t_one.append((time.perf_counter() - t1) / 8); t_set1.append((t1 - t0) / 8)
    tot = int(done.sum())
    opt = float(np.median(t_opt)); st = float(np.median(t_set)); one = float(np.median(t_one))
    chi = [float(s.log()[:int(done[w])][-1, 0]) for w, s in enumerate(wins[:n_distinct])]
    out = {"metric": "local_ba_lm_iters_per_sec_batched", "windows": W, "workload": f"{label}: {W} windows of {n_kf} keyframes x {n_pts} landmarks x "
           f"{len(probs[0]['ecam'])} observations, {iters} LM iterations each ({n_distinct} distinct problems)",
           "value": round(tot / opt, 1), "unit": "iters/s", "ms_per_round": round(opt / max(int(done.max()), 1) * 1e3, 3),
           "value_incl_set_problem": round(tot / (opt + st), 1), "set_problem_ms_per_window": round(st / W * 1e3, 4),
           "single_window_iters_per_s": round(iters / one, 1), "speedup_vs_one_at_a_time": round((tot / opt) / (iters / one), 1),
           "single_window_set_problem_ms": round(float(np.median(t_set1)) * 1e3, 3),
           "single_window_iters_per_s_incl_set_problem": round(iters / (one + float(np.median(t_set1))), 1),
           "speedup_incl_set_problem": round((tot / (opt + st)) / (iters / (one + float(np.median(t_set1)))), 1),
           "iters_done_min_max": [int(done.min()), int(done.max())], "final_robust_chi2_first_windows": chi}
    pmb = _pmc("batch") if (W == 64 and n_kf == 50) else {}
    if "_stale" in pmb:
        from . import _profiles
        out["schur_counters"] = _profiles.stale_block(_PMC_REL, pmb["_stale"])
    elif "k_bb_schur" in pmb:                              # the round's largest kernel: how much of its gathered Hpl traffic reaches the fabric (profile run, 64 windows)
        e = pmb["k_bb_schur"]; d = e["derived"]
        out["schur_counters"] = {"source": _PMC_REL, "kernel": "k_bb_schur", "us_per_round": e["mean_us"],
                                 "hbm_bytes_per_round": int(d["hbm_read_bytes"] + d["hbm_write_bytes"]),
                                 "hbm_bytes_per_round_fetch_uncorrected": int(d["hbm_read_bytes_uncorrected"] + d["hbm_write_bytes"]),
                                 "hbm_gbps": d["hbm_gbps"], "hbm_frac_of_peak": round(d["hbm_gbps"] / 8000.0, 4),
                                 "algorithmic_bytes_per_round": int(W * (len(probs[0]["ecam"]) * 144 + n_pts * 96))}
    batch.close()
    for s in wins:
        s.close()
    return out


def run_global(ctx, n_kf=5000, n_pts=500000, iters=5, window=20, dist=None, device=None):
    """BASELINE configs[4]: 5 000 keyframes x 500 k landmarks (~3.9 M observations, band +-2*window keyframes); with `dist`
    the landmarks are sharded over the ranks (strong scaling: the problem is fixed) and S / b~ are all-reduced over RCCL.
    The dense oracle cannot hold this size; correctness is carried by the banded-vs-dense and sharded-vs-single parity
    tests plus the properties reported here (monotone chi2, every step accepted, errors shrink)."""
    pr = synth.ba_problem_sequence(n_kf=n_kf, n_pts=n_pts, window=window)
    world = dist.get_world_size() if dist is not None else 1
    s = ba.GlobalBA(ctx, pr["K"], dist, device) if dist is not None else ba.LocalBA(ctx, pr["K"])
    t0 = time.perf_counter()
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    ctx.synchronize()
    t_setup = time.perf_counter() - t0
    s.optimize(1)                                                                      # warm-up
    t0 = time.perf_counter()
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    ctx.synchronize()
    t_setup2 = time.perf_counter() - t0                                                # buffers exist: uploads (host arrays in) + the structure build
    t0 = time.perf_counter()
    done = s.optimize(iters)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    comm_ms, comm_bytes = s.comm() if dist is not None else (0.0, 0)
    ctx.timing(True)
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    s.optimize(iters)
    kt = ctx.timing_read()
    ctx.timing(False)
    lg = s.log()[:done]
    _, t = s.poses(); pts = s.points_all() if dist is not None else s.points()
    out = {"metric": "global_ba_lm_iters_per_sec", "n_gpus": world, "scaling": "strong",
           "allreduce_ms_per_iter": round(comm_ms / max(done, 1), 3), "allreduce_mb_per_iter": round(comm_bytes / max(done, 1) / 1e6, 2), "value": round(done / dt, 2), "unit": "iters/s", "iters": int(done),
           "ms_per_iter": round(dt / done * 1e3, 2), "dtype": "f64", "set_problem_s": round(t_setup, 2), "set_problem_ms_warm": round(t_setup2 * 1e3, 2),
           "workload": f"configs[4]: {n_kf} keyframes x {n_pts} landmarks x {len(pr['ecam'])} observations, band +-{2 * window} keyframes",
           "robust_chi2": [float(v) for v in lg[:, 0]], "accepted": [int(v) for v in lg[:, 3]], "trials": int(lg[:, 2].sum()),
           "kernel_ms_per_iter": {k: round(v[0] / done, 3) for k, v in sorted(kt.items()) if k.startswith("k_ba")},
           "pose_err_mean_before_m": float(np.abs(pr["t"] - pr["tgt"]).mean()), "pose_err_mean_after_m": float(np.abs(t - pr["tgt"]).mean()),
           "point_err_mean_before_m": float(np.abs(pr["pts"] - pr["Xgt"]).mean()), "point_err_mean_after_m": float(np.abs(pts - pr["Xgt"]).mean())}
    # matrix-core roofline of the reduced-camera-system solve (the only GEMM-shaped stage): the sliding-window kernel issues 4
    # v_mfma_f64_16x16x4 per 16x16 tile, D(D+1)/2 tiles per block column (D = band in tiles); it runs on ONE CU by design
    n6 = 6 * int((1 - pr["fixed"]).sum()); bw = min(n6 - 1, 6 * 2 * window + 5); D = (bw + 15) // 16; nblk = (n6 + 15) // 16
    flops = nblk * D * (D + 1) // 2 * 4 * 2048
    ms = out["kernel_ms_per_iter"].get("k_ba_solve", 0.0)
    if ms > 0 and D <= 16:
        bcr = n6 >= 8 * 256 and bw >= 15 and os.environ.get("BH_BA_CHOL_BCR") != "0" and os.environ.get("BH_BA_CHOL") != "panel"
        if bcr:
            # block cyclic reduction in 256-blocks: the work it ISSUES (dense blocks: Cholesky + two triangular solves + three
            # products per eliminated block) next to the band's own flops, which is what `achieved` is priced on
            N = (n6 + 255) // 256
            issued, k = 0, N
            while k >= 1:
                ne = (k + 1) // 2; ns = k // 2
                issued += ne * (800 * 4) * 2048                         # on-chip Cholesky of a 256-block: 800 tile products
                issued += (2 * ns if k > 1 else 0) * 16 * 544 * 2048   # U = L^-1 M, one per (eliminated, survivor) edge of the chain
                issued += (2 * ns if k > 1 else 0) * 136 * 64 * 2048 + max(ns - 1, 0) * 256 * 64 * 2048
                if k == 1:
                    break
                k = ns
            out["solve_roofline"] = {"bound": "mfma", "kernel": "k_ba_chol_dense<true> + k_bcr_trsm + k_bcr_update + k_bcr_back",
                                     "achieved": round(flops / (ms * 1e-3) / 1e12, 5), "unit": "TFLOP/s", "peak": 78.6,
                                     "frac": round(flops / (ms * 1e-3) / 78.6e12, 6), "cus_used": 256,
                                     "mfma_flops_per_solve": int(flops), "mfma_flops_issued": int(issued),
                                     "issued_frac": round(issued / (ms * 1e-3) / 78.6e12, 5),
                                     "note": "block cyclic reduction: %d blocks of 256, %d levels; the band needs %.2f GFLOP, the nested-dissection "
                                             "order issues %.1f GFLOP on dense blocks; the chain of on-chip 256-block factorisations (one per level) "
                                             "bounds the time" % (N, int(np.ceil(np.log2(N))) + 1, flops / 1e9, issued / 1e9)}
        else:
            cus = 2 if (D == 16 and nblk >= 96 and os.environ.get("BH_BA_CHOL_TWOWAY") != "0") else 1      # two-way schedule: one CU per end
            out["solve_roofline"] = {"bound": "mfma", "kernel": "k_ba_chol_sw", "achieved": round(flops / (ms * 1e-3) / 1e12, 5), "unit": "TFLOP/s",
                                     "peak": 78.6, "frac": round(flops / (ms * 1e-3) / 78.6e12, 6), "cus_used": cus,
                                     "peak_of_cus_used": round(78.6 * cus / 256, 4), "frac_of_cus_used": round(flops / (ms * 1e-3) / (78.6e12 * cus / 256), 4),
                                     "mfma_flops_per_solve": int(flops),
                                     "note": "two dependency chains of ~%d block columns, one CU each; the dense f64 MFMA peak of 78.6 TFLOP/s is for 256 CUs" % (nblk // cus)}
    pm = _pmc("global") if dist is None else {}
    stale = "_stale" in pm
    if pm and "solve_roofline" in out and "k_bcr" in out["solve_roofline"]["kernel"]:
        # launches per solve from the profile itself: dispatches of each kernel over the dispatches of k_ba_err (one per trial)
        trials_prof = 1 if stale else max(pm.get("k_ba_err", {}).get("dispatches", 0), 1)
        names = [] if stale else [(k, pm[k]["dispatches"] / trials_prof) for k in ("k_ba_chol_dense<true>", "k_bcr_trsm", "k_bcr_update", "k_bcr_back", "k_bcr_extract") if k in pm]
        cb = _counter_block(pm, names, ms)
        if cb:
            out["solve_roofline"]["counters"] = cb
    sms = out["kernel_ms_per_iter"].get("k_ba_schur", 0.0)
    if sms > 0:
        E, P = len(pr["ecam"]), len(pr["pts"])
        alg = E * 144 + P * 96 + n6 * (bw + 1) * 8
        # the global form: one workgroup per row of S with the row camera's blocks staged in LDS (k_ba_schur_row: S and b~ in one launch);
        # BH_BA_SCHUR_ROW=0 keeps the wave-per-pair kernels of round 3
        row = os.environ.get("BH_BA_SCHUR_ROW") != "0"
        names = [("k_ba_dinv", 1), ("k_ba_schur_row", 1)] if row else [("k_ba_dinv", 1), ("k_ba_schur_S<64>", 1), ("k_ba_schur_b", 1)]
        out["schur_roofline"] = {"bound": "hbm", "kernel": " + ".join(n for n, _ in names), "algorithmic_bytes": int(alg), "ms": sms,
                                 "achieved": round(alg / (sms * 1e-3) / 1e9, 1), "unit": "GB/s", "peak": 8000, "frac": round(alg / (sms * 1e-3) / 8e12, 5)}
        cb = _counter_block(pm, names, sms) if pm and (stale or all(n in pm for n, _ in names)) else None
        if cb:
            out["schur_roofline"]["counters"] = cb
            if not cb.get("stale"):
                out["schur_roofline"]["traffic_over_algorithmic"] = round(cb["hbm_bytes"] / alg, 2)
    lms = out["kernel_ms_per_iter"].get("k_ba_linearize", 0.0)
    if lms > 0 and pm:
        cb = _counter_block(pm, [("k_ba_linearize", 1)], lms)
        if cb:
            out["linearize_counters"] = cb
    if dist is not None:                                      # the exchange per rank and Levenberg trial, by kind; the replicated part of the solve
        det = s.comm_detail()
        tr = max(out["trials"], 1)
        out["solver"] = s.solver()
        out["exchange_per_trial_per_rank"] = {
            "segment_solve": det["segment_solve"], "blocks_of_256_rows_this_rank": det["blocks_this_rank"], "blocks_per_rank": det["blocks_per_rank"],
            "reduce_scatter_S_mb": round(det["reduce_scatter_bytes"] / tr / 1e6, 3), "all_gather_mb": round(det["all_gather_bytes"] / tr / 1e6, 3),
            "allreduce_mb": round(det["allreduce_bytes"] / tr / 1e6, 3),
            "replicated_solve_ms": round(det["replicated_ms"] / max(det["solves"], 1), 4),
            "note": "bytes RECEIVED by this rank: its own band rows of S (reduce-scatter), the other segments' end blocks and step rows (all-gathers), "
                    "Hpp / bp / b~ / scalars (all-reduces); replicated_solve_ms = the chain of 2 x world end blocks, the only arithmetic every rank repeats"}
    s.close()
    return out
