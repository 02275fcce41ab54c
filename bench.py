#!/usr/bin/env python3
"""bench.py — RGB-D frames/s of ORB detect+describe + BF-Hamming cross-check match (BASELINE.json configs[1]).

A step = one pass of the front-end hot path over one HBM-resident batch: B synthetic 640x480 RGB-D frames ->
ORB (nfeatures=1000, the reference's other parameters, camera.py:49-56) -> match frame i against frame i-1
(direction of slam/tracking.py:56).  N>1: one process per GPU, each with its own B frames (frames are independent:
weak scaling, no data-path collective); rank 0 prints ONE JSON line.

The JSON line also carries
  roofline      the dominant kernel's algorithmic bytes / its hipEvent-measured duration vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (OpenMP over frames) timed on the host cores on a bounded sample of the same frames
  local_ba      secondary metric of BASELINE.json: LM iterations/s of the local BA on config 3 (N=1)
  tracking      pose-only BA and solvePnPRansac problems/s at the reference's live problem size (N=1)
  bow_query     configs[3] (1 query vs 100 k entries x 1000 words): queries/s, entries sharded over the N ranks, per-rank top-k
                all-gathered and merged (every N)
  global_ba     configs[4] (5 000 keyframes x 500 k landmarks): LM iterations/s, landmark-sharded over the N ranks with
                RCCL all-reduce of the reduced camera system (strong scaling; every N)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_FRAME = 921600 + 614400 + 32 * 1000 + 24 * 1000        # 1 592 000 B/frame at N=1000
HBM_PEAK_GBS = 8000.0                                                  # MI355X_MICROARCH.md: 8.0 TB/s spec


def gen_frames_device(torch, B, dev, start=0, chunk=50):
    """Synthetic frames rendered on the GPU (data generation, not timed): same scene as boslam_amd.synth."""
    from boslam_amd import synth
    tex = torch.from_numpy(synth._texture()).to(dev)
    bgr = torch.empty((B, synth.HEIGHT, synth.WIDTH, 3), dtype=torch.uint8, device=dev)
    depth = torch.empty((B, synth.HEIGHT, synth.WIDTH), dtype=torch.int16, device=dev)
    u = torch.arange(synth.WIDTH, dtype=torch.float64, device=dev)
    v = torch.arange(synth.HEIGHT, dtype=torch.float64, device=dev)
    gains = torch.tensor([1.0, 0.9, 0.8], dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(2025 + start)
    rng = np.random.default_rng(2026 + start)
    for c0 in range(0, B, chunk):
        n = min(chunk, B - c0)
        k = torch.arange(start + c0, start + c0 + n, dtype=torch.float64, device=dev)
        sx = (2.0 * (u[None, :] - synth.CX) / synth.FX + 0.002 * k[:, None]) * 370.0 + 640.0      # [n][W]
        sy = (2.0 * (v[None, :] - synth.CY) / synth.FY + 0.001 * k[:, None]) * 370.0 + 480.0      # [n][H]
        x0 = sx.floor().clamp(0, 2046).long(); fx = (sx - sx.floor()).float()
        y0 = sy.floor().clamp(0, 2046).long(); fy = (sy - sy.floor()).float()
        yy = y0[:, :, None]; xx = x0[:, None, :]
        t00 = tex[yy, xx]; t01 = tex[yy, xx + 1]; t10 = tex[yy + 1, xx]; t11 = tex[yy + 1, xx + 1]
        top = t00 * (1 - fx)[:, None, :] + t01 * fx[:, None, :]
        bot = t10 * (1 - fx)[:, None, :] + t11 * fx[:, None, :]
        img = top * (1 - fy)[:, :, None] + bot * fy[:, :, None]
        img = img + 4.0 * torch.randn(img.shape, generator=g, device=dev, dtype=torch.float32)
        bgr[c0:c0 + n] = (img[..., None] * gains).round().clamp(0, 255).to(torch.uint8)
        d = torch.full((n, synth.HEIGHT, synth.WIDTH), 2000, dtype=torch.int16, device=dev)
        X = torch.arange(synth.WIDTH, device=dev)[None, None, :]; Y = torch.arange(synth.HEIGHT, device=dev)[None, :, None]
        for _ in range(12):
            w = torch.from_numpy(rng.integers(20, 60, size=n)).to(dev)[:, None, None]
            h = torch.from_numpy(rng.integers(15, 45, size=n)).to(dev)[:, None, None]
            xh = torch.from_numpy(rng.integers(0, synth.WIDTH - 60, size=n)).to(dev)[:, None, None]
            yh = torch.from_numpy(rng.integers(0, synth.HEIGHT - 45, size=n)).to(dev)[:, None, None]
            d[(X >= xh) & (X < xh + w) & (Y >= yh) & (Y < yh + h)] = 0
        depth[c0:c0 + n] = d
    return bgr, depth


def cpu_baseline(bgr_host, depth_host, nfeatures, max_kp):
    """Oracle (OpenMP over frames) on the host cores, bounded sample; returns dict for the JSON line."""
    import oracle
    P = oracle.orb_params(nfeatures=nfeatures)
    cores = min(len(os.sched_getaffinity(0)), 16)      # the box's CPU share for one GPU is 16 cores
    oracle.set_num_threads(cores)
    n = len(bgr_host)
    reps = 4                                            # ~10 s of CPU work on 16 cores
    t0 = time.perf_counter()
    for _ in range(reps):
        kps, desc, nk, rc = oracle.orb_batch(bgr_host, depth_host, P, max_kp)
        oracle.match_consecutive(desc, nk, max_kp)
    dt = time.perf_counter() - t0
    return {"value": n * reps / dt, "unit": "frames/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"{reps} passes over {n} of the benchmark's own frames: ORB(nfeatures={nfeatures}) + {n - 1} consecutive-frame "
                      f"cross-check matches per pass, CPU oracle (C, -O3, OpenMP over frames), {dt:.1f} s wall"}, (kps, desc, nk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1000, help="frames per GPU per step (BASELINE config 2: 1000)")
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--cpu-sample", type=int, default=1000, help="frames in the cpu_baseline sample (0 = skip)")
    ap.add_argument("--pipelines", type=int, default=0, help="chunk pipelines (streams) inside the library; 0 = library default (4)")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-global-ba", action="store_true", help="skip the configs[4] global BA (landmark-sharded over the ranks)")
    ap.add_argument("--global-ba-timeout", type=float, default=240.0, help="watchdog for the sharded extras (BoW query, global BA), seconds")
    ap.add_argument("--no-bow", action="store_true", help="skip the configs[3] BoW L1 query (entries sharded over the ranks)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: anything native libraries print there (RCCL / gloo banners) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("BH_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # BH_BENCH_DEVICE: rehearsals of N ranks on one GPU
    backend = os.environ.get("BH_BENCH_BACKEND", "nccl")                               # "gloo" for such rehearsals (RCCL refuses two ranks per GPU)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend=backend)

    from boslam_amd import Context, frontend
    ctx = Context(local)
    params = frontend.make_params(nfeatures=args.nfeatures)
    B = args.frames
    bgr, depth = gen_frames_device(torch, B, dev, start=rank * B)
    fe = frontend.DeviceFrontEnd(ctx, B, 480, 640, params, torch_device=dev)
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(); ctx.synchronize()

    if args.pipelines > 0:
        os.environ["BH_ORB_PIPELINES"] = str(args.pipelines)
    for _ in range(args.warmup):
        fe.run(bgr, depth)
    barrier()
    ctx.status()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fe.run(bgr, depth)
    barrier()
    dt = time.perf_counter() - t0
    ctx.status()
    # Roofline pass: the library overlaps several chunk pipelines on separate streams, which makes single-launch durations
    # non-additive (a launch shares the chip with other pipelines' kernels).  Per-kernel durations are therefore taken in a
    # second timed region of the same K steps with ONE pipeline, hipEvents around every launch on the launch stream.
    os.environ["BH_ORB_PIPELINES"] = "1"
    fe.run(bgr, depth)
    barrier()
    ctx.timing(True)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        fe.run(bgr, depth)
    barrier()
    dt_serial = time.perf_counter() - t1
    ktimes = ctx.timing_read()
    ctx.timing(False)
    if args.pipelines > 0:
        os.environ["BH_ORB_PIPELINES"] = str(args.pipelines)
    else:
        os.environ.pop("BH_ORB_PIPELINES", None)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        n_kp = fe.n_kp.cpu().numpy(); n_match = fe.n_match.cpu().numpy()
        fps = world * B * args.steps / dt
        fps_serial = B * args.steps / dt_serial
        # dominant kernel by summed hipEvent time; its own compulsory traffic per frame
        per_frame_bytes = {"k_gray_mask": 921600 + 614400 + 2 * 307200, "k_pyr_down": 2 * (926546 + 643332),
                           "k_fast": 3 * 950532, "k_blur": 2 * 950532, "k_select": 950532 + 56 * args.nfeatures,
                           "k_describe": 56 * args.nfeatures + 2 * 950532, "k_match": 2 * 32 * args.nfeatures + 12 * args.nfeatures}
        dom = max(ktimes, key=lambda k: ktimes[k][0])
        dom_ms, dom_n = ktimes[dom]
        launches_per_step = dom_n / args.steps
        frames_per_launch = B / launches_per_step if dom != "k_pyr_down" else B / (launches_per_step / (params.nlevels - 1))
        if dom == "k_pyr_down":
            alg_bytes = per_frame_bytes[dom] * frames_per_launch / (params.nlevels - 1)
        else:
            alg_bytes = per_frame_bytes[dom] * frames_per_launch
        avg_s = dom_ms / dom_n / 1e3
        achieved = alg_bytes / avg_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                per_frame = json.load(open(pmc)).get(dom)          # HBM bytes per frame from the PMC passes (tools/pmc_traffic.py)
                traffic = int(per_frame * frames_per_launch / ((params.nlevels - 1) if dom == "k_pyr_down" else 1)) if per_frame else None
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "avg_launch_ms": round(dom_ms / dom_n, 4), "alg_bytes_per_launch": int(alg_bytes),
                    "measured_in": f"second timed region of {args.steps} steps with one chunk pipeline ({round(dt_serial / args.steps * 1e3, 3)} "
                                   f"ms/step, {round(fps_serial, 1)} frames/s): with several pipelines in flight a launch shares the chip",
                    "note": "every front-end kernel except k_gray_mask is VALU-issue- or latency-bound rather than HBM-bound (DESIGN.md section 4, "
                            "profiles/r01_final_pmc_summary.json); the HBM fraction is what the measurement contract asks for",
                    "path_achieved": round(fps / world * ALG_BYTES_PER_FRAME / 1e9, 2),
                    "path_frac": round(fps / world * ALG_BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 5),
                    "kernel_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in sorted(ktimes.items())}}
        out = {"metric": "rgbd_frames_per_sec_orb_match_640x480", "value": round(fps, 1), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
               "data": "synthetic",
               "config": {"workload": "configs[1]: batched ORB detect+describe + BF-Hamming cross-check match, "
                                      f"{B} frames 640x480 RGB-D per GPU per step, nfeatures={args.nfeatures}",
                          "frames_per_gpu": B, "mean_kp_per_frame": float(n_kp.mean()),
                          "mean_matches_per_pair": float(n_match[1:].mean()) if B > 1 else 0.0,
                          "pipelines_per_gpu": args.pipelines or 4,
                          "parallelism": f"frames sharded x{world}, no collective"},
               "roofline": roofline}
        if args.cpu_sample > 0 and world == 1:                # cpu_baseline: rank 0 at N=1 only
            ns = min(args.cpu_sample, B)
            bh = bgr[:ns].cpu().numpy(); dh = depth[:ns].cpu().numpy().view(np.uint16)
            cb, (okps, odesc, onk) = cpu_baseline(bh, dh, args.nfeatures, fe.max_kp)
            out["cpu_baseline"] = cb
            # the same sample doubles as a parity spot check of the timed outputs
            ddesc = fe.desc[:ns].cpu().numpy()
            same = bool(np.array_equal(onk, n_kp[:ns])) and all(
                np.array_equal(ddesc[f, :onk[f]], odesc[f, :onk[f]]) for f in range(ns))
            out["parity_sample_ok"] = same
        if not args.no_ba and world == 1:                     # local BA does not shard (replicas only): N=1 figure
            from boslam_amd import ba_bench
            out["local_ba"] = ba_bench.run(ctx, dev)
        if not args.no_ba and world == 1:                     # tracking stages (pose-only BA, PnP-RANSAC): N=1 figures
            from boslam_amd import track_bench
            out["tracking"] = track_bench.run(ctx)

    # Sharded extras, every N: configs[3] BoW query (entries sharded, per-rank top-k all-gathered) and configs[4] global BA (landmarks
    # sharded, RCCL all-reduce hook).  They run after the headline measurement is complete and under a watchdog: should a multi-GPU
    # collective path stall or a rank fail, rank 0 still prints the JSON line (with the reason) and every rank leaves, instead of
    # losing the run to a time-out.
    extras = [] if args.no_bow else ["bow_query"]
    if not args.no_global_ba:
        extras.append("global_ba")
    if extras:
        import threading
        stage = {"name": extras[0]}

        def bail():
            if rank == 0:
                out[stage["name"]] = {"error": f"no result within {args.global_ba_timeout} s (watchdog)"}
                emit(out)
            os._exit(0)

        dog = threading.Timer(args.global_ba_timeout, bail)
        dog.daemon = True
        dog.start()
        for name in extras:
            stage["name"] = name
            try:
                if name == "bow_query":
                    from boslam_amd import bow_bench
                    res = bow_bench.run_sharded(ctx, dist=dist, device=dev)
                else:
                    from boslam_amd import ba_bench
                    res = ba_bench.run_global(ctx, dist=dist, device=dev)
            except Exception as e:                              # noqa: BLE001 - reported in the line, the headline stands
                res = {"error": f"{type(e).__name__}: {e}"}
                if world > 1:                                   # the other ranks may be inside a collective: do not join them again
                    dog.cancel()
                    if rank == 0:
                        out[name] = res
                        emit(out)
                    os._exit(0)
            if rank == 0:
                out[name] = res
        dog.cancel()
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
