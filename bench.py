#!/usr/bin/env python3
"""bench.py — RGB-D frames/s of ORB detect+describe + BF-Hamming cross-check match (BASELINE.json configs[1]).

A step = one pass of the front-end hot path over one HBM-resident batch: B synthetic 640x480 RGB-D frames ->
ORB (nfeatures=1000, the reference's other parameters, camera.py:49-56) -> match frame i against frame i-1
(direction of slam/tracking.py:56).  N>1: one process per GPU, rank g holds frames [g*B, (g+1)*B) of ONE continuous sequence
(weak scaling); frames are independent for ORB, and the pair across a shard boundary (query = first frame of rank g, train =
last frame of rank g-1) is matched by rank g after a point-to-point copy of that one descriptor row (<= 38 KB) from rank g-1:
N*B - 1 pairs in all, no data-path collective.  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` creates the N ranks itself (N child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
started before this process has touched the GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
the launcher's environment is used as is.  `--gpus` that disagrees with WORLD_SIZE is an error.  Any failure (exception, watchdog,
parity_sample_ok false) still prints the JSON line, with the reason, and exits non-zero.

The JSON line also carries
  roofline      dominant kernel: SURVEY 8(d)'s 1 592 000 algorithmic B/frame x frames per launch / its hipEvent-measured duration
                vs the 8 TB/s HBM peak (`frac`); the same kernel priced on its own compulsory bytes is `kernel_frac`
  cpu_baseline  the CPU oracle (OpenMP over frames) timed on the host cores on a bounded sample of the same frames; plus a probe
                for the real reference libraries (cv2 / g2o): timed and diffed when importable, "absent" otherwise
  inputs        the same step on a low-texture and on a high-corner-density scene (N=1)
  local_ba      LM iterations/s of the local BA on configs[2] (does not shard: replicas, one per rank, summed)
  tracking      pose-only BA and solvePnPRansac problems/s at the reference's live problem size (replicas, summed)
  bow_query     configs[3] (1 query vs 100 k entries x 1000 words): entries sharded over the ranks, per-rank top-k all-gathered
  global_ba     configs[4] (5 000 keyframes x 500 k landmarks): LM iterations/s, sharded over the ranks, RCCL all-reduce
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_FRAME = 921600 + 614400 + 32 * 1000 + 24 * 1000        # 1 592 000 B/frame at N=1000
HBM_PEAK_GBS = 8000.0                                                  # MI355X_MICROARCH.md: 8.0 TB/s spec

SCENES = {                                                             # zoom (texture px per metre), contrast about 128, noise sigma
    "default": (370.0, 1.0, 4.0),
    "low_texture": (370.0, 0.22, 1.5),
    "high_corner_density": (1500.0, 1.0, 4.0),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1000, help="frames per GPU per step (BASELINE config 2: 1000)")
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--cpu-sample", type=int, default=1000, help="frames in the cpu_baseline sample (0 = skip)")
    ap.add_argument("--pipelines", type=int, default=0, help="chunk pipelines (streams) inside the library; 0 = library default")
    ap.add_argument("--no-ba", action="store_true", help="skip local BA and the tracking stages")
    ap.add_argument("--no-inputs", action="store_true", help="skip the low-texture / high-corner-density input variants")
    ap.add_argument("--no-global-ba", action="store_true", help="skip the configs[4] global BA")
    ap.add_argument("--no-bow", action="store_true", help="skip the configs[3] BoW L1 query")
    ap.add_argument("--global-ba-timeout", type=float, default=240.0, help="watchdog for the sharded extras (BoW query, global BA), seconds")
    ap.add_argument("--timeout", type=float, default=1500.0, help="watchdog for the whole run, seconds")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ rank creation (GPU-free parent)
def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes.  This parent never imports torch or touches
    the GPU; rank 0's stdout (the JSON line) is this process's stdout; the exit code is the first non-zero rank code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                c = procs[r].poll()
                if c is not None:
                    pending.discard(r)
                    if c != 0 and rc == 0:
                        rc = c if c > 0 else 1
            if rc != 0 and pending:                      # a rank failed: give the others a moment to leave by themselves, then end them
                t_end = time.time() + 20
                while time.time() < t_end and any(procs[r].poll() is None for r in pending):
                    time.sleep(0.2)
                for r in pending:
                    if procs[r].poll() is None:
                        procs[r].kill()
                break
            time.sleep(0.1)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------ synthetic input (not timed)
def gen_frames_device(torch, B, dev, start=0, chunk=50, scene="default"):
    """Synthetic frames rendered on the GPU (data generation, not timed): same scene as boslam_amd.synth for "default"."""
    from boslam_amd import synth
    ppm, contrast, noise = SCENES[scene]
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
        sx = (2.0 * (u[None, :] - synth.CX) / synth.FX + 0.002 * k[:, None]) * ppm + 640.0      # [n][W]
        sy = (2.0 * (v[None, :] - synth.CY) / synth.FY + 0.001 * k[:, None]) * ppm + 480.0      # [n][H]
        sx = sx.remainder(2046.0); sy = sy.remainder(2046.0)                                    # long sequences wrap around the texture
        x0 = sx.floor().clamp(0, 2046).long(); fx = (sx - sx.floor()).float()
        y0 = sy.floor().clamp(0, 2046).long(); fy = (sy - sy.floor()).float()
        yy = y0[:, :, None]; xx = x0[:, None, :]
        t00 = tex[yy, xx]; t01 = tex[yy, xx + 1]; t10 = tex[yy + 1, xx]; t11 = tex[yy + 1, xx + 1]
        top = t00 * (1 - fx)[:, None, :] + t01 * fx[:, None, :]
        bot = t10 * (1 - fx)[:, None, :] + t11 * fx[:, None, :]
        img = top * (1 - fy)[:, :, None] + bot * fy[:, :, None]
        if contrast != 1.0:
            img = 128.0 + (img - 128.0) * contrast
        img = img + noise * torch.randn(img.shape, generator=g, device=dev, dtype=torch.float32)
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


# ------------------------------------------------------------------------------------------------ CPU baseline (checker side)
def cpu_baseline(bgr_host, depth_host, nfeatures, max_kp):
    """Oracle (OpenMP over frames) on the host cores, bounded sample; returns dict for the JSON line.  Also probes for the real
    reference libraries: when cv2 / g2o import, the reference's own calls are timed on a slice of the same frames and
    diffed against the oracle; otherwise the line says they are absent."""
    import oracle
    from oracle import refprobe
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
    out = {"value": n * reps / dt, "unit": "frames/s", "cores": oracle.num_threads(), "kind": "port",
           "sample": f"{reps} passes over {n} of the benchmark's own frames: ORB(nfeatures={nfeatures}) + {n - 1} consecutive-frame "
                     f"cross-check matches per pass, CPU oracle (C, -O3, OpenMP over frames), {dt:.1f} s wall"}
    try:
        out["reference_libs"] = refprobe.run(bgr_host[:min(n, 60)], depth_host[:min(n, 60)], nfeatures, kps, desc, nk)
    except Exception as e:                              # noqa: BLE001 - the probe must never cost the run its headline
        out["reference_libs"] = {"error": f"{type(e).__name__}: {e}"}
    return out, (kps, desc, nk)


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    # stdout carries exactly ONE line, the JSON result: anything native libraries print there (RCCL / gloo banners) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    from boslam_amd import dist as shard
    rank, world, _ = shard.env_rank_world()
    out = {"metric": "rgbd_frames_per_sec_orb_match_640x480", "value": None, "unit": "frames/s", "n_gpus": world}
    failed = []

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)
        os.dup2(2, 1)

    import threading
    stage = {"name": "front-end headline"}

    def bail():                                          # a stall (hung collective / kernel) must not end as a green run
        if rank == 0:
            out.setdefault("errors", []).append(f"{stage['name']}: no result within the watchdog limit")
            emit(out)
        os._exit(3)

    whole = threading.Timer(args.timeout, bail)
    whole.daemon = True
    whole.start()

    import torch
    backend = os.environ.get("BH_BENCH_BACKEND", "nccl")            # "gloo": rehearsal of N ranks on fewer GPUs (RCCL wants one GPU per rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
    local = int(os.environ.get("BH_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if backend != "nccl":
        local %= torch.cuda.device_count()
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
    f_start, f_stop, _halo = shard.frame_shard(world * B, rank, world)   # rank g: frames [gB, (g+1)B) of ONE sequence; the halo row travels in DeviceFrontEnd
    assert f_stop - f_start == B
    bgr, depth = gen_frames_device(torch, B, dev, start=f_start)
    fe = frontend.DeviceFrontEnd(ctx, B, 480, 640, params, torch_device=dev, dist=dist, backend=backend)
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(); ctx.synchronize()

    def sum_over_ranks(v):
        if dist is None:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        return float(t.item())

    def timed(front, frames_b, frames_d, steps, warm):
        for _ in range(warm):
            front.run(frames_b, frames_d)
        barrier()
        ctx.status()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            front.run(frames_b, frames_d)
        barrier()
        el = time.perf_counter() - t0
        ctx.status()
        return el

    if args.pipelines > 0:
        os.environ["BH_ORB_PIPELINES"] = str(args.pipelines)
    dt = timed(fe, bgr, depth, args.steps, args.warmup)
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
    dt = shard.reduce_max(dt, dist, dev if backend == "nccl" else "cpu")           # MAX over ranks

    n_kp = fe.n_kp.cpu().numpy(); n_match = fe.n_match.cpu().numpy()
    p_start, p_stop = shard.pair_shard(world * B, rank, world)
    pairs_local = p_stop - p_start
    pairs_total = int(sum_over_ranks(pairs_local if B > 1 or rank > 0 else 0))
    if rank == 0:
        fps = world * B * args.steps / dt
        fps_serial = B * args.steps / dt_serial
        # dominant kernel by summed hipEvent time; its own compulsory traffic per frame
        nf = args.nfeatures
        per_frame_bytes = {"k_gray_mask": 921600 + 614400 + 307200 + 38400,     # BGR + depth in, gray + mask bits out
                           "k_pyr_down": 926546 + 643332 + (926546 + 643332) // 8,   # levels 0-6 in, 1-7 out, bit-packed mask pyramid alongside
                           "k_fast": 2 * 950532 + 16 * nf,           # pyramid pixels in, blurred pyramid out, ~2 n corner records of 8 B
                           "k_select": 16 * nf + 108 * 2 * nf + 8 * nf,  # corner records, the Harris windows of ~2 n candidates (9 rows x 12 B), staging slots
                           "k_describe": 56 * nf + 2 * 950532, "k_match": 2 * 32 * nf + 12 * nf}
        dom = max(ktimes, key=lambda k: ktimes[k][0])
        dom_ms, dom_n = ktimes[dom]
        launches_per_step = dom_n / args.steps
        per_level = (params.nlevels - 1) if dom == "k_pyr_down" else 1          # k_pyr_down: one launch per level transition
        frames_per_launch = B / (launches_per_step / per_level)
        avg_s = dom_ms / dom_n / 1e3
        alg_bytes = ALG_BYTES_PER_FRAME * frames_per_launch / per_level          # 1 592 000 B per frame
        achieved = alg_bytes / avg_s / 1e9
        k_bytes = per_frame_bytes.get(dom, ALG_BYTES_PER_FRAME) * frames_per_launch / per_level
        # HBM bytes per frame from the PMC passes (tools/pmc_traffic.py): a property of the kernel as committed - a summary collected on another
        # build of the kernel source is not combined with this run's times (boslam_amd/_profiles.py): traffic stays null and the reason is given
        from boslam_amd import _profiles
        traffic, traffic_note = None, None
        tr, why = _profiles.load("profiles/r05_pmc_traffic.json", ["orb.hip", "match.hip"])
        if tr is not None and why is None:
            per_frame = tr.get({"k_gray_mask": "k_gray_bits", "k_pyr_down": "k_pyr_down_bits"}.get(dom, dom)) or tr.get(dom)
            traffic = int(per_frame * frames_per_launch / per_level) if per_frame else None
        else:
            traffic_note = why
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, **({"traffic_note": traffic_note} if traffic_note else {}),
                    "avg_launch_ms": round(dom_ms / dom_n, 4), "alg_bytes_per_launch": int(alg_bytes),
                    "byte_model": "SURVEY 8(d): 1 592 000 B/frame (BGR + depth in, descriptors + keypoints out) x frames per launch",
                    "kernel_achieved": round(k_bytes / avg_s / 1e9, 2), "kernel_frac": round(k_bytes / avg_s / 1e9 / HBM_PEAK_GBS, 5),
                    "kernel_bytes_per_launch": int(k_bytes),
                    "kernel_byte_model": "the dominant kernel's own compulsory reads + writes per frame (DESIGN.md section 4) x frames per launch",
                    "measured_in": f"second timed region of {args.steps} steps with one chunk pipeline ({round(dt_serial / args.steps * 1e3, 3)} "
                                   f"ms/step, {round(fps_serial, 1)} frames/s): with several pipelines in flight a launch shares the chip",
                    "path_achieved": round(fps / world * ALG_BYTES_PER_FRAME / 1e9, 2),
                    "path_frac": round(fps / world * ALG_BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 5),
                    "kernel_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in sorted(ktimes.items())}}
        # The front end is bound by vector-instruction ISSUE, not by HBM: price the dominant kernel on that too.
        # peak = what tools/valu_rate.hip measured on this chip for the instruction class the kernel is made of (packed 16-bit, v_perm,
        # three-operand integer ops: 4 cycles per wave64 instruction; only a handful of plain 32-bit ops issue in 2); achieved = the
        # kernel's SQ_INSTS_VALU per launch (profiles/r05_pmc_summary.json, launches of 128 frames) over the live launch time.
        try:
            vr = json.load(open(os.path.join(ROOT, "profiles", "r04_valu_rate.json")))["results"]      # (a measurement of the chip, not of a kernel)
            ps, why = _profiles.load("profiles/r05_pmc_summary.json", ["orb.hip", "match.hip"])
            if ps is None or why:
                roofline["valu"] = _profiles.stale_block("profiles/r05_pmc_summary.json", why or "absent")
                raise LookupError
            cls = ["PERM", "PKADD", "PKSUB", "PKMIN", "PKMAX", "PKMIN3", "PKMAX3", "LSHLOR", "LSHLADD", "MAD24", "ALIGNBIT", "DOT4", "DOT2", "BFE", "MINU", "MAXI"]
            peak4 = float(np.median([vr[k]["w8"]["chip_winstr_per_s"] for k in cls if k in vr]))
            peak2 = float(np.median([vr[k]["w8"]["chip_winstr_per_s"] for k in ("XOR", "SUB", "OR", "AND", "LSHR", "ASHR", "MOV", "FMA") if k in vr]))
            e = ps[{"k_gray_mask": "k_gray_bits", "k_pyr_down": "k_pyr_down_bits"}.get(dom, dom)]
            instr = e["SQ_INSTS_VALU"] * frames_per_launch / 128.0
            roofline["valu"] = {"peak_wave_instr_per_s": round(peak4, 0), "achieved": round(instr / avg_s, 0), "frac": round(instr / avg_s / peak4, 4),
                                "wave_instr_per_launch": int(instr), "valu_per_wave": e.get("derived", {}).get("valu_per_wave"),
                                "peak_two_cycle_class_wave_instr_per_s": round(peak2, 0),
                                "note": "tools/valu_rate.hip (profiles/r04_valu_rate.json): packed-16 / v_perm / min3 / dot / three-operand integer ops issue one wave64 "
                                        "instruction per 4 cycles per SIMD at any occupancy; v_add/sub/and/or/xor/lshr/mov and f32 add/mul/fma reach 2.4 - 2.8 cycles "
                                        "only in unbroken runs (alternating the two classes: 3.8)"}
        except Exception:                                           # noqa: BLE001 (profiles absent: the line goes out without the block)
            pass
        out.update({"value": round(fps, 1), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                    "config": {"workload": "configs[1]: batched ORB detect+describe + BF-Hamming cross-check match, "
                                           f"{B} frames 640x480 RGB-D per GPU per step, nfeatures={args.nfeatures}",
                               "frames_per_gpu": B, "mean_kp_per_frame": float(n_kp.mean()),
                               "mean_matches_per_pair": float(n_match[1:].mean()) if B > 1 else 0.0,
                               "pairs_matched_per_step": pairs_total,
                               "pipelines_per_gpu": args.pipelines or int(os.environ.get("BH_ORB_PIPELINES", "2")),      # the library's default (orb.hip: pipe_cfg)
                               "parallelism": f"frames sharded x{world}, one-descriptor-row halo to the next rank, no collective"},
                    "roofline": roofline})
        if args.cpu_sample > 0 and world == 1:                # cpu_baseline: rank 0 at N=1 only
            stage["name"] = "cpu_baseline"
            ns = min(args.cpu_sample, B)
            bh = bgr[:ns].cpu().numpy(); dh = depth[:ns].cpu().numpy().view(np.uint16)
            cb, (okps, odesc, onk) = cpu_baseline(bh, dh, args.nfeatures, fe.max_kp)
            out["cpu_baseline"] = cb
            # the same sample doubles as a parity spot check of the timed outputs
            ddesc = fe.desc[:ns].cpu().numpy()
            same = bool(np.array_equal(onk, n_kp[:ns])) and all(
                np.array_equal(ddesc[f, :onk[f]], odesc[f, :onk[f]]) for f in range(ns))
            out["parity_sample_ok"] = same
            if not same:
                failed.append("parity_sample_ok is false: the timed descriptors differ from the oracle's")

    if not args.no_inputs and world == 1:                     # the data-dependent paths (survivor lists, rescan) on two other scenes
        stage["name"] = "inputs"
        res = {}
        for scene in ("low_texture", "high_corner_density"):
            nb = min(B, 250)
            sb, sd = gen_frames_device(torch, nb, dev, scene=scene)
            f2 = frontend.DeviceFrontEnd(ctx, nb, 480, 640, params, torch_device=dev)
            el = timed(f2, sb, sd, 5, 1)
            res[scene] = {"frames_per_s": round(nb * 5 / el, 1), "frames": nb, "mean_kp_per_frame": float(f2.n_kp.float().mean().item()),
                          "mean_matches_per_pair": float(f2.n_match[1:].float().mean().item())}
            del f2, sb, sd
        out["inputs"] = res

    if not args.no_ba:                                        # local BA / tracking do not shard: one replica per rank, values summed
        from boslam_amd import ba_bench, track_bench
        def local_ba():
            r = ba_bench.run(ctx, dev)
            if rank == 0:                                       # W windows in one batch (bh_ba_batch_optimize): the replicas of SURVEY 8(e) inside one GPU
                r["batched"] = {"configs2_windows": ba_bench.run_batch(ctx, W=64, repeats=2),
                                "live_windows": ba_bench.run_batch(ctx, W=64, n_kf=10, n_pts=600, iters=7, repeats=2,
                                                                   label="live size (slam/covisibility_graph.py:220: 7 iterations, ~10 keyframes x 600 map points)")}
            return r
        def tracking_stage():
            r = track_bench.run(ctx)
            # ORB + back-projection + the whole _track_wrt_refkf chain (match, filter, gathers, CamOnlyBA, inlier count) for frame i against
            # frame i-1 on the benchmark's own HBM-resident frames: bh_track_refkf, no host hop
            K5 = np.array([384.239013671875, 384.239013671875, 322.432373046875, 239.6533203125, 0.06])      # config.py:36-41
            trk = frontend.DeviceTracker(ctx, B, 480, 640, params, K5, torch_device=dev)
            el = timed(trk, bgr, depth, 5, 2)
            st = trk.status[1:].cpu().numpy()
            r["frames_per_s"] = round(B * 5 / el, 1)
            r["device_resident_chain"] = {"frames_per_s": round(B * 5 / el, 1), "ms_per_step": round(el / 5 * 1e3, 3), "frames": B,
                                          "stages": "ORB detect+describe -> kp_arr + cloud_kp -> match(i, i-1) -> distance < 30 -> CamOnlyBA -> >= 15 inliers",
                                          "tracked_pairs": int((st == 0).sum()), "pairs": int(len(st)),
                                          "mean_matches": float(trk.n_match[1:].float().mean().item()), "mean_inliers": float(trk.n_inlier[1:].float().mean().item())}
            # ... and the second stage behind it: Tracker._track_local_map (slam/tracking.py:91-136, bh_track_local_map) on the pose the first stage
            # produced.  Local map of frame i = the back-projected keypoints of frames i-1 and i-2 in frame i-1's coordinates (their descriptors as the
            # medoids, normals facing the camera): ~2 000 candidates per frame, the live system's order of magnitude
            m_ = trk.max_kp
            prev2 = torch.cat([trk.cloud[:1], trk.cloud[:-1]], 0); prev2d = torch.cat([trk.desc[:1], trk.desc[:-1]], 0)
            mp_pts = torch.cat([torch.cat([trk.cloud[:1], trk.cloud[:-1]], 0), torch.cat([prev2[:1], prev2[:-1]], 0)], 1)      # row i: clouds of i-1, i-2
            mp_feat = torch.cat([torch.cat([trk.desc[:1], trk.desc[:-1]], 0), torch.cat([prev2d[:1], prev2d[:-1]], 0)], 1)
            mp_nrm = torch.zeros_like(mp_pts); mp_nrm[..., 2] = -1.0
            trk.set_local_map(mp_pts, mp_nrm, mp_feat, torch.full((B,), 2 * m_, dtype=torch.int32, device=dev))
            el2 = timed(trk, bgr, depth, 5, 2)
            lo = trk.lm["out"]; st2 = lo["status"][1:].cpu().numpy()
            r["local_map_chain"] = {"frames_per_s": round(B * 5 / el2, 1), "ms_per_step": round(el2 / 5 * 1e3, 3), "frames": B, "candidates_per_frame": 2 * m_,
                                    "stages": "the chain above -> visibility gate over the local map -> gather of the survivors' descriptors / points -> match `<= 30` -> "
                                              "CamOnlyBA -> >= 15 inliers (Tracker.update's two tracking stages, no host hop; a frame the first stage lost is skipped, "
                                              "slam/tracking.py:205-212)",
                                    "tracked_frames": int((st2 == 0).sum()), "skipped_after_stage1": int((st2 == 5).sum()), "mean_gated": float(lo["n_gate"][1:].float().mean().item()),
                                    "mean_matches": float(lo["n_match"][1:].float().mean().item()), "mean_inliers": float(lo["n_inlier"][1:].float().mean().item())}
            del trk
            if rank == 0:
                # the reference's live use: ONE frame per call from HOST memory at camera rate (camera.py:141, slam/tracking.py:54-89), results
                # back on the host - ORB, back-projection and the tracking step against the previous frame; latency, not throughput
                from boslam_amd import tracking
                himg = bgr[:2].cpu().numpy(); hdep = depth[:2].cpu().numpy()
                prev, lat = None, []
                for rep in range(24):
                    k = rep & 1
                    t0 = time.perf_counter()
                    kps, desc, nk = frontend.orb_detect_describe(ctx, himg[k][None], depth=hdep[k][None], params=params)
                    xy, cloud = frontend.kp_backproject(ctx, kps, nk, hdep[k][None], K5[0], K5[1], K5[2], K5[3])
                    o = tracking.track_refkf(ctx, desc, nk, xy, prev[0], prev[1], prev[2], K5) if prev is not None else None
                    lat.append(time.perf_counter() - t0)
                    prev = (desc, nk, cloud)
                lt = frontend.LiveTracker(ctx, 480, 640, params, K5, torch_device=dev)      # the same loop with the previous frame kept on the device
                lat2, o2 = [], None
                for rep in range(24):
                    t0 = time.perf_counter()
                    o2 = lt.step(himg[rep & 1], hdep[rep & 1])
                    lat2.append(time.perf_counter() - t0)
                lt.close()
                lt = frontend.LiveTracker(ctx, 480, 640, params, K5, torch_device=dev, use_graph=True)      # ... and each frame's launches replayed as one HIP graph
                lat3, o3 = [], None
                for rep in range(24):
                    t0 = time.perf_counter()
                    o3 = lt.step(himg[rep & 1], hdep[rep & 1])
                    lat3.append(time.perf_counter() - t0)
                lt.close()
                r["live_single_frame_graph"] = {"ms_per_frame": round(float(np.median(lat3[6:])) * 1e3, 3), "frames_per_s": round(1.0 / float(np.median(lat3[6:])), 1),
                                                "status_last": o3["status"], "n_inlier_last": o3["n_inlier"],
                                                "equals_enqueued": bool(o3["status"] == o2["status"] and o3["n_inlier"] == o2["n_inlier"] and np.array_equal(o3["R"], o2["R"])),
                                                "stages": "the loop of live_single_frame_device_state with use_graph=True: staging copies + every launch of the frame + the verdict "
                                                          "copy captured once per slot parity (torch.cuda.CUDAGraph on the stream the library has adopted) and replayed"}
                r["live_single_frame_device_state"] = {"ms_per_frame": round(float(np.median(lat2[4:])) * 1e3, 3), "frames_per_s": round(1.0 / float(np.median(lat2[4:])), 1),
                                                       "status_last": o2["status"], "n_inlier_last": o2["n_inlier"],
                                                       "stages": "frontend.LiveTracker: one pinned H2D of the frame, ORB + back-projection + bh_track_refkf against the previous "
                                                                 "frame's descriptors / cloud kept in HBM, one 112-B verdict back"}
                r["live_single_frame"] = {"ms_per_frame": round(float(np.median(lat[4:])) * 1e3, 3), "frames_per_s": round(1.0 / float(np.median(lat[4:])), 1),
                                          "status_last": int(o["status"][0]), "n_inlier_last": int(o["n_inlier"][0]),
                                          "stages": "host frame (1.5 MB over PCIe) -> ORB -> kp_arr + cloud_kp -> match vs previous frame -> CamOnlyBA -> verdict, "
                                                    "everything back on the host; three synchronous C-ABI calls per frame"}
            return r
        for name, fn in (("local_ba", local_ba), ("tracking", tracking_stage)):
            stage["name"] = name
            try:
                res = fn()
            except Exception as e:                              # noqa: BLE001
                res = {"error": f"{type(e).__name__}: {e}"}
            if "error" in res:
                failed.append(f"{name}: {res['error']}")
                if world > 1:                                   # the other ranks are on their way into a collective: leave now
                    if rank == 0:
                        out[name] = res; out["errors"] = failed; emit(out)
                    os._exit(3)
            elif name == "local_ba":
                res["replicas"] = world
                res["value_all_replicas"] = round(sum_over_ranks(res["value"]), 1)
            else:
                res["replicas"] = world
                for k in ("pose_only_ba", "pnp_ransac"):
                    res[k]["problems_per_s_all_replicas"] = round(sum_over_ranks(res[k]["problems_per_s"]), 1)
            if rank == 0:
                out[name] = res

    # Sharded extras, every N: configs[3] BoW query (entries sharded, per-rank top-k all-gathered) and configs[4] global BA.
    # They run after the headline measurement is complete and under their own watchdog.
    extras = [] if args.no_bow else ["bow_query"]
    if not args.no_global_ba:
        extras.append("global_ba")
    if extras:
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
            except Exception as e:                              # noqa: BLE001 - reported in the line and in the exit code
                res = {"error": f"{type(e).__name__}: {e}"}
                failed.append(f"{name}: {res['error']}")
                if world > 1:                                   # the other ranks may be inside a collective: do not join them again
                    dog.cancel()
                    if rank == 0:
                        out[name] = res; out["errors"] = failed; emit(out)
                    os._exit(3)
            if rank == 0:
                out[name] = res
        dog.cancel()
    whole.cancel()
    if rank == 0:
        if failed:
            out["errors"] = failed
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return 1 if failed else 0


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}; launch one rank per GPU or drop the launcher\n")
        sys.exit(2)
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
