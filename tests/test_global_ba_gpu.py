"""Landmark-sharded global BA: the all-reduce hook path against the single-GPU solver."""
import os, socket, subprocess, sys
import numpy as np
import pytest

from boslam_amd import ba, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _single(ctx, pr, iters):
    s = ba.LocalBA(ctx, pr["K"])
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    done = s.optimize(iters)
    out = (s.poses(), s.points(), s.log()[:done])
    s.close()
    return out


@pytest.fixture(scope="module")
def ctx():
    from boslam_amd import Context
    c = Context(0)
    yield c
    c.close()


def test_hook_world1_rccl_is_bit_identical(ctx):
    """world_size 1 over the RCCL backend: exercises the in-place device all-reduce path; with one rank every sum has the
    single-GPU order, so poses, landmarks and the LM log must be bitwise equal."""
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    pr = synth.ba_problem_sequence(n_kf=150, n_pts=4000, window=6, seed=3)
    (R0, t0), p0, lg0 = _single(ctx, pr, 5)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        s = ba.GlobalBA(ctx, pr["K"], dist)
        s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
        done = s.optimize(5)
        R, t = s.poses(); pts = s.points_all(); lg = s.log()[:done]
        ms, nbytes = s.comm()
        s.close()
    finally:
        dist.destroy_process_group()
    assert np.array_equal(lg, lg0) and np.array_equal(R, R0) and np.array_equal(t, t0) and np.array_equal(pts, p0)
    assert nbytes > 0


def test_two_ranks_match_single_gpu(ctx, tmp_path):
    """Two ranks (gloo, both on this box's one GPU; the hook stages through host memory): same LM decisions, poses and
    landmarks within 1e-9 of the single-GPU run, and bitwise equal across the two ranks."""
    n_kf, n_pts, window, iters = 150, 4000, 6, 5
    pr = synth.ba_problem_sequence(n_kf=n_kf, n_pts=n_pts, window=window, seed=3)
    (R0, t0), p0, lg0 = _single(ctx, pr, iters)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_global_ba_worker.py"), str(tmp_path), str(n_kf), str(n_pts),
                               str(window), str(iters)], env=dict(env, RANK=str(r))) for r in range(2)]
    try:
        rcs = [p.wait(timeout=300) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0, 0]
    a = np.load(tmp_path / "rank0.npz"); b = np.load(tmp_path / "rank1.npz")
    for k in ("R", "t", "pts", "log"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["log"][:, 2:], lg0[:, 2:]) and np.allclose(a["log"][:, 0], lg0[:, 0], rtol=1e-9)
    assert np.abs(a["R"] - R0).max() < 1e-9 and np.abs(a["t"] - t0).max() < 1e-9 and np.abs(a["pts"] - p0).max() < 1e-9
    assert float(a["comm_bytes"]) > 0


def _run_ranks(tmp_path, world, n_kf, n_pts, window, iters, extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_global_ba_worker.py"), str(tmp_path), str(n_kf), str(n_pts), str(window), str(iters)],
                              env=dict(env, RANK=str(r))) for r in range(world)]
    try:
        rcs = [p.wait(timeout=600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0] * world
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("world,window", [(2, 6), (3, 20)])
def test_segment_wise_solve_matches_single_gpu(ctx, tmp_path, world, window):
    """The sharded solve of long banded systems: S reduce-scattered by keyframe segment, every rank eliminates the
    interior of its segment, the segments' end blocks are all-gathered and only that chain of 2 x world blocks is solved on every
    rank.  700 keyframes (n = 4 188: 17 blocks of 256 -> 9 + 8 on two ranks, 6 + 6 + 5 on three; window 20 is the widest band the
    block solver takes).  gloo ranks sharing this box's GPU: same Levenberg decisions as the single-GPU run, poses and landmarks within
    1e-9, bitwise equal across ranks; bytes received per rank and trial stay near S / world; the old scheme (BH_BA_DSOLVE=0) agrees too."""
    n_kf, n_pts, iters = 700, 20000, 4
    pr = synth.ba_problem_sequence(n_kf=n_kf, n_pts=n_pts, window=window, seed=3)
    (R0, t0), p0, lg0 = _single(ctx, pr, iters)
    outs = _run_ranks(tmp_path, world, n_kf, n_pts, window, iters)
    a = outs[0]
    for b in outs[1:]:
        for k in ("R", "t", "pts", "log"):
            assert np.array_equal(a[k], b[k]), k
    assert str(a["solver"]) == "bcr-segments" and bool(a["det_segment_solve"])
    assert sorted(int(o["det_blocks_this_rank"]) for o in outs) == sorted([9, 8] if world == 2 else [6, 6, 5])
    assert np.array_equal(a["log"][:, 2:], lg0[:, 2:]) and np.allclose(a["log"][:, 0], lg0[:, 0], rtol=1e-9)
    assert np.abs(a["R"] - R0).max() < 1e-9 and np.abs(a["t"] - t0).max() < 1e-9 and np.abs(a["pts"] - p0).max() < 1e-9
    trials = int(a["log"][:, 2].sum())
    n6 = 6 * (n_kf - 2); bw = min(n6 - 1, 6 * 2 * window + 5)
    s_bytes = 8.0 * n6 * (bw + 1)
    per_trial = (float(a["det_reduce_scatter_bytes"]) + float(a["det_all_gather_bytes"])) / trials
    assert float(a["det_reduce_scatter_bytes"]) / trials <= 1.35 * s_bytes / world            # its own rows of S (+ chunk padding)
    assert per_trial <= 1.35 * s_bytes / world + (world - 1) * (4 * 65536 + 512) * 8 + 8.0 * (n6 + 256 * world * 2)
    assert int(a["det_solves"]) == trials and float(a["det_replicated_ms"]) > 0
    # the library's own accounting equals the plan arithmetic tests/test_host_cpu.py uses for configs[4] (the bytes are deterministic)
    from boslam_amd import ba as ba_mod
    model = ba_mod.exchange_bytes_per_trial(n6, bw, world)
    assert float(a["det_reduce_scatter_bytes"]) == model["reduce_scatter"] * trials and float(a["det_all_gather_bytes"]) == model["all_gather"] * trials
    if world == 2:                                            # the round-2 scheme stays selectable and agrees
        old = _run_ranks(tmp_path, world, n_kf, n_pts, window, iters, {"BH_BA_DSOLVE": "0"})
        assert str(old[0]["solver"]) == "bcr" and not bool(old[0]["det_segment_solve"])
        assert np.array_equal(old[0]["log"][:, 2:], lg0[:, 2:]) and np.abs(old[0]["t"] - t0).max() < 1e-9
        old_model = ba_mod.exchange_bytes_per_trial(n6, bw, world, segments=False)
        assert float(old[0]["det_allreduce_bytes"]) >= old_model["total"] * trials                   # S + b~ per trial (+ Hpp, bp, scalars per iteration)
        # (bytes: at this size - S is 2.6 MB, an end-block share 2.1 MB - the old scheme moves less; at configs[4], S = 59 MB on 8 ranks, it
        #  is 59 MB against 7.4 + 14.7 + 0.2 MB per rank and trial: DESIGN.md section 7)


def _with_env(name, value, fn):
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return fn()
    finally:
        if old is None: del os.environ[name]
        else: os.environ[name] = old


@pytest.mark.parametrize("duplicates", [False, True])
def test_row_form_of_the_schur_complement_equals_the_wave_per_pair_form(ctx, duplicates):
    """The global-BA form of S and b~ (k_ba_schur_row: camera s1's blocks staged once per row in LDS) against the wave-per-pair kernels
    (BH_BA_SCHUR_ROW=0, read at set_problem): same LM decisions, costs and damping to 1e-9, poses and landmarks to 1e-9.  The problem is large enough
    for the one-wave-per-pair switch (>= 4096 camera pairs), and one keyframe sees 1 300 extra landmarks: more than the 1 024 blocks a
    row keeps in LDS, so the tail of that row takes the global-memory path.  duplicates: 200 landmarks are observed TWICE by one of their keyframes
    (the diagonal block then has cross terms between the two observations, which the row kernel may not fold into its staging pass)."""
    pr = synth.ba_problem_sequence(n_kf=700, n_pts=20000, window=6, seed=5)
    rng = np.random.default_rng(1)
    hub = 350
    extra = 1300
    P0 = len(pr["pts"])
    local = np.stack([rng.uniform(-1, 1, extra), rng.uniform(-0.7, 0.7, extra), rng.uniform(2.5, 5.0, extra)], axis=1)
    Xc = pr["tgt"][hub] + local @ pr["Rgt"][hub].T
    ecam, ept, meas = [pr["ecam"]], [pr["ept"]], [pr["meas"]]
    fx, fy, cx, cy = pr["K"][:4]
    for c in (hub - 1, hub, hub + 1):
        pc = (Xc - pr["tgt"][c]) @ pr["Rgt"][c]                                   # R'(X - t)
        uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], axis=1)
        ecam.append(np.full(extra, c, np.int32)); ept.append(np.arange(P0, P0 + extra, dtype=np.int32)); meas.append(uv + rng.normal(0, 1.0, uv.shape))
    pr = dict(pr, pts=np.concatenate([pr["pts"], Xc + rng.normal(0, 0.02, Xc.shape)]), ecam=np.concatenate(ecam), ept=np.concatenate(ept), meas=np.concatenate(meas))
    assert np.bincount(pr["ecam"])[hub] > 1024
    if duplicates:
        pick = rng.choice(len(pr["ecam"]), 200, replace=False)
        pr = dict(pr, ecam=np.concatenate([pr["ecam"], pr["ecam"][pick]]), ept=np.concatenate([pr["ept"], pr["ept"][pick]]),
                  meas=np.concatenate([pr["meas"], pr["meas"][pick] + rng.normal(0, 0.5, (200, 2))]))
    (Rr, tr), pr_pts, lgr = _with_env("BH_BA_SCHUR_ROW", "1", lambda: _single(ctx, pr, 4))
    (Rw, tw), pw_pts, lgw = _with_env("BH_BA_SCHUR_ROW", "0", lambda: _single(ctx, pr, 4))
    assert len(lgr) == len(lgw) == 4
    assert np.array_equal(lgr[:, 2:], lgw[:, 2:]) and np.allclose(lgr[:, :2], lgw[:, :2], rtol=1e-9, atol=0)
    assert lgr[-1, 0] < 0.8 * lgr[0, 0]                                           # and it optimises
    assert np.abs(Rr - Rw).max() < 1e-9 and np.abs(tr - tw).max() < 1e-9 and np.abs(pr_pts - pw_pts).max() < 1e-9
