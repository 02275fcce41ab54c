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
