"""BASELINE configs[0]: the 100-frame two-way PnP tracking sequence (counterpart of experiments/pnp_two_way_tracking.py).

The script feeds every frame's fused (R, t) back into the next frame's PnP input (camera.py:32-43) and RANSAC's inlier test is a
hard threshold, so anything short of bit-identical poses would eventually flip an inlier and let the two trajectories drift
apart.  The HIP solvePnPRansac therefore reproduces the oracle bit for bit (point-order sums, detmath.h), and both checks are
exact: (1) teacher-forced, every one of the 198 problems of the oracle's replay (pinned by tests/golden/config1_golden.npz in
tests/test_oracle_cpu.py) returns the oracle's inliers, rvec and tvec; (2) the free-running GPU replay equals the golden log
for all 100 frames."""
import os
import numpy as np
import pytest

import oracle
from boslam_amd import frontend, pnp, synth, two_way_tracking, cv2_shim

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1_golden.npz")
K4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY])


@pytest.fixture(scope="module")
def seq(ctx):
    g = np.load(GOLD)
    nfr = len(g["state"])
    bgr, depth = synth.render_batch(nfr, traj="config1")
    params = frontend.make_params(nfeatures=600)
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params)
    xy, cloud = frontend.kp_backproject(ctx, kps, n, depth, *K4)
    return dict(g=g, bgr=bgr, depth=depth, desc=desc, n=n, xy=xy, cloud=cloud)


def test_teacher_forced_every_frame_matches_the_oracle(ctx, seq):
    g, desc, n, xy, cloud = seq["g"], seq["desc"], seq["n"], seq["xy"], seq["cloud"]
    assert np.array_equal(n, g["n_kp"])
    stats = dict(frames=0, problems=0)

    def match_fn(p, f):
        qi, ti, d, nm = frontend.match_pairs(ctx, desc[p:p + 1], n[p:p + 1], desc[f:f + 1], n[f:f + 1])
        oq, ot, od = oracle.match_crosscheck(desc[p, :n[p]], desc[f, :n[f]])
        assert np.array_equal(qi[0, :nm[0]], oq) and np.array_equal(ti[0, :nm[0]], ot) and np.array_equal(d[0, :nm[0]], od)
        return oq, ot, od

    def pnp2_fn(a, b, b1, a1):
        S = len(a)
        obj = np.stack([a, b1]); img = np.stack([b, a1])
        rv, tv, inl, ni, ok = pnp.solve_pnp_ransac_batch(ctx, obj, img, np.array([S, S], np.int32), K4)
        out = (oracle.solve_pnp_ransac(a, b, K4), oracle.solve_pnp_ransac(b1, a1, K4))
        for k in range(2):
            o_ok, o_r, o_t, o_in = out[k]
            assert bool(ok[k]) == o_ok
            assert np.array_equal(inl[k][:ni[k]], o_in)
            assert np.array_equal(rv[k], o_r) and np.array_equal(tv[k], o_t), (rv[k] - o_r, tv[k] - o_t)     # bit for bit
            assert np.array_equal(cv2_shim.Rodrigues(rv[k])[0], oracle.rodrigues(rv[k]))
            stats["problems"] += 1
        stats["frames"] += 1
        return out                                                               # continue on the oracle's state

    log = two_way_tracking.replay(desc, n, xy, cloud, match_fn, pnp2_fn, oracle.rodrigues)
    for k in ("state", "num_matches", "num_inliers", "num_union"):
        assert np.array_equal(log[k], g[k]), k
    assert np.abs(log["R"] - g["R"]).max() < 1e-12 and np.abs(log["t"] - g["t"]).max() < 1e-12
    assert stats["frames"] == 99 and stats["problems"] == 198, stats


def test_free_running_gpu_replay_equals_the_golden_log(ctx, seq):
    g = seq["g"]
    log = two_way_tracking.run(ctx, seq["bgr"], seq["depth"], K4)
    assert (log["state"][1:] == 1).all()
    for k in ("state", "num_matches", "num_inliers", "num_union"):
        assert np.array_equal(log[k], g[k]), k
    for k in ("R", "t", "point_diff", "norm_tvec", "cam_R", "cam_t_norm"):
        assert np.array_equal(log[k], g[k]), (k, np.abs(log[k] - g[k]).max())                                  # all 100 frames, bit for bit
