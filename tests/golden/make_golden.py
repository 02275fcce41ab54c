"""Regenerates tests/golden/*.npz from the CPU oracle on seeded synthetic inputs.

The reference has no golden vectors for this path and cannot be imported here (cv2 / g2o absent — an ordinary
ModuleNotFoundError), so these fixtures pin the ORACLE's behaviour (regression) and give the -m gpu
tests a second, committed target.  Run:  python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle                      # noqa: E402
from boslam_amd import synth       # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    for f in range(2):
        bgr, depth = synth.render_frame(f)
        kps, desc, rc = oracle.orb_detect_describe(bgr, (depth > 0).astype(np.uint8) * 255, oracle.orb_params(nfeatures=600))
        assert rc == 0
        out[f"desc{f}"] = desc
        for fld in ("x", "y", "angle", "response", "size", "octave"):
            out[f"kp{f}_{fld}"] = kps[fld]
    qi, ti, d = oracle.match_crosscheck(out["desc1"], out["desc0"])
    out["match_1_vs_0"] = np.stack([qi, ti, d], 1)
    np.savez_compressed(os.path.join(HERE, "frontend_golden.npz"), **out)

    p = synth.pose_problem(n=120, seed=11)
    o = oracle.pose_only_ba(p["pix"], p["pts"], p["K"])
    pr = synth.ba_problem(n_kf=6, n_pts=150, seed=1)
    b = oracle.ba_optimize(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"], pr["K"], 7, np.sqrt(7.815))
    np.savez_compressed(os.path.join(HERE, "ba_golden.npz"), pose_R=o["R"], pose_t=o["t"], pose_inlier=o["inlier"],
                        pose_chi2=o["chi2"], ba_R=b["R"], ba_t=b["t"], ba_pts=b["pts"], ba_chi2=b["chi2"], ba_log=b["log"])
    np.savez_compressed(os.path.join(HERE, "config1_golden.npz"), **config1_oracle(100))
    print("golden vectors written to", HERE)


def config1_oracle(nfr):
    """BASELINE configs[0] on the CPU oracle: the replay of boslam_amd.two_way_tracking with oracle back ends."""
    from boslam_amd import two_way_tracking
    bgr, depth = synth.render_batch(nfr, traj="config1")
    P = oracle.orb_params(nfeatures=600)
    kps, desc, nk, rc = oracle.orb_batch(bgr, depth, P, 768)
    assert np.all(np.asarray(rc) == 0)
    xy = np.zeros((nfr, 768, 2), np.int64); cloud = np.zeros((nfr, 768, 3))
    for f in range(nfr):
        xy[f, :nk[f]] = oracle.kp_to_int(kps[f, :nk[f]])
        cloud[f, :nk[f]] = oracle.backproject(depth[f], xy[f, :nk[f]], synth.FX, synth.FY, synth.CX, synth.CY)
    K4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY])

    def match_fn(p, f):
        return oracle.match_crosscheck(desc[p, :nk[p]], desc[f, :nk[f]])

    def pnp2_fn(a, b, b1, a1):
        return oracle.solve_pnp_ransac(a, b, K4), oracle.solve_pnp_ransac(b1, a1, K4)

    log = two_way_tracking.replay(desc, nk, xy, cloud, match_fn, pnp2_fn, oracle.rodrigues)
    log["n_kp"] = nk
    return log


if __name__ == "__main__":
    main()
