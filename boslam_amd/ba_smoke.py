"""Tiny BA pass for __graft_entry__.smoke(): returns the GPU results; the caller (smoke) checks them with the oracle."""
from . import ba, synth


def run(ctx):
    p = synth.pose_problem(n=80, seed=2)
    R, t, chi2, inl, st = ba.pose_only_ba_batch(ctx, p["pix"][None], p["pts"][None], [len(p["pix"])], p["K"])
    pr = synth.ba_problem(n_kf=5, n_pts=100, seed=4)
    s = ba.LocalBA(ctx, pr["K"])
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    s.optimize(5)
    _, tt = s.poses()
    pts = s.points()
    s.close()
    return dict(pose_problem=p, pose=(R[0], t[0], inl[0], st[0]), ba_problem=pr, ba=(tt, pts))
