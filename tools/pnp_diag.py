"""GPU diagnostic for the a9 parity gap: (1) device f64 primitives vs host libm, bit for bit; (2) teacher-forced config-1 replay,
every PnP problem whose inlier set or pose differs from the oracle is saved (gpurun_out/pnp_fail.npz) and traced on both sides;
the first differing trace field is printed."""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from boslam_amd import Context, _lib, frontend, pnp, synth, two_way_tracking

FIELDS = [("it", 0, 1), ("idx", 1, 6), ("cws", 8, 20), ("alphas", 20, 40), ("MtM", 40, 184), ("dm", 184, 196), ("ut", 196, 340),
          ("L", 340, 400), ("rho", 400, 406)]
for N in range(3):
    b = 406 + 21 * N
    FIELDS += [(f"betas0_N{N+1}", b, b + 4), (f"betas_N{N+1}", b + 4, b + 8), (f"R_N{N+1}", b + 8, b + 17), (f"t_N{N+1}", b + 17, b + 20),
               (f"err_N{N+1}", b + 20, b + 21)]
FIELDS += [("R", 469, 478), ("t", 478, 481), ("good", 481, 482)]
K4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY])


def gpu_trace(ctx, obj, img):
    n = len(obj)
    obj = np.ascontiguousarray(obj[None]); img = np.ascontiguousarray(img[None])
    rv = np.zeros((1, 3)); tv = np.zeros((1, 3)); inl = np.zeros((1, n), np.int32); ni = np.zeros(1, np.int32); ok = np.zeros(1, np.int32)
    tr = np.zeros((1, 32, 512))
    nn = np.array([n], np.int32)
    ctx.check(_lib.lib().bh_debug_pnp_trace(ctx.h, _lib.ptr(obj), _lib.ptr(img), _lib.ptr(nn), 1, n, _lib.ptr(K4), 100, 8.0, 0.99,
                                            _lib.ptr(rv), _lib.ptr(tv), _lib.ptr(inl), _lib.ptr(ni), _lib.ptr(ok), _lib.ptr(tr)))
    return tr[0], rv[0], tv[0], inl[0, :ni[0]]


def oracle_trace(obj, img):
    tr = np.zeros((32, 512))
    oracle.lib().orc_pnp_set_trace(tr.ctypes.data_as(C.c_void_p), 32)
    out = oracle.solve_pnp_ransac(obj, img, K4)
    oracle.lib().orc_pnp_set_trace(None, 0)
    return tr, out


def compare(tg, to, tag):
    for h in range(32):
        if to[h, 481] == 0 and tg[h, 481] == 0 and h > 0 and not to[h].any() and not tg[h].any():
            break
        for name, a, b in FIELDS:
            x, y = tg[h, a:b], to[h, a:b]
            if not np.array_equal(x.view(np.uint64), y.view(np.uint64)):
                bad = np.flatnonzero(x.view(np.uint64) != y.view(np.uint64))
                rel = np.abs(x[bad] - y[bad]) / np.maximum(np.abs(y[bad]), 1e-300)
                print(f"{tag}: hyp {h}: first differing field {name}: {len(bad)} of {b - a} differ, max rel {rel.max():.3e}; gpu {x[bad][:3]} oracle {y[bad][:3]}")
                return h, name
    print(f"{tag}: traces identical over the hypotheses run")
    return None


def ulp_report(ctx):
    rng = np.random.default_rng(5)
    n = 1 << 20
    a = np.concatenate([rng.uniform(-4, 4, n // 2), rng.lognormal(0, 8, n // 2) * rng.choice([-1, 1], n // 2)])
    b = rng.lognormal(0, 6, n) * rng.choice([-1, 1], n)
    out = np.zeros((8, n))
    ctx.check(_lib.lib().bh_debug_f64_ops(ctx.h, _lib.ptr(a), _lib.ptr(b), n, _lib.ptr(out)))
    with np.errstate(all="ignore"):
        ref = [a / b, np.sqrt(np.abs(a)), np.sin(a), np.cos(a), np.arccos(np.clip(a, -1, 1)), np.log(np.abs(a)), np.abs(a) ** 5.0, np.rint(a)]
    for name, g, r in zip(["div", "sqrt", "sin", "cos", "acos", "log", "pow5", "rint"], out, ref):
        fin = np.isfinite(r) & np.isfinite(g)
        d = np.abs(g[fin].view(np.int64) - r[fin].view(np.int64))
        print(f"f64 {name}: {int((d != 0).sum())} of {int(fin.sum())} differ from host libm, max {int(d.max())} ulp")


def main():
    ctx = Context(0)
    ulp_report(ctx)
    g = np.load(os.path.join(ROOT, "tests", "golden", "config1_golden.npz"))
    nfr = len(g["state"])
    bgr, depth = synth.render_batch(nfr, traj="config1")
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=frontend.make_params(nfeatures=600))
    xy, cloud = frontend.kp_backproject(ctx, kps, n, depth, *K4)
    fails = []

    def match_fn(p, f):
        return oracle.match_crosscheck(desc[p, :n[p]], desc[f, :n[f]])

    def pnp2_fn(a, b, b1, a1):
        S = len(a)
        rv, tv, inl, ni, ok = pnp.solve_pnp_ransac_batch(ctx, np.stack([a, b1]), np.stack([b, a1]), np.array([S, S], np.int32), K4)
        out = (oracle.solve_pnp_ransac(a, b, K4), oracle.solve_pnp_ransac(b1, a1, K4))
        for k, (o, i) in enumerate(((a, b), (b1, a1))):
            same = np.array_equal(inl[k][:ni[k]], out[k][3])
            err = max(np.abs(rv[k] - out[k][1]).max(), np.abs(tv[k] - out[k][2]).max())
            if not same or err > 1e-9:
                print(f"problem {len(fails)}: n {S} inliers gpu {ni[k]} oracle {len(out[k][3])} same {same} pose err {err:.3e}")
                fails.append((o.copy(), i.copy()))
        return out

    two_way_tracking.replay(desc, n, xy, cloud, match_fn, pnp2_fn, oracle.rodrigues)
    print(f"{len(fails)} differing problems of 198")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "pnp_fail.npz"), **{f"obj{i}": f[0] for i, f in enumerate(fails)},
             **{f"img{i}": f[1] for i, f in enumerate(fails)})
    for i, (o, im) in enumerate(fails[:12]):
        tg, rv, tv, inl = gpu_trace(ctx, o, im)
        to, out = oracle_trace(o, im)
        compare(tg, to, f"problem {i}")
        print(f"   final: gpu rvec {rv} tvec {tv} | oracle rvec {out[1]} tvec {out[2]} | inl xor {np.setxor1d(inl, out[3])}")
    ctx.close()


if __name__ == "__main__":
    main()
