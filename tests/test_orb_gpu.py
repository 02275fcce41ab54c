"""GPU parity: the HIP ORB pipeline vs the CPU oracle (bit-exact), stage by stage and end to end."""
import numpy as np
import pytest

import oracle
from boslam_amd import frontend, synth

pytestmark = pytest.mark.gpu


def _oracle_frame(bgr, depth, nfeatures, debug=False):
    P = oracle.orb_params(nfeatures=nfeatures)
    mask = None if depth is None else (depth > 0).astype(np.uint8) * 255
    return oracle.orb_detect_describe(bgr, mask, P, max_kp=4 * nfeatures + 256, debug=debug)


def _unpitch(buf, g, key_off="off"):
    """pitched device pyramid -> tight per-level arrays"""
    out = []
    for l in range(len(g["w"])):
        w, h, p, o = int(g["w"][l]), int(g["h"][l]), int(g["pitch"][l]), int(g["off"][l])
        out.append(buf[o:o + p * h].reshape(h, p)[:, :w])
    return out


@pytest.mark.parametrize("nfeatures", [600, 1000])
def test_orb_intermediates_and_output_match_oracle(ctx, nfeatures):
    B = 3
    bgr, depth = synth.render_batch(B)
    params = frontend.make_params(nfeatures=nfeatures)
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params)
    for f in range(B):
        ok, od, rc, dbg = _oracle_frame(bgr[f], depth[f], nfeatures, debug=True)
        assert rc == 0
        dev = frontend.orb_debug_fetch(ctx, f, 480, 640, params)
        for name in ("pyr", "mask", "nms", "blur"):
            lv = _unpitch(dev[name], dev)
            for l in range(params.nlevels):
                w, h, o = int(dbg["w"][l]), int(dbg["h"][l]), int(dbg["off"][l])
                ref = dbg[name][o:o + w * h].reshape(h, w)
                assert lv[l].shape == ref.shape
                bad = np.argwhere(lv[l] != ref)
                assert len(bad) == 0, f"{name} level {l} frame {f}: {len(bad)} px differ, first {bad[:3]}"
        assert n[f] == len(ok), (n[f], len(ok))
        k = kps[f, :n[f]]
        for fld in ("x", "y", "size", "response", "angle"):
            assert np.array_equal(k[fld].view(np.uint32), ok[fld].view(np.uint32)), fld
        assert np.array_equal(k["octave"], ok["octave"])
        assert np.array_equal(desc[f, :n[f]], od)


def test_orb_gray_input_no_mask(ctx):
    bgr, _ = synth.render_batch(2)
    gray = np.ascontiguousarray(bgr[..., 1])
    params = frontend.make_params(nfeatures=500)
    kps, desc, n = frontend.orb_detect_describe(ctx, gray, params=params)
    for f in range(2):
        ok, od, rc = oracle.orb_detect_describe(gray[f], None, oracle.orb_params(nfeatures=500), max_kp=2000)
        assert n[f] == len(ok)
        assert np.array_equal(desc[f, :n[f]], od)
        assert np.array_equal(kps[f, :n[f]]["angle"].view(np.uint32), ok["angle"].view(np.uint32))


def test_match_crosscheck_matches_oracle(ctx):
    rng = np.random.default_rng(5)
    cases = [(1000, 1000), (600, 600), (600, 2500), (1, 1), (37, 5), (5, 37), (0, 10), (10, 0), (1500, 700)]
    Sq = max(c[0] for c in cases); St = max(c[1] for c in cases)
    q = rng.integers(0, 256, size=(len(cases), Sq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, size=(len(cases), St, 32), dtype=np.uint8)
    # plant near-duplicates and exact ties so the first-minimum rules are exercised
    for b, (nq, nt) in enumerate(cases):
        for k in range(min(nq, nt) // 2):
            t[b, (k * 7) % nt] = q[b, (k * 3) % nq] ^ np.uint8(1 << (k % 8))
        if nt > 3 and nq > 3:
            t[b, 1] = t[b, 0]; q[b, 2] = q[b, 1]
        if nt > 40 and nq > 40:                                   # all-ones / all-zeros rows: |q| - 2 q.t goes down to -256 in the MFMA form
            q[b, 30] = 255; t[b, 31] = 255; q[b, 33] = 0; t[b, 35] = 0; t[b, 36] = 255; q[b, 38] = 255
    nq = np.array([c[0] for c in cases], np.int32); nt = np.array([c[1] for c in cases], np.int32)
    qi, ti, d, nm = frontend.match_pairs(ctx, q, nq, t, nt)
    for b, (a, c) in enumerate(cases):
        oq, ot, od = oracle.match_crosscheck(q[b, :a], t[b, :c])
        assert nm[b] == len(oq), (b, nm[b], len(oq))
        assert np.array_equal(qi[b, :nm[b]], oq) and np.array_equal(ti[b, :nm[b]], ot) and np.array_equal(d[b, :nm[b]], od)


def test_backproject_matches_oracle(ctx):
    bgr, depth = synth.render_batch(2)
    params = frontend.make_params(nfeatures=600)
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params)
    xy, cloud = frontend.kp_backproject(ctx, kps, n, depth, synth.FX, synth.FY, synth.CX, synth.CY)
    for f in range(2):
        oxy = oracle.kp_to_int(kps[f, :n[f]])
        assert np.array_equal(xy[f, :n[f]], oxy)
        oc = oracle.backproject(depth[f], oxy, synth.FX, synth.FY, synth.CX, synth.CY)
        assert np.array_equal(cloud[f, :n[f]], oc)
        assert (oc[:, 2] > 0).all()            # the ORB mask keeps keypoints off depth holes


def test_medoid_descriptor_matches_oracle_and_numpy(ctx):
    rng = np.random.default_rng(12)
    M = 700
    desc = rng.integers(0, 256, size=(M, 10, 32), dtype=np.uint8)
    k = rng.integers(0, 11, size=M).astype(np.int32)
    for m in range(0, M, 5):                                           # clustered observations: near-duplicates and exact ties
        base = desc[m, 0].copy()
        for j in range(1, 10):
            desc[m, j] = base ^ (rng.random(32) < 0.03 * j).astype(np.uint8) * np.uint8(1 << (j % 8))
    desc[3, 1] = ~desc[3, 0]                                           # distance 256 wraps to 0 in the reference's uint8 matrix
    k[3] = 2
    got = frontend.medoid_descriptors(ctx, desc, k)
    for m in range(M):
        ref = oracle.medoid_index(desc[m, :k[m]]) if k[m] > 0 else -1
        assert got[m] == ref, (m, k[m], got[m], ref)
        if 0 < k[m] and m % 7 == 0:                                    # numpy restatement of nodes.py:147-153
            d = np.zeros((k[m], k[m]), np.uint8)
            for a in range(k[m]):
                for b in range(a + 1, k[m]):
                    d[a, b] = d[b, a] = np.unpackbits(desc[m, a] ^ desc[m, b]).sum() & 255
            assert got[m] == int(np.argmin(np.median(d, axis=0)))


def test_config2_full_size_batch(ctx):
    """BASELINE configs[1] at its full size: one 1000-frame batch through both entry modes.  40 distinct rendered frames are tiled
    to 1000 (shuffled); the oracle pins the 40, and size-independent properties pin the rest: a frame's result does not depend on
    where it sits in the batch (chunks of 128 / 256 frames, up to four stream pipelines), the HBM-resident path (BH_MEM_DEVICE)
    equals the host path bit for bit, and a frame matched against itself is the identity with distance 0."""
    import torch
    nd, B = 40, 1000
    bgr, depth = synth.render_batch(nd)
    rng = np.random.default_rng(7)
    pick = np.concatenate([np.arange(nd), rng.integers(0, nd, B - nd)])
    rng.shuffle(pick)
    params = frontend.make_params(nfeatures=1000)
    big_bgr = bgr[pick]; big_depth = depth[pick]
    kps, desc, n = frontend.orb_detect_describe(ctx, big_bgr, depth=big_depth, params=params)
    P = oracle.orb_params(nfeatures=1000)
    first = {}
    for i, f in enumerate(pick):
        if f not in first:
            first[f] = i
            okp, odesc, rc = oracle.orb_detect_describe(bgr[f], (depth[f] > 0).astype(np.uint8) * 255, P)
            assert rc == 0 and n[i] == len(okp) and np.array_equal(desc[i, :n[i]], odesc)
            for fld in ("x", "y", "angle", "response", "size", "octave"):
                assert np.array_equal(kps[fld][i, :n[i]].view(np.uint32), okp[fld].view(np.uint32)), fld
        else:
            j = first[f]
            assert n[i] == n[j] and np.array_equal(desc[i], desc[j]) and np.array_equal(kps[i].tobytes(), kps[j].tobytes())
    # HBM-resident path, consecutive-frame matching included
    fe = frontend.DeviceFrontEnd(ctx, B, 480, 640, params)
    fe.run(torch.from_numpy(big_bgr).cuda(), torch.from_numpy(big_depth.view(np.int16)).cuda())
    torch.cuda.synchronize(); ctx.synchronize(); ctx.status()
    assert np.array_equal(fe.n_kp.cpu().numpy(), n) and np.array_equal(fe.desc.cpu().numpy(), desc)
    nm = fe.n_match.cpu().numpy(); qi = fe.qidx.cpu().numpy(); ti = fe.tidx.cpu().numpy(); dd = fe.dist.cpu().numpy()
    same = np.nonzero(pick[1:] == pick[:-1])[0] + 1                       # a frame followed by its own copy
    assert len(same) > 5
    for i in same:
        # every match of a frame against itself has distance 0 and pairs identical descriptors (duplicates may swap partners)
        assert nm[i] > 0 and (dd[i, :nm[i]] == 0).all()
        assert np.array_equal(desc[i][qi[i, :nm[i]]], desc[i - 1][ti[i, :nm[i]]])
    for i in (1, 500, 999):                                               # and the matcher against the oracle on a few pairs
        oq, ot, od = oracle.match_crosscheck(desc[i, :n[i]], desc[i - 1, :n[i - 1]])
        assert np.array_equal(qi[i, :nm[i]], oq) and np.array_equal(ti[i, :nm[i]], ot) and np.array_equal(dd[i, :nm[i]], od)


@pytest.mark.parametrize("B", [9, 21, 141])
def test_ragged_batches_do_not_depend_on_the_launch_mapping(ctx, B):
    """Batch sizes that are not multiples of eight (the kernels tie frames to XCDs by launch index when they are, and fall back when not;
    k_describe's frame-per-step layout leaves part of its last wave empty): every frame's keypoints and descriptors equal those of the
    same image run alone, through the host path and through the HBM-resident path (141 = one full 128-frame chunk + a ragged one)."""
    import torch
    nd = 5
    bgr, depth = synth.render_batch(nd)
    params = frontend.make_params(nfeatures=600)
    ref = []
    for f in range(nd):
        k1, d1, n1 = frontend.orb_detect_describe(ctx, bgr[f:f + 1], depth=depth[f:f + 1], params=params)
        ref.append((k1[0, :n1[0]].tobytes(), d1[0, :n1[0]].copy(), int(n1[0])))             # (rows beyond n_kp are not written)
    pick = (np.arange(B) * 3 + 1) % nd
    big_bgr = np.ascontiguousarray(bgr[pick]); big_depth = np.ascontiguousarray(depth[pick])
    kps, desc, n = frontend.orb_detect_describe(ctx, big_bgr, depth=big_depth, params=params)
    for i, f in enumerate(pick):
        assert n[i] == ref[f][2] and np.array_equal(desc[i, :n[i]], ref[f][1]) and kps[i, :n[i]].tobytes() == ref[f][0], (i, f)
    fe = frontend.DeviceFrontEnd(ctx, B, 480, 640, params)
    fe.run(torch.from_numpy(big_bgr).cuda(), torch.from_numpy(big_depth.view(np.int16)).cuda())
    torch.cuda.synchronize(); ctx.synchronize(); ctx.status()
    dn = fe.n_kp.cpu().numpy(); dd = fe.desc.cpu().numpy()
    assert np.array_equal(dn, n) and all(np.array_equal(dd[i, :n[i]], desc[i, :n[i]]) for i in range(B))


def test_plain_c_consumer_of_the_abi_matches_the_python_mirror(ctx, tmp_path):
    """tests/c_abi/consumer.c - a C99 program that dlopens the library and runs detectAndCompute on two frames + one cross-check match,
    as a cgo / JNI / C host of the reference's call sites would - must print the counts and checksums the Python mirror produces."""
    import os, subprocess
    from boslam_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "consumer"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), "-o", str(exe),
                        os.path.join(root, "tests", "c_abi", "consumer.c"), "-ldl"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    bgr, depth = synth.render_batch(2)
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    with open(tmp_path / "frames.bin", "wb") as f:
        f.write(np.ascontiguousarray(bgr, dtype=np.uint8).tobytes()); f.write(depth.tobytes())
    r = subprocess.run([str(exe), _lib.LIB_PATH, str(tmp_path / "frames.bin"), "480", "640"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr)
    tok = r.stdout.split()
    got = dict(n_kp=(int(tok[1]), int(tok[2])), n_match=int(tok[4]), kps=tok[6], desc=tok[8], match=tok[10])

    def fnv(b, h):
        for x in bytes(b):
            h = ((h ^ x) * 1099511628211) & 0xffffffffffffffff
        return h
    params = frontend.make_params(nfeatures=1000)
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params, max_kp=1200)
    hk = hd = hm = 1469598103934665603
    for f in range(2):
        hk = fnv(kps[f, :n[f]].tobytes(), hk); hd = fnv(desc[f, :n[f]].tobytes(), hd)
    qi, ti, di, nm = frontend.match_pairs(ctx, desc[0:1], n[0:1], desc[1:2], n[1:2])
    for a in (qi, ti, di):
        hm = fnv(np.ascontiguousarray(a[0, :nm[0]], dtype=np.int32).tobytes(), hm)
    assert got["n_kp"] == (int(n[0]), int(n[1])) and got["n_match"] == int(nm[0]) and got["n_match"] > 100
    assert got["kps"] == "%016x" % hk and got["desc"] == "%016x" % hd and got["match"] == "%016x" % hm
