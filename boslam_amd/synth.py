"""Seeded synthetic RGB-D inputs.  Data generation only — no hot-path arithmetic here.

The reference has no synthetic frame source (frames come from a RealSense, camera.py:131-137, or a pickle,
slam/main.py:53-55); tests and bench.py therefore render a textured plane through the reference's default
pinhole intrinsics (config.py:36-41) with a plain numpy bilinear warp.  BA problems follow the recipe of the
reference's only synthetic generator, tests/ba_test.py:166-210, scaled up as SURVEY.md §8d states.
"""
import numpy as np

# config.py:36-41
FX = FY = 384.239013671875
CX, CY = 322.432373046875, 239.6533203125
BASELINE = 0.06
WIDTH, HEIGHT = 640, 480

_TEX = None


def _texture():
    """2048x2048 texture: 5000 overlapping random rectangles, 30..160 px log-uniform sides (seed 2024).

    Chosen so that every one of the 8 pyramid levels yields more FAST candidates than 2*n_l at nfeatures=1000
    (both retainBest stages are exercised and each frame produces exactly 1000 keypoints, BASELINE config 2)."""
    global _TEX
    if _TEX is None:
        rng = np.random.default_rng(2024)
        t = np.full((2048, 2048), 128, np.uint8)
        for _ in range(5000):
            w = int(np.exp(rng.uniform(np.log(30), np.log(160))))
            h = int(np.exp(rng.uniform(np.log(30), np.log(160))))
            x = int(rng.integers(-w // 2, 2048)); y = int(rng.integers(-h // 2, 2048))
            t[max(y, 0):y + h, max(x, 0):x + w] = rng.integers(0, 256)
        _TEX = t.astype(np.float32)
    return _TEX


def render_frame(k, traj="config2", width=WIDTH, height=HEIGHT):
    """Frame k of the plane-at-2m sequence -> (bgr u8[H][W][3], depth u16[H][W])."""
    tex = _texture()
    if traj == "config1":
        tx, ty = 0.005 * k, 0.0
    else:
        tx, ty = 0.002 * k, 0.001 * k
    ppm = 370.0                                  # texture pixels per metre
    u = np.arange(width, dtype=np.float64)
    v = np.arange(height, dtype=np.float64)
    X = 2.0 * (u - CX) / FX + tx                 # plane z = 2 m, R = I
    Y = 2.0 * (v - CY) / FY + ty
    sx = X * ppm + 640.0
    sy = Y * ppm + 480.0
    x0 = np.floor(sx).astype(np.int64); fx = (sx - x0).astype(np.float32)
    y0 = np.floor(sy).astype(np.int64); fy = (sy - y0).astype(np.float32)
    x0 = np.clip(x0, 0, 2046); y0 = np.clip(y0, 0, 2046)
    t00 = tex[np.ix_(y0, x0)]; t01 = tex[np.ix_(y0, x0 + 1)]
    t10 = tex[np.ix_(y0 + 1, x0)]; t11 = tex[np.ix_(y0 + 1, x0 + 1)]
    top = t00 * (1 - fx)[None, :] + t01 * fx[None, :]
    bot = t10 * (1 - fx)[None, :] + t11 * fx[None, :]
    img = top * (1 - fy)[:, None] + bot * fy[:, None]
    img = img + np.random.default_rng(2025 + 7919 * k).normal(0.0, 4.0, size=img.shape).astype(np.float32)
    gains = np.array([1.0, 0.9, 0.8], dtype=np.float32)            # B, G, R
    bgr = np.clip(np.rint(img[:, :, None] * gains[None, None, :]), 0, 255).astype(np.uint8)
    depth = np.full((height, width), 2000, dtype=np.uint16)
    rng = np.random.default_rng(2026 + 104729 * k)        # ~5 % of the pixels without depth, in 12 coherent holes
    for _ in range(12):
        w = int(rng.integers(20, 60)); h = int(rng.integers(15, 45))
        x = int(rng.integers(0, width - w)); y = int(rng.integers(0, height - h))
        depth[y:y + h, x:x + w] = 0
    return np.ascontiguousarray(bgr), depth


def render_batch(n, traj="config2", start=0):
    bgr = np.empty((n, HEIGHT, WIDTH, 3), dtype=np.uint8)
    depth = np.empty((n, HEIGHT, WIDTH), dtype=np.uint16)
    for i in range(n):
        bgr[i], depth[i] = render_frame(start + i, traj)
    return bgr, depth


def random_descriptors(seed, n):
    """utils.py:53-55 int2orb-style synthetic descriptors: u8[n][32]."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)


# ------------------------------------------------------------------------------------------------ BA problems
def _look_at(pos, target=np.zeros(3)):
    """Camera-in-world rotation whose +z axis looks from pos to target (x right, y down)."""
    z = target - pos; z = z / np.linalg.norm(z)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(up, z)
    if np.linalg.norm(x) < 1e-9:
        x = np.array([1.0, 0.0, 0.0])
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=1)          # columns = camera axes in world


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def project(R, t, X, K=(FX, FY, CX, CY)):
    """SBACam convention: (R, t) camera-in-world; pixel of world point X."""
    pc = (X - t) @ R                      # R' (X - t)
    return np.stack([K[0] * pc[..., 0] / pc[..., 2] + K[2], K[1] * pc[..., 1] / pc[..., 2] + K[3]], axis=-1), pc[..., 2]


def ba_problem(n_kf=50, n_pts=10000, obs_per_pt=8, seed=7, radius=4.0, pix_noise=1.0, rot_noise=0.01,
               trans_noise=0.02, pt_noise=0.02, n_fixed=2, outlier_frac=0.0):
    """SURVEY.md §8(d) config 3 generator (defaults) — a scaled-up cousin of tests/ba_test.py:166-210.
    Returns dict with ground truth, perturbed initial state, edges and measurements."""
    rng = np.random.default_rng(seed)
    ang = 2 * np.pi * np.arange(n_kf) / n_kf
    pos = np.stack([radius * np.cos(ang), radius * np.sin(ang), 0.3 * np.sin(3 * ang)], axis=1)
    Rgt = np.stack([_look_at(p) for p in pos]); tgt = pos.copy()
    Xgt = rng.uniform(-1, 1, size=(n_pts, 3))
    ecam, ept, meas = [], [], []
    for j in range(n_pts):
        uv, z = zip(*[project(Rgt[i], tgt[i], Xgt[j]) for i in range(n_kf)]) if n_kf <= 8 else (None, None)
        if uv is None:
            pc = np.einsum('cij,ci->cj', Rgt, Xgt[j][None, :] - tgt)           # R'(X - t) per camera
            uvs = np.stack([FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY], axis=1); zs = pc[:, 2]
        else:
            uvs = np.stack(uv); zs = np.array(z)
        vis = np.where((zs > 0.1) & (uvs[:, 0] >= 0) & (uvs[:, 0] < WIDTH) & (uvs[:, 1] >= 0) & (uvs[:, 1] < HEIGHT))[0]
        if len(vis) < 2:
            continue
        sel = rng.choice(vis, size=min(obs_per_pt, len(vis)), replace=False)
        for i in np.sort(sel):
            ecam.append(i); ept.append(j); meas.append(uvs[i] + rng.normal(0, pix_noise, 2))
    ecam = np.array(ecam, np.int32); ept = np.array(ept, np.int32); meas = np.array(meas)
    if outlier_frac > 0:
        bad = rng.random(len(meas)) < outlier_frac
        meas[bad] += rng.normal(0, 40.0, size=(int(bad.sum()), 2))
    R0 = Rgt.copy(); t0 = tgt.copy()
    for i in range(n_fixed, n_kf):
        R0[i] = Rgt[i] @ _rodrigues(rng.normal(0, rot_noise, 3)); t0[i] = tgt[i] + rng.normal(0, trans_noise, 3)
    X0 = Xgt + rng.normal(0, pt_noise, Xgt.shape)
    fixed = np.zeros(n_kf, np.uint8); fixed[:n_fixed] = 1
    return dict(Rgt=Rgt, tgt=tgt, Xgt=Xgt, R=R0, t=t0, pts=X0, fixed=fixed, ecam=ecam, ept=ept, meas=meas,
                K=np.array([FX, FY, CX, CY, BASELINE]))


def pose_problem(n=200, seed=3, outlier_frac=0.2, int_pixels=True):
    """One CamOnlyBA problem (slam/tracking.py:73): n 3-D points in front of a camera a few cm from identity,
    pixels truncated to int like kp_arr (camera.py:142), a fraction of gross outliers."""
    rng = np.random.default_rng(seed)
    R = _rodrigues(rng.normal(0, 0.03, 3)); t = rng.normal(0, 0.05, 3)
    X = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(1.0, 4.0, n)], axis=1)
    uv, z = project(R, t, X)
    uv = uv + rng.normal(0, 0.7, uv.shape)
    bad = rng.random(n) < outlier_frac
    uv[bad] += rng.normal(0, 60.0, size=(int(bad.sum()), 2))
    if int_pixels:
        uv = np.trunc(uv)
    return dict(pix=uv.astype(np.float64), pts=X, R=R, t=t, outlier=bad, K=np.array([FX, FY, CX, CY, BASELINE]))


def tracking_pairs(B=8, n_ref=600, n_frame=600, seed=5, frac_seen=0.7, flip_bits=6, outlier_frac=0.1):
    """B (frame, reference keyframe) pairs shaped like Tracker._track_wrt_refkf's inputs (slam/tracking.py:54-73): the reference holds
    n_ref descriptors with 3-D points (kf_ref.cloud_kp), the frame re-observes a fraction of them from a pose a few cm away - descriptors
    with a few flipped bits, integer pixels (camera.py:142) - plus unrelated features; some re-observations carry a wrong pixel.
    Returns q_desc u8[B][n_frame][32], nq, q_xy i64[B][n_frame][2], t_desc u8[B][n_ref][32], nt, t_pts f64[B][n_ref][3], K, poses."""
    rng = np.random.default_rng(seed)
    q_desc = rng.integers(0, 256, (B, n_frame, 32), dtype=np.uint8); t_desc = rng.integers(0, 256, (B, n_ref, 32), dtype=np.uint8)
    q_xy = np.zeros((B, n_frame, 2), np.int64); t_pts = np.zeros((B, n_ref, 3))
    nq = np.zeros(B, np.int32); nt = np.zeros(B, np.int32)
    Rs, ts = [], []
    for b in range(B):
        nt[b] = n_ref - (b % 3) * 7; nq[b] = n_frame - (b % 4) * 5
        R = _rodrigues(rng.normal(0, 0.03, 3)); t = rng.normal(0, 0.05, 3)
        X = np.stack([rng.uniform(-1.5, 1.5, n_ref), rng.uniform(-1.0, 1.0, n_ref), rng.uniform(1.0, 4.0, n_ref)], axis=1)
        t_pts[b] = X
        uv, _ = project(R, t, X)
        uv = uv + rng.normal(0, 0.7, uv.shape)
        q_xy[b] = rng.integers(0, 480, (n_frame, 2))
        seen = rng.permutation(int(nt[b]))[:int(frac_seen * min(nq[b], nt[b]))]
        slots = rng.permutation(int(nq[b]))[:len(seen)]
        d = t_desc[b, seen].copy()
        for k in range(len(seen)):
            for bit in rng.integers(0, 256, rng.integers(0, flip_bits + 1)):
                d[k, bit >> 3] ^= np.uint8(1 << (bit & 7))
        q_desc[b, slots] = d
        px = uv[seen].copy()
        bad = rng.random(len(seen)) < outlier_frac
        px[bad] += rng.normal(0, 60.0, (int(bad.sum()), 2))
        q_xy[b, slots] = np.trunc(px).astype(np.int64)
        Rs.append(R); ts.append(t)
    return dict(q_desc=q_desc, nq=nq, q_xy=q_xy, t_desc=t_desc, nt=nt, t_pts=t_pts, K=np.array([FX, FY, CX, CY, BASELINE]), R=np.stack(Rs), t=np.stack(ts))


def local_map_frames(B=6, n_frame=600, n_mp=900, seed=9, dup_frac=0.35, flip_bits=6, outlier_frac=0.1, away_frac=0.2):
    """B frames shaped like Tracker._track_local_map's inputs (slam/tracking.py:91-136): the local map's candidates in loop order - every
    (local keyframe, its map point) pair, so `dup_frac` of the n_mp map points appear two or three times - with medoid descriptors, points
    and viewing normals (a share `away_frac` faces away from the camera and fails the cos60 test, some points lie outside the image);
    the frame re-observes a part of the map (descriptors with a few flipped bits, int pixels, some wrong pixels) from a pose near identity.
    Returns q_desc, nq, q_xy, R, t (frame.R, frame.t), mp_pts, mp_normals, mp_feat, mp_id (the map point of each candidate), M, K."""
    rng = np.random.default_rng(seed)
    Sm = int(n_mp * (1 + 2 * dup_frac)) + 8
    q_desc = rng.integers(0, 256, (B, n_frame, 32), dtype=np.uint8)
    q_xy = rng.integers(0, 480, (B, n_frame, 2)).astype(np.int64)
    mp_pts = np.zeros((B, Sm, 3)); mp_nrm = np.zeros((B, Sm, 3)); mp_feat = np.zeros((B, Sm, 32), np.uint8); mp_id = -np.ones((B, Sm), np.int64)
    nq = np.zeros(B, np.int32); M = np.zeros(B, np.int32)
    Rs, ts = [], []
    for b in range(B):
        nq[b] = n_frame - (b % 4) * 5
        R = _rodrigues(rng.normal(0, 0.02, 3)); t = rng.normal(0, 0.04, 3)
        X = np.stack([rng.uniform(-2.5, 2.5, n_mp), rng.uniform(-1.8, 1.8, n_mp), rng.uniform(1.0, 4.0, n_mp)], axis=1)
        feat = rng.integers(0, 256, (n_mp, 32), dtype=np.uint8)
        see = R @ np.array([0.0, 0.0, 1.0]); see /= np.linalg.norm(see)
        nrm = -see + rng.normal(0, 0.35, (n_mp, 3))
        away = rng.random(n_mp) < away_frac
        nrm[away] = see + rng.normal(0, 0.2, (int(away.sum()), 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        reps = np.ones(n_mp, np.int64); dup = rng.random(n_mp) < dup_frac; reps[dup] = rng.integers(2, 4, int(dup.sum()))
        order = rng.permutation(np.repeat(np.arange(n_mp), reps))[:Sm]           # loop order: keyframe by keyframe, a map point wherever its keyframes come up
        M[b] = len(order)
        mp_pts[b, :M[b]] = X[order]; mp_nrm[b, :M[b]] = nrm[order]; mp_feat[b, :M[b]] = feat[order]; mp_id[b, :M[b]] = order
        uv, _ = project(R, t, X)
        uv = uv + rng.normal(0, 0.7, uv.shape)
        seen = rng.permutation(n_mp)[:int(0.6 * min(nq[b], n_mp))]
        slots = rng.permutation(int(nq[b]))[:len(seen)]
        d = feat[seen].copy()
        for k in range(len(seen)):
            for bit in rng.integers(0, 256, rng.integers(0, flip_bits + 1)):
                d[k, bit >> 3] ^= np.uint8(1 << (bit & 7))
        q_desc[b, slots] = d
        px = uv[seen].copy()
        bad = rng.random(len(seen)) < outlier_frac
        px[bad] += rng.normal(0, 60.0, (int(bad.sum()), 2))
        q_xy[b, slots] = np.trunc(px).astype(np.int64)
        Rs.append(R); ts.append(t)
    return dict(q_desc=q_desc, nq=nq, q_xy=q_xy, R=np.stack(Rs), t=np.stack(ts), mp_pts=mp_pts, mp_normals=mp_nrm, mp_feat=mp_feat, mp_id=mp_id, M=M,
                K=np.array([FX, FY, CX, CY, BASELINE]))


def ba_problem_frontal(n_kf=6, n_pts=120, seed=21, pix_noise=0.7):
    """Local-window flavour of the live system: keyframes a few cm / degrees from identity looking down +z at points
    2..5 m away (so the reference's 640x480 admission gate, optimizer.py:52-54, admits most edges); int pixels."""
    rng = np.random.default_rng(seed)
    Rgt = np.stack([_rodrigues(rng.normal(0, 0.04, 3)) for _ in range(n_kf)])
    tgt = rng.normal(0, 0.15, size=(n_kf, 3))
    Rgt[0] = np.eye(3); tgt[0] = 0
    Xgt = np.stack([rng.uniform(-2, 2, n_pts), rng.uniform(-1.5, 1.5, n_pts), rng.uniform(2, 5, n_pts)], axis=1)
    ecam, ept, meas = [], [], []
    for j in range(n_pts):
        for i in range(n_kf):
            uv, z = project(Rgt[i], tgt[i], Xgt[j])
            if z > 0.1 and 0 <= uv[0] < WIDTH and 0 <= uv[1] < HEIGHT and rng.random() < 0.8:
                ecam.append(i); ept.append(j); meas.append(np.trunc(uv + rng.normal(0, pix_noise, 2)))
    R0 = Rgt.copy(); t0 = tgt.copy()
    for i in range(1, n_kf):
        R0[i] = Rgt[i] @ _rodrigues(rng.normal(0, 0.01, 3)); t0[i] = tgt[i] + rng.normal(0, 0.02, 3)
    fixed = np.zeros(n_kf, np.uint8); fixed[0] = 1
    return dict(Rgt=Rgt, tgt=tgt, Xgt=Xgt, R=R0, t=t0, pts=Xgt + rng.normal(0, 0.02, Xgt.shape), fixed=fixed,
                ecam=np.array(ecam, np.int32), ept=np.array(ept, np.int32), meas=np.array(meas, np.float64),
                K=np.array([FX, FY, CX, CY, BASELINE]))


# ------------------------------------------------------------------------------------------------ BoW database
def bow_database(n_entries=100000, words_per_entry=1000, vocab=10**6, seed=11, query_entry=12345, redraw=0.2):
    """entries with `words_per_entry` distinct random words out of a 10^6-word vocabulary
    (k=10, L=6), TF-IDF-like weights U(0,1) L1-normalised per entry; query = entry `query_entry` with 20 % of its words
    re-drawn.  Returns CSR (start i64, words u32 ascending per entry, vals f64) and the query (words, vals)."""
    rng = np.random.default_rng(seed)
    words = rng.integers(0, vocab, size=(n_entries, words_per_entry), dtype=np.int64)
    words.sort(axis=1)
    # make the words of an entry distinct: bump duplicates (rare) by re-sorting unique-ified rows lazily
    dup = (np.diff(words, axis=1) == 0).any(axis=1)
    for e in np.nonzero(dup)[0]:
        w = np.unique(words[e])
        while len(w) < words_per_entry:
            w = np.unique(np.concatenate([w, rng.integers(0, vocab, size=words_per_entry - len(w))]))
        words[e] = w
    vals = rng.random((n_entries, words_per_entry))
    vals /= vals.sum(axis=1, keepdims=True)
    start = np.arange(n_entries + 1, dtype=np.int64) * words_per_entry
    qe = min(query_entry, n_entries - 1)
    qw = words[qe].copy(); qv = vals[qe].copy()
    nre = int(redraw * words_per_entry)
    pos = rng.choice(words_per_entry, size=nre, replace=False)
    qw[pos] = rng.integers(0, vocab, size=nre)
    qw, ui = np.unique(qw, return_index=True)
    qv = qv[ui] / qv[ui].sum()
    return dict(start=start, words=words.reshape(-1).astype(np.uint32), vals=vals.reshape(-1), q_words=qw.astype(np.uint32),
                q_vals=qv, query_entry=qe)


def vocabulary(k=10, L=4, seed=21, stop_frac=0.01):
    """Synthetic k-ary vocabulary tree of depth L (the real /DBow3/orbvoc.dbow3, k=10 L=6, is not in the reference repo):
    random 256-bit node descriptors, children stored contiguously in BFS order, leaf weights = idf-like U(0.5, 8), a few
    stop words with weight 0.  Returns the flat arrays bh_voc_create / the oracle take."""
    rng = np.random.default_rng(seed)
    n_nodes = (k ** (L + 1) - 1) // (k - 1)
    n_inner = (k ** L - 1) // (k - 1)
    child_start = np.zeros(n_nodes + 1, np.int32)
    child_start[1:n_inner + 1] = np.arange(1, n_inner + 1) * k
    child_start[n_inner + 1:] = n_inner * k
    child_idx = np.arange(1, n_inner * k + 1, dtype=np.int32)            # node i's children are 1 + i*k .. i*k + k
    # children within a parent in a shuffled order, so that "first minimum in stored order" is exercised
    child_idx = child_idx.reshape(n_inner, k)
    for i in range(n_inner):
        rng.shuffle(child_idx[i])
    child_idx = child_idx.reshape(-1)
    node_desc = rng.integers(0, 256, size=(n_nodes, 32), dtype=np.uint8)
    node_weight = np.zeros(n_nodes); node_word = np.full(n_nodes, -1, np.int32)
    leaves = np.arange(n_inner, n_nodes)
    node_word[leaves] = rng.permutation(len(leaves)).astype(np.int32)
    node_weight[leaves] = rng.uniform(0.5, 8.0, len(leaves))
    node_weight[leaves[rng.random(len(leaves)) < stop_frac]] = 0.0
    return dict(child_start=child_start, child_idx=child_idx, node_desc=node_desc, node_weight=node_weight, node_word=node_word,
                k=k, L=L, n_words=len(leaves))


def ba_problem_sequence(n_kf=5000, n_pts=500000, obs_per_pt=8, window=20, seed=13, pix_noise=1.0, rot_noise=0.005,
                        trans_noise=0.02, pt_noise=0.02, n_fixed=2):
    """SURVEY.md §8(d) config 5 generator: keyframes along a smooth 3-D path, every landmark seen by `obs_per_pt` keyframes
    within +-`window` of a centre keyframe, so the reduced camera system is block-banded (half-bandwidth 2*window).
    Fully vectorised (4 M observations in a few seconds)."""
    rng = np.random.default_rng(seed)
    i = np.arange(n_kf)
    pos = np.stack([0.1 * i, 0.3 * np.sin(0.013 * i), 0.2 * np.sin(0.021 * i + 1.0)], axis=1)
    yaw = 0.15 * np.sin(0.009 * i)
    Rgt = np.zeros((n_kf, 3, 3))
    Rgt[:, 0, 0] = np.cos(yaw); Rgt[:, 0, 2] = np.sin(yaw); Rgt[:, 1, 1] = 1.0; Rgt[:, 2, 0] = -np.sin(yaw); Rgt[:, 2, 2] = np.cos(yaw)
    tgt = pos.copy()
    centre = rng.integers(0, n_kf, size=n_pts)
    local = np.stack([rng.uniform(-1.0, 1.0, n_pts), rng.uniform(-0.7, 0.7, n_pts), rng.uniform(2.5, 5.0, n_pts)], axis=1)
    Xgt = tgt[centre] + np.einsum('pij,pj->pi', Rgt[centre], local)
    offs = np.argsort(rng.random((n_pts, 2 * window + 1)), axis=1)[:, :obs_per_pt] - window
    cams = np.clip(centre[:, None] + offs, 0, n_kf - 1)
    cams = np.sort(cams, axis=1)
    keep = np.ones_like(cams, dtype=bool); keep[:, 1:] = cams[:, 1:] != cams[:, :-1]          # clipping may duplicate a keyframe
    pid = np.repeat(np.arange(n_pts), obs_per_pt).reshape(n_pts, obs_per_pt)
    ecam = cams[keep].astype(np.int32); ept = pid[keep].astype(np.int32)
    pc = np.einsum('eji,ej->ei', Rgt[ecam], Xgt[ept] - tgt[ecam])                               # R'(X - t)
    uv = np.stack([FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY], axis=1)
    vis = (pc[:, 2] > 0.1) & (uv[:, 0] >= 0) & (uv[:, 0] < WIDTH) & (uv[:, 1] >= 0) & (uv[:, 1] < HEIGHT)
    ecam, ept, uv = ecam[vis], ept[vis], uv[vis]
    meas = uv + rng.normal(0, pix_noise, uv.shape)
    R0 = Rgt.copy(); t0 = tgt.copy()
    w = rng.normal(0, rot_noise, (n_kf, 3)); w[:n_fixed] = 0
    th = np.linalg.norm(w, axis=1); th[th < 1e-12] = 1e-12
    k = w / th[:, None]
    Kx = np.zeros((n_kf, 3, 3)); Kx[:, 0, 1] = -k[:, 2]; Kx[:, 0, 2] = k[:, 1]; Kx[:, 1, 0] = k[:, 2]; Kx[:, 1, 2] = -k[:, 0]; Kx[:, 2, 0] = -k[:, 1]; Kx[:, 2, 1] = k[:, 0]
    dR = np.eye(3)[None] + np.sin(th)[:, None, None] * Kx + (1 - np.cos(th))[:, None, None] * (Kx @ Kx)
    R0 = Rgt @ dR
    dt = rng.normal(0, trans_noise, (n_kf, 3)); dt[:n_fixed] = 0
    t0 = tgt + dt
    fixed = np.zeros(n_kf, np.uint8); fixed[:n_fixed] = 1
    return dict(Rgt=Rgt, tgt=tgt, Xgt=Xgt, R=R0, t=t0, pts=Xgt + rng.normal(0, pt_noise, Xgt.shape), fixed=fixed,
                ecam=ecam, ept=ept, meas=meas, K=np.array([FX, FY, CX, CY, BASELINE]), window=window)
