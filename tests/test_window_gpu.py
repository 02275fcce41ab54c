"""Local-BA window / problem construction: HIP against the line-by-line Python restatement."""
import numpy as np
import pytest

import oracle
from boslam_amd import window

pytestmark = pytest.mark.gpu


def random_graph(K, n_mp, seed, sym=True, obs_per_kf=(40, 400), deg=(0, 12)):
    rng = np.random.default_rng(seed)
    adj = [dict() for _ in range(K)]
    for i in range(K):
        for j in rng.choice(K, size=rng.integers(*deg) if deg[1] > deg[0] else deg[0], replace=False):
            j = int(j)
            if j == i:
                continue
            c = int(rng.integers(0, 40))
            adj[i][j] = c
            if sym:
                adj[j][i] = c
            elif rng.random() < 0.5:
                adj[j][i] = int(rng.integers(0, 40))                      # asymmetric bookkeeping: exercises the `outer` branch
    nbr_start = np.zeros(K + 1, np.int32); nbr_id, nbr_common = [], []
    for i in range(K):
        ks = list(adj[i].keys()); rng.shuffle(ks)                           # neighbour lists in arbitrary order
        nbr_id += ks; nbr_common += [adj[i][k] for k in ks]; nbr_start[i + 1] = len(nbr_id)
    obs_start = np.zeros(K + 1, np.int32); obs_mp, obs_px = [], []
    for i in range(K):
        n = int(rng.integers(*obs_per_kf))
        centre = int(rng.integers(0, n_mp))
        mps = np.unique((centre + rng.integers(-600, 600, n)) % n_mp)       # nearby keyframes share map points
        rng.shuffle(mps)
        obs_mp += [int(m) for m in mps]; obs_px += [(int(rng.integers(0, 640)), int(rng.integers(0, 480))) for _ in mps]
        obs_start[i + 1] = len(obs_mp)
    return (nbr_start, np.array(nbr_id, np.int32), np.array(nbr_common, np.int32), obs_start, np.array(obs_mp, np.int32),
            np.array(obs_px, np.int64).reshape(-1, 2))


def same(a, b):
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


@pytest.mark.parametrize("K,n_mp,sym,seed", [(30, 3000, True, 1), (200, 20000, True, 2), (200, 20000, False, 3), (1500, 100000, False, 4)])
def test_window_problem_matches_restatement(ctx, K, n_mp, sym, seed):
    g = random_graph(K, n_mp, seed, sym)
    hits = 0
    for kf in np.random.default_rng(seed).choice(K, size=12, replace=False):
        o = oracle.window_problem(*g, kf=int(kf))
        h = window.window_problem(ctx, *g, n_mp=n_mp, kf=int(kf))
        same(o, h)
        hits += len(o["edge_pose"]) > 0
        if sym:
            assert o["n_outer"] == 0            # symmetric bookkeeping: the reference's outer set is always inside inner
    assert hits >= 3


def test_window_edge_cases(ctx):
    # an isolated keyframe, a keyframe whose neighbours all share < 15 points ("nothing to optimize"), empty observation lists
    nbr_start = np.array([0, 0, 2, 3, 4], np.int32); nbr_id = np.array([2, 3, 1, 1], np.int32); nbr_common = np.array([3, 14, 3, 14], np.int32)
    obs_start = np.array([0, 0, 3, 3, 5], np.int32); obs_mp = np.array([5, 6, 7, 6, 9], np.int32)
    obs_px = np.arange(10, dtype=np.int64).reshape(5, 2)
    for kf in range(4):
        o = oracle.window_problem(nbr_start, nbr_id, nbr_common, obs_start, obs_mp, obs_px, kf=kf)
        h = window.window_problem(ctx, nbr_start, nbr_id, nbr_common, obs_start, obs_mp, obs_px, n_mp=10, kf=kf)
        same(o, h)
        assert len(h["win_kf"]) == 0
    nbr_common[:] = 20                           # now keyframe 1 has two inner neighbours; map point 6 is seen twice
    o = oracle.window_problem(nbr_start, nbr_id, nbr_common, obs_start, obs_mp, obs_px, kf=1)
    h = window.window_problem(ctx, nbr_start, nbr_id, nbr_common, obs_start, obs_mp, obs_px, n_mp=10, kf=1)
    same(o, h)
    assert list(h["win_kf"]) == [1, 2, 3] and list(h["win_fixed"]) == [True, False, False] and list(h["pt_mp"]) == [6] and len(h["edge_pose"]) == 2


def test_window_feeds_local_ba(ctx):
    """End to end: window -> edge list -> bh_ba_set_problem runs (poses / points from a synthetic sequence)."""
    from boslam_amd import ba, synth
    pr = synth.ba_problem_sequence(n_kf=60, n_pts=2500, window=6, seed=9)
    K = 60
    # covisibility CSR from the observations
    by_kf = [set(pr["ept"][pr["ecam"] == i].tolist()) for i in range(K)]
    nbr_start = np.zeros(K + 1, np.int32); nbr_id, nbr_common = [], []
    for i in range(K):
        for j in range(K):
            c = len(by_kf[i] & by_kf[j]) if i != j else 0
            if c > 0:
                nbr_id.append(j); nbr_common.append(c)
        nbr_start[i + 1] = len(nbr_id)
    order = np.argsort(pr["ecam"], kind="stable")
    obs_start = np.searchsorted(pr["ecam"][order], np.arange(K + 1)).astype(np.int32)
    obs_mp = pr["ept"][order]; obs_px = np.rint(pr["meas"][order]).astype(np.int64)
    w = window.window_problem(ctx, nbr_start, nbr_id, nbr_common, obs_start, obs_mp, obs_px, n_mp=2500, kf=30)
    o = oracle.window_problem(nbr_start, np.array(nbr_id, np.int32), np.array(nbr_common, np.int32), obs_start, obs_mp, obs_px, kf=30)
    same(o, w)
    assert w["n_inner"] >= 4 and w["win_fixed"][0] and not w["win_fixed"][1:].any()
    s = ba.LocalBA(ctx, pr["K"])
    s.set_problem(pr["R"][w["win_kf"]], pr["t"][w["win_kf"]], w["win_fixed"].astype(np.uint8), pr["pts"][w["pt_mp"]], w["edge_pose"],
                  w["edge_point"], w["edge_px"].astype(np.float64))
    done = s.optimize(7)                                                    # :218 optimizer.optimize(7)
    lg = s.log()[:done]
    assert done >= 1 and lg[-1, 0] < lg[0, 0] * 1.0001
    s.close()
