"""Frame-sharded front end: two ranks (gloo, both on this box's one GPU) each run ORB on their half of one
sequence; rank 1 receives rank 0's last descriptor row point to point and matches the pair across the shard boundary.  All 2B-1
pairs must equal the single-process result."""
import os, socket, subprocess, sys
import numpy as np
import pytest

from boslam_amd import frontend, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cross_shard_pair_equals_single_process(ctx, tmp_path):
    B = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_halo_worker.py"), str(tmp_path), str(B)], env=dict(env, RANK=str(r)))
             for r in range(2)]
    try:
        rcs = [p.wait(timeout=300) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0, 0]
    # single process over the whole sequence
    bgr, depth = synth.render_batch(2 * B)
    params = frontend.make_params(nfeatures=600)
    kps, desc, n = frontend.orb_detect_describe(ctx, bgr, depth=depth, params=params)
    qi, ti, d, nm = frontend.match_pairs(ctx, desc[1:], n[1:], desc[:-1], n[:-1])             # pair i-1: query frame i, train frame i-1
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    for k in range(2):
        assert np.array_equal(r[k]["n_kp"], n[k * B:(k + 1) * B])
        for f in range(B):
            g = k * B + f
            assert np.array_equal(r[k]["desc"][f, :n[g]], desc[g, :n[g]])
            if g == 0:
                continue                                                                     # frame 0 has no predecessor
            m = nm[g - 1]
            assert r[k]["n_match"][f] == m and m > 100, (k, f, r[k]["n_match"][f], m)
            for name, ref in (("qidx", qi), ("tidx", ti), ("dist", d)):
                assert np.array_equal(r[k][name][f, :m], ref[g - 1, :m]), (k, f, name)
