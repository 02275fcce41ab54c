"""The C ABI from two host threads at once (the tracker runs on the main thread, local mapping on a daemon thread,
slam/main.py:37-47; ctypes drops the GIL inside every call).  One bh_ctx per thread, as the header prescribes: thread A runs the
front end + bh_track_refkf, thread B builds and optimises a local window and queries the BoW database, 50 rounds each, concurrently.
Every result must equal the serial run of the same calls bit for bit - the library keeps no process-wide mutable state."""
import threading

import numpy as np
import pytest

from boslam_amd import Context, ba, bow, frontend, synth, tracking

pytestmark = pytest.mark.gpu
ROUNDS = 50


def _tracker_work(c, bgr, depth, params, pairs):
    kps, desc, n = frontend.orb_detect_describe(c, bgr, depth=depth, params=params)
    t = tracking.track_refkf(c, pairs["q_desc"], pairs["nq"], pairs["q_xy"], pairs["t_desc"], pairs["nt"], pairs["t_pts"], pairs["K"])
    return [n.copy(), desc[0, :n[0]].copy(), kps[1, :n[1]]["angle"].view(np.uint32).copy(), t["status"].copy(), t["n_inlier"].copy(),
            t["R"].copy(), t["t"].copy()]


def _mapper_work(c, pr, db):
    s = ba.LocalBA(c, pr["K"])
    s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
    s.optimize(5)
    _, tt = s.poses(); pts = s.points()
    s.close()
    d = bow.BowDatabase(c)
    d.set_db(db["start"], db["words"], db["vals"])
    out = [tt.copy(), pts.copy()]
    for mode in ("scan", "inverted"):
        d.set_mode(mode)
        ids, sc = d.query(db["q_words"], db["q_vals"], 5)
        out += [ids.copy(), sc.copy()]
    d.close()
    return out


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def test_two_threads_two_contexts_equal_the_serial_run():
    bgr, depth = synth.render_batch(2)
    params = frontend.make_params(nfeatures=500)
    pairs = synth.tracking_pairs(B=3, n_ref=500, n_frame=500, seed=21)
    pr = synth.ba_problem(n_kf=6, n_pts=150, seed=9)
    db = synth.bow_database(n_entries=2000, words_per_entry=120, query_entry=77, seed=3)
    ca, cb = Context(0), Context(0)
    try:
        ref_a = _tracker_work(ca, bgr, depth, params, pairs)
        ref_b = _mapper_work(cb, pr, db)
        bad, errs = [], []
        start = threading.Barrier(2)

        def run(fn, ref, tag):
            try:
                start.wait()
                for i in range(ROUNDS):
                    if not _same(fn(), ref):
                        bad.append((tag, i))
            except Exception as e:                                   # a worker's exception must fail the test, not vanish with the thread
                errs.append((tag, repr(e)))

        ta = threading.Thread(target=run, args=(lambda: _tracker_work(ca, bgr, depth, params, pairs), ref_a, "tracker"))
        tb = threading.Thread(target=run, args=(lambda: _mapper_work(cb, pr, db), ref_b, "mapper"), daemon=True)
        ta.start(); tb.start(); ta.join(600); tb.join(600)
        assert not ta.is_alive() and not tb.is_alive()
        assert not errs, errs
        assert not bad, f"results changed under concurrency: {bad[:5]}"
    finally:
        ca.close(); cb.close()


def test_default_context_is_per_thread():
    from boslam_amd import _lib
    got = {}
    alive = threading.Barrier(4)                                   # all four alive together (a finished thread's ident may be handed out again)

    def grab(k):
        got[k] = _lib.default_context(0)
        alive.wait()

    ts = [threading.Thread(target=grab, args=(i,)) for i in range(4)]
    for t in ts: t.start()
    for t in ts: t.join()
    got["main"] = _lib.default_context(0)
    assert len({id(c) for c in got.values()}) == 5                  # one context per calling thread
    assert _lib.default_context(0) is got["main"]
