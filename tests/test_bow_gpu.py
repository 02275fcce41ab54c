"""GPU parity: local-map visibility gate and DBoW3-style L1 query vs the CPU oracle (indices exact, scores bit-exact)."""
import numpy as np
import pytest

import oracle
from boslam_amd import bow, synth

pytestmark = pytest.mark.gpu


def test_visibility_gate_matches_oracle(ctx):
    rng = np.random.default_rng(2)
    for M in (1, 77, 1024, 5000):
        pts = np.stack([rng.uniform(-3, 3, M), rng.uniform(-2, 2, M), rng.uniform(-1, 6, M)], 1)
        nrm = rng.normal(size=(M, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        R = synth._rodrigues(rng.normal(0, 0.1, 3)); t = rng.normal(0, 0.2, 3)
        see = R @ np.array([0.0, 0.0, 1.0]); see /= np.linalg.norm(see)            # camera.py:26-29
        pts[0] = [0.0, 0.0, 0.0]                                                     # z may be ~0: inf/nan pixel -> rejected
        got = bow.visibility_gate(ctx, pts, nrm, R, t, see, synth.FX, synth.CX, synth.CY)
        ref = oracle.visibility_gate(pts, nrm, R, t, synth.FX, synth.CX, synth.CY, 640, 480, see)
        assert np.array_equal(got, ref)
        assert 0 < len(ref) < M or M == 1
    assert len(bow.visibility_gate(ctx, np.zeros((0, 3)), np.zeros((0, 3)), np.eye(3), np.zeros(3), [0, 0, 1], 1, 0, 0)) == 0


@pytest.mark.parametrize("n_entries,wpe,k", [(300, 40, 5), (5000, 333, 10), (20000, 1000, 10)])
def test_bow_query_matches_oracle(ctx, n_entries, wpe, k):
    db = synth.bow_database(n_entries=n_entries, words_per_entry=wpe, query_entry=n_entries // 3, seed=n_entries)
    d = bow.BowDatabase(ctx)
    d.set_db(db["start"], db["words"], db["vals"])
    oi, os_ = oracle.bow_query_l1(db["q_words"], db["q_vals"], db["start"], db["words"], db["vals"], k)
    for mode in ("scan", "inverted"):                                         # the streaming scan and the inverted-file walk (DBoW3's own structure)
        d.set_mode(mode)
        ids, sc = d.query(db["q_words"], db["q_vals"], k)
        assert np.array_equal(ids, oi), mode
        assert np.array_equal(sc.view(np.uint64), os_.view(np.uint64)), "scores must be bit-identical (same summation order)"
    assert ids[0] == db["query_entry"] and sc[0] > 0.5                        # relocalisation finds the perturbed entry
    i1, s1 = d.query(db["q_words"], db["q_vals"], 1)                          # the reference asks for top-1 (bow_db.py:39)
    assert i1[0] == oi[0] and s1[0] == os_[0]
    d.close()


def test_bow_ragged_and_empty_entries(ctx):
    rng = np.random.default_rng(4)
    lens = np.array([0, 1, 5, 0, 257, 3, 1024, 2], np.int64)
    start = np.concatenate([[0], np.cumsum(lens)])
    words = np.concatenate([np.sort(rng.choice(5000, size=n, replace=False)) for n in lens]).astype(np.uint32)
    vals = rng.random(len(words))
    qw = np.sort(rng.choice(5000, size=600, replace=False)).astype(np.uint32); qv = rng.random(600)
    d = bow.BowDatabase(ctx)
    d.set_db(start, words, vals)
    oi, os_ = oracle.bow_query_l1(qw, qv, start, words, vals, 8)
    for mode in ("scan", "inverted"):
        d.set_mode(mode)
        ids, sc = d.query(qw, qv, 8)
        assert np.array_equal(ids, oi) and np.array_equal(sc, os_), mode
        assert 0 not in ids and 3 not in ids                                   # empty entries never appear
    # a query sharing no word with anything returns nothing
    ids, sc = d.query(np.array([6000, 6001], np.uint32), np.array([0.5, 0.5]), 3)
    assert len(ids) == 0
    d.close()


def test_bow_config4_full_size_and_roofline(ctx):
    """BASELINE configs[3]: 1 query vs 100 k entries x 1000 words (1.2 GB of (word, weight) pairs)."""
    db = synth.bow_database()
    d = bow.BowDatabase(ctx)
    d.set_db(db["start"], db["words"], db["vals"])
    oi, os_ = oracle.bow_query_l1(db["q_words"], db["q_vals"], db["start"], db["words"], db["vals"], 10)
    for mode in ("scan", "inverted"):
        d.set_mode(mode)
        ids, sc = d.query(db["q_words"], db["q_vals"], 10)
        assert np.array_equal(ids, oi) and np.array_equal(sc, os_), mode
        assert ids[0] == 12345
    d.close()


def test_bow_inverted_file_follows_the_database_and_falls_back(ctx):
    """The inverted file is rebuilt after add / erase; a query that hits one bucket of 64 entries more often than its hit list holds
    (64 near-copies of the query side by side) is answered by the scan instead - same ids, same bits; subsets and top-k > 16 too."""
    rng = np.random.default_rng(8)
    qw = np.sort(rng.choice(200000, size=900, replace=False)).astype(np.uint32); qv = rng.random(900); qv /= qv.sum()
    rows = []
    for e in range(400):
        if 128 <= e < 192:                                                     # one whole bucket of near-duplicates of the query: 64 x ~850 hits
            w = np.sort(rng.choice(qw, size=850, replace=False)).astype(np.uint32)
        else:
            w = np.sort(np.concatenate([rng.choice(qw, size=rng.integers(0, 30), replace=False), rng.choice(np.arange(200000, 400000), size=200, replace=False)])).astype(np.uint32)
        v = rng.random(len(w)); rows.append((w, v / v.sum()))
    start = np.concatenate([[0], np.cumsum([len(r[0]) for r in rows])]).astype(np.int64)
    words = np.concatenate([r[0] for r in rows]); vals = np.concatenate([r[1] for r in rows])
    d = bow.BowDatabase(ctx)
    d.set_db(start, words, vals)
    oi, os_ = oracle.bow_query_l1(qw, qv, start, words, vals, 10)
    ids, sc = d.query(qw, qv, 10)                                              # inverted by default -> overflow -> scan
    assert np.array_equal(ids, oi) and np.array_equal(sc, os_) and 128 <= ids[0] < 192
    sub = [3, 7, 150, 151, 399]
    oi2, os2 = oracle.bow_query_l1_subset(qw, qv, start, words, vals, sub, 3)
    ids, sc = d.query(qw, qv, 3, subset=sub)
    assert np.array_equal(ids, oi2) and np.array_equal(sc, os2)
    ids, sc = d.query(qw, qv, 40)                                              # more results than a bucket reports candidates for
    oi3, os3 = oracle.bow_query_l1(qw, qv, start, words, vals, 40)
    assert np.array_equal(ids, oi3) and np.array_equal(sc, os3)
    # a sparse query stays on the inverted file; then the database changes under it
    q2 = qw[::9].copy(); v2 = np.full(len(q2), 1.0 / len(q2))
    oi4, os4 = oracle.bow_query_l1(q2, v2, start, words, vals, 5)
    ids, sc = d.query(q2, v2, 5)
    assert np.array_equal(ids, oi4) and np.array_equal(sc, os4)
    ids, sc = d.query(q2, v2, 4, subset=[1, 2, 130, 140, 300])
    oi5, os5 = oracle.bow_query_l1_subset(q2, v2, start, words, vals, [1, 2, 130, 140, 300], 4)
    assert np.array_equal(ids, oi5) and np.array_equal(sc, os5)
    new = d.add(q2, v2)                                                        # an exact copy of the query: must win after the rebuild
    ids, sc = d.query(q2, v2, 2)
    assert ids[0] == new and abs(sc[0] - 1.0) < 1e-12
    d.erase(new)
    ids, sc = d.query(q2, v2, 5)
    assert np.array_equal(ids, oi4) and np.array_equal(sc, os4)
    d.close()


@pytest.mark.parametrize("n_entries", [20000, 70000])
def test_bow_several_heavy_buckets_in_one_query(ctx, monkeypatch, n_entries):
    """Round 5: k_inv_finish - every workgroup sums its light buckets, then takes its buckets whose hit lists overflowed; the workgroup that
    arrives last selects (bucket minima dealt round-robin to its 16 waves, so a database of a few hundred buckets is answered too).  Three near-copies of the query in three different buckets of 64 entries (plus two in one
    bucket) make three heavy buckets in one query; the inverted file must answer (scores bit-identical to the scan and to the oracle), also
    with the stream-synchronising host path (BH_BOW_SYNC=1) instead of the polled result tag, and through the batched entry."""
    rng = np.random.default_rng(18)
    qw = np.sort(rng.choice(300000, size=1000, replace=False)).astype(np.uint32); qv = rng.random(1000); qv /= qv.sum()
    rows = []
    dup = {70: 900, 71: 700, 500: 950, 1500: 800}                             # entry -> words shared with the query (buckets 1, 1, 7, 23)
    for e in range(n_entries):
        other = np.unique(rng.integers(300000, 600000, size=120)).astype(np.uint32)
        if e in dup:
            w = np.sort(np.concatenate([rng.choice(qw, size=dup[e], replace=False), other[:60]])).astype(np.uint32)
        else:
            w = np.sort(np.concatenate([qw[rng.integers(0, 1000, size=rng.integers(0, 3))], other])).astype(np.uint32)
            w = np.unique(w)
        v = rng.random(len(w)); rows.append((w, v / v.sum()))
    start = np.concatenate([[0], np.cumsum([len(r[0]) for r in rows])]).astype(np.int64)
    words = np.concatenate([r[0] for r in rows]); vals = np.concatenate([r[1] for r in rows])
    d = bow.BowDatabase(ctx)
    d.set_db(start, words, vals)
    oi, os_ = oracle.bow_query_l1(qw, qv, start, words, vals, 10)
    assert set(oi[:4].tolist()) == set(dup)
    d.set_mode("scan")
    si, ss = d.query(qw, qv, 10)
    d.set_mode("inverted")
    for _ in range(4):                                                         # (the file is built once the database has been quiet for three queries)
        ids, sc = d.query(qw, qv, 10)
    ctx.timing(True)
    ids, sc = d.query(qw, qv, 10)
    kt = ctx.timing_read(); ctx.timing(False)
    assert "k_bow_inverted" in kt and "k_bow_scan" not in kt, "the inverted file must answer this query itself"
    assert np.array_equal(ids, oi) and np.array_equal(sc, os_) and np.array_equal(ids, si) and np.array_equal(sc, ss)
    monkeypatch.setenv("BH_BOW_SYNC", "1")
    ids2, sc2 = d.query(qw, qv, 10)
    monkeypatch.delenv("BH_BOW_SYNC")
    assert np.array_equal(ids2, oi) and np.array_equal(sc2, os_)
    q3 = qw[::3].copy(); v3 = np.full(len(q3), 1.0 / len(q3))
    got = d.query_batch([(qw, qv), (q3, v3), (qw, qv)], 7)
    o3i, o3s = oracle.bow_query_l1(q3, v3, start, words, vals, 7)
    assert np.array_equal(got[0][0], oi[:7]) and np.array_equal(got[0][1], os_[:7]) and np.array_equal(got[2][0], oi[:7])
    assert np.array_equal(got[1][0], o3i) and np.array_equal(got[1][1], o3s)
    d.close()


@pytest.mark.parametrize("k,L,n", [(10, 3, 600), (10, 4, 1000), (4, 5, 37), (7, 2, 4096)])
def test_vocabulary_transform_matches_oracle(ctx, k, L, n):
    voc = synth.vocabulary(k=k, L=L, seed=k * 10 + L)
    v = bow.Vocabulary(ctx, voc)
    d = synth.random_descriptors(n, n)
    d[n // 3] = d[1]; d[n - 1] = d[1]                                 # repeated words: the accumulation order matters
    for i in range(0, n, 7):
        d[i] = voc["node_desc"][-1 - (i % voc["n_words"])]          # exact leaf descriptors: distance-0 walks and ties on the way
    w, x = v.transform(d)
    ow, ox = oracle.voc_transform(voc, d)
    assert np.array_equal(w, ow)
    assert np.array_equal(x.view(np.uint64), ox.view(np.uint64)), "BoW weights must be bit-identical"
    assert abs(x.sum() - 1.0) < 1e-12 and np.all(np.diff(w.astype(np.int64)) > 0)
    # batched, ragged
    D = np.zeros((3, n, 32), np.uint8); D[0] = d; D[1, :5] = d[:5]; ns = [n, 5, 0]
    W, X, NW = v.transform_batch(D, ns)
    assert NW[0] == len(ow) and NW[2] == 0
    w1, x1 = oracle.voc_transform(voc, d[:5])
    assert np.array_equal(W[1, :NW[1]], w1) and np.array_equal(X[1, :NW[1]], x1)
    v.close()


def test_dbow_shim_end_to_end_relocalisation(ctx):
    """slam/bow_db.py flow: add keyframes (descriptors -> tree -> database), query a noisy revisit, erase, query again."""
    voc = synth.vocabulary(k=10, L=4, seed=5)

    class Frame:
        def __init__(self, i, des): self.id = 100 + i; self.des = des
    rng = np.random.default_rng(9)
    frames = [Frame(i, synth.random_descriptors(50 + i, 400 + 13 * i)) for i in range(12)]
    db = bow.Dbow(voc, ctx=ctx)
    for f in frames:
        db.add(f)
    db.add(frames[3])                                                  # second add of the same frame id is a no-op
    assert len(db.frame2entry) == 12
    q = frames[7].des.copy()
    flip = rng.random(q.shape) < 0.02
    q ^= (flip * (1 << rng.integers(0, 8, q.shape))).astype(np.uint8)  # a few bit errors per descriptor
    probe = Frame(99, q)
    fid, score = db.query(probe)
    # oracle: same pipeline on the CPU
    rows = [oracle.voc_transform(voc, f.des) for f in frames]
    start = np.concatenate([[0], np.cumsum([len(r[0]) for r in rows])]).astype(np.int64)
    words = np.concatenate([r[0] for r in rows]); vals = np.concatenate([r[1] for r in rows])
    qw, qv = oracle.voc_transform(voc, q)
    oi, os_ = oracle.bow_query_l1(qw, qv, start, words, vals, 1)
    assert fid == frames[int(oi[0])].id == 107 and score == os_[0]
    db.erase(107)
    fid2, score2 = db.query(probe)
    oi2, os2 = oracle.bow_query_l1(qw, qv, start, words, vals, 2)
    assert fid2 == frames[int(oi2[1])].id and score2 == os2[1]
    assert db.distance(frames[0].des[0], frames[0].des[0]) == 0 and db.distance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


def test_real_shape_vocabulary_k10_L6_from_a_dbow3_file(ctx, tmp_path):
    """The shape of the reference's /DBow3/orbvoc.dbow3 (slam/bow_db.py:17-18): k = 10, L = 6, 10^6 words, read back from a DBoW3
    binary file; quantisation on the GPU is bit-identical to the oracle's walk of the same tree."""
    from boslam_amd import dbow3_io
    voc = synth.vocabulary(k=10, L=6, seed=3)
    p = tmp_path / "orbvoc.dbow3"
    dbow3_io.save_binary(p, voc)
    loaded = dbow3_io.load_vocabulary(p)
    v = bow.Vocabulary(ctx, loaded)
    d = synth.random_descriptors(77, 1000)
    d[::9] = loaded["node_desc"][np.flatnonzero(loaded["node_word"] >= 0)[:len(d[::9])]]
    w, x = v.transform(d)
    ow, ox = oracle.voc_transform(loaded, d)
    assert len(ow) > 900 and np.array_equal(w, ow) and np.array_equal(x.view(np.uint64), ox.view(np.uint64))
    ow2, ox2 = oracle.voc_transform(voc, d)                      # and equal to the tree the file was written from
    assert np.array_equal(w, ow2) and np.array_equal(x, ox2)
    v.close()


def test_dbow_from_file_subset_query_and_compare_frames(ctx, tmp_path):
    """slam/bow_db.py end to end from a vocabulary FILE: Dbow(path), query(frame, subset_inds) (the fork's target_inds), compareFrames
    (the fork's compareBowsL1), and the live call of slam/loop_closing.py:15 with the local window excluded."""
    from boslam_amd import dbow3_io, loop_closing
    voc = synth.vocabulary(k=10, L=3, seed=8)
    p = tmp_path / "voc.yml.gz"
    dbow3_io.save_yaml(p, voc)

    class Frame:
        def __init__(self, i, des): self.id = 500 + i; self.des = des
    frames = [Frame(i, synth.random_descriptors(70 + i, 300 + 7 * i)) for i in range(10)]
    frames[6].des[:200] = frames[2].des[:200]                         # 6 revisits 2
    db = bow.Dbow(str(p), ctx=ctx)
    for f in frames:
        db.add(f)
    rows = [oracle.voc_transform(voc, f.des) for f in frames]
    start = np.concatenate([[0], np.cumsum([len(r[0]) for r in rows])]).astype(np.int64)
    words = np.concatenate([r[0] for r in rows]); vals = np.concatenate([r[1] for r in rows])
    qw, qv = rows[6]
    # unrestricted: the frame itself wins
    assert db.query(frames[6])[0] == 506
    # restricted to a subset that excludes itself: the revisited frame wins; scores equal the oracle's bit for bit
    sub = [0, 1, 2, 3, 9]
    fid, sc = db.query(frames[6], subset_inds=sub)
    oi, os_ = oracle.bow_query_l1_subset(qw, qv, start, words, vals, sub, 1)
    assert fid == 500 + int(oi[0]) == 502 and sc == os_[0]
    ids, scs = db.db.query(qw, qv, 4, subset=sub)
    oi4, os4 = oracle.bow_query_l1_subset(qw, qv, start, words, vals, sub, 4)
    assert np.array_equal(ids, oi4) and np.array_equal(scs, os4)
    # compareFrames: (common words, score) as a float32 (2, 1) array
    c = db.compareFrames(frames[6], frames[2])
    oc, osc = oracle.bow_compare_l1(start, words, vals, 6, 2)
    assert c.shape == (2, 1) and c.dtype == np.float32 and int(c[0, 0]) == oc and abs(float(c[1, 0]) - osc) < 1e-6 and oc > 50
    c2 = db.compareFrames(frames[0], frames[1])
    oc2, osc2 = oracle.bow_compare_l1(start, words, vals, 0, 1)
    assert int(c2[0, 0]) == oc2 and abs(float(c2[1, 0]) - osc2) < 1e-6
    # loop-closure candidate (slam/loop_closing.py:15) with the local window {5, 6, 7} excluded
    fid, sc = loop_closing.loop_candidate_detection(db, frames[6], exclude_frame_ids=[505, 507])
    keep = [e for e in range(10) if e not in (5, 6, 7)]
    oi, os_ = oracle.bow_query_l1_subset(qw, qv, start, words, vals, keep, 1)
    assert fid == 502 == 500 + int(oi[0]) and sc == os_[0]
    # pyDBoW3.Database-shaped access (the object slam/loop_closing.py holds)
    D = bow.Database(ctx); D.loadVocabulary(str(p), True, 2)
    for f in frames:
        D.add(f.des)
    res = D.query(frames[6].des, [], 1, -1)
    assert len(res) == 1 and res[0].Id == 6 and abs(res[0].Score - 1.0) < 1e-12
    assert [r.Id for r in D.query(frames[6].des, [], 3, 5)] == [int(i) for i in oracle.bow_query_l1_subset(qw, qv, start, words, vals, range(6), 3)[0]]


def test_bow_query_batch_equals_single_queries(ctx):
    """bh_bow_query_l1_batch: 37 queries (more than two rounds of 16 slots; near-duplicates of stored entries, unrelated word sets, an
    empty query, one that shares nothing) against the same queries one at a time and against the oracle - ids and score bits equal,
    in both modes, and single queries keep working afterwards (slot 0 of the enlarged scratch)."""
    db = synth.bow_database(n_entries=6000, words_per_entry=400, query_entry=2000, seed=9)
    d = bow.BowDatabase(ctx)
    d.set_db(db["start"], db["words"], db["vals"])
    rng = np.random.default_rng(2)
    queries = [(db["q_words"], db["q_vals"])]
    for e in rng.integers(0, 6000, 20):                                        # perturbed copies of stored entries
        a, b = db["start"][e], db["start"][e + 1]
        w = db["words"][a:b].copy(); v = db["vals"][a:b].copy()
        keep = rng.random(len(w)) > 0.2
        queries.append((w[keep], v[keep] / v[keep].sum()))
    for _ in range(14):                                                        # random word sets
        w = np.unique(rng.integers(0, 1 << 20, int(rng.integers(1, 900)))).astype(np.uint32)
        v = rng.random(len(w)); queries.append((w, v / v.sum()))
    queries.append((np.zeros(0, np.uint32), np.zeros(0)))
    queries.append((np.array([(1 << 20) - 1], np.uint32), np.array([1.0])))
    for mode in ("inverted", "scan"):
        d.set_mode(mode)
        for k in (1, 7):
            got = d.query_batch(queries, k)
            assert len(got) == len(queries)
            for (w, v), (ids, sc) in zip(queries, got):
                i1, s1 = d.query(w, v, k) if len(w) else (np.zeros(0, np.int32), np.zeros(0))
                assert np.array_equal(ids, i1) and np.array_equal(sc.view(np.uint64), s1.view(np.uint64)), mode
                if len(w):
                    oi, os_ = oracle.bow_query_l1(w, v, db["start"], db["words"], db["vals"], k)
                    assert np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint64), os_.view(np.uint64))
    assert d.query_batch([], 3) == []
    d.close()


def test_device_resident_queries_take_the_inverted_file(ctx):
    """bh_bow_query_l1 with BH_MEM_DEVICE: the query arrays and the result arrays live in HBM, nothing is synchronised by the
    call; the inverted file answers (same ids, scores bit-identical to the host paths), the predicated scan stays out of the way - and takes
    over when the inverted file cannot answer (max_results > 16, scan mode)."""
    torch = pytest.importorskip("torch")
    db = synth.bow_database(n_entries=20000, words_per_entry=600, query_entry=777, seed=21)
    d = bow.BowDatabase(ctx)
    d.set_db(db["start"], db["words"], db["vals"])
    qw = torch.from_numpy(db["q_words"].astype(np.int64)).to(torch.int32).cuda(); qv = torch.from_numpy(db["q_vals"]).cuda()      # u32 word ids < 2^31 as int32 bits
    for k, modes in ((5, ("inverted", "scan")), (16, ("inverted",)), (40, ("inverted",))):
        oi, os_ = oracle.bow_query_l1(db["q_words"], db["q_vals"], db["start"], db["words"], db["vals"], k)
        for mode in modes:
            d.set_mode(mode)
            ids = torch.full((k,), -7, dtype=torch.int32, device="cuda"); sc = torch.zeros(k, dtype=torch.float64, device="cuda"); n = torch.zeros(1, dtype=torch.int32, device="cuda")
            for rep in range(3):                                          # back to back on the stream: the slot's counters are reset by each query
                d.query_device(qw.data_ptr(), qv.data_ptr(), len(db["q_words"]), k, ids.data_ptr(), sc.data_ptr(), n.data_ptr())
            ctx.synchronize()
            m = int(n.item())
            assert m == len(oi) and np.array_equal(ids[:m].cpu().numpy(), oi), (k, mode)
            assert np.array_equal(sc[:m].cpu().numpy().view(np.uint64), os_.view(np.uint64)), (k, mode)
    # host queries after device queries on the same handle (shared slot scratch) still agree
    d.set_mode("inverted")
    i2, s2 = d.query(db["q_words"], db["q_vals"], 5)
    oi, os_ = oracle.bow_query_l1(db["q_words"], db["q_vals"], db["start"], db["words"], db["vals"], 5)
    assert np.array_equal(i2, oi) and np.array_equal(s2, os_)
    d.close()
