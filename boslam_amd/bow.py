"""Local-map visibility gate and DBoW3-style L1 database query over the C ABI, plus the keyframe-sharded multi-GPU
query (per-rank top-k + all-gather + merge).

Mirrors
  Tracker._track_local_map's per-point gate           slam/tracking.py:97-111
  Dbow.query -> pyDBoW3 Database.query(des, [], 1, -1) slam/bow_db.py:37-42, docker/pydbow3/src/dbow3.cpp:147-152
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import BH_MEM_HOST

COS60 = float(np.cos(np.pi / 3))            # slam/tracking.py:19


def visibility_gate(ctx, pts, normals, R, t, see_vector, f, cx, cy, width=640, height=480, cos_thr=COS60):
    """One frame: returns the indices (input order) of the map points that project inside the image and face the camera."""
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(1, -1, 3)
    normals = np.ascontiguousarray(normals, dtype=np.float64).reshape(1, -1, 3)
    M = pts.shape[1]
    if M == 0:
        return np.zeros(0, np.int32)
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(1, 9); t = np.ascontiguousarray(t, dtype=np.float64).reshape(1, 3)
    see = np.ascontiguousarray(see_vector, dtype=np.float64).reshape(1, 3)
    idx = np.zeros((1, M), np.int32); n = np.zeros(1, np.int32); Ms = np.array([M], np.int32)
    ctx.check(_lib.lib().bh_visibility_gate(ctx.h, _lib.ptr(pts), _lib.ptr(normals), _lib.ptr(Ms), 1, M, _lib.ptr(R), _lib.ptr(t),
                                            _lib.ptr(see), float(f), float(cx), float(cy), int(width), int(height),
                                            float(cos_thr), _lib.ptr(idx), _lib.ptr(n), BH_MEM_HOST))
    return idx[0, :n[0]].copy()


class BowDatabase:
    """Device-resident BoW database (CSR over entries) with L1 top-k query."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(_lib.lib().bh_bow_create(ctx.h, C.byref(h)))
        self.h = h
        self.n_entries = 0

    def set_db(self, start, words, vals):
        start = np.ascontiguousarray(start, dtype=np.int64); words = np.ascontiguousarray(words, dtype=np.uint32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        self.n_entries = len(start) - 1
        self.ctx.check(_lib.lib().bh_bow_set_db(self.h, self.n_entries, _lib.ptr(start), _lib.ptr(words), _lib.ptr(vals), BH_MEM_HOST))

    def set_db_device(self, start, words, vals):
        """Same, from device tensors (torch used as the allocator only): start i64[E+1], words i32/u32[nnz], vals f64[nnz]."""
        self.n_entries = int(start.numel()) - 1
        self.ctx.check(_lib.lib().bh_bow_set_db(self.h, self.n_entries, _lib.ptr(start), _lib.ptr(words), _lib.ptr(vals), _lib.BH_MEM_DEVICE))

    def add(self, words, vals):
        """Database.add at the BoW-vector level -> entry id."""
        words = np.ascontiguousarray(words, dtype=np.uint32); vals = np.ascontiguousarray(vals, dtype=np.float64)
        eid = C.c_int(-1)
        self.ctx.check(_lib.lib().bh_bow_add_entry(self.h, _lib.ptr(words), _lib.ptr(vals), len(words), C.byref(eid)))
        self.n_entries = eid.value + 1
        return eid.value

    def erase(self, entry_id):
        self.ctx.check(_lib.lib().bh_bow_erase_entry(self.h, int(entry_id)))

    def query(self, q_words, q_vals, max_results=1, subset=None):
        """Top `max_results` (entry id, score).  `subset` = the fork's target_inds: only these entries may be returned (empty = all)."""
        q_words = np.ascontiguousarray(q_words, dtype=np.uint32); q_vals = np.ascontiguousarray(q_vals, dtype=np.float64)
        ids = np.zeros(max_results, np.int32); sc = np.zeros(max_results, np.float64); n = C.c_int(0)
        if subset is not None and len(subset):
            sub = np.ascontiguousarray(subset, dtype=np.int32)
            self.ctx.check(_lib.lib().bh_bow_query_l1_subset(self.h, _lib.ptr(q_words), _lib.ptr(q_vals), len(q_words), _lib.ptr(sub), len(sub),
                                                             int(max_results), _lib.ptr(ids), _lib.ptr(sc), C.byref(n)))
        else:
            self.ctx.check(_lib.lib().bh_bow_query_l1(self.h, _lib.ptr(q_words), _lib.ptr(q_vals), len(q_words), int(max_results),
                                                      _lib.ptr(ids), _lib.ptr(sc), C.byref(n), BH_MEM_HOST))
        return ids[:n.value].copy(), sc[:n.value].copy()

    def query_device(self, q_words, q_vals, nq, max_results, ids, scores, n_results):
        """A device-resident query (bh_bow_query_l1, BH_MEM_DEVICE): every argument but nq / max_results is a device address (int) - words u32
        strictly ascending, weights f64; ids int32[max_results], scores f64[max_results], n_results int32[1].  Enqueued on the context's
        stream, no synchronisation; the inverted file answers when it can, the scan otherwise (set_mode as for host queries)."""
        v = C.c_void_p
        self.ctx.check(_lib.lib().bh_bow_query_l1(self.h, v(q_words), v(q_vals), int(nq), int(max_results), v(ids), v(scores), v(n_results), _lib.BH_MEM_DEVICE))

    def query_batch(self, queries, max_results=1):
        """Many queries in one call (bh_bow_query_l1_batch): queries = sequence of (words, vals).  Returns a list of (ids, scores), each
        what query() returns for that query; with the inverted file up to 16 queries share every launch."""
        Q = len(queries)
        S = max([len(w) for w, _ in queries] + [1])
        qw = np.zeros((Q, S), np.uint32); qv = np.zeros((Q, S), np.float64); nq = np.zeros(Q, np.int32)
        for i, (w, v) in enumerate(queries):
            nq[i] = len(w); qw[i, :len(w)] = w; qv[i, :len(w)] = v
        ids = np.zeros((Q, max_results), np.int32); sc = np.zeros((Q, max_results), np.float64); n = np.zeros(Q, np.int32)
        if Q:
            self.ctx.check(_lib.lib().bh_bow_query_l1_batch(self.h, _lib.ptr(qw), _lib.ptr(qv), _lib.ptr(nq), Q, S, int(max_results), _lib.ptr(ids), _lib.ptr(sc), _lib.ptr(n)))
        return [(ids[i, :n[i]].copy(), sc[i, :n[i]].copy()) for i in range(Q)]

    def set_mode(self, mode):
        """'inverted' (default: inverted-file query, scan as the fallback) or 'scan' (always stream the database)."""
        self.ctx.check(_lib.lib().bh_bow_set_mode(self.h, {"inverted": 0, "scan": 1}[mode]))

    def compare(self, i, j):
        """compareBowsL1 of two stored entries -> (number of common words, L1 score)."""
        cnt = C.c_int(0); score = C.c_double(0.0)
        self.ctx.check(_lib.lib().bh_bow_compare_l1(self.h, int(i), int(j), C.byref(cnt), C.byref(score)))
        return cnt.value, score.value

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().bh_bow_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk(id_lists, score_lists, k):
    """Merge per-shard (global id, score) lists: best score first, ties by ascending id."""
    ids = np.concatenate([np.asarray(i, dtype=np.int64) for i in id_lists]) if id_lists else np.zeros(0, np.int64)
    sc = np.concatenate([np.asarray(s, dtype=np.float64) for s in score_lists]) if score_lists else np.zeros(0)
    order = np.lexsort((ids, -sc))
    return ids[order][:k], sc[order][:k]


def sharded_query(local_query, entry_base, k, dist=None, device="cpu"):
    """Keyframe-sharded query: `local_query()` -> (local ids, scores) on this rank's shard whose first entry has global
    id `entry_base`; every rank returns the global top-k.  The only collective is an all-gather of k (id, score) pairs
    per rank (8*k*16 B on an 8-GPU node)."""
    ids, sc = local_query()
    ids = np.asarray(ids, dtype=np.int64) + int(entry_base)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return merge_topk([ids], [sc], k)
    import torch
    pad_i = torch.full((k,), -1, dtype=torch.int64, device=device); pad_s = torch.full((k,), -1.0, dtype=torch.float64, device=device)
    pad_i[:len(ids)] = torch.from_numpy(ids).to(device); pad_s[:len(sc)] = torch.from_numpy(np.asarray(sc, dtype=np.float64)).to(device)
    world = dist.get_world_size()
    gi = [torch.empty_like(pad_i) for _ in range(world)]; gs = [torch.empty_like(pad_s) for _ in range(world)]
    dist.all_gather(gi, pad_i); dist.all_gather(gs, pad_s)
    li, ls = [], []
    for a, b in zip(gi, gs):
        a = a.cpu().numpy(); b = b.cpu().numpy()
        li.append(a[a >= 0]); ls.append(b[a >= 0])
    return merge_topk(li, ls, k)


class Vocabulary:
    """Device-resident vocabulary tree (DBoW3 Vocabulary stand-in): descriptors -> BoW vector."""

    def __init__(self, ctx, voc):
        self.ctx = ctx
        cs = np.ascontiguousarray(voc["child_start"], dtype=np.int32); ci = np.ascontiguousarray(voc["child_idx"], dtype=np.int32)
        nd = np.ascontiguousarray(voc["node_desc"], dtype=np.uint8); nw = np.ascontiguousarray(voc["node_weight"], dtype=np.float64)
        ni = np.ascontiguousarray(voc["node_word"], dtype=np.int32)
        h = C.c_void_p()
        ctx.check(_lib.lib().bh_voc_create(ctx.h, len(cs) - 1, _lib.ptr(cs), _lib.ptr(ci), _lib.ptr(nd), _lib.ptr(nw), _lib.ptr(ni), C.byref(h)))
        self.h = h

    def transform_batch(self, desc, n):
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        B, S = desc.shape[:2]
        n = np.ascontiguousarray(n, dtype=np.int32)
        words = np.zeros((B, S), np.uint32); vals = np.zeros((B, S), np.float64); nw = np.zeros(B, np.int32)
        self.ctx.check(_lib.lib().bh_voc_transform(self.h, _lib.ptr(desc), _lib.ptr(n), B, S, _lib.ptr(words), _lib.ptr(vals), _lib.ptr(nw), BH_MEM_HOST))
        return words, vals, nw

    def transform(self, desc):
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(1, -1, 32)
        if desc.shape[1] == 0:
            return np.zeros(0, np.uint32), np.zeros(0)
        w, v, n = self.transform_batch(desc, [desc.shape[1]])
        return w[0, :n[0]].copy(), v[0, :n[0]].copy()

    @staticmethod
    def distance(f1, f2):
        """Vocabulary.distance -> Hamming distance of two descriptors (slam/bow_db.py:44-45, slam/nodes.py medoid)."""
        return int(np.unpackbits(np.bitwise_xor(np.asarray(f1, np.uint8).ravel(), np.asarray(f2, np.uint8).ravel())).sum())

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().bh_voc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dbow:
    """Mirror of the reference's slam/bow_db.py Dbow (add / erase / query / distance / compareFrames) over the HIP vocabulary +
    database.  `voc` is a vocabulary file (the reference's default is "/DBow3/orbvoc.dbow3", slam/bow_db.py:17; DBoW3 binary or
    YAML, see dbow3_io) or an already flat tree dict.  The reference's `setVocabulary(voc, True, 2)` also switches on DBoW3's direct
    index, which none of the reference's calls read; it is not kept."""

    DEFAULT_VOCABULARY = "/DBow3/orbvoc.dbow3"

    def __init__(self, voc=DEFAULT_VOCABULARY, ctx=None):
        self.ctx = ctx or _lib.default_context()
        if not isinstance(voc, dict):
            from . import dbow3_io
            voc = dbow3_io.load_vocabulary(voc)
        self.voc = Vocabulary(self.ctx, voc)
        self.db = BowDatabase(self.ctx)
        self.entry2frame = {}
        self.frame2entry = {}

    def add(self, frame):
        if frame.id not in self.frame2entry:
            w, v = self.voc.transform(frame.des)
            i = self.db.add(w, v)
            self.entry2frame[i] = frame.id
            self.frame2entry[frame.id] = i

    def erase(self, i):
        j = self.frame2entry[i]
        self.db.erase(j)
        del self.frame2entry[i]
        del self.entry2frame[j]

    def query(self, frame, subset_inds=()):
        """slam/bow_db.py:37-42: best (frame id, score) among the entries in subset_inds (empty: all), or (None, None)."""
        w, v = self.voc.transform(frame.des)
        ids, sc = self.db.query(w, v, 1, subset=list(subset_inds))
        if len(ids) > 0:
            return self.entry2frame[int(ids[0])], float(sc[0])
        return None, None

    def distance(self, f1, f2):
        return self.voc.distance(f1, f2)

    def compareFrames(self, frame1, frame2):
        """slam/bow_db.py:47-57: float32 (2, 1) array [number of common words, L1 score], as compare_bows returns its cv::Mat."""
        if frame1.id not in self.frame2entry:
            self.add(frame1)
        if frame2.id not in self.frame2entry:
            self.add(frame2)
        cnt, score = self.db.compare(self.frame2entry[frame1.id], self.frame2entry[frame2.id])
        return np.array([[cnt], [score]], dtype=np.float32)


class Result:
    """DBoW3::Result as pyDBoW3 exposes it (slam/bow_db.py:41 reads .Id and .Score)."""
    __slots__ = ("Id", "Score")

    def __init__(self, Id, Score):
        self.Id = int(Id); self.Score = float(Score)

    def __repr__(self):
        return f"Result(Id={self.Id}, Score={self.Score})"


class Database:
    """Call-compatible stand-in for pyDBoW3.Database (docker/pydbow3/src/dbow3.cpp:121-193) for the calls the reference makes:
    setVocabulary (slam/bow_db.py:18), add (:26), erase (:33), query(des, target_inds, max_results, max_id) (:39 and
    slam/loop_closing.py:15), compare_bows (:57)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or _lib.default_context()
        self.voc = None
        self.db = BowDatabase(self.ctx)

    def setVocabulary(self, vocabulary, use_di=False, di_levels=0):
        self.voc = vocabulary if isinstance(vocabulary, Vocabulary) else Vocabulary(self.ctx, vocabulary)

    def loadVocabulary(self, filename, use_di=False, di_levels=0):
        from . import dbow3_io
        self.setVocabulary(dbow3_io.load_vocabulary(filename), use_di, di_levels)

    def add(self, features):
        w, v = self.voc.transform(features)
        return self.db.add(w, v)

    def erase(self, i):
        self.db.erase(i)

    def query(self, features, target_inds=(), max_results=1, max_id=-1):
        w, v = self.voc.transform(features)
        subset = list(target_inds)
        if max_id >= 0:                                            # DBoW3: only entries with id <= max_id
            subset = [i for i in (subset or range(self.db.n_entries)) if i <= max_id]
            if not subset:
                return []
        ids, sc = self.db.query(w, v, max_results, subset=subset)
        return [Result(i, s) for i, s in zip(ids, sc)]

    def compare_bows(self, i, j):
        cnt, score = self.db.compare(i, j)
        return np.array([[cnt], [score]], dtype=np.float32)
