"""DBoW3 vocabulary files -> the flat tree arrays of bh_voc_create (and back, for tests and tools).

The reference loads its vocabulary with `self.voc.load("/DBow3/orbvoc.dbow3")` (slam/bow_db.py:17); neither that file nor the
DBoW3 sources are in the reference repository, so the two on-disk formats are restated from DBoW3's published
Vocabulary::load / toStream / save [upstream, from memory - unpinned; the reference's fork is BOpermanis/DBow3 HEAD]:

  binary (what `.dbow3` files are; Vocabulary::toStream / fromStream):
        u64 signature 88877711233 | u8 compressed | u32 nnodes (root included) |
        i32 k | i32 L | i32 scoring | i32 weighting |
        nnodes-1 node records, written by a depth-first walk with an explicit stack (so NOT in id order; the children of one
        parent are consecutive and in their stored order):
            u32 nodeId | u32 parentId | f64 weight | descriptor (DescManip::toStream: i32 cols, i32 rows, i32 cv type, bytes)
        u32 nwords | nwords x (u32 nodeId, u32 wordId)
        A node's children are its records in file order (fromStream push_backs them as it reads); leaves are the nodes named in
        the word table.  The QuickLZ-compressed variant (compressed = 1) is not supported: re-save the vocabulary uncompressed
        (Vocabulary::save(path, binary_compressed=False)) or as YAML.
  YAML (OpenCV FileStorage, optionally gzip):  vocabulary: { k, L, scoringType, weightingType,
        nodes: [ { nodeId, parentId, weight, descriptor:"dbw3 <type> <cols> b0 b1 ..." | "b0 b1 ... b31" } ],
        words: [ { wordId, nodeId } ] }

Children keep their file order: Vocabulary::transform takes the FIRST child with the smallest Hamming distance.
Only TF-IDF weighting with L1 scoring (DBoW3's defaults, the ones orbvoc uses) is what bh_voc_transform computes."""
import gzip
import re
import struct
import numpy as np

SIGNATURE = 88877711233
TF_IDF, L1_NORM = 0, 0


def _flat(parent, desc, weight, is_leaf, word_of_node=None, k=None, L=None, scoring=L1_NORM, weighting=TF_IDF, child_rank=None):
    """parent[i] (node 0 = root, parent[0] ignored), desc u8[n][32], weight f64[n], is_leaf bool[n] -> bh_voc_create arrays.
    child_rank[i] (optional): position of node i's record in the file; siblings keep that order (default: id order)."""
    n = len(parent)
    parent = np.asarray(parent, np.int64)
    if child_rank is None:
        order = np.argsort(parent[1:], kind="stable") + 1                  # children grouped by parent, id (= file) order kept
    else:
        order = np.lexsort((np.asarray(child_rank)[1:], parent[1:])) + 1   # grouped by parent, siblings by file position
    counts = np.bincount(parent[1:], minlength=n)
    child_start = np.zeros(n + 1, np.int32); child_start[1:] = np.cumsum(counts)
    child_idx = order.astype(np.int32)
    node_word = np.full(n, -1, np.int32)
    if word_of_node is None:
        node_word[np.flatnonzero(is_leaf)] = np.arange(int(np.count_nonzero(is_leaf)), dtype=np.int32)      # file order
    else:
        node_word[:] = word_of_node
    nw = np.where(node_word >= 0, np.asarray(weight, np.float64), 0.0)
    return dict(child_start=child_start, child_idx=child_idx, node_desc=np.ascontiguousarray(desc, dtype=np.uint8), node_weight=nw,
                node_word=node_word, k=k, L=L, n_words=int((node_word >= 0).sum()), scoring=scoring, weighting=weighting)


def _check(voc, path):
    if voc["weighting"] != TF_IDF or voc["scoring"] != L1_NORM:
        raise ValueError(f"{path}: weighting {voc['weighting']} / scoring {voc['scoring']}: only TF-IDF + L1 (DBoW3 defaults) are implemented")
    if voc["n_words"] > 1 << 20:
        raise ValueError(f"{path}: {voc['n_words']} words; the device query bitmap holds 2^20")
    return voc


def _desc_header(h, path, what):
    """DescManip::toStream writes (cols, rows, type); a 1 x 32 CV_8U ORB descriptor is (32, 1, 0).  (type, cols, rows) = (0, 32, 1),
    the order an older DBoW3 revision used, is accepted too - for ORB the two cannot be confused."""
    h = tuple(int(v) for v in h)
    if h in ((32, 1, 0), (0, 32, 1)):
        return 32
    raise ValueError(f"{path}: {what}: descriptor header {h}; only 1 x 32 CV_8U (ORB) descriptors are supported")


def load_binary(path):
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 13:
        raise ValueError(f"{path}: truncated DBoW3 header")
    sig, compressed, nnodes = struct.unpack_from("<QBI", raw, 0)
    if sig != SIGNATURE:
        raise ValueError(f"{path}: not a DBoW3 binary vocabulary (signature {sig})")
    if compressed:
        raise ValueError(f"{path}: QuickLZ-compressed DBoW3 stream; re-save uncompressed or as YAML")
    if nnodes == 0:
        raise ValueError(f"{path}: empty vocabulary")
    off = 13
    k, L, scoring, weighting = struct.unpack_from("<iiii", raw, off); off += 16
    rec = 4 + 4 + 8 + 12 + 32                                     # nodeId, parentId, weight, (cols, rows, type), 32 bytes
    need = off + rec * (nnodes - 1) + 4
    if len(raw) < need:
        raise ValueError(f"{path}: truncated: {nnodes} nodes need {need} bytes, the file has {len(raw)}")
    a = np.frombuffer(raw, dtype=np.uint8, count=rec * (nnodes - 1), offset=off).reshape(nnodes - 1, rec)
    off += rec * (nnodes - 1)
    nid = a[:, 0:4].copy().view("<u4")[:, 0].astype(np.int64)
    pid = a[:, 4:8].copy().view("<u4")[:, 0].astype(np.int64)
    hdr = a[:, 16:28].copy().view("<i4")
    if nnodes > 1:
        _desc_header(hdr[0], path, f"node {int(nid[0])}")
        if not np.all(hdr == hdr[0]):
            bad = int(np.flatnonzero(np.any(hdr != hdr[0], axis=1))[0])
            _desc_header(hdr[bad], path, f"node {int(nid[bad])}")
    if np.any(nid <= 0) or np.any(nid >= nnodes) or np.any(pid >= nnodes) or len(np.unique(nid)) != nnodes - 1:
        raise ValueError(f"{path}: node ids are not a permutation of 1..{nnodes - 1}")
    parent = np.zeros(nnodes, np.int64); weight = np.zeros(nnodes); desc = np.zeros((nnodes, 32), np.uint8)
    parent[nid] = pid
    weight[nid] = a[:, 8:16].copy().view("<f8")[:, 0]
    desc[nid] = a[:, 28:60]
    nwords, = struct.unpack_from("<I", raw, off); off += 4
    if len(raw) < off + 8 * nwords:
        raise ValueError(f"{path}: truncated word table ({nwords} words)")
    wt = np.frombuffer(raw, dtype="<u4", count=2 * nwords, offset=off).reshape(nwords, 2).astype(np.int64)
    if np.any(wt[:, 0] >= nnodes) or np.any(wt[:, 1] >= nwords) or len(np.unique(wt[:, 1])) != nwords:
        raise ValueError(f"{path}: word table does not map {nwords} word ids onto nodes")
    word_of_node = np.full(nnodes, -1, np.int32)
    word_of_node[wt[:, 0]] = wt[:, 1]
    if np.any(word_of_node[pid] >= 0):
        raise ValueError(f"{path}: a word node has children")
    # children in FILE order (fromStream: m_nodes[parent].children.push_back as records arrive): rank nodes by file position
    file_pos = np.zeros(nnodes, np.int64); file_pos[nid] = np.arange(1, nnodes)
    return _check(_flat(parent, desc, weight, word_of_node >= 0, word_of_node=word_of_node, k=k, L=L, scoring=scoring, weighting=weighting,
                        child_rank=file_pos), path)


_NODE = re.compile(r"nodeId:\s*(\d+)\s*,\s*parentId:\s*(\d+)\s*,\s*weight:\s*([-+.eE\d]+)\s*,\s*descriptor:\s*\"([^\"]*)\"", re.S)
_WORD = re.compile(r"wordId:\s*(\d+)\s*,\s*nodeId:\s*(\d+)")


def load_yaml(path):
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rt") as f:
        txt = f.read()
    def scalar(name):
        m = re.search(r"^\s*" + name + r":\s*(-?\d+)", txt, re.M)
        if not m:
            raise ValueError(f"{path}: no '{name}' in the vocabulary")
        return int(m.group(1))
    k, L, scoring, weighting = scalar("k"), scalar("L"), scalar("scoringType"), scalar("weightingType")
    nodes = _NODE.findall(txt)
    n = len(nodes) + 1
    parent = np.zeros(n, np.int64); weight = np.zeros(n); desc = np.zeros((n, 32), np.uint8)
    ids = np.zeros(len(nodes), np.int64)
    for j, (nid, pid, w, d) in enumerate(nodes):
        nid = int(nid); ids[j] = nid
        tok = d.split()
        if tok and tok[0] == "dbw3":
            tok = tok[3:]                                       # "dbw3 <type> <cols>" header of DBoW3's DescManip::toString
        if len(tok) != 32:
            raise ValueError(f"{path}: node {nid}: {len(tok)} descriptor bytes")
        if nid >= n:
            raise ValueError(f"{path}: node ids are not 1..N")
        parent[nid] = int(pid); weight[nid] = float(w); desc[nid] = np.array(tok, dtype=np.int64).astype(np.uint8)
    word_of_node = np.full(n, -1, np.int32)
    for wid, nid in _WORD.findall(txt):
        word_of_node[int(nid)] = int(wid)
    if not np.array_equal(ids, np.arange(1, n)):                # children must keep file order: re-rank parents by appearance
        raise ValueError(f"{path}: nodes are not listed in id order")
    return _check(_flat(parent, desc, weight, word_of_node >= 0, word_of_node=word_of_node, k=k, L=L, scoring=scoring, weighting=weighting), path)


def load_vocabulary(path):
    """Vocabulary::load: the binary format when the file starts with DBoW3's signature, else OpenCV-FileStorage YAML (.gz allowed)."""
    with open(path, "rb") as f:
        head = f.read(8)
    if len(head) == 8 and struct.unpack("<Q", head)[0] == SIGNATURE:
        return load_binary(path)
    return load_yaml(path)


def _parents(voc):
    n = len(voc["node_word"])
    parent = np.zeros(n, np.int64)
    cs, ci = np.asarray(voc["child_start"], np.int64), np.asarray(voc["child_idx"], np.int64)
    parent[ci] = np.repeat(np.arange(n), np.diff(cs))
    return parent


def _stack_order(voc):
    """The order Vocabulary::toStream writes node records in: parents = [root]; pop the LAST one, write its children in stored order,
    push those that have children.  Returns the node ids (of `voc`) in that order, root excluded."""
    cs, ci = np.asarray(voc["child_start"], np.int64), np.asarray(voc["child_idx"], np.int64)
    out = []
    stack = [0]
    while stack:
        pid = stack.pop()
        kids = ci[cs[pid]:cs[pid + 1]]
        out.append(kids)
        stack.extend(int(c) for c in kids if cs[c + 1] > cs[c])
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def _file_order(voc):
    """A node numbering 1..N in which every parent's children are consecutive and in their stored order (breadth first); what the
    YAML writer uses (its reader wants ids in file order).  Returns new_id[old_id]."""
    n = len(voc["node_word"])
    cs, ci = np.asarray(voc["child_start"], np.int64), np.asarray(voc["child_idx"], np.int64)
    level = np.array([0], np.int64); order = [level]
    while len(level):
        cnt = cs[level + 1] - cs[level]
        if cnt.sum() == 0:
            break
        idx = np.repeat(cs[level], cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        level = ci[idx]; order.append(level)
    order = np.concatenate(order)
    new_id = np.zeros(n, np.int64); new_id[np.array(order)] = np.arange(n)
    return new_id


def renumber_words_in_file_order(voc):
    """Copy of the tree whose word ids follow the leaves' breadth-first order (kept for tools; no file format needs it)."""
    new_id = _file_order(voc)
    old = np.argsort(new_id)
    out = dict(voc)
    nw = np.full(len(new_id), -1, np.int32)
    leaves = old[voc["node_word"][old] >= 0]
    nw[leaves] = np.arange(len(leaves), dtype=np.int32)
    out["node_word"] = nw
    return out


def save_binary(path, voc):
    """Uncompressed DBoW3 binary stream as Vocabulary::toStream lays it out (module docstring): records in stack order with explicit
    node and parent ids, then the (nodeId, wordId) table in word-id order."""
    n = len(voc["node_word"])
    o = _stack_order(voc)
    if len(o) != n - 1:
        raise ValueError("save_binary: the tree does not reach every node from the root")
    parent = _parents(voc)
    rec = np.zeros((n - 1, 60), np.uint8)
    rec[:, 0:4] = o.astype("<u4").view(np.uint8).reshape(-1, 4)
    rec[:, 4:8] = parent[o].astype("<u4").view(np.uint8).reshape(-1, 4)
    rec[:, 8:16] = np.asarray(voc["node_weight"], np.float64)[o].astype("<f8").view(np.uint8).reshape(-1, 8)
    rec[:, 16:28] = np.tile(np.array([32, 1, 0], "<i4").view(np.uint8), (n - 1, 1))          # cols 32, rows 1, CV_8U
    rec[:, 28:60] = voc["node_desc"][o]
    nw = np.asarray(voc["node_word"], np.int64)
    nodes_of_words = np.flatnonzero(nw >= 0)
    nodes_of_words = nodes_of_words[np.argsort(nw[nodes_of_words], kind="stable")]
    if not np.array_equal(nw[nodes_of_words], np.arange(len(nodes_of_words))):
        raise ValueError("save_binary: word ids must be 0..nwords-1")
    tab = np.stack([nodes_of_words, nw[nodes_of_words]], axis=1).astype("<u4")
    with open(path, "wb") as f:
        f.write(struct.pack("<QBI", SIGNATURE, 0, n))
        f.write(struct.pack("<iiii", int(voc.get("k") or 10), int(voc.get("L") or 0), L1_NORM, TF_IDF))
        f.write(rec.tobytes())
        f.write(struct.pack("<I", len(tab)))
        f.write(tab.tobytes())


def save_yaml(path, voc):
    new_id = _file_order(voc)
    n = len(new_id)
    old = np.argsort(new_id)
    parent = _parents(voc)
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "wt") as f:
        f.write("%YAML:1.0\n---\nvocabulary:\n")
        f.write(f"   k: {int(voc.get('k') or 10)}\n   L: {int(voc.get('L') or 0)}\n   scoringType: {L1_NORM}\n   weightingType: {TF_IDF}\n   nodes:\n")
        for j in range(1, n):
            o = old[j]
            d = "dbw3 0 32 " + " ".join(str(int(b)) for b in voc["node_desc"][o]) + " "
            f.write(f"      - {{ nodeId:{j}, parentId:{int(new_id[parent[o]])}, weight:{float(voc['node_weight'][o])!r},\n          descriptor:\"{d}\" }}\n")
        f.write("   words:\n")
        for j in range(1, n):
            o = old[j]
            if voc["node_word"][o] >= 0:
                f.write(f"      - {{ wordId:{int(voc['node_word'][o])}, nodeId:{j} }}\n")
