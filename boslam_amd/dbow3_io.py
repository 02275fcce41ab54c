"""DBoW3 vocabulary files -> the flat tree arrays of bh_voc_create (and back, for tests and tools).

The reference loads its vocabulary with `self.voc.load("/DBow3/orbvoc.dbow3")` (slam/bow_db.py:17); neither that file nor the
DBoW3 sources are in the reference repository, so the two on-disk formats are restated from DBoW3's published
Vocabulary::load / toStream / save [upstream, from memory - unpinned; the reference's fork is BOpermanis/DBow3 HEAD]:

  binary (what `.dbow3` files are):  u64 signature 88877711233 | u8 compressed | u32 nnodes |
        i32 k | i32 L | i32 scoring | i32 weighting |
        for node 1 .. nnodes-1:  u32 parent | descriptor (i32 cv type, i32 cols, i32 rows, cols*rows bytes) | f64 weight | u8 isLeaf
        Leaves get word ids in file order.  The QuickLZ-compressed variant (compressed = 1) is not supported: re-save the
        vocabulary uncompressed (DBoW3's Vocabulary::save(path, binary_compressed=False)) or as YAML.
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


def _flat(parent, desc, weight, is_leaf, word_of_node=None, k=None, L=None, scoring=L1_NORM, weighting=TF_IDF):
    """parent[i] (node 0 = root, parent[0] ignored), desc u8[n][32], weight f64[n], is_leaf bool[n] -> bh_voc_create arrays."""
    n = len(parent)
    parent = np.asarray(parent, np.int64)
    order = np.argsort(parent[1:], kind="stable") + 1                      # children grouped by parent, file order kept
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


def load_binary(path):
    with open(path, "rb") as f:
        raw = f.read()
    sig, compressed, nnodes = struct.unpack_from("<QBI", raw, 0)
    if sig != SIGNATURE:
        raise ValueError(f"{path}: not a DBoW3 binary vocabulary (signature {sig})")
    if compressed:
        raise ValueError(f"{path}: QuickLZ-compressed DBoW3 stream; re-save uncompressed or as YAML")
    off = 13
    k, L, scoring, weighting = struct.unpack_from("<iiii", raw, off); off += 16
    parent = np.zeros(nnodes, np.int64); weight = np.zeros(nnodes); leaf = np.zeros(nnodes, bool)
    desc = np.zeros((nnodes, 32), np.uint8)
    # fixed-size records when every descriptor is 1 x 32 CV_8U (ORB): parse them in one shot
    rec = 4 + 12 + 32 + 8 + 1
    if len(raw) - off == rec * (nnodes - 1):
        a = np.frombuffer(raw, dtype=np.uint8, count=rec * (nnodes - 1), offset=off).reshape(nnodes - 1, rec)
        hdr = a[:, 4:16].copy().view("<i4")
        if np.all(hdr[:, 1] * hdr[:, 2] == 32):
            parent[1:] = a[:, 0:4].copy().view("<u4")[:, 0]
            desc[1:] = a[:, 16:48]
            weight[1:] = a[:, 48:56].copy().view("<f8")[:, 0]
            leaf[1:] = a[:, 56] != 0
            return _check(_flat(parent, desc, weight, leaf, k=k, L=L, scoring=scoring, weighting=weighting), path)
    for i in range(1, nnodes):
        parent[i], = struct.unpack_from("<I", raw, off); off += 4
        _type, cols, rows = struct.unpack_from("<iii", raw, off); off += 12
        nb = cols * rows
        if nb != 32:
            raise ValueError(f"{path}: node {i} has a {rows}x{cols} descriptor; ORB descriptors are 32 bytes")
        desc[i] = np.frombuffer(raw, np.uint8, 32, off); off += 32
        weight[i], = struct.unpack_from("<d", raw, off); off += 8
        leaf[i] = raw[off] != 0; off += 1
    return _check(_flat(parent, desc, weight, leaf, k=k, L=L, scoring=scoring, weighting=weighting), path)


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


def _file_order(voc):
    """Node numbering a DBoW3 file needs: ids 1..N in an order in which every parent's children appear in their stored order and
    leaves are numbered by word id (the binary format derives word ids from leaf order).  Returns new_id[old_id]."""
    n = len(voc["node_word"])
    cs, ci = voc["child_start"], voc["child_idx"]
    # BFS, level by level: keeps each parent's children contiguous and in their stored order
    cs, ci = np.asarray(cs, np.int64), np.asarray(ci, np.int64)
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
    """Copy of the tree whose word ids follow the leaves' file order (what the binary format implies)."""
    new_id = _file_order(voc)
    old = np.argsort(new_id)
    out = dict(voc)
    nw = np.full(len(new_id), -1, np.int32)
    leaves = old[voc["node_word"][old] >= 0]
    nw[leaves] = np.arange(len(leaves), dtype=np.int32)
    out["node_word"] = nw
    return out


def save_binary(path, voc):
    """Uncompressed DBoW3 binary stream.  Word ids are implied by leaf order in the file, so the tree must number its leaves in
    that order (true for trees this module loaded; renumber_words_in_file_order makes any tree so)."""
    new_id = _file_order(voc)
    n = len(new_id)
    old = np.argsort(new_id)
    parent = _parents(voc)
    leaf_words = voc["node_word"][old][voc["node_word"][old] >= 0]
    if not np.array_equal(leaf_words, np.arange(len(leaf_words))):
        raise ValueError("save_binary: word ids must follow the leaves' file order")
    o = old[1:]
    rec = np.zeros((n - 1, 57), np.uint8)
    rec[:, 0:4] = new_id[parent[o]].astype("<u4").view(np.uint8).reshape(-1, 4)
    rec[:, 4:16] = np.tile(np.array([0, 32, 1], "<i4").view(np.uint8), (n - 1, 1))          # CV_8U, cols 32, rows 1
    rec[:, 16:48] = voc["node_desc"][o]
    rec[:, 48:56] = np.asarray(voc["node_weight"], np.float64)[o].astype("<f8").view(np.uint8).reshape(-1, 8)
    rec[:, 56] = voc["node_word"][o] >= 0
    with open(path, "wb") as f:
        f.write(struct.pack("<QBI", SIGNATURE, 0, n))
        f.write(struct.pack("<iiii", int(voc.get("k") or 10), int(voc.get("L") or 0), L1_NORM, TF_IDF))
        f.write(rec.tobytes())


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
