"""DBoW3 vocabulary file reader (boslam_amd/dbow3_io.py): the two on-disk formats behind slam/bow_db.py:17
`self.voc.load("/DBow3/orbvoc.dbow3")`.  The real file is not in the reference repository (UNPINNED): the binary layout is the one
of DBoW3's Vocabulary::toStream / fromStream as published, checked here against a byte stream assembled by hand with struct.pack
(independent of this module's writer), by round trips through the module's own writers, and by quantising descriptors through the
loaded tree with the CPU oracle."""
import struct
import numpy as np
import pytest

import oracle
from boslam_amd import dbow3_io, synth


def _same_quantisation(a, b, n=300, seed=5):
    d = synth.random_descriptors(seed, n)
    d[:40] = a["node_desc"][np.flatnonzero(a["node_word"] >= 0)[:40]]          # exact leaf hits
    wa, va = oracle.voc_transform(a, d); wb, vb = oracle.voc_transform(b, d)
    assert len(wa) > 10 and np.array_equal(wa, wb) and np.array_equal(va, vb)


@pytest.mark.parametrize("k,L", [(4, 3), (10, 2), (3, 5)])
def test_binary_round_trip(tmp_path, k, L):
    voc = synth.vocabulary(k=k, L=L, seed=k * 10 + L)                           # shuffled children, permuted word ids: the file carries both
    p = tmp_path / "voc.dbow3"
    dbow3_io.save_binary(p, voc)
    sig, comp, nn = struct.unpack("<QBI", open(p, "rb").read(13))
    assert sig == 88877711233 and comp == 0 and nn == len(voc["node_word"])
    got = dbow3_io.load_vocabulary(p)
    assert got["k"] == k and got["L"] == L and got["n_words"] == k ** L
    _same_quantisation(voc, got)


def test_yaml_round_trip_plain_and_gz(tmp_path):
    voc = synth.vocabulary(k=5, L=3, seed=9)                                   # arbitrary word ids: YAML carries them explicitly
    for name in ("voc.yml", "voc.yml.gz"):
        dbow3_io.save_yaml(tmp_path / name, voc)
        got = dbow3_io.load_vocabulary(tmp_path / name)
        assert got["k"] == 5 and got["L"] == 3 and got["n_words"] == 125
        _same_quantisation(voc, got)
    txt = open(tmp_path / "voc.yml").read()
    assert txt.startswith("%YAML:1.0") and "descriptor:\"dbw3 0 32 " in txt
    # DBoW2-style descriptor strings (no "dbw3 <type> <cols>" header) load too
    open(tmp_path / "old.yml", "w").write(txt.replace("dbw3 0 32 ", ""))
    _same_quantisation(voc, dbow3_io.load_vocabulary(tmp_path / "old.yml"))


def test_real_shape_k10_L6_binary(tmp_path):
    """The shape of orbvoc.dbow3 (slam/bow_db.py:17-18: k = 10, L = 6, 10^6 words): 1 111 111 nodes, a 63 MB file."""
    voc = synth.vocabulary(k=10, L=6, seed=3)
    p = tmp_path / "orbvoc.dbow3"
    dbow3_io.save_binary(p, voc)
    got = dbow3_io.load_vocabulary(p)
    assert got["k"] == 10 and got["L"] == 6 and got["n_words"] == 10 ** 6 and len(got["node_word"]) == 1111111
    _same_quantisation(voc, got, n=500)


def _hand_built_stream(desc_header=(32, 1, 0)):
    """A k = 2, L = 2 vocabulary written byte by byte the way Vocabulary::toStream does (DBoW3 Vocabulary.cpp): header, k / L /
    scoring / weighting, then the records of an explicit-stack depth-first walk - root's children (ids 1, 2), then, the stack
    being LIFO, node 2's children BEFORE node 1's - each record nodeId | parentId | weight | DescManip (cols, rows, type, bytes);
    then the word table as (nodeId, wordId) pairs.  Node ids and word ids are deliberately not in file order."""
    rng = np.random.default_rng(77)
    desc = {i: rng.integers(0, 256, 32, dtype=np.uint8) for i in range(1, 7)}
    weight = {1: 0.0, 2: 0.0, 3: 1.5, 4: 2.5, 5: 0.75, 6: 3.25}
    # tree: 0 -> (1, 2);  1 -> (5, 3) in THAT stored order;  2 -> (6, 4)
    records = [(1, 0), (2, 0), (6, 2), (4, 2), (5, 1), (3, 1)]
    words = [(3, 2), (4, 0), (5, 3), (6, 1)]                                     # (nodeId, wordId), table in arbitrary order
    b = struct.pack("<QBI", 88877711233, 0, 7) + struct.pack("<iiii", 2, 2, 0, 0)
    for nid, pid in records:
        b += struct.pack("<IId", nid, pid, weight[nid]) + struct.pack("<iii", *desc_header) + desc[nid].tobytes()
    b += struct.pack("<I", len(words))
    for nid, wid in words:
        b += struct.pack("<II", nid, wid)
    return b, desc, weight


@pytest.mark.parametrize("desc_header", [(32, 1, 0), (0, 32, 1)])
def test_hand_built_binary_stream_in_the_upstream_layout(tmp_path, desc_header):
    raw, desc, weight = _hand_built_stream(desc_header)
    p = tmp_path / "hand.dbow3"
    open(p, "wb").write(raw)
    v = dbow3_io.load_vocabulary(p)
    assert v["k"] == 2 and v["L"] == 2 and v["n_words"] == 4 and len(v["node_word"]) == 7
    cs, ci = v["child_start"], v["child_idx"]
    kids = lambda i: list(ci[cs[i]:cs[i + 1]])
    assert kids(0) == [1, 2] and kids(1) == [5, 3] and kids(2) == [6, 4] and all(kids(i) == [] for i in (3, 4, 5, 6))
    assert list(v["node_word"]) == [-1, -1, -1, 2, 0, 3, 1]
    for i in range(1, 7):
        assert np.array_equal(v["node_desc"][i], desc[i])
    assert list(v["node_weight"]) == [0.0, 0.0, 0.0, 1.5, 2.5, 0.75, 3.25]
    # quantisation through the loaded tree: a descriptor equal to node 5's, but one bit towards node 2's side, still lands where the
    # first-minimum walk puts it (python walk over the hand-made tree)
    def walk(d):
        node = 0
        while kids(node):
            dist = [int(np.unpackbits(np.bitwise_xor(d, desc[c])).sum()) for c in kids(node)]
            node = kids(node)[int(np.argmin(dist))]
        return int(v["node_word"][node])
    q = np.stack([desc[5], desc[3], desc[6], desc[4], desc[1], desc[2]])
    words, vals = oracle.voc_transform(v, q)
    assert sorted(set(walk(d) for d in q)) == list(words)
    # the module's writer emits the same bytes for the same tree (stack order, explicit ids, word table in word-id order)
    if desc_header == (32, 1, 0):
        p2 = tmp_path / "again.dbow3"
        dbow3_io.save_binary(p2, v)
        raw2 = open(p2, "rb").read()
        n_rec = 13 + 16 + 6 * 60
        assert raw2[:n_rec] == raw[:n_rec]                                         # header + records byte for byte
        tab = np.frombuffer(raw2[n_rec + 4:], "<u4").reshape(-1, 2)
        assert [tuple(r) for r in tab] == [(4, 0), (6, 1), (3, 2), (5, 3)]


def test_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / "c.dbow3"
    open(p, "wb").write(struct.pack("<QBI", 88877711233, 1, 5) + b"\0" * 64)
    with pytest.raises(ValueError, match="QuickLZ"):
        dbow3_io.load_vocabulary(p)
    raw, _, _ = _hand_built_stream()
    open(p, "wb").write(raw[:200])
    with pytest.raises(ValueError, match="truncated"):
        dbow3_io.load_vocabulary(p)
    open(p, "wb").write(raw[:-8])
    with pytest.raises(ValueError, match="truncated word table"):
        dbow3_io.load_vocabulary(p)
    bad = bytearray(raw); bad[13 + 16:13 + 20] = struct.pack("<I", 9)              # a node id outside 1..6
    open(p, "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="permutation"):
        dbow3_io.load_vocabulary(p)
    bad, _, _ = _hand_built_stream((64, 1, 5))                                      # not a 32-byte CV_8U descriptor
    open(p, "wb").write(bad)
    with pytest.raises(ValueError, match="descriptor header"):
        dbow3_io.load_vocabulary(p)
