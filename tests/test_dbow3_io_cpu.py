"""DBoW3 vocabulary file reader (boslam_amd/dbow3_io.py): the two on-disk formats behind slam/bow_db.py:17
`self.voc.load("/DBow3/orbvoc.dbow3")`, checked by round trips through this module's own writers and by quantising descriptors
through the loaded tree with the CPU oracle."""
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
    voc = dbow3_io.renumber_words_in_file_order(synth.vocabulary(k=k, L=L, seed=k * 10 + L))
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
    voc = dbow3_io.renumber_words_in_file_order(synth.vocabulary(k=10, L=6, seed=3))
    p = tmp_path / "orbvoc.dbow3"
    dbow3_io.save_binary(p, voc)
    got = dbow3_io.load_vocabulary(p)
    assert got["k"] == 10 and got["L"] == 6 and got["n_words"] == 10 ** 6 and len(got["node_word"]) == 1111111
    _same_quantisation(voc, got, n=500)


def test_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / "c.dbow3"
    open(p, "wb").write(struct.pack("<QBI", 88877711233, 1, 5) + b"\0" * 64)
    with pytest.raises(ValueError, match="QuickLZ"):
        dbow3_io.load_vocabulary(p)
    voc = dict(synth.vocabulary(k=3, L=2, seed=1))
    with pytest.raises(ValueError, match="file order"):
        dbow3_io.save_binary(tmp_path / "x.dbow3", voc)                        # random word ids cannot be expressed in the binary format
