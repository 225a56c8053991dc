"""TF checkpoint-bundle reader/writer (reference network.py:47, :122 restore path)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, MODEL_PREFIX
from roomnet_amd import tf_bundle
from roomnet_amd.graph import build_graph


def test_crc32c_known_answers():
    # RFC 3720 / LevelDB crc32c test vectors
    assert tf_bundle.crc32c(b"") == 0
    assert tf_bundle.crc32c(b"123456789") == 0xE3069283
    assert tf_bundle.crc32c(bytes(32)) == 0x8A9136AA
    assert tf_bundle.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tf_bundle.crc32c(bytes(range(32))) == 0x46DD794E
    for v in (0, 1, 0xDEADBEEF, 0xFFFFFFFF):
        assert tf_bundle.unmask_crc(tf_bundle.mask_crc(v)) == v


def test_all_79_tensor_crcs_verify():
    r = tf_bundle.BundleReader(MODEL_PREFIX)
    assert r.header["num_shards"] == 1
    assert len(r.entries) == 79
    assert all(r.verify(k) for k in r.keys())
    assert sum(e.num_elements for e in r.entries.values()) == 178062


def test_index_matches_reference_fixture():
    ref = json.load(open(os.path.join(GOLDEN, "bundle_index.json")))
    r = tf_bundle.BundleReader(MODEL_PREFIX)
    assert len(ref["entries"]) == 79
    for e in ref["entries"]:
        got = r.entries[e["name"]]
        assert list(got.shape) == e["shape"]
        assert (got.offset, got.size, got.crc32c) == (e["offset"], e["size"], e["crc32c_masked"])
    # data shard is the tensors back to back in lexicographic key order
    off = 0
    for k in r.keys():
        assert r.entries[k].offset == off
        off += r.entries[k].size
    assert off == 712248 == os.path.getsize(MODEL_PREFIX + ".data-00000-of-00001")


def test_checkpoint_matches_graph_variables():
    r = tf_bundle.BundleReader(MODEL_PREFIX)
    g = build_graph(6, 224)
    assert {k: tuple(e.shape) for k, e in r.entries.items()} == g.variable_shapes()


def test_write_read_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"a/kernel": rng.standard_normal((3, 3, 4, 5)).astype(np.float32),
               "a/bias": rng.standard_normal((5,)).astype(np.float32),
               "z_last": np.arange(7, dtype=np.float32),
               "scalar_like": np.ones((1,), np.float32)}
    prefix = str(tmp_path / "ckpt" / "model--12")
    tf_bundle.write_bundle(prefix, tensors)
    r = tf_bundle.BundleReader(prefix)
    assert sorted(r.keys()) == sorted(tensors)
    for k, v in tensors.items():
        np.testing.assert_array_equal(r.get(k), v)


def test_rewrite_of_reference_checkpoint_is_byte_identical_data(tmp_path, weights):
    prefix = str(tmp_path / "roomnet")
    tf_bundle.write_bundle(prefix, weights)
    assert open(prefix + ".data-00000-of-00001", "rb").read() == \
        open(MODEL_PREFIX + ".data-00000-of-00001", "rb").read()
    r = tf_bundle.BundleReader(prefix)
    ref = tf_bundle.BundleReader(MODEL_PREFIX)
    for k in ref.keys():
        assert r.entries[k].crc32c == ref.entries[k].crc32c
        assert r.entries[k].offset == ref.entries[k].offset


def test_corruption_is_detected(tmp_path, weights):
    prefix = str(tmp_path / "bad")
    tf_bundle.write_bundle(prefix, {"x": np.arange(100, dtype=np.float32)})
    p = prefix + ".data-00000-of-00001"
    raw = bytearray(open(p, "rb").read())
    raw[17] ^= 0x40
    open(p, "wb").write(bytes(raw))
    r = tf_bundle.BundleReader(prefix)
    assert not r.verify("x")
    with pytest.raises(tf_bundle.BundleError):
        r.get("x")
    # index corruption
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[3] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(tf_bundle.BundleError):
        tf_bundle.BundleReader(prefix)


def test_missing_and_bad_files(tmp_path):
    with pytest.raises(tf_bundle.BundleError):
        tf_bundle.BundleReader(str(tmp_path / "nope"))
    p = tmp_path / "junk.index"
    p.write_bytes(b"not a table" * 10)
    with pytest.raises(tf_bundle.BundleError):
        tf_bundle.BundleReader(str(tmp_path / "junk"))
