"""Flax checkpoint wire format: savit_amd/flax_ckpt.py against the published flax.serialization layout.
CPU-only: the format is host code.  The engine round trip is in test_model_gpu.py."""
import os
import struct
import sys

import msgpack
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import savit_amd  # noqa: E402,F401
from savit_amd import flax_ckpt as fc  # noqa: E402


def test_ndarray_ext_known_bytes():
    """A float32 [2,3] array is ExtType 1 whose payload is msgpack((shape, dtype name, raw bytes)) - built by hand here."""
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    payload = msgpack.packb(([2, 3], "float32", a.tobytes()), use_bin_type=True)
    expected = msgpack.packb({"w": msgpack.ExtType(1, payload)}, use_bin_type=True)
    assert fc.msgpack_serialize({"w": a}) == expected
    # and the raw layout: fixmap(1) 'w' ext8/ext16 type 1 ...
    assert expected[0] == 0x81 and expected[1:3] == b"\xa1w"
    assert expected[3] in (0xc7, 0xc8) and expected[5 if expected[3] == 0xc7 else 6] == 1
    back = fc.msgpack_restore(expected)
    assert back["w"].dtype == np.float32 and np.array_equal(back["w"], a)


def test_numpy_scalar_and_tuple_state():
    tree = {"step": np.int32(5), "opt_state": ({}, {"count": np.asarray(5, np.int32), "mu": {"a": np.ones(3, np.float32)}}, {})}
    back = fc.msgpack_restore(fc.msgpack_serialize(tree))
    assert int(back["step"]) == 5
    assert set(back["opt_state"].keys()) == {"0", "1", "2"}  # tuples become dicts keyed '0', '1', ... (to_state_dict)
    assert int(back["opt_state"]["1"]["count"]) == 5
    np.testing.assert_array_equal(back["opt_state"]["1"]["mu"]["a"], np.ones(3, np.float32))


def test_chunked_array_layout_and_roundtrip():
    a = np.random.default_rng(0).standard_normal((37, 11)).astype(np.float32)
    blob = fc.msgpack_serialize({"big": a}, max_chunk_bytes=256)
    raw = msgpack.unpackb(blob, raw=False, strict_map_key=False)
    assert raw["big"]["__msgpack_chunked_array__"] is True
    assert raw["big"]["shape"] == {"0": 37, "1": 11}
    assert len(raw["big"]["chunks"]) == -(-a.nbytes // 256)
    np.testing.assert_array_equal(fc.msgpack_restore(blob)["big"], a)


def test_bfloat16_payload_widens_to_float32():
    vals = np.array([1.0, -2.5, 3.140625], dtype=np.float32)
    bits = (vals.view(np.uint32) >> 16).astype(np.uint16)  # exact in bf16
    payload = msgpack.packb(([3], "bfloat16", bits.tobytes()), use_bin_type=True)
    blob = msgpack.packb({"x": msgpack.ExtType(1, payload)}, use_bin_type=True)
    np.testing.assert_array_equal(fc.msgpack_restore(blob)["x"], vals)
    assert fc.msgpack_restore(blob, bf16_as_f32=False)["x"].dtype == np.uint16


def test_latest_checkpoint_picks_highest_step(tmp_path):
    for s in (3, 20, 100):
        (tmp_path / f"checkpoint_{s}").write_bytes(fc.msgpack_serialize({"step": np.int32(s)}))
    (tmp_path / "checkpoint_7.tmp").write_bytes(b"")
    (tmp_path / "checkpoint_200.pt").write_bytes(b"")
    assert os.path.basename(fc.latest_checkpoint(str(tmp_path))) == "checkpoint_100"
    assert int(fc.read_train_state(fc.latest_checkpoint(str(tmp_path)))["step"]) == 100


def test_module_tree_accepts_variables_or_params_level():
    inner = {"Dense_0": {"kernel": np.zeros((2, 2), np.float32)}}
    assert fc._module_tree({"params": inner}) == {"params": inner}
    assert fc._module_tree({"params": {"params": inner}}) == {"params": inner}
    assert fc._module_tree(inner) == {"params": inner}


def test_unknown_type_is_rejected():
    with pytest.raises(TypeError):
        fc.msgpack_serialize({"bad": object()})
