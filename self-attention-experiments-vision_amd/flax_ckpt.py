"""Flax checkpoint format: read and write the msgpack files `flax.training.checkpoints` produces.

The reference saves its TrainState with `checkpoints.save_checkpoint(dir, train_state, step, keep=3)`
(/root/reference/train.py:123-127): a file `checkpoint_<step>` holding `flax.serialization.to_bytes(train_state)`, i.e. the
msgpack encoding of the state dict

    {'step': i32[], 'params': {'params': {<module tree>}},                      # train.py:29-37: TrainState.params = model.init(...)
     'opt_state': {'0': {},                                                     # clip_by_global_norm        (train.py:25-27)
                   '1': {'count': i32[], 'mu': {'params': …}, 'nu': {'params': …}},  # scale_by_adam
                   '2': {}, '3': {'count': i32[]}}}                             # additive_weight_decay, scale_by_schedule

Flax is not installed here, so the wire format is restated from its published definition (flax/serialization.py):
  * every ndarray / numpy scalar is a msgpack ExtType: code 1 (ndarray) or 3 (numpy scalar) whose payload is itself
    msgpack: (shape tuple, dtype name, raw little-endian bytes); code 2 is a native Python complex (packed (real, imag));
  * an array above 2**30 bytes is written as a dict {'__msgpack_chunked_array__': True, 'shape': […], 'chunks': {'0': ext, …}}
    of flat chunks;
  * dict keys are strings; tuples / lists of states become dicts keyed '0', '1', ….
The module tree is the one `engine.ParamLayout.flax_tree` exposes (Appendix A.6 of SURVEY.md), so a JAX-written checkpoint of
the reference loads into the engine's flat buffers and a checkpoint written here restores in Flax.  Host-side code: numpy only.
"""
from __future__ import annotations

import glob
import os
import re
from typing import Any, Dict, Optional

import msgpack
import numpy as np

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
MAX_CHUNK_BYTES = 2 ** 30


def _dtype_from_name(name: str) -> np.dtype:
    if name == "bfloat16":  # numpy has no bfloat16: keep the raw 16-bit patterns (to_float32 below widens them)
        return np.dtype(np.uint16)
    return np.dtype(name)


def _ndarray_to_ext(a: np.ndarray, code: int = _EXT_NDARRAY) -> msgpack.ExtType:
    a = np.asarray(a)
    if a.dtype.byteorder == ">":
        a = a.astype(a.dtype.newbyteorder("<"))
    payload = msgpack.packb((list(a.shape), a.dtype.name, np.ascontiguousarray(a).tobytes()), use_bin_type=True)
    return msgpack.ExtType(code, payload)


def _ext_to_ndarray(data: bytes, bf16_as_f32: bool) -> np.ndarray:
    shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
    arr = np.frombuffer(buf, dtype=_dtype_from_name(dtype_name)).reshape(shape)
    if dtype_name == "bfloat16" and bf16_as_f32:
        arr = (arr.astype(np.uint32) << 16).view(np.float32)
    return arr


def _encode(obj: Any, max_chunk: int) -> Any:
    if isinstance(obj, dict):
        return {str(k): _encode(v, max_chunk) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return {str(i): _encode(v, max_chunk) for i, v in enumerate(obj)}
    if isinstance(obj, np.generic):
        return _ndarray_to_ext(np.asarray(obj), _EXT_NPSCALAR)
    if isinstance(obj, np.ndarray):
        if obj.nbytes > max_chunk:
            flat = np.ascontiguousarray(obj).reshape(-1)
            per = max(1, max_chunk // flat.dtype.itemsize)
            chunks = {str(i): _ndarray_to_ext(flat[s:s + per]) for i, s in enumerate(range(0, flat.size, per))}
            return {"__msgpack_chunked_array__": True, "shape": {str(i): int(d) for i, d in enumerate(obj.shape)}, "chunks": chunks}
        return _ndarray_to_ext(obj)
    if isinstance(obj, complex):
        return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((obj.real, obj.imag)))
    if hasattr(obj, "detach") and hasattr(obj, "cpu"):  # torch tensor
        return _encode(obj.detach().cpu().numpy(), max_chunk)
    if obj is None or isinstance(obj, (bool, int, float, str, bytes)):
        return obj
    raise TypeError(f"cannot serialise {type(obj).__name__} into a Flax checkpoint")


def msgpack_serialize(tree: Any, max_chunk_bytes: int = MAX_CHUNK_BYTES) -> bytes:
    """flax.serialization.msgpack_serialize: nested dict / list of numpy arrays (or torch tensors) -> bytes."""
    return msgpack.packb(_encode(tree, max_chunk_bytes), use_bin_type=True)


def msgpack_restore(data: bytes, bf16_as_f32: bool = True) -> Any:
    """flax.serialization.msgpack_restore: bytes -> nested dict of numpy arrays (chunked arrays re-assembled)."""

    def ext_hook(code, payload):
        if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
            a = _ext_to_ndarray(payload, bf16_as_f32)
            return a[()] if code == _EXT_NPSCALAR and a.shape == () else a
        if code == _EXT_COMPLEX:
            re_, im = msgpack.unpackb(payload)
            return complex(re_, im)
        return msgpack.ExtType(code, payload)

    def unchunk(obj):
        if isinstance(obj, dict):
            if obj.get("__msgpack_chunked_array__"):
                shape = [obj["shape"][k] for k in sorted(obj["shape"], key=int)] if isinstance(obj["shape"], dict) else list(obj["shape"])
                ch = obj["chunks"]
                parts = [ch[k] for k in sorted(ch, key=int)] if isinstance(ch, dict) else list(ch)
                return np.concatenate([np.asarray(p).reshape(-1) for p in parts]).reshape(shape)
            return {k: unchunk(v) for k, v in obj.items()}
        return obj

    return unchunk(msgpack.unpackb(data, ext_hook=ext_hook, raw=False, strict_map_key=False))


# ------------------------------------------------------------------------------------------------ TrainState <-> engine
def latest_checkpoint(ckpt_dir: str, prefix: str = "checkpoint_") -> Optional[str]:
    """flax.training.checkpoints.latest_checkpoint: the file `<prefix><step>` with the largest step (no extension)."""
    best, best_step = None, -1
    for f in glob.glob(os.path.join(ckpt_dir, prefix + "*")):
        m = re.fullmatch(re.escape(prefix) + r"(\d+)", os.path.basename(f))
        if m and int(m.group(1)) > best_step:
            best, best_step = f, int(m.group(1))
    return best


def read_train_state(path: str) -> Dict[str, Any]:
    with open(path, "rb") as f:
        return msgpack_restore(f.read())


def _module_tree(node: Any) -> Dict[str, Any]:
    """TrainState.params is the whole variables dict {'params': tree}; accept either level."""
    while isinstance(node, dict) and set(node.keys()) == {"params"}:
        node = node["params"]
    return {"params": node}


def load_into_engine(eng, state: Dict[str, Any], load_optimizer: bool = True) -> int:
    """Copy a restored TrainState (or a bare variables / params dict) into the engine's flat buffers.  Returns the step."""
    import torch

    from .engine import _copy_tree

    params = state["params"] if "opt_state" in state or "step" in state else state
    eng.load_params(_module_tree(params))
    step = int(np.asarray(state.get("step", 0))) if isinstance(state, dict) else 0
    adam = None
    if load_optimizer and isinstance(state.get("opt_state"), dict):
        for v in state["opt_state"].values():
            if isinstance(v, dict) and "mu" in v and "nu" in v:
                adam = v
    if adam is not None:
        if eng.adam_m is None:
            eng.adam_m = torch.zeros_like(eng.params)
            eng.adam_v = torch.zeros_like(eng.params)
        _copy_tree(eng.layout.flax_tree(eng.adam_m), _module_tree(adam["mu"]))
        _copy_tree(eng.layout.flax_tree(eng.adam_v), _module_tree(adam["nu"]))
        eng.step_count = int(np.asarray(adam.get("count", step)))
    return step


def train_state_dict(eng, step: int) -> Dict[str, Any]:
    """The state dict `flax.serialization.to_state_dict(train_state)` would give for the reference's optimizer chain."""
    def tree(flat):
        return _to_numpy(eng.layout.flax_tree(flat))

    def _to_numpy(t):
        if isinstance(t, dict):
            return {k: _to_numpy(v) for k, v in t.items()}
        return t.detach().cpu().numpy().copy()

    count = np.asarray(eng.step_count, dtype=np.int32)
    if eng.adam_m is not None:
        mu, nu = tree(eng.adam_m), tree(eng.adam_v)
    else:
        zeros = eng.params.new_zeros(eng.params.shape)
        mu, nu = tree(zeros), tree(zeros)
    return {"step": np.asarray(step, dtype=np.int32), "params": tree(eng.params),
            "opt_state": {"0": {}, "1": {"count": count, "mu": mu, "nu": nu}, "2": {}, "3": {"count": count}}}


def save_from_engine(eng, ckpt_dir: str, step: int, keep: int = 3, prefix: str = "checkpoint_") -> str:
    """checkpoints.save_checkpoint(dir, train_state, step, keep=3) (train.py:123-127): atomic write, oldest files pruned."""
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f"{prefix}{step}")
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(msgpack_serialize(train_state_dict(eng, step)))
    os.replace(tmp, path)
    files = sorted((f for f in glob.glob(os.path.join(ckpt_dir, prefix + "*")) if re.fullmatch(re.escape(prefix) + r"\d+", os.path.basename(f))),
                   key=lambda p: int(os.path.basename(p)[len(prefix):]))
    for old in files[:-keep] if keep > 0 else []:
        os.remove(old)
    return path
