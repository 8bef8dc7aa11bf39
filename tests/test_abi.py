"""CPU-side checks of the drop-in boundary: libsavit.so builds, loads, and exports exactly the symbols
include/savit.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    import savit_amd

    return savit_amd


def _declared():
    src = open(os.path.join(ROOT, "include", "savit.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(savit_\w+)\s*\(", src)))


def test_header_symbols_exported(built):
    lib = built.lib.load()
    names = _declared()
    assert names, "no declarations parsed"
    for n in names:
        assert hasattr(lib, n), f"{n} declared in savit.h but not exported"
    assert names == built.lib.exported_symbols(), "lib.py signature table and savit.h disagree"
    assert lib.savit_abi_version() == 1


def test_gemm_args_struct_layout(built):
    """ctypes mirror must match the C struct: 9 pointers then 19 4-byte scalars (+ padding to 8)."""
    assert ctypes.sizeof(built.lib.GemmArgs) == 9 * 8 + 19 * 4 + 4
    assert built.lib.GemmArgs.M.offset == 72 and built.lib.GemmArgs.tile.offset == 72 + 17 * 4


def test_no_cpu_fallback(built):
    import torch

    from savit_amd import ops

    with pytest.raises(ValueError, match="no CPU path"):
        ops.layernorm_fwd(torch.zeros((4, 64)), torch.ones(64), torch.zeros(64))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "self-attention-experiments-vision_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M), f"{f} imports the oracle"
                assert "from oracle" not in s and "import oracle" not in s
