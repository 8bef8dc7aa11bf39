"""Fingerprints of the device code inside libsavit.so (pure Python, no torch, no GPU).

Why: `roofline.traffic` in bench.py's JSON line comes from committed rocprofv3 --pmc passes (profiles/rNN_pmc_traffic.json) - a
counter pass cannot run inside the bench process.  A committed figure silently goes stale when the kernel it was measured on is
edited or renamed.  `tools/pmc_summary.py` therefore stores, per kernel, the SHA-256 of the kernel's machine code as it stood in the
library the counters were collected on, plus the library's symbol-set hash and build id; bench.py compares them with the RUNNING
library and reports `"traffic_stale": true` instead of a number that belongs to other code.

The library is a host ELF whose `.hip_fatbin` section holds one clang offload bundle per translation unit; each bundle carries one
gfx950 code object (an ELF64 of its own) whose symbol table names every kernel (`STT_FUNC`, with size)."""
from __future__ import annotations

import hashlib
import os
import re
import struct
from typing import Dict, List, Optional, Tuple

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _bundles(blob: bytes) -> List[Tuple[str, bytes]]:
    """[(target triple, code object bytes)] of every uncompressed offload bundle in `blob`."""
    out = []
    pos = blob.find(_MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", blob, pos + len(_MAGIC))
        cur = pos + len(_MAGIC) + 8
        if 0 < n < 64:
            for _ in range(n):
                off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
                triple = blob[cur + 24:cur + 24 + tlen].decode("ascii", "replace")
                cur += 24 + tlen
                out.append((triple, blob[pos + off:pos + off + size]))
        pos = blob.find(_MAGIC, pos + len(_MAGIC))
    return out


def _elf_functions(obj: bytes) -> Dict[str, bytes]:
    """{symbol name: its bytes} for every defined STT_FUNC symbol of an ELF64 little-endian object."""
    if obj[:4] != b"\x7fELF" or obj[4] != 2 or obj[5] != 1:
        return {}
    shoff, = struct.unpack_from("<Q", obj, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", obj, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", obj, shoff + i * shentsize) for i in range(shnum)]
    funcs: Dict[str, bytes] = {}
    for (_, typ, _, _, off, size, link, _, _, entsize) in secs:
        if typ != 2 or not entsize:  # SHT_SYMTAB
            continue
        stroff = secs[link][4]
        for i in range(size // entsize):
            name_i, info, _, shndx, value, ssize = struct.unpack_from("<IBBHQQ", obj, off + i * entsize)
            if (info & 0xF) != 2 or shndx == 0 or shndx >= shnum or ssize == 0:  # STT_FUNC, defined
                continue
            end = obj.index(b"\0", stroff + name_i)
            name = obj[stroff + name_i:end].decode("ascii", "replace")
            _, _, _, saddr, soff, ssz, _, _, _, _ = secs[shndx]
            lo = value - saddr + soff
            if 0 <= lo and lo + ssize <= len(obj):
                funcs[name] = obj[lo:lo + ssize]
    return funcs


_TARG = re.compile(r"L([a-z])(n?)(\d+)E")


def short_name(mangled: str) -> str:
    """`_Z23gemm_wgrad_group_kernelILi256ELi256ELi2ELi4ELi3ELi32EEv...` -> `gemm_wgrad_group_kernel<256,256,2,4,3,32>` (the form
    rocprofv3 and bench.kernel_symbol use).  Only integral / bool template arguments are rendered - every kernel template of this
    library takes nothing else; a name this cannot read is returned as it is."""
    if not mangled.startswith("_Z"):
        return mangled
    pos, name = 2, None
    nested = mangled.startswith("_ZN")
    if nested:
        pos = 3
    while True:  # <source-name>s; inside N...E the LAST one before the template arguments / the closing E is the function
        m = re.compile(r"(\d+)").match(mangled, pos)
        if not m:
            break
        n = int(m.group(1))
        name = mangled[m.end():m.end() + n]
        pos = m.end() + n
        if not nested:
            break
    if name is None:
        return mangled
    rest = mangled[pos:]
    if not rest.startswith("I"):
        return name
    args, pos = [], 1
    while pos < len(rest) and rest[pos] != "E":
        a = _TARG.match(rest, pos)
        if not a:
            return mangled
        v = ("-" if a.group(2) else "") + a.group(3)
        args.append(("true" if v != "0" else "false") if a.group(1) == "b" else v)
        pos = a.end()
    return f"{name}<{','.join(args)}>"


def library_fingerprint(path: str) -> dict:
    """{"build_id": hex of the host ELF's GNU build id (sha256 of the file when it has none), "symbols_hash": sha256 over the sorted
    gfx950 kernel symbols, "kernels": {short name: sha256 of the kernel's machine code}}"""
    blob = open(path, "rb").read()
    kernels: Dict[str, str] = {}
    names: List[str] = []
    for triple, obj in _bundles(blob):
        if "gfx950" not in triple:
            continue
        for sym, code in _elf_functions(obj).items():
            names.append(sym)
            kernels[short_name(sym)] = hashlib.sha256(code).hexdigest()
    return {"build_id": _build_id(blob), "symbols_hash": hashlib.sha256("\n".join(sorted(names)).encode()).hexdigest(), "kernels": kernels}


def _build_id(blob: bytes) -> str:
    i = blob.find(b"GNU\0", 0)
    while i >= 12:
        namesz, descsz, typ = struct.unpack_from("<III", blob, i - 12)
        if namesz == 4 and typ == 3 and 8 <= descsz <= 64:  # NT_GNU_BUILD_ID
            return blob[i + 4:i + 4 + descsz].hex()
        i = blob.find(b"GNU\0", i + 4)
    return "sha256:" + hashlib.sha256(blob).hexdigest()


def default_library() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsavit.so")


def traffic_is_stale(pmc_json: dict, kernel: str, running: Optional[dict] = None) -> Tuple[bool, str]:
    """Is the committed PMC figure of `kernel` still a measurement of the code that is running?  -> (stale, why).
    Fresh only if the file carries a fingerprint and the kernel's code hash equals the running library's."""
    running = running if running is not None else library_fingerprint(default_library())
    fp = pmc_json.get("library")
    if not fp:
        return True, "the PMC file carries no library fingerprint (written before round 5)"
    want = (fp.get("kernels") or {}).get(kernel)
    have = running["kernels"].get(kernel)
    if have is None:
        return True, f"the running library has no kernel {kernel}"
    if want is None:
        return True, f"the PMC file has no code hash for {kernel}"
    if want != have:
        return True, f"{kernel} was rebuilt since the counters were collected (code hash {want[:12]} -> {have[:12]})"
    return False, "kernel code identical to the profiled build" + ("" if fp.get("build_id") == running["build_id"] else " (other kernels of the library changed)")
