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
    assert lib.savit_abi_version() == built.lib.ABI_VERSION == 2


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


def test_shape_contracts_are_checked_on_the_host(built):
    """Every entry point validates its arguments BEFORE touching the GPU (a wrong shape must never reach a hand-written kernel):
    contract violations return SAVIT_EINVAL (1001) - checked here without a GPU, with fake non-null pointers."""
    L = built.lib.load()
    P, EINVAL = 0x1000, 1001  # 16-B aligned dummy address, never dereferenced on these paths
    # one-wave-per-sequence attention accepts exactly 16 tokens x 4 heads x 16 padded columns
    assert L.savit_seq16_attention_fwd(P, P, 8, 16, 6, 16, 192, None) == EINVAL
    assert L.savit_seq16_attention_fwd(P, P, 8, 32, 4, 16, 192, None) == EINVAL
    assert L.savit_seq16_attention_bwd(P, P, P, 8, 16, 4, 16, 200, 1.0, None) == EINVAL
    # tiled attention: head_dim outside {16, 32, 48, 64}, too many tokens
    assert L.savit_attention_fwd(P, P, None, 2, 197, 3, 40, 3 * 3 * 40, None) == EINVAL
    assert L.savit_attention_fwd(P, P, None, 2, 70000, 3, 64, 3 * 3 * 64, None) == EINVAL
    # per-image transpose: pitches must be multiples of 8 and cover the matrix; residual in/out come together
    assert L.savit_transpose_bf16(P, 196 * 128, 128, P, 128 * 190, 190, 2, 196, 128, None, None, 0, None, 0, None) == EINVAL  # ld_dst < R
    assert L.savit_transpose_bf16(P, 196 * 124, 124, P, 128 * 200, 200, 2, 196, 128, None, None, 0, None, 0, None) == EINVAL  # ld_src < Cc
    assert L.savit_transpose_bf16(P, 196 * 128, 128, None, 0, 128, 2, 128, 196, P, None, 0, None, 0, None) == EINVAL        # resid without out
    # column-sum helpers are written for d <= 1024, LayerNorm for d % 4 == 0
    assert L.savit_cast_colsum(P, P, P, 10, 2048, None) == EINVAL
    assert L.savit_tnt_inner2outer_split(P, P, P, None, 2, 1, 64, None) == EINVAL  # needs a cls row plus at least one patch row
    assert L.savit_layernorm_fwd(P, P, P, P, None, None, 4, 30, 32, 1e-6, 1, None) == EINVAL
    assert L.savit_tnt_pixel_gather(P, P, 2, 224, 16, 5, 3, 80, None) == EINVAL  # patch % transformed patch != 0
    # GEMM: K must be a multiple of the 32-deep K-step, pitches multiples of 8
    a = built.lib.GemmArgs()
    a.A, a.Bt, a.C, a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.epilogue, a.rows_per_sample = P, P, P, 64, 64, 40, 40, 40, 64, 0, 1
    assert L.savit_gemm_bf16_tn(ctypes.byref(a), None) == EINVAL
    a.K, a.ldb, a.lda = 64, 64, 20  # an operand pitch that is not a multiple of 8 elements
    assert L.savit_gemm_bf16_tn(ctypes.byref(a), None) == EINVAL


def test_product_library_has_no_experiment_switches(built):
    """the shipped library reads no environment variable, has no run-time ablation path, and rejects the
    timing-only tile ids (they exist only in tools/build_variant.sh builds, which define SAVIT_EXPERIMENTS)."""
    import subprocess

    so = built.lib.LIB_PATH
    text = subprocess.run(["strings", so], check=True, capture_output=True, text=True).stdout
    bad = [ln for ln in text.splitlines() if re.search(r"SAVIT_\w*(ABL|DEBUG|TILE|VARIANT|PP_|NO_TAIL|GENERAL|FUSED)", ln)]
    assert not bad, bad
    undefined = subprocess.run(["nm", "-D", "--undefined-only", so], check=True, capture_output=True, text=True).stdout
    assert "getenv" not in undefined, "libsavit.so imports getenv"
    L = built.lib.load()
    P = 0x1000
    a = built.lib.GemmArgs()
    a.A, a.Bt, a.C, a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.epilogue, a.rows_per_sample = P, P, P, 4096, 4096, 4096, 4096, 4096, 4096, 0, 1
    for tile in (100, 101, 102, 103, 104, 107, 108, 109, 110, 255):
        a.tile = tile
        assert L.savit_gemm_bf16_tn(ctypes.byref(a), None) == 1001, tile
    # no per-launch driver calls: the dynamic-LDS limit is raised once per kernel symbol (SAVIT_LDS_ONCE), never inline
    csrc = os.path.join(ROOT, "self-attention-experiments-vision_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith(".hip"):
            s = open(os.path.join(csrc, f)).read()
            assert "hipFuncSetAttribute" not in s, f"{f}: per-launch hipFuncSetAttribute"
            assert "getenv" not in s, f"{f}: getenv in the library"


def test_every_compiled_tn_tile_is_reachable_from_the_auto_heuristic(built):
    """libsavit.so holds no GEMM tile nobody selects.  tests/golden/gemm_shapes.json lists every TN GEMM the engines
    of the four families launch (tools/dump_gemm_shapes.py, run on the GPU box: ViT-Ti/S/B/L, CaiT-S24/XXS24, Mixer-B/16, S/32, TNT-S/B at
    the bench batch sizes); the kernel families compiled into gemm_tn.o must be exactly the tiles savit_gemm_tn_auto_tile_cus returns for
    them (whole device and with 32 CUs reserved for a resident all-reduce), and every other tile id is rejected."""
    import json
    import shutil
    import subprocess
    import tempfile

    L = built.lib.load()
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "gemm_shapes.json")))
    reach = set()
    for v in shapes.values():
        for M, N, K, epi, alias in v["shapes"]:
            for cus in (0, 224):
                t = L.savit_gemm_tn_auto_tile_cus(M, N, K, epi, cus)
                reach.add(13 if (t in (20, 21, 22) and alias) else t)  # the ping-pong kernels do not take aliased rows: savit_gemm_bf16_tn falls back
    family = {"gemm_tn_ring_kernel<128, 128, 2, 2, 4,": 6, "gemm_tn_pair_kernel<128, 128, 2, 2, 2,": 12, "gemm_tn_pair_kernel<256, 256, 2, 4, 2,": 13,
              "gemm_tn_pair_kernel<192, 128, 2, 2, 2,": 17, "gemm_tn_pair_tail_kernel<192, 128, 128, 2, 2, 2,": 18, "gemm_tn_pp_kernel<": 20,
              "gemm_tn_pp320_kernel<": 21, "gemm_tn_pp320p_kernel<": 22, "gemm_tn_rows_kernel<": 24}
    assert reach == set(family.values()), sorted(reach)
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(readelf) and os.path.exists(objdump) and shutil.which("c++filt")):
        pytest.skip("llvm-readelf / c++filt not available")
    with tempfile.TemporaryDirectory() as d:
        dst = os.path.join(d, "gemm_tn.o")
        shutil.copy(os.path.join(ROOT, "self-attention-experiments-vision_amd", "csrc", "gemm_tn.o"), dst)
        subprocess.run([objdump, "--offloading", dst], check=True, capture_output=True)
        co = [f for f in os.listdir(d) if "gfx950" in f][0]
        notes = subprocess.run([readelf, "--notes", os.path.join(d, co)], check=True, capture_output=True, text=True).stdout
    mangled = re.findall(r"\.name:\s+(\S+)", notes)
    names = subprocess.run(["c++filt"] + mangled, check=True, capture_output=True, text=True).stdout.splitlines()
    kernels = sorted({re.sub(r"^void \(anonymous namespace\)::", "", n).split("(")[0] for n in names if "gemm_tn_" in n})
    assert kernels, "no TN GEMM kernel found in gemm_tn.o"
    compiled = set()
    for k in kernels:
        hit = [t for pre, t in family.items() if k.startswith(pre)]
        assert len(hit) == 1, f"{k}: a kernel family no tile of the auto heuristic maps to"
        compiled.add(hit[0])
    assert compiled == reach, (sorted(compiled), sorted(reach))
    P = 0x1000
    a = built.lib.GemmArgs()
    a.A, a.Bt, a.C, a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.epilogue, a.rows_per_sample = P, P, P, 4096, 4096, 4096, 4096, 4096, 4096, 0, 1
    for tile in sorted(set(range(1, 40)) - reach):
        a.tile = tile
        assert L.savit_gemm_bf16_tn(ctypes.byref(a), None) == 1001, tile


def test_tile_choice_follows_the_cu_budget(built):
    """A data-parallel rank that leaves CUs to the resident all-reduce (train.py:96) gets grids priced for the CUs that remain: DeiT-B's
    N = 768 products are ONE round of 237 tiles of 320 rows on 256 (or 240) CUs and two rounds on 224 - the heuristic must move off that tile - and
    the column-sum slab height follows the tile."""
    L = built.lib.load()
    M = 128 * 197
    assert L.savit_gemm_tn_auto_tile_cus(M, 768, 768, 0, 0) == 21 == L.savit_gemm_tn_auto_tile_epi(M, 768, 768, 0)
    assert L.savit_gemm_tn_auto_tile_cus(M, 768, 768, 0, 256) == 21 and L.savit_gemm_tn_auto_tile_cus(M, 768, 768, 0, 1000) == 21
    assert L.savit_gemm_tn_auto_tile_cus(M, 768, 768, 0, 240) == 21  # 237 tiles still fit one round of 240 CUs
    t = L.savit_gemm_tn_auto_tile_cus(M, 768, 768, 0, 224)            # ... not one of 224
    assert t in (17, 18, 20)
    for cus in (0, 240, 224):
        tile = L.savit_gemm_tn_auto_tile_cus(M, 3072, 768, 3, cus)
        assert L.savit_gemm_colsum_rows_cus(M, 3072, 768, 0, cus) == L.savit_gemm_colsum_rows(M, 3072, 768, tile)
    assert L.savit_gemm_wgrad_group_tiles(384, 1152, 640) == 5 and L.savit_gemm_wgrad_group_tiles(384, 384, 640) == 2
    assert L.savit_gemm_wgrad_group_tiles(384, 1536, 640) == 6 and L.savit_gemm_wgrad_group_tiles(1536, 384, 640) == 6
    assert L.savit_gemm_wgrad_group_tiles(384, 1152, 384) == 9 and L.savit_gemm_wgrad_group_tiles(768, 768, 256) == 9
    assert L.savit_gemm_wgrad_group_tiles(768, 768, 192) == 0


def _device_isa(obj_name, tmp_path):
    """Disassemble the gfx950 code object embedded in csrc/<obj_name> (llvm-objdump --offloading extracts next to its input)."""
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    src = os.path.join(ROOT, "self-attention-experiments-vision_amd", "csrc", obj_name)
    dst = os.path.join(str(tmp_path), obj_name)
    shutil.copy(src, dst)
    subprocess.run([objdump, "--offloading", dst], check=True, capture_output=True)
    cos = [f for f in os.listdir(str(tmp_path)) if "gfx950" in f]
    assert len(cos) == 1, cos
    return subprocess.run([objdump, "-d", os.path.join(str(tmp_path), cos[0])], check=True, capture_output=True, text=True).stdout


def test_layernorm_forward_has_no_packed_fp32_math(built, tmp_path):
    """The LayerNorm forward kernels must not contain v_pk_*_f32 (csrc/layernorm_fwd.hip explains why: the in-place packed
    subtraction of the mean was the one instruction that misbehaved under GPU time-slicing between processes)."""
    isa = _device_isa("layernorm_fwd.o", tmp_path)
    assert "ln_fwd_kernel" in isa and "ln_fwd_narrow_kernel" in isa
    assert not re.search(r"\bv_pk_\w+_f32\b", isa), "packed fp32 math in the LayerNorm forward object"


def test_no_hand_timed_permlane_swaps(built, tmp_path):
    """Lane swaps go through the compiler builtins (common.h), so hipcc places their hazard wait states: the library sources hold
    no inline-asm v_permlane*_swap any more."""
    csrc = os.path.join(ROOT, "self-attention-experiments-vision_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            s = open(os.path.join(csrc, f)).read()
            assert not re.search(r'"[^"\n]*v_permlane(16|32)_swap', s), f"inline-asm permlane swap in {f}"


def test_talking_heads_coefficient_loads_are_not_touched_in_flight(built, tmp_path):
    """The packed head mixes of the talking-heads row kernels fetch a coefficient matrix with inline-asm s_load_dwordx16 and wait for
    it later (ThCoef::issue / wait, csrc/attention.hip).  hipcc does not know the destination SGPRs are in flight in between: the
    built object must not read or copy them before the wait (tools/check_inflight_sgprs.py)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("check_inflight_sgprs", os.path.join(ROOT, "tools", "check_inflight_sgprs.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    isa = _device_isa("attention.o", tmp_path)
    kernels = re.split(r"\n(?=[0-9a-f]+ <)", isa)
    seen = 0
    for k in kernels:
        head = k.split("\n", 1)[0]
        if "s_load_dwordx16" in k:  # EVERY kernel that uses the in-flight scalar loads, whatever its name
            assert "th_softmax_fwd_kernel" in head or "th_softmax_bwd_kernel" in head, f"{head}: a new user of ThCoef - add it here knowingly"
            n, bad = chk.violations(k)
            seen += n
            assert not bad, (head, bad[:3])
    assert seen >= 4 * 2 + 4 * 3  # H = 8: 4 quads x (2 matrices forward, 3 backward); H = 4 adds one quad per matrix use


def test_overlapped_kernels_hold_no_scratch_and_no_builtin_waits(built, tmp_path):
    """The persistent attention backward keeps an LDS-DMA in flight across its passes.  Two things silently turn that into no overlap at
    all: a spilled register (its reload is a VMEM operation that waits for everything issued before it) and a
    `s_waitcnt vmcnt(0)` that hipcc puts in front of LDS reads it can attribute (the ds_read_tr16 builtin, typed float reads).  So: no
    scratch in the instantiations the BASELINE geometries launch, no scratch_* instruction in them, and between the two barriers of a
    pass no full vmcnt wait in the loop bodies (the only vmcnt(0) of the kernel are the ones in front of its barriers)."""
    import re
    import shutil
    import subprocess

    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(readelf) and shutil.which("c++filt")):
        pytest.skip("llvm-readelf / c++filt not available")
    isa = _device_isa("attention.o", tmp_path)
    co = [f for f in os.listdir(str(tmp_path)) if "gfx950" in f][0]
    notes = subprocess.run([readelf, "--notes", os.path.join(str(tmp_path), co)], check=True, capture_output=True, text=True).stdout
    scratch = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        scratch[name] = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
    checked = 0
    for mangled, bytes_ in scratch.items():
        if not re.search(r"attn_bwd_persl?_kernelILi[567]E|attn_fwdl?_kernelILi7E|attn_[fb]wd2_kernelILi4E", mangled):
            continue
        checked += 1
        assert bytes_ == 0, f"{mangled}: {bytes_} bytes of scratch"
        body = isa.split("<" + mangled + ">:")[1].split("\n\n")[0]
        assert "scratch_" not in body, mangled
    assert checked >= 10, checked
    # the persistent backward for DeiT's N = 197: every full vmcnt wait sits right in front of an s_barrier (its passes have none)
    body = isa.split("<" + [m for m in scratch if "attn_bwd_pers_kernelILi7E" in m][0] + ">:")[1].split("\n\n")[0]
    lines = [ln.split("//")[0].strip() for ln in body.splitlines() if ln.strip()]
    ops = [ln for ln in lines if re.match(r"^(s_waitcnt|s_barrier|v_mfma|buffer_load|global_load|global_store|s_cbranch)", ln)]
    for i, op in enumerate(ops):
        if op.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", op):
            window = ops[i + 1:i + 4]
            assert any(w.startswith("s_barrier") or (w.startswith("s_waitcnt") and "vmcnt(0)" in w) for w in window) or \
                not any(w.startswith("v_mfma") for w in ops[i + 1:i + 3]), ("a full vmcnt wait inside a pass", i, op, window)


def _load_tool(name):
    import importlib.util

    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


INFLIGHT_KERNELS = {  # object -> kernels whose LDS reads are inline asm with a hand-placed wait
    "attention.o": r"attn_bwd_persl?_kernelILi[1-7]E|attn_[fb]wd2_kernelILi[1-4]E|attn_fwdl?_kernelILi[1-8]E|th_softmax_(fwd|bwd)_kernel",
    "gemm_wgrad.o": r"gemm_wgrad_group_kernel|gemm_wgrad_group_mixed_kernel|gemm_wgrad_ring_kernel",
    "gemm_tn.o": r"gemm_tn_pp320_kernel|gemm_tn_pp_kernel|gemm_tn_pers_kernel",
}


@pytest.mark.parametrize("obj", sorted(INFLIGHT_KERNELS))
def test_no_vgpr_is_touched_while_its_lds_read_is_in_flight(built, tmp_path, obj):
    """The transposed / 16-byte LDS reads of the overlapped kernels are inline asm whose wait comes later by hand (TrFrag, lds_read_f4,
    SAVIT_TR_READ, the park reads): hipcc believes the destination VGPRs are defined at the asm and may read, copy or re-use them
    before the `s_waitcnt lgkmcnt` that retires the read.  tools/check_inflight_vgprs.py follows the LDS queue through the control
    flow of the BUILT kernels and fails on any such instruction."""
    chk = _load_tool("check_inflight_vgprs")
    kernels = chk.split_kernels(_device_isa(obj, tmp_path))
    pat = re.compile(INFLIGHT_KERNELS[obj])
    seen = reads = 0
    for name, body in kernels.items():
        if not pat.search(name):
            continue
        n, bad = chk.violations(body)
        assert not bad, (name, bad[:4])
        assert chk.violations.last_unreached == 0, (name, "the control-flow walk missed instructions", chk.violations.last_unreached)
        seen += 1
        reads += n
    assert seen >= 4 and reads >= 100, (seen, reads)
    if obj == "gemm_wgrad.o":  # the kernels the verdict names must be among them
        assert any("gemm_wgrad_group_kernel" in k for k in kernels) and any("gemm_wgrad_group_mixed_kernel" in k for k in kernels)
    if obj == "attention.o":
        for need in ("attn_bwd_pers_kernelILi7E", "attn_bwd_persl_kernelILi7E", "attn_fwdl_kernelILi7E", "attn_fwd2_kernelILi4E", "attn_bwd2_kernelILi4E"):
            assert any(need in k for k in kernels), need


def test_inflight_vgpr_guard_fails_on_a_copy_before_the_wait(built, tmp_path):
    """The guard must FAIL on an object that has the defect: (a) a hand-made kernel, compiled here, with a `v_mov` of the destination
    between the asm `ds_read_b64_tr_b16` and its `s_waitcnt`; (b) the real weight-gradient kernel's listing with such a copy spliced
    in behind its first transposed read, and with the read's wait removed (a use behind the loop exit)."""
    import shutil
    import subprocess

    chk = _load_tool("check_inflight_vgprs")
    if shutil.which("hipcc"):
        src = tmp_path / "bad.hip"
        src.write_text(r'''
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
__global__ void bad_copy_kernel(unsigned* out, int n) {
  __shared__ unsigned buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = i * n;
  __syncthreads();
  unsigned addr = (unsigned)(threadIdx.x * 8);
  u2 v; unsigned stale;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  asm volatile("v_mov_b32 %0, %1" : "=v"(stale) : "v"(v.x));   // the copy a live-range split would make: reads v in flight
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
  out[threadIdx.x] = v.x + v.y + stale;
}
__global__ void good_kernel(unsigned* out, int n) {
  __shared__ unsigned buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = i * n;
  __syncthreads();
  unsigned addr = (unsigned)(threadIdx.x * 8);
  u2 v; unsigned later;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
  asm volatile("v_mov_b32 %0, %1" : "=v"(later) : "v"(v.x));
  out[threadIdx.x] = v.x + v.y + later;
}
''')
        obj = tmp_path / "bad.o"
        subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-c", str(src), "-o", str(obj)], check=True, capture_output=True)
        sub = tmp_path / "x"
        sub.mkdir()
        shutil.copy(str(obj), str(sub / "bad.o"))
        objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
        subprocess.run([objdump, "--offloading", str(sub / "bad.o")], check=True, capture_output=True)
        co = [f for f in os.listdir(str(sub)) if "gfx950" in f][0]
        isa = subprocess.run([objdump, "-d", str(sub / co)], check=True, capture_output=True, text=True).stdout
        ks = chk.split_kernels(isa)
        nb, bad = chk.violations([b for k, b in ks.items() if "bad_copy_kernel" in k][0])
        assert nb >= 1 and bad and any("v_mov_b32" in c for _, c in bad), bad
        ng, good = chk.violations([b for k, b in ks.items() if "good_kernel" in k][0])
        assert ng >= 1 and not good, good
    # (b) the real kernel, with the defect spliced into its listing
    kernels = chk.split_kernels(_device_isa("gemm_wgrad.o", tmp_path))
    body = [b for k, b in kernels.items() if "gemm_wgrad_group_kernelILi256ELi256" in k][0]
    assert not chk.violations(body)[1]
    lines = body.splitlines()
    at = next(i for i, ln in enumerate(lines) if "ds_read_b64_tr_b16" in ln)
    dst = re.search(r"v\[(\d+):\d+\]", lines[at]).group(1)
    spliced = lines[:at + 1] + [f"\tv_mov_b32_e32 v255, v{dst}                // 00000FFFFFF0: 7E000000"] + lines[at + 1:]
    assert any("v255" in c for _, c in chk.violations("\n".join(spliced))[1])
    no_wait = [ln for ln in lines if not re.search(r"s_waitcnt\s+lgkmcnt", ln.split("//")[0])]
    assert chk.violations("\n".join(no_wait))[1], "with every lgkmcnt wait removed the MFMAs read fragments in flight"


def test_persistent_attention_backward_barrier_b_waits_for_the_lds_dma(built, tmp_path):
    """barrier b of attn_bwd_pers_kernel orders every wave's K / V LDS-DMA before pass A reads rows other waves staged.  The
    wait must not depend on what hipcc attaches to __syncthreads(): the source issues `s_waitcnt vmcnt(0)` itself, and the built kernel
    must show that bare wait directly in front of an s_barrier (only further waits between them)."""
    isa = _device_isa("attention.o", tmp_path)
    found = 0
    for n in (5, 6, 7):
        body = isa.split("attn_bwd_pers_kernelILi%dE" % n, 1)[1].split("s_endpgm")[0]
        ops = [ln.split("//")[0].strip() for ln in body.splitlines() if ln.strip()]
        ops = [o for o in ops if re.match(r"^[sv]_|^ds_|^buffer_|^global_", o)]
        ok = False
        for i, o in enumerate(ops):
            if o == "s_waitcnt vmcnt(0)":
                j = i + 1
                while j < len(ops) and ops[j].startswith("s_waitcnt"):
                    j += 1
                ok = ok or (j < len(ops) and ops[j] == "s_barrier")
        assert ok, f"attn_bwd_pers_kernel<{n}>: no explicit vmcnt(0) in front of a barrier"
        found += 1
    assert found == 3


def test_every_entry_point_is_named_in_integration_md():
    """INTEGRATION.md is the switch-over guide of the boundary: every function include/savit.h declares appears there by its full name
    (round 5: two new kernels were in the header, the library and the engines - and not in the guide)."""
    import re

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    header = open(os.path.join(root, "include", "savit.h")).read()
    guide = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(savit_[a-z0-9_]+)\s*\(", header)))
    assert len(names) > 80
    missing = [n for n in names if n not in guide]
    assert not missing, missing


def test_loader_wave_kernels_publish_their_lds_dma_behind_a_full_vmcnt_wait(built, tmp_path):
    """Round 6: attn_fwdl_kernel / attn_bwd_persl_kernel give every LDS-DMA of a workgroup to ONE extra wave; the other waves read the
    images after a bare barrier.  The loader's `s_waitcnt vmcnt(0)` must therefore sit directly in front of the barriers that publish its
    requests (source: inline asm, not something hipcc attaches to a builtin), and no other wave may issue an LDS-DMA: the only
    `buffer_load ... lds` instructions of these kernels are the loader branch's - checked as a count against the launch geometry."""
    isa = _device_isa("attention.o", tmp_path)

    def ops_of(sym):
        body = isa.split(sym, 1)[1].split("s_endpgm")[0]
        ops = [ln.split("//")[0].strip() for ln in body.splitlines() if ln.strip()]
        return [o for o in ops if re.match(r"^[sv]_|^ds_|^buffer_|^global_", o)]

    def waits_before_barrier(ops):
        n = 0
        for i, o in enumerate(ops):
            if o == "s_waitcnt vmcnt(0)":
                j = i + 1
                while j < len(ops) and ops[j].startswith("s_waitcnt"):
                    j += 1
                n += j < len(ops) and ops[j] == "s_barrier"
        return n

    for nt in (5, 6, 7):
        bwd, fwd = ops_of("attn_bwd_persl_kernelILi%dE" % nt), ops_of("attn_fwdl_kernelILi%dE" % nt)
        # backward loader: prologue (barrier p), K / V (barrier b), next item's O rows + Q / dO (barrier c); forward loader: one per item
        assert waits_before_barrier(bwd) >= 3, (nt, waits_before_barrier(bwd))
        assert waits_before_barrier(fwd) >= 1, nt
        dma = lambda ops: sum(1 for o in ops if o.startswith("buffer_load_dwordx4") and " lds" in o)  # noqa: E731
        # backward: Q, dO images + O rows in the prologue and again per item, K, V per item: (2 * 4 nt + 4 nt) * 2 + 2 * 4 nt = 32 nt
        assert dma(bwd) == 32 * nt, (nt, dma(bwd))
        # forward: stage_image's loop over the instructions is rolled (one LDS-DMA instruction per image and call site: 2 images x 2 sites)
        assert 1 <= dma(fwd) <= 4 * 4 * nt, (nt, dma(fwd))
