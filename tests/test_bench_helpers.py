"""bench.py's accounting helpers (no GPU): launch label -> kernel class / rocprofv3 symbol / algorithmic flops, and the choice of the
dominant kernel - the parts of the `roofline` object that are not measurements."""
import types

import bench
import savit_amd  # noqa: F401
from savit_amd import lib as _lib
from savit_amd.config import get_config


def _fake_engine(cfg, B):
    eng = types.SimpleNamespace()
    eng.L = _lib.load()
    eng.M = B * cfg.seq_len
    eng.group_flops = {"wgrad.group.0.l11-l9": 8.0e11}
    return eng


def test_kernel_class_of_every_label_kind():
    kc = bench.kernel_class
    assert kc("l3.qkv") == "gemm_tn" and kc("l3.fc2.dgrad") == "gemm_tn" and kc("head") == "gemm_tn"
    assert kc("wgrad.group.2.l7-l5") == "gemm_wgrad" and kc("l0.Wo.wgrad") == "gemm_wgrad" and kc("l0.Wo.wgrad.reduce") == "wgrad_reduce"
    assert kc("l1.attn") == "attention_fwd" and kc("l1.attn.bwd") == "attention_bwd"
    assert kc("l1.ln2") == "layernorm_fwd" and kc("l1.ln2.bwd") == "layernorm_bwd" and kc("lnf.bwd") == "layernorm_bwd"
    assert kc("zero.grads") == "memset" and kc("adamw") == "optimizer" and kc("sumsq") == "optimizer" and kc("cast W1") == "optimizer"
    assert kc("cast W1#1") == "optimizer"  # second launch of a label inside one instrumented step
    assert kc("xent") == "other"


def test_symbol_tables_and_dominant_kernel():
    cfg = get_config("vit_b_patch16")
    B = 128
    eng = _fake_engine(cfg, B)
    M, d, F = eng.M, cfg.embed_dim, cfg.hidden
    times = {}
    for l in range(12):
        times.update({f"l{l}.qkv": 0.085, f"l{l}.proj": 0.055, f"l{l}.fc1": 0.135, f"l{l}.fc2": 0.111, f"l{l}.fc2.dgrad": 0.157,
                      f"l{l}.fc1.dgrad": 0.089, f"l{l}.proj.dgrad": 0.033, f"l{l}.qkv.dgrad": 0.078, f"l{l}.attn": 0.045, f"l{l}.ln1": 0.023})
    times["wgrad.group.0.l11-l9"] = 0.70
    sym, cls = bench.symbol_tables(times, cfg, eng, B)
    plain = "gemm_tn_pp320_kernel<0>"
    assert sym[plain]["n"] == 48 and abs(sym[plain]["ms"] - 12 * (0.085 + 0.089 + 0.033 + 0.078)) < 1e-9
    assert abs(sym[plain]["flops"] - 12 * 2.0 * M * d * (3 * d + F + d + 3 * d)) < 1.0
    assert sym["gemm_tn_pp320_kernel<1>"]["flops"] == 12 * 2.0 * M * d * F
    assert sym["gemm_wgrad_group_kernel<256,256,2,4,3,32>"]["flops"] == 8.0e11
    assert cls["gemm_tn"]["n"] == 96 and cls["attention_fwd"]["n"] == 12
    assert bench.pick_dominant(sym) == plain  # 3.42 ms against 0.70
    # within 5 % of each other: the kernel with more flops per launch wins, whatever the order
    times["wgrad.group.0.l11-l9"] = sym[plain]["ms"] * 0.97
    assert bench.pick_dominant(bench.symbol_tables(times, cfg, eng, B)[0]).startswith("gemm_wgrad_group_kernel")
    times["wgrad.group.0.l11-l9"] = sym[plain]["ms"] * 1.04
    assert bench.pick_dominant(bench.symbol_tables(times, cfg, eng, B)[0]).startswith("gemm_wgrad_group_kernel")
    times["wgrad.group.0.l11-l9"] = sym[plain]["ms"] * 0.90
    assert bench.pick_dominant(bench.symbol_tables(times, cfg, eng, B)[0]) == plain
