"""bench.py's accounting helpers (no GPU): launch label -> kernel class / rocprofv3 symbol / algorithmic flops, and the choice of the
dominant kernel - the parts of the `roofline` object that are not measurements."""
import json
import os
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
    # DeiT-B: every TN product takes the one-tile 320 x 256 kernel (tile 21; the persistent tile 22 is chosen for launches of >= 6 rounds
    # with K >= 1024 only - ViT-L at 256 images)
    plain = "gemm_tn_pp320_kernel<0>"
    assert sym[plain]["n"] == 48 and abs(sym[plain]["ms"] - 12 * (0.085 + 0.089 + 0.033 + 0.078)) < 1e-9
    assert abs(sym[plain]["flops"] - 12 * 2.0 * M * d * (3 * d + F + d + 3 * d)) < 1.0
    assert sym["gemm_tn_pp320_kernel<1>"]["flops"] == 12 * 2.0 * M * d * F
    assert bench.tn_symbol(eng.L, 256 * 577, 3072, 1024, 0) == "gemm_tn_pp320p_kernel<0>" and bench.tn_symbol(eng.L, 256 * 577, 1024, 4096, 2) == "gemm_tn_pp320p_kernel<2>"
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
    # ... net of what an empty bracket costs per launch (under rocprofv3 several us: 48 brackets against 1 must not flip the choice)
    assert bench.pick_dominant(bench.symbol_tables(times, cfg, eng, B)[0], pair_overhead_ms=0.008).startswith("gemm_wgrad_group_kernel")


def test_library_fingerprint_and_traffic_staleness(tmp_path):
    """`roofline.traffic` must be impossible to go stale: tools/pmc_summary.py stores the code hash of every kernel it
    has figures for (fingerprint.py reads them out of libsavit.so); bench.traffic_lookup returns the figure only while the running
    library's hash of that kernel is the same, and says "traffic_stale": true otherwise."""
    import json

    from savit_amd import fingerprint as fpr

    fp = fpr.library_fingerprint(fpr.default_library())
    dom = "gemm_wgrad_group_kernel<256,256,2,4,3,32>"
    assert dom in fp["kernels"] and "gemm_tn_pp320_kernel<1>" in fp["kernels"] and "attn_bwd_pers_kernel<7>" in fp["kernels"]
    assert len(fp["kernels"]) > 150 and len(fp["symbols_hash"]) == 64 and not fp["build_id"].startswith("sha256:"), "link with --build-id"
    assert fpr.short_name("_ZN12_GLOBAL__N_123gemm_wgrad_group_kernelILi256ELi256ELi2ELi4ELi3ELi32EEEvNS_16WgradGroupParamsE") == dom
    assert fpr.short_name("_ZN12_GLOBAL__N_112adamw_kernelILb1EEEvPf") == "adamw_kernel<true>"
    pj = {"kernels": {dom: {"traffic_bytes": 2169000000}}, "library": {"build_id": fp["build_id"], "symbols_hash": fp["symbols_hash"],
                                                                        "kernels": {dom: fp["kernels"][dom]}}}
    (tmp_path / "r05_pmc_traffic.json").write_text(json.dumps(pj))
    (tmp_path / "r04_pmc_traffic.json").write_text(json.dumps({"kernels": {dom: {"traffic_bytes": 1}}}))  # older round: must not be picked
    r = bench.traffic_lookup(dom, True, str(tmp_path), fp)
    assert r["traffic"] == 2169000000 and r["traffic_stale"] is False and "r05_pmc_traffic.json" in r["traffic_source"]
    # the kernel was edited since the counters were collected
    changed = dict(fp, kernels=dict(fp["kernels"], **{dom: "0" * 64}))
    r = bench.traffic_lookup(dom, True, str(tmp_path), changed)
    assert r["traffic"] is None and r["traffic_stale"] is True and "rebuilt" in r["traffic_check"]
    # renamed / removed kernel, and a PMC file from before the fingerprints existed
    r = bench.traffic_lookup("gemm_wgrad_group_kernel<256,256,2,4,3,64>", True, str(tmp_path), fp)
    assert r["traffic"] is None and r["traffic_stale"] is True
    (tmp_path / "r05_pmc_traffic.json").write_text(json.dumps({"kernels": pj["kernels"]}))
    r = bench.traffic_lookup(dom, True, str(tmp_path), fp)
    assert r["traffic"] is None and r["traffic_stale"] is True and "no library fingerprint" in r["traffic_check"]
    # not the headline workload: no figure, no claim
    assert bench.traffic_lookup(dom, False, str(tmp_path), fp) == {"traffic": None}


def test_every_symbol_bench_can_name_exists_in_the_library():
    """bench.tn_symbol / WG_GROUP_TILES / kernel_symbol spell kernel symbols by hand; each must be a kernel of the built library (so
    that `roofline.kernel` always matches a row of the rocprofv3 CSV), for every TN shape the engines launch."""
    import json
    import os

    from savit_amd import fingerprint as fpr

    have = set(fpr.library_fingerprint(fpr.default_library())["kernels"])
    L = _lib.load()
    shapes = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gemm_shapes.json")))
    seen = set()
    for v in shapes.values():
        for M, N, K, epi, alias in v["shapes"]:
            for cus in (0, 240):
                seen.add(bench.tn_symbol(L, M, N, K, epi, 0, cus))
    assert len(seen) >= 8
    for sname in seen | set(bench.WG_GROUP_TILES.values()) | set(bench.WG_VARIANTS.values()):
        assert sname.replace(", ", ",") in have, sname


def test_step_roofline_leads_with_executed_flops():
    """`frac` is matrix-pipe utilisation (executed flops), `frac_counted` keeps SURVEY 8d's dense count; they differ
    for the ViT family only (last layer on the cls rows), by exactly what config.cls_only_saved_flops_per_image prices."""
    from savit_amd.config import cls_only_saved_flops_per_image, executed_flops_per_image, train_flops_per_image

    cfg = get_config("vit_b_patch16")
    fpi = train_flops_per_image(cfg)
    assert abs(fpi - 105.151758336e9) < 1 and abs(executed_flops_per_image(cfg, True, True) - 98.553212928e9) < 1
    # backward priced at 2 x forward for every product, the attention included: forward share = 1/3 of what both save
    d, N = 768, 197
    dense, attn = 2.0 * 196 * d * d + 4.0 * 196 * d * 3072, 4.0 * 196 * N * d
    assert cls_only_saved_flops_per_image(cfg, True) == 3.0 * (dense + attn) and cls_only_saved_flops_per_image(cfg, False) == 2.0 * dense
    r = bench.step_roofline(cfg, 8000.0, True, True)
    assert r["frac"] < r["frac_counted"] and abs(r["frac_counted"] - 8000 * fpi / 1e12 / bench.MFMA_BF16_PEAK_TFLOPS) < 1e-4
    assert abs(r["achieved"] - 8000 * 98.553212928e9 / 1e12) < 0.01 and r["flops_per_image"] == fpi and "executed_note" in r
    r0 = bench.step_roofline(cfg, 8000.0, False, False)
    assert r0["frac"] == r0["frac_counted"] and "executed_note" not in r0
    c = bench.step_roofline(get_config("cait_s_24"), 6800.0, False, False)
    assert c["frac"] == c["frac_counted"] and abs(c["flops_per_image"] - 55.848e9) < 1e7


def test_hbm_kernels_from_instrumented_labels():
    """north_star's "achieved HBM GB/s on the memory-bound softmax / LayerNorm" as part of the JSON line: algorithmic bytes per dense
    launch / mean instrumented launch time.  The cls-row launches of the last layer and the final LayerNorm are not bandwidth launches
    and must not dilute the averages."""
    cfg = get_config("vit_b_patch16")
    B, M = 128, 128 * 197
    labels = {"lnf": 0.004, "lnf.bwd": 0.004, "adamw": 0.43, "xent": 0.01}
    for l in range(12):
        labels.update({f"l{l}.ln1": 0.0211, f"l{l}.ln2": 0.0211 if l < 11 else 0.003, f"l{l}.ln1.bwd": 0.050,
                       f"l{l}.ln2.bwd": 0.050 if l < 11 else 0.004, f"l{l}.attn": 0.041 if l < 11 else 0.020,
                       f"l{l}.attn.bwd": 0.113 if l < 11 else 0.032, f"l{l}.qkv": 0.09})
    h = bench.hbm_kernels(labels, cfg, M, B, 86_567_656, True, True)
    assert set(h) == {"ln_fwd", "ln_bwd", "attn_fwd", "attn_bwd", "adamw"}
    assert h["ln_fwd"]["launches"] == 23 and h["ln_bwd"]["launches"] == 23 and h["attn_fwd"]["launches"] == 11 and h["attn_bwd"]["launches"] == 11
    assert h["ln_fwd"]["algorithmic_bytes"] == 6 * M * 768 + 8 * M and abs(h["ln_fwd"]["avg_us"] - 21.1) < 0.05
    assert abs(h["ln_fwd"]["TB/s"] - h["ln_fwd"]["algorithmic_bytes"] / 21.1e-6 / 1e12) < 0.01
    assert abs(h["ln_bwd"]["TB/s"] - (16 * M * 768 + 8 * M) / 50e-6 / 1e12) < 0.01 and 0.9 < h["ln_bwd"]["frac_of_6.3"] < 1.05
    assert abs(h["attn_bwd"]["TB/s"] - (16 * M * 768 + 4 * 128 * 12 * 197) / 113e-6 / 1e12) < 0.01
    assert h["adamw"]["algorithmic_bytes"] == 30 * 86_567_656
    # the dense plan: the last layer's launches are bandwidth launches too
    hd = bench.hbm_kernels(labels, cfg, M, B, 86_567_656, False, False)
    assert hd["ln_fwd"]["launches"] == 24 and hd["attn_bwd"]["launches"] == 12


def test_engine_options_env_and_keywords(monkeypatch):
    """options.EngineOptions: keyword > options object > SAVIT_* environment (read at construction) > default; unknown names raise."""
    import pytest
    from savit_amd.options import EngineOptions

    for k in [v[0] for v in EngineOptions.ENV.values()]:
        monkeypatch.delenv(k, raising=False)
    assert EngineOptions.resolve(None).non_default() == {}
    monkeypatch.setenv("SAVIT_CLS_ONLY_LAST", "0")
    monkeypatch.setenv("SAVIT_RESERVED_CUS", "24")
    monkeypatch.setenv("SAVIT_WGRAD_GROUP", "auto")
    monkeypatch.setenv("SAVIT_OVERLAP_WGRAD", "1")
    o = EngineOptions.resolve(None, reserved_cus=None, wgrad_max_lag=3)
    assert o.non_default() == {"cls_only_last": False, "reserved_cus": 24, "wgrad_max_lag": 3, "overlap_wgrad": True}
    assert EngineOptions.resolve(None, reserved_cus=8, cls_only_last=True).non_default() == {"reserved_cus": 8, "overlap_wgrad": True}
    base = EngineOptions(rows_tile=False)
    assert EngineOptions.resolve(base).non_default() == {"rows_tile": False}  # an explicit object: the environment is not consulted
    with pytest.raises(TypeError, match="unknown engine option"):
        EngineOptions.resolve(None, clz_only_last=False)
    assert set(EngineOptions.ENV) == set(EngineOptions().as_dict())


def test_sweep_watchdog_prints_the_headline_and_leaves():
    """N > 1: a reserved_cus sweep that hangs behind the measured headline must not lose the line - after the timeout rank 0 prints it
    (with the reason) and every rank leaves with status 0; other ranks print nothing."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "rank = int(sys.argv[1])\n"
            "bench.arm_sweep_watchdog(0.3, {'metric': 'images_per_sec', 'value': 1.0} if rank == 0 else {}, rank)\n"
            "time.sleep(30)\nprint('not reached')\n" % root)
    for rank in (0, 1):
        r = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=25)
        assert r.returncode == 0 and "not reached" not in r.stdout
        if rank == 0:
            line = json.loads(r.stdout.strip().splitlines()[-1])
            assert line["value"] == 1.0 and "watchdog" in line["reserved_cus_sweep_error"]
        else:
            assert r.stdout.strip() == ""
    # a cancelled watchdog never fires
    code2 = ("import sys, time; sys.path.insert(0, %r); import bench\n"
             "t = bench.arm_sweep_watchdog(0.2, {'value': 1.0}, 0); t.cancel(); time.sleep(0.6); print('done')\n" % root)
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=25)
    assert r.returncode == 0 and r.stdout.strip() == "done"
