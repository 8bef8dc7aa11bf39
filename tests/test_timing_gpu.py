"""The measurement utilities of SURVEY 8d (savit_timer_*, savit_spin, timing.LaunchTimer / instrumented_steps) on the GPU."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine():
    import savit_amd  # noqa: F401
    from savit_amd.config import ModelConfig
    from savit_amd.engine import ViTEngine

    cfg = ModelConfig(kind="vit", img_size=32, patch=8, embed_dim=64, num_heads=1, num_layers=2, num_classes=16)
    eng = ViTEngine(cfg, 4)
    eng.init_params(1)
    eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes) * 0.1)
    return cfg, eng


def test_timer_measures_a_gate_kernel():
    from savit_amd import lib as _lib
    from savit_amd.timing import LaunchTimer

    L = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    t = LaunchTimer(4)
    for us in (2000, 500):
        k = t.begin(f"spin{us}", s)
        assert L.savit_spin(us, s) == 0
        t.end(k, s)
    torch.cuda.synchronize()
    (l0, ms0), (l1, ms1) = t.results()
    assert (l0, l1) == ("spin2000", "spin500")
    assert 2.0 <= ms0 <= 3.0, ms0  # the spin sleeps between clock reads: it overshoots, it never undershoots
    assert 0.5 <= ms1 <= 1.0, ms1
    assert L.savit_spin(10 ** 7, s) == _lib.SAVIT_EINVAL  # bounded: a gate can never be a hang
    t.close()


def test_timer_only_brackets_tracked_labels_and_reports_drops():
    from savit_amd.timing import LaunchTimer

    s = torch.cuda.current_stream().cuda_stream
    t = LaunchTimer(2, only=["a"])
    assert t.begin("b", s) == -1
    k0 = t.begin("a", s); t.end(k0, s)
    k1 = t.begin("a", s); t.end(k1, s)
    assert (k0, k1) == (0, 1)
    assert t.begin("a", s) == -1 and t.dropped == 1
    torch.cuda.synchronize()
    assert [lbl for lbl, _ in t.results()] == ["a", "a"]
    t.close()


def test_instrumented_steps_add_up_and_leave_the_engine_as_it_was():
    from savit_amd.timing import LaunchTimer, instrumented_steps

    cfg, eng = _engine()
    img = torch.randn(4, 32, 32, 3, device="cuda").to(torch.bfloat16)
    lab = torch.randint(0, cfg.num_classes, (4,), device="cuda", dtype=torch.int32)
    eng.forward(img)
    eng.loss_backward(lab)
    g_ref = eng.grads.clone()

    def one():
        eng.forward()
        eng.loss_backward(lab)

    res = instrumented_steps(eng, one, reps=2)
    assert eng.launch_timer is None
    labels = res["labels"]
    assert "l0.qkv" in labels and "l1.fc2.dgrad" in labels and "xent" in labels and "zero.grads" in labels
    assert all(v > 0 for v in labels.values())
    for r in res["reps"]:
        assert r["gate_reached"]
        assert r["sum_ms"] <= 1.05 * r["span_ms"]  # pairs are disjoint intervals of one stream: they cannot add up to more than their span
    # the bracketed launches are the ordinary ones (this tiny geometry takes the split weight-gradient kernel with fp32 atomics: not bitwise)
    assert torch.allclose(eng.grads, g_ref, rtol=1e-4, atol=1e-6)
    # profile_step is the same thing through the engine's own method
    t = eng.profile_step(lab, reps=1)
    assert set(t) == set(labels)

    # a live timer on selected launches of ordinary steps
    live = LaunchTimer(16, only=["l0.qkv", "l1.qkv"])
    eng.launch_timer = live
    for _ in range(3):
        eng.forward()
        eng.loss_backward(lab)
        eng.optimizer_step(1e-3)
    eng.launch_timer = None
    torch.cuda.synchronize()
    got = live.results()
    assert [lbl for lbl, _ in got] == ["l0.qkv", "l1.qkv"] * 3
    assert all(0 < ms < 5 for _, ms in got)
    live.close()


def test_zero_bytes_and_hold_cus():
    from savit_amd import lib as _lib

    L = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    x = torch.ones(1000, device="cuda")
    assert L.savit_zero_bytes(x.data_ptr() + 4 * 10, 4 * 100, s) == 0
    torch.cuda.synchronize()
    assert float(x.sum()) == 900.0 and float(x[10:110].abs().sum()) == 0.0
    assert L.savit_hold_cus(8, 200, s) == 0
    torch.cuda.synchronize()
    assert L.savit_hold_cus(0, 200, s) == _lib.SAVIT_EINVAL
