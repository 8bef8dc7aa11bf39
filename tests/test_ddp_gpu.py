"""Data-parallel step on the GPU engine: two ranks, each with half the batch, bucketed gradient all-reduce fired
from the backward hooks (with the weight-gradient GEMMs on their side stream), must end at the parameters a single engine
reaches on the whole batch.  Both ranks share cuda:0 and exchange through gloo (RCCL needs one GPU per rank; the hook / bucket
/ stream-join logic under test is the same)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _make(cfg_name, B):
    import dataclasses

    import savit_amd  # noqa: F401
    from savit_amd.config import get_config
    from savit_amd.model import create_model

    cfg = get_config(cfg_name, num_classes=64)
    if cfg.kind != "vit":
        cfg = dataclasses.replace(cfg, num_layers=3)  # the other families: three layers are enough for several buckets
    model = create_model(cfg_name, num_classes=64, dtype=torch.bfloat16)
    model.cfg = cfg
    eng = model.engine(B)
    eng.init_params(5)
    g = torch.Generator().manual_seed(3)
    scale = cfg.embed_dim ** -0.5 if cfg.kind != "tnt" else 0.02  # TNT has no LayerNorm before the head
    eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes, generator=g) * scale)
    eng.weights_stale = True
    return cfg, eng


def _data(cfg, B):
    g = torch.Generator(device="cuda").manual_seed(11)
    img = torch.randn(B, cfg.img_size, cfg.img_size, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, cfg.num_classes, (B,), device="cuda", generator=g, dtype=torch.int32)
    return img, lab


def _rank_main(rank, world, port, cfg_name, B, steps, out_path):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from savit_amd import ddp

    cfg, eng = _make(cfg_name, B // world)
    img, lab = _data(cfg, B)
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    ddp.broadcast_params(eng.params)
    sync = ddp.GradSync(eng.grads, ddp.plan_buckets_for(eng.layout, 1 << 18))  # small buckets: several hooks fire
    assert len(sync.buckets) > 2
    eng.bwd_hooks = sync.hooks()
    grads = []
    for _ in range(steps):
        eng.forward(img[lo:hi].contiguous())
        eng.loss_backward(lab[lo:hi].contiguous(), label_smoothing=0.1)
        sync.wait()
        torch.cuda.synchronize()
        grads.append((eng.grads * sync.grad_scale).cpu())  # the mean over ranks, as AdamW sees it
        eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0, grad_scale=sync.grad_scale)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"params": eng.params.cpu(), "grads": grads}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cfg_name,side_streams", [("vit_ti_patch16", "1"), ("vit_ti_patch16", "3"), ("vit_ti_patch16", "0"),
                                                   ("mixer_s_patch32", "1"), ("tnt_b_patch16", "1"), ("cait_xxs_24", "1")])
def test_two_rank_step_matches_single_engine(tmp_path, cfg_name, side_streams, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    # "0": every launch on one stream (the ViT engines' default); otherwise weight gradients on that many side streams
    monkeypatch.setenv("SAVIT_OVERLAP_WGRAD", "0" if side_streams == "0" else "1")
    monkeypatch.setenv("SAVIT_SIDE_STREAMS", side_streams if side_streams != "0" else "1")
    B, steps = 8, 2
    out = str(tmp_path / "p.pt")
    port = 29600 + int(side_streams) + 10 * ["vit_ti_patch16", "mixer_s_patch32", "tnt_b_patch16", "cait_xxs_24"].index(cfg_name)
    mp.spawn(_rank_main, args=(2, port, cfg_name, B, steps, out), nprocs=2, join=True)
    dp = torch.load(out)

    cfg, eng = _make(cfg_name, B)
    img, lab = _data(cfg, B)
    for k in range(steps):
        eng.forward(img)
        eng.loss_backward(lab, label_smoothing=0.1)
        torch.cuda.synchronize()
        if k == 0:
            # step 1 starts from identical parameters: the rank-averaged gradient equals the whole-batch gradient up to fp32
            # summation order (per-sample math is identical; dlogits' 1/B scaling differs by an exact power of two)
            g_ref, g_dp = eng.grads.cpu().double(), dp["grads"][0].double()
            rel = ((g_dp - g_ref).norm() / g_ref.norm()).item()
            # fp32 summation order is the only difference: ~1e-7 (printed); the bar is 5e-6.  (Round 1 ran with 1e-3 after a
            # LayerNorm-forward launch was seen perturbed when two processes time-slice one GPU; that kernel no longer contains the
            # instruction form that misbehaved - csrc/layernorm_fwd.hip - so a recurrence anywhere must fail this test.)
            print(f"[ddp {cfg_name}] rank-averaged vs whole-batch gradient: rel {rel:.2e}")
            assert rel < 5e-6, rel
        eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    torch.cuda.synchronize()
    ref = eng.params.cpu()
    # after Adam: a weight whose gradient is ~0 can move by +-lr in either direction (sign of noise), so compare in the mean
    assert (dp["params"] - ref).abs().mean().item() < 2e-5
    assert (dp["params"] - ref).abs().max().item() < 2.5e-3  # <= 2*lr + rounding over two steps


RCCL_TEST_CHANNELS = 12  # (not RCCL's default for any topology, so seeing it in the log shows the variable took effect)


def _rccl_rank_main(rank, port, out_path):
    """One rank on RCCL (backend 'nccl'): the exchange itself is the identity, what runs is RCCL's initialisation on this device,
    the broadcast, the async all-reduce of every bucket from the backward hooks on RCCL's stream, and the stream hand-offs of wait()."""
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from savit_amd import ddp

    # the channel bounds of a data-parallel rank (ddp.rccl_channel_env), exported before the first communicator exists; RCCL's own
    # log (NCCL_DEBUG=INFO, to a file) must then report no more channels than that
    for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS"):
        os.environ.pop(k, None)
    in_effect = ddp.apply_rccl_channel_env(RCCL_TEST_CHANNELS, pin=True)  # (the opt-in pin: SAVIT_PIN_RCCL_CHANNELS=1)
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_FILE=out_path + ".nccl.%p.log")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

    cfg, eng = _make("vit_ti_patch16", 8)
    img, lab = _data(cfg, 8)
    ddp.broadcast_params(eng.params)
    sync = ddp.GradSync(eng.grads, ddp.plan_buckets_for(eng.layout, 1 << 18))
    eng.bwd_hooks = sync.hooks()
    eng.forward(img)
    eng.loss_backward(lab, label_smoothing=0.1)
    sync.wait()
    torch.cuda.synchronize()
    g = eng.grads.cpu()
    eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0, grad_scale=sync.grad_scale)
    torch.cuda.synchronize()
    torch.save({"grads": g, "params": eng.params.cpu(), "nbuckets": len(sync.buckets), "backend": dist.get_backend(),
                "rccl_env": in_effect, "rccl_channels": ddp.rccl_channels_in_effect()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_single_rank(tmp_path):
    """The RCCL path of bench.py / train.py (backend 'nccl', device_id, bucket hooks) on the one GPU this box has: a 1-rank group."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    out = str(tmp_path / "r.pt")
    mp.spawn(_rccl_rank_main, args=(29711, out), nprocs=1, join=True)
    got = torch.load(out)
    assert got["backend"] == "nccl" and got["nbuckets"] > 2
    # RCCL ran under the channel bounds that keep it inside the CUs backward is planned to leave
    assert got["rccl_env"] == {"NCCL_MAX_NCHANNELS": str(RCCL_TEST_CHANNELS), "NCCL_MIN_NCHANNELS": str(RCCL_TEST_CHANNELS)}
    assert got["rccl_channels"] == RCCL_TEST_CHANNELS
    import glob

    from savit_amd import ddp

    logs = "".join(open(f, errors="replace").read() for f in glob.glob(out + ".nccl.*.log"))
    assert "NCCL_MAX_NCHANNELS" in logs, "RCCL's log does not mention the variable it was given: " + logs[-600:]
    n = ddp.parse_rccl_channels(logs)
    print(f"[rccl world 1] channels reported by RCCL: {n} (bound {RCCL_TEST_CHANNELS})")
    if n is not None:  # (a 1-rank communicator may build no rings at all; when it reports channels they must respect the bound)
        assert n <= RCCL_TEST_CHANNELS, n
    cfg, eng = _make("vit_ti_patch16", 8)
    img, lab = _data(cfg, 8)
    eng.forward(img)
    eng.loss_backward(lab, label_smoothing=0.1)
    torch.cuda.synchronize()
    g0, g1 = eng.grads.cpu().double(), got["grads"].double()
    rel = float((g0 - g1).norm() / g0.norm())
    print(f"[rccl world 1] gradient vs the plain engine: rel {rel:.2e}")
    assert rel < 5e-6  # a 1-rank all-reduce is the identity; what differs is fp32 summation order in the few atomic reductions left
    eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    torch.cuda.synchronize()
    assert float((eng.params.cpu() - got["params"]).abs().max()) < 2.5e-3  # <= 2 lr where a ~0 gradient flips sign


def test_bench_two_ranks_gloo_rehearsal_with_reserved_cus_sweep():
    """Round 6: the complete N > 1 code path of bench.py - the self-launching parent, two rank processes sharing this
    GPU over gloo (SAVIT_DIST_BACKEND=gloo: RCCL needs a GPU per rank), hooks and bucket all-reduces, the `distributed` block of the JSON
    line, and the reserved_cus 0 / 8 / 16 / 32 sweep that follows the headline measurement - on a small model so that it takes seconds.
    What it cannot show is RCCL over xGMI: no multi-GPU node was available in any round."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SAVIT_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "vit_ti_patch16",
                        "--batch", "16", "--sweep-steps", "2", "--no-cpu-baseline", "--no-other-configs"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 32 and d["value"] > 0
    dist_info = d["config"]["distributed"]
    assert dist_info["world_size"] == 2 and dist_info["backend"] == "gloo" and dist_info["buckets"] >= 1 and len(dist_info["ranks"]) == 2
    assert d["reserved_cus"] == 16 and d["config"]["engine_options"]["reserved_cus_in_effect"] == 16 and d["config"]["engine_options"]["bucket_mb"] == 48.0
    assert d["rccl_channels"] is None  # the channel pin is opt-in (SAVIT_PIN_RCCL_CHANNELS=1)
    sweep = d["reserved_cus_sweep"]
    assert [row["reserved_cus"] for row in sweep] == [0, 8, 16, 32]
    assert all(row["ms_per_step"] > 0 and row["value"] > 0 and row["allreduce_exposed_ms"] >= 0 for row in sweep)
    assert "reserved_cus_sweep_error" not in d
