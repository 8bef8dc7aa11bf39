"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel gradient exchange (ddp.py): the same
GradSync / plan_buckets code that bench.py drives over RCCL on the GPUs, here with CPU tensors standing in for
the engine's flat gradient buffer.  Checks the pmean semantics of /root/reference/train.py:96."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layout():
    import savit_amd  # noqa: F401
    from savit_amd.config import ModelConfig
    from savit_amd.engine import ParamLayout

    cfg = ModelConfig(kind="vit", num_layers=6, num_heads=2, embed_dim=128, patch=8, num_classes=16, img_size=32)
    return ParamLayout(cfg)


def test_plan_buckets_cover_and_order():
    from savit_amd import ddp

    lay = _layout()
    for min_elems in (1, lay.layer_stride, 3 * lay.layer_stride, 10 ** 9):
        b = ddp.plan_buckets(lay.layer_start, lay.final_start, lay.total, min_elems)
        # contiguous cover of [0, total), produced from the END of the buffer towards the start
        assert b[0][1] == lay.total and b[-1][0] == 0 and b[-1][2] == "Wpe.wgrad"
        for (s0, e0, _), (s1, e1, _) in zip(b, b[1:]):
            assert s0 == e1 and s0 < e0
        # a bucket's trigger is the LAST backward launch that writes into it: layer j's ln1.bwd closes layer j
        for s, e, label in b[:-1]:
            j = int(label[1:label.index(".")])
            assert lay.layer_start[j] == s
        if min_elems == 10 ** 9:  # everything above layer 0 in one bucket; layer 0 + embeddings form the (exposed) last one
            assert len(b) == 2 and b[1][1] == lay.layer_start[1]
    # DeiT-B sized layout: 48 MB buckets -> a handful of buckets, each >= 48 MB except possibly the last
    from savit_amd.config import get_config
    from savit_amd.engine import ParamLayout

    big = ParamLayout(get_config("vit_b_patch16"))
    assert big.total >= 86_530_024
    bb = ddp.plan_buckets(big.layer_start, big.final_start, big.total, 48 * 2 ** 20 // 4)
    assert 3 <= len(bb) <= 8 and all((e - s) * 4 >= 48 * 2 ** 20 for s, e, _ in bb[:-2])
    assert bb[-1][1] == big.layer_start[1] and (bb[-1][1] - bb[-1][0]) * 4 < 40 * 2 ** 20  # exposed tail: embeddings + layer 0


def _worker(rank, world, port, total, buckets, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from savit_amd import ddp

        torch.manual_seed(100 + rank)
        params = torch.randn(total)
        ddp.broadcast_params(params)  # replicate (train.py:228)
        grads = torch.zeros(total)
        sync = ddp.GradSync(grads, buckets)
        hooks = sync.hooks()
        local = torch.randn(total)  # this rank's local-batch gradient
        # "backward": buckets become final from the end of the buffer; fire each hook right after filling its slice
        for s, e, label in buckets:
            grads[s:e] = local[s:e]
            hooks[label]()
        sync.wait()
        mean = grads * sync.grad_scale  # the 1/world factor the fused AdamW kernel applies
        loss = ddp.allreduce_scalar_mean(torch.tensor([float(rank + 1)]))
        # second step re-uses the object (works/launched lists must reset)
        for s, e, label in buckets:
            grads[s:e] = 1.0
            hooks[label]()
        sync.wait()
        q.put((rank, params.double().sum().item(), local.numpy(), mean.numpy(), float(loss), float(grads.mean())))
        # a skipped bucket must be reported, not silently ignored
        hooks[buckets[0][2]]()
        try:
            sync.wait()
            q.put((rank, "no-error"))
        except RuntimeError:
            for w in sync.works:
                w.wait()
            q.put((rank, "raised"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_gradsync_gloo_world2():
    from savit_amd import ddp

    lay = _layout()
    buckets = ddp.plan_buckets(lay.layer_start, lay.final_start, lay.total, 2 * lay.layer_stride)
    assert len(buckets) >= 3
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lay.total, buckets, q)) for r in range(world)]
    for p in procs:
        p.start()
    res, flags = {}, {}
    for _ in range(2 * world):
        item = q.get(timeout=120)
        if len(item) == 2:
            flags[item[0]] = item[1]
        else:
            res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0]  # parameters replicated from rank 0
    expect = (res[0][1] + res[1][1]) / 2  # pmean of the local gradients
    for r in range(world):
        assert np.allclose(res[r][2], expect, atol=1e-6)
        assert abs(res[r][3] - 1.5) < 1e-6  # psum(loss)/n
        assert abs(res[r][4] - 2.0) < 1e-6  # second step: sum of ones over 2 ranks
        assert flags[r] == "raised"


def test_bench_self_launches_its_ranks(repo_root):
    """`python bench.py --gpus 2` with no launcher around it starts two rank processes itself (the parent stays GPU-free) and rank 0
    prints one JSON line.  SAVIT_BENCH_RANK_PROBE makes the ranks stop after the rendezvous, so this runs without a GPU."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SAVIT_BENCH_RANK_PROBE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    # ... RCCL keeps its own channel choice by default (round 6, ADVICE r5: pinning NCCL_MAX/MIN_NCHANNELS to the reserve was never measured on
    # more than one GPU and may cost all-reduce bandwidth); the plan still reserves 16 CUs
    assert d == {"launcher_probe": True, "world": 2, "rank_sum": 1.0, "ipc_mode_legacy": "0", "rccl_channels": None,
                 "nccl_min_nchannels": None, "reserved_cus": 16}

    def probe(**extra):
        rr = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                            env=dict(env, **extra), capture_output=True, text=True, timeout=300)
        dd = json.loads([ln for ln in rr.stdout.splitlines() if ln.startswith("{")][0])
        return dd["rccl_channels"], dd["nccl_min_nchannels"], dd["reserved_cus"]

    # opt-in pin (SAVIT_PIN_RCCL_CHANNELS=1): the rank processes receive the bounds that keep the resident all-reduce inside the CUs
    # backward is planned to leave it (one channel = one workgroup = one CU, 16 by default); the user's own settings win, the plan follows
    assert probe(SAVIT_PIN_RCCL_CHANNELS="1") == (16, "16", 16)
    assert probe(SAVIT_PIN_RCCL_CHANNELS="1", SAVIT_RESERVED_CUS="24") == (24, "24", 24)
    assert probe(SAVIT_PIN_RCCL_CHANNELS="1", NCCL_MAX_NCHANNELS="8") == (8, "8", 8)
    # unpinned, a user-bounded RCCL is still what the plan follows - clamped below the CU count (a bound of 512 must not make the engine raise)
    assert probe(NCCL_MAX_NCHANNELS="8") == (8, None, 8)
    assert probe(NCCL_MAX_NCHANNELS="512") == (512, None, 255)
    assert probe(SAVIT_RESERVED_CUS="24") == (None, None, 24)


def test_bench_rejects_mismatched_world(repo_root):
    import subprocess
    import sys

    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "!= WORLD_SIZE" in r.stderr


def test_default_reserved_cus(monkeypatch):
    """A data-parallel rank plans its backward launches for the CUs the resident all-reduce leaves (engine reserved_cus): 0 alone, 16 of
    256 at world > 1, SAVIT_RESERVED_CUS overrides (train.py:96 / :228-230 have no counterpart: XLA plans its own collectives)."""
    from savit_amd import ddp

    monkeypatch.delenv("SAVIT_RESERVED_CUS", raising=False)
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.delenv("NCCL_MIN_NCHANNELS", raising=False)
    assert ddp.default_reserved_cus(1) == 0 and ddp.default_reserved_cus(2) == 16 and ddp.default_reserved_cus(8) == 16
    monkeypatch.setenv("SAVIT_RESERVED_CUS", "24")
    assert ddp.default_reserved_cus(1) == 24 and ddp.default_reserved_cus(8) == 24


def test_rccl_channel_bounds_follow_the_reserve():
    """ddp.rccl_channel_env: the channel bounds RCCL must run under so that its resident workgroups (one per channel, one CU each) stay
    inside `reserved_cus`; the reverse when the user bounded RCCL themselves; and bench.py's torch-free restatement of it (the
    launcher parent must not import torch) agrees on every case."""
    import bench
    from savit_amd import ddp

    PIN = {"SAVIT_PIN_RCCL_CHANNELS": "1"}
    assert ddp.rccl_channel_env(16, env={}) == {} and ddp.rccl_channel_env(16, env={"NCCL_MAX_NCHANNELS": "8"}) == {}  # opt-in only
    assert ddp.rccl_channel_env(16, env=PIN) == {"NCCL_MAX_NCHANNELS": "16", "NCCL_MIN_NCHANNELS": "16"} == ddp.rccl_channel_env(16, env={}, pin=True)
    assert ddp.rccl_channel_env(0, env=PIN) == {}
    assert ddp.rccl_channel_env(16, env=dict(PIN, NCCL_MAX_NCHANNELS="8")) == {"NCCL_MAX_NCHANNELS": "8", "NCCL_MIN_NCHANNELS": "8"}
    assert ddp.rccl_channel_env(16, env=dict(PIN, NCCL_MIN_NCHANNELS="4")) == {"NCCL_MAX_NCHANNELS": "16", "NCCL_MIN_NCHANNELS": "4"}
    assert ddp.default_reserved_cus(8, env={"NCCL_MAX_NCHANNELS": "12"}) == 12  # the plan follows the user's bound
    assert ddp.default_reserved_cus(8, env={"NCCL_MAX_NCHANNELS": "512"}) == 255 and ddp.default_reserved_cus(8, env={"NCCL_MAX_NCHANNELS": "512"}, n_cus=304) == 303
    assert ddp.default_reserved_cus(1, env={"NCCL_MAX_NCHANNELS": "12"}) == 0   # alone: nothing resident
    for env0 in ({}, {"SAVIT_RESERVED_CUS": "24"}, {"NCCL_MAX_NCHANNELS": "8"}, {"SAVIT_RESERVED_CUS": "0"}, {"NCCL_MIN_NCHANNELS": "2"},
                 {"SAVIT_RESERVED_CUS": "32", "NCCL_MAX_NCHANNELS": "20"}, {"NCCL_MAX_NCHANNELS": "512"}):
        for env in (env0, dict(env0, **PIN)):
            for world in (1, 2, 8):
                want = ddp.rccl_channel_env(ddp.default_reserved_cus(world, env=env), env=env) if world > 1 else {}
                assert bench.rccl_env_for(world, env) == want, (env, world)
    log = "host:1:1 [0] NCCL INFO Channel 00/16 : 0\nhost:1:1 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels"
    assert ddp.parse_rccl_channels(log) == 16 and ddp.parse_rccl_channels("nothing here") is None
