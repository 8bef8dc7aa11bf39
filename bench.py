#!/usr/bin/env python3
"""Headline benchmark: DeiT-B/16 224^2 bf16 TRAIN STEP throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one synthetic batch already resident in HBM:
  forward (patch-embed, 12 encoder blocks, head) + label-smoothed CE + full backward + gradient all-reduce
  (RCCL, bucketed, overlapped with backward; N > 1) + fused AdamW + bf16 operand refresh.
Per-GPU batch is fixed at 128 images (BASELINE config 3: 1024 on 8 GPUs), i.e. weak scaling; `value` is the
whole-job images/s.  Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      the dominant GEMM kernel (by total time per step; picked from an instrumented forward+backward after warm-up):
                achieved = algorithmic FLOPs per launch / mean launch duration, measured live INSIDE the timed region with HIP
                timing events (hipEventDisableSystemFence) around every launch of that kernel on its launch stream;
                peak = dense bf16 MFMA peak of MI355X.  `traffic` comes from the committed PMC passes (separate rocprofv3 runs;
                profiles/ and DESIGN.md).
  kernel_breakdown_ms / roofline_valid   three instrumented whole steps after the timed region (every launch bracketed, each step
                enqueued behind a gate kernel so that no event pair contains host time, per-label minimum), and the checks
                that the accounting adds up: launches x avg of the dominant kernel <= ms_per_step, sum of all launches <= 1.05 x
                ms_per_step (net of what an empty bracket costs, measured), live and instrumented averages of the dominant kernel within 10 %.  A failed check sets
                "roofline_valid": false and dumps the per-label table to stderr.
  step_roofline whole-step MFMA utilisation: images/s/GPU x the flops the step EXECUTES (98.55 GFLOP per image on DeiT-B: the last
                encoder layer runs on the cls rows) / peak = `frac`; SURVEY.md 8d's dense count (105.152 GFLOP) gives `frac_counted`.
  hbm_kernels   the memory-bound kernels of the step (LayerNorm forward / backward, AdamW, attention forward / backward): algorithmic
                bytes per launch, mean launch time from the instrumented steps, achieved TB/s and the fraction of the 6.3 TB/s this
                chip sustains (north_star: "achieved HBM GB/s on the memory-bound softmax / LayerNorm").
  cpu_baseline  the CPU restatement (oracle/torch_ref.py, JAX absent: SURVEY 8c) of BASELINE config 1 (ViT-Ti/16, batch 8,
                fp32: forward+loss+backward = `value`, forward+loss alone = `forward_loss_value`) on this host's cores, bounded
                sample, rank 0 at N=1 only; `headline_model_value` is the same restatement of the headline model.
  other_configs (N=1, headline workload only) a short timing (10 steps after 3 warm-up steps) of BASELINE configs 2, 4 and 5 on
                the same GPU, each with images_per_gpu, ms_per_step, step_roofline.frac and its own dominant-kernel roofline.

`python bench.py --gpus N` with N > 1 and no launcher around it starts the N ranks itself: the parent makes no GPU call (it does
not even import torch), runs `python -m torch.distributed.run --nproc-per-node N ... bench.py <same flags>` as a child process and
returns its exit code; rank 0 of the children prints the JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC between the ranks' RCCL peers: the only mode this host driver supports

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2516.6  # 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md chip table)


TN_TILES = {6: "gemm_tn_ring_kernel<128,128,2,2,4,%d,false>",
            12: "gemm_tn_pair_kernel<128,128,2,2,2,%d>", 13: "gemm_tn_pair_kernel<256,256,2,4,2,%d>",
            17: "gemm_tn_pair_kernel<192,128,2,2,2,%d>", 18: "gemm_tn_pair_tail_kernel<192,128,128,2,2,2,%d>",
            20: "gemm_tn_pp_kernel<%d,0>", 21: "gemm_tn_pp320_kernel<%d>", 22: "gemm_tn_pp320p_kernel<%d>", 24: "gemm_tn_rows_kernel<%d>"}
WG_VARIANTS = {1: "gemm_wgrad_ring_kernel<128,128,2,2,4,false>", 3: "gemm_wgrad_ring_kernel<256,256,2,4,4,false>"}
EPI_OF = {"qkv": 0, "proj": 2, "fc1": 1, "fc2": 2, "fc2.dgrad": 3, "fc1.dgrad": 0, "proj.dgrad": 0, "qkv.dgrad": 0}


def kernel_symbol(label: str, lib, M: int, d: int, F: int) -> str:
    """Launch label of the engine's plan -> the kernel symbol rocprofv3 reports (template arguments as in csrc/)."""
    parts = label.split(".")
    if label.startswith("wgrad.group"):
        # engine.wgrad_group_tile: 256 x 384 / 384 x 256 tiles for the d = 384 models, else 256 x 256 (edge tiles hang over the matrix)
        return "gemm_wgrad_group_mixed_kernel<3>" if (d % 256 or F % 256) and d % 384 == 0 and F % 384 == 0 else "gemm_wgrad_group_kernel<256,256,2,4,3,32>"
    op = ".".join(parts[1:]) if parts[0].startswith("l") and parts[0][1:].isdigit() else label
    shapes = {"qkv": (3 * d, d), "proj": (d, d), "fc1": (F, d), "fc2": (d, F), "fc2.dgrad": (F, d), "fc1.dgrad": (d, F),
              "proj.dgrad": (d, d), "qkv.dgrad": (d, 3 * d)}
    if op in shapes:
        N, K = shapes[op]
        return tn_symbol(lib, M, N, K, EPI_OF[op])
    wshapes = {"Wqkv.wgrad": (d, 3 * d), "Wo.wgrad": (d, d), "W1.wgrad": (d, F), "W2.wgrad": (F, d)}
    if op in wshapes:
        return WG_VARIANTS[lib.savit_gemm_wgrad_auto_variant(wshapes[op][0], wshapes[op][1], 0)]
    if op.endswith(".wgrad.reduce"):
        return "wgrad_reduce_kernel"
    if "attn" in op:
        return "attn_bwd_persl_kernel" if op.endswith(".bwd") else "attn_fwdl_kernel"  # the N <= 224, head_dim 64 kernels with a loader wave (csrc/attention.hip, round 6)
    if op.startswith("ln"):
        return "ln_bwd_kernel(+finalize)" if op.endswith(".bwd") else "ln_fwd_kernel"
    return "other(" + op + ")"


WG_GROUP_TILES = {256: "gemm_wgrad_group_kernel<256,256,2,4,3,32>", 640: "gemm_wgrad_group_mixed_kernel<3>",
                  384: "gemm_wgrad_group_kernel<128,384,2,4,3,32>", 128: "gemm_wgrad_group_kernel<128,128,2,2,4,32>"}


def tn_symbol(lib, M: int, N: int, K: int, epilogue: int, tile: int = 0, cu_budget: int = 0) -> str:
    """The kernel symbol savit_gemm_bf16_tn launches for this problem (tile 0 = the library's own choice for cu_budget CUs)."""
    if tile == 0:
        tile = lib.savit_gemm_tn_auto_tile_cus(M, N, K, epilogue, cu_budget)
    if tile == 24:  # the few-rows kernel: 32 x 32 outputs per wave where that leaves waves for every CU, else 16 x 16 (launch_rows)
        return "gemm_tn_rows_kernel<%d,%d>" % (epilogue, 2 if -(-M // 16) * -(-N // 16) >= 1536 else 1)
    return TN_TILES.get(tile, "gemm_tn tile %d <%%d>" % tile) % epilogue


def plan_symbols(eng) -> dict:
    """{launch label: kernel symbol} read off the engine's recorded launch plans (any family): every TN GEMM from the argument
    block it will be launched with, every grouped weight-gradient launch from its tile code.  Labels of other launches are absent
    (kernel_symbol / kernel_class name them)."""
    out = {}
    lib = getattr(eng, "L", None)
    flops = eng.__dict__.setdefault("plan_flops", {}) if hasattr(eng, "__dict__") else {}
    for plan in (getattr(eng, "_fwd_plan", None), getattr(eng, "_bwd_plan", None), getattr(eng, "_bwd_plan_serial", None)):
        for fn, args, label in (plan.calls if plan is not None else ()):
            name = getattr(fn, "__name__", "")
            if name == "savit_gemm_bf16_tn":
                a = args[0]._obj
                out[label] = tn_symbol(lib, a.M, a.N, a.K, a.epilogue, a.tile, a.cu_budget)
                flops[label] = 2.0 * a.M * a.N * a.K  # algorithmic (operands padded with zeros - head, Mixer tokens - count as stored)
            elif name in ("savit_gemm_bf16_wgrad_grouped", "savit_gemm_bf16_wgrad_grouped_ex"):
                out[label] = WG_GROUP_TILES.get(int(args[2]), "gemm_wgrad_group tile %d" % int(args[2]))
    return out


def kernel_class(label: str) -> str:
    label = label.split("#")[0]
    if label.startswith("zero."):
        return "memset"
    if label in ("sumsq", "adamw") or label.startswith("cast "):
        return "optimizer"
    if label.startswith("wgrad.group"):
        return "gemm_wgrad"
    if label.endswith(".wgrad.reduce"):
        return "wgrad_reduce"
    if label.endswith(".wgrad") or label == "Wpe.wgrad":
        return "gemm_wgrad"
    if label.endswith(".dgrad") or label.split(".")[-1] in ("qkv", "proj", "fc1", "fc2", "q", "kv", "iqkv", "iproj", "ifc1", "ifc2", "fc") \
            or label in ("patch_embed", "pixel_embed", "head"):
        return "gemm_tn"
    if "attn" in label:
        return "attention_bwd" if label.endswith(".bwd") else "attention_fwd"
    if ".ls" in label:
        return "layerscale_bwd"
    if ".tok.T" in label or ".tok.dT" in label:
        return "token_transpose"
    if "ln" in label:
        return "layernorm_bwd" if label.endswith(".bwd") else "layernorm_fwd"
    return "other"


OTHER_CONFIGS = (("vit_s_patch16", 256, 224, "2. DeiT-S/16 224^2, 256 img"), ("cait_s_24", 256, 224, "4. CaiT-S24 224^2, 256 img"),
                 ("vit_l_patch16", 256, 384, "5. ViT-L/16 384^2, 256 img/GPU"))


def launch_ranks(n: int) -> int:
    """Parent of a multi-rank run: start one child process per rank through torch.distributed.run and return its exit code.
    Nothing here touches the GPU (no torch import): a process that has initialised HIP must never fork / exec rank processes."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"  # dmabuf IPC between the ranks' RCCL peers (the only mode this host driver supports)
    env.setdefault("OMP_NUM_THREADS", "4")   # torch.distributed.run would set 1; the CPU side of a rank only drives launches
    env.update(rccl_env_for(n, env))         # (opt-in, SAVIT_PIN_RCCL_CHANNELS=1) RCCL's channels bounded by the CUs backward leaves it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rccl_env_for(world: int, env) -> dict:
    """NCCL_MAX/MIN_NCHANNELS for the rank processes of a `world`-rank job - only with SAVIT_PIN_RCCL_CHANNELS=1 (ddp.rccl_channel_env:
    opt-in, unverified on more than one GPU; restated here without importing torch: the launcher parent must stay GPU-free).
    tests/test_ddp_cpu.py checks both against each other."""
    if world <= 1 or env.get("SAVIT_PIN_RCCL_CHANNELS", "0") != "1":
        return {}
    if env.get("SAVIT_RESERVED_CUS"):
        reserved = int(env["SAVIT_RESERVED_CUS"])
    elif env.get("NCCL_MAX_NCHANNELS"):
        reserved = max(0, min(int(env["NCCL_MAX_NCHANNELS"]), 255))
    else:
        reserved = 16
    if reserved <= 0:
        return {}
    mx = env.get("NCCL_MAX_NCHANNELS") or str(reserved)
    return {"NCCL_MAX_NCHANNELS": mx, "NCCL_MIN_NCHANNELS": env.get("NCCL_MIN_NCHANNELS") or str(min(int(mx), reserved))}


def rank_probe():
    """SAVIT_BENCH_RANK_PROBE=1: the ranks only rendezvous (gloo, CPU) and rank 0 prints what it saw - tests/test_ddp_cpu.py uses
    this to cover the launcher without a GPU."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    t = torch.tensor([float(dist.get_rank())])
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        from savit_amd import ddp

        print(json.dumps({"launcher_probe": True, "world": dist.get_world_size(), "rank_sum": t.item(),
                          "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                          "rccl_channels": ddp.rccl_channels_in_effect(), "nccl_min_nchannels": os.environ.get("NCCL_MIN_NCHANNELS"),
                          "reserved_cus": ddp.default_reserved_cus(dist.get_world_size())}))
    dist.barrier()
    dist.destroy_process_group()


def build_engine(cfg, B, reserved_cus=None, **opts):
    """opts: options.EngineOptions fields (wgrad_max_lag, overlap_wgrad, cls_only_last, ...); None = the environment's / the default."""
    from savit_amd.engine import ViTEngine

    if cfg.kind == "cait":
        from savit_amd.cait_engine import CaiTEngine

        return CaiTEngine(cfg, B, reserved_cus=reserved_cus, **opts)
    if cfg.kind == "mixer":
        from savit_amd.mixer_engine import MixerEngine

        return MixerEngine(cfg, B, reserved_cus=reserved_cus, **opts)
    if cfg.kind == "tnt":
        from savit_amd.tnt_engine import TNTEngine

        return TNTEngine(cfg, B, reserved_cus=reserved_cus, **opts)
    return ViTEngine(cfg, B, reserved_cus=reserved_cus, **opts)


def engine_plan_facts(eng, bucket_mb=None) -> dict:
    """What the launch plan of this engine was built with - printed in the JSON line (`config.engine_options`) so that a measured
    number names the plan it ran: every EngineOptions field, plus the quantities derived from them."""
    out = dict(eng.opt.as_dict())
    out.update({"overlap_wgrad_in_effect": bool(getattr(eng, "overlap_wgrad", False)), "reserved_cus_in_effect": int(getattr(eng, "reserved_cus", 0)),
                "cu_budget": int(getattr(eng, "cu_budget", 0)), "wgrad_tile": int(getattr(eng, "wgrad_tile", 0) or 0),
                "wgrad_lag_layers": int(getattr(eng, "wgrad_lag", 0) or 0), "cls_only_last_in_effect": bool(getattr(eng, "cls_only_last", False)),
                "cls_fwd_in_effect": bool(getattr(eng, "cls_fwd", False))})
    if bucket_mb is not None:
        out["bucket_mb"] = bucket_mb
    return out


SWEEP_RESERVED = (0, 8, 16, 32)


def reserved_cus_sweep(cfg, B, world, rank, dist, bucket_mb, steps, warmup, opts) -> list:
    """N > 1 only (every rank takes part): the train step timed with backward planned for 256 - r CUs, r in SWEEP_RESERVED, in ONE job -
    same process group, same RCCL communicator, a fresh engine per point - so that the first lease of a multi-GPU node yields the
    scaling number AND the tuning of the one planning constant nobody could measure (`reserved_cus = 16` was chosen
    against a stand-in, profiles/r04_cu_thief.log).  -> [{reserved_cus, ms_per_step, value, allreduce_exposed_ms}] (rank 0; max over ranks)."""
    import torch

    from savit_amd import ddp

    rows = []
    for r in SWEEP_RESERVED:
        eng = build_engine(cfg, B, reserved_cus=r, **opts)
        init_bench_params(eng, cfg)
        ddp.broadcast_params(eng.params)
        sync = ddp.GradSync(eng.grads, ddp.plan_buckets_for(eng.layout, int(bucket_mb * 2 ** 20 / 4)))
        eng.bwd_hooks = sync.hooks()
        eng.refresh_weights()
        step, _ = make_step(eng, cfg, B, world, rank, sync)
        for i in range(warmup):
            step(i)
        del step.exposed[:]
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0, sum(a.elapsed_time(b) for a, b in step.exposed) / max(1, len(step.exposed))],
                         device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, ex = float(t[0].item()), float(t[1].item())
        rows.append({"reserved_cus": r, "steps": steps, "warmup": warmup, "ms_per_step": round(el / steps * 1e3, 3),
                     "value": round(B * world * steps / el, 1), "allreduce_exposed_ms": round(ex, 4)})
        del step, sync, eng
        torch.cuda.empty_cache()
    return rows


def arm_sweep_watchdog(seconds: float, out: dict, rank: int):
    """The headline is measured before the sweep starts; a sweep that hangs (a collective one rank never enters) must not lose it.  After
    `seconds` every rank leaves the process; rank 0 prints the line first, with the reason.  -> the timer (cancel() it when the sweep is done)."""
    import threading

    def fire():
        if rank == 0:
            line = dict(out, reserved_cus_sweep_error=f"watchdog: the sweep was still running after {seconds:g} s; headline printed without it")
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
        os._exit(0)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def make_step(eng, cfg, B, world, rank, sync):
    """Engine + synthetic batches resident in HBM -> step(i): N(0,1) NHWC images cast to bf16 (train.py:81), uniform labels,
    seed 42+rank; lr*bs/512 (train.py:171,214-220), wd 1e-4 (:172-176), label smoothing 0.1, clip 1.0."""
    import torch

    gd = torch.Generator(device="cuda").manual_seed(42 + rank)
    S = cfg.img_size
    batches = [(torch.randn(B, S, S, 3, device="cuda", generator=gd).to(torch.bfloat16),
                torch.randint(0, cfg.num_classes, (B,), device="cuda", generator=gd, dtype=torch.int32)) for _ in range(2)]
    lr, wd = 5e-4 * (B * world) / 512.0, 1e-4

    if cfg.kind == "cait":
        eng.gen.manual_seed(42 + rank)  # per-rank stochastic-depth masks

    exposed = []  # per step: (event after the last backward launch, event after sync.wait() returned) on the compute stream

    def step(i):
        img, lab = batches[i & 1]
        if cfg.kind == "cait":
            eng.forward(img, is_training=True)  # stochastic depth active (cait.py:38,49)
        else:
            eng.forward(img)
        eng.loss_backward(lab, label_smoothing=0.1)
        if sync is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()   # behind the last backward kernel
            sync.wait()   # the compute stream waits here for the bucket all-reduces still in flight
            e1.record()
            exposed.append((e0, e1))
        eng.optimizer_step(lr=lr, weight_decay=wd, max_norm=1.0, grad_scale=(sync.grad_scale if sync else 1.0))

    step.exposed = exposed
    return step, batches


def rank_identity(local_rank: int) -> dict:
    """What this rank actually runs on - gathered over all ranks into the N > 1 JSON line, so that the line itself shows that N
    different GPUs took part."""
    import torch

    p = torch.cuda.get_device_properties(local_rank)
    ident = {"rank": int(os.environ.get("RANK", "0")), "local_rank": local_rank, "device": torch.cuda.current_device(), "name": p.name,
             "pci_bus_id": None, "uuid": None, "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))}
    try:
        ident["pci_bus_id"] = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    except AttributeError:
        pass
    try:
        ident["uuid"] = str(p.uuid)
    except AttributeError:
        pass
    return ident


def init_bench_params(eng, cfg):
    import torch

    eng.init_params(seed=42)  # train.py:187-189 default seed
    # the reference zero-initialises the head kernel (vit.py:98); a zero operand would make the first backward
    # passes run on zeros (higher clocks), so the benchmark gives the head a lecun-normal kernel.
    g = torch.Generator().manual_seed(7)
    eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes, generator=g) * cfg.embed_dim ** -0.5)


def symbol_tables(times, cfg, eng, B):
    """{launch label: ms} -> per kernel symbol (what rocprofv3 --kernel-trace --stats lists): total ms, launches, algorithmic flops,
    the labels behind it; and the same per kernel class."""
    d, F, M = cfg.embed_dim, cfg.hidden, eng.M
    gemm_flops = {"qkv": 2.0 * M * d * 3 * d, "proj": 2.0 * M * d * d, "fc1": 2.0 * M * d * F, "fc2": 2.0 * M * d * F,
                  "tok": 2.0 * B * d * cfg.n_patches * cfg.tokens_hidden}  # MLP-Mixer token-mixing GEMMs (unpadded, algorithmic)
    names = {"Wqkv": "qkv", "Wo": "proj", "W1": "fc1", "W2": "fc2", "tW1": "tok", "tW2": "tok"}
    sym, cls = {}, {}
    psym = plan_symbols(eng)
    for key, t_ms in times.items():
        label = key.split("#")[0]
        c = kernel_class(label)
        parts = label.split(".")
        fl = 0.0
        if c in ("gemm_tn", "gemm_wgrad") and len(parts) >= 2 and parts[0].startswith("l") and parts[0][1:].isdigit():
            fl = gemm_flops.get(names.get(parts[1], parts[1]), 0.0)
        elif label.startswith("wgrad.group"):
            fl = getattr(eng, "group_flops", {}).get(label, 0.0)
        elif label in ("patch_embed", "Wpe.wgrad"):
            fl = 2.0 * B * cfg.n_patches * cfg.patch_dim * d
        elif label.startswith("head"):
            fl = 2.0 * B * d * cfg.num_classes
        elif c == "gemm_tn" and cfg.kind != "vit":
            fl = getattr(eng, "plan_flops", {}).get(label, 0.0)  # class-attention layers, TNT's inner stream, ...: 2 M N K of the launch
        sname = psym.get(label) or (kernel_symbol(label, eng.L, M, d, F) if cfg.kind == "vit" else c)
        for table, k in ((sym, sname), (cls, c)):
            e = table.setdefault(k, {"ms": 0.0, "n": 0, "flops": 0.0, "labels": []})
            e["ms"] += t_ms
            e["n"] += 1
            e["flops"] += fl
            e["labels"].append(label)
    return sym, cls


def traffic_lookup(kernel: str, headline: bool, profiles_dir: str = None, running: dict = None) -> dict:
    """`roofline.traffic` for `kernel`: HBM-side bytes per launch from the newest committed rocprofv3 --pmc passes of the headline
    workload (tools/pmc_summary.py -> profiles/rNN_pmc_traffic.json; a counter pass cannot run inside this process) - but only while
    that file was measured on THIS code: it carries the SHA-256 of each kernel's machine code (fingerprint.py), compared here with the
    library that is running.  A mismatch (kernel edited, renamed, or a file from before round 5) yields "traffic": null and
    "traffic_stale": true with the reason, never another build's number."""
    from savit_amd import fingerprint as fpr

    res = {"traffic": None}
    if not headline:
        return res
    pdir = profiles_dir or os.path.join(ROOT, "profiles")
    try:
        files = sorted((f for f in os.listdir(pdir) if f.startswith("r") and f.endswith("_pmc_traffic.json") and f[1:3].isdigit()), reverse=True)
        if not files:
            return res
        src = files[0]
        pj = json.load(open(os.path.join(pdir, src)))
        pm = pj["kernels"].get(kernel)
        stale, why = fpr.traffic_is_stale(pj, kernel, running)
        res["traffic_source"] = f"profiles/{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, FETCH doubled)"
        res["traffic_stale"] = bool(stale or pm is None)
        res["traffic_check"] = why
        if pm and not stale:
            res["traffic"] = pm["traffic_bytes"]
    except (OSError, ValueError, KeyError) as e:
        res["traffic_stale"] = True
        res["traffic_check"] = f"could not read the PMC summary: {e}"
    return res


HBM_ACHIEVABLE_TBS = 6.3  # MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s sustained by a streaming kernel


def step_roofline(cfg, per_gpu: float, cls_only_last: bool, cls_fwd: bool) -> dict:
    """Whole-step MFMA figure.  `frac` / `achieved` use the flops the step EXECUTES (the ViT engines run the last
    encoder layer on the cls rows where only those are read - exact, engine.py - so the dense count overstates what the matrix pipes
    did); `frac_counted` / `achieved_counted` keep SURVEY 8d's dense count (the north-star's "% of MFMA peak" as BASELINE.md prices it:
    40 % = 9 573 img/s on DeiT-B).  Equal for every family but ViT."""
    from savit_amd.config import executed_flops_per_image, train_flops_per_image

    fpi = train_flops_per_image(cfg)
    ex = executed_flops_per_image(cfg, cls_only_last, cls_fwd)
    out = {"bound": "mfma", "achieved": round(per_gpu * ex / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(per_gpu * ex / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "executed_flops_per_image": ex,
           "achieved_counted": round(per_gpu * fpi / 1e12, 2), "frac_counted": round(per_gpu * fpi / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
           "flops_per_image": fpi}
    if ex != fpi:
        out["executed_note"] = ("last encoder layer behind its qkv projection on the cls rows only (vit.py:57,95: row 0 alone reaches the "
                                "head); frac = executed flops, frac_counted = SURVEY 8d's dense count")
    return out


def hbm_kernels(labels_ms: dict, cfg, M: int, B: int, n_params: int, cls_only_last: bool, cls_fwd: bool) -> dict:
    """The memory-bound kernels of a ViT step from the per-label times of the instrumented steps ({label: ms}): per kernel the
    ALGORITHMIC bytes of one dense launch (what the kernel must read and write once), launches, mean time, TB/s and
    the fraction of the sustained HBM rate.  Only launches over all M = B * N token rows count (the cls-row launches of the last layer
    and the final LayerNorm are latency-shaped, not bandwidth-shaped)."""
    d, N, H, NL = cfg.embed_dim, cfg.seq_len, cfg.num_heads, cfg.num_layers
    per = {
        "ln_fwd": (4 + 2) * M * d + 8 * M,                 # fp32 residual in, bf16 out, mean / rstd out
        "ln_bwd": (2 + 4 + 4 + 4 + 2) * M * d + 8 * M,     # dy bf16, x fp32, residual gradient in / out fp32, bf16 copy out, statistics
        "attn_fwd": (6 + 2) * M * d + 4 * B * H * N,       # packed q|k|v in, o out, LSE out
        "attn_bwd": (6 + 2 + 2 + 6) * M * d + 4 * B * H * N,  # q|k|v, o, do in; dq|dk|dv out; LSE in
        "adamw": 30 * n_params,                            # p, g, m, v in; p, m, v out (fp32) + the bf16 operand mirror
    }
    last = f"l{NL - 1}."
    acc = {k: [0.0, 0] for k in per}
    for key, ms in labels_ms.items():
        label = key.split("#")[0]
        op = label.split(".", 1)[1] if label.startswith("l") and "." in label and label.split(".")[0][1:].isdigit() else label
        skip_last_fwd = cls_only_last and cls_fwd and label.startswith(last)
        if op in ("ln1", "ln2") and not (skip_last_fwd and op == "ln2"):
            k = "ln_fwd"
        elif op in ("ln1.bwd", "ln2.bwd") and not (cls_only_last and label.startswith(last) and op == "ln2.bwd"):
            k = "ln_bwd"
        elif op == "attn" and not skip_last_fwd:
            k = "attn_fwd"
        elif op == "attn.bwd" and not skip_last_fwd:
            k = "attn_bwd"
        elif label == "adamw":
            k = "adamw"
        else:
            continue
        acc[k][0] += ms
        acc[k][1] += 1
    out = {}
    for k, (ms, n) in acc.items():
        if n == 0 or ms <= 0:
            continue
        avg_s = ms / n * 1e-3
        tbs = per[k] / avg_s / 1e12
        out[k] = {"algorithmic_bytes": int(per[k]), "launches": n, "avg_us": round(avg_s * 1e6, 1), "TB/s": round(tbs, 2),
                  "frac_of_6.3": round(tbs / HBM_ACHIEVABLE_TBS, 3)}
    return out


def pick_dominant(sym, pair_overhead_ms: float = 0.0):
    """The GEMM kernel symbol with the largest total time; totals within 5 % of the largest (run-to-run noise: the grouped
    weight-gradient kernel and the plain-epilogue 320x256 kernel are both ~20 % of the DeiT-B step) are broken towards the kernel with
    more flops per launch, so that the reported kernel does not flip between runs.  Totals are taken net of what an EMPTY bracket costs
    (pair_overhead_ms per launch: ~1 us, several under rocprofv3, where it made the 46-launch kernel overtake the 5-launch one and the
    profiled run reported another kernel than the un-profiled one - profiles/r05_serial_bench.json)."""
    gemm = [k for k in sym if k.startswith("gemm") and sym[k]["flops"] > 0]
    net = {k: sym[k]["ms"] - pair_overhead_ms * sym[k]["n"] for k in gemm}
    top = max(net.values())
    return max((k for k in gemm if net[k] >= 0.95 * top), key=lambda k: sym[k]["flops"] / sym[k]["n"])


def measure(eng, cfg, B, world, rank, sync, steps, warmup, dist=None, breakdown=True):
    """warm-up, [instrumented forward+backward: which kernel dominates], the TIMED region with that kernel's launches bracketed by
    timing events on the launch stream, [instrumented whole steps: per-class breakdown + consistency checks].
    -> (elapsed seconds of the timed region (max over ranks), step fn, dict for the JSON line)"""
    import torch

    from savit_amd.timing import LaunchTimer, instrumented_steps

    step, batches = make_step(eng, cfg, B, world, rank, sync)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()

    # ---- which GEMM kernel dominates this step?  (every rank, no collective, no parameter update: forward + loss + backward only)
    saved_hooks, eng.bwd_hooks = eng.bwd_hooks, {}
    lab0 = batches[0][1]
    eng.set_images(batches[0][0])  # the instrumented passes run forward() on the resident image buffer: batch 0's images WITH batch 0's labels

    def fwd_bwd():
        if cfg.kind == "cait":
            eng.forward(is_training=True)
        else:
            eng.forward()
        eng.loss_backward(lab0, label_smoothing=0.1)

    pre = instrumented_steps(eng, fwd_bwd, reps=2)
    eng.bwd_hooks = saved_hooks
    sym0, _ = symbol_tables(pre["labels"], cfg, eng, B)
    dom = pick_dominant(sym0, pre["pair_overhead_ms"])
    dom_labels = sorted(set(sym0[dom]["labels"]))
    live = LaunchTimer(steps * len(dom_labels) + 8, only=dom_labels)

    # ---- the timed region
    del step.exposed[:]
    eng.launch_timer = live
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.launch_timer = None
    final_loss = float(eng.loss.item())  # the loss of the LAST TIMED step, read before any instrumented pass touches the engine
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_step = elapsed / steps * 1e3
    info = {"final_loss": round(final_loss, 4)}
    if rank != 0:
        live.close()
        return elapsed, step, info

    # ---- roofline of the dominant kernel: algorithmic flops per launch / mean launch duration over the timed region
    lt = live.results()
    live.close()
    n_l = len(lt)
    tot_ms = sum(ms for _, ms in lt)
    per_label_fl = {}
    for lbl in dom_labels:
        per_label_fl[lbl] = symbol_tables({lbl: 0.0}, cfg, eng, B)[0][dom]["flops"]
    fl_tot = sum(per_label_fl[lbl] for lbl, _ in lt)
    avg_ms = tot_ms / max(1, n_l)
    ach = fl_tot / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    launches_per_step = n_l / steps
    roof = {"bound": "mfma", "kernel": dom, "launches_per_step": round(launches_per_step, 2), "launches_timed": n_l,
            "avg_launch_ms": round(avg_ms, 4), "min_launch_ms": round(min(ms for _, ms in lt), 4), "max_launch_ms": round(max(ms for _, ms in lt), 4),
            "share_of_step": round(launches_per_step * avg_ms / ms_step, 3), "flops_per_launch": fl_tot / max(1, n_l),
            "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
            "traffic": None,
            "how": "HIP timing events (hipEventDisableSystemFence) around every launch of this kernel on its launch stream, over all timed steps"}
    checks = {"dominant_fits_in_step": launches_per_step * avg_ms <= ms_step, "no_events_dropped": live.dropped == 0}
    info["roofline"] = roof

    # ---- per-class breakdown: whole steps (incl. optimizer + operand refresh), every launch bracketed, behind a gate kernel
    if breakdown:
        saved_hooks, eng.bwd_hooks = eng.bwd_hooks, {}
        lr, wd = 5e-4 * (B * world) / 512.0, 1e-4

        def whole():
            fwd_bwd()
            eng.optimizer_step(lr=lr, weight_decay=wd, max_norm=1.0)

        # the instrumented steps really update the parameters (5 optimizer steps on one batch): the training state is put back after
        eng.set_images(batches[0][0])
        snap = {k: getattr(eng, k).clone() for k in ("params", "adam_m", "adam_v", "params_bf16") if getattr(eng, k, None) is not None}
        snap_step = eng.step_count
        post = instrumented_steps(eng, whole, reps=3)
        for k, v in snap.items():
            getattr(eng, k).copy_(v)
        eng.step_count = snap_step
        del snap
        eng._mirror_fresh = False
        eng.refresh_weights()
        eng.bwd_hooks = saved_hooks
        sym1, cls1 = symbol_tables(post["labels"], cfg, eng, B)
        total = sum(v["ms"] for v in cls1.values())
        serial = not getattr(eng, "overlap_wgrad", False)  # engines with side streams run their serial plan when instrumented
        best_span = min(r["span_ms"] for r in post["reps"])
        # every bracket contains its own marker packets on top of its kernel (measured with empty brackets: ~1 us, several under
        # rocprofv3): the sum is checked net of that
        net = total - post["launches"] * post["pair_overhead_ms"]
        # under rocprofv3 (its tool library is preloaded) every dispatch carries the profiler's own packets: 1.10 there
        profiled = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))
        lim = 1.10 if profiled else 1.05
        checks["sum_le_1p05_step"] = (net <= lim * ms_step) if serial else (net <= lim * best_span)
        if profiled:
            info["profiler_attached"] = True
        checks["gate_reached"] = all(r["gate_reached"] for r in post["reps"])
        if dom in sym1:
            inst_avg = sym1[dom]["ms"] / sym1[dom]["n"]
            roof["instrumented_avg_launch_ms"] = round(inst_avg, 4)
            if world == 1:  # (with an all-reduce resident in the timed region the live figure is SUPPOSED to differ from the instrumented, exchange-free one)
                checks["live_vs_instrumented_within_10pct"] = abs(inst_avg - avg_ms) <= 0.10 * avg_ms
        kb = {c: round(v["ms"], 3) for c, v in sorted(cls1.items(), key=lambda kv: -kv[1]["ms"])}
        opt_ms = cls1.get("optimizer", {"ms": 0.0})["ms"] + sum(ms for k, ms in post["labels"].items() if k.split("#")[0] == "zero.gnorm")
        kb["sum_fwd_bwd"] = round(total - opt_ms, 3)
        kb["sum_step"] = round(total, 3)
        kb["sum_step_net_of_brackets"] = round(net, 3)
        info["bracket_overhead_us"] = round(post["pair_overhead_ms"] * 1e3, 2)
        info["launches_per_train_step"] = int(post["launches"])  # forward + loss + backward + optimizer + operand refresh, serial plan
        info["kernel_breakdown_ms"] = kb
        info["kernel_breakdown_how"] = (f"{len(post['reps'])} instrumented steps, every launch bracketed, each step enqueued behind a "
                                        f"{post['gate_us']} us gate kernel, per-label minimum; span of one instrumented step {best_span:.3f} ms")
        info["gemm_class_tflops"] = {c: round(cls1[c]["flops"] / (cls1[c]["ms"] * 1e-3) / 1e12, 2) for c in ("gemm_tn", "gemm_wgrad") if c in cls1 and cls1[c]["ms"] > 0}
        info["top_kernels_ms"] = {k: {"total_ms": round(v["ms"], 3), "launches": v["n"], "avg_us": round(v["ms"] / v["n"] * 1e3, 1)}
                                  for k, v in sorted(sym1.items(), key=lambda kv: -kv[1]["ms"])[:8]}
        if cfg.kind == "vit":
            info["hbm_kernels"] = hbm_kernels(post["labels"], cfg, eng.M, B, int(eng.params.numel()), bool(getattr(eng, "cls_only_last", False)),
                                              bool(getattr(eng, "cls_fwd", False)))
        info["_post_labels"] = post["labels"]
    info["roofline_valid"] = all(bool(v) for v in checks.values())
    info["roofline_checks"] = checks
    if not info["roofline_valid"]:
        print("bench.py: the per-kernel accounting failed its consistency checks:", checks, file=sys.stderr)
        for k, ms in info.get("_post_labels", {}).items():
            print(f"  {k:28s} {ms * 1e3:9.1f} us", file=sys.stderr)
        for lbl, ms in lt[:200]:
            print(f"  live {lbl:24s} {ms * 1e3:9.1f} us", file=sys.stderr)
    info.pop("_post_labels", None)
    return elapsed, step, info


def time_other_config(model, B, img_size, steps=10, warmup=3):
    """Short single-GPU timing of another BASELINE config (same step definition and the same per-kernel roofline as the headline)."""
    import torch

    from savit_amd.config import get_config

    cfg = get_config(model, img_size=img_size)
    eng = build_engine(cfg, B)
    init_bench_params(eng, cfg)
    eng.refresh_weights()
    el, step, info = measure(eng, cfg, B, 1, 0, None, steps, warmup, breakdown=False)
    ips = B * steps / el
    res = {"model": model, "img_size": img_size, "images_per_gpu": B, "steps": steps, "warmup": warmup, "value": round(ips, 1),
           "unit": "images/s", "ms_per_step": round(el / steps * 1e3, 3),
           "step_roofline": step_roofline(cfg, ips, bool(getattr(eng, "cls_only_last", False)), bool(getattr(eng, "cls_fwd", False))),
           "roofline": {k: info["roofline"][k] for k in ("kernel", "launches_per_step", "avg_launch_ms", "achieved", "frac")},
           "roofline_valid": info["roofline_valid"], "final_loss": info["final_loss"]}
    del step, eng
    torch.cuda.empty_cache()
    return res


def time_config1_fp32(steps=100, warmup=10):
    """BASELINE config 1 on the GPU: ViT-Ti/16, batch 8, fp32 arithmetic, forward + loss (what simple_train.py's plumbing run computes
    before its backward; the CPU restatement of the same thing is cpu_baseline.forward_loss_value)."""
    import torch

    from savit_amd.config import get_config
    from savit_amd.engine_f32 import ViTEngineF32

    cfg = get_config("vit_ti_patch16")
    eng = ViTEngineF32(cfg, 8)
    init_bench_params(eng, cfg)
    g = torch.Generator(device="cuda").manual_seed(42)
    img = torch.randn(8, 224, 224, 3, device="cuda", generator=g)
    lab = torch.randint(0, 1000, (8,), device="cuda", generator=g, dtype=torch.int32)
    for _ in range(warmup):
        eng.forward(img)
        eng.loss_fn(lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(img)
        eng.loss_fn(lab)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res = {"model": "vit_ti_patch16", "dtype": "f32", "images_per_gpu": 8, "steps": steps, "warmup": warmup, "value": round(8 * steps / el, 1),
           "unit": "images/s", "ms_per_step": round(el / steps * 1e3, 3), "final_loss": round(float(eng.loss.item()), 4)}
    # the whole fp32 train step of simple_train.py:72-90 (forward, loss, backward, clip + Adam): what cpu_baseline.value times on the host
    def train_step():
        eng.forward(img)
        eng.loss_backward(lab, 0.1)
        eng.optimizer_step(lr=3e-3 * 8 / 512.0, weight_decay=0.0, max_norm=1.0)
    for _ in range(warmup):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        train_step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res["train_step"] = {"value": round(8 * steps / el, 1), "unit": "images/s", "ms_per_step": round(el / steps * 1e3, 3),
                         "final_loss": round(float(eng.loss.item()), 4)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)  # 20 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="vit_b_patch16", help="workload model (default: DeiT-B/16, the metric's config)")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU")
    ap.add_argument("--img-size", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--bucket-mb", type=float, default=48.0)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short timings of BASELINE configs 2, 4, 5")
    # the data-parallel plan's knobs (options.EngineOptions; unset = SAVIT_* environment, else the defaults)
    ap.add_argument("--reserved-cus", type=int, default=None, help="CUs backward leaves to the resident all-reduce (default: 16 at N > 1, 0 alone)")
    ap.add_argument("--wgrad-max-lag", type=int, default=None, help="layers a weight gradient may wait for a full grouped launch")
    ap.add_argument("--overlap-wgrad", type=int, choices=(0, 1), default=None, help="weight gradients on a side stream")
    ap.add_argument("--no-reserved-cus-sweep", action="store_true", help="N > 1: skip the reserved_cus 0 / 8 / 16 / 32 timings behind the headline")
    ap.add_argument("--sweep-steps", type=int, default=10)
    ap.add_argument("--sweep-timeout", type=float, default=240.0,
                    help="N > 1: seconds after which a sweep that has not returned is abandoned (the headline line is printed without it)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))  # this process stays GPU-free; the ranks are its grandchildren
    if os.environ.get("SAVIT_BENCH_RANK_PROBE"):
        return rank_probe()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU execution path for the product")
    # one rank per GPU (RCCL).  SAVIT_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the ranks then
    # share devices (local_rank modulo the device count) and the exchange goes through gloo - same hooks, same bucket plan.
    backend = os.environ.get("SAVIT_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.update(rccl_env_for(world, os.environ))  # (a rank started by an external launcher: before the first communicator)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import savit_amd  # noqa: F401
    from savit_amd import ddp
    if os.environ.get("SAVIT_EXP_LIB"):  # A/B runs against an experiment build of the library (tools/build_variant.sh)
        from savit_amd import lib as _l
        _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_%s.so" % os.environ["SAVIT_EXP_LIB"])
    from savit_amd.config import get_config

    cfg = get_config(args.model, img_size=args.img_size)
    B = args.batch
    opts = {"wgrad_max_lag": args.wgrad_max_lag, "overlap_wgrad": None if args.overlap_wgrad is None else bool(args.overlap_wgrad)}
    reserved = args.reserved_cus if args.reserved_cus is not None else ddp.default_reserved_cus(world)
    eng = build_engine(cfg, B, reserved_cus=reserved, **opts)  # backward is planned for the CUs the all-reduce leaves
    init_bench_params(eng, cfg)
    sync = None
    if world > 1:
        ddp.broadcast_params(eng.params)
        buckets = ddp.plan_buckets_for(eng.layout, int(args.bucket_mb * 2 ** 20 / 4))
        sync = ddp.GradSync(eng.grads, buckets)
        eng.bwd_hooks = sync.hooks()
    eng.refresh_weights()
    S = cfg.img_size
    elapsed, step, info = measure(eng, cfg, B, world, rank, sync, args.steps, args.warmup, dist)
    loss = info.pop("final_loss")  # read right behind the timed loop (the instrumented passes used to overwrite it)
    dist_info = None
    if dist is not None:
        # self-evidence of the multi-GPU run: the backend torch.distributed reports, its world size, every rank's device identity
        # (all-gathered) and the all-reduce time NOT hidden behind backward (HIP events on the compute stream, max over ranks)
        ids = [None] * world
        dist.all_gather_object(ids, rank_identity(local_rank))
        ex = torch.tensor([sum(a.elapsed_time(b) for a, b in step.exposed) / max(1, len(step.exposed))], device="cuda", dtype=torch.float64)
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ids,
                     "distinct_devices": len({(d.get("pci_bus_id"), d.get("uuid"), d.get("device")) for d in ids}),
                     "gradient_bytes_per_step": int(eng.grads.numel() * 4), "buckets": len(sync.buckets),
                     "allreduce_exposed_ms": round(float(ex.item()), 4), "reserved_cus": int(getattr(eng, "reserved_cus", 0)),
                     "rccl_channels": ddp.rccl_channels_in_effect(), "rccl_min_channels": os.environ.get("NCCL_MIN_NCHANNELS")}

    out = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        per_gpu = value / world
        out = {
            "metric": "images/sec DeiT-B/16 224^2 bf16 train step" if args.model == "vit_b_patch16" else f"images/sec {args.model} bf16 train step",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.model} ({'DeiT-B/16' if args.model == 'vit_b_patch16' else args.model}) {S}x{S} train step: "
                                   "fwd + label-smoothed CE + bwd + grad all-reduce + AdamW",
                       "images_per_gpu": B, "global_batch": B * world, "seq_len": cfg.seq_len, "parallelism": f"dp{world}",
                       "final_loss": round(loss, 4), "engine_options": engine_plan_facts(eng, args.bucket_mb if world > 1 else None),
                       **({"distributed": dist_info} if dist_info else {})},
            **({"allreduce_exposed_ms": dist_info["allreduce_exposed_ms"], "reserved_cus": dist_info["reserved_cus"],
                "rccl_channels": dist_info["rccl_channels"]} if dist_info else {}),
            "step_roofline": step_roofline(cfg, per_gpu, bool(getattr(eng, "cls_only_last", False)), bool(getattr(eng, "cls_fwd", False))),
        }
        out.update(info)
        dom = out["roofline"]["kernel"]
        d, F, M = cfg.embed_dim, cfg.hidden, eng.M
        # HBM-side traffic of that kernel: PMC passes cannot run inside this process, so the figure is the committed
        # rocprofv3 --pmc measurement of the same workload (tools/pmc_summary.py -> profiles/*_pmc_traffic.json), when it
        # covers this kernel and this is the headline workload; null otherwise.
        out["roofline"].update(traffic_lookup(dom, headline=(args.model == "vit_b_patch16" and B == 128 and args.img_size == 224)))
        if dom.startswith("gemm_wgrad_group"):
            # operands read once (X and dY of every weight of a launch, by its share of the weight's tiles) + dW written once (fp32; round 5:
            # first touch, nothing is read back), over the grouped launches of the backward plan
            plan = eng._serial_bwd_plan() if not eng.overlap_wgrad else eng._current_bwd_plan()
            tot = 0.0
            for arr in plan.wgrad_arrays:
                for q in arr:
                    alltiles = int(eng.L.savit_gemm_wgrad_group_tiles(q.Kin, q.Nout, eng.wgrad_tile))
                    cnt = q.tile_count if q.tile_count > 0 else alltiles - q.tile_begin
                    tot += cnt / alltiles * (2.0 * q.M * (q.Kin + q.Nout) + (4.0 if q.overwrite else 8.0) * q.Kin * q.Nout)
            out["roofline"]["algorithmic_bytes_per_launch"] = int(tot / max(1, len(plan.wgrad_arrays)))

    headline = args.model == "vit_b_patch16" and B == 128 and args.img_size == 224
    # ---- N > 1: the same step with backward planned for 256 - {0, 8, 16, 32} CUs, behind the headline measurement (all ranks)
    if world > 1 and not args.no_reserved_cus_sweep:
        del step, eng, sync
        torch.cuda.empty_cache()
        dog = arm_sweep_watchdog(args.sweep_timeout, out if rank == 0 else {}, rank)
        try:
            rows = reserved_cus_sweep(cfg, B, world, rank, dist, args.bucket_mb, args.sweep_steps, 3, opts)
            if rank == 0:
                out["reserved_cus_sweep"] = rows
        except Exception as e:  # (the headline above is already measured: a failure here - the same on every rank - must not lose the line)
            if rank == 0:
                out["reserved_cus_sweep_error"] = f"{type(e).__name__}: {e}"
        dog.cancel()
        eng = step = None

    # ---- other BASELINE configs on this GPU (rank 0, N=1, headline workload only): short timings, after the headline engine is freed
    if rank == 0 and world == 1 and headline and not args.no_other_configs:
        del step, eng
        torch.cuda.empty_cache()
        out["other_configs"] = {}
        for model, ob, osz, name in OTHER_CONFIGS:
            out["other_configs"][name] = time_other_config(model, ob, osz)
        out["other_configs"]["1. ViT-Ti/16 224^2 fp32, batch 8, forward + loss; train_step = + backward + Adam (fp32-input MFMA path)"] = time_config1_fp32()

    # ---- CPU baseline leg (rank 0, N=1 only): the oracle's torch-CPU restatement of BASELINE config 1, bounded samples
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_ref, vit_ref

        c1 = vit_ref.get_cfg("vit_ti_patch16", img_size=224)
        fb = torch_ref.time_train_step(c1, batch=8, seconds=args.cpu_seconds * 0.5)
        fo = torch_ref.time_train_step(c1, batch=8, seconds=args.cpu_seconds * 0.25, backward=False)
        out["cpu_baseline"] = {"value": round(fb["images_per_s"], 3), "unit": "images/s", "cores": fb["cores"], "kind": "port",
                               "forward_loss_value": round(fo["images_per_s"], 3),
                               "sample": f"BASELINE config 1 (ViT-Ti/16 224^2, batch 8, fp32), torch-CPU eager restatement of the reference "
                                         f"(JAX absent): {fb['steps']} steps of fwd+loss+bwd in {fb['seconds']:.1f} s (= value), "
                                         f"{fo['steps']} steps of fwd+loss in {fo['seconds']:.1f} s (= forward_loss_value)"}
        if headline:
            hb = torch_ref.time_train_step(vit_ref.get_cfg(args.model, img_size=args.img_size), batch=8, seconds=args.cpu_seconds * 0.5)
            out["cpu_baseline"]["headline_model_value"] = round(hb["images_per_s"], 3)
            out["cpu_baseline"]["headline_model_sample"] = f"{hb['steps']} fp32 fwd+loss+bwd steps of {args.model} at batch 8, {hb['seconds']:.1f} s"
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
