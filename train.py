#!/usr/bin/env python3
"""train.py-shaped training loop on the MI355X engine.

Mirrors the reference's CLI and step semantics (/root/reference/train.py:130-255) with its defects fixed
(descent sign, single 1/n scaling, host-side logging, one parameter convention):

  flags            --data_dir --img_size --num_epochs --batch_size --label_smoothing --augmentation --model_name --lr
                   --weight_decay --clip_grad --checkpoint_dir --seed            (train.py:130-190; same names, same defaults)
  schedule         lr * batch_size / 512, 5 warm-up epochs then cosine to 1e-5   (train.py:214-220)
  step             forward, label-smoothed CE (+ mix labels), backward, gradient mean over ranks, AdamW, top-1
                   (train.py:77-109)
  eval             loss (no smoothing) + top-1/top-5 every 5 epochs, summed over ranks        (train.py:112-120,239-252)
  checkpoint       rank 0, every 10 epochs, keep 3 - and RESTORE on start (the reference only saves: train.py:123-127)

The reference's input pipeline (`input_pipeline.load`, train.py:48-74) does not exist in its tree (defect B1) and its
TF-data augmentation stack is out of scope, so batches come from `--data synthetic` (seeded N(0,1) images, uniform labels,
delivered in the loader's [H, W, C, N] fp32 layout of train.py:80) or from `--data npz:<file>` (arrays `images` NHWC
uint8/float and `labels`).  One process per GPU: `python -m torch.distributed.run --nproc-per-node 8 train.py ...`.
"""
import argparse
import glob
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def warmup_cosine(step, peak, warmup_steps, total_steps, end_value=1e-5):
    """optax.warmup_cosine_decay_schedule(init 0, peak, warmup, decay_steps=total, end) as used at train.py:214-220."""
    if step < warmup_steps:
        return peak * step / max(1, warmup_steps)
    t = min(1.0, (step - warmup_steps) / max(1, total_steps - warmup_steps))
    return end_value + (peak - end_value) * 0.5 * (1.0 + math.cos(math.pi * t))


class SyntheticData:
    """Seeded stand-in for the reference loader's batch contract {'images': [H,W,C,N] fp32, 'labels': [N]}."""

    def __init__(self, img_size, batch, num_classes, steps_per_epoch, seed, device):
        import torch

        self.g = torch.Generator(device=device).manual_seed(seed)
        self.shape = (img_size, img_size, 3, batch)
        self.batch, self.C, self.n, self.dev = batch, num_classes, steps_per_epoch, device

    def __iter__(self):
        import torch

        for _ in range(self.n):
            yield {"images": torch.randn(self.shape, device=self.dev, generator=self.g),
                   "labels": torch.randint(0, self.C, (self.batch,), device=self.dev, generator=self.g, dtype=torch.int32)}


class NpzData:
    def __init__(self, path, batch, rank, world, device, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        import numpy as np
        import torch

        z = np.load(path)
        imgs = torch.as_tensor(z["images"])[rank::world]
        self.labels = torch.as_tensor(z["labels"]).to(torch.int32)[rank::world]
        if imgs.dtype == torch.uint8:  # data/constants.py:7-8 normalisation
            imgs = (imgs.float() / 255.0 - torch.tensor(mean)) / torch.tensor(std)
        self.images, self.batch, self.dev = imgs.float(), batch, device

    def __iter__(self):
        for i in range(0, self.images.shape[0] - self.batch + 1, self.batch):
            img = self.images[i:i + self.batch].to(self.dev).permute(1, 2, 3, 0).contiguous()  # loader layout H W C N
            yield {"images": img, "labels": self.labels[i:i + self.batch].to(self.dev)}


def save_checkpoint(eng, ckpt_dir, step, keep=3):
    """checkpoints.save_checkpoint(dir, train_state, step, keep=3) (reference train.py:123-127): the SAME file name and wire
    format (msgpack of the TrainState state dict, savit_amd/flax_ckpt.py), so the file restores in Flax and vice versa."""
    from savit_amd import flax_ckpt

    return flax_ckpt.save_from_engine(eng, ckpt_dir, step, keep=keep)


def restore_checkpoint(eng, ckpt_dir):
    """Latest `checkpoint_<step>` (Flax msgpack; the reference never restores, defect B-list) - or a round-1 `.pt` file."""
    import torch

    from savit_amd import flax_ckpt

    path = flax_ckpt.latest_checkpoint(ckpt_dir)
    if path is not None:
        step = flax_ckpt.load_into_engine(eng, flax_ckpt.read_train_state(path))
        eng.weights_stale = True
        return step
    files = sorted(glob.glob(os.path.join(ckpt_dir, "checkpoint_*.pt")), key=lambda p: int(p.rsplit("_", 1)[1][:-3]))
    if not files:
        return 0
    ck = torch.load(files[-1], map_location="cpu")
    eng.params.copy_(ck["params"])
    if ck["m"] is not None:
        eng.adam_m = ck["m"].to(eng.params.device)
        eng.adam_v = ck["v"].to(eng.params.device)
    eng.step_count = ck["opt_step"]
    eng.weights_stale = True
    return int(ck["step"])


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data_dir", default=None)
    ap.add_argument("--data", default="synthetic", help="synthetic | npz:<train.npz>[,<eval.npz>]")
    ap.add_argument("--img_size", type=int, default=224)
    ap.add_argument("--num_epochs", type=int, default=300)
    ap.add_argument("--batch_size", type=int, default=32, help="GLOBAL batch (train.py:142-147), split over ranks")
    ap.add_argument("--label_smoothing", type=float, default=0.1)
    ap.add_argument("--augmentation", default=None, help="accepted for CLI compatibility; augmentation ops are out of scope")
    ap.add_argument("--model_name", default="vit_b_patch16")
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--weight_decay", type=float, default=1e-4)
    ap.add_argument("--clip_grad", type=float, default=None)
    ap.add_argument("--checkpoint_dir", default=None)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--steps_per_epoch", type=int, default=None, help="synthetic data: steps per epoch (ImageNet-1k: 1281167 // batch)")
    ap.add_argument("--max_steps", type=int, default=None)
    ap.add_argument("--eval_every_epochs", type=int, default=5)
    ap.add_argument("--save_every_epochs", type=int, default=10)
    ap.add_argument("--log_every", type=int, default=10)
    # GPU-side input path (the reference does these in its TF host pipeline: data/preprocess/augment_utils.py:85-136)
    ap.add_argument("--mixup_alpha", type=float, default=0.0, help="batch mixup Beta parameter (reference pipeline: 0.8); 0 = off")
    ap.add_argument("--cutmix_alpha", type=float, default=0.0, help="batch cutmix Beta parameter (reference pipeline: 1.0); 0 = off")
    ap.add_argument("--mix_prob", type=float, default=1.0, help="probability of applying the drawn mix augmentation to a batch")
    ap.add_argument("--dtype", default="bfloat16", choices=("bfloat16", "float32"),
                    help="bfloat16 = what train.py:222-224 passes to create_model; float32 = create_model's default, the arithmetic of "
                         "simple_train.py:72-90 (one GPU, no mixup: the exact-fp32 engines)")
    args = ap.parse_args(argv)

    import torch

    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X: there is no CPU execution path")
    torch.cuda.set_device(local)
    import savit_amd  # noqa: F401
    from savit_amd import ddp

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a data-parallel rank plans its backward launches for the CUs the resident all-reduce leaves (engine reserved_cus, ddp.py) -
        # and, with SAVIT_PIN_RCCL_CHANNELS=1 (opt-in: unverified on > 1 GPU), bounds RCCL's channels (one channel = one resident
        # workgroup = one CU) to that reserve before the first communicator exists (jax.lax.pmean, /root/reference/train.py:96, has no counterpart: XLA schedules its own collectives)
        os.environ.setdefault("SAVIT_RESERVED_CUS", str(ddp.default_reserved_cus(world)))
        rccl_env = ddp.apply_rccl_channel_env(int(os.environ["SAVIT_RESERVED_CUS"]))
        if rank == 0 and rccl_env:
            print("[train] RCCL channel bounds:", rccl_env, flush=True)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    bs, ragged = divmod(args.batch_size, world)
    if ragged:
        raise ValueError(f"Batch size {args.batch_size} must be divisible by num devices {world}")  # train.py:43-47

    from savit_amd.model import create_model

    fp32 = args.dtype == "float32"
    model = create_model(args.model_name, num_classes=1000, dtype=torch.float32 if fp32 else torch.bfloat16, img_size=args.img_size)  # train.py:222-224
    if fp32 and (world > 1 or args.mixup_alpha > 0 or args.cutmix_alpha > 0):
        raise SystemExit("--dtype float32 (create_model's default arithmetic, simple_train.py; what the reference always computes CaiT in: "
                         "cait.py:147-154) trains every family on one GPU without mix augmentation; the data-parallel / mixup paths train in bfloat16")
    model.init(args.seed, torch.ones(1, args.img_size, args.img_size, 3, device="cuda"), is_training=False)  # train.py:29-31
    eng = model.engine(bs)
    start = restore_checkpoint(eng, args.checkpoint_dir) if args.checkpoint_dir else 0
    sync = None
    if world > 1:
        ddp.broadcast_params(eng.params)  # replicate, train.py:228
        sync = ddp.GradSync(eng.grads, ddp.plan_buckets_for(eng.layout, 48 * 2 ** 20 // 4))
        eng.bwd_hooks = sync.hooks()
    eng.refresh_weights()

    spe = args.steps_per_epoch or max(1, 1281167 // args.batch_size)
    total = args.num_epochs * spe
    peak = args.lr * args.batch_size / 512.0
    dev = torch.device("cuda", local)
    eval_src = None
    if args.data == "synthetic":
        train_src = lambda ep: SyntheticData(args.img_size, bs, 1000, spe, args.seed + 1000 * ep + rank, dev)  # noqa: E731
        eval_src = lambda: SyntheticData(args.img_size, bs, 1000, 2, args.seed - 1 - rank, dev)  # noqa: E731
    elif args.data.startswith("npz:"):
        paths = args.data[4:].split(",")
        train_src = lambda ep: NpzData(paths[0], bs, rank, world, dev)  # noqa: E731
        if len(paths) > 1:
            eval_src = lambda: NpzData(paths[1], bs, rank, world, dev)  # noqa: E731
    else:
        raise SystemExit("--data must be 'synthetic' or 'npz:<file>'")

    def evaluate():
        tot = torch.zeros(4, device=dev)  # loss sum, top1, top5, count
        for batch in eval_src():
            eng.forward(batch["images"])
            eng.loss.zero_()
            eng.labels.copy_(batch["labels"])
            from savit_amd import lib as _lib

            _lib.check(eng.L.savit_softmax_xent(eng.logits.data_ptr(), eng.cfg.num_classes, eng.labels.data_ptr(), None, None, 0.0, 1.0 / bs,
                                                eng.loss_rows.data_ptr(), eng.loss.data_ptr(), None, 0, None, eng.top1.data_ptr(),
                                                eng.top5.data_ptr(), bs, eng.cfg.num_classes, torch.cuda.current_stream().cuda_stream),
                       "savit_softmax_xent")
            tot += torch.stack([eng.loss_rows.sum(), eng.top1.sum(), eng.top5.sum(), torch.tensor(float(bs), device=dev)])
        if dist is not None:
            dist.all_reduce(tot)  # psum, train.py:120
        return (tot[0] / tot[3]).item(), (tot[1] / tot[3]).item(), (tot[2] / tot[3]).item()

    mix_on = args.mixup_alpha > 0 or args.cutmix_alpha > 0
    mix_gen = torch.Generator(device=dev).manual_seed(args.seed + 7919 * (rank + 1)) if mix_on else None
    step, t0, seen = start, time.perf_counter(), 0
    for epoch in range(start // spe, args.num_epochs):
        for batch in train_src(epoch):
            lr = warmup_cosine(step, peak, 5 * spe, total)
            images = batch["images"]
            if mix_on and "mix_labels" not in batch:
                from savit_amd import augment, ops as _ops

                S = args.img_size
                nhwc = _ops.hwcn_to_nhwc_bf16(images) if tuple(images.shape) == (S, S, 3, bs) and images.dtype == torch.float32 \
                    else images.to(torch.bfloat16).contiguous()
                images, _, ml, ratio = augment.mix_batch(nhwc, batch["labels"].to(torch.int32), args.mixup_alpha, args.cutmix_alpha,
                                                         args.mix_prob, mix_gen)
                if ml is not None:
                    batch = dict(batch, mix_labels=ml, ratio=ratio)
            if eng.cfg.kind == "cait":
                from savit_amd.cait_engine import stochastic_depth_seed

                eng.forward(images, is_training=True, sd_seed=stochastic_depth_seed(args.seed, rank, step))
            else:
                eng.forward(images)
            if fp32:
                eng.loss_backward(batch["labels"], args.label_smoothing)
            else:
                eng.loss_backward(batch["labels"], args.label_smoothing, batch.get("mix_labels"), batch.get("ratio"))
            if sync is not None:
                sync.wait()
            eng.optimizer_step(lr=lr, weight_decay=args.weight_decay, max_norm=args.clip_grad or 0.0,
                               grad_scale=sync.grad_scale if sync else 1.0)
            step += 1
            seen += args.batch_size
            if step % args.log_every == 0:
                stats = torch.stack([eng.loss[0], eng.top1.mean()])
                if dist is not None:
                    ddp.allreduce_scalar_mean(stats)
                if rank == 0:
                    dt = time.perf_counter() - t0
                    print(json.dumps({"step": step, "epoch": epoch, "train/loss": round(stats[0].item(), 4),
                                      "train/top-1-acc": round(stats[1].item(), 4), "lr": lr, "images_per_s": round(seen / dt, 1)}), flush=True)
                    t0, seen = time.perf_counter(), 0
            if args.max_steps and step >= args.max_steps:
                break
        if args.checkpoint_dir and rank == 0 and (epoch + 1) % args.save_every_epochs == 0:
            save_checkpoint(eng, args.checkpoint_dir, step)
        if eval_src is not None and (epoch + 1) % args.eval_every_epochs == 0:
            loss, t1, t5 = evaluate()
            if rank == 0:
                print(json.dumps({"step": step, "eval/loss": round(loss, 4), "eval/top-1-acc": round(t1, 4), "eval/top-5-acc": round(t5, 4)}), flush=True)
        if args.max_steps and step >= args.max_steps:
            break
    if args.checkpoint_dir and rank == 0:
        save_checkpoint(eng, args.checkpoint_dir, step)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return step


if __name__ == "__main__":
    main()
