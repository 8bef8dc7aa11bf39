"""Batch mix augmentations on the GPU: host-side sampling of the random draws + the HIP kernels.

Mirrors /root/reference/data/preprocess/augment_utils.py:85-136 (`create_mix_augment`: pick mixup or cutmix per batch, apply
with probability `prob_to_apply`) and augment_ops.py:98-181 (`batch_cutmix`, `batch_mixup`).  The reference runs these in its
TF host pipeline on one-hot labels; here the images are mixed by `ops.batch_mixup` / `ops.batch_cutmix` on the bf16 batch that
train_step consumes, and the label mix is handed to the loss kernel in the two-label form train.py:83-88 already uses
(`ratio*y + (1-ratio)*y1`), which is the same arithmetic as mixing the one-hot rows.  TF's stateless RNG stream cannot be
reproduced, so the draws come from a torch generator; their DISTRIBUTIONS are the reference's:
  mixup : u ~ U[0,1), mix = u**(1/alpha) / 2, mix = max(mix, 1-mix), partner = random permutation      (augment_ops.py:162-176)
  cutmix: u ~ U[0,1), w = u**(1/beta) / 2 (box area fraction AND own-label weight), ratio = sqrt(w),
          box h = int(ratio*H), w = int(ratio*W), offsets uniform then `% (size - box)`, partner = B-1-b  (augment_ops.py:119-141,70-81)
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops


def sample_mixup(batch: int, alpha: float = 0.8, generator: Optional[torch.Generator] = None, device="cuda"):
    """-> (weight fp32 [B] in [0.5, 1], index int32 [B])"""
    u = torch.rand(batch, generator=generator, device=device)
    mix = u.pow(1.0 / alpha) / 2
    mix = torch.maximum(mix, 1 - mix)
    index = torch.randperm(batch, generator=generator, device=device).to(torch.int32)
    return mix.float().contiguous(), index.contiguous()


def sample_cutmix(batch: int, height: int, width: int, beta: float = 1.0, generator: Optional[torch.Generator] = None, device="cuda"):
    """-> (weight fp32 [B] in [0, 0.5] = area kept from the own image, box int32 [B,4] = y0,y1,x0,x1, index int32 [B] = B-1-b)"""
    u = torch.rand(batch, generator=generator, device=device)
    w = u.pow(1.0 / beta) / 2
    ratio = w.sqrt()
    mh = (ratio * height).to(torch.int64)
    mw = (ratio * width).to(torch.int64)
    xs = torch.randint(0, width, (batch,), generator=generator, device=device)
    ys = torch.randint(0, height, (batch,), generator=generator, device=device)
    xs = xs % (width - mw)    # "avoid shifting too much" (augment_ops.py:80-81); mw <= 0.707*W so the modulus is positive
    ys = ys % (height - mh)
    box = torch.stack([ys, ys + mh, xs, xs + mw], dim=1).to(torch.int32).contiguous()
    index = torch.arange(batch - 1, -1, -1, device=device, dtype=torch.int32)
    return w.float().contiguous(), box, index


def mix_batch(images: torch.Tensor, labels: torch.Tensor, mixup_alpha: float = 0.8, cutmix_alpha: float = 1.0, prob_to_apply: float = 1.0,
              generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    """create_mix_augment (augment_utils.py:85-136) for one bf16 [B,H,W,C] GPU batch.
    Returns (images', labels, mix_labels, ratio): the loss is ratio*CE(labels) + (1-ratio)*CE(mix_labels) (train.py:83-88);
    mix_labels / ratio are None when the augmentation is not applied."""
    B, H, W, _ = images.shape
    dev = images.device
    branches = [b for b, a in (("mixup", mixup_alpha), ("cutmix", cutmix_alpha)) if a]
    if not branches or prob_to_apply == 0:
        return images, labels, None, None
    if prob_to_apply < 1.0 and float(torch.rand((), generator=generator, device=dev)) >= prob_to_apply:
        return images, labels, None, None
    which = branches[int(torch.randint(0, len(branches), (), generator=generator, device=dev))]
    if which == "mixup":
        weight, index = sample_mixup(B, mixup_alpha, generator, dev)
        out = ops.batch_mixup(images, weight, index)
    else:
        weight, box, index = sample_cutmix(B, H, W, cutmix_alpha, generator, dev)
        out = ops.batch_cutmix(images, box, index)
    return out, labels, labels[index.long()].contiguous(), weight
