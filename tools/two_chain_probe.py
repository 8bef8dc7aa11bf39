"""Experiment: two half-batch engines on two streams against one full-batch engine (forward + loss + backward, no optimizer).
The chain of a train step is ~190 dependent launches with a 4-5 us dispatch gap each, tile rounds with idle CUs at their tails and
HBM-bound launches (LayerNorm) during which the matrix pipes idle: two INDEPENDENT chains could fill each other's gaps.
usage: python tools/two_chain_probe.py [model] [batch] [img]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import savit_amd  # noqa: F401,E402
from savit_amd.config import get_config  # noqa: E402
from savit_amd.engine import ViTEngine  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vit_b_patch16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    img = int(sys.argv[3]) if len(sys.argv) > 3 else 224
    cfg = get_config(name, img_size=img)
    g = torch.Generator(device="cuda").manual_seed(0)
    imgs = torch.randn(B, img, img, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, cfg.num_classes, (B,), device="cuda", generator=g, dtype=torch.int32)
    full = ViTEngine(cfg, B)
    full.init_params(0)
    halves = [ViTEngine(cfg, B // 2) for _ in range(2)]
    for h in halves:
        h.init_params(0)
    streams = [torch.cuda.Stream() for _ in range(2)]
    parts = [(imgs[:B // 2].contiguous(), lab[:B // 2].contiguous()), (imgs[B // 2:].contiguous(), lab[B // 2:].contiguous())]
    full.set_images(imgs)
    for h, (im, _) in zip(halves, parts):
        h.set_images(im)

    def one():
        full.forward()
        full.loss_backward(lab)

    def two():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        # interleave the issue layer by layer is not possible through the public calls: issue forward A, forward B, backward A, backward B
        for h, s in zip(halves, streams):
            with torch.cuda.stream(s):
                h.forward()
        for h, s, (_, lb) in zip(halves, streams, parts):
            with torch.cuda.stream(s):
                h.loss_backward(lb)
        for s in streams:
            cur.wait_stream(s)

    def seq():
        for h, (_, lb) in zip(halves, parts):
            h.forward()
            h.loss_backward(lb)

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n

    for rep in range(2):
        print(f"{name} B={B}: one engine {timed(one):.3f} ms | two half-batch engines, two streams {timed(two):.3f} ms | the same two, one stream "
              f"{timed(seq):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
