"""Experiment: does replaying the train step as a captured hipGraph shorten it?  (The step is ~195 launches whose host issue takes
~2 ms of a ~16 ms step - the host is never the limiter - so a graph can only win what the command processor saves between
pre-recorded dispatches.)  Scalars that change per step (AdamW's step count / learning rate) are frozen in the capture: timing only.
usage: python tools/graph_probe.py [model] [batch] [img]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import savit_amd  # noqa: F401,E402
from savit_amd.config import get_config  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vit_b_patch16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    img = int(sys.argv[3]) if len(sys.argv) > 3 else 224
    cfg = get_config(name, img_size=img)
    if cfg.kind == "cait":
        from savit_amd.cait_engine import CaiTEngine as Eng
    else:
        from savit_amd.engine import ViTEngine as Eng
    eng = Eng(cfg, B)
    eng.init_params(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    imgs = torch.randn(B, img, img, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, cfg.num_classes, (B,), device="cuda", generator=g, dtype=torch.int32)

    def step():
        eng.forward(imgs)
        eng.loss_backward(lab)
        eng.optimizer_step(lr=1e-4, weight_decay=0.05, max_norm=1.0)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n

    for _ in range(5):
        step()
    eager = [timed(step, 20) for _ in range(3)]
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(graph):
        step()
    for _ in range(3):
        graph.replay()
    rep = [timed(graph.replay, 20) for _ in range(3)]
    eager2 = [timed(step, 20) for _ in range(3)]
    print(f"{name} B={B}: eager {min(eager):.3f} ms (runs {['%.3f' % x for x in eager]}), graph replay {min(rep):.3f} ms "
          f"({['%.3f' % x for x in rep]}), eager again {min(eager2):.3f} ms; loss {float(eng.loss):.4f}")


if __name__ == "__main__":
    main()
