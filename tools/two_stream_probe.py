"""Timing probe: backward with the weight-gradient GEMMs spread over several side streams and fewer K-splits (timing only:
cross-stream hazards between the side streams are ignored)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd.config import get_config
from savit_amd.engine import ViTEngine

cfg = get_config("vit_b_patch16"); B = 128
eng = ViTEngine(cfg, B); eng.init_params(42)
eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes) * cfg.embed_dim ** -0.5)
img = torch.randn(B, 224, 224, 3, device="cuda").to(torch.bfloat16)
lab = torch.randint(0, 1000, (B,), device="cuda", dtype=torch.int32)
for _ in range(3):
    eng.forward(img); eng.loss_backward(lab); eng.optimizer_step(1e-4, 1e-4, 1.0)
plan = eng._serial_bwd_plan()
sides = [torch.cuda.Stream() for _ in range(4)]
def run(nside, splits):
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    t0 = time.perf_counter()
    for it in range(5):
        k = 0
        for fn, args, label in plan.calls:
            if nside and label.endswith(".wgrad"):
                s = sides[k % nside]; k += 1
                ev = torch.cuda.Event(); ev.record(main); s.wait_event(ev)
                a = list(args)
                if splits and not label.startswith(("head", "Wpe")): a[9] = splits
                rc = fn(*a, s.cuda_stream)
            else:
                rc = fn(*args, main.cuda_stream)
            assert rc == 0, label
        for s in sides[:max(nside, 0)]:
            ev2 = torch.cuda.Event(); ev2.record(s); main.wait_event(ev2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3
for nside in (0, 1):
    for splits in (0, 4, 2, 1):
        if nside == 0 and splits: continue
        print(f"side streams {nside} splits {splits or 'auto'}: bwd {run(nside, splits):.3f} ms", flush=True)

