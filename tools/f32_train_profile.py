"""ViT-Ti/16 fp32 train step (BASELINE config 1, batch 8) under rocprofv3 --kernel-trace --stats: where the step goes (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import savit_amd  # noqa: F401
from savit_amd.config import get_config
from savit_amd.engine_f32 import ViTEngineF32

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = get_config("vit_ti_patch16")
eng = ViTEngineF32(cfg, B)
eng.init_params(0)
img = torch.randn(B, 224, 224, 3, device="cuda")
lab = torch.randint(0, 1000, (B,), device="cuda", dtype=torch.int32)
for _ in range(5):
    eng.forward(img)
    eng.loss_backward(lab, 0.1)
    eng.optimizer_step(lr=1e-4, max_norm=1.0)
torch.cuda.synchronize()
# per-label timing of one backward plan
evs = []
s = torch.cuda.current_stream().cuda_stream
pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in eng._bwd.calls]
for a, b in pool:  # events are created lazily at their first record: do that outside the timed launches
    a.record(); b.record()
torch.cuda.synchronize()
for (fn, args, label), (a, b) in zip(eng._bwd.calls, pool):
    a.record(); fn(*args, s); b.record()
    evs.append((label, a, b))
torch.cuda.synchronize()
agg = {}
for label, a, b in evs:
    k = label.split(".", 1)[1] if label[0] == "l" and label[1].isdigit() else label.split(".")[0]
    agg[k] = agg.get(k, 0.0) + a.elapsed_time(b)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{k:20s} {v:8.3f} ms")
print("backward total", sum(agg.values()))
