"""Per-launch timing of one CaiT train step (dev tool)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import lib as _l
if os.environ.get("SAVIT_EXP_LIB"):
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_%s.so" % os.environ["SAVIT_EXP_LIB"])
from savit_amd.config import get_config
from savit_amd.cait_engine import CaiTEngine
model = sys.argv[1] if len(sys.argv) > 1 else "cait_s_24"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = get_config(model)
eng = CaiTEngine(cfg, B); eng.init_params(42)
img = torch.randn(B, 224, 224, 3, device="cuda").to(torch.bfloat16)
lab = torch.randint(0, 1000, (B,), device="cuda", dtype=torch.int32)
for _ in range(2):
    eng.forward(img, is_training=True); eng.loss_backward(lab); eng.optimizer_step(1e-4, 1e-4, 1.0)
acc = collections.defaultdict(list)
for _ in range(2):
    eng.set_images(img)
    t = eng.profile_step(lab)
    for k, v in t.items():
        kk = ".".join(k.split(".")[1:]) if k[0] in "lc" and k[1].isdigit() else k
        acc[kk].append(v)
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v) / 2
    print(f"{k:18s} x{len(v)/2:4.0f}  avg {sum(v)/len(v)*1e3:8.1f} us  total {sum(v)/2:7.3f} ms")
print("sum", tot)
