"""Per-launch timing of one train step (dev tool): timing.instrumented_steps over forward + loss + backward + optimizer."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import lib as _l
if os.environ.get("SAVIT_EXP_LIB"):
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_%s.so" % os.environ["SAVIT_EXP_LIB"])
from savit_amd.config import get_config
from savit_amd.timing import instrumented_steps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_engine

model = sys.argv[1] if len(sys.argv) > 1 else "vit_b_patch16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
IMG = int(sys.argv[3]) if len(sys.argv) > 3 else 224
cfg = get_config(model, img_size=IMG)
eng = build_engine(cfg, B)
eng.init_params(42)
eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes) * cfg.embed_dim ** -0.5)
img = torch.randn(B, IMG, IMG, 3, device="cuda").to(torch.bfloat16)
lab = torch.randint(0, 1000, (B,), device="cuda", dtype=torch.int32)
train = cfg.kind == "cait"


def step():
    eng.forward(is_training=True) if train else eng.forward()
    eng.loss_backward(lab)
    eng.optimizer_step(1e-4, 1e-4, 1.0)


eng.set_images(img)
for _ in range(3):
    step()
res = instrumented_steps(eng, step, reps=3)
acc = collections.OrderedDict()
for k, v in res["labels"].items():
    k = k.split("#")[0]
    kk = ".".join(k.split(".")[1:]) if k.startswith("l") and k[1].isdigit() else (k[:11] if k.startswith("wgrad.group") else k)
    acc.setdefault(kk, []).append(v)
d, F, M = cfg.embed_dim, cfg.hidden, eng.M
fl = {"qkv": 2.0*M*d*3*d, "proj": 2.0*M*d*d, "fc1": 2.0*M*d*F, "fc2": 2.0*M*d*F}
tot = 0
for k, v in acc.items():
    avg = sum(v) / len(v)
    tot += sum(v)
    key = {"Wqkv": "qkv", "Wo": "proj", "W1": "fc1", "W2": "fc2"}.get(k.split(".")[0], k.split(".")[0])
    tf = f"{fl[key]/avg/1e9:7.1f} TF/s" if key in fl else ""
    print(f"{k:18s} x{len(v):4d}  avg {avg*1e3:8.1f} us  total {sum(v):7.3f} ms  {tf}")
if os.environ.get("SAVIT_PROFILE_LAYER"):  # the launches of one layer one by one (e.g. the last layer, which runs on the cls rows)
    pre = "l%s." % os.environ["SAVIT_PROFILE_LAYER"]
    for k, v in res["labels"].items():
        if k.startswith(pre):
            print(f"  {k:24s} {v*1e3:8.1f} us")
print("sum %.3f ms; reps:" % tot, [{k: (round(x, 3) if isinstance(x, float) else x) for k, x in r.items()} for r in res["reps"]], "gate", res["gate_us"], "us")
