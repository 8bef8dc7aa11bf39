"""Every TN GEMM shape (M, N, K, epilogue, lda < K) the engines of all four families launch at the bench batch sizes -> tests/golden/gemm_shapes.json
(the fixture tests/test_abi.py uses to check that every tile compiled into libsavit.so is reachable from the auto heuristic).  Needs a GPU
(the engines allocate their buffers); run on the GPU box: python tools/dump_gemm_shapes.py > gpurun_out/gemm_shapes.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd  # noqa: F401
from savit_amd import lib as _lib
from savit_amd.config import get_config
from bench import build_engine

CASES = [("vit_ti_patch16", 256, 224), ("vit_s_patch16", 256, 224), ("vit_b_patch16", 128, 224), ("vit_l_patch16", 32, 384), ("cait_s_24", 64, 224),
         ("cait_xxs_24", 64, 224), ("mixer_b_patch16", 32, 224), ("mixer_s_patch32", 32, 224), ("tnt_s_patch16", 16, 224), ("tnt_b_patch16", 16, 224)]
out = {}
for name, B, S in CASES:
    cfg = get_config(name, img_size=S)
    eng = build_engine(cfg, B)
    eng.init_params(0)
    plans = [eng._build_fwd_plan(), eng._serial_bwd_plan()]
    shapes = set()
    for P in plans:
        for a in P.keep:
            if isinstance(a, _lib.GemmArgs):
                shapes.add((a.M, a.N, a.K, a.epilogue, int(a.lda < a.K and a.epilogue != _lib.EPI_PATCH)))
    # batch-size independent form: rows per image
    out[name] = {"batch": B, "img": S, "shapes": sorted(shapes)}
    del eng
    torch.cuda.empty_cache()
print(json.dumps(out))
