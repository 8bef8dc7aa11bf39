"""Weight-gradient variants on the d = 384 shapes (DeiT-S / CaiT-S, 256 images): one process per variant (SAVIT_WGRAD_VARIANT is read once)."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, savit_amd
    from savit_amd import ops
    bf16 = torch.bfloat16
    M = 256 * 196
    flush = torch.empty(512 * 2 ** 20, dtype=torch.uint8, device="cuda")
    for name, Kin, Nout in (("Wqkv", 384, 1152), ("Wo", 384, 384), ("W1", 384, 1536), ("W2", 1536, 384), ("B:W1", 768, 3072), ("B:Wo", 768, 768)):
        Mv = M if not name.startswith("B:") else 25216
        X = torch.randn(Mv, Kin, device="cuda").to(bf16); dY = torch.randn(Mv, Nout, device="cuda").to(bf16)
        dW = torch.zeros(Kin, Nout, device="cuda")
        ws = ops.wgrad_workspace(Mv, Kin, Nout, 0)
        ts = []
        for rep in range(8):
            flush.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ops.gemm_wgrad(X, dY, dW, workspace=ws); b.record(); torch.cuda.synchronize()
            if rep >= 2: ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        print(f"variant {sys.argv[1]} {name:5s} {Kin}x{Nout}: median {ts[len(ts)//2]:7.1f} us  {2.0*Mv*Kin*Nout/ts[len(ts)//2]/1e6:6.0f} TF/s", flush=True)
else:
    for v in ("1", "2", "3", "4", "5"):
        subprocess.run([sys.executable, __file__, v], env=dict(os.environ, SAVIT_WGRAD_VARIANT=v))
