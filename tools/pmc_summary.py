"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-kernel HBM-side traffic per launch.
Correction per MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports HALF the bytes of wide
(16 B/lane) coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores and fp32 atomics."""
import collections, csv, glob, importlib.util, json, os, re, sys


def fingerprint():
    """Code hashes of the library the counters were just collected on (run this script on the GPU box, right behind the passes):
    bench.py compares them with the library it runs and reports "traffic_stale" on a mismatch (fingerprint.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("savit_fingerprint", os.path.join(root, "self-attention-experiments-vision_amd", "fingerprint.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.library_fingerprint(os.environ.get("SAVIT_LIB_PATH") or m.default_library())


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        m = re.search(r"(gemm_\w+<[^>]*>|attn_\w+<\d+>|ln_\w+<\d+>|adamw_kernel|th_\w+(<\d+>)?|layerscale_bwd_kernel<\d+>|class_attn_\w+<\d+>|transpose_bf16_kernel|token_mean_\w+|seq16_\w+_kernel|cast_colsum_kernel|inner2outer_\w+_kernel|pixel_gather_kernel|colsum_finalize_kernel|wgrad_reduce_kernel)", name)
        if m:
            d[m.group(1).replace(" ", "")].append(float(r["Counter_Value"]))
    return d

fdir, wdir, out = sys.argv[1:4]
f = agg((glob.glob(fdir + "/*/*counter_collection.csv") + glob.glob(fdir + "/*counter_collection.csv"))[0])
w = agg((glob.glob(wdir + "/*/*counter_collection.csv") + glob.glob(wdir + "/*counter_collection.csv"))[0])
res = {}
for k in sorted(set(f) | set(w)):
    fk = sum(f.get(k, [0])) / max(1, len(f.get(k, [0])))
    wk = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    res[k] = {"launches": len(f.get(k, [])), "fetch_bytes_corrected": round(2 * fk * 1024), "write_bytes": round(wk * 1024),
              "traffic_bytes": round((2 * fk + wk) * 1024)}
note = sys.argv[4] if len(sys.argv) > 4 else "DeiT-B/16, 128 img"
fp = fingerprint()
fp["kernels"] = {k: v for k, v in fp["kernels"].items() if k in res}  # only the kernels this file has figures for
json.dump({"note": f"per-launch averages over one bench.py run ({note}); FETCH_SIZE doubled per MI355X_MICROARCH.md", "kernels": res,
           "library": fp}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes"])[:14]:
    print(f"{k:50s} fetch {v['fetch_bytes_corrected']/1e6:8.1f} MB  write {v['write_bytes']/1e6:8.1f} MB")
