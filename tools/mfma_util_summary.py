"""Per-kernel MFMA pipe utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) of bench.py.
util = MFMA busy cycles summed over all SIMDs / (cycles the launch was active x 1024 SIMDs); GRBM_GUI_ACTIVE is reported summed
over the 8 XCDs, so the active cycles of the launch are that / 8.  The in-kernel clock follows as active cycles / duration."""
import collections, csv, glob, json, re, sys

d = sys.argv[1]
out = sys.argv[2]
cc = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
kt = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    m = re.search(r"(gemm_\w+<[^>]*>|attn_\w+<\d+>)", r["Kernel_Name"])
    if not m:
        continue
    k = m.group(1).replace(" ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[k]["_dur"].append(dur.get(r["Dispatch_Id"], 0))
res = {}
for k, v in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or "GRBM_GUI_ACTIVE" not in v:
        continue
    n = len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
    busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / n
    act = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"]) / 8.0
    ns = sum(v["_dur"]) / max(1, len(v["_dur"])) / max(1, len(v) - 1)
    res[k] = {"launches": n, "mfma_busy_cycles": round(busy), "active_cycles": round(act), "mfma_util": round(busy / (act * 1024.0), 4),
              "avg_ns": round(sum(v["_dur"]) / max(1, len(v["_dur"]))), "clock_ghz": round(act / (sum(v["_dur"]) / max(1, len(v["_dur"]))), 3)}
json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on bench.py (DeiT-B/16, 128 img, one stream); util = busy / (active x 1024 SIMDs)",
           "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["mfma_util"]):
    print(f"{k:48s} util {v['mfma_util']*100:5.1f} %  clock {v['clock_ghz']:.2f} GHz  avg {v['avg_ns']/1e3:7.1f} us  x{v['launches']}")
