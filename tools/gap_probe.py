"""Gaps between consecutive kernels of one train step, from a rocprofv3 --kernel-trace CSV (dev tool): which launch boundaries cost what.
Usage: python tools/gap_probe.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, (s, e, n) in enumerate(rows) if "adamw" in n]
if len(ends) < 4:
    sys.exit("need >= 4 steps in the trace")
lo, hi = ends[-3], ends[-2]   # one step: behind one adamw launch up to and including the next
step = rows[lo:hi + 1]


def short(n):
    m = re.search(r"::(\w+(?:<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:48]


gaps = [(step[i + 1][0] - step[i][1], short(step[i][2]), short(step[i + 1][2])) for i in range(len(step) - 1)]
tot = sum(g for g, _, _ in gaps)
span = step[-1][1] - step[0][1]
print(f"launches {len(step) - 1}, span {span / 1e6:.3f} ms, kernel time {sum(e - s for s, e, _ in step[1:]) / 1e6:.3f} ms, gaps {tot / 1e6:.3f} ms "
      f"(mean {tot / len(gaps) / 1e3:.2f} us)")
by_pred, by_succ = collections.defaultdict(list), collections.defaultdict(list)
for g, a, b in gaps:
    by_pred[a].append(g)
    by_succ[b].append(g)
for title, d in (("gap BEHIND kernel", by_pred), ("gap IN FRONT OF kernel", by_succ)):
    print(title)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:16]:
        print(f"  {sum(v) / 1e3:8.1f} us total  x{len(v):3d}  mean {sum(v) / len(v) / 1e3:6.2f}  min {min(v) / 1e3:6.2f}  max {max(v) / 1e3:6.2f}  {k}")
print("largest single gaps:")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:7.2f} us  {a}  ->  {b}")
