"""Timeline of one train step from a rocprofv3 --kernel-trace CSV (dev tool): how much of the step two kernels really overlap.
Usage: python tools/timeline.py <kernel_trace.csv> [steps_in_trace]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
# one step = from one adamw_kernel end to the next
ends = [e for s, e, n, *_ in rows if "adamw" in n]
if len(ends) < 3:
    sys.exit("need >= 3 steps")
lo, hi = ends[-3], ends[-2]
step = [r for r in rows if r[0] >= lo and r[1] <= hi]
span = hi - lo
tot = sum(e - s for s, e, *_ in step)
# union of busy intervals and time with >= 2 kernels in flight
ev = sorted([(s, 1) for s, e, *_ in step] + [(e, -1) for s, e, *_ in step])
busy = two = 0
cur = 0
last = ev[0][0]
for t, d in ev:
    if cur >= 1: busy += t - last
    if cur >= 2: two += t - last
    cur += d; last = t
# per-queue busy time
qb = collections.defaultdict(int)
for s_, e_, n_, q_, st_ in step: qb[(q_, st_)] += e_ - s_
print("per (queue, stream) kernel time (ms):", {k: round(v / 1e6, 3) for k, v in qb.items()})
print(f"step span {span/1e6:.3f} ms, sum of kernel durations {tot/1e6:.3f} ms, busy (>=1 kernel) {busy/1e6:.3f} ms, >=2 kernels in flight {two/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, *_ in step:
    import re
    m = re.search(r"::(\w+(?:<[^>]*>)?)", n)
    k = (m.group(1) if m else n)[:70]
    by[k][0] += 1; by[k][1] += e - s
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{t/1e6:8.3f} ms  x{c:4d}  avg {t/c/1e3:8.1f} us  {k}")
