"""Offline check for the inline-asm scalar loads of the talking-heads row kernels (ThCoef::issue / wait in csrc/attention.hip): between
an `s_load_dwordx16` and the next `s_waitcnt lgkmcnt(0)` no instruction may READ or COPY the destination SGPRs (they are in flight;
hipcc does not know that).  usage: check_inflight_sgprs.py kernel.s  (ISA text: `hipcc -S --cuda-device-only` or `llvm-objdump -d`)"""
import re
import sys

_RNG = re.compile(r"s\[(\d+):(\d+)\]|\bs(\d+)\b")


def _sregs(text):
    out = set()
    for m in _RNG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def violations(isa_text):
    """-> (number of s_load_dwordx16 seen, [offending instruction lines])"""
    loads, bad, inflight = 0, [], set()
    for raw in isa_text.splitlines():
        l = raw.split("//")[0].strip()
        if not l or l.startswith((";", ".")) or l.endswith(":"):
            continue
        op = l.split()[0]
        if op == "s_load_dwordx16":
            loads += 1
            inflight |= _sregs(l.split(None, 1)[1].split(",")[0])
            continue
        if op == "s_waitcnt" and "lgkmcnt(0)" in l:
            inflight.clear()
            continue
        if inflight and " " in l and _sregs(l.split(None, 1)[1]) & inflight:
            bad.append(l)
    return loads, bad


if __name__ == "__main__":
    n, bad = violations(open(sys.argv[1]).read())
    for l in bad:
        print("touches in-flight SGPRs:", l)
    print(f"s_load_dwordx16: {n}; in-flight SGPR violations: {len(bad)}")
    sys.exit(1 if bad else 0)
