"""Offline exposure list for the packed-fp32 hazard of DESIGN section 5: disassemble every code object of
csrc/*.o and list, per kernel, the IN-PLACE packed fp32 instructions with an op_sel / op_sel_hi modifier -

    v_pk_{add,mul,fma}_f32 vD, ..., op_sel:[..]      with vD[0:1] also a source operand

- the one instruction form whose low-half result was lost now and then in round 1's LayerNorm forward when two PROCESSES time-sliced one
GPU (never with one process per GPU).  No GPU is needed: this reads ISA.   python tools/pk_f32_scan.py [--all]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "self-attention-experiments-vision_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK = re.compile(r"\b(v_pk_(?:add|mul|fma)_f32)\s+(v\[\d+:\d+\])\s*,\s*(.*)")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def scan(obj):
    res = {}
    with tempfile.TemporaryDirectory() as td:
        dst = os.path.join(td, os.path.basename(obj))
        subprocess.run(["cp", obj, dst], check=True)
        subprocess.run([OBJDUMP, "--offloading", dst], check=True, capture_output=True)
        cos = [f for f in os.listdir(td) if "gfx950" in f]
        if not cos:
            return res
        isa = subprocess.run([OBJDUMP, "-d", os.path.join(td, cos[0])], check=True, capture_output=True, text=True).stdout
    cur = None
    for ln in isa.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = m.group(1)
            res[cur] = {"pk": 0, "inplace_opsel": 0}
            continue
        m = PK.search(ln)
        if m and cur:
            res[cur]["pk"] += 1
            dst_reg, rest = m.group(2), m.group(3).split("//")[0]
            if "op_sel" in rest and dst_reg in rest.split("op_sel")[0]:
                res[cur]["inplace_opsel"] += 1
    return res


def main():
    show_all = "--all" in sys.argv
    rows = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".o"):
            for k, v in scan(os.path.join(CSRC, f)).items():
                rows.append((f, k, v["pk"], v["inplace_opsel"]))
    names = demangle([r[1] for r in rows])
    print(f"{'object':22s} {'packed fp32':>11s} {'in-place+op_sel':>15s}  kernel")
    tot = 0
    for f, k, pk, ip in rows:
        if ip or (show_all and pk):
            nm = re.sub(r"\(anonymous namespace\)::", "", names.get(k, k))
            print(f"{f:22s} {pk:11d} {ip:15d}  {nm[:150]}")
        tot += ip
    print(f"kernels scanned: {len(rows)}; with the in-place op_sel form: {sum(1 for r in rows if r[3])}; instructions of that form: {tot}")


if __name__ == "__main__":
    main()
