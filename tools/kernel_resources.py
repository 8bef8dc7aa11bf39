"""Per-kernel register / LDS / scratch use of a built object (dev tool): tools/kernel_resources.py attention.o [name filter]"""
import os, re, shutil, subprocess, sys, tempfile
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "/usr/bin/c++filt"
src = sys.argv[1]
if not os.path.exists(src):
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "self-attention-experiments-vision_amd", "csrc", src)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    dst = os.path.join(d, os.path.basename(src))
    shutil.copy(src, dst)
    subprocess.run([OBJDUMP, "--offloading", dst], check=True, capture_output=True)
    co = [f for f in os.listdir(d) if "gfx950" in f][0]
    notes = subprocess.run([READELF, "--notes", os.path.join(d, co)], check=True, capture_output=True, text=True).stdout
rows = []
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    rows.append((name, "0" if False else blk.split()[0], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"),
                 g("vgpr_spill_count"), g("sgpr_spill_count")))
names = subprocess.run([CXXFILT] + [r[0] for r in rows], check=True, capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':90s} agpr vgpr sgpr   lds scratch vspill sspill")
for n, r in zip(names, rows):
    n = re.sub(r"^void \(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)
    if flt in n:
        print(f"{n[:90]:90s} {r[1]:>4s} {r[2]:>4s} {r[3]:>4s} {r[4]:>5s} {r[5]:>7s} {r[6]:>6s} {r[7]:>6s}")
