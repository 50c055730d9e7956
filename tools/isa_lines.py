"""Static VALU instruction count per source line of one kernel (device ISA via hipcc -S -gline-tables-only).
usage: isa_lines.py file.hip "kernel-name-filter" [top_n]"""
import collections
import re
import subprocess
import sys

src, flt = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
out = "/tmp/isa_lines.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                "-fhip-fp32-correctly-rounded-divide-sqrt", "-gline-tables-only", "-x", "hip", "-S", "--cuda-device-only", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
files = {}
cur, loc = None, None
per_line = collections.Counter()
total = 0
for line in open(out):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r"^(_Z\w+):", line)
    if m:
        dem = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = m.group(1) if flt in dem else None
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
    if not cur:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    s = line.strip()
    if line.startswith("\t") and s.startswith("v_"):
        per_line[loc] += 1
        total += 1
print("total VALU", total)
per_file = collections.Counter()
for (f, l), n in per_line.items():
    per_file[f] += n
print(dict(per_file))
for (f, l), n in per_line.most_common(top):
    print(f"{n:5d}  {f}:{l}")
