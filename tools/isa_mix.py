"""Static instruction mix per kernel of a HIP source (device ISA via hipcc -S). usage: isa_mix.py file.hip [name-filter]"""
import collections
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/isa_mix.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                "-fhip-fp32-correctly-rounded-divide-sqrt", "-x", "hip", "-S",
                "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
cur, body = None, collections.defaultdict(list)
for line in open(out):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
    if cur and line.startswith("\t") and not line.strip().startswith((".", ";", "//")):
        body[cur].append(line.strip().split()[0])
for name, ins in body.items():
    dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem)
    if flt and flt not in dem:
        continue
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
    lds = sum(v for k, v in c.items() if k.startswith("ds_"))
    salu = sum(v for k, v in c.items() if k.startswith("s_"))
    print(f"== {dem}: total {len(ins)} valu {valu} salu {salu} vmem {vmem} lds {lds}")
    g = collections.Counter()
    for k, v in c.items():
        if k.startswith("v_"):
            g[re.sub(r"_e32|_e64|_dpp|_sdwa", "", k)] += v
    print("   ", ", ".join(f"{k}:{v}" for k, v in g.most_common(45)))
