"""Print per-kernel register/LDS/occupancy usage for a HIP source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-x", "hip", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\d+:\d+:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    print(f"{name:60s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('TotalSGPRs', r.get('SGPRs','?')):>4} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>5} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?'):>6}")
