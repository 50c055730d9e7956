"""Folds a PMC summary (tools/pmc_collect.sh -> tools/pmc_summary.py text, as committed under profiles/) into
profiles/roofline_static.json, the per-workload record bench.py's `roofline` object quotes beside its live numbers.

    python tools/pmc_to_static.py <workload> <profiles/rN_pmc_*.txt> <kernel substring> <width> <height> [flops_per_px]

Per launch of the named kernel: VALU wave-instructions (SQ_INSTS_VALU), VALU issue cycles summed over the SIMDs (SQ_ACTIVE_INST_VALU
counts quad-cycles: x 4), HBM traffic (FETCH_SIZE is in KiB-like units of 1000 B on this stack and reports half of a wide streaming
read on gfx950 — MI355X_MICROARCH.md "HBM": doubled here — plus WRITE_SIZE).
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path, kernel):
    cur, vals = None, {}
    for line in open(path):
        if not line.startswith(" ") and not line.startswith("pmc group"):
            cur = line.strip()
            continue
        m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)", line)
        if m and cur and kernel in cur:
            vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    if not vals:
        raise SystemExit(f"no kernel matching '{kernel}' in {path}")
    name = max(vals, key=lambda k: vals[k].get("SQ_INSTS_VALU", 0))  # the dominant instantiation
    return name, vals[name]


def main():
    workload, path, kernel, w, h = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    flops = float(sys.argv[6]) if len(sys.argv) > 6 else None
    name, v = parse(path, kernel)
    out_path = os.path.join(ROOT, "profiles", "roofline_static.json")
    table = json.load(open(out_path)) if os.path.exists(out_path) else {}
    rec = {"kernel": name, "pixels": w * h, "source": os.path.relpath(path, ROOT),
           "valu_wave_insts_per_launch": v["SQ_INSTS_VALU"], "valu_active_cycles_per_launch": v["SQ_ACTIVE_INST_VALU"] * 4.0}
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rec["hbm_traffic_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1000.0
    if "GRBM_GUI_ACTIVE" in v:
        rec["xcd_cycles_per_launch"] = v["GRBM_GUI_ACTIVE"] / 8.0
    if flops:
        rec["flops_per_px"] = flops
    elif workload in table and "flops_per_px" in table[workload]:
        rec["flops_per_px"] = table[workload]["flops_per_px"]
    table[workload] = rec
    json.dump(table, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
