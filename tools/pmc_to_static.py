"""Folds a PMC summary (tools/pmc_collect.sh -> tools/pmc_summary.py text, as committed under profiles/) into
profiles/roofline_static.json, the per-workload record bench.py's `roofline` object quotes beside its live numbers.

    python tools/pmc_to_static.py <workload> <profiles/rN_pmc_*.txt> <kernel substring> <width> <height> [flops_per_px]

Per launch of the named kernel: VALU wave-instructions (SQ_INSTS_VALU), rocprof's VALUBusy numerator (SQ_ACTIVE_INST_VALU counts
quad-cycles: x 4), HBM traffic (FETCH_SIZE is in KiB-like units of 1000 B on this stack and reports half of a wide streaming read on
gfx950 — MI355X_MICROARCH.md "HBM": doubled here — plus WRITE_SIZE).  The guide states the factor for wide coalesced reads; round 5 calibrated it for
the narrow accesses of this path as well (tools/microbench/fetch_calib.hip, profiles/r5_pmc_gather_calibration.txt): every L2 miss fetches one
128-byte line whatever the access width — 2-byte gathers and PCF footprints included — and FETCH_SIZE tallies it as 64 bytes, so x 2 holds there too.

SQ_ACTIVE_INST_VALU charges every instruction a whole quad-cycle (its mean is 4.09 cycles per instruction on every kernel here), while
MI355X issues fp32 mul / add / fma / mov in ~2.3 cycles (profiles/r1_valu_issue_cost.txt), so VALUBusy OVERSTATES the issue time of a
kernel made of those (it reads 1.14 for the tiled kernel).  The record therefore also carries a MODEL: the kernel's static VALU mix
(hipcc -S of its source, every instruction weighted once) priced with the measured issue costs gives cycles per VALU instruction,
times the measured SQ_INSTS_VALU.  bench.py quotes the model as valu_issue.frac and VALUBusy beside it.
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


# measured issue cost classes, cycles per wave-instruction per SIMD (profiles/r1_valu_issue_cost.txt)
FAST = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_mac_f32", "v_fmamk_f32", "v_fmaak_f32")
MID = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32",
       "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_bfe_u32", "v_and_or_b32", "v_or3_b32", "v_bfi_b32")
SLOW = ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_rcp_f16", "v_sqrt_f16", "v_rsq_f16", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32",
        "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_rcp_iflag_f32")
SOURCES = {"k_lighting_fast": "lighting.hip", "k_lighting_tiled": "lighting_tiled.hip", "k_tonemap": "tonemap.hip", "k_rtao": "rt.hip",
           "k_sun_shadow_mask": "rt.hip"}


def issue_cost(op):
    base = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$|_e64_dpp$", "", op)
    if base in FAST:
        return 2.3
    if base in MID:
        return 3.2
    if base in SLOW:
        return 8.3
    return 4.3  # fma_mix, min / max / med3, conversions, compares, cndmask, packed and f64 arithmetic, 64-bit shifts, mul_lo, ...


def static_cycles_per_valu_inst(kernel_name):
    """(cycles per VALU instruction, number of static VALU instructions) of the kernel's code as compiled with the library's flags"""
    import collections
    import subprocess
    sys.path.insert(0, ROOT)
    from androidrenderer_amd import build as B
    stem = re.sub(r"^void\s+", "", kernel_name).split("<")[0].split("::")[-1]
    src = SOURCES.get(stem)
    if not src:
        return None, 0
    flags = [f for f in B.FLAGS if f not in ("-fPIC",)]
    out = f"/tmp/static_{stem}.s"
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    want = re.sub(r"^void\s+", "", kernel_name).replace(" ", "")
    cur, bodies = None, collections.defaultdict(list)
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        if cur and line.startswith("\t") and not line.strip().startswith((".", ";", "//")):
            bodies[cur].append(line.strip().split()[0])
    for sym, ins in bodies.items():
        dem = subprocess.run(["c++filt", sym], stdout=subprocess.PIPE, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void\s+", "", dem)).split("(")[0].replace(" ", "")
        if dem == want:
            v = [i for i in ins if i.startswith("v_")]
            return (sum(issue_cost(i) for i in v) / len(v), len(v)) if v else (None, 0)
    return None, 0


def main():
    workload, path, kernel, w, h = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    flops = float(sys.argv[6]) if len(sys.argv) > 6 else None
    name, v = parse(path, kernel)
    out_path = os.path.join(ROOT, "profiles", "roofline_static.json")
    table = json.load(open(out_path)) if os.path.exists(out_path) else {}
    rec = {"kernel": name, "pixels": w * h, "source": os.path.relpath(path, ROOT),
           "valu_wave_insts_per_launch": v["SQ_INSTS_VALU"], "valu_active_cycles_per_launch": v["SQ_ACTIVE_INST_VALU"] * 4.0}
    cpi, n_static = static_cycles_per_valu_inst(name)
    if cpi:
        rec["valu_model_cycles_per_inst"] = round(cpi, 3)
        rec["valu_model_static_insts"] = n_static
        rec["valu_model_issue_cycles_per_launch"] = v["SQ_INSTS_VALU"] * cpi
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
