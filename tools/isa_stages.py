"""Static per-STAGE instruction budget of one kernel: every instruction of the kernel's ISA is attributed, through the inline chain
llvm-symbolizer reports for its address, to the first stage whose rule matches a frame of that chain (innermost frame first), and priced with
the measured issue costs of MI355X (profiles/r1_valu_issue_cost.txt).  The counts are STATIC: an instruction inside a loop or under a
wave-uniform branch counts once, whatever its trip count — read them beside a dynamic count (SQ_INSTS_VALU).

usage: isa_stages.py <source.hip> <kernel name substring, demangled> <stage file.json> [--per N] [--extra-flags ...]

stage file: {"stages": [{"name": "...", "match": [rule, ...]}, ...]}; rule = {"func": substring of a frame's function, optional "file":
basename, optional "lines": [first, last]}.  Stages are tried in order; `--per N` divides the totals by N (pixels per thread)."""
import collections
import json
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin/"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-gline-tables-only", "-x", "hip", "-c", "--cuda-device-only", "--no-gpu-bundle-output"]

# issue cycles of one wave64 instruction when nothing else limits (tools/microbench/valu_rate.hip)
FULL_RATE = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mov_b32", "v_add_f16", "v_mul_f16",
             "v_sub_f16", "v_fma_f16", "v_fmac_f16", "v_mad_f32", "v_fmaak_f32", "v_fmamk_f32")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def issue_cycles(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith(TRANS):
        return 8.0
    if base in FULL_RATE:
        return 2.3
    if base.startswith(("v_add_u32", "v_sub_u32", "v_and_", "v_or_", "v_xor_", "v_lshl", "v_lshr", "v_ashr", "v_add_co", "v_sub_co", "v_subrev_u32", "v_add3",
                        "v_lshl_add", "v_and_or", "v_or3", "v_bfe", "v_bfi", "v_not", "v_add_lshl", "v_lshl_or", "v_subrev_co", "v_addc", "v_subb")):
        return 3.0
    return 4.3


def main():
    src, flt, stage_file = sys.argv[1], sys.argv[2], sys.argv[3]
    per = 1
    extra = []
    rest = sys.argv[4:]
    while rest:
        if rest[0] == "--per":
            per = int(rest[1])
            rest = rest[2:]
        elif rest[0] == "--extra-flags":
            extra = rest[1:]
            rest = []
        elif rest[0] == "--ops":
            rest = rest[1:]
        else:
            raise SystemExit(f"unknown argument {rest[0]}")
    stages = json.load(open(stage_file))["stages"]
    co = "/tmp/isa_stages.co"
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [src, "-o", co], check=True, stderr=subprocess.DEVNULL)
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, stdout=subprocess.PIPE, text=True).stdout
    cur, insts = None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            dem = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = dem if flt in dem else None
            continue
        if not cur:
            continue
        m = re.match(r"^\s+(\S+).*//\s*([0-9A-F]+):", line)
        if m:
            insts.append((int(m.group(2), 16), m.group(1)))
    if not insts:
        raise SystemExit(f"no kernel matches {flt!r}")
    sym = subprocess.run([LLVM + "llvm-symbolizer", f"--obj={co}", "--inlines", "--functions=short"], input="\n".join(hex(a) for a, _ in insts) + "\n",
                         stdout=subprocess.PIPE, text=True, check=True).stdout
    chains = []
    for block in sym.strip().split("\n\n"):
        lines = block.strip().split("\n")
        frames = []
        for i in range(0, len(lines) - 1, 2):
            m = re.match(r"(.*):(\d+):(\d+)$", lines[i + 1])
            frames.append((lines[i], m.group(1).split("/")[-1] if m else "?", int(m.group(2)) if m else 0))
        chains.append(frames)
    assert len(chains) == len(insts), (len(chains), len(insts))

    def stage_of(frames):
        for st in stages:
            for rule in st["match"]:
                for fn, fl, ln in frames:
                    if rule.get("func", "") not in fn:
                        continue
                    if "file" in rule and rule["file"] != fl:
                        continue
                    if "lines" in rule and not (rule["lines"][0] <= ln <= rule["lines"][1]):
                        continue
                    return st["name"]
        return "(unmatched)"
    ops = collections.defaultdict(collections.Counter)
    tab = collections.OrderedDict((st["name"], collections.Counter()) for st in stages)
    tab["(unmatched)"] = collections.Counter()
    unmatched = collections.Counter()
    for (addr, op), frames in zip(insts, chains):
        st = stage_of(frames)
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "lds" if op.startswith("ds_") else "other"
        tab[st][kind] += 1
        if kind == "valu":
            tab[st]["cycles"] += issue_cycles(op)
            ops[st][re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)] += 1
        if st == "(unmatched)" and kind == "valu":
            unmatched[(frames[0][1], frames[0][2], frames[-1][1], frames[-1][2])] += 1
    tot = collections.Counter()
    for c in tab.values():
        tot.update(c)
    print(f"# {flt}: static instructions per stage" + (f", divided by {per} (per pixel)" if per > 1 else "") + "; issue cycles = VALU priced at 2.3 / 3 / 4.3 / 8")
    print(f"{'stage':44s} {'VALU':>8s} {'%':>6s} {'issue cyc':>10s} {'%':>6s} {'VMEM':>6s} {'LDS':>6s} {'SALU':>6s}")
    for name, c in tab.items():
        if not sum(c.values()):
            continue
        print(f"{name:44s} {c['valu'] / per:8.1f} {100 * c['valu'] / max(tot['valu'], 1):6.1f} {c['cycles'] / per:10.1f} {100 * c['cycles'] / max(tot['cycles'], 1):6.1f} "
              f"{c['vmem'] / per:6.1f} {c['lds'] / per:6.1f} {c['salu'] / per:6.1f}")
    print(f"{'total':44s} {tot['valu'] / per:8.1f} {100.0:6.1f} {tot['cycles'] / per:10.1f} {100.0:6.1f} {tot['vmem'] / per:6.1f} {tot['lds'] / per:6.1f} {tot['salu'] / per:6.1f}")
    if "--ops" in sys.argv:
        for name, c in ops.items():
            print(f"# {name}: " + ", ".join(f"{k} {v / per:g}" for k, v in c.most_common(14)))
    if unmatched:
        print("# unmatched VALU by (innermost file:line, outermost file:line):")
        for k, n in unmatched.most_common(12):
            print(f"#   {n:5d}  {k[0]}:{k[1]}  <- {k[2]}:{k[3]}")


if __name__ == "__main__":
    main()
