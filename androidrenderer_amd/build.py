"""Builds androidrenderer_amd/libsah_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m androidrenderer_amd.build [--force]

-ffp-contract=off is part of the numerics contract: every fp32 operator is individually rounded.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("SAH_HIP_LIBRARY") or os.path.join(HERE, "libsah_hip.so")
SOURCES = ["api.cpp", "api_post.cpp", "api_raster.cpp", "lighting.hip", "lighting_tiled.hip", "post.hip", "lpv.hip", "probes.hip", "sky_luts.hip",
           "raster.hip", "vpl.hip"]
# -fno-slp-vectorize: on MI355X v_pk_{mul,add,fma}_f32 issue in 4 cycles against 2 for the scalar forms (profiles/r1_valu_issue_cost.txt),
# so the SLP vectoriser's packed pairs gain nothing and cost the register shuffles that feed them.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function", "-x", "hip"]
FLAGS += os.environ.get("SAH_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DSAH_EXP_...); never set by the driver


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sah_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
