"""Builds androidrenderer_amd/libsah_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m androidrenderer_amd.build [--force]

One object per source under androidrenderer_amd/_build/ (compiled in parallel, rebuilt when the source, any header of csrc/ or
include/, or the flags changed), then one link.  -ffp-contract=off is part of the numerics contract: every fp32
operator is individually rounded unless a kernel opts into contraction locally (the tolerance mode of the Lighting pass does, with
`#pragma clang fp contract(fast)` around the arithmetic it relaxes).
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get("SAH_HIP_CSRC") or os.path.join(HERE, "csrc")  # (experiments: a patched copy of the sources, tools/experiments/r4/variants.py)
INCLUDE = os.path.join(HERE, "..", "include")
OUT = os.environ.get("SAH_HIP_LIBRARY") or os.path.join(HERE, "libsah_hip.so")
OBJDIR = os.path.join(HERE, "_build", os.path.splitext(os.path.basename(OUT))[0])  # one object directory per output (A/B builds)
SOURCES = ["api.cpp", "api_post.cpp", "api_raster.cpp", "api_rt.cpp", "rt.hip", "api_ipc.cpp", "api_chain.cpp", "ipc.hip", "lighting.hip", "lighting_tiled.hip", "post.hip", "tonemap.hip", "tonemap_tol.hip", "lpv.hip",
           "probes.hip", "sky_luts.hip", "raster.hip", "vpl.hip"]
# -fno-slp-vectorize: on MI355X v_pk_{mul,add,fma}_f32 issue in 4 cycles against 2 for the scalar forms (profiles/r1_valu_issue_cost.txt),
# so the SLP vectoriser's packed pairs gain nothing and cost the register shuffles that feed them.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function", "-x", "hip"]
FLAGS += os.environ.get("SAH_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DSAH_EXP_...); never set by the driver
# Per-source options.  lighting_tiled.hip without the post-RA machine scheduler: its kernels are long unrolled chains (eight probes, the light
# loop) whose order the pre-RA scheduler already fixed around the loads; the post-RA pass re-orders them for a latency model that does not hold
# at four waves per SIMD and costs 1.8 % on the cache-GI kernel (0.3626 -> 0.3559 ms), 1.0-1.2 % on the light workloads
# (tools/experiments/r4/r4_sched.sh: eight scheduling options measured; on lighting.hip every one of them loses).  Same instructions, other order.
# tonemap*.hip with LLVM's wave-priority pass (s_setprio raised until a wave's loads are out): the composite's waves stage texels from
# global memory before each filter stage, and the ones still issuing loads then go ahead of the ones that are filtering — tolerance
# composite 0.1824 -> 0.1793 ms, strict 0.2536 -> 0.2503 (r4_sched3.sh, two alternating runs each); nothing for the ray tracer or the bloom.
SOURCE_FLAGS = {"lighting_tiled.hip": ["-mllvm", "-enable-post-misched=0"],
                "tonemap_tol.hip": ["-mllvm", "-amdgpu-set-wave-priority"], "tonemap.hip": ["-mllvm", "-amdgpu-set-wave-priority"]}


def _headers_digest():
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(SOURCE_FLAGS.items()))).encode())
    for d in (CSRC, INCLUDE):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hpp", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def _stale(src, obj, stamp, digest):
    if not os.path.exists(obj) or not os.path.exists(stamp):
        return True
    if open(stamp).read() != digest:
        return True
    return os.path.getmtime(src) > os.path.getmtime(obj)


def needs_build():
    if not os.path.exists(OUT):
        return True
    digest = _headers_digest()
    for s in SOURCES:
        obj = os.path.join(OBJDIR, s + ".o")
        if _stale(os.path.join(CSRC, s), obj, obj + ".stamp", digest) or os.path.getmtime(obj) > os.path.getmtime(OUT):
            return True
    return False


def _compile(hipcc, src, obj, verbose):
    cmd = [hipcc] + FLAGS + SOURCE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJDIR, exist_ok=True)
    digest = _headers_digest()
    todo = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s + ".o")
        if force or _stale(src, obj, obj + ".stamp", digest):
            todo.append((src, obj))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as ex:
        for f in [ex.submit(_compile, hipcc, src, obj, verbose) for src, obj in todo]:
            f.result()
    for _, obj in todo:
        open(obj + ".stamp", "w").write(digest)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(OBJDIR, s + ".o") for s in SOURCES] + ["-o", OUT, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
