"""Timing-only ablation builds of the two lighting kernels ("a per-stage budget").  Each variant is the product
source with ONE stage replaced by a stand-in that keeps the data flow alive (so that nothing else is optimised away) — its images are
wrong on purpose.  The patches live here, not in csrc/: a variant is built from a patched COPY of the sources into build_ab/<name>.so,
which androidrenderer_amd/lib.py loads when SAH_HIP_LIBRARY points at it.

    python tools/experiments/r4/variants.py            # build every variant (needs an up-to-date base build)
    python tools/experiments/r4/variants.py NAME ...   # some of them
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from androidrenderer_amd import build as base_build  # noqa: E402

CSRC = os.path.join(ROOT, "androidrenderer_amd", "csrc")
OUT = os.path.join(ROOT, "build_ab")
FAST = ["lighting.hip", "lighting_tiled.hip"]  # the two translation units that include the lighting headers

# name -> (sources to recompile, [(file, old, new), ...])
VARIANTS = {
    # ---- k_lighting_fast<CSM, LPV, 4> ------------------------------------------------------------------------------------------------
    "fast_no_pcf": (["lighting.hip"], [("lighting_fast.hpp", "        float pcf = __builtin_fmaf(wx0 * wy0, (pcf_ref < dtap[0]) ? 1.0f : 0.0f, 0.0f);\n"
                                        "        pcf = __builtin_fmaf(pcf_fx * wy0, (pcf_ref < dtap[1]) ? 1.0f : 0.0f, pcf);\n"
                                        "        pcf = __builtin_fmaf(wx0 * pcf_fy, (pcf_ref < dtap[2]) ? 1.0f : 0.0f, pcf);\n"
                                        "        pcf = __builtin_fmaf(pcf_fx * pcf_fy, (pcf_ref < dtap[3]) ? 1.0f : 0.0f, pcf);\n",
                                        "        float pcf = pcf_ref < 0.5f ? wx0 * wy0 : 1.0f;\n"),
                                       ("lighting_fast.hpp", "        for (int k = 0; k < 4; k++) raw[k] = *reinterpret_cast<const uint16_t*>(sm.ptr + pcf_off[k]);",
                                        "        for (int k = 0; k < 4; k++) raw[k] = (uint16_t)(pcf_off[k] >> 3);")]),
    "fast_no_brdf": (["lighting.hip"], [("lighting_fast.hpp", "            const F3 b = brdf_fast(s, L, V, brdf_out_of_domain);",
                                         "            brdf_out_of_domain = false;\n            const F3 b = s.base_color * dot(s.normal, V);")]),
    "fast_no_lpv_taps": (["lighting.hip"], [("lighting_common.hpp", "            const uint4 q = *reinterpret_cast<const uint4*>(packed + ro[r] + 16u * (uint32_t)j);",
                                             "            const uint4 q = make_uint4(ro[r], ro[r] + (uint32_t)j, ro[r] ^ 0x3c00u, 0x3c003c00u);"),
                                            ("lighting_common.hpp", "        for (int k = 0; k < 8; k++) {  // tap order: x fastest, then y, then z",
                                             "        for (int k = 0; k < 1; k++) {  // ABLATION: one tap")]),
    "fast_no_lpv_loads": (["lighting.hip"], [("lighting_common.hpp", "            const uint4 q = *reinterpret_cast<const uint4*>(packed + ro[r] + 16u * (uint32_t)j);",
                                              "            const uint4 q = make_uint4(ro[r], ro[r] + (uint32_t)j, ro[r] ^ 0x3c00u, 0x3c003c00u);")]),
    "fast_no_lpv_select": (["lighting.hip"], [("lighting_fast.hpp", "        if (!__all(in0 || !ok || sky_px)) {", "        if (false) {")]),
    "fast_no_geometry": (["lighting.hip"], [("lighting_fast.hpp", "        const FastGeom g = fast_geometry(a, f, colx_glsl, rowy_glsl, D, si, dn, tab, ok);",
                                             "        FastGeom g;\n        {\n            const float zz = 0.05f / D;\n"
                                             "            g.N = F3{Fn(si.normal[0]), Fn(si.normal[1]), Fn(si.normal[2])};\n"
                                             "            g.ws = F3{Fn(-7.0f + zz), Fn(1.0f - rowy_glsl * zz), Fn(colx_glsl * zz)};\n"
                                             "            g.V = F3{Fn(0.8f), Fn(rowy_glsl), Fn(colx_glsl)};\n            g.vsz = Fn(-zz);\n        }")]),
    "fast_no_deferral": (["lighting.hip"], [("lighting.hip", "        const bool mine = ((deferred_mask >> i) & 1u) && !(SKY && ((sky_mask >> i) & 1u));",
                                             "        const bool mine = false;")]),
    "fast_no_decode": (["lighting.hip"], [("lighting_common.hpp", "    s.color[0] = lut[p.color & 0xffu];\n    s.color[1] = lut[(p.color >> 8) & 0xffu];\n    s.color[2] = lut[(p.color >> 16) & 0xffu];",
                                           "    s.color[0] = __uint_as_float(0x3e000000u | (p.color << 8));\n    s.color[1] = __uint_as_float(0x3e000000u | p.color);\n"
                                           "    s.color[2] = __uint_as_float(0x3e000000u | (p.color >> 8));"),
                                          ("lighting_common.hpp", "    s.rough = lut[256 + ((p.data >> 8) & 0xffu)];\n    s.metal = lut[256 + ((p.data >> 16) & 0xffu)];",
                                           "    s.rough = __uint_as_float(0x3e000000u | p.data);\n    s.metal = __uint_as_float(0x3e000000u | (p.data >> 8));")]),
    # ---- k_lighting_tiled<RT, CACHE> --------------------------------------------------------------------------------------------------
    "tiled_no_rt_sun": (["lighting_tiled.hip"], [("lighting_tiled.hip", "            sun_rt(a, x, y, p, si, add);", "            add[0] = add[1] = add[2] = p.mask;")]),
    "tiled_no_cache_brdf": (["lighting_tiled.hip"], [("lighting_gi_ext.hpp", "    const H3 b = brdf_sl(s, s.normal, V);  // == Fd(s, N, V) + Fr(s, N, V)",
                                                      "    const H3 b = s.base_color * dot(s.normal, V);")]),
    "tiled_no_cheb": (["lighting_tiled.hip"], [("lighting_gi_ext.hpp", "        Fn cheb = Fn(div_nr(variance.v, cden.v));\n        cheb = nmax(cheb * cheb * cheb, Fn(0.f));\n        cheb = behind ? cheb : Fn(1.f);\n"
                                                "        probe_weight = probe_weight * nmax(Fn(0.05f), cheb);\n        probe_weight = nmax(Fn(0.000001f), probe_weight);\n"
                                                "        const Fn crush = Fn(0.2f);\n        const Fn crushed = probe_weight * ((probe_weight * probe_weight) * (Fn(1.f) / (crush * crush)));\n"
                                                "        probe_weight = probe_weight.v < crush.v ? crushed : probe_weight;\n",
                                                "        probe_weight = behind ? cden : variance;\n")]),
    "tiled_no_depth_dir": (["lighting_tiled.hip"], [("lighting_gi_ext.hpp", "            const Fn l1 = nabs(dir_to_probe.x) + nabs(dir_to_probe.y) + nabs(dir_to_probe.z);\n"
                                                     "            if (i == 0) pbad = !(l1.v > 0.f);  // the point ON the probe: no direction; every other corner has a component >= 2^-40\n"
                                                     "            const Fn inv = Fn(rcp_nr(l1.v));\n            const F2 uv = {-dir_to_probe.x * inv, -dir_to_probe.y * inv};\n"
                                                     "            if (jz == 0) {\n                depth_oct = uv;\n            } else {\n"
                                                     "                const Fn rx = Fn(1.f) - nabs(uv.y), ry = Fn(1.f) - nabs(uv.x);\n"
                                                     "                depth_oct = {jx == 0 ? rx : -rx, jy == 0 ? ry : -ry};\n            }\n",
                                                     "            depth_oct = {dir_to_probe.x, dir_to_probe.y};\n")]),
    "tiled_no_depth_lookup": (["lighting_tiled.hip"], [("lighting_gi_ext.hpp", "        const uint2 d0 = load_pair(c.depth.ptr, drow0), d1 = load_pair(c.depth.ptr, drow0 + c.depth.row_pitch);\n"
                                                        "        const uint32_t dw[4] = {d0.x, d0.y, d1.x, d1.y};  // tap order (x0,y0) (x1,y0) (x0,y1) (x1,y1)\n"
                                                        "        const float dwt[4] = {dxa.w0 * dya.w0, dxa.w1 * dya.w0, dxa.w0 * dya.w1, dxa.w1 * dya.w1};\n"
                                                        "        float dt0 = 0.f, dt1 = 0.f;\n#pragma unroll\n        for (int k = 0; k < 4; k++) {\n"
                                                        "            dt0 = fma_mix_lo(dwt[k], dw[k], dt0);\n            dt1 = fma_mix_hi(dwt[k], dw[k], dt1);\n        }\n",
                                                        "        float dt0 = dxa.w0 + __uint_as_float(drow0 | 0x3f000000u), dt1 = dya.w0 * dt0;\n")]),
    "tiled_no_irr_taps": (["lighting_tiled.hip"], [("lighting_gi_ext.hpp", "        const float4 it[4] = {irow[0], irow[1], irow_next[0], irow_next[1]};",
                                                    "        const float tt = __uint_as_float((irow0 & 0xffffu) | 0x3f000000u);\n"
                                                    "        const float4 it[4] = {make_float4(tt, tt, tt, 0.f), make_float4(tt, 1.f, tt, 0.f), make_float4(1.f, tt, tt, 0.f), make_float4(tt, tt, 1.f, 0.f)};\n"
                                                    "        (void)irow; (void)irow_next;")]),
    "tiled_no_pixel_setup": (["lighting_tiled.hip"], [("lighting_gi_ext.hpp", "    const F3 location = worldspace_location_slang(a, (float)x, (float)y, p.depth);\n"
                                                       "    const H3 V = to_h(normalize(location - F3{Fn(a.view_pos[0]), Fn(a.view_pos[1]), Fn(a.view_pos[2])}));\n    uint32_t cascade_index = 5;",
                                                       "    const float zz = 0.05f / p.depth;\n"
                                                       "    const F3 location = {Fn(-7.0f + zz), Fn(1.0f - ((float)y * (1.0f / 1080.0f) - 1.0f) * 0.767f * zz), Fn(((float)x * (1.0f / 1920.0f) - 1.0f) * 1.364f * zz)};\n"
                                                       "    const H3 V = to_h(F3{Fn(0.8f), Fn(0.1f), Fn(0.2f)});\n    uint32_t cascade_index = 5;")]),
}
# ---- scheduling experiments on the headline kernel (results stay correct: same operators, other order / fewer wave votes) ----
VARIANTS["fast_lpv_first"] = (["lighting.hip"], [
    ("lighting_fast.hpp", "    // ---------------- a1: sun, CSM mode ----------------\n    // direct = ((ndotl * brdf) * colour) * shadow is exactly 0",
     "    Fn indirect[3];\n    if constexpr (GI == SAH_GI_LPV) lpv_fetch_packed(lpv, f.lpv_packed, f.pk_row_pitch, f.pk_slice_pitch, lpv_u, lpv_v, lpv_w, nc, indirect);\n"
     "    // ---------------- a1: sun, CSM mode ----------------\n    // direct = ((ndotl * brdf) * colour) * shadow is exactly 0"),
    ("lighting_fast.hpp", "        Fn indirect[3];\n        lpv_fetch_packed(lpv, f.lpv_packed, f.pk_row_pitch, f.pk_slice_pitch, lpv_u, lpv_v, lpv_w, nc, indirect);\n", "")])
VARIANTS["fast_no_vote1"] = (["lighting.hip"], [("lighting_fast.hpp", "    if (__any(ok && !sky_px && ndotl_sun.v > 0.f)) {\n        uint32_t cascade = 0;", "    {\n        uint32_t cascade = 0;")])
VARIANTS["fast_no_votes"] = (["lighting.hip"], VARIANTS["fast_no_vote1"][1] + [
    ("lighting_fast.hpp", "        if (__any(ok && !sky_px && ndotl_sun.v > 0.f && shadow != 0.0f)) {", "        {")])
VARIANTS["fast_lpv_first_no_vote1"] = (["lighting.hip"], VARIANTS["fast_lpv_first"][1] + VARIANTS["fast_no_vote1"][1])

VARIANTS["tm_hoist"] = (["tonemap_tol.hip"], [("tonemap_tol.hip", "#define SAH_TM_HOIST 0", "#define SAH_TM_HOIST 1")])

# ---- a9 light loop: what the builder-owned spec choices cost (results differ from the spec: timing only) ----
VARIANTS["lights_pow5_f32"] = (["lighting_tiled.hip"], [
    ("numerics.hpp", "    const Fn light_scatter = one + (f90 - one) * npow5(nclamp(one - NoL, zero, one));",
     "    const Fn ls_u = nclamp(one - NoL, zero, one), ls_u2 = ls_u * ls_u;\n    const Fn light_scatter = one + (f90 - one) * (ls_u2 * ls_u2 * ls_u);"),
    ("numerics.hpp", "    const F3 Fv = F_Schlick(VoH, p.f0, one);\n    // V_SmithGGXCorrelated\n    const Fn argL = (-NoL * p.a2 + NoL) * NoL + p.a2;",
     "    const Fn fv_u = nclamp(one - VoH, zero, one), fv_u2 = fv_u * fv_u, fv_p = fv_u2 * fv_u2 * fv_u;\n"
     "    const F3 Fv = {p.f0.x + (one - p.f0.x) * fv_p, p.f0.y + (one - p.f0.y) * fv_p, p.f0.z + (one - p.f0.z) * fv_p};\n"
     "    // V_SmithGGXCorrelated\n    const Fn argL = (-NoL * p.a2 + NoL) * NoL + p.a2;")])
VARIANTS["lights_inv_r"] = (["lighting_tiled.hip"], [("lighting_gi_ext.hpp", "    const Fn xr = Fn(div_nr(dist.v, pl.radius));", "    const Fn xr = dist * Fn(__builtin_amdgcn_rcpf(pl.radius));")])
VARIANTS["lights_both"] = (["lighting_tiled.hip"], VARIANTS["lights_pow5_f32"][1] + VARIANTS["lights_inv_r"][1])

VARIANTS["bloom_big_tiles_mip1"] = (["post.hip"], [("post.hip", "if ((uint64_t)cols * ((rows + 15) / 16) >= 1024) {", "if ((uint64_t)cols * ((rows + 15) / 16) >= 500) {")])

# compound variants
# ---- k_copy_bloom_mip0 (copy scene + bloom mip 0 in one pass): what do the copy's four taps per cell, the antialiased stores and the
# filter cost? -------------------------------------------------------------------------------------------------------------------------
VARIANTS["cbm_one_tap"] = (["post.hip"], [("post.hip", "                t[q][1] = *reinterpret_cast<const uint2*>(lit + (cy[q].o0 + cx[q].o1));\n"
                                           "                t[q][2] = *reinterpret_cast<const uint2*>(lit + (cy[q].o1 + cx[q].o0));\n"
                                           "                t[q][3] = *reinterpret_cast<const uint2*>(lit + (cy[q].o1 + cx[q].o1));\n",
                                           "                t[q][1] = t[q][2] = t[q][3] = t[q][0];\n")])
VARIANTS["cbm_no_aa_store"] = (["post.hip"], [("post.hip", "                if (ax >= own_x0 && ax < own_x1 && ay >= own_y0 && ay < own_y1)\n", "                if (ax == -77)\n")])
VARIANTS["cbm_no_filter"] = (["post.hip"], [("post.hip", "    // 3. the mip 0 tile, as k_bloom_downsample_lds filters it\n    const uint32_t col = tid & 63u, x = bx + col;\n    if (x >= dw) return;",
                                             "    // 3. the mip 0 tile, as k_bloom_downsample_lds filters it\n    const uint32_t col = tid & 63u, x = bx + col;\n    if (x >= dw || g.mw != 77u) return;")])
VARIANTS["cbm_one_tap_no_filter"] = (["post.hip"], VARIANTS["cbm_one_tap"][1] + VARIANTS["cbm_no_filter"][1])
# the tiled kernel WITHOUT a light list as 64 x 1 pixel strips per wave (64 x 4 per workgroup) instead of 8 x 8 squares in a 16 x 16 tile
for _n, _xy, _grid in (("tiled_strips_64x1", ("blockIdx.x * 64u + (threadIdx.x & 63u)", "blockIdx.y * 4u + (threadIdx.x >> 6)"), "dim3((a.width + 63) / 64, (rows + 3) / 4)"),
                       ("tiled_strips_32x2", ("blockIdx.x * 32u + (threadIdx.x & 31u)", "blockIdx.y * 8u + (threadIdx.x >> 5)"), "dim3((a.width + 31) / 32, (rows + 7) / 8)"),
                       ("tiled_strips_16x4", ("blockIdx.x * 16u + (threadIdx.x & 15u)", "blockIdx.y * 16u + (threadIdx.x >> 4)"), "dim3((a.width + 15) / 16, (rows + 15) / 16)")):
    VARIANTS[_n] = (["lighting_tiled.hip"], [
        ("lighting_tiled.hip", "    const uint32_t x = blockIdx.x * 16u + (threadIdx.x & 7u) + ((threadIdx.x >> 3) & 8u);\n"
         "    const uint32_t y = a.row_begin + blockIdx.y * 16u + ((threadIdx.x >> 3) & 7u) + ((threadIdx.x >> 4) & 8u);",
         "    const uint32_t x = LIGHTS ? blockIdx.x * 16u + (threadIdx.x & 7u) + ((threadIdx.x >> 3) & 8u) : " + _xy[0] + ";\n"
         "    const uint32_t y = a.row_begin + (LIGHTS ? blockIdx.y * 16u + ((threadIdx.x >> 3) & 7u) + ((threadIdx.x >> 4) & 8u) : " + _xy[1] + ");"),
        ("lighting_tiled.hip", "    else hipLaunchKernelGGL((k_lighting_tiled<SUN, GI, false>), grid, block,", "    else hipLaunchKernelGGL((k_lighting_tiled<SUN, GI, false>), " + _grid + ", block,")])
# the tiled kernel WITH a light list: waves of 16 x 4 pixels of the 16 x 16 tile instead of 8 x 8 squares
VARIANTS["tiled_lights_16x4"] = (["lighting_tiled.hip"], [
    ("lighting_tiled.hip", "    const uint32_t x = LIGHTS ? blockIdx.x * 16u + (threadIdx.x & 7u) + ((threadIdx.x >> 3) & 8u) : blockIdx.x * 32u + (threadIdx.x & 31u);",
     "    const uint32_t x = LIGHTS ? blockIdx.x * 16u + (threadIdx.x & 15u) : blockIdx.x * 32u + (threadIdx.x & 31u);"),
    ("lighting_tiled.hip", "    const uint32_t y = a.row_begin + (LIGHTS ? blockIdx.y * 16u + ((threadIdx.x >> 3) & 7u) + ((threadIdx.x >> 4) & 8u) : blockIdx.y * 8u + (threadIdx.x >> 5));",
     "    const uint32_t y = a.row_begin + (LIGHTS ? blockIdx.y * 16u + (threadIdx.x >> 4) : blockIdx.y * 8u + (threadIdx.x >> 5));")])
VARIANTS["tm_always16"] = (["tonemap_tol.hip"], [("tonemap_tol.hip", "    if ((uint64_t)cols * ((rows + 31) / 32) >= 2 * 768) hipLaunchKernelGGL", "    if (false) hipLaunchKernelGGL")])
VARIANTS["tiled_skeleton"] = (["lighting_tiled.hip"], VARIANTS["tiled_no_cheb"][1] + VARIANTS["tiled_no_depth_dir"][1] + VARIANTS["tiled_no_depth_lookup"][1] +
                              VARIANTS["tiled_no_irr_taps"][1])
VARIANTS["tiled_no_lookups"] = (["lighting_tiled.hip"], VARIANTS["tiled_no_depth_lookup"][1] + VARIANTS["tiled_no_irr_taps"][1])


# ---- compiler scheduling options on the two lighting translation units (same source, same bits: timing only) ----------------------------
FLAG_VARIANTS = {
    "sched_max_ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "sched_max_clause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "sched_iter_ilp": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    "sched_iter_minreg": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"],
    "sched_bias0": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
    "sched_bias100": ["-mllvm", "-amdgpu-schedule-metric-bias=100"],
    "sched_no_post": ["-mllvm", "-enable-post-misched=0"],
    "sched_relaxed_occ": ["-mllvm", "-amdgpu-schedule-relaxed-occupancy=true"],
}
for _n in FLAG_VARIANTS:
    VARIANTS[_n] = (FAST, [])
# the tiled translation unit alone
for _n, _f in (("tiled_no_post", ["-mllvm", "-enable-post-misched=0"]),
               ("tiled_no_post_clause", ["-mllvm", "-enable-post-misched=0", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]),
               ("tiled_no_post_ilp", ["-mllvm", "-enable-post-misched=0", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
               ("tiled_clause", ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"])):
    FLAG_VARIANTS[_n] = _f
    VARIANTS[_n] = (["lighting_tiled.hip"], [])


for _n, _f, _src in (("prio_fast", ["-mllvm", "-amdgpu-set-wave-priority"], ["lighting.hip"]),
                     ("prio_tiled", ["-mllvm", "-enable-post-misched=0", "-mllvm", "-amdgpu-set-wave-priority"], ["lighting_tiled.hip"]),
                     ("prio_both", ["-mllvm", "-amdgpu-set-wave-priority"], ["lighting.hip"]),
                     ("o2_fast", ["-O2"], ["lighting.hip"]),
                     ("nocluster_fast", ["-mllvm", "-misched-cluster=0"], ["lighting.hip"]),
                     ("revlocal_fast", ["-mllvm", "-greedy-reverse-local-assignment"], ["lighting.hip"]),
                     ("clause4_fast", ["-mllvm", "-amdgpu-max-memory-clause=4"], ["lighting.hip"]),
                     ("prio_post", ["-mllvm", "-amdgpu-set-wave-priority"], ["post.hip", "tonemap_tol.hip"]),
                     ("prio_tm", ["-mllvm", "-amdgpu-set-wave-priority"], ["tonemap_tol.hip"]),
                     ("prio_np_tm", ["-mllvm", "-amdgpu-set-wave-priority", "-mllvm", "-enable-post-misched=0"], ["tonemap_tol.hip"]),
                     ("prio_tm_strict", ["-mllvm", "-amdgpu-set-wave-priority"], ["tonemap.hip"]),
                     ("prio_rt", ["-mllvm", "-amdgpu-set-wave-priority"], ["rt.hip"]),
                     ("tm_prio_ilp", ["-mllvm", "-amdgpu-set-wave-priority", "-mllvm", "-amdgpu-sched-strategy=max-ilp"], ["tonemap_tol.hip"]),
                     ("tm_prio_clause", ["-mllvm", "-amdgpu-set-wave-priority", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause"], ["tonemap_tol.hip"]),
                     ("tm_prio_bias100", ["-mllvm", "-amdgpu-set-wave-priority", "-mllvm", "-amdgpu-schedule-metric-bias=100"], ["tonemap_tol.hip"]),
                     ("tm_prio_revlocal", ["-mllvm", "-amdgpu-set-wave-priority", "-mllvm", "-greedy-reverse-local-assignment"], ["tonemap_tol.hip"]),
                     ("post_ilp", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"], ["post.hip"]),
                     ("post_clause", ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"], ["post.hip"])):
    FLAG_VARIANTS[_n] = _f
    VARIANTS[_n] = (_src, [])
# ... and of every other translation unit with kernels in it, one at a time
for _src in ("post.hip", "tonemap_tol.hip", "tonemap.hip", "rt.hip", "raster.hip", "lpv.hip", "probes.hip"):
    _n = "np_" + _src.split(".")[0]
    FLAG_VARIANTS[_n] = ["-mllvm", "-enable-post-misched=0"]
    VARIANTS[_n] = ([_src], [])


def build_variant(name):
    sources, patches = VARIANTS[name]
    src_root = os.path.join(OUT, "src", name)
    csrc = os.path.join(src_root, "androidrenderer_amd", "csrc")
    shutil.rmtree(src_root, ignore_errors=True)
    shutil.copytree(CSRC, csrc)
    os.makedirs(os.path.join(src_root, "include"), exist_ok=True)
    for f in os.listdir(os.path.join(ROOT, "include")):
        shutil.copy(os.path.join(ROOT, "include", f), os.path.join(src_root, "include", f))
    for fname, old, new in patches:
        p = os.path.join(csrc, fname)
        s = open(p).read()
        if old not in s:
            raise SystemExit(f"variant {name}: the text to replace is not in {fname}:\n{old}")
        open(p, "w").write(s.replace(old, new, 1))
    objdir = os.path.join(OUT, "obj", name)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for s in base_build.SOURCES:
        if s in sources:
            obj = os.path.join(objdir, s + ".o")
            subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + base_build.FLAGS + (FLAG_VARIANTS[name] if name in FLAG_VARIANTS else base_build.SOURCE_FLAGS.get(s, [])) +
                                  ["-Wno-unused-variable", "-Wno-unused-but-set-variable", "-c", os.path.join(csrc, s), "-o", obj])
        else:
            obj = os.path.join(base_build.OBJDIR, s + ".o")  # the product build's object
            if not os.path.exists(obj):
                raise SystemExit("build the product library first (python -m androidrenderer_amd.build)")
        objs.append(obj)
    out = os.path.join(OUT, name + ".so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out, "-ldl"])
    shutil.rmtree(src_root, ignore_errors=True)
    return name


if __name__ == "__main__":
    names = sys.argv[1:] or sorted(VARIANTS)
    base_build.build()
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        for n in ex.map(build_variant, names):
            print("built", n, flush=True)
