"""Times every C-ABI pass at a given resolution on one GPU (HIP events on the launch stream) and prints achieved
algorithmic GB/s next to the 8 TB/s HBM roofline.  Diagnostic companion of bench.py (which owns the headline number).

    python tools/bench_passes.py [--width 3840 --height 2160 --iters 50]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--lights", type=int, default=256)
    ap.add_argument("--light-radius", type=float, default=4.0)
    ap.add_argument("--ramp-ms", type=float, default=150.0, help="untimed: each pass runs back to back for this long before its timed launches (the same clock "
                    "ramp bench.py gives its step: profiles/r3_clock_ramp.txt)")
    ap.add_argument("--only", default="", help="run only the passes whose name contains one of these comma-separated substrings")
    ap.add_argument("--json", action="store_true", help="also print the results as one JSON object")
    args = ap.parse_args()
    import torch

    from androidrenderer_amd import _abi, images, lib, synth
    from tests import util

    W, H = args.width, args.height
    ctx = lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    px = W * H
    results = {}

    def wanted(name):
        return not args.only or any(part.strip().lower() in name.lower() for part in args.only.split(","))

    def timeit(name, fn, bytes_per_call):
        if not wanted(name):
            return
        import time
        t_end = time.perf_counter() + args.ramp_ms * 1e-3
        while True:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            if time.perf_counter() >= t_end:
                break
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        results[name] = {"ms": round(ms, 4), "Mpx_s": round(px / ms / 1e3, 1), "GB_s": round(bytes_per_call / ms / 1e6, 1),
                         "hbm_frac": round(bytes_per_call / ms / 1e6 / 8000.0, 4)}
        print(f"{name:34s} {ms:9.4f} ms  {px / ms / 1e3:10.1f} Mpx/s  {bytes_per_call / ms / 1e6:8.1f} GB/s ({bytes_per_call / ms / 1e6 / 80.0:5.1f} % of 8 TB/s)", flush=True)

    # ---- lighting variants ----------------------------------------------------------------------------
    def lighting_case(name, sun, gi, flavour, bpp, lights=None, flags=_abi.LIGHTING_DEFAULT_FLAGS, general=False):
        if not wanted(name):
            return None
        f = util.LightingFrame(W, H, seed=2, sun_mode=sun, gi=gi, flavour=flavour, shadowmap_res=4096, lights=lights, flags=flags)
        dev = f.device_arrays()
        lit = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
        d, keep = f.describe(dev, lit)
        ctx.debug_set(force_general=general)
        timeit(name, lambda: ctx.lighting(d), bpp * px)
        ctx.debug_set(force_general=False)
        return lit

    lighting_case("lighting CSM+LPV atrium (fast)", _abi.SHADOW_MODE_CSM, _abi.GI_LPV, "atrium", 36)
    lighting_case("lighting CSM+LPV random (fast)", _abi.SHADOW_MODE_CSM, _abi.GI_LPV, "random", 36)
    lighting_case("lighting CSM only atrium (fast)", _abi.SHADOW_MODE_CSM, _abi.GI_NONE, "atrium", 32)
    lighting_case("lighting RT only atrium (fast)", _abi.SHADOW_MODE_RT, _abi.GI_NONE, "atrium", 36)
    lighting_case("lighting RT+LPV atrium (fast)", _abi.SHADOW_MODE_RT, _abi.GI_LPV, "atrium", 40)
    lighting_case("lighting off/none atrium (fast)", _abi.SHADOW_MODE_OFF, _abi.GI_NONE, "atrium", 32)
    lighting_case("lighting RT+cache atrium (tiled)", _abi.SHADOW_MODE_RT, _abi.GI_CACHE, "atrium", 36)
    lighting_case("lighting RT+rtgi atrium (tiled)", _abi.SHADOW_MODE_RT, _abi.GI_RTGI, "atrium", 52)
    from androidrenderer_amd import scene
    view = scene.SceneView.default(W, H)
    pl = synth.point_lights(view, args.lights, args.light_radius, seed=8)
    lighting_case(f"lighting CSM + {args.lights} lights (tiled)", _abi.SHADOW_MODE_CSM, _abi.GI_NONE, "atrium", 32, lights=pl)

    # ---- post chain -------------------------------------------------------------------------------------
    scene_img = util.to_torch(synth.hdr_scene(W, H, seed=11).view(np.uint16))
    sp = images.plane(scene_img, _abi.FORMAT_R16G16B16A16_SFLOAT)
    aa = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    ap_ = images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)
    timeit("copy scene", lambda: ctx.copy_scene(sp, ap_), 16 * px)
    mips = [torch.zeros((mh, mw, 4), dtype=torch.int16, device="cuda") for (mw, mh) in images.bloom_mip_sizes(W, H, 6)]
    chain = images.mipchain(mips)
    timeit("bloom chain (6 mips)", lambda: ctx.bloom(sp, chain), int((8 + 2.667 + 2.667) * px))
    timeit("copy scene + bloom mip 0 (one pass)", lambda: ctx.copy_scene_bloom_mip0(sp, ap_, chain, (0, H), (0, mips[0].shape[0])), int((8 + 8 + 2) * px))
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    op = images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)
    timeit("tonemap composite", lambda: ctx.tonemap(sp, chain, op), int((8 + 4 + 2.667) * px))
    timeit("tonemap composite, tolerance 1 code", lambda: ctx.tonemap(sp, chain, op, flags=_abi.TONEMAP_TOLERANCE_1CODE), int((8 + 4 + 2.667) * px))

    # ---- LPV maintenance ----------------------------------------------------------------------------------
    vols = synth.lpv_volumes(4, 5)
    a_t = [util.to_torch(v.view(np.uint16)) for v in vols]
    b_t = [torch.zeros_like(t) for t in a_t]
    av = [images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in a_t]
    bv = [images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in b_t]
    timeit("lpv propagate x32", lambda: ctx.lpv_propagate(av, bv, 4, 32), 32 * 6 * 1024 * 1024)
    timeit("lpv clear", lambda: ctx.lpv_clear(av[0], av[1], av[2], bv[0], 4), 4 * 1024 * 1024)

    # ---- probe maintenance (a11) and sky LUTs (f3): tiny, launch bound -------------------------------------------------------
    if wanted("probe copy") or wanted("probe update x256"):
        atl, trace, ids = synth.probe_maintenance_inputs(seed=12, num_probes=256)
        a_t = {k: util.to_torch(v.view(np.uint16) if v.dtype == np.float16 else v) for k, v in atl.items()}
        b_t = {k: torch.zeros_like(v) for k, v in a_t.items()}
        a_d, b_d = util.probe_atlases_desc(a_t), util.probe_atlases_desc(b_t)
        tr_t = util.to_torch(trace.view(np.uint16))
        ids_t = torch.from_numpy(ids.view(np.int32)).cuda()
        tv = images.volume(tr_t, _abi.FORMAT_R16G16B16A16_SFLOAT)
        atlas_bytes = sum(int(v.numel() * v.element_size()) for v in a_t.values())
        timeit("probe copy (5 atlases)", lambda: ctx.probe_copy(a_d, b_d, [[1, 0, 0], [0, 1, 0], [0, 0, -1], [0, 0, 0]]), 2 * atlas_bytes)
        timeit("probe update x256", lambda: ctx.probe_update(a_d, tv, ids_t.data_ptr(), 256), 256 * 20 * 20 * 8)
    if wanted("sky LUT update"):
        luts = [torch.zeros(shape, dtype=torch.int16, device="cuda") for shape in ((64, 256, 4), (32, 32, 4), (200, 200, 4))]
        lp = [images.plane(a, _abi.FORMAT_R16G16B16A16_SFLOAT) for a in luts]
        timeit("sky LUT update (3 LUTs)", lambda: ctx.sky_update_luts(lp[0], lp[1], lp[2], (0.3, -0.8, 0.52)), (256 * 64 + 32 * 32 + 200 * 200) * 8)
    # ---- scene rasteriser (f1, f2): the atrium at two tessellations --------------------------------------------------------------
    if wanted("raster"):
        from androidrenderer_amd import mesh, scene
        view = scene.SceneView.default(W, H)
        sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_CSM)
        constants = sun.update_shadow_cascades(view, resolution=4096)
        gb = {"color": torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"), "normals": torch.zeros((H, W, 4), dtype=torch.int16, device="cuda"),
              "data": torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"), "emission": torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"),
              "depth": torch.zeros((H, W), dtype=torch.float32, device="cuda")}
        gbd = images.gbuffer(gb)
        sm = torch.zeros((4, 4096, 4096), dtype=torch.int16, device="cuda")
        smv = images.volume(sm, _abi.FORMAT_D16_UNORM)
        for subdiv in (1, 24):
            arrays = mesh.atrium(subdiv).arrays()
            dev = mesh.to_device(arrays)
            geo = mesh.geometry(dev, [])
            tris = int(arrays["indices"].shape[0] // 3)
            timeit(f"raster gbuffer {tris} tris", lambda: ctx.gbuffer_render(geo, view.gpu_data, gbd), 24 * px)
            timeit(f"raster shadow 4x4096^2 {tris} tris", lambda: ctx.shadow_render(geo, constants, 4, smv), 4 * 4096 * 4096 * 2)
    # ---- LPV injection chain (f4): RSM 4 x 128^2, VPL extraction and injection for the four cascades ----------------------------------
    if wanted("lpv inject"):
        from androidrenderer_amd import mesh, scene
        view = scene.SceneView.default(W, H)
        sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_CSM)
        casc = scene.LpvCascades()
        casc.update_cascade_transforms(view, sun)
        arrays = mesh.atrium(24).arrays()
        dev = mesh.to_device(arrays)
        geo = mesh.geometry(dev, [])
        rsm_t = {"flux": torch.zeros((4, 128, 128, 4), dtype=torch.uint8, device="cuda"), "normals": torch.zeros((4, 128, 128, 4), dtype=torch.uint8, device="cuda"),
                 "depth": torch.zeros((4, 128, 128), dtype=torch.int16, device="cuda")}
        rsm = _abi.RsmTargets(images.volume(rsm_t["flux"], _abi.FORMAT_R8G8B8A8_SRGB), images.volume(rsm_t["normals"], _abi.FORMAT_R8G8B8A8_UNORM),
                              images.volume(rsm_t["depth"], _abi.FORMAT_D16_UNORM))
        lists = torch.zeros((4, 4096, 4), dtype=torch.int32, device="cuda")
        counts = torch.zeros(4, dtype=torch.int32, device="cuda")
        vols_t = [torch.zeros((32, 32, 128, 4), dtype=torch.int16, device="cuda") for _ in range(3)]
        vd = [images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in vols_t]
        timeit("lpv inject: rsm render 4x128^2", lambda: ctx.rsm_render(geo, sun.constants, casc.matrices, 4, rsm), 4 * 128 * 128 * 10)

        def extract_and_inject():
            for c in range(4):
                ctx.lpv_extract_vpls(rsm, casc.matrices, c, 0.25, lists[c].data_ptr(), counts[c:].data_ptr())
                ctx.lpv_inject_vpls(lists[c].data_ptr(), counts[c:].data_ptr(), 4096, casc.matrices, c, 4, vd)
        timeit("lpv inject: extract + inject x4", extract_and_inject, 4 * 128 * 128 * 10 + 4 * 4096 * 16)
    if args.json:
        print(json.dumps(results))


if __name__ == "__main__":
    main()
