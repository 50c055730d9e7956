"""Randomised HIP-vs-oracle sweep of the producer passes (G-buffer, shadow cascades, RSM, VPL extraction / injection): triangle soups
of every size from sub-pixel to screen-filling, cameras inside the geometry (near-plane and guard-band clipping), odd resolutions,
instanced and cutout draws.  Complements tests/test_raster.py and tests/test_lpv_inject.py; prints one line per case and a summary.

    python tools/stress_raster.py [--cases 40]
"""
import argparse
import ctypes as C
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from androidrenderer_amd import _abi, images, lib, mesh, scene, synth  # noqa: E402
from tests import util  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--textured", action="store_true", help="random textures, samplers and texcoords on every material slot (clipped triangles included)")
    args = ap.parse_args()
    import torch
    ctx = lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    o = util.oracle()
    bad = 0
    for case in range(args.cases):
        g = synth.rng(9000 + case)
        w, h = [(320, 180), (257, 131), (64, 64), (511, 77), (96, 200)][case % 5]
        extent = float(g.choice([1.5, 6.0, 40.0]))
        size = (0.01, float(g.choice([0.5, 5.0, 60.0])))
        arrays = mesh.random_soup(100 + case, triangles=int(g.choice([200, 800, 3000])), extent=extent, size=size, textured=args.textured).arrays()
        view = scene.SceneView()
        view.rotate(float(g.uniform(-1.2, 1.2)), float(g.uniform(0, 2 * math.pi)))
        view.set_position(g.uniform(-extent, extent, 3))
        view.set_render_resolution(w, h)
        view.set_perspective_projection(float(g.choice([40.0, 75.0, 110.0])), w / h, float(g.choice([0.05, 0.5])))
        view.update_transforms()
        if args.textured:
            view.gpu_data.material_texture_mip_bias = float(g.choice([0.0, -1.0, 0.75]))
        sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_CSM)
        sun.set_direction(g.normal(size=3))
        res = int(g.choice([64, 200, 512]))
        constants = sun.update_shadow_cascades(view, max_shadow_distance=float(g.choice([16.0, 128.0])), resolution=res)
        lpv = scene.LpvCascades()
        lpv.update_cascade_transforms(view, sun)
        host_geo = mesh.geometry(mesh.with_counts(arrays), [])
        dev = mesh.to_device(arrays)
        dev_geo = mesh.geometry(dev, [])
        fails = []
        # G-buffer
        want = {"color": np.zeros((h, w, 4), np.uint8), "normals": np.zeros((h, w, 4), np.uint16), "data": np.zeros((h, w, 4), np.uint8),
                "emission": np.zeros((h, w, 4), np.uint8), "depth": np.zeros((h, w), np.float32)}
        wd = images.gbuffer(want)
        ws = np.zeros(8, np.uint32)
        assert o.orc_gbuffer_render(C.byref(host_geo), C.byref(view.gpu_data), C.byref(wd), ws.ctypes.data) == 0
        got = {"color": torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda"), "normals": torch.zeros((h, w, 4), dtype=torch.int16, device="cuda"),
               "data": torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda"), "emission": torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda"),
               "depth": torch.zeros((h, w), dtype=torch.float32, device="cuda")}
        gs = torch.zeros(8, dtype=torch.int32, device="cuda")
        ctx.gbuffer_render(dev_geo, view.gpu_data, images.gbuffer(got), gs.data_ptr())
        torch.cuda.synchronize()
        for k in want:
            if not np.array_equal(got[k].cpu().numpy().view(np.uint8), want[k].view(np.uint8)):
                fails.append(f"gbuffer.{k}")
        if list(gs.cpu().numpy().view(np.uint32)[:4]) != list(ws[:4]):
            fails.append("gbuffer.stats")
        # shadow cascades
        want_sm = np.zeros((4, res, res), np.uint16)
        assert o.orc_shadow_render(C.byref(host_geo), C.byref(constants), 4, C.byref(images.volume(want_sm, _abi.FORMAT_D16_UNORM)), None) == 0
        got_sm = torch.zeros((4, res, res), dtype=torch.int16, device="cuda")
        ctx.shadow_render(dev_geo, constants, 4, images.volume(got_sm, _abi.FORMAT_D16_UNORM))
        torch.cuda.synchronize()
        if not np.array_equal(got_sm.cpu().numpy().view(np.uint16), want_sm):
            fails.append("shadow")
        # RSM -> VPLs -> LPV
        rsm_np = {"flux": np.zeros((4, 128, 128, 4), np.uint8), "normals": np.zeros((4, 128, 128, 4), np.uint8), "depth": np.zeros((4, 128, 128), np.uint16)}
        rsm_t = {k: torch.zeros(v.shape, dtype=torch.int16 if v.dtype == np.uint16 else torch.uint8, device="cuda") for k, v in rsm_np.items()}

        def desc(a):
            return _abi.RsmTargets(images.volume(a["flux"], _abi.FORMAT_R8G8B8A8_SRGB), images.volume(a["normals"], _abi.FORMAT_R8G8B8A8_UNORM),
                                   images.volume(a["depth"], _abi.FORMAT_D16_UNORM))
        assert o.orc_rsm_render(C.byref(host_geo), C.byref(sun.constants), lpv.matrices, 4, C.byref(desc(rsm_np)), None) == 0
        ctx.rsm_render(dev_geo, sun.constants, lpv.matrices, 4, desc(rsm_t))
        torch.cuda.synchronize()
        for k in rsm_np:
            if not np.array_equal(rsm_t[k].cpu().numpy().view(np.uint8), rsm_np[k].view(np.uint8)):
                fails.append(f"rsm.{k}")
        vols_np = [np.zeros((32, 32, 128, 4), np.uint16) for _ in range(3)]
        vols_t = [torch.zeros((32, 32, 128, 4), dtype=torch.int16, device="cuda") for _ in range(3)]
        vd_np = (_abi.Volume * 3)(*[images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in vols_np])
        vd_t = [images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in vols_t]
        lights = 0
        for c in range(4):
            vp, cnt = np.zeros((4096, 4), np.uint32), np.zeros(1, np.uint32)
            assert o.orc_lpv_extract_vpls(C.byref(desc(rsm_np)), lpv.matrices, c, 0.25, vp.ctypes.data, cnt.ctypes.data) == 0
            assert o.orc_lpv_inject_vpls(vp.ctypes.data, cnt.ctypes.data, 4096, lpv.matrices, c, 4, vd_np) == 0
            lt, ct = torch.zeros((4096, 4), dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
            ctx.lpv_extract_vpls(desc(rsm_t), lpv.matrices, c, 0.25, lt.data_ptr(), ct.data_ptr())
            ctx.lpv_inject_vpls(lt.data_ptr(), ct.data_ptr(), 4096, lpv.matrices, c, 4, vd_t)
            torch.cuda.synchronize()
            n = int(cnt[0])
            lights += n
            if int(ct.item()) != n or not np.array_equal(lt.cpu().numpy().view(np.uint32)[:n], vp[:n]):
                fails.append(f"vpl[{c}]")
        for c in range(3):
            if not np.array_equal(vols_t[c].cpu().numpy().view(np.uint16), vols_np[c]):
                fails.append(f"lpv[{c}]")
        covered = float((want["depth"] > 0).mean())
        print(f"case {case:3d} {w}x{h} res {res} tris {arrays['indices'].shape[0] // 3:5d} stats {[int(v) for v in ws[:4]]} covered {covered:.2f} lights {lights:5d} "
              f"{'OK' if not fails else 'MISMATCH ' + ','.join(fails)}", flush=True)
        bad += bool(fails)
    print(f"{args.cases - bad} of {args.cases} cases bit-identical")
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
