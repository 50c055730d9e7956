"""GPU: one frame through the C++ host façade (include/sah_host.hpp, tests/cpp/host_frame.cpp) — LPV propagate →
lighting → copy scene → bloom → tonemap — compared with the oracle running the same chain on the uniform blocks the
C++ program produced."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from androidrenderer_amd import _abi, images, scene, synth
from tests import util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_frame")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "host_frame.cpp")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "sah_host.hpp"))):
        return
    libdir = os.path.join(ROOT, "androidrenderer_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", EXE, "-L", libdir,
                           "-lsah_hip", f"-Wl,-rpath,{libdir}"])


@pytest.mark.parametrize("sun_mode", [_abi.SHADOW_MODE_RT])
def test_frame_through_cpp_facade(tmp_path, sun_mode):
    _build()
    W, H, steps = 192, 108, 4
    view = scene.SceneView.default(W, H)
    g = synth.atrium_gbuffer(W, H, view, seed=2)
    ao = synth.ao_plane(W, H, 3)
    mask = synth.shadow_mask(W, H, 4)
    luts = synth.sky_luts(7)
    vols = [v.view(np.uint16) for v in synth.lpv_volumes(4, 5)]
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(np.array([W, H, sun_mode, steps], dtype=np.uint32).tobytes())
        for a in (g["color"], g["normals"], g["data"], g["emission"], g["depth"], ao, mask, luts["transmittance"], luts["sky_view"], *vols):
            f.write(np.ascontiguousarray(a).tobytes())
    subprocess.check_call([EXE, str(inp), str(outp)])
    blob = open(outp, "rb").read()
    off = 0

    def take(n):
        nonlocal off
        b = blob[off:off + n]
        off += n
        return b

    view_c = _abi.ViewData.from_buffer_copy(take(432))
    sun_c = _abi.SunLightConstants.from_buffer_copy(take(640))
    casc = (_abi.LpvCascadeMatrices * 4).from_buffer_copy(take(1024))
    lpv_out = [np.frombuffer(take(128 * 32 * 32 * 8), dtype=np.uint16).reshape(32, 32, 128, 4) for _ in range(3)]
    lit = np.frombuffer(take(W * H * 8), dtype=np.uint16).reshape(H, W, 4)
    final = np.frombuffer(take(W * H * 4), dtype=np.uint8).reshape(H, W, 4)

    o = util.oracle()
    # 1. LPV propagation
    a_np = [v.copy() for v in vols]
    b_np = [np.zeros_like(v) for v in vols]
    a_v = (_abi.Volume * 3)(*[images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in a_np])
    b_v = (_abi.Volume * 3)(*[images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in b_np])
    assert o.orc_lpv_propagate(a_v, b_v, 4, steps) == 0
    for c in range(3):
        assert np.array_equal(lpv_out[c], a_np[c])
    # 2. lighting with the blocks the C++ façade built
    fr = util.LightingFrame(W, H, gbuffer=g, seed=0, sun_mode=sun_mode, gi=_abi.GI_LPV)
    fr.arrays["ao"], fr.arrays["shadow_mask"] = ao, mask
    fr.arrays["sky_t"], fr.arrays["sky_v"] = luts["transmittance"], luts["sky_view"]
    fr.arrays["lpv_r"], fr.arrays["lpv_g"], fr.arrays["lpv_b"] = a_np
    fr.view.gpu_data = view_c
    fr.sun.constants = sun_c
    fr.lpv.matrices = casc
    ref_lit = fr.run_oracle()
    d = util.f16_ulp_diff(lit, ref_lit)
    print(util.report_ulp("façade lit_scene", d))
    assert d.max() <= 1
    # 3. post chain
    aa = np.zeros_like(ref_lit)
    sp, ap = images.plane(ref_lit, _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert o.orc_copy_scene(C.byref(sp), C.byref(ap)) == 0
    mips = [np.zeros((mh, mw, 4), dtype=np.uint16) for (mw, mh) in images.bloom_mip_sizes(W, H, 6)]
    chain = images.mipchain(mips)
    assert o.orc_bloom(C.byref(ap), C.byref(chain)) == 0
    out = np.zeros((H, W, 4), dtype=np.uint8)
    op = images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)
    assert o.orc_tonemap(C.byref(ap), C.byref(chain), C.byref(op), 0, 0) == 0
    dc = np.abs(final.astype(np.int32) - out.astype(np.int32))
    print(f"façade final image: max code diff {dc.max()}, {int((dc > 0).sum())} differ")
    assert dc.max() <= 1
    # the camera the C++ SceneView builds is the reference's start-up camera, like scene.py's
    assert np.allclose(np.array(view_c.view[:]), np.array(view.gpu_data.view[:]), atol=1e-5)
    assert math.isclose(view_c.inverse_projection[0], view.gpu_data.inverse_projection[0], rel_tol=1e-5)


RASTER_EXE = os.path.join(ROOT, "tests", "cpp", "host_raster")


def _build_raster():
    src = os.path.join(ROOT, "tests", "cpp", "host_raster.cpp")
    if os.path.exists(RASTER_EXE) and os.path.getmtime(RASTER_EXE) > max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "sah_host.hpp"))):
        return
    libdir = os.path.join(ROOT, "androidrenderer_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", RASTER_EXE, "-L", libdir,
                           "-lsah_hip", f"-Wl,-rpath,{libdir}"])


def test_producer_passes_through_cpp_facade(tmp_path):
    """update_shadow_cascades + render_shadows + GbufferPhase::render in C++ against the oracle fed with the blocks C++ built, and
    the C++ cascade fitting against scene.py's."""
    from androidrenderer_amd import mesh
    _build_raster()
    W, H, R = 160, 90, 256
    m = mesh.atrium(2)
    # material textures through the facade's TextureDescriptorPool: base colour / normal / data on every material, emission on the lamps
    from androidrenderer_amd import synth
    g = synth.rng(78)
    tex = [m.add_texture(*mesh.random_texture(g, 64, 64, None, srgb, mesh.sampler())) for srgb in (True, False, False, True)]
    m.material_textures = [(tex[0], tex[1], tex[2], tex[3] if i == 5 else _abi.TEXTURE_NONE) for i in range(len(m.materials))]
    arrays = m.arrays()
    counts = mesh.with_counts(arrays)["counts"]
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(np.array([W, H, R, counts["vertices"], counts["indices"], counts["primitives"], counts["materials"], len(arrays["textures"])],
                         dtype=np.uint32).tobytes())
        for k in ("positions", "vertex_data", "indices", "primitives", "materials"):
            f.write(np.ascontiguousarray(arrays[k]).tobytes())
        for (mips, fmt, smp) in arrays["textures"]:
            f.write(np.array([fmt, len(mips)], np.uint32).tobytes())
            f.write(bytes(smp))
            for level in mips:
                f.write(np.array([level.shape[1], level.shape[0]], np.uint32).tobytes())
                f.write(np.ascontiguousarray(level).tobytes())
        noise = g.integers(0, 256, (128, 128, 4), dtype=np.uint8)
        f.write(noise.tobytes())  # (read before the material-texture table: see host_raster.cpp)
        f.write(np.ascontiguousarray(arrays["material_textures"], dtype=np.uint32).tobytes())
    subprocess.check_call([RASTER_EXE, str(inp), str(outp)])
    blob = open(outp, "rb").read()
    off = 0

    def take(n):
        nonlocal off
        b = blob[off:off + n]
        off += n
        return b

    view_c = _abi.ViewData.from_buffer_copy(take(432))
    sun_c = _abi.SunLightConstants.from_buffer_copy(take(640))
    sm = np.frombuffer(take(4 * R * R * 2), dtype=np.uint16).reshape(4, R, R)
    got = {"color": np.frombuffer(take(W * H * 4), np.uint8).reshape(H, W, 4), "normals": np.frombuffer(take(W * H * 8), np.uint16).reshape(H, W, 4),
           "data": np.frombuffer(take(W * H * 4), np.uint8).reshape(H, W, 4), "emission": np.frombuffer(take(W * H * 4), np.uint8).reshape(H, W, 4),
           "depth": np.frombuffer(take(W * H * 4), np.float32).reshape(H, W)}
    casc = (_abi.LpvCascadeMatrices * 4).from_buffer_copy(take(1024))
    lpv_got = [np.frombuffer(take(128 * 32 * 32 * 8), dtype=np.uint16).reshape(32, 32, 128, 4) for _ in range(3)]
    ao_got = np.frombuffer(take(W * H * 4), np.float32).reshape(H, W)
    mask_got = np.frombuffer(take(W * H * 4), np.float32).reshape(H, W)
    sky_t = np.frombuffer(take(256 * 64 * 8), np.uint16).reshape(64, 256, 4)
    sky_v = np.frombuffer(take(200 * 200 * 8), np.uint16).reshape(200, 200, 4)
    rb_got = np.frombuffer(take(W * H * 8), np.uint16).reshape(H, W, 4)
    ri_got = np.frombuffer(take(W * H * 8), np.uint16).reshape(H, W, 4)
    atl_got = {"rtgi": np.frombuffer(take(32 * 256 * 224 * 4), np.uint32).reshape(32, 256, 224),
               "light_cache": np.frombuffer(take(32 * 416 * 416 * 4), np.uint32).reshape(32, 416, 416),
               "depth": np.frombuffer(take(32 * 384 * 384 * 4), np.uint16).reshape(32, 384, 384, 2),
               "average": np.frombuffer(take(32 * 32 * 32 * 4), np.uint32).reshape(32, 32, 32),
               "validity": np.frombuffer(take(32 * 32 * 32), np.uint8).reshape(32, 32, 32)}
    assert off == len(blob)
    o = util.oracle()
    keep = []
    g = mesh.geometry(mesh.with_counts(arrays), keep)
    # LPV injection chain with the cascades C++ built: RSM -> VPLs -> volumes
    rsm = {"flux": np.zeros((4, 128, 128, 4), np.uint8), "normals": np.zeros((4, 128, 128, 4), np.uint8), "depth": np.zeros((4, 128, 128), np.uint16)}
    rd = _abi.RsmTargets(images.volume(rsm["flux"], _abi.FORMAT_R8G8B8A8_SRGB), images.volume(rsm["normals"], _abi.FORMAT_R8G8B8A8_UNORM),
                         images.volume(rsm["depth"], _abi.FORMAT_D16_UNORM))
    assert o.orc_rsm_render(C.byref(g), C.byref(sun_c), casc, 4, C.byref(rd), None) == 0
    lpv_want = [np.zeros((32, 32, 128, 4), np.uint16) for _ in range(3)]
    vols = (_abi.Volume * 3)(*[images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in lpv_want])
    injected = 0
    for c in range(4):
        vpls, count = np.zeros((4096, 4), np.uint32), np.zeros(1, np.uint32)
        assert o.orc_lpv_extract_vpls(C.byref(rd), casc, c, 0.25, vpls.ctypes.data, count.ctypes.data) == 0
        assert o.orc_lpv_inject_vpls(vpls.ctypes.data, count.ctypes.data, 4096, casc, c, 4, vols) == 0
        injected += int(count[0])
    assert injected > 200
    for c in range(3):
        assert np.array_equal(lpv_got[c], lpv_want[c]), f"injected LPV volume {c}"
    py_lpv = scene.LpvCascades()
    py_lpv.update_cascade_transforms(scene.SceneView.default(W, H), scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_CSM))
    for c in range(4):  # C++ and Python build the same RSM frusta (up to the rounding of their inverses)
        assert np.allclose(np.array(casc[c].rsm_vp[:]), np.array(py_lpv.matrices[c].rsm_vp[:]), rtol=2e-3, atol=2e-3)
        assert np.allclose(np.array(casc[c].world_to_cascade[:]), np.array(py_lpv.matrices[c].world_to_cascade[:]), rtol=1e-5, atol=1e-5)
    want_sm = np.zeros_like(sm)
    vol = images.volume(want_sm, _abi.FORMAT_D16_UNORM)
    assert o.orc_shadow_render(C.byref(g), C.byref(sun_c), 4, C.byref(vol), None) == 0
    assert np.array_equal(sm, want_sm)
    assert (sm != 0xffff).mean() > 0.2
    want = {"color": np.zeros((H, W, 4), np.uint8), "normals": np.zeros((H, W, 4), np.uint16), "data": np.zeros((H, W, 4), np.uint8),
            "emission": np.zeros((H, W, 4), np.uint8), "depth": np.zeros((H, W), np.float32)}
    gb = images.gbuffer(want)
    assert o.orc_gbuffer_render(C.byref(g), C.byref(view_c), C.byref(gb), None) == 0
    for k in want:
        assert np.array_equal(got[k].view(np.uint8), want[k].view(np.uint8)), k
    assert (want["depth"] > 0).mean() > 0.9
    # ray tracing through the facade (RaytracingScene::finalize, AmbientOcclusionPhase::generate_ao in RTAO mode, DirectionalLight::raytrace)
    # against the oracle's brute-force loop over every triangle, on the G-buffer checked above
    ao_want, mask_want = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
    sun_rt = _abi.SunLightConstants.from_buffer_copy(bytes(sun_c))
    sun_rt.num_shadow_samples = 2.0
    pd, pn = images.plane(want["depth"], _abi.FORMAT_D32_SFLOAT), images.plane(want["normals"], _abi.FORMAT_R16G16B16A16_SFLOAT)
    pz = images.plane(noise, _abi.FORMAT_R8G8B8A8_UNORM)
    pa, pm = images.plane(ao_want, _abi.FORMAT_R32_SFLOAT), images.plane(mask_want, _abi.FORMAT_R32_SFLOAT)
    assert o.orc_rtao(C.byref(g), C.byref(view_c), C.byref(pd), C.byref(pn), C.byref(pz), 1, 8.0, C.byref(pa)) == 0
    assert o.orc_sun_shadow_mask(C.byref(g), C.byref(view_c), C.byref(sun_rt), C.byref(pd), C.byref(pn), C.byref(pz), C.byref(pm)) == 0
    assert np.array_equal(ao_got.view(np.uint32), ao_want.view(np.uint32)), "RTAO through the facade"
    assert np.array_equal(mask_got.view(np.uint32), mask_want.view(np.uint32)), "sun shadow mask through the facade"
    assert (ao_want == 0).any() and (mask_want < 1).any()
    # the GI rays through the facade (RayTracedGlobalIllumination::pre_render -> IrradianceCache::dispatch_probe_updates, ::post_render)
    # against the oracle, fed with the sky LUTs C++ generated
    sky = _abi.SkyLuts(images.plane(sky_t, _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(sky_v, _abi.FORMAT_R16G16B16A16_SFLOAT))
    rb_want, ri_want = np.zeros((H, W, 4), np.uint16), np.zeros((H, W, 4), np.uint16)  # create_texture clears to zero
    assert o.orc_rtgi_trace(C.byref(g), C.byref(view_c), C.byref(sun_rt), C.byref(sky), C.byref(pd), C.byref(pn), C.byref(pz),
                            C.byref(images.plane(rb_want, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.plane(ri_want, _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
    assert np.array_equal(rb_got, rb_want), "RTGI ray buffer through the facade"
    assert np.array_equal(ri_got, ri_want), "RTGI ray irradiance through the facade"
    assert (rb_want.view(np.float16)[..., 3] > 0).mean() > 0.3
    atl_want = {k: np.zeros_like(v) for k, v in atl_got.items()}  # fresh atlases, no cascade movement: the copy pass moves zeros
    ids = np.array([(16, 3, 16), (15, 2, 17), (10, 9, 20), (18, 12, 14), (16, 19, 16), (16, 27, 15)], np.uint32)
    trace = np.zeros((len(ids), 20, 20, 4), np.uint16)
    d = _abi.ProbeTraceDesc()
    for c in range(4):
        spacing = 0.5 * (1 << c)
        d.cascades[c].probe_spacing = spacing
        d.cascades[c].min[0], d.cascades[c].min[1], d.cascades[c].min[2] = -16.0 * spacing + 0.013, 1.0 - 4.0 * spacing, -16.0 * spacing + 0.021
    d.probes_to_update, d.num_probes = ids.ctypes.data, len(ids)
    d.sun, d.sky, d.noise = C.pointer(sun_rt), C.pointer(sky), C.pointer(pz)
    d.probe_irradiance = images.volume(atl_want["rtgi"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32)
    d.probe_depth = images.volume(atl_want["depth"], _abi.FORMAT_R16G16_SFLOAT)
    d.probe_validity = images.volume(atl_want["validity"], _abi.FORMAT_R8_UNORM)
    d.probe_size[0], d.probe_size[1] = 5, 6
    d.trace_results = images.volume(trace, _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert o.orc_probe_trace(C.byref(g), C.byref(d)) == 0
    tv = images.volume(trace, _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert o.orc_probe_update(C.byref(util.probe_atlases_desc(atl_want)), C.byref(tv), ids.ctypes.data, len(ids)) == 0
    for k in atl_want:
        assert np.array_equal(atl_got[k], atl_want[k]), f"irradiance cache atlas {k} after tracing and folding six probes through the facade"
    assert atl_want["rtgi"].any() and atl_want["validity"].any()
    # cascade fitting: same algorithm as scene.py (glm restated twice); the inverses differ in rounding, so compare loosely
    view = scene.SceneView.default(W, H)
    sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_CSM)
    py = sun.update_shadow_cascades(view, resolution=R)
    for c in range(4):
        a, b = np.array(sun_c.cascade_matrices[c][:]), np.array(py.cascade_matrices[c][:])
        assert np.allclose(a, b, rtol=2e-3, atol=2e-3), (c, a, b)
        assert math.isclose(sun_c.data[c][0], py.data[c][0], rel_tol=1e-5)
    assert sun_c.csm_resolution[0] == R
