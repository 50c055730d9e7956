"""Golden images (tests/golden/, made by tools/gen_golden.py — an independent vectorised-numpy restatement of the shader
text) against the oracle on the CPU and against the HIP library on the GPU.  The reference holds no golden images of its own
(SURVEY.md §8-c), so this is a two-implementations-agree check, not a pin to reference output: parity stays "unpinned"."""
import ctypes as C
import os

import numpy as np
import pytest

from androidrenderer_amd import _abi, images, synth
from tests import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LIGHTING = ("lighting_csm_lpv", "lighting_rt", "lighting_csm", "lighting_rt_rtgi", "lighting_csm_lights",
            "lighting_rt_cache_sky",   # a4 irradiance-cache gather + a6 sky fill: the reference's default configuration
            "lighting_csm_lpv_sky")    # a6 on the fast path (its own sky kernel)


def _frame(name):
    g = np.load(os.path.join(GOLDEN, f"{name}_64x36.npz"))
    f = util.golden_lighting_frame(64, 36, int(g["seed"]), int(g["sun_mode"]), int(g["gi"]), sky=bool(int(g["sky"])) if "sky" in g.files else False)
    assert f.inputs_sha256() == str(g["inputs_sha256"]), "synthetic input generators drifted: re-run tools/gen_golden.py"
    if "lights" in g.files:  # point-light list of the a9 fixture (stored with the image)
        f.lights = np.ascontiguousarray(g["lights"], dtype=np.float32)
        f.arrays["lights"] = f.lights
    return f, g["lit"]


def _lpv_fixture():
    g = np.load(os.path.join(GOLDEN, "lpv_propagate_2c_3steps.npz"))
    return [g[f"in{i}"].copy() for i in range(3)], [g[f"out{i}"] for i in range(3)]


def _probe_update_fixture():
    """Inputs (regenerated from the seed) and the expected atlases: the inputs with the golden's changed texels applied."""
    g = np.load(os.path.join(GOLDEN, "probe_update_48.npz"))
    atl, trace, ids = synth.probe_maintenance_inputs(seed=int(g["seed"]), num_probes=int(g["num_probes"]))
    want = {}
    for k, v in atl.items():
        w = (v.view(np.uint16) if v.dtype == np.float16 else v).copy()
        idx = tuple(g[f"{k}_idx"].T.astype(np.int64))
        w[idx] = g[f"{k}_val"]
        assert len(idx[0]) > 0
        want[k] = w
    return atl, trace, ids, want


def _post_inputs():
    g = np.load(os.path.join(GOLDEN, "post_64x36.npz"))
    scene_img = synth.hdr_scene(64, 36, seed=int(g["seed"])).view(np.uint16)
    return scene_img, [g[f"mip{i}"] for i in range(6)], g["final"]


@pytest.mark.parametrize("name", LIGHTING)
def test_oracle_matches_golden_lighting(name):
    f, want = _frame(name)
    got = f.run_oracle()
    d = util.f16_ulp_diff(got, want)
    assert d.max() == 0, util.report_ulp(name, d)
    assert (want[..., :3] != 0).mean() > 0.5  # the image is not trivially black


def test_oracle_matches_golden_post():
    scene_img, mips_want, final_want = _post_inputs()
    o = util.oracle()
    mips = [np.zeros((h, w, 4), np.uint16) for (w, h) in images.bloom_mip_sizes(64, 36, 6)]
    sp = images.plane(scene_img, _abi.FORMAT_R16G16B16A16_SFLOAT)
    mc = images.mipchain(mips)
    assert o.orc_bloom(C.byref(sp), C.byref(mc)) == 0
    for i, (m, w) in enumerate(zip(mips, mips_want)):
        d = util.f16_ulp_diff(m[..., :3], w[..., :3])
        assert d.max() == 0, util.report_ulp(f"mip{i}", d)
    out = np.zeros((36, 64, 4), np.uint8)
    op = images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)
    assert o.orc_tonemap(C.byref(sp), C.byref(mc), C.byref(op), 0, 0) == 0
    assert np.array_equal(out, final_want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", LIGHTING)
@pytest.mark.parametrize("force_general", [0, 1])
def test_hip_matches_golden_lighting(hip_ctx, name, force_general):
    f, want = _frame(name)
    hip_ctx.debug_set(force_general=force_general, force_ppt=0)
    try:
        got = f.run_hip(hip_ctx)
    finally:
        hip_ctx.debug_set(force_general=0, force_ppt=0)
    d = util.f16_ulp_diff(got, want)
    assert d.max() == 0, util.report_ulp(name, d)


@pytest.mark.gpu
def test_hip_matches_golden_post(hip_ctx):
    import torch
    scene_img, mips_want, final_want = _post_inputs()
    dev_scene = util.to_torch(scene_img)
    mips = [torch.zeros((h, w, 4), dtype=torch.int16, device="cuda") for (w, h) in images.bloom_mip_sizes(64, 36, 6)]
    sp = images.plane(dev_scene, _abi.FORMAT_R16G16B16A16_SFLOAT)
    mc = images.mipchain(mips)
    hip_ctx.bloom(sp, mc)
    out = torch.zeros((36, 64, 4), dtype=torch.uint8, device="cuda")
    op = images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)
    hip_ctx.tonemap(sp, mc, op, 0, 0)
    torch.cuda.synchronize()
    for i, (m, w) in enumerate(zip(mips, mips_want)):
        d = util.f16_ulp_diff(util.from_torch(m, np.uint16)[..., :3], w[..., :3])
        assert d.max() == 0, util.report_ulp(f"mip{i}", d)
    assert np.array_equal(out.cpu().numpy(), final_want)


def test_oracle_matches_golden_copy_scene_and_lpv_propagate():
    o = util.oracle()
    src = np.load(os.path.join(GOLDEN, "lighting_csm_lpv_64x36.npz"))["lit"]
    want = np.load(os.path.join(GOLDEN, "copy_scene_64x36.npz"))["out"]
    out = np.zeros_like(src)
    sp, op = images.plane(src, _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(out, _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert o.orc_copy_scene(C.byref(sp), C.byref(op)) == 0
    assert np.array_equal(out, want)
    # LPV propagate: 3 steps A -> B -> A -> B, two cascades; the result of an odd number of steps is in B
    a, want_b = _lpv_fixture()
    b = [np.zeros_like(v) for v in a]
    av = (_abi.Volume * 3)(*[images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in a])
    bv = (_abi.Volume * 3)(*[images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in b])
    assert o.orc_lpv_propagate(av, bv, 2, 3) == 0
    for c in range(3):
        d = util.f16_ulp_diff(b[c], want_b[c])
        assert d.max() == 0, util.report_ulp(f"lpv channel {c}", d)
    assert any(int((w != 0).sum()) > 500 for w in want_b)  # the light did spread


@pytest.mark.gpu
def test_hip_matches_golden_copy_scene_and_lpv_propagate(hip_ctx):
    import torch
    src = np.load(os.path.join(GOLDEN, "lighting_csm_lpv_64x36.npz"))["lit"]
    want = np.load(os.path.join(GOLDEN, "copy_scene_64x36.npz"))["out"]
    s_t, o_t = util.to_torch(src), torch.zeros((36, 64, 4), dtype=torch.int16, device="cuda")
    hip_ctx.copy_scene(images.plane(s_t, _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(o_t, _abi.FORMAT_R16G16B16A16_SFLOAT))
    a, want_b = _lpv_fixture()
    a_t = [util.to_torch(v) for v in a]
    b_t = [torch.zeros_like(t) for t in a_t]
    hip_ctx.lpv_propagate([images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in a_t], [images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in b_t], 2, 3)
    torch.cuda.synchronize()
    assert np.array_equal(util.from_torch(o_t, np.uint16), want)
    for c in range(3):
        assert np.array_equal(util.from_torch(b_t[c], np.uint16), want_b[c]), f"lpv channel {c}"


def test_oracle_matches_golden_probe_update():
    atl, trace, ids, want = _probe_update_fixture()
    o = util.oracle()
    a = util.probe_atlases_desc(atl)
    tv = images.volume(trace.view(np.uint16), _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert o.orc_probe_update(C.byref(a), C.byref(tv), ids.ctypes.data, len(ids)) == 0
    for k, w in want.items():
        got = atl[k].view(np.uint16) if atl[k].dtype == np.float16 else atl[k]
        assert np.array_equal(got, w), f"atlas {k}: {int((got != w).sum())} words differ"


@pytest.mark.gpu
def test_hip_matches_golden_probe_update(hip_ctx):
    import torch
    atl, trace, ids, want = _probe_update_fixture()
    a_t = {k: util.to_torch(v.view(np.uint16) if v.dtype == np.float16 else v) for k, v in atl.items()}
    tr_t = util.to_torch(trace.view(np.uint16))
    ids_t = torch.from_numpy(ids.view(np.int32)).cuda()
    hip_ctx.probe_update(util.probe_atlases_desc(a_t), images.volume(tr_t, _abi.FORMAT_R16G16B16A16_SFLOAT), ids_t.data_ptr(), len(ids))
    torch.cuda.synchronize()
    for k, w in want.items():
        got = a_t[k].cpu().numpy()
        assert np.array_equal(got.view(w.dtype).reshape(w.shape), w), f"atlas {k} differs"


def _raster_golden():
    g = np.load(os.path.join(GOLDEN, "raster_gbuffer_64x36.npz"))
    return {k: g[k] for k in ("color", "normals", "data", "emission", "depth")}


def test_oracle_matches_golden_textured_gbuffer():
    """f1: the depth pre-pass + G-buffer pass with material textures against the independent numpy rasteriser of tools/gen_golden.py"""
    from tests.test_raster import _oracle_gbuffer
    m, view = util.golden_raster_scene()
    got, stats = _oracle_gbuffer(m.arrays(), view, 64, 36)
    want = _raster_golden()
    for k in want:
        assert np.array_equal(got[k], want[k]), f"{k}: {int((got[k] != want[k]).sum())} values differ"
    assert (want["depth"] > 0).sum() > 400 and stats[1] == 2  # the two triangles with the opposite winding are culled


@pytest.mark.gpu
def test_hip_matches_golden_textured_gbuffer(hip_ctx):
    from tests.test_raster import _hip_gbuffer
    m, view = util.golden_raster_scene()
    got, _ = _hip_gbuffer(hip_ctx, m.arrays(), view, 64, 36)
    want = _raster_golden()
    for k in want:
        assert np.array_equal(got[k], want[k]), f"{k}: {int((got[k] != want[k]).sum())} values differ"


def test_oracle_matches_golden_anisotropic_gbuffer():
    """the same scene with anisotropic samplers (8x / 2.5x / 16x / 4x): N taps along the major axis of the footprint, per sah_hip.h"""
    from tests.test_raster import _oracle_gbuffer
    m, view = util.golden_raster_scene(anisotropic=True)
    got, _ = _oracle_gbuffer(m.arrays(), view, 64, 36)
    want = np.load(os.path.join(GOLDEN, "raster_gbuffer_aniso_64x36.npz"))
    iso = _raster_golden()
    for k in iso:
        assert np.array_equal(got[k], want[k]), f"{k}: {int((got[k] != want[k]).sum())} values differ"
    assert (want["color"] != iso["color"]).any() and (want["data"] != iso["data"]).any()  # the oblique wall really takes the anisotropic path
    # (the depth plane moves too: the wall is alpha-tested with the base-colour texture's alpha)


@pytest.mark.gpu
def test_hip_matches_golden_anisotropic_gbuffer(hip_ctx):
    from tests.test_raster import _hip_gbuffer
    m, view = util.golden_raster_scene(anisotropic=True)
    got, _ = _hip_gbuffer(hip_ctx, m.arrays(), view, 64, 36)
    want = np.load(os.path.join(GOLDEN, "raster_gbuffer_aniso_64x36.npz"))
    for k in ("color", "normals", "data", "emission", "depth"):
        assert np.array_equal(got[k], want[k]), f"{k}: {int((got[k] != want[k]).sum())} values differ"


def test_oracle_matches_golden_shadow_cascades():
    """f2: two sun shadow cascades of the same scene (D16, LESS, texture-alpha cutouts) against the numpy rasteriser"""
    from tests.test_raster import _oracle_shadow
    m, view = util.golden_raster_scene()
    sun = util.golden_raster_sun(view)
    got, _ = _oracle_shadow(m.arrays(), sun.constants, 2, (48, 48))
    want = np.load(os.path.join(GOLDEN, "raster_shadow_2x48.npz"))["shadowmap"]
    assert np.array_equal(got, want), int((got != want).sum())
    assert (want != 0xFFFF).sum() > 100


@pytest.mark.gpu
def test_hip_matches_golden_shadow_cascades(hip_ctx):
    from tests.test_raster import _hip_shadow
    m, view = util.golden_raster_scene()
    sun = util.golden_raster_sun(view)
    got, _ = _hip_shadow(hip_ctx, m.arrays(), sun.constants, 2, (48, 48))
    want = np.load(os.path.join(GOLDEN, "raster_shadow_2x48.npz"))["shadowmap"]
    assert np.array_equal(got, want), int((got != want).sum())


def _rsm_of_golden_scene():
    m, view = util.golden_raster_scene()
    sun = util.golden_raster_sun(view)
    return m, sun, util.golden_raster_lpv(view, sun), np.load(os.path.join(GOLDEN, "raster_rsm_4x32.npz"))


def test_oracle_matches_golden_rsm():
    """f4 (first stage): the LPV's reflective shadow map of the same scene against the numpy rasteriser"""
    from tests.test_lpv_inject import _oracle_rsm
    m, sun, lpv, want = _rsm_of_golden_scene()
    got = _oracle_rsm(m.arrays(), sun, lpv, res=32)
    for k in ("depth", "flux", "normals"):
        assert np.array_equal(got[k], want[k]), f"{k}: {int((got[k] != want[k]).sum())} values differ"
    assert (want["depth"] != 0xFFFF).sum() > 100


@pytest.mark.gpu
def test_hip_matches_golden_rsm(hip_ctx):
    from tests.test_lpv_inject import _hip_rsm
    m, sun, lpv, want = _rsm_of_golden_scene()
    got = _hip_rsm(hip_ctx, m.arrays(), sun, lpv, res=32)
    for k in ("depth", "flux", "normals"):
        g = got[k].cpu().numpy()
        g = g.view(np.uint16) if k == "depth" else g
        assert np.array_equal(g, want[k]), f"{k}: {int((g != want[k]).sum())} values differ"


def _inject_golden():
    g = np.load(os.path.join(GOLDEN, "lpv_inject_4x32.npz"))
    want = [np.zeros((32, 32, 128, 4), np.uint16) for _ in range(3)]
    at = tuple(g["cells"].T.astype(np.int64))
    for i in range(3):
        want[i][at] = g[f"vol_{i}"]
    return [g[f"vpls_{c}"] for c in range(4)], want


def test_oracle_matches_golden_vpl_extraction_and_injection():
    """f4: VPL lists of the golden RSM (ascending invocation index) and the volumes after injecting them, cascade by cascade"""
    from tests.test_lpv_inject import _empty_volumes, _oracle_extract, _oracle_inject
    _, _, lpv, rsm = _rsm_of_golden_scene()
    rsm = {k: rsm[k] for k in ("flux", "normals", "depth")}
    lists, want = _inject_golden()
    vols = _empty_volumes()
    for c in range(4):
        vpls, count = _oracle_extract(rsm, lpv, c)
        assert count == len(lists[c]) and np.array_equal(vpls[:count], lists[c]), f"cascade {c}"
        _oracle_inject(vpls, count, lpv, c, vols)
    for i in range(3):
        assert np.array_equal(vols[i], want[i]), f"volume {i}: {int((vols[i] != want[i]).sum())} values differ"
    assert sum(len(l) for l in lists) > 40


@pytest.mark.gpu
def test_hip_matches_golden_vpl_extraction_and_injection(hip_ctx):
    import torch
    from tests.test_lpv_inject import CELL, _rsm_desc
    _, _, lpv, rsm = _rsm_of_golden_scene()
    rsm_t = {"flux": torch.from_numpy(rsm["flux"]).cuda(), "normals": torch.from_numpy(rsm["normals"]).cuda(),
             "depth": torch.from_numpy(rsm["depth"].view(np.int16)).cuda()}
    lists, want = _inject_golden()
    vols_t = [torch.zeros((32, 32, 128, 4), dtype=torch.int16, device="cuda") for _ in range(3)]
    vol_desc = [images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in vols_t]
    cap = 16 * 16
    for c in range(4):
        list_t = torch.zeros((cap, 4), dtype=torch.int32, device="cuda")
        count_t = torch.zeros(1, dtype=torch.int32, device="cuda")
        hip_ctx.lpv_extract_vpls(_rsm_desc(rsm_t), lpv.matrices, c, CELL, list_t.data_ptr(), count_t.data_ptr())
        torch.cuda.synchronize()
        n = int(count_t.item())
        assert n == len(lists[c]) and np.array_equal(list_t.cpu().numpy().view(np.uint32)[:n], lists[c]), f"cascade {c}"
        hip_ctx.lpv_inject_vpls(list_t.data_ptr(), count_t.data_ptr(), cap, lpv.matrices, c, 4, vol_desc)
    torch.cuda.synchronize()
    for i in range(3):
        assert np.array_equal(vols_t[i].cpu().numpy().view(np.uint16), want[i]), f"volume {i}"


def _check_sky_luts(t, m, s):
    g = np.load(os.path.join(GOLDEN, "sky_luts.npz"))
    assert np.array_equal(t[::4], g["transmittance_rows"]), "transmittance LUT"
    assert np.array_equal(m, g["multiscattering"]), "multiple-scattering LUT"
    assert np.array_equal(s[::4], g["sky_view_rows"]), "sky-view LUT"


def test_oracle_matches_golden_sky_luts():
    """f3: the three sky LUT generators against their numpy restatement (tools/gen_golden.py)"""
    from tests.test_sky_luts import _oracle_luts
    g = np.load(os.path.join(GOLDEN, "sky_luts.npz"))
    _check_sky_luts(*_oracle_luts(tuple(float(c) for c in g["light"])))


@pytest.mark.gpu
def test_hip_matches_golden_sky_luts(hip_ctx):
    import torch
    g = np.load(os.path.join(GOLDEN, "sky_luts.npz"))
    t, m, s = (torch.zeros(shape, dtype=torch.int16, device="cuda") for shape in ((64, 256, 4), (32, 32, 4), (200, 200, 4)))
    hip_ctx.sky_update_luts(*[images.plane(a, _abi.FORMAT_R16G16B16A16_SFLOAT) for a in (t, m, s)], tuple(float(c) for c in g["light"]))
    torch.cuda.synchronize()
    _check_sky_luts(*[a.cpu().numpy().view(np.uint16) for a in (t, m, s)])


def _probe_copy_fixture():
    import hashlib
    g = np.load(os.path.join(GOLDEN, "probe_copy.npz"))
    src, _, _ = synth.probe_maintenance_inputs(seed=int(g["seed"]), num_probes=4)
    dst = {k: np.full_like(v, 0x55 if v.dtype == np.uint8 else 0x3555 if v.dtype == np.uint32 else 7.0) for k, v in src.items()}
    want = {k: str(g[f"sha256_{k}"]) for k in src}
    digest = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    return src, dst, [[float(c) for c in row] for row in g["movement"]], want, digest


def test_oracle_matches_golden_probe_copy():
    """a11: the probe scroll (copy_cascades) against the sequential numpy restatement; the fixture holds the digest of every atlas"""
    from tests.test_probes import _oracle_copy
    src, dst, movement, want, digest = _probe_copy_fixture()
    _oracle_copy(src, dst, movement)
    for k in want:
        assert digest(dst[k]) == want[k], f"atlas {k}"


@pytest.mark.gpu
def test_hip_matches_golden_probe_copy(hip_ctx):
    import torch
    src, dst, movement, want, digest = _probe_copy_fixture()
    s_t = {k: util.to_torch(v.view(np.uint16) if v.dtype == np.float16 else v) for k, v in src.items()}
    d_t = {k: util.to_torch(v.view(np.uint16) if v.dtype == np.float16 else v) for k, v in dst.items()}
    hip_ctx.probe_copy(util.probe_atlases_desc(s_t), util.probe_atlases_desc(d_t), movement)
    torch.cuda.synchronize()
    for k in want:
        assert digest(d_t[k].cpu().numpy()) == want[k], f"atlas {k}"


# ---- ray tracing (f4, first slice): RTAO and the sun shadow mask ------------------------------------------------------------------
def _rt_fixture():
    from tests import test_rt
    g = np.load(os.path.join(GOLDEN, "rt_64x36.npz"))
    m, view, sun, noise = util.golden_rt_scene()
    case = test_rt.RtCase(m, 64, 36, view=view, gbuffer={"depth": g["depth"].copy(), "normals": g["normals"].copy()})
    case.noise = noise
    case.sun = sun
    return case, g


def test_oracle_matches_golden_rt():
    case, g = _rt_fixture()
    # the planes the rays start from are the golden rasteriser's; the oracle's rasteriser gives the same ones
    from androidrenderer_amd import mesh
    want = {"color": np.zeros((36, 64, 4), np.uint8), "normals": np.zeros((36, 64, 4), np.uint16), "data": np.zeros((36, 64, 4), np.uint8),
            "emission": np.zeros((36, 64, 4), np.uint8), "depth": np.zeros((36, 64), np.float32)}
    gb = images.gbuffer(want)
    assert util.oracle().orc_gbuffer_render(C.byref(case.host_geo), C.byref(case.view.gpu_data), C.byref(gb), None) == 0
    assert np.array_equal(want["depth"].view(np.uint32), g["depth"].view(np.uint32)) and np.array_equal(want["normals"], g["normals"])
    ao, mask = case.oracle_rtao(spp=2, radius=3.0), case.oracle_mask()
    assert np.array_equal(ao.view(np.uint32), g["ao"].view(np.uint32)), f"rtao: {int((ao != g['ao']).sum())} texels differ"
    assert np.array_equal(mask.view(np.uint32), g["mask"].view(np.uint32)), f"shadow mask: {int((mask != g['mask']).sum())} texels differ"
    assert (g["ao"] == 0).sum() > 50 and len(np.unique(g["mask"])) >= 3


@pytest.mark.gpu
def test_hip_matches_golden_rt(hip_ctx):
    case, g = _rt_fixture()
    case.hip_build(hip_ctx)
    ao, mask = case.hip_rtao(hip_ctx, spp=2, radius=3.0), case.hip_mask(hip_ctx)
    assert np.array_equal(ao.view(np.uint32), g["ao"].view(np.uint32))
    assert np.array_equal(mask.view(np.uint32), g["mask"].view(np.uint32))


def _check_rt_gi(rb, ri, trace):
    g = np.load(os.path.join(GOLDEN, "rt_gi_64x36.npz"))
    # texels the ray generator skips (sky pixels) keep the fill of the test's planes; the golden holds zeros there
    traced = (np.load(os.path.join(GOLDEN, "rt_64x36.npz"))["depth"] != 0)[..., None]
    for name, got, want in (("ray_buffer", rb, g["ray_buffer"]), ("ray_irradiance", ri, g["ray_irradiance"])):
        got = np.where(traced, got.view(np.uint16), 0)
        assert np.array_equal(got, want), f"{name}: {int((got != want).any(-1).sum())} texels differ"
        assert (got.view(np.uint16)[~traced[..., 0]] == 0).all()
    assert np.array_equal(trace.view(np.uint16), g["trace"]), f"probe trace: {int((trace.view(np.uint16) != g['trace']).any(-1).sum())} texels differ"
    d = g["trace"].view(np.float16)[..., 3].astype(np.float32)
    assert (d > 0).sum() > 2000 and (d < 0).sum() > 100 and (g["trace"][-1] == 0).all()  # hits, back faces, and the probe outside the cascades


def test_oracle_matches_golden_rt_gi_rays():
    """f4: one GI ray per pixel and twelve probes x 400 rays — closest hit, hit shading with its shadow ray, sky and next-cascade misses —
    against the independent numpy restatement of tools/gen_golden.py"""
    case, _ = _rt_fixture()
    rb, ri = case.oracle_rtgi()
    _check_rt_gi(rb, ri, case.oracle_probe_trace(util.golden_rt_gi_inputs()["probes"]))


def _check_rt_gi_bounces(run_rtgi, set_bounces):
    g = np.load(os.path.join(GOLDEN, "rt_gi_bounces_64x36.npz"))
    base = np.load(os.path.join(GOLDEN, "rt_gi_64x36.npz"))
    traced = (np.load(os.path.join(GOLDEN, "rt_64x36.npz"))["depth"] != 0)[..., None]
    try:
        for nb in (1, 2):
            set_bounces(nb)
            rb, ri = run_rtgi()
            assert np.array_equal(np.where(traced, rb.view(np.uint16), 0), base["ray_buffer"])  # directions, first-hit distances: as without
            got, want = np.where(traced, ri.view(np.uint16), 0), g[f"ray_irradiance_{nb}"]
            assert np.array_equal(got, want), f"{nb} bounce(s): {int((got != want).any(-1).sum())} texels differ"
            assert (want.view(np.float16).astype(np.float32) > base["ray_irradiance"].view(np.float16).astype(np.float32)).any()
    finally:
        set_bounces(0)


def test_oracle_matches_golden_rt_gi_bounces():
    """the bounce branch of the GI hit stage (gltf_basic_pbr.slang:481-517; brdf() = Fd + Fr there, Fd alone for the sun) against the numpy
    restatement, 1 and 2 bounces"""
    case, _ = _rt_fixture()
    _check_rt_gi_bounces(case.oracle_rtgi, lambda n: util.oracle().orc_rt_set_bounces(n))


@pytest.mark.gpu
def test_hip_matches_golden_rt_gi_bounces(hip_ctx):
    case, _ = _rt_fixture()
    case.hip_build(hip_ctx)
    _check_rt_gi_bounces(lambda: case.hip_rtgi(hip_ctx), hip_ctx.rt_set_bounces)


@pytest.mark.gpu
def test_hip_matches_golden_rt_gi_rays(hip_ctx):
    case, _ = _rt_fixture()
    case.hip_build(hip_ctx)
    rb, ri = case.hip_rtgi(hip_ctx)
    _check_rt_gi(rb, ri, case.hip_probe_trace(hip_ctx, util.golden_rt_gi_inputs()["probes"]))
