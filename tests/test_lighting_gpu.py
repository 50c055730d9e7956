"""GPU parity: the fused HIP Lighting pass vs the CPU oracle on the same seeded inputs (bar: <= 1 ULP per channel
of the stored RGBA16F value; SURVEY.md §8-c, BASELINE.json north_star)."""
import numpy as np
import pytest

from androidrenderer_amd import _abi
from tests import util

pytestmark = pytest.mark.gpu

MAX_ULP = 1  # tolerance stated by BASELINE.json north_star: within 1 ULP per channel (fp16 storage)


def _check(frame, hip_ctx, name):
    """Both device paths (fast kernel + fix-up, and the general kernel) against the oracle, and against each other."""
    ref = frame.run_oracle()
    dev = frame.device_arrays()
    outs = {}
    # (the fast kernel has three bodies — 4, 2 and 1 pixels per thread — and picks one by the size of the launch: frames of this size
    # get 1, so the other two are asked for by name; widths they cannot take fall back by themselves)
    for path, general, ppt in (("fast", False, 0), ("fast, 4 px per thread", False, 4), ("fast, 2 px per thread", False, 2), ("general", True, 0)):
        hip_ctx.debug_set(force_general=general, force_ppt=ppt)
        got = frame.run_hip(hip_ctx, dev)
        outs[path] = got
        d = util.f16_ulp_diff(got, ref)
        print(util.report_ulp(f"{name} [{path}]", d))
        assert d.max() <= MAX_ULP, util.report_ulp(f"{name} [{path}]", d)
    hip_ctx.debug_set(force_general=False, force_ppt=0)
    for path in outs:
        assert np.array_equal(outs[path], outs["general"]), f"the {path} kernel and the general kernel disagree"


@pytest.mark.parametrize("sun_mode", [_abi.SHADOW_MODE_OFF, _abi.SHADOW_MODE_CSM, _abi.SHADOW_MODE_RT])
@pytest.mark.parametrize("gi", [_abi.GI_NONE, _abi.GI_LPV])
def test_lighting_random_gbuffer(hip_ctx, sun_mode, gi):
    f = util.LightingFrame(256, 144, seed=11 + sun_mode * 3 + gi, sun_mode=sun_mode, gi=gi, flavour="random")
    _check(f, hip_ctx, f"random sun={sun_mode} gi={gi}")


@pytest.mark.parametrize("sun_mode", [_abi.SHADOW_MODE_CSM, _abi.SHADOW_MODE_RT])
def test_lighting_atrium_lpv(hip_ctx, sun_mode):
    f = util.LightingFrame(320, 180, seed=5, sun_mode=sun_mode, gi=_abi.GI_LPV, flavour="atrium")
    _check(f, hip_ctx, f"atrium sun={sun_mode} lpv")


@pytest.mark.parametrize("gi", [_abi.GI_NONE, _abi.GI_LPV])
def test_lighting_atrium_with_its_own_shadow_map(hip_ctx, gi):
    """Shadow map ray-cast from the atrium's boxes (synth.atrium_shadowmap): large coherent lit / shadowed regions, so whole
    waves take the 'no lane is lit and unshadowed' exit that random depth noise never triggers."""
    f = util.LightingFrame(320, 180, seed=6, sun_mode=_abi.SHADOW_MODE_CSM, gi=gi, flavour="atrium", shadowmap_res=512, shadow="scene")
    _check(f, hip_ctx, f"atrium + scene shadow map gi={gi}")
    lit = f.run_oracle().view(np.float16).astype(np.float32)[..., :3].sum(-1)
    surf = (f.arrays["depth"] != 0) & (f.arrays["emission"][..., :3].sum(-1) == 0)
    if gi == _abi.GI_NONE:
        frac = float((lit[surf] > 0).mean())
        assert 0.05 < frac < 0.6, frac  # some of the atrium is in the sun, most of it is not


def test_lighting_ragged_width(hip_ctx):
    # width not a multiple of 4: scalar path
    f = util.LightingFrame(131, 37, seed=3, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV)
    _check(f, hip_ctx, "ragged 131x37")


def test_row_shard_equals_full(hip_ctx):
    """Sharded == unsharded bit for bit (SURVEY §8-e determinism requirement)."""
    f = util.LightingFrame(128, 64, seed=9, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_LPV)
    dev = f.device_arrays()
    full = f.run_hip(hip_ctx, dev)
    import torch
    lit = torch.zeros((64, 128, 4), dtype=torch.int16, device="cuda")
    for r0, r1 in ((0, 16), (16, 48), (48, 64)):
        f.row_begin, f.row_end = r0, r1
        d, keep = f.describe(dev, lit)
        hip_ctx.lighting(d)
    torch.cuda.synchronize()
    assert np.array_equal(util.from_torch(lit, np.uint16), full)


def _poison(g, rng):
    """Adversarial texels: zero / huge / inf / NaN normals, roughness 0, denormal / inf / NaN / negative depth."""
    h, w = g["depth"].shape
    n = g["normals"].view(np.uint16)
    d = g["depth"].view(np.uint32)
    for k, (nb, db) in enumerate([((0, 0, 0), None), ((0x7BFF, 0x7BFF, 0x7BFF), None), ((0x7C00, 0x3C00, 0), None),
                                  ((0x7E00, 0x3C00, 0x3C00), None), (None, 0x00000001), (None, 0x7F800000), (None, 0x7FC00000),
                                  (None, 0xBF000000), ((0x0001, 0, 0), None), (None, 0x00800000), ((0x8000, 0x8000, 0x3C00), 0x3F7FFFFF)]):
        ys, xs = rng.integers(0, h, 40), rng.integers(0, w, 40)
        if nb is not None:
            for c in range(3):
                n[ys, xs, c] = nb[c]
        if db is not None:
            d[ys, xs] = db
    ys, xs = rng.integers(0, h, 300), rng.integers(0, w, 300)
    g["data"][ys, xs, 1] = 0  # roughness 0: the NaN-producing corner of D_GGX
    ys, xs = rng.integers(0, h, 100), rng.integers(0, w, 100)
    g["data"][ys, xs, 1] = 0
    g["normals"][ys, xs, :3] = g["normals"][ys, xs, :3]  # keep


@pytest.mark.parametrize("sun_mode", [_abi.SHADOW_MODE_CSM, _abi.SHADOW_MODE_RT])
@pytest.mark.parametrize("gi", [_abi.GI_NONE, _abi.GI_LPV])
def test_lighting_adversarial_inputs(hip_ctx, sun_mode, gi):
    from androidrenderer_amd import synth
    g = synth.random_gbuffer(192, 96, seed=77)
    _poison(g, np.random.default_rng(5))
    f = util.LightingFrame(192, 96, gbuffer=g, seed=78, sun_mode=sun_mode, gi=gi)
    f.arrays["ao"][3, 5] = np.nan
    f.arrays["ao"][7, 9] = np.inf
    # normals exactly along / against the sun (ndotl == 1: sqrt(1 - ndotl^2) == 0 in the PCF bias; ndotl == 0) and axis aligned
    sd = np.array(f.sun.constants.direction_and_tan_size[:3], dtype=np.float32)
    L = (-sd / np.linalg.norm(sd)).astype(np.float16)
    rng = np.random.default_rng(6)
    for vec in (L, -L, np.array([0, 1, 0], np.float16), np.array([0, 0, -1], np.float16), (L.astype(np.float32) * 1e-4).astype(np.float16)):
        ys, xs = rng.integers(0, 96, 60), rng.integers(0, 192, 60)
        f.arrays["normals"][ys, xs, :3] = vec
    _check(f, hip_ctx, f"adversarial sun={sun_mode} gi={gi}")


def test_lighting_nonfinite_lpv_volume(hip_ctx):
    """An inf / NaN texel in an LPV volume disables the 'specular quirk is inert' shortcut for the whole frame."""
    f = util.LightingFrame(160, 90, seed=31, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")
    v = f.arrays["lpv_g"].view(np.uint16)
    v[5, 7, 9, 1] = 0x7C00
    v[20, 11, 40, 2] = 0x7E00
    _check(f, hip_ctx, "lpv with inf/nan texels")
    f2 = util.LightingFrame(160, 90, seed=31, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")
    _check(f2, hip_ctx, "lpv finite again (state flag re-armed)")


def test_lighting_no_quirk_flag(hip_ctx):
    f = util.LightingFrame(128, 72, seed=13, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flags=0)
    _check(f, hip_ctx, "additive sun blend (quirk off)")


def test_lighting_no_sky(hip_ctx):
    f = util.LightingFrame(128, 72, seed=14, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_NONE, sky=False)
    _check(f, hip_ctx, "no sky")


def test_ao_off_mode_clear_equals_no_ao_plane(hip_ctx):
    """a12: AO mode Off clears the AO target to 1.0 (ambient_occlusion_phase.cpp:167-179); the LPV overlay must then give exactly
    what it gives with a constant-one AO plane, on a pitched plane whose padding stays untouched."""
    import torch
    f = util.LightingFrame(160, 90, seed=91, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")
    f.arrays["ao"] = np.ones((90, 160), dtype=np.float32)
    want = f.run_oracle()
    dev = f.device_arrays()
    padded = torch.full((90, 192), 7.0, dtype=torch.float32, device="cuda")  # row pitch 768 bytes, 160 texels used
    ao_view = padded[:, :160]
    hip_ctx.ao_clear(_abi.Plane(padded.data_ptr(), 160, 90, 192 * 4, _abi.FORMAT_R32_SFLOAT))
    torch.cuda.synchronize()
    assert bool((ao_view == 1.0).all()) and bool((padded[:, 160:] == 7.0).all())
    dev["ao"] = ao_view.contiguous()
    got = f.run_hip(hip_ctx, dev)
    assert np.array_equal(got, want)


def test_lpv_generation_keeps_and_drops_the_gather_copy(hip_ctx):
    """sah_gi::lpv_generation: the fast kernel's interleaved copy of the LPV is rebuilt when the counter or a volume descriptor changes, when a
    library call writes the volumes, and on every call at generation 0 — and a kept copy keeps its non-finite verdict."""
    import torch
    from androidrenderer_amd import images
    f = util.LightingFrame(256, 144, seed=61, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium", shadowmap_res=256)
    dev = f.device_arrays()
    f.lpv_generation = 7
    first = f.run_hip(hip_ctx, dev)
    assert np.array_equal(first, f.run_oracle())
    assert np.array_equal(f.run_hip(hip_ctx, dev), first)  # second call: copy kept
    # new contents + new counter -> the new image
    g = np.random.default_rng(5)
    for k in ("lpv_r", "lpv_g", "lpv_b"):
        f.arrays[k] = (f.arrays[k].view(np.float16) * np.float16(0.5) + g.uniform(0, 0.05, f.arrays[k].shape).astype(np.float16)).view(f.arrays[k].dtype)
        dev[k].copy_(util.to_torch(f.arrays[k]))
    f.lpv_generation = 8
    second = f.run_hip(hip_ctx, dev)
    want = f.run_oracle()
    assert np.array_equal(second, want) and not np.array_equal(second, first)
    # a library call that writes the volumes drops the copy although the counter stays
    vols = [images.volume(dev[k], _abi.FORMAT_R16G16B16A16_SFLOAT) for k in ("lpv_r", "lpv_g", "lpv_b")]
    hip_ctx.lpv_clear(vols[0], vols[1], vols[2], None, 4)
    torch.cuda.synchronize()
    for k in ("lpv_r", "lpv_g", "lpv_b"):
        f.arrays[k] = np.zeros_like(f.arrays[k])
    assert np.array_equal(f.run_hip(hip_ctx, dev), f.run_oracle())
    # a kept copy with an inf texel: every pixel goes through the general restatement on both calls
    f.arrays["lpv_g"] = f.arrays["lpv_g"].copy()
    f.arrays["lpv_g"].view(np.uint16)[3, 4, 5, 1] = 0x7c00
    dev["lpv_g"].copy_(util.to_torch(f.arrays["lpv_g"]))
    f.lpv_generation = 9
    want = f.run_oracle()
    assert np.array_equal(f.run_hip(hip_ctx, dev), want) and np.array_equal(f.run_hip(hip_ctx, dev), want)
    # ... and the next finite copy is clean again (the tag belongs to the pack that found the texel)
    f.arrays["lpv_g"].view(np.uint16)[3, 4, 5, 1] = 0
    dev["lpv_g"].copy_(util.to_torch(f.arrays["lpv_g"]))
    f.lpv_generation = 10
    assert np.array_equal(f.run_hip(hip_ctx, dev), f.run_oracle())
    assert hip_ctx.deferred_pixels() < 256 * 144 // 4


def test_probe_generation_keeps_and_drops_the_irradiance_copy(hip_ctx):
    """sah_gi::probe_generation: the tiled kernel's fp32 copy of the irradiance atlas is rebuilt when the counter or the atlas descriptor changes,
    when sah_probe_update writes the atlas through this context, and on every call at generation 0; inf / NaN texels widen to themselves."""
    import torch
    from androidrenderer_amd import images, synth
    f = util.LightingFrame(192, 108, seed=62, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_CACHE, flavour="atrium")
    dev = f.device_arrays()
    f.probe_generation = 3
    first = f.run_hip(hip_ctx, dev)
    assert np.array_equal(first, f.run_oracle())
    assert np.array_equal(f.run_hip(hip_ctx, dev), first)  # copy kept
    # new atlas contents under the same counter are NOT picked up (the caller's contract) ... until the counter moves
    f.arrays["probe_irr"] = synth.probe_atlases(999)["irradiance"]
    f.arrays["probe_irr"][3, 40:60, 30:50] = 0x7ff << 11 | 0x7c0  # G = NaN (exponent all ones, mantissa != 0), R = inf
    dev["probe_irr"].copy_(util.to_torch(f.arrays["probe_irr"]))
    assert np.array_equal(f.run_hip(hip_ctx, dev), first)
    f.probe_generation = 4
    second = f.run_hip(hip_ctx, dev)
    assert np.array_equal(second, f.run_oracle()) and not np.array_equal(second, first)
    # a probe update through the library drops the copy although the counter stays
    atl, trace, ids = synth.probe_maintenance_inputs(seed=5, num_probes=16)
    o_atl = {"rtgi": f.arrays["probe_irr"].copy(), "light_cache": atl["light_cache"], "depth": f.arrays["probe_depth"].copy(), "average": atl["average"],
             "validity": f.arrays["probe_val"].copy()}
    import ctypes as C
    tv = images.volume(trace.view(np.uint16), _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert util.oracle().orc_probe_update(C.byref(util.probe_atlases_desc(o_atl)), C.byref(tv), ids.ctypes.data, len(ids)) == 0
    h_atl = {"rtgi": dev["probe_irr"], "light_cache": util.to_torch(atl["light_cache"]), "depth": dev["probe_depth"], "average": util.to_torch(atl["average"]),
             "validity": dev["probe_val"]}
    ids_t, tr_t = util.to_torch(ids.reshape(-1)), util.to_torch(trace.view(np.uint16))
    hip_ctx.probe_update(util.probe_atlases_desc(h_atl), images.volume(tr_t, _abi.FORMAT_R16G16B16A16_SFLOAT), ids_t.data_ptr(), len(ids))
    f.arrays["probe_irr"], f.arrays["probe_depth"], f.arrays["probe_val"] = o_atl["rtgi"], o_atl["depth"], o_atl["validity"]
    third = f.run_hip(hip_ctx, dev)
    assert np.array_equal(third, f.run_oracle()) and not np.array_equal(third, second)
    # generation 0: rebuilt on every call
    f.probe_generation = 0
    f.arrays["probe_irr"] = synth.probe_atlases(1000)["irradiance"]
    dev["probe_irr"].copy_(util.to_torch(f.arrays["probe_irr"]))
    assert np.array_equal(f.run_hip(hip_ctx, dev), f.run_oracle())


def test_fixup_hint_across_calls(hip_ctx):
    """The fix-up launch returns at once unless a wave of the fast kernel raised the context's hint word; the two words are used in turn by
    consecutive calls and cleared by the call before.  A frame that lists pixels (random G-buffer: roughness 0 under the LPV) and one that lists
    none (atrium), in every order of succession, each against the general kernel's image of the same frame."""
    a = util.LightingFrame(256, 144, seed=31, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="random")
    b = util.LightingFrame(256, 144, seed=32, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")
    dev = {id(f): f.device_arrays() for f in (a, b)}
    want = {}
    hip_ctx.debug_set(force_general=True)
    for f in (a, b):
        want[id(f)] = f.run_hip(hip_ctx, dev[id(f)])
    listed = {}
    for ppt in (0, 4):
        hip_ctx.debug_set(force_general=False, force_ppt=ppt)
        for k, f in enumerate((a, b, b, a, a, b, a, b, b, b, a)):
            got = f.run_hip(hip_ctx, dev[id(f)])
            listed[id(f)] = hip_ctx.deferred_pixels()
            assert np.array_equal(got, want[id(f)]), f"call {k} (pixels per thread {ppt or 'auto'}): the fast path's image differs from the general kernel's"
    hip_ctx.debug_set(force_general=False, force_ppt=0)
    assert listed[id(a)] > 0 and listed[id(b)] == 0, listed  # (the two frames are what the test takes them for)
