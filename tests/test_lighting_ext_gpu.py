"""GPU parity for the tiled Lighting kernel: irradiance-cache overlay (a4), RTGI reconstruction (a5) and the point-light
extension (a9) with LDS tile culling. The culled result must be IDENTICAL to brute-force shading (SURVEY §8-c fixture iv)."""
import numpy as np
import pytest

from androidrenderer_amd import _abi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _vs_oracle(frame, hip_ctx, name, max_ulp=1):
    ref = frame.run_oracle()
    got = frame.run_hip(hip_ctx)
    d = util.f16_ulp_diff(got, ref)
    print(util.report_ulp(name, d))
    assert d.max() <= max_ulp, util.report_ulp(name, d)
    return got


@pytest.mark.parametrize("sun_mode", [_abi.SHADOW_MODE_RT, _abi.SHADOW_MODE_CSM])
@pytest.mark.parametrize("flavour", ["random", "atrium"])
def test_cache_overlay(hip_ctx, sun_mode, flavour):
    f = util.LightingFrame(160, 96, seed=41, sun_mode=sun_mode, gi=_abi.GI_CACHE, flavour=flavour)
    _vs_oracle(f, hip_ctx, f"cache gi sun={sun_mode} {flavour}")


def test_cache_overlay_debug_colours(hip_ctx):
    f = util.LightingFrame(96, 48, seed=42, sun_mode=_abi.SHADOW_MODE_OFF, gi=_abi.GI_CACHE, cache_debug_mode=1)
    _vs_oracle(f, hip_ctx, "cache gi debug mode")


@pytest.mark.parametrize("extra", [0, 3])
def test_rtgi_overlay(hip_ctx, extra):
    f = util.LightingFrame(144, 80, seed=43, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_RTGI, num_extra_rays=extra, flavour="atrium")
    _vs_oracle(f, hip_ctx, f"rtgi extra_rays={extra}")


@pytest.mark.parametrize("count,radius", [(64, 6.0), (256, 4.0), (1300, 3.0)])
def test_point_lights_culled_equals_brute_force_equals_oracle(hip_ctx, count, radius):
    base = util.LightingFrame(128, 80, seed=44, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_NONE, flavour="atrium")
    lights = synth.point_lights(base.view, count, radius, seed=45)
    f = util.LightingFrame(128, 80, seed=44, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_NONE, flavour="atrium", lights=lights)
    culled = _vs_oracle(f, hip_ctx, f"{count} point lights (tile-culled)")
    f.flags |= _abi.LIGHTING_BRUTE_FORCE_LIGHTS
    brute = f.run_hip(hip_ctx)
    assert np.array_equal(culled, brute), "tile culling changed the image"
    # the lights did something
    plain = base.run_hip(hip_ctx)
    assert not np.array_equal(plain, culled)


def test_point_lights_with_lpv_and_ragged_tiles(hip_ctx):
    base = util.LightingFrame(75, 45, seed=46, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_LPV)
    lights = synth.point_lights(base.view, 48, 8.0, seed=47)
    f = util.LightingFrame(75, 45, seed=46, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_LPV, lights=lights)
    _vs_oracle(f, hip_ctx, "48 lights + LPV, 75x45 (partial tiles)")


def test_point_lights_adversarial_light_data(hip_ctx):
    """Lights that leave the hot form's preconditions (tiled kernel: per-light uniform check, per-pixel fallback): a light sitting
    exactly on a shaded surface point (d2 == 0), zero / denormal / huge / infinite / NaN radius, infinite and NaN colour or
    intensity, NaN and infinite positions, and pixels in the razor-thin shell where the attenuation window underflows."""
    base = util.LightingFrame(96, 64, seed=48, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_NONE, flavour="atrium")
    lights = synth.point_lights(base.view, 24, 5.0, seed=49)
    # world position of a few pixels, recomputed as the shader does, to drop lights exactly onto surfaces
    o = util.oracle()
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    lights[1, 3] = 0.0            # radius 0: xr = d / 0
    lights[2, 3] = 1e-42          # denormal radius
    lights[3, 3] = 3e38           # huge radius: every pixel is "near"
    lights[4, 3] = inf
    lights[5, 3] = nan
    lights[6, 4] = inf            # colour
    lights[7, 7] = inf            # intensity
    lights[8, 7] = nan
    lights[9, 0] = nan            # position
    lights[10, 1] = inf
    lights[11, 7] = -3.0          # negative intensity: terms of either sign, and -0 products
    lights[12, 3] = 2.0 ** -30    # tiny but normal radius
    f = util.LightingFrame(96, 64, seed=48, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_NONE, flavour="atrium", lights=lights)
    culled = _vs_oracle(f, hip_ctx, "adversarial light list", max_ulp=0)
    f.flags |= _abi.LIGHTING_BRUTE_FORCE_LIGHTS
    assert np.array_equal(culled, f.run_hip(hip_ctx))


def test_point_lights_on_the_surface(hip_ctx):
    """d2 == 0 for one pixel (light placed at that pixel's reconstructed world position) and d ~ r for many (window near 0)."""
    base = util.LightingFrame(64, 48, seed=50, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_NONE, flavour="atrium")
    # reconstruct world positions the way directional_light.frag does, in float64 (close is enough for the second part; for the
    # exact hit the light position is refined below with the float32 value the kernel will compute)
    v = base.view.gpu_data
    depth = base.arrays["depth"]
    H, W = depth.shape
    ip = np.array(v.inverse_projection[:], dtype=np.float32).reshape(4, 4).T  # row-major [row][col]
    iv = np.array(v.inverse_view[:], dtype=np.float32).reshape(4, 4).T
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    f32 = np.float32
    tx = ((xs + f32(0.5)) + f32(0.5)) / f32(v.render_resolution[0])
    ty = ((ys + f32(0.5)) + f32(0.5)) / f32(v.render_resolution[1])
    ndc = np.stack([tx * f32(2) - f32(1), ty * f32(2) - f32(1), depth, np.ones_like(depth)], axis=-1)
    with np.errstate(all="ignore"):
        vs = np.stack([((ip[i, 0] * ndc[..., 0] + ip[i, 1] * ndc[..., 1]) + ip[i, 2] * ndc[..., 2]) + ip[i, 3] * ndc[..., 3] for i in range(4)], axis=-1)
        vs3 = np.stack([vs[..., i] / vs[..., 3] for i in range(3)] + [np.ones_like(depth)], axis=-1)
        ws = np.stack([((iv[i, 0] * vs3[..., 0] + iv[i, 1] * vs3[..., 1]) + iv[i, 2] * vs3[..., 2]) + iv[i, 3] * vs3[..., 3] for i in range(3)], axis=-1)
    surf = np.argwhere(depth != 0)
    lights = np.zeros((6, 8), dtype=np.float32)
    for k in range(6):
        y, x = surf[(k * 977) % len(surf)]
        lights[k, 0:3] = ws[y, x]
        lights[k, 3] = 0.75 + 0.5 * k
        lights[k, 4:7] = (1.0, 0.9, 0.8)
        lights[k, 7] = 4000.0
    f = util.LightingFrame(64, 48, seed=50, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_NONE, flavour="atrium", lights=lights)
    _vs_oracle(f, hip_ctx, "lights placed on surface points", max_ulp=0)


@pytest.mark.parametrize("gi,extra", [(_abi.GI_CACHE, 0), (_abi.GI_RTGI, 2)])
@pytest.mark.parametrize("sun_mode", [_abi.SHADOW_MODE_RT, _abi.SHADOW_MODE_CSM])
def test_tiled_kernel_adversarial_surface_inputs(hip_ctx, sun_mode, gi, extra):
    """Zero, NaN, infinite and denormal normals / depths and roughness 0 through the TILED kernel's Slang passes (the half root is
    v_sqrt_f32 alone, whose NaN may carry another sign or payload than the IEEE expansion the oracle uses — DESIGN.md §3 leaves NaN bit
    patterns unspecified; what must hold is that a NaN is a NaN in the same channels of the same pixels, and every finite value is the
    oracle's: util.f16_ulp_diff counts NaN against NaN as equal and NaN against a number as 65535)."""
    from tests.test_lighting_gpu import _poison
    g = synth.random_gbuffer(160, 96, seed=91)
    _poison(g, np.random.default_rng(7))
    f = util.LightingFrame(160, 96, gbuffer=g, seed=92, sun_mode=sun_mode, gi=gi, num_extra_rays=extra)
    if "shadow_mask" in f.arrays:
        f.arrays["shadow_mask"][4, 6] = np.nan
        f.arrays["shadow_mask"][9, 11] = np.inf
    got = _vs_oracle(f, hip_ctx, f"tiled adversarial sun={sun_mode} gi={gi}", max_ulp=0)
    nan = (got & 0x7FFF) > 0x7C00
    print(f"NaN channels in the image: {int(nan.sum())}")


@pytest.mark.parametrize("volumes", ["finite", "nonfinite"])
@pytest.mark.parametrize("sun_mode", [_abi.SHADOW_MODE_CSM, _abi.SHADOW_MODE_RT])
def test_point_lights_over_lpv_fast_overlay_adversarial(hip_ctx, sun_mode, volumes):
    """Round 6: with a light list the LPV overlay of the tiled kernel is the fast kernel's (gather from the packed copy) wherever its proofs hold.
    Poisoned surface inputs (zero / NaN / infinite / denormal normals and depths, roughness 0), NaN and infinite AO texels and — `nonfinite` — an
    inf and a NaN in the volumes (every pixel must then take the general overlay) against the oracle, against the context forced onto the
    general kernels, and with the gather copy kept across calls (lpv_generation != 0: no rebuild on the second call)."""
    from tests.test_lighting_gpu import _poison
    g = synth.random_gbuffer(160, 96, seed=93)
    _poison(g, np.random.default_rng(9))
    base = util.LightingFrame(160, 96, gbuffer=g, seed=94, sun_mode=sun_mode, gi=_abi.GI_LPV)
    lights = synth.point_lights(base.view, 40, 8.0, seed=95)
    f = util.LightingFrame(160, 96, gbuffer=g, seed=94, sun_mode=sun_mode, gi=_abi.GI_LPV, lights=lights)
    f.arrays["ao"][3, 5] = np.nan
    f.arrays["ao"][8, 13] = np.inf
    f.arrays["ao"][20, 40] = -1.0
    if volumes == "nonfinite":
        f.arrays["lpv_g"] = f.arrays["lpv_g"].copy()
        f.arrays["lpv_g"][5, 6, 7, 1] = np.float16(np.inf)
        f.arrays["lpv_b"] = f.arrays["lpv_b"].copy()
        f.arrays["lpv_b"][9, 3, 50, 0] = np.float16(np.nan)
    got = _vs_oracle(f, hip_ctx, f"lights + LPV (tiled, fast overlay) sun={sun_mode} {volumes}", max_ulp=0)
    hip_ctx.debug_set(force_general=True)
    try:
        general = f.run_hip(hip_ctx)
    finally:
        hip_ctx.debug_set(force_general=False)
    assert int(util.f16_ulp_diff(got, general).max()) == 0
    before = hip_ctx.copy_rebuilds()[0]
    f.lpv_generation = 7
    dev = f.device_arrays()
    first, second = f.run_hip(hip_ctx, dev), f.run_hip(hip_ctx, dev)
    assert hip_ctx.copy_rebuilds()[0] == before + 1, "the kept gather copy was rebuilt (or never built) by the tiled kernel's Lighting calls"
    assert np.array_equal(first, got) and np.array_equal(second, got)
