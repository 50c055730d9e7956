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
