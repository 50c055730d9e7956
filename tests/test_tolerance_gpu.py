"""GPU, experiment builds only: the "within 1 ULP" mode of the Lighting pass (csrc/params.hpp: kExpLightingTolerance1Ulp) against the strict mode — and the
strict mode is the one the oracle pins bit for bit (tests/test_lighting_gpu.py).  Bar: every channel of every pixel within 1 ULP of
the stored fp16 value (BASELINE.json north_star); any pixel beyond that is a bug in a guard band, not noise.  The ULP histogram and
the share of pixels the guards send to the strict restatement are printed for profiles/."""
import numpy as np
import pytest

from androidrenderer_amd import _abi, synth
from tests import util

import os

# The mode is 2.4x slower than the strict kernel (profiles/r2_tolerance_mode_v1.txt) and therefore not in the ABI; the relaxed body is
# compiled only with SAH_EXTRA_HIPCC_FLAGS=-DSAH_EXP_TOLERANCE_1ULP (python -m androidrenderer_amd.build --force), and these tests run
# only when SAH_EXP_TOLERANCE_1ULP=1 says the library under test is such a build.
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("SAH_EXP_TOLERANCE_1ULP") != "1", reason="needs an experiment build (-DSAH_EXP_TOLERANCE_1ULP)")]


def _both(ctx, f, dev=None):
    dev = dev or f.device_arrays()
    strict = f.run_hip(ctx, dev)
    n_strict = ctx.deferred_pixels()
    f.flags |= _abi.EXP_LIGHTING_TOLERANCE_1ULP
    relaxed = f.run_hip(ctx, dev)
    n_relaxed = ctx.deferred_pixels()
    f.flags &= ~_abi.EXP_LIGHTING_TOLERANCE_1ULP
    return strict, relaxed, n_strict, n_relaxed


def _report(name, strict, relaxed, n_strict, n_relaxed):
    d = util.f16_ulp_diff(relaxed, strict)
    hist = np.bincount(np.minimum(d.reshape(-1), 4), minlength=5)
    px = strict.shape[0] * strict.shape[1]
    print(f"{name}: ulp histogram [0,1,2,3,>=4] = {hist.tolist()}, deferred strict {n_strict} ({100.0 * n_strict / px:.3f} %) "
          f"relaxed {n_relaxed} ({100.0 * n_relaxed / px:.3f} %)")
    return d


@pytest.mark.parametrize("flavour", ["random", "atrium"])
@pytest.mark.parametrize("sun_mode,gi,flags", [(_abi.SHADOW_MODE_CSM, _abi.GI_LPV, _abi.LIGHTING_DEFAULT_FLAGS), (_abi.SHADOW_MODE_CSM, _abi.GI_NONE, _abi.LIGHTING_DEFAULT_FLAGS),
                                                (_abi.SHADOW_MODE_OFF, _abi.GI_LPV, _abi.LIGHTING_DEFAULT_FLAGS), (_abi.SHADOW_MODE_CSM, _abi.GI_LPV, 0)])
def test_tolerance_mode_within_one_ulp(hip_ctx, flavour, sun_mode, gi, flags):
    worst = 0
    for seed in (101, 102, 103):
        f = util.LightingFrame(512, 288, seed=seed, sun_mode=sun_mode, gi=gi, flavour=flavour, flags=flags, shadowmap_res=512,
                               shadow="scene" if (flavour == "atrium" and seed == 103) else "noise")
        d = _report(f"{flavour} sun={sun_mode} gi={gi} flags={flags} seed={seed}", *_both(hip_ctx, f))
        worst = max(worst, int(d.max()))
    assert worst <= 1


def test_tolerance_mode_matches_the_oracle_within_one_ulp(hip_ctx):
    f = util.LightingFrame(320, 180, seed=7, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")
    ref = f.run_oracle()
    f.flags |= _abi.EXP_LIGHTING_TOLERANCE_1ULP
    got = f.run_hip(hip_ctx)
    d = util.f16_ulp_diff(got, ref)
    print(util.report_ulp("tolerance mode vs oracle", d))
    assert d.max() <= 1


def test_tolerance_mode_shiny_and_dark_corners(hip_ctx):
    """Low roughness near the highlight (D_GGX denominator), grazing view angles, near-black results and negative overlays."""
    g = synth.random_gbuffer(512, 256, seed=55)
    rng = np.random.default_rng(9)
    g["data"][..., 1] = rng.integers(1, 40, g["data"].shape[:2])        # roughness 1/255 .. 39/255
    g["data"][..., 2] = rng.integers(0, 256, g["data"].shape[:2])
    g["color"][::2, :, :3] = rng.integers(0, 6, (128, 512, 3))           # nearly black base colours
    f = util.LightingFrame(512, 256, gbuffer=g, seed=56, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV)
    # normals clustered around the half vector of a typical pixel would need the view vector; random normals with a narrow lobe do:
    # 1 - NoH^2 spans [0, 1] over the frame, and with roughness this low the guard on the denominator must catch the peak
    for k in ("lpv_r", "lpv_g", "lpv_b"):  # signed volumes: the overlay cancels against the sun term on some pixels
        v = f.arrays[k].view(np.float16)
        v[..., 1:] *= np.float16(-1.5)
    d = _report("shiny / dark / signed", *_both(hip_ctx, f))
    assert d.max() <= 1


@pytest.mark.parametrize("flavour", ["atrium", "random"])
def test_tolerance_mode_4k(hip_ctx, flavour):
    f = util.LightingFrame(3840, 2160, seed=2, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour=flavour, shadowmap_res=4096, synth_device="cuda")
    d = _report(f"4k_deferred_gi {flavour}", *_both(hip_ctx, f))
    assert d.max() <= 1
