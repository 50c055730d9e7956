"""GPU parity: the fused HIP Lighting pass vs the CPU oracle on the same seeded inputs (bar: <= 1 ULP per channel
of the stored RGBA16F value; SURVEY.md §8-c, BASELINE.json north_star)."""
import numpy as np
import pytest

from androidrenderer_amd import _abi
from tests import util

pytestmark = pytest.mark.gpu

MAX_ULP = 1  # tolerance stated by BASELINE.json north_star: within 1 ULP per channel (fp16 storage)


def _check(frame, hip_ctx, name):
    ref = frame.run_oracle()
    got = frame.run_hip(hip_ctx)
    d = util.f16_ulp_diff(got, ref)
    print(util.report_ulp(name, d))
    assert d.max() <= MAX_ULP, util.report_ulp(name, d)
    return d


@pytest.mark.parametrize("sun_mode", [_abi.SHADOW_MODE_OFF, _abi.SHADOW_MODE_CSM, _abi.SHADOW_MODE_RT])
@pytest.mark.parametrize("gi", [_abi.GI_NONE, _abi.GI_LPV])
def test_lighting_random_gbuffer(hip_ctx, sun_mode, gi):
    f = util.LightingFrame(256, 144, seed=11 + sun_mode * 3 + gi, sun_mode=sun_mode, gi=gi, flavour="random")
    _check(f, hip_ctx, f"random sun={sun_mode} gi={gi}")


@pytest.mark.parametrize("sun_mode", [_abi.SHADOW_MODE_CSM, _abi.SHADOW_MODE_RT])
def test_lighting_atrium_lpv(hip_ctx, sun_mode):
    f = util.LightingFrame(320, 180, seed=5, sun_mode=sun_mode, gi=_abi.GI_LPV, flavour="atrium")
    _check(f, hip_ctx, f"atrium sun={sun_mode} lpv")


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
