"""GPU: the library's own exchange step (sah_comm_unique_id -> sah_create -> sah_lighting on a row range -> sah_allgather_rows), i.e.
every RCCL call the N-rank path makes.  world = 1 runs on the one-GPU box through a one-rank communicator; world = 2 runs when two
GPUs are visible, each rank a fresh child process that starts before anything touches the GPU (tests/dist_child.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from androidrenderer_amd import _abi, images, lib
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sharded_gather(ctx, f, dev, shards, rows_per, world_slots):
    """Shade `shards` into one padded target and gather it in place; returns the (height, width, 4) uint16 frame."""
    import torch
    lit = torch.zeros((rows_per * world_slots, f.width, 4), dtype=torch.int16, device="cuda")
    for r0, r1 in shards:
        f.row_begin, f.row_end = r0, r1
        d, keep = f.describe(dev, lit[:f.height])
        ctx.lighting(d)
    ctx.allgather_rows(images.plane(lit[:f.height], _abi.FORMAT_R16G16B16A16_SFLOAT), rows_per, rows_per * world_slots)
    ctx.sync()
    torch.cuda.synchronize()
    return util.from_torch(lit[:f.height], np.uint16)


@pytest.mark.parametrize("height", [96, 97])
@pytest.mark.parametrize("side_stream", [False, True])
def test_one_rank_communicator_gathers_hip_shaded_rows(height, side_stream):
    import torch
    ctx = lib.Context(device=0, rank=0, world=1, comm_id=lib.comm_unique_id())
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    side = torch.cuda.Stream() if side_stream else None
    if side is not None:
        ctx.comm_set_stream(side.cuda_stream)
    try:
        f = util.LightingFrame(160, height, seed=21, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")
        dev = f.device_arrays()
        full = f.run_hip(ctx, dev)
        got = _sharded_gather(ctx, f, dev, [(0, 32), (32, 80), (80, height)], height, 1)
        ctx.comm_wait()
        torch.cuda.synchronize()
        assert np.array_equal(got, full), "rows shaded shard by shard and gathered through RCCL differ from the unsharded frame"
        assert np.array_equal(full, util.LightingFrame(160, height, seed=21, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium").run_oracle())
        # raw byte gather through the same communicator
        buf = torch.arange(4096, dtype=torch.int32, device="cuda")
        ctx.allgather_bytes(buf.data_ptr(), 4096 * 4)
        ctx.comm_wait()
        torch.cuda.synchronize()
        assert bool((buf == torch.arange(4096, dtype=torch.int32, device="cuda")).all())
    finally:
        torch.cuda.synchronize()
        ctx.close()


def test_allgather_rows_rejects_bad_slots(hip_ctx):
    import torch
    img = torch.zeros((37, 16, 4), dtype=torch.int16, device="cuda")
    p = images.plane(img, _abi.FORMAT_R16G16B16A16_SFLOAT)
    with pytest.raises(lib.SahError) as e:
        hip_ctx.allgather_rows(p, 36, 37)  # one rank, 36-row slot: row 36 would never be gathered
    assert e.value.status == _abi.SAH_ERR_INVALID_ARGUMENT
    with pytest.raises(lib.SahError) as e:
        hip_ctx.allgather_rows(p, 40, 37)  # slot larger than the allocation
    assert e.value.status == _abi.SAH_ERR_INVALID_ARGUMENT
    hip_ctx.allgather_rows(p, 37, 37)  # no communicator, one rank: a no-op


@pytest.mark.parametrize("height", [96, 97])
def test_two_rank_gather_equals_unsharded(tmp_path, height):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the round-end 8-GPU node; the one-GPU box runs the world = 1 tests above)")
    comm_id = lib.comm_unique_id()
    (tmp_path / "id.bin").write_bytes(comm_id)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_child.py"), str(r), "2", str(height), str(tmp_path)], env=env)
             for r in range(2)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0, 0], rcs
    ref = util.LightingFrame(160, height, seed=21, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium").run_oracle()
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"rank{r}.npy"), ref), f"rank {r}: gathered frame differs from the oracle's unsharded frame"
