"""CPU-only, world_size 2 over gloo: the N>1 path of bench.py — contiguous row shards, each rank shades its rows, an
all-gather re-assembles the frame on every rank — must equal the unsharded frame bit for bit.
The per-rank compute stand-in is the oracle (there is no GPU here); the shard arithmetic and the gather are the code under
test."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard_rows(height, world, rank):
    """Row block of `rank` (androidrenderer_amd/shard.py, the rule bench.py uses): ceil(height / world) rows each, clipped to the
    image: equal-sized gather slots, the last ones possibly short or empty."""
    sys.path.insert(0, ROOT)
    from androidrenderer_amd import shard
    return shard.lighting_rows(height, world, rank)


def _worker(rank, world, port, height, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from androidrenderer_amd import _abi
    from tests import util

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = util.LightingFrame(64, height, seed=17, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")
    r0, r1 = shard_rows(height, world, rank)
    f.row_begin, f.row_end = r0, r1
    per = -(-height // world)
    shaded = f.run_oracle().view(np.uint8)  # bytes, as the gather moves them; only rows [r0, r1) are written
    padded = torch.zeros((per * world,) + shaded.shape[1:], dtype=torch.uint8)  # equal slots; the tail slot may be short
    padded[r0:r1] = torch.from_numpy(shaded[r0:r1])
    parts = [padded[i * per:(i + 1) * per] for i in range(world)]
    dist.all_gather(parts, parts[rank].clone())
    full = padded[:height].numpy().view(np.uint16)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height", [36, 37])  # even split and a remainder row
def test_row_shards_allgather_equals_unsharded(tmp_path, height):
    import torch.multiprocessing as mp
    from androidrenderer_amd import _abi
    from tests import util

    world = 2
    port = 29500 + (os.getpid() % 2000) + height
    mp.spawn(_worker, args=(world, port, height, str(tmp_path)), nprocs=world, join=True)
    ref = util.LightingFrame(64, height, seed=17, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium").run_oracle()
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert np.array_equal(got, ref), f"rank {r}: gathered frame differs from the unsharded frame"


def test_shard_rows_partition():
    for h in (1, 7, 270, 2160, 2161):
        for w in (1, 2, 4, 8):
            rows = [shard_rows(h, w, r) for r in range(w)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            assert all(0 <= b - a <= -(-h // w) for a, b in rows)


def _ramp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import types

    import torch
    import torch.distributed as dist

    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = types.SimpleNamespace(args=types.SimpleNamespace(ramp_ms=200.0), torch=torch, dist=dist, torch_pg=True, red_dev=torch.device("cpu"), world=world, rank=rank)
    # a clock of its own per rank: rank 1's ramp begins 30 ms "later" and its steps are a little faster, so that each rank, reading its own
    # clock alone, would stop after a different number of rounds (rank 0: 3 rounds of 8 x 10 ms, rank 1: 4 rounds of 8 x 7.5 ms)
    now = [0.0 if rank == 0 else 0.030]
    per_step = 0.010 if rank == 0 else 0.0075
    steps = []

    def step(k):
        now[0] += per_step
        t = torch.tensor([k], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the exchange inside a step: a collective every rank must join, or it never returns
        assert int(t.item()) == k, "the ranks are not at the same step"
        steps.append(k)

    n = bench.clock_ramp(R, step, lambda: None, lambda: None, clock=lambda: now[0])
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64))
    with open(os.path.join(out_dir, f"ramp{rank}.txt"), "w") as f:
        f.write(" ".join(str(int(c.item())) for c in counts) + f" {len(steps)}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_clock_ramp_stops_on_every_rank_after_the_same_step(tmp_path):
    """bench.py's clock ramp runs for a TIME; its steps hold collectives.  Two ranks whose clocks disagree about when the time is up must
    still make the same number of steps (round 5: a four-rank rehearsal left one rank's last gathers without partners)."""
    import torch.multiprocessing as mp

    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_ramp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    seen = [open(tmp_path / f"ramp{r}.txt").read().split() for r in range(world)]
    assert seen[0] == seen[1], seen
    n0, n1, made = (int(v) for v in seen[0])
    assert n0 == n1 == made and made % 8 == 0 and 0 < made <= 32, seen
