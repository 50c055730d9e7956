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


@pytest.mark.parametrize("height", [97])
def test_two_ranks_on_one_gpu_direct_exchange(tmp_path, height):
    """A REAL two-rank run on the one-GPU box: two fresh child processes share GPU 0 (HIP IPC allows what RCCL refuses) and exchange bloom
    mip 0 and the final image by storing their rows straight into the peer's buffers (sah_ipc_*).  Every rank's gathered image must equal
    the unsharded chain — the HIP one for all frames, and the oracle's for the first."""
    _run_chain_children(tmp_path, height, [])
    _check_chain_children(tmp_path, height)


@pytest.mark.parametrize("no_split", [False, True])
def test_two_rank_chain_through_rccl(tmp_path, no_split):
    """The same frames, one rank per GPU, through the library's RCCL communicator: bloom mip 0 on the parent communicator, the final image
    on the reversed one — made by ncclCommSplit, or (SAH_COMM_NO_SPLIT=1) exchanged with grouped ncclSend / ncclRecv — step by step and
    with two frames in flight on two work streams and a side stream.  This is the loop bench.py runs at N > 1."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the round-end 8-GPU node; on the one-GPU box the direct-exchange test above runs the same frames)")
    _run_chain_children(tmp_path, 97, ["rccl"], {"SAH_COMM_NO_SPLIT": "1"} if no_split else {})
    _check_chain_children(tmp_path, 97)


def test_two_rank_chain_direct_exchange_across_devices(tmp_path):
    """The direct (IPC) exchange with one rank per GPU — what the one-GPU test cannot show: that a peer DEVICE's stores into this rank's
    buffers are visible to its next kernels after the counter handshake (same-device peers share an L2).  The first box with two GPUs
    decides it; RCCL stays bench.py's default until then."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the round-end 8-GPU node)")
    _run_chain_children(tmp_path, 97, ["ipc2"])
    _check_chain_children(tmp_path, 97)


@pytest.mark.parametrize("pipelined", [False, True])
def test_direct_exchange_gives_up_and_recovers(tmp_path, pipelined):
    """The give-up path of the direct exchange: one rank skips a gather for three seconds.  The other's sah_sync must report
    SAH_ERR_COMM after about two, stay failed, and have copied nothing into the late peer; the late peer must not copy into the rank that
    gave up; and sync -> barrier -> sah_ipc_reset -> barrier must bring both back (tests/ipc_giveup_child.py, two processes on GPU 0).
    `pipelined`: the rank that is alone has three gathers enqueued before it looks (the reset must not depend on counters nobody signalled)."""
    import json
    port = 29800 + (os.getpid() % 150)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port += 11 if pipelined else 0
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_giveup_child.py"), str(r), str(tmp_path), str(port)] + (["pipelined"] if pipelined else []),
                              env=env) for r in range(2)]
    rcs = [p.wait(timeout=120) for p in procs]
    assert rcs == [0, 0], rcs
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in range(2))
    for r in (r0, r1):
        assert r["gather1_ok"] and r["gather3_ok"] and r["gather4_ok"], r
        assert r["gather2_status"] == _abi.SAH_ERR_COMM and r["gather_after_giveup_status"] == _abi.SAH_ERR_COMM, r
        assert r["peer_slot_untouched"], r
        assert r["gather_after_giveup_seconds"] < 0.5, r
    assert 1.5 < r0["gather2_seconds"] < 4.0, r0  # the two-second bound of the waiting kernel


def _run_chain_children(tmp_path, height, extra, extra_env=None):
    port = 29300 + (os.getpid() % 500)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_child.py"), str(r), "2", str(height), str(tmp_path), str(port)] + extra,
                              env=env) for r in range(2)]
    rcs = [p.wait(timeout=300) for p in procs]
    assert rcs == [0, 0], rcs


def _check_chain_children(tmp_path, height):
    import ctypes as C
    import torch
    from androidrenderer_amd import chain
    # the unsharded chain in this process
    ctx = lib.Context(device=0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        f = util.LightingFrame(160, height, seed=37, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_CACHE, flavour="atrium")
        dev = f.device_arrays()
        plain = chain.ShardedChain(ctx, f, dev, 0, 1)
        plain.step(gather=False)
        torch.cuda.synchronize()
        want_step = plain.out.cpu().numpy().copy()
        g = torch.Generator(device="cpu").manual_seed(11)
        masks = [torch.rand((height, 160), generator=g).cuda() for _ in range(5)]
        want = {}
        for i, m in enumerate(masks):
            dev["shadow_mask"].copy_(m)
            plain.step(gather=False)
            torch.cuda.synchronize()
            want[i] = plain.out.cpu().numpy().copy()
    finally:
        torch.cuda.synchronize()
        ctx.close()
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"rank{r}_step.npy"), want_step), f"rank {r}: stepwise chain"
        for i in (1, 2, 3, 4):
            assert np.array_equal(np.load(tmp_path / f"rank{r}_frame{i}.npy"), want[i]), f"rank {r}: pipelined frame {i}"
    assert not np.array_equal(want[1], want[2])
    # the first image against the oracle's unsharded chain
    o = util.oracle()
    lit = f.run_oracle()
    aa = np.zeros_like(lit)
    assert o.orc_copy_scene(C.byref(images.plane(lit, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
    mips = [np.zeros((mh, mw, 4), np.uint16) for (mw, mh) in images.bloom_mip_sizes(160, height, 6)]
    assert o.orc_bloom(C.byref(images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.mipchain(mips))) == 0
    out = np.zeros((height, 160, 4), np.uint8)
    assert o.orc_tonemap(C.byref(images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.mipchain(mips)),
                         C.byref(images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)), 0, 0) == 0
    assert np.array_equal(want_step, out)


@pytest.mark.gpu
@pytest.mark.parametrize("fail_preflight", [False, True])
def test_bench_rehearsal_of_two_ranks_verifies_its_frames(tmp_path, fail_preflight):
    """bench.py's whole N > 1 control flow (process group, row plan, two frames in flight, both exchanges, max over ranks) with two ranks on
    the one GPU of the box (gloo + the direct exchange): the line it prints must say that the sharded loop's last frames equal the
    unsharded frame on every rank.  With the pre-flight check made to fail (test hook) every rank must fall back to the one-frame-at-a-time
    loop, and that loop's frames must be right too."""
    import json
    port = 29800 + (os.getpid() % 150)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **({"SAH_BENCH_FAIL_PREFLIGHT": "1"} if fail_preflight else {}))
    port += 7 if fail_preflight else 0
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--rehearse-on-one-gpu"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["sharded_equals_unsharded"] is True and d["value"] is not None and "error" not in d
    assert d["config"]["speedup_vs_same_workload_on_one_gpu"] is None  # a rehearsal shares one GPU: no speed-up is claimed
    pre = d["config"]["preflight"]
    if fail_preflight:
        assert d["config"]["frames_in_flight"] == 1 and pre["fallback"] == "one frame at a time" and pre["fallback_ok"] is True
    else:
        assert d["config"]["frames_in_flight"] == 2 and pre == {"two_frames_in_flight_ok": True, "fallback": None}


@pytest.mark.gpu
def test_bench_reports_a_failed_verification_as_a_failure(tmp_path):
    """A sharded loop whose last frames differ from the unsharded frame (test hook) has no throughput: value null, an `error`, exit code 3
    on every rank."""
    import json
    port = 29950 + (os.getpid() % 40)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SAH_BENCH_FAIL_VERIFY="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--rehearse-on-one-gpu", "--ramp-ms", "0"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode != 0, out.stdout[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] is None and "differ" in d["error"] and d["config"]["sharded_equals_unsharded"] is False
    assert d["config"]["speedup_vs_same_workload_on_one_gpu"] is None


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with no torch.distributed.run around it (the shape of the driver's one-GPU command): the file starts its ranks as
    a child process and relays their one line and exit code.  The line names the chain's metric and carries the one-GPU reference at its top level."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--rehearse-on-one-gpu", "--ramp-ms", "0"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is not None and d["config"]["sharded_equals_unsharded"] is True
    assert "final-image" in d["metric"] and "4k_probe_gi_chain" in d["metric"]
    assert d["same_workload_on_one_gpu"]["value"] > 0 and "speedup_vs_same_workload_on_one_gpu" in d
