"""The sharded full chain (androidrenderer_amd/shard.py, chain.py): rows per rank, two exchanges (bloom mip 0 in rank order, final
RGBA8 image in reversed rank order), result identical to the unsharded frame.
CPU: plan properties; world-size-2 gloo run with the oracle as the per-rank compute stand-in, every buffer poisoned outside the
rows the plan says a rank computes.  GPU: the HIP passes with several ranks emulated on one device (slot copies stand in for the
gathers), and the real exchange through a one-rank RCCL communicator."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from androidrenderer_amd import _abi, images, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POISON16 = 0x7E01  # an fp16 NaN: any read of a row nobody computed shows in the result


def test_plans_cover_the_frame_and_their_dependencies():
    for H in (1, 2, 7, 36, 37, 270, 1081, 2160, 4320):
        for N in (1, 2, 3, 4, 8):
            plans = [shard.chain_plan(H, N, r) for r in range(N)]
            outs = sorted(p.out_rows for p in plans)
            assert sum(b - a for a, b in outs) == H and all(x[1] == y[0] or y[1] == y[0] for x, y in zip(outs, outs[1:]))
            h0 = max(1, H // 2)
            h1 = max(1, h0 // 2)
            assert sum(p.mip1_rows[1] - p.mip1_rows[0] for p in plans) == h1  # mip 1 is partitioned (and gathered) ...
            assert all(p.mip0_height == h0 and p.mip1_height == h1 for p in plans)
            for p in plans:
                assert p.out_rows == shard._clip(p.out_slot * p.rows_per_rank, (p.out_slot + 1) * p.rows_per_rank, H)
                for y in range(*p.out_rows):  # composite: scene row H - 1 - y and its neighbours (clamped at the edges) ...
                    assert p.aa_rows[0] <= max(H - 2 - y, 0) and min(H - y, H - 1) < p.aa_rows[1]
                    # ... and the mip 0 rows its tent taps can touch: v = 1 - (y + 0.5) / H, one texel either side, each a bilinear pair
                    pos = (1.0 - (y + 0.5) / H) * h0 - 0.5
                    lo, hi = int(np.floor(pos - 1.0 - 1e-3)), int(np.floor(pos + 1.0 + 1e-3)) + 1
                    assert p.mip0_rows[0] <= min(max(lo, 0), h0 - 1) and min(max(hi, 0), h0 - 1) < p.mip0_rows[1]
                def sources(j, hs, hd):  # rows within two texels of the tap centre, with the bilinear partner
                    c = (j + 0.5) * hs / hd - 0.5
                    return min(max(int(np.floor(c - 2.0 - 1e-3)), 0), hs - 1), min(max(int(np.floor(c + 2.0 + 1e-3)) + 1, 0), hs - 1)
                for j in range(*p.mip1_rows):  # ... mip 0 is not: every rank computes the rows its mip 1 rows read
                    lo, hi = sources(j, h0, h1)
                    assert p.mip0_rows[0] <= lo and hi < p.mip0_rows[1]
                for j in range(*p.mip0_rows):
                    lo, hi = sources(j, H, h0)
                    assert p.aa_rows[0] <= lo and hi < p.aa_rows[1]
                have = set(range(*p.lit_rows)) | set(range(*p.lit_wrap_rows))
                for j in range(*p.aa_rows):  # the copy's sampler repeats: row -1 is row H - 1, row H is row 0
                    assert {(j - 1) % H, j, (j + 1) % H} <= have


def _oracle_chain_rank(f, plan, gather_mip, gather_final):
    """One rank with the oracle as compute: whole-plane oracle passes, then everything outside the rows the plan assigns to this
    rank is poisoned before the next stage may read it."""
    from tests import util
    o = util.oracle()
    H, W = f.height, f.width
    lit = np.full((H, W, 4), POISON16, np.uint16)
    for rows in (plan.lit_rows, plan.lit_wrap_rows):
        if rows[1] > rows[0]:
            f.row_begin, f.row_end = rows
            lit[rows[0]:rows[1]] = f.run_oracle()[rows[0]:rows[1]]
    f.row_begin = f.row_end = 0
    aa = np.zeros_like(lit)
    assert o.orc_copy_scene(C.byref(images.plane(lit, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
    aa[:plan.aa_rows[0]] = POISON16
    aa[plan.aa_rows[1]:] = POISON16
    sizes = images.bloom_mip_sizes(W, H, 6)
    mip0 = np.zeros((sizes[0][1], sizes[0][0], 4), np.uint16)
    assert o.orc_bloom_downsample(C.byref(images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.plane(mip0, _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
    mip0[:plan.mip0_rows[0]] = POISON16  # mip 0 is not exchanged: the rank has the rows it computed for itself and nothing else
    mip0[plan.mip0_rows[1]:] = POISON16
    q = plan.mip1_rows_per_rank
    mip1_alloc = np.full((q * plan.world, sizes[1][0], 4), POISON16, np.uint16)
    mip1 = np.zeros((sizes[1][1], sizes[1][0], 4), np.uint16)
    assert o.orc_bloom_downsample(C.byref(images.plane(mip0, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.plane(mip1, _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
    mip1_alloc[plan.mip1_rows[0]:plan.mip1_rows[1]] = mip1[plan.mip1_rows[0]:plan.mip1_rows[1]]
    gather_mip(mip1_alloc, q)
    mips = [mip0, np.ascontiguousarray(mip1_alloc[:sizes[1][1]])] + [np.zeros((mh, mw, 4), np.uint16) for (mw, mh) in sizes[2:]]
    for m in range(2, 6):
        assert o.orc_bloom_downsample(C.byref(images.plane(mips[m - 1], _abi.FORMAT_R16G16B16A16_SFLOAT)),
                                      C.byref(images.plane(mips[m], _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
    out_alloc = np.full((plan.rows_per_rank * plan.world, W, 4), 0x5A, np.uint8)
    out = np.zeros((H, W, 4), np.uint8)
    if plan.out_rows[1] > plan.out_rows[0]:
        assert o.orc_tonemap(C.byref(images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.mipchain(mips)),
                             C.byref(images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)), plan.out_rows[0], plan.out_rows[1]) == 0
        out_alloc[plan.out_rows[0]:plan.out_rows[1]] = out[plan.out_rows[0]:plan.out_rows[1]]
    gather_final(out_alloc, plan.rows_per_rank)
    return out_alloc[:H]


def _frame(height):
    from tests import util
    return util.LightingFrame(96, height, seed=19, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")


def _unsharded_oracle(height):
    plan = shard.chain_plan(height, 1, 0)
    return _oracle_chain_rank(_frame(height), plan, lambda a, q: None, lambda a, q: None)


def _worker(rank, world, port, height, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = shard.chain_plan(height, world, rank)

    def gather(slot_of_rank):
        def run(alloc, rows):
            t = torch.from_numpy(alloc.view(np.uint8).reshape(alloc.shape[0], -1))
            parts = [t[i * rows:(i + 1) * rows] for i in range(world)]
            mine = slot_of_rank(rank)
            got = [torch.empty_like(parts[0]) for _ in range(world)]
            dist.all_gather(got, parts[mine].clone())
            for r in range(world):  # rank r's contribution is slot slot_of_rank(r)
                parts[slot_of_rank(r)].copy_(got[r])
        return run
    final = _oracle_chain_rank(_frame(height), plan, gather(lambda r: r), gather(lambda r: world - 1 - r))
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), final)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height", [54, 37])
def test_sharded_chain_over_gloo_equals_unsharded(tmp_path, height):
    import torch.multiprocessing as mp
    world = 2
    port = 29700 + (os.getpid() % 1500) + height
    mp.spawn(_worker, args=(world, port, height, str(tmp_path)), nprocs=world, join=True)
    ref = _unsharded_oracle(height)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"rank{r}.npy"), ref), f"rank {r}: final image differs from the unsharded chain"


@pytest.mark.parametrize("height,world", [(150, 3), (37, 3), (149, 4), (75, 8)])
def test_oracle_sharded_chain_with_emulated_ranks(height, world):
    """every rank in turn with the oracle as compute and poisoned rows outside its plan (no processes: the mip-1 slots of a first pass are
    handed to the second) — odd heights, where a mip is not exactly half of its source and the downsample's rows drift"""
    f, ref = _frame(height), _unsharded_oracle(height)
    slots = {}

    def run(rank, collect):
        plan = shard.chain_plan(height, world, rank)

        def gather_mip(alloc, q):
            if collect:
                slots[rank] = alloc[rank * q:(rank + 1) * q].copy()
            else:
                for r, v in slots.items():
                    alloc[r * q:(r + 1) * q] = v
        return plan, _oracle_chain_rank(f, plan, gather_mip, lambda alloc, per: None)
    for r in range(world):
        run(r, True)
    for r in range(world):
        plan, out = run(r, False)
        rows = slice(*plan.out_rows)
        assert np.array_equal(out[rows], ref[rows]), f"rank {r} of {world}: its rows of the final image differ from the unsharded chain"


# ---- GPU -----------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_hip_sharded_chain_with_emulated_ranks(hip_ctx, world):
    """N ranks on one device, one after the other, each with its own poisoned buffers; the two gathers are slot copies."""
    import torch
    from androidrenderer_amd import chain
    from tests import util
    W, H = 256, 150
    f = util.LightingFrame(W, H, seed=23, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")
    dev = f.device_arrays()
    ref = chain.ShardedChain(hip_ctx, f, dev, 0, 1)
    ref.step(gather=False)
    torch.cuda.synchronize()
    want = ref.out.cpu().numpy()
    ranks = [chain.ShardedChain(hip_ctx, f, dev, r, world) for r in range(world)]
    for c in ranks:
        for t in (c.lit, c.aa, c.mips[0], c.mip1_alloc):
            t.fill_(POISON16)
        c.out_alloc.fill_(0x5A)
        c.lighting()
        c.reduce()
    for c in ranks:  # exchange 1: rank r's slot r of mip 1 goes to everybody
        q = c.plan.mip1_rows_per_rank
        for src in ranks:
            c.mip1_alloc[src.plan.rank * q:(src.plan.rank + 1) * q] = src.mip1_alloc[src.plan.rank * q:(src.plan.rank + 1) * q]
    for c in ranks:
        c.composite()
    for c in ranks:  # exchange 2: rank r's slot world - 1 - r of the final image
        per = c.plan.rows_per_rank
        for src in ranks:
            s = src.plan.out_slot
            c.out_alloc[s * per:(s + 1) * per] = src.out_alloc[s * per:(s + 1) * per]
    torch.cuda.synchronize()
    for c in ranks:
        assert np.array_equal(c.out.cpu().numpy(), want), f"rank {c.plan.rank} of {world}: final image differs from the unsharded chain"


@pytest.mark.gpu
def test_hip_chain_through_a_one_rank_communicator():
    """The real exchange entry points (ncclAllGather on the communicator and on its reversed split) in the chain's order."""
    import torch
    from androidrenderer_amd import chain, lib
    from tests import util
    ctx = lib.Context(device=0, rank=0, world=1, comm_id=lib.comm_unique_id())
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    side = torch.cuda.Stream()
    try:
        f = util.LightingFrame(200, 117, seed=29, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_CACHE, flavour="atrium")
        dev = f.device_arrays()
        plain = chain.ShardedChain(ctx, f, dev, 0, 1)
        plain.step(gather=False)
        torch.cuda.synchronize()
        want = plain.out.cpu().numpy()
        ctx.comm_set_stream(side.cuda_stream)
        c = chain.ShardedChain(ctx, f, dev, 0, 1)
        c.step(gather=True)
        ctx.comm_wait()
        torch.cuda.synchronize()
        assert np.array_equal(c.out.cpu().numpy(), want)
        # and against the oracle's unsharded chain
        o = util.oracle()
        lit = f.run_oracle()
        aa = np.zeros_like(lit)
        assert o.orc_copy_scene(C.byref(images.plane(lit, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
        mips = [np.zeros((mh, mw, 4), np.uint16) for (mw, mh) in images.bloom_mip_sizes(200, 117, 6)]
        assert o.orc_bloom(C.byref(images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.mipchain(mips))) == 0
        out = np.zeros((117, 200, 4), np.uint8)
        assert o.orc_tonemap(C.byref(images.plane(aa, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.mipchain(mips)),
                             C.byref(images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)), 0, 0) == 0
        assert np.array_equal(want, out)
    finally:
        torch.cuda.synchronize()
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("native", [False, True, "graphs", "three-streams", "three-streams-graphs"],
                         ids=["python-loop", "library-loop", "library-loop-graphs", "library-loop-3-streams", "library-loop-3-streams-graphs"])
@pytest.mark.parametrize("no_split,two_streams", [(False, False), (True, False), (False, True), (True, True), (None, True), (None, False)])
def test_hip_pipelined_chain_keeps_its_frames_apart(monkeypatch, no_split, two_streams, native):
    """Two frames in flight (chain.PipelinedChain: both exchanges on the side stream of a one-rank communicator), a different AO plane
    per frame, warm-up / flush / more frames as bench.py drives it: every frame's final image equals the unpipelined chain's.
    `two_streams`: the B halves (mips 1.., tonemap, final exchange) on a second work stream beside the next frame's lighting.
    `native`: the same loop inside the library ("three-streams": lighting, copy + mip rows and mips 2.. + composite each on a stream of its own) (sah_chain_create / _submit / _flush through chain.NativePipelinedChain); "graphs": with its
    halves captured into HIP graphs and replayed (SAH_CHAIN_CAPTURE) — the AO plane still changes every frame, the descriptors do not."""
    import torch
    from androidrenderer_amd import chain, lib
    from tests import util
    if no_split:  # the reversed exchange without ncclCommSplit (grouped point-to-point on the parent communicator)
        monkeypatch.setenv("SAH_COMM_NO_SPLIT", "1")
    # (no_split None: no communicator at all — bench.py --frames-in-flight 2 on one GPU; the exchanges are no-ops)
    ctx = lib.Context(device=0, rank=0, world=1, comm_id=None if no_split is None else lib.comm_unique_id())
    previous_stream = torch.cuda.current_stream()
    if native in ("graphs", "three-streams-graphs"):
        torch.cuda.set_stream(torch.cuda.Stream())  # a work stream of its own: the null stream cannot be captured
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    side = torch.cuda.Stream()
    try:
        f = util.LightingFrame(160, 90, seed=33, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium", shadowmap_res=256)
        dev = f.device_arrays()
        g = torch.Generator(device="cpu").manual_seed(5)
        ao = [torch.rand((90, 160), generator=g).cuda() for _ in range(7)]
        plain = chain.ShardedChain(ctx, f, dev, 0, 1)
        want = []
        for a in ao:
            dev["ao"].copy_(a)
            plain.step(gather=False)
            torch.cuda.synchronize()
            want.append(plain.out.cpu().numpy().copy())
        assert not np.array_equal(want[0], want[1])
        if native:
            three = isinstance(native, str) and native.startswith("three-streams")
            pc = chain.NativePipelinedChain(ctx, f, dev, 0, 1, side, torch.cuda.Stream() if (two_streams or three) else None, capture=isinstance(native, str) and native.endswith("graphs"),
                                            reduce_stream=torch.cuda.Stream() if three else None)
        else:
            pc = chain.PipelinedChain(ctx, f, dev, 0, 1, side, torch.cuda.Stream() if two_streams else None)
        got = {}
        for i, a in enumerate(ao):
            dev["ao"].copy_(a)             # on the work stream: ordered with the frames around it
            pc.submit()
            if i in (2, 4):                # warm-up boundary as in bench.py: flush, then carry on
                pc.flush()
                torch.cuda.synchronize()
                got[i] = pc.image(i).cpu().numpy().copy()
                got[i - 1] = pc.image(i - 1).cpu().numpy().copy()
        pc.flush()
        ctx.comm_wait()
        torch.cuda.synchronize()
        got[6] = pc.image(6).cpu().numpy().copy()
        got[5] = pc.image(5).cpu().numpy().copy()
        for i, img in sorted(got.items()):
            assert np.array_equal(img, want[i]), f"frame {i}"
        assert pc.submitted == 7 and (native or pc.finished == 7)
        if isinstance(native, str) and native.endswith("graphs"):  # seven frames over two buffer sets: each part direct once, captured once, replayed afterwards
            replays, captures, failed = pc.graphs()
            assert not failed and captures >= 4 and replays >= 4, (replays, captures, failed)
        if native:
            pc.close()
    finally:
        torch.cuda.synchronize()
        torch.cuda.set_stream(previous_stream)
        ctx.close()


def test_library_source_row_rule_is_the_plans():
    """sah_bloom_source_rows (include/sah_hip.h): the C ABI's statement of which source rows a band of a bloom mip reads — the rule a C caller
    needs to shard an odd-height frame — is the one the Python plan uses (shard._downsample_sources), for every band of several pyramids."""
    import ctypes as C
    from androidrenderer_amd import lib, shard
    L = lib.load()
    out = (C.c_uint32 * 2)()
    for hs, hd in ((2160, 1080), (1080, 540), (75, 37), (37, 18), (149, 74), (9, 4), (3, 1), (2, 1), (1, 1)):
        for j0 in range(0, hd, max(1, hd // 7)):
            for j1 in (j0 + 1, min(hd, j0 + 5), hd):
                if j1 <= j0:
                    continue
                assert L.sah_bloom_source_rows(hs, hd, j0, j1, out) == 0
                lo, hi = shard._downsample_sources(j0, j1, hs, hd)
                assert (out[0], out[1]) == shard._clip(lo, hi, hs), (hs, hd, j0, j1)
    assert L.sah_bloom_source_rows(10, 5, 3, 3, out) == 0 and (out[0], out[1]) == (0, 0)  # an empty band reads nothing
    assert L.sah_bloom_source_rows(10, 5, 4, 3, out) != 0 and L.sah_bloom_source_rows(10, 5, 0, 6, out) != 0 and L.sah_bloom_source_rows(0, 5, 0, 1, out) != 0


def test_cpp_facade_plan_is_the_python_plan(tmp_path):
    """include/sah_host.hpp: sah::shard_chain_plan — what a C++ host hands to sah_chain_create — restates shard.chain_plan; every field of every
    rank's plan must agree, on 4K and 8K and on the odd heights whose mips are not exactly half as high (tests/cpp/shard_plan.cpp, CPU only)."""
    import subprocess
    from androidrenderer_amd import shard
    exe = str(tmp_path / "shard_plan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "shard_plan.cpp"), "-o", exe])
    cases = [(2160, 8), (2160, 4), (2160, 2), (2160, 1), (4320, 8), (1080, 6), (720, 16), (97, 2), (150, 3), (37, 3), (149, 4), (75, 8), (9, 4), (5, 7), (2, 2), (1, 3)]
    out = subprocess.check_output([exe] + [str(v) for c in cases for v in c], text=True)
    lines = iter(out.strip().splitlines())
    for height, world in cases:
        for r in range(world):
            got = [int(v) for v in next(lines).split()]
            p = shard.chain_plan(height, world, r)
            want = [height, world, r, *p.out_rows, *p.mip1_rows, *p.mip0_rows, *p.aa_rows, *p.lit_rows, *p.lit_wrap_rows, p.rows_per_rank, p.mip1_rows_per_rank,
                    p.rows_per_rank * world, p.mip1_rows_per_rank * world, p.mip0_height, p.mip1_height]
            assert got == want, (height, world, r, got, want)
