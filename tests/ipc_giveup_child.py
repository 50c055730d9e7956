"""One rank of tests/test_comm_gpu.py::test_direct_exchange_gives_up_and_recovers: two fresh processes share GPU 0 and gather a small buffer
through the direct exchange (sah_ipc_*).  Gather 1 works.  Gather 2 is made by rank 0 only: rank 1 is "busy" for three seconds.  Expected
(include/sah_hip.h, sah_ipc_reset): rank 0's sah_sync returns SAH_ERR_COMM after about two seconds, its next gather fails at once, and it
has copied nothing into rank 1's buffer; rank 1, arriving late, finds its own ready-wait satisfied, does NOT copy into rank 0 (which has
posted its give-up note), and gives up on the done-wait in turn.  Then the documented way out — sync, barrier, sah_ipc_reset, barrier —
and gather 3 works on both.  With a fourth argument "pipelined" rank 0 enqueues THREE gathers before it looks (the ordinary pipelined
case: the host learns of the give-up two seconds late; the later gathers are counted on rank 0 and never signalled — ADVICE r5), and two
gathers must work after the reset.  Results go to <out_dir>/rank<r>.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, out_dir, port = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    pipelined = len(sys.argv) > 4 and sys.argv[4] == "pipelined"
    world = 2
    import torch
    import torch.distributed as dist
    from androidrenderer_amd import _abi, chain, lib

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

    def allgather(b):
        out = [None] * world
        dist.all_gather_object(out, b)
        return out

    torch.cuda.set_device(0)
    ctx = lib.Context(device=0, rank=rank, world=world, comm_id=None)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    chain.connect_direct_exchange(ctx, allgather)
    n = 1 << 16  # bytes per rank
    buf = torch.zeros(2 * n, dtype=torch.uint8, device="cuda")
    ctx.ipc_register(buf.data_ptr(), 2 * n, allgather(ctx.ipc_export(buf.data_ptr(), 2 * n)))
    res = {}

    def fill(tag):
        buf.fill_(0xEE)
        buf[rank * n:(rank + 1) * n] = tag + rank
        torch.cuda.synchronize()

    def gathered_ok(tag):
        return bool((buf[:n] == tag).all()) and bool((buf[n:] == tag + 1).all())

    # gather 1: both ranks
    fill(10)
    dist.barrier()
    ctx.allgather_bytes(buf.data_ptr(), n)
    ctx.sync()
    res["gather1_ok"] = gathered_ok(10)
    dist.barrier()
    # gather 2: rank 0 alone; rank 1 arrives three seconds late
    fill(20)
    dist.barrier()
    if rank == 1:
        time.sleep(3.0)
    t0 = time.perf_counter()
    for _ in range(3 if (pipelined and rank == 0) else 1):
        ctx.allgather_bytes(buf.data_ptr(), n)
    try:
        ctx.sync()
        res["gather2_status"] = 0
    except lib.SahError as e:
        res["gather2_status"] = e.status
    res["gather2_seconds"] = time.perf_counter() - t0
    torch.cuda.synchronize()
    other = 1 - rank
    res["peer_slot_untouched"] = bool((buf[other * n:(other + 1) * n] == 0xEE).all())  # nobody copied into a rank that had given up / been given up on
    try:  # sticky: the next gather fails at once
        t1 = time.perf_counter()
        ctx.allgather_bytes(buf.data_ptr(), n)
        res["gather_after_giveup_status"] = 0
    except lib.SahError as e:
        res["gather_after_giveup_status"] = e.status
    res["gather_after_giveup_seconds"] = time.perf_counter() - t1
    # the way out: drain, meet, reset, meet
    try:
        ctx.sync()
    except lib.SahError:
        pass
    dist.barrier()
    ctx.ipc_reset()
    dist.barrier()
    fill(30)
    dist.barrier()
    ctx.allgather_bytes(buf.data_ptr(), n)
    ctx.sync()
    res["gather3_ok"] = gathered_ok(30)
    dist.barrier()
    fill(40)
    dist.barrier()
    ctx.allgather_bytes(buf.data_ptr(), n)
    ctx.sync()
    res["gather4_ok"] = gathered_ok(40)
    dist.barrier()
    ctx.ipc_unregister(buf.data_ptr())
    json.dump(res, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
