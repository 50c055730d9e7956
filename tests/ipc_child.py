"""One rank of tests/test_comm_gpu.py::test_two_ranks_on_one_gpu_direct_exchange: a fresh process (nothing has touched the GPU before
this file runs).  Both ranks share GPU 0 — RCCL refuses two ranks on one device, HIP IPC does not — and run the sharded full chain with
the direct exchange backend: once step by step, then five frames through PipelinedChain (two work streams, a side stream, two frames
in flight).  Handles travel between the ranks over a gloo process group.  The final images go to <out_dir>/rank<r>_*.npy.

With a sixth argument "ipc2" (test_two_rank_chain_direct_exchange_across_devices, needs two GPUs) the direct exchange runs one rank per
GPU: the stores then cross a link, and the peer's consumer kernels read them from another device's memory system.
With a sixth argument "rccl" (test_two_rank_chain_through_rccl, needs two GPUs) the same frames run one rank per GPU through the library's
RCCL communicator instead — the parent communicator for bloom mip 0, the reversed one (ncclCommSplit, or grouped send / recv under
SAH_COMM_NO_SPLIT=1, which the parent test sets in the environment) for the final image."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, height, out_dir, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    rccl = len(sys.argv) > 6 and sys.argv[6] == "rccl"
    two_devices = rccl or (len(sys.argv) > 6 and sys.argv[6] == "ipc2")  # "ipc2": the direct exchange, one rank per GPU (two GPUs)
    import torch
    import torch.distributed as dist
    from androidrenderer_amd import _abi, chain, lib
    from tests import util

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

    def allgather(b):
        out = [None] * world
        dist.all_gather_object(out, b)
        return out

    device = rank if two_devices else 0
    torch.cuda.set_device(device)
    comm_id = None
    if rccl:
        comm_id = allgather(lib.comm_unique_id() if rank == 0 else None)[0]
    ctx = lib.Context(device=device, rank=rank, world=world, comm_id=comm_id)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if not rccl:
        chain.connect_direct_exchange(ctx, allgather)
    f = util.LightingFrame(160, height, seed=37, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_CACHE, flavour="atrium")
    dev = f.device_arrays()
    # 1. the chain step by step, exchanges on the work stream
    sc = chain.ShardedChain(ctx, f, dev, rank, world)
    for t in (sc.lit, sc.aa, sc.mips[0], sc.mip1_alloc):
        t.fill_(0x7E01)  # an fp16 NaN: a row nobody computed or received shows
    sc.out_alloc.fill_(0x5A)
    if not rccl:
        sc.register_direct_exchange(allgather)
    sc.step()
    ctx.sync()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"rank{rank}_step.npy"), sc.out.cpu().numpy())
    if not rccl:  # the registrations go back before the pipelined chain takes their slots (sah_ipc_unregister: lowest free slot next)
        dist.barrier()  # nobody closes a mapping a peer may still be copying through
        sc.unregister_direct_exchange()
        try:
            ctx.ipc_unregister(sc.out_alloc.data_ptr())
            raise AssertionError("unregistering twice must fail")
        except lib.SahError:
            pass
    # 2. two frames in flight, a different shadow mask per frame
    side, second = torch.cuda.Stream(), torch.cuda.Stream()
    pc = chain.PipelinedChain(ctx, f, dev, rank, world, side, second)
    if not rccl:
        pc.register_direct_exchange(allgather)
    g = torch.Generator(device="cpu").manual_seed(11)
    masks = [torch.rand((height, 160), generator=g).cuda() for _ in range(5)]
    images_out = {}
    for i, m in enumerate(masks):
        dev["shadow_mask"].copy_(m)
        pc.submit()
        if i == 2:
            pc.flush()
            torch.cuda.synchronize()
            images_out[1] = pc.image(1).cpu().numpy().copy()
            images_out[2] = pc.image(2).cpu().numpy().copy()
    pc.flush()
    ctx.comm_wait()
    torch.cuda.synchronize()
    images_out[3] = pc.image(3).cpu().numpy().copy()
    images_out[4] = pc.image(4).cpu().numpy().copy()
    for i, img in images_out.items():
        np.save(os.path.join(out_dir, f"rank{rank}_frame{i}.npy"), img)
    dist.barrier()  # nobody unmaps a buffer a peer may still be copying into
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
