"""One rank of tests/test_comm_gpu.py::test_two_rank_gather_equals_unsharded: a fresh process per GPU (nothing has touched the GPU
before this file runs), library-owned communicator, HIP-shaded rows, in-place RCCL all-gather, result saved for the parent."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, height, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    from androidrenderer_amd import _abi, images, lib
    from tests import util

    torch.cuda.set_device(rank)
    comm_id = open(os.path.join(out_dir, "id.bin"), "rb").read()
    ctx = lib.Context(device=rank, rank=rank, world=world, comm_id=comm_id)
    with torch.cuda.device(rank):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        f = util.LightingFrame(160, height, seed=21, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_LPV, flavour="atrium")
        dev = f.device_arrays(f"cuda:{rank}")
        per = -(-height // world)
        r0 = min(rank * per, height)
        r1 = min(r0 + per, height)
        lit = torch.zeros((per * world, f.width, 4), dtype=torch.int16, device=f"cuda:{rank}")
        f.row_begin, f.row_end = r0, r1
        d, keep = f.describe(dev, lit[:height])
        if r1 > r0:
            ctx.lighting(d)
        ctx.allgather_rows(images.plane(lit[:height], _abi.FORMAT_R16G16B16A16_SFLOAT), per, per * world)
        ctx.sync()
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), util.from_torch(lit[:height], np.uint16))
    ctx.close()


if __name__ == "__main__":
    main()
