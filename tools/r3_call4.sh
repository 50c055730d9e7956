#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r3_call4
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_post_gpu.py -x -q -m gpu -s -k "tolerance" > $O/tests.log 2>&1; echo "tests rc=$?"; grep "histogram\|passed\|failed\|Error" $O/tests.log | tail -30
python3 - <<'P' 2>&1 | tail -20
import torch, numpy as np, sys
sys.path.insert(0,'.')
from androidrenderer_amd import _abi, images, lib, synth
from tests import util
ctx = lib.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
W,H=3840,2160
scene = util.to_torch(synth.hdr_scene(W,H,seed=5).view(np.uint16))
mips=[torch.zeros((mh,mw,4),dtype=torch.int16,device='cuda') for (mw,mh) in images.bloom_mip_sizes(W,H,6)]
sp=images.plane(scene,_abi.FORMAT_R16G16B16A16_SFLOAT); mc=images.mipchain(mips)
ctx.bloom(sp,mc)
outs=[]
for flags in (0,1):
    out=torch.zeros((H,W,4),dtype=torch.uint8,device='cuda'); op=images.plane(out,_abi.FORMAT_R8G8B8A8_SRGB)
    for _ in range(20): ctx.tonemap(sp,mc,op,flags=flags)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ctx.tonemap(sp,mc,op,flags=flags)
    e1.record(); torch.cuda.synchronize()
    print('flags',flags,'ms',e0.elapsed_time(e1)/100)
    outs.append(out.cpu().numpy())
d=np.abs(outs[0].astype(int)-outs[1].astype(int)); print('4K hist', np.bincount(d.reshape(-1),minlength=3)[:4], 'max', d.max())
P
