"""Debug aid: tonemap HIP vs oracle with a single live bloom mip (others zero), tells which mip / which pixels differ."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from androidrenderer_amd import _abi, images, synth, lib
from tests import util

w, h = 256, 144
ctx = lib.Context()
o = util.oracle()
scene = synth.hdr_scene(w, h, seed=21).view(np.uint16)
sizes = images.bloom_mip_sizes(w, h, 6)
mips_ref = [np.zeros((mh, mw, 4), dtype=np.uint16) for (mw, mh) in sizes]
chain = images.mipchain(mips_ref)
sp = images.plane(scene, _abi.FORMAT_R16G16B16A16_SFLOAT)
assert o.orc_bloom(C.byref(sp), C.byref(chain)) == 0
for live in list(range(6)) + [None]:
    ms = [m.copy() if (live is None or i == live) else np.zeros_like(m) for i, m in enumerate(mips_ref)]
    ms = [m * 0 + m * 40 if False else m for m in ms]
    ch = images.mipchain(ms)
    ref = np.zeros((h, w, 4), dtype=np.uint8)
    zs = np.zeros_like(scene)
    zp = images.plane(zs, _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert o.orc_tonemap(C.byref(zp), C.byref(ch), C.byref(images.plane(ref, _abi.FORMAT_R8G8B8A8_SRGB)), 0, 0) == 0
    tm = [util.to_torch(m) for m in ms]
    tch = images.mipchain(tm)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    ctx.tonemap(images.plane(util.to_torch(zs), _abi.FORMAT_R16G16B16A16_SFLOAT), tch, images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    d = np.abs(got.astype(int) - ref.astype(int))[..., :3].max(axis=2)
    ys, xs = np.nonzero(d)
    print("live mip", live, "max diff", d.max(), "count", len(ys), "ref max code", ref[..., :3].max())
    if len(ys):
        print("   x range", xs.min(), xs.max(), "y range", ys.min(), ys.max(), "cols mod 32 hist", np.bincount(xs % 32, minlength=32).tolist())
        print("   rows mod 32 hist", np.bincount(ys % 32, minlength=32).tolist())
        print("   sample got/ref", got[ys[0], xs[0]], ref[ys[0], xs[0]], (xs[0], ys[0]))
