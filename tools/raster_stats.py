"""Prints the rasteriser's statistics words for the atrium at two tessellations (records, bin entries): sizing aid for DESIGN.md."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from androidrenderer_amd import _abi, images, lib, mesh, scene  # noqa: E402

W, H = 3840, 2160
ctx = lib.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
view = scene.SceneView.default(W, H)
sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_CSM)
constants = sun.update_shadow_cascades(view, resolution=4096)
gb = {"color": torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"), "normals": torch.zeros((H, W, 4), dtype=torch.int16, device="cuda"),
      "data": torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"), "emission": torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"),
      "depth": torch.zeros((H, W), dtype=torch.float32, device="cuda")}
sm = torch.zeros((4, 4096, 4096), dtype=torch.int16, device="cuda")
stats = torch.zeros(8, dtype=torch.int32, device="cuda")
for subdiv in (1, 24):
    dev = mesh.to_device(mesh.atrium(subdiv).arrays())
    geo = mesh.geometry(dev, [])
    ctx.gbuffer_render(geo, view.gpu_data, images.gbuffer(gb), stats.data_ptr())
    torch.cuda.synchronize()
    print("gbuffer", subdiv, stats.cpu().numpy().tolist(), "covered", float((gb["depth"] > 0).float().mean()))
    ctx.shadow_render(geo, constants, 4, images.volume(sm, _abi.FORMAT_D16_UNORM), stats.data_ptr())
    torch.cuda.synchronize()
    print("shadow ", subdiv, stats.cpu().numpy().tolist(), "covered per cascade", [float((sm[c] != -1).float().mean()) for c in range(4)])
