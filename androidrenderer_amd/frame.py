"""One Lighting pass worth of inputs (planes, side tables, uniform blocks) and its `sah_lighting_desc` — input plumbing
shared by bench.py, tools/ and tests/.  The same object describes itself over host arrays (numpy) or device arrays (torch)."""
import ctypes as C
import hashlib
import math

import numpy as np

from . import _abi, images, scene, synth


def to_torch(a, device="cuda"):
    import torch
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16)).to(device)
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).to(device)
    return torch.from_numpy(a).to(device)


def from_torch(t, dtype):
    return t.cpu().numpy().view(dtype)


class LightingInputs:
    def __init__(self, width, height, gbuffer=None, seed=1, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_NONE, sky=True, flavour="random",
                 flags=_abi.LIGHTING_DEFAULT_FLAGS, shadowmap_res=256, lights=None, cache_debug_mode=0, num_extra_rays=0, synth_device="cpu",
                 shadow="noise"):
        self.width, self.height = width, height
        self.view = scene.SceneView.default(width, height)
        self.sun = scene.DirectionalLight(shadow_mode=sun_mode)
        self.sun_mode = sun_mode
        self.flags = flags
        self.gi_kind = gi
        if gbuffer is None:
            gbuffer = (synth.random_gbuffer(width, height, seed) if flavour == "random"
                       else synth.atrium_gbuffer(width, height, self.view, seed, device=synth_device))
        self.arrays = dict(gbuffer)
        self.arrays["ao"] = synth.ao_plane(width, height, seed + 100)
        self.has_sky = sky
        if sky:
            luts = synth.sky_luts(seed + 200)
            self.arrays["sky_t"] = luts["transmittance"]
            self.arrays["sky_v"] = luts["sky_view"]
        if sun_mode == _abi.SHADOW_MODE_CSM:
            self.sun.update_shadow_cascades(self.view, resolution=shadowmap_res)
            if shadow == "scene" and flavour == "atrium":  # depth of the atrium's own boxes as seen from the sun
                self.arrays["shadowmap"] = synth.atrium_shadowmap(self.sun.constants, shadowmap_res, 4, device=synth_device)
            else:  # D16 noise in [0.3, 0.7]
                self.arrays["shadowmap"] = synth.shadowmap(shadowmap_res, 4, seed + 300)
        if sun_mode == _abi.SHADOW_MODE_RT:
            self.arrays["shadow_mask"] = synth.shadow_mask(width, height, seed + 400)
        self.lpv = None
        if gi == _abi.GI_LPV:
            self.lpv = scene.LpvCascades()
            self.lpv.update_cascade_transforms(self.view, self.sun)
            r, g, b = synth.lpv_volumes(4, seed + 500)
            self.arrays["lpv_r"], self.arrays["lpv_g"], self.arrays["lpv_b"] = r, g, b
        if gi == _abi.GI_CACHE:
            at = synth.probe_atlases(seed + 600)
            self.arrays["probe_irr"], self.arrays["probe_depth"], self.arrays["probe_val"] = at["irradiance"], at["depth"], at["validity"]
        if gi == _abi.GI_RTGI:
            rt = synth.rtgi_planes(width, height, seed + 700)
            self.arrays["ray_buffer"], self.arrays["ray_irr"], self.arrays["noise"] = rt["ray_buffer"], rt["ray_irradiance"], rt["noise"]
        self.cache_debug_mode = cache_debug_mode
        self.num_extra_rays = num_extra_rays
        self.lights = lights  # (N, 8) float32 or None
        if lights is not None:
            self.arrays["lights"] = np.ascontiguousarray(lights, dtype=np.float32)
        self.row_begin = self.row_end = 0

    def bytes_per_pixel(self):
        """Algorithmic plane traffic of one pass: G-buffer 24 B read + lit 8 B written, plus the per-pixel
        input planes the selected sub-passes read (AO 4 — LPV overlay only, shadow mask 4, RTGI ray buffer 8 + irradiance 8)."""
        b = 24 + 8
        if self.gi_kind == _abi.GI_LPV:
            b += 4
        if self.sun_mode == _abi.SHADOW_MODE_RT:
            b += 4
        if self.gi_kind == _abi.GI_RTGI:
            b += 16
        return b

    def describe(self, arrays, lit):
        """Build a LightingDesc over `arrays` (numpy or torch, same keys) writing into `lit`. Returns (desc, keepalive)."""
        keep = []
        gb = images.gbuffer(arrays)
        lit_p = images.plane(lit, _abi.FORMAT_R16G16B16A16_SFLOAT)
        ao_p = images.plane(arrays["ao"], _abi.FORMAT_R32_SFLOAT)
        d = _abi.LightingDesc()
        d.gbuffer = C.pointer(gb)
        d.lit = C.pointer(lit_p)
        d.ao = C.pointer(ao_p)
        d.view = C.pointer(self.view.gpu_data)
        d.sun = C.pointer(self.sun.constants)
        keep += [gb, lit_p, ao_p]
        if "shadowmap" in arrays:
            sm = images.volume(arrays["shadowmap"], _abi.FORMAT_D16_UNORM)
            d.shadowmap = C.pointer(sm)
            keep.append(sm)
        if "shadow_mask" in arrays:
            m = images.plane(arrays["shadow_mask"], _abi.FORMAT_R32_SFLOAT)
            d.shadow_mask = C.pointer(m)
            keep.append(m)
        if self.has_sky:
            sk = _abi.SkyLuts(images.plane(arrays["sky_t"], _abi.FORMAT_R16G16B16A16_SFLOAT),
                              images.plane(arrays["sky_v"], _abi.FORMAT_R16G16B16A16_SFLOAT))
            d.sky = C.pointer(sk)
            keep.append(sk)
        if self.gi_kind != _abi.GI_NONE:
            gi = _abi.GI()
            gi.kind = self.gi_kind
            if self.gi_kind == _abi.GI_LPV:
                gi.lpv_red = images.volume(arrays["lpv_r"], _abi.FORMAT_R16G16B16A16_SFLOAT)
                gi.lpv_green = images.volume(arrays["lpv_g"], _abi.FORMAT_R16G16B16A16_SFLOAT)
                gi.lpv_blue = images.volume(arrays["lpv_b"], _abi.FORMAT_R16G16B16A16_SFLOAT)
                gi.lpv_cascades = C.cast(self.lpv.matrices, C.POINTER(_abi.LpvCascadeMatrices))
                gi.lpv_num_cascades = getattr(self, "lpv_num_cascades", 4)  # (volumes are (32 * cascades) x 32 x 32)
                gi.lpv_exposure = float(np.float32(math.pi) * np.float32(10.0))
                gi.lpv_generation = getattr(self, "lpv_generation", 0)  # 0: the library rebuilds its gather copy of the volumes on every call
            elif self.gi_kind == _abi.GI_CACHE:
                gi.probe_irradiance = images.volume(arrays["probe_irr"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32)
                gi.probe_depth = images.volume(arrays["probe_depth"], _abi.FORMAT_R16G16_SFLOAT)
                gi.probe_validity = images.volume(arrays["probe_val"], _abi.FORMAT_R8_UNORM)
                for c, (cmin, spacing) in enumerate(self.probe_cascades()):
                    gi.probe_cascades[c].probe_spacing = spacing
                    for i in range(3):
                        gi.probe_cascades[c].min[i] = cmin[i]
                gi.probe_size[0], gi.probe_size[1] = 5, 6  # irradiance_cache.cpp:298-299
                gi.cache_debug_mode = self.cache_debug_mode
                gi.probe_generation = getattr(self, "probe_generation", 0)  # 0: the library rebuilds its fp32 gather copy on every call
            elif self.gi_kind == _abi.GI_RTGI:
                gi.ray_buffer = images.plane(arrays["ray_buffer"], _abi.FORMAT_R16G16B16A16_SFLOAT)
                gi.ray_irradiance = images.plane(arrays["ray_irr"], _abi.FORMAT_R16G16B16A16_SFLOAT)
                gi.noise = images.plane(arrays["noise"], _abi.FORMAT_R8G8B8A8_UNORM)
                gi.num_extra_rays = self.num_extra_rays
                gi.extra_ray_radius = 16.0
            d.gi = C.pointer(gi)
            keep.append(gi)
        if self.lights is not None:
            la = arrays["lights"]
            ptr = la.ctypes.data if isinstance(la, np.ndarray) else la.data_ptr()
            ll = _abi.LightList(ptr, self.lights.shape[0])
            d.lights = C.pointer(ll)
            keep.append(ll)
        d.flags = self.flags
        d.row_begin, d.row_end = self.row_begin, self.row_end
        keep.append(arrays)
        return d, keep

    def probe_cascades(self):
        """Stand-in placement of the four irradiance-cache cascades: centred on the camera, spacing 0.5 m doubling per cascade.
        Returns [(min xyz, spacing)] as Python floats (the ABI struct rounds them to fp32)."""
        pos = self.view.position
        out = []
        for c in range(4):
            spacing = 0.5 * (2.0 ** c)
            ext = (32 * spacing, 8 * spacing, 32 * spacing)
            out.append(([float(pos[i]) - ext[i] / 2.0 + 0.013 * (c + 1) for i in range(3)], spacing))
        return out

    def inputs_sha256(self):
        m = hashlib.sha256()
        for k in sorted(self.arrays):
            m.update(k.encode())
            m.update(np.ascontiguousarray(self.arrays[k]).tobytes())
        return m.hexdigest()

    def device_arrays(self, device="cuda"):
        return {k: to_torch(v, device) for k, v in self.arrays.items()}

    def run_hip(self, ctx, dev=None):
        import torch
        dev = dev or self.device_arrays()
        lit = torch.zeros((self.height, self.width, 4), dtype=torch.int16, device="cuda")
        d, keep = self.describe(dev, lit)
        ctx.lighting(d)
        torch.cuda.synchronize()
        return from_torch(lit, np.uint16)
