"""Shared test plumbing: the oracle binding (checker), frame builders that feed the same inputs to the oracle
(host pointers) and to the HIP library (device pointers), and ULP metrics on stored formats."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

from androidrenderer_amd import _abi, images, scene, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")

_oracle = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return ORACLE_SO


def oracle():
    """ctypes handle on oracle/liboracle.so — test infrastructure only."""
    global _oracle
    if _oracle is not None:
        return _oracle
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    o = C.CDLL(ORACLE_SO)
    o.orc_lighting.argtypes = [C.POINTER(_abi.LightingDesc)]
    o.orc_copy_scene.argtypes = [C.POINTER(_abi.Plane), C.POINTER(_abi.Plane)]
    o.orc_bloom.argtypes = [C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain)]
    o.orc_bloom_downsample.argtypes = [C.POINTER(_abi.Plane), C.POINTER(_abi.Plane)]
    o.orc_tonemap.argtypes = [C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain), C.POINTER(_abi.Plane), C.c_uint32, C.c_uint32]
    o.orc_lpv_clear.argtypes = [C.POINTER(_abi.Volume)] * 4 + [C.c_uint32]
    o.orc_lpv_propagate.argtypes = [C.POINTER(_abi.Volume), C.POINTER(_abi.Volume), C.c_uint32, C.c_uint32]
    o.orc_f32_to_f16.argtypes = [C.c_float]
    o.orc_f32_to_f16.restype = C.c_uint16
    o.orc_f16_to_f32.argtypes = [C.c_uint16]
    o.orc_f16_to_f32.restype = C.c_float
    o.orc_srgb8_to_linear.argtypes = [C.c_uint8]
    o.orc_srgb8_to_linear.restype = C.c_float
    o.orc_linear_to_srgb8.argtypes = [C.c_float]
    o.orc_linear_to_srgb8.restype = C.c_uint8
    o.orc_r11g11b10_decode.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
    o.orc_r11g11b10_encode.argtypes = [C.POINTER(C.c_float)]
    o.orc_r11g11b10_encode.restype = C.c_uint32
    for f in (o.orc_brdf_f32, o.orc_brdf_f16):
        f.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float),
                      C.POINTER(C.c_float)]
    o.orc_sample_bilinear.argtypes = [C.POINTER(_abi.Plane), C.c_float, C.c_float, C.c_int, C.POINTER(C.c_float)]
    o.orc_sample_trilinear.argtypes = [C.POINTER(_abi.Volume), C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_float)]
    o.orc_sample_shadow.argtypes = [C.POINTER(_abi.Volume), C.c_float, C.c_float, C.c_int, C.c_float]
    o.orc_sample_shadow.restype = C.c_float
    _oracle = o
    return o


# ---- ULP metrics -----------------------------------------------------------------------------------

def f16_ordered(bits):
    """Map fp16 bit patterns to integers that are monotone in the represented value (+0 == -0)."""
    b = bits.astype(np.int32)
    mag = b & 0x7FFF
    return np.where(b & 0x8000, -mag, mag)


def f16_ulp_diff(a_bits, b_bits):
    """Per-element |ULP distance| between two uint16 fp16 images; NaN vs NaN counts as 0, NaN vs number as 65535."""
    a, b = a_bits.astype(np.uint16), b_bits.astype(np.uint16)
    a_nan = (a & 0x7FFF) > 0x7C00
    b_nan = (b & 0x7FFF) > 0x7C00
    d = np.abs(f16_ordered(a) - f16_ordered(b))
    d = np.where(a_nan & b_nan, 0, d)
    d = np.where(a_nan ^ b_nan, 65535, d)
    return d


def report_ulp(name, d):
    n = d.size
    nz = int((d > 0).sum())
    return f"{name}: max {int(d.max())} ulp, {nz}/{n} differ ({100.0 * nz / n:.4f} %), >1ulp: {int((d > 1).sum())}"


# ---- device helpers ----------------------------------------------------------------------------------

def to_torch(a, device="cuda"):
    import torch
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16)).to(device)
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).to(device)
    return torch.from_numpy(a).to(device)


def from_torch(t, dtype):
    return t.cpu().numpy().view(dtype)


# ---- a lighting frame ---------------------------------------------------------------------------------

class LightingFrame:
    """All inputs of one Lighting pass as numpy arrays plus the uniform blocks; can describe itself to the oracle
    (host pointers) or to the HIP library (device pointers)."""

    def __init__(self, width, height, gbuffer=None, seed=1, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_NONE, sky=True, flavour="random",
                 flags=_abi.LIGHTING_DEFAULT_FLAGS, shadowmap_res=256, lights=None, cache_debug_mode=0, num_extra_rays=0):
        self.width, self.height = width, height
        self.view = scene.SceneView.default(width, height)
        self.sun = scene.DirectionalLight(shadow_mode=sun_mode)
        self.flags = flags
        self.gi_kind = gi
        if gbuffer is None:
            gbuffer = synth.random_gbuffer(width, height, seed) if flavour == "random" else synth.atrium_gbuffer(width, height, self.view, seed)
        self.arrays = dict(gbuffer)
        self.arrays["ao"] = synth.ao_plane(width, height, seed + 100)
        self.has_sky = sky
        if sky:
            luts = synth.sky_luts(seed + 200)
            self.arrays["sky_t"] = luts["transmittance"]
            self.arrays["sky_v"] = luts["sky_view"]
        if sun_mode == _abi.SHADOW_MODE_CSM:
            self.sun.update_shadow_cascades(self.view, resolution=shadowmap_res)
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
                gi.lpv_num_cascades = 4
                gi.lpv_exposure = float(np.float32(math.pi) * np.float32(10.0))
            elif self.gi_kind == _abi.GI_CACHE:
                gi.probe_irradiance = images.volume(arrays["probe_irr"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32)
                gi.probe_depth = images.volume(arrays["probe_depth"], _abi.FORMAT_R16G16_SFLOAT)
                gi.probe_validity = images.volume(arrays["probe_val"], _abi.FORMAT_R8_UNORM)
                pos = self.view.position
                for c in range(4):
                    spacing = 0.5 * (2.0 ** c)  # stand-in placement: cascades centred on the camera
                    gi.probe_cascades[c].probe_spacing = spacing
                    ext = (32 * spacing, 8 * spacing, 32 * spacing)
                    for i in range(3):
                        gi.probe_cascades[c].min[i] = float(pos[i]) - ext[i] / 2.0 + 0.013 * (c + 1)
                gi.probe_size[0], gi.probe_size[1] = 5, 6  # irradiance_cache.cpp:298-299
                gi.cache_debug_mode = self.cache_debug_mode
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

    def inputs_sha256(self):
        import hashlib
        m = hashlib.sha256()
        for k in sorted(self.arrays):
            m.update(k.encode())
            m.update(np.ascontiguousarray(self.arrays[k]).tobytes())
        return m.hexdigest()

    def run_oracle(self):
        lit = np.zeros((self.height, self.width, 4), dtype=np.uint16)
        d, keep = self.describe(self.arrays, lit)
        rc = oracle().orc_lighting(C.byref(d))
        assert rc == 0, rc
        return lit

    def device_arrays(self):
        return {k: to_torch(v) for k, v in self.arrays.items()}

    def run_hip(self, ctx, dev=None):
        import torch
        dev = dev or self.device_arrays()
        lit = torch.zeros((self.height, self.width, 4), dtype=torch.int16, device="cuda")
        d, keep = self.describe(dev, lit)
        ctx.lighting(d)
        torch.cuda.synchronize()
        return from_torch(lit, np.uint16)


def golden_lighting_frame(width, height, seed, sun_mode, gi):
    """The inputs behind tests/golden/lighting_*.npz (tools/gen_golden.py): atrium G-buffer, no sky LUTs, a 128² CSM, and
    roughness bytes >= 1 so the LPV specular quirk term is exactly zero (the golden generator does not model it)."""
    f = LightingFrame(width, height, seed=seed, sun_mode=sun_mode, gi=gi, flavour="atrium", sky=False, shadowmap_res=128)
    f.arrays["data"] = f.arrays["data"].copy()
    f.arrays["data"][..., 1] = np.maximum(f.arrays["data"][..., 1], 1)
    return f
