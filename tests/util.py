"""Shared test plumbing: the oracle binding (checker), frame builders that feed the same inputs to the oracle
(host pointers) and to the HIP library (device pointers), and ULP metrics on stored formats."""
import ctypes as C
import os
import subprocess

import numpy as np

from androidrenderer_amd import _abi, frame, synth
from androidrenderer_amd.frame import from_torch, to_torch  # noqa: F401  (re-exported for the tests)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("SAH_ORACLE_SO") or os.path.join(ORACLE_DIR, "liboracle.so")  # (override: the sanitizer build, tools/sanitize.sh)

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
    o.orc_sky_update_luts.argtypes = [C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.POINTER(C.c_float)]
    o.orc_dir_to_sh.argtypes =[C.POINTER(C.c_float), C.POINTER(C.c_float)]
    o.orc_octahedral_coordinates.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    o.orc_probe_uv.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
    o.orc_octahedral_direction_of_texel.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    o.orc_viewspace_position_glsl.argtypes = [C.POINTER(_abi.ViewData), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)]
    o.orc_worldspace_location_slang.argtypes = [C.POINTER(_abi.ViewData), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)]
    o.orc_probe_copy.argtypes =[C.POINTER(_abi.ProbeAtlases), C.POINTER(_abi.ProbeAtlases), C.POINTER(C.c_float * 3)]
    o.orc_probe_update.argtypes = [C.POINTER(_abi.ProbeAtlases), C.POINTER(_abi.Volume), C.c_void_p, C.c_uint32]
    o.orc_shadow_render.argtypes = [C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.SunLightConstants), C.c_uint32, C.POINTER(_abi.Volume), C.c_void_p]
    o.orc_gbuffer_render.argtypes = [C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.ViewData), C.POINTER(_abi.GBuffer), C.c_void_p]
    o.orc_rsm_render.argtypes = [C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.SunLightConstants), C.POINTER(_abi.LpvCascadeMatrices), C.c_uint32,
                                 C.POINTER(_abi.RsmTargets), C.c_void_p]
    o.orc_lpv_extract_vpls.argtypes = [C.POINTER(_abi.RsmTargets), C.POINTER(_abi.LpvCascadeMatrices), C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]
    o.orc_lpv_inject_vpls.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(_abi.LpvCascadeMatrices), C.c_uint32, C.c_uint32, C.POINTER(_abi.Volume)]
    o.orc_rt_stats.argtypes = [C.POINTER(_abi.SceneGeometry), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    o.orc_rtao.argtypes = [C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.ViewData), C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.c_uint32,
                           C.c_float, C.POINTER(_abi.Plane)]
    o.orc_sun_shadow_mask.argtypes = [C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.ViewData), C.POINTER(_abi.SunLightConstants), C.POINTER(_abi.Plane),
                                      C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.POINTER(_abi.Plane)]
    o.orc_probe_trace.argtypes = [C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.ProbeTraceDesc)]
    o.orc_rt_set_bounces.argtypes = [C.c_uint32]
    o.orc_rtgi_trace.argtypes = [C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.ViewData), C.POINTER(_abi.SunLightConstants), C.POINTER(_abi.SkyLuts)] + \
        [C.POINTER(_abi.Plane)] * 5
    _oracle = o
    return o


def probe_atlases_desc(arrays):
    """dict of numpy / torch atlases (synth.probe_maintenance_inputs) -> _abi.ProbeAtlases"""
    from androidrenderer_amd import images
    return _abi.ProbeAtlases(images.volume(arrays["rtgi"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32),
                             images.volume(arrays["light_cache"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32),
                             images.volume(arrays["depth"], _abi.FORMAT_R16G16_SFLOAT),
                             images.volume(arrays["average"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32),
                             images.volume(arrays["validity"], _abi.FORMAT_R8_UNORM))


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


# ---- a lighting frame ---------------------------------------------------------------------------------

class LightingFrame(frame.LightingInputs):
    """androidrenderer_amd.frame.LightingInputs (inputs + sah_lighting_desc over host or device arrays) plus the checker."""

    def run_oracle(self):
        lit = np.zeros((self.height, self.width, 4), dtype=np.uint16)
        d, keep = self.describe(self.arrays, lit)
        rc = oracle().orc_lighting(C.byref(d))
        assert rc == 0, rc
        return lit


def golden_lighting_frame(width, height, seed, sun_mode, gi, sky=False):
    """The inputs behind tests/golden/lighting_*.npz (tools/gen_golden.py): atrium G-buffer, no sky LUTs, a 128² CSM, and
    roughness bytes >= 1 so the LPV specular quirk term is exactly zero (the golden generator does not model it)."""
    f = LightingFrame(width, height, seed=seed, sun_mode=sun_mode, gi=gi, flavour="atrium", sky=sky, shadowmap_res=128)
    f.arrays["data"] = f.arrays["data"].copy()
    f.arrays["data"][..., 1] = np.maximum(f.arrays["data"][..., 1], 1)
    return f


def golden_raster_scene(anisotropic=False):
    """(anisotropic=True: the same scene with anisotropic samplers, 8x / 2.5x / 16x / 4x — tests/golden/raster_gbuffer_aniso_64x36.npz.)
    The scene behind tests/golden/raster_gbuffer_64x36.npz (tools/gen_golden.py): a textured, alpha-tested wall seen at an angle
    (two CUTOUT triangles, every material slot bound to a texture with its own sampler), a SOLID triangle in front of it drawn twice
    at the same depth with two materials (the later draw stays) and once with the opposite winding (culled), random vertex colours,
    normals and tangents.  Returns (mesh.Mesh, scene.SceneView); nothing is clipped."""
    from androidrenderer_amd import mesh, scene
    g = synth.rng(120)
    view = scene.SceneView.default(64, 36)  # at (-7, 1, 0) looking along +x
    view.gpu_data.material_texture_mip_bias = 0.25
    m = mesh.Mesh()
    an = (8.0, 2.5, 16.0, 4.0) if anisotropic else (0.0, 0.0, 0.0, 0.0)
    tex = [m.add_texture(*mesh.random_texture(g, 32, 32, None, True, mesh.sampler(max_anisotropy=an[0]), alpha=(0, 256))),  # base colour: trilinear REPEAT
           m.add_texture(*mesh.random_texture(g, 16, 8, None, False, mesh.sampler(mag=0, min=1, mipmap=0, max_anisotropy=an[1],
                                                                                     address_u=_abi.ADDRESS_MIRRORED_REPEAT, address_v=_abi.ADDRESS_CLAMP_TO_EDGE))),
           m.add_texture(*mesh.random_texture(g, 64, 64, 3, False, mesh.sampler(mag=1, min=0, mipmap=1, bias=-0.5, max_lod=1.5, max_anisotropy=an[2]))),
           m.add_texture(*mesh.random_texture(g, 8, 8, None, True, mesh.sampler(mag=0, min=0, mipmap=0, min_lod=1.0, max_anisotropy=an[3])))]
    wall = m.add_material(mesh.material(base=(0.9, 0.8, 1.0, 1.0), rough=0.7, metal=0.4, emission=(1.5, 0.5, 2.0, 0.0), opacity_threshold=0.35),
                          base_color=tex[0], normal=tex[1], data=tex[2], emission=tex[3])
    red = m.add_material(mesh.material(base=(1.0, 0.1, 0.1, 1.0), rough=0.2, metal=0.9))
    blue = m.add_material(mesh.material(base=(0.1, 0.1, 1.0, 1.0), rough=0.9, metal=0.0, normal_texel=(0.4, 0.6, 1.0, 1.0)), data=tex[2])

    def attrs(n):
        nrm = g.normal(size=(n, 3)).astype(np.float32)
        tan = np.concatenate([g.normal(size=(n, 3)), g.choice([-1.0, 1.0], (n, 1))], axis=1).astype(np.float32)
        col = g.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        return nrm, tan, col
    # the wall: x from -3 to -1 across its width (perspective: texcoords are not affine in window space), texcoords 0..3 / 0..2
    pos = np.array([(-3.0, -2.0, -3.0), (-1.0, -2.0, 3.5), (-1.0, 4.0, 3.5), (-3.0, 4.0, -3.0)], np.float32)
    uv = np.array([(0.0, 0.0), (3.0, 0.0), (3.0, 2.0), (0.0, 2.0)], np.float32)
    nrm, tan, col = attrs(4)
    col |= np.uint32(0x60000000)  # keep vertex alpha away from zero: the texture's alpha decides
    m.add_primitive(pos, nrm, (0, 1, 2, 0, 2, 3), wall, ptype=_abi.PRIMITIVE_TYPE_CUTOUT, colors=col, tangents=tan, texcoords=uv)
    tri = np.array([(-4.5, 0.0, -1.0), (-4.5, 0.0, 1.0), (-4.5, 2.0, 0.0)], np.float32)
    nrm, tan, col = attrs(3)
    uv3 = np.array([(0.1, 0.2), (0.9, 0.3), (0.5, 0.8)], np.float32)
    for mat, order in ((red, (0, 1, 2)), (blue, (0, 1, 2)), (red, (0, 2, 1)), (blue, (0, 2, 1))):
        m.add_primitive(tri, nrm, order, mat, colors=col, tangents=tan, texcoords=uv3)
    return m, view


def golden_rt_scene():
    """The scene behind tests/golden/rt_64x36.npz (tools/gen_golden.py): golden_raster_scene() with a floor under it and a SOLID slab
    above the floor (so that rays leave from, and end on, opaque and alpha-tested geometry), a sun from above-left with three shadow
    samples, and a seeded 128 x 128 stand-in for the blue-noise layer.  Returns (mesh, view, sun, noise)."""
    from androidrenderer_amd import mesh, scene
    m, view = golden_raster_scene()
    grey = m.add_material(mesh.material(base=(0.5, 0.5, 0.5, 1.0)))
    m.add_primitive([(-5, -2, -6), (4, -2, -6), (4, -2, 6), (-5, -2, 6)], [(0, 1, 0)] * 4, (0, 2, 1, 0, 3, 2), grey)
    m.add_primitive([(-5, 0.5, -6), (0, 0.5, -6), (0, 0.5, -1), (-5, 0.5, -1)], [(0, -1, 0)] * 4, (0, 1, 2, 0, 2, 3, 0, 2, 1, 0, 3, 2), grey)
    sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_RT, num_shadow_samples=3.0)
    sun.set_direction([0.5, -1.0, 0.3])
    noise = synth.rng(121).integers(0, 256, (128, 128, 4), dtype=np.uint8)
    return m, view, sun, noise


def rt_gi_cascades(centre, spacing0):
    """[(min xyz, spacing)] of the four irradiance-cache cascades around `centre` as the ray-tracing tests place them"""
    out = []
    for c in range(4):
        spacing = spacing0 * (2.0 ** c)
        ext = (32 * spacing, 8 * spacing, 32 * spacing)
        out.append(([centre[i] - ext[i] / 2.0 + 0.013 * (c + 1) for i in range(3)], spacing))
    return out


def golden_rt_gi_inputs():
    """inputs of the GI rays of tests/golden/rt_gi_64x36.npz besides golden_rt_scene(): sky LUTs, the irradiance cache the probe misses sample,
    its cascades and the probes traced (all four cascades, inside and outside the geometry, one outside the cascades)"""
    luts, at = synth.sky_luts(203), synth.probe_atlases(603)
    probes = np.array([(16, 3, 16), (14, 2, 17), (10, 4, 12), (18, 5, 14), (15, 11, 16), (17, 12, 15), (16, 19, 16), (15, 20, 17), (16, 27, 16), (17, 28, 15),
                       (3, 1, 30), (5, 40, 7)], np.uint32)
    return {"sky_t": luts["transmittance"], "sky_v": luts["sky_view"], "irr": at["irradiance"], "pdepth": at["depth"], "val": at["validity"],
            "cascades": rt_gi_cascades((0.0, 1.0, 0.0), 0.5), "probes": probes}


def golden_raster_sun(view):
    """Sun whose cascades are fitted to `view` (directional_light.cpp:164-260 through scene.DirectionalLight)."""
    from androidrenderer_amd import scene
    sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_CSM)
    sun.update_shadow_cascades(view, resolution=48)
    return sun


def golden_raster_lpv(view, sun):
    from androidrenderer_amd import scene
    lpv = scene.LpvCascades()
    lpv.update_cascade_transforms(view, sun)
    return lpv
