"""ctypes mirror of include/sah_hip.h (the C ABI). Field order and sizes must match the header exactly;
tests/test_abi.py checks the struct sizes against the values the library reports."""
import ctypes as C

SAH_OK = 0
SAH_ERR_INVALID_ARGUMENT = -1
SAH_ERR_UNSUPPORTED_FORMAT = -2
SAH_ERR_HIP = -3
SAH_ERR_NO_DEVICE = -4
SAH_ERR_COMM = -5
SAH_ERR_UNSUPPORTED = -6

FORMAT_R8_UNORM = 9
FORMAT_R8G8B8A8_UNORM = 37
FORMAT_R8G8B8A8_SRGB = 43
FORMAT_R16_SFLOAT = 76
FORMAT_R16G16_SFLOAT = 83
FORMAT_R16G16B16A16_SFLOAT = 97
FORMAT_R32_SFLOAT = 100
FORMAT_B10G11R11_UFLOAT_PACK32 = 122
FORMAT_D16_UNORM = 124
FORMAT_D32_SFLOAT = 126

FORMAT_BPP = {9: 1, 37: 4, 43: 4, 76: 2, 83: 4, 97: 8, 100: 4, 122: 4, 124: 2, 126: 4}

SHADOW_MODE_OFF, SHADOW_MODE_CSM, SHADOW_MODE_RT = 0, 1, 2
GI_NONE, GI_LPV, GI_CACHE, GI_RTGI = 0, 1, 2, 3
LIGHTING_QUIRK_SUN_BLEND = 1 << 0
LIGHTING_BRUTE_FORCE_LIGHTS = 1 << 1
LIGHTING_DEFAULT_FLAGS = LIGHTING_QUIRK_SUN_BLEND
MAX_BLOOM_MIPS = 8
TONEMAP_TOLERANCE_1CODE = 1 << 0
GENERATION_TRACKED = 0xFFFFFFFF  # sah_gi::lpv_generation / probe_generation: the context keeps its gather copies current itself


class Plane(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("row_pitch_bytes", C.c_uint32),
                ("format", C.c_uint32)]


class Volume(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("depth", C.c_uint32),
                ("row_pitch_bytes", C.c_uint32), ("slice_pitch_bytes", C.c_uint32), ("format", C.c_uint32)]


class ProbeAtlases(C.Structure):  # sah_probe_atlases
    _fields_ = [("rtgi", Volume), ("light_cache", Volume), ("depth", Volume), ("average", Volume), ("validity", Volume)]


PRIMITIVE_TYPE_SOLID, PRIMITIVE_TYPE_CUTOUT = 0, 1
RASTER_STATS_WORDS = 8
RT_STATS_WORDS = 4
IPC_HANDLE_BYTES = 128


class VertexData(C.Structure):  # sah_vertex_data, 40 bytes
    _fields_ = [("normal", C.c_float * 3), ("tangent", C.c_float * 4), ("texcoord", C.c_float * 2), ("color", C.c_uint32)]


class Material(C.Structure):  # sah_material, 112 bytes
    _fields_ = [("base_color_tint", C.c_float * 4), ("emission_factor", C.c_float * 4), ("metalness_factor", C.c_float),
                ("roughness_factor", C.c_float), ("opacity_threshold", C.c_float), ("padding1", C.c_float),
                ("base_color_texel", C.c_float * 4), ("normal_texel", C.c_float * 4), ("data_texel", C.c_float * 4),
                ("emission_texel", C.c_float * 4)]


class Primitive(C.Structure):  # sah_primitive, 96 bytes
    _fields_ = [("model", C.c_float * 16), ("first_index", C.c_uint32), ("index_count", C.c_uint32), ("vertex_offset", C.c_int32),
                ("type", C.c_uint32), ("material", C.c_uint32), ("padding", C.c_uint32 * 3)]


FILTER_NEAREST, FILTER_LINEAR = 0, 1
ADDRESS_REPEAT, ADDRESS_MIRRORED_REPEAT, ADDRESS_CLAMP_TO_EDGE = 0, 1, 2
MAX_TEXTURE_MIPS = 14
TEXTURE_NONE = 0xFFFFFFFF


class Sampler(C.Structure):  # sah_sampler, 40 bytes
    _fields_ = [("mag_filter", C.c_uint32), ("min_filter", C.c_uint32), ("mipmap_mode", C.c_uint32), ("address_u", C.c_uint32),
                ("address_v", C.c_uint32), ("mip_lod_bias", C.c_float), ("min_lod", C.c_float), ("max_lod", C.c_float),
                ("max_anisotropy", C.c_float), ("reserved", C.c_uint32)]


class Texture(C.Structure):  # sah_texture, 384 bytes
    _fields_ = [("mips", Plane * MAX_TEXTURE_MIPS), ("num_mips", C.c_uint32), ("padding", C.c_uint32), ("sampler", Sampler)]


class MaterialTextures(C.Structure):  # sah_material_textures, 16 bytes
    _fields_ = [("base_color", C.c_uint32), ("normal", C.c_uint32), ("data", C.c_uint32), ("emission", C.c_uint32)]


class SceneGeometry(C.Structure):  # sah_scene_geometry
    _fields_ = [("vertex_positions", C.c_void_p), ("vertex_data", C.c_void_p), ("indices", C.c_void_p), ("primitives", C.c_void_p),
                ("materials", C.c_void_p), ("num_vertices", C.c_uint32), ("num_indices", C.c_uint32), ("num_primitives", C.c_uint32),
                ("num_materials", C.c_uint32), ("textures", C.c_void_p), ("material_textures", C.c_void_p), ("num_textures", C.c_uint32),
                ("padding", C.c_uint32)]


class RsmTargets(C.Structure):  # sah_rsm_targets
    _fields_ = [("flux", Volume), ("normals", Volume), ("depth", Volume)]


class GBuffer(C.Structure):
    _fields_ = [("color", Plane), ("normals", Plane), ("data", Plane), ("emission", Plane), ("depth", Plane)]


class MipChain(C.Structure):
    _fields_ = [("mips", Plane * MAX_BLOOM_MIPS), ("num_mips", C.c_uint32)]


class ViewData(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("projection", C.c_float * 16), ("inverse_view", C.c_float * 16),
                ("inverse_projection", C.c_float * 16), ("last_frame_view", C.c_float * 16),
                ("last_frame_projection", C.c_float * 16), ("frustum", C.c_float * 4), ("z_near", C.c_float),
                ("material_texture_mip_bias", C.c_float), ("render_resolution", C.c_float * 2), ("jitter", C.c_float * 2),
                ("previous_jitter", C.c_float * 2)]


class SunLightConstants(C.Structure):
    _fields_ = [("direction_and_tan_size", C.c_float * 4), ("color", C.c_float * 4), ("csm_resolution", C.c_uint32 * 4),
                ("data", (C.c_float * 4) * 4), ("cascade_matrices", (C.c_float * 16) * 4),
                ("cascade_inverse_matrices", (C.c_float * 16) * 4), ("shadow_mode", C.c_uint32),
                ("num_shadow_samples", C.c_float), ("padding1", C.c_uint32), ("padding2", C.c_uint32)]


class LpvCascadeMatrices(C.Structure):
    _fields_ = [("rsm_vp", C.c_float * 16), ("inverse_rsm_vp", C.c_float * 16), ("world_to_cascade", C.c_float * 16),
                ("cascade_to_world", C.c_float * 16)]


class ProbeCascade(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("probe_spacing", C.c_float)]


class GI(C.Structure):
    _fields_ = [("kind", C.c_uint32),
                ("lpv_red", Volume), ("lpv_green", Volume), ("lpv_blue", Volume),
                ("lpv_cascades", C.POINTER(LpvCascadeMatrices)), ("lpv_num_cascades", C.c_uint32), ("lpv_exposure", C.c_float),
                ("probe_irradiance", Volume), ("probe_depth", Volume), ("probe_validity", Volume),
                ("probe_cascades", ProbeCascade * 4), ("probe_size", C.c_uint32 * 2), ("cache_debug_mode", C.c_uint32),
                ("ray_buffer", Plane), ("ray_irradiance", Plane), ("noise", Plane), ("num_extra_rays", C.c_uint32),
                ("extra_ray_radius", C.c_float), ("lpv_generation", C.c_uint32), ("probe_generation", C.c_uint32)]


class SkyLuts(C.Structure):
    _fields_ = [("transmittance", Plane), ("sky_view", Plane)]


class PointLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("radius", C.c_float), ("color", C.c_float * 3), ("intensity", C.c_float)]


class LightList(C.Structure):
    _fields_ = [("lights", C.c_void_p), ("count", C.c_uint32)]


class LightingDesc(C.Structure):
    _fields_ = [("gbuffer", C.POINTER(GBuffer)), ("ao", C.POINTER(Plane)), ("lit", C.POINTER(Plane)),
                ("view", C.POINTER(ViewData)), ("sun", C.POINTER(SunLightConstants)), ("shadowmap", C.POINTER(Volume)),
                ("shadow_mask", C.POINTER(Plane)), ("lights", C.POINTER(LightList)), ("gi", C.POINTER(GI)),
                ("sky", C.POINTER(SkyLuts)), ("flags", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32)]


class ChainPlan(C.Structure):  # sah_chain_plan
    _fields_ = [("aa_rows", C.c_uint32 * 2), ("mip0_rows", C.c_uint32 * 2), ("mip1_rows", C.c_uint32 * 2), ("out_rows", C.c_uint32 * 2),
                ("mip1_rows_per_rank", C.c_uint32), ("mip1_allocated_rows", C.c_uint32), ("rows_per_rank", C.c_uint32),
                ("out_allocated_rows", C.c_uint32)]


class ChainFrame(C.Structure):  # sah_chain_frame
    _fields_ = [("lighting", C.POINTER(LightingDesc) * 2), ("lit", Plane), ("antialiased", Plane), ("bloom", MipChain), ("out", Plane)]


CHAIN_NO_EXCHANGE = 1 << 0
CHAIN_CAPTURE = 1 << 1

assert C.sizeof(ViewData) == 432
assert C.sizeof(SunLightConstants) == 640
assert C.sizeof(LpvCascadeMatrices) == 256
assert C.sizeof(ProbeCascade) == 16
assert C.sizeof(PointLight) == 32


class ProbeTraceDesc(C.Structure):  # sah_probe_trace_desc
    _fields_ = [("cascades", ProbeCascade * 4), ("probes_to_update", C.c_void_p), ("num_probes", C.c_uint32),
                ("sun", C.POINTER(SunLightConstants)), ("sky", C.POINTER(SkyLuts)), ("noise", C.POINTER(Plane)),
                ("probe_irradiance", Volume), ("probe_depth", Volume), ("probe_validity", Volume), ("probe_size", C.c_uint32 * 2),
                ("trace_results", Volume)]
