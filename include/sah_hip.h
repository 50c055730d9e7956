/* sah_hip.h — C ABI of the MI355X-native deferred lighting + GI + post hot path.
 *
 * This is the drop-in boundary.  The reference renderer has no FFI of its
 * own; each entry point below replaces the GPU work recorded by one of the reference's C++ seams, cited
 * as `RenderCore/...:line` (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns int: 0 = SAH_OK, < 0 = sah_status; nothing throws across the ABI;
 *   - one sah_ctx per device; calls on one ctx are serialised by the caller (the reference records all
 *     passes from one thread, RenderCore/render/scene_renderer.cpp:121-470);
 *   - all work is enqueued on the ctx's HIP stream and is asynchronous; sah_sync() waits for it;
 *   - the caller owns every buffer; image pointers are DEVICE pointers to linear row-major storage in
 *     the VkFormat the reference allocates for that resource (format values are VkFormat enumerants);
 *   - uniform blocks (sah_view_data, sah_sun_light_constants, cascade tables) are HOST pointers, read
 *     during the call and passed to the kernels by value, like the reference's UBO uploads;
 *   - matrices are column-major float[16] (glm / GLSL layout, m[col*4 + row]).
 */
#ifndef SAH_HIP_H
#define SAH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAH_ABI_VERSION 6

/* sah_gi::lpv_generation / probe_generation: the library keeps its gather copies current itself (see there) */
#define SAH_GENERATION_TRACKED 0xffffffffu

typedef enum sah_status {
    SAH_OK = 0,
    SAH_ERR_INVALID_ARGUMENT = -1,
    SAH_ERR_UNSUPPORTED_FORMAT = -2,
    SAH_ERR_HIP = -3,
    SAH_ERR_NO_DEVICE = -4,
    SAH_ERR_COMM = -5,
    SAH_ERR_UNSUPPORTED = -6
} sah_status;

/* VkFormat values used on the path (Vulkan 1.4 core enumerants). */
enum {
    SAH_FORMAT_R8_UNORM = 9,
    SAH_FORMAT_R8G8B8A8_UNORM = 37,
    SAH_FORMAT_R8G8B8A8_SRGB = 43,
    SAH_FORMAT_R16_SFLOAT = 76,
    SAH_FORMAT_R16G16_SFLOAT = 83,
    SAH_FORMAT_R16G16B16A16_SFLOAT = 97,
    SAH_FORMAT_R32_SFLOAT = 100,
    SAH_FORMAT_B10G11R11_UFLOAT_PACK32 = 122,
    SAH_FORMAT_D16_UNORM = 124,
    SAH_FORMAT_D32_SFLOAT = 126
};

/* ---- resources ------------------------------------------------------------------------------ */

/* 2D image, linear row-major.  Stands in for the reference's TextureHandle
 * (RenderCore/render/backend/handles.hpp:3-5). */
typedef struct sah_plane {
    void* ptr;
    uint32_t width, height;
    uint32_t row_pitch_bytes;
    uint32_t format;
} sah_plane;

/* 3D image or 2D array (depth = slices / layers), linear. */
typedef struct sah_volume {
    void* ptr;
    uint32_t width, height, depth;
    uint32_t row_pitch_bytes, slice_pitch_bytes;
    uint32_t format;
} sah_volume;

/* RenderCore/render/gbuffer.hpp:5-11; formats from RenderCore/render/scene_renderer.cpp:580-649. */
typedef struct sah_gbuffer {
    sah_plane color;    /* R8G8B8A8_SRGB   */
    sah_plane normals;  /* R16G16B16A16_SFLOAT, xyz = world normal (not normalised) */
    sah_plane data;     /* R8G8B8A8_UNORM, g = roughness, b = metalness */
    sah_plane emission; /* R8G8B8A8_SRGB   */
    sah_plane depth;    /* D32_SFLOAT, reversed-Z infinite far, 0 = sky */
} sah_gbuffer;

/* Bloom pyramid (RenderCore/render/bloomer.cpp:268-285): mip 0 = output/2, RGBA16F, tightly packed. */
#define SAH_MAX_BLOOM_MIPS 8
typedef struct sah_mipchain {
    sah_plane mips[SAH_MAX_BLOOM_MIPS];
    uint32_t num_mips;
} sah_mipchain;

/* ---- uniform blocks (byte-identical to RenderCore/shared/) ----------------------------------- */

/* RenderCore/shared/view_data.hpp:6-40 — 432 bytes. */
typedef struct sah_view_data {
    float view[16];
    float projection[16];
    float inverse_view[16];
    float inverse_projection[16];
    float last_frame_view[16];
    float last_frame_projection[16];
    float frustum[4];
    float z_near;
    float material_texture_mip_bias;
    float render_resolution[2];
    float jitter[2];
    float previous_jitter[2];
} sah_view_data;

#define SAH_SHADOW_MODE_OFF 0
#define SAH_SHADOW_MODE_CSM 1
#define SAH_SHADOW_MODE_RT 2

/* RenderCore/shared/sun_light_constants.hpp:10-44 — 640 bytes. */
typedef struct sah_sun_light_constants {
    float direction_and_tan_size[4];
    float color[4];
    uint32_t csm_resolution[4];
    float data[4][4]; /* split depth in .x */
    float cascade_matrices[4][16];
    float cascade_inverse_matrices[4][16];
    uint32_t shadow_mode;
    float num_shadow_samples;
    uint32_t padding1, padding2;
} sah_sun_light_constants;

/* RenderCore/shared/lpv.hpp:6-11 — 256 bytes. */
typedef struct sah_lpv_cascade_matrices {
    float rsm_vp[16];
    float inverse_rsm_vp[16];
    float world_to_cascade[16];
    float cascade_to_world[16];
} sah_lpv_cascade_matrices;

/* RenderCore/shared/gi_probe.hpp:5-8 — 16 bytes. */
typedef struct sah_probe_cascade {
    float min[3];
    float probe_spacing;
} sah_probe_cascade;

/* ---- lighting pass --------------------------------------------------------------------------- */

typedef enum sah_gi_kind { SAH_GI_NONE = 0, SAH_GI_LPV = 1, SAH_GI_CACHE = 2, SAH_GI_RTGI = 3 } sah_gi_kind;

/* What IGlobalIlluminator::render_to_lit_scene binds (RenderCore/render/gi/global_illuminator.hpp:38-40):
 *   LPV   RenderCore/render/gi/light_propagation_volume.cpp:274-306
 *   CACHE RenderCore/render/gi/irradiance_cache.cpp:287-306
 *   RTGI  RenderCore/render/gi/rtgi.cpp:160-188 */
typedef struct sah_gi {
    uint32_t kind; /* sah_gi_kind */

    /* LPV: three RGBA16F 3D volumes, (32*num_cascades) x 32 x 32 */
    sah_volume lpv_red, lpv_green, lpv_blue;
    const sah_lpv_cascade_matrices* lpv_cascades; /* host, lpv_num_cascades entries (<= 4) */
    uint32_t lpv_num_cascades;
    float lpv_exposure; /* r.GI.LPV.Exposure, default pi*10 */

    /* Irradiance cache: 2D arrays with 32 layers */
    sah_volume probe_irradiance; /* B10G11R11_UFLOAT 224x256x32 */
    sah_volume probe_depth;      /* R16G16_SFLOAT   384x384x32 */
    sah_volume probe_validity;   /* R8_UNORM        32x32x32   */
    sah_probe_cascade probe_cascades[4];
    uint32_t probe_size[2]; /* push constants, (5,6) in the reference */
    uint32_t cache_debug_mode;

    /* RTGI reconstruction */
    sah_plane ray_buffer;     /* RGBA16F: ray direction xyz, distance */
    sah_plane ray_irradiance; /* RGBA16F */
    sah_plane noise;          /* RGBA8_UNORM 128x128 blue-noise layer */
    uint32_t num_extra_rays;  /* r.GI.Reconstruction.NumSamples */
    float extra_ray_radius;   /* r.GI.Reconstruction.Size */

    /* LPV: change counter of the three volumes, kept by whoever writes them (the reference re-propagates once per frame:
     * light_propagation_volume.cpp:970-1063).  The fast Lighting kernel gathers from an interleaved copy of the volumes that it otherwise
     * rebuilds on every call (5 us + a launch); with a non-zero counter the copy is rebuilt only when the counter or one of the three
     * volume descriptors differs from the previous sah_lighting call of this context.  0 = rebuild every call.  sah_lpv_clear,
     * sah_lpv_propagate and sah_lpv_inject_vpls on this context drop the copy regardless.
     * SAH_GENERATION_TRACKED: "these volumes are written through this context only".  The LAST step of sah_lpv_propagate then writes the
     * interleaved copy beside the volumes it stores anyway (same texels, no extra pass), and the Lighting pass that follows gathers from
     * it without a rebuild: a frame that propagates and then shades — the reference's order — never runs the 5 us copy pass. */
    uint32_t lpv_generation;
    /* Irradiance cache: the same for the irradiance atlas (written by sah_probe_update / sah_probe_copy, or by the caller).  The tiled
     * Lighting kernel gathers the bilinear taps from an fp32 copy of the R11G11B10 atlas — the widening is exact, so the arithmetic is
     * the same — that it rebuilds on every call unless this counter is non-zero and, like the atlas descriptor, unchanged since the
     * previous sah_lighting call of this context.  0 = rebuild every call.  sah_probe_update and sah_probe_copy on this context drop
     * the copy regardless — except under SAH_GENERATION_TRACKED ("this atlas is written through this context only"): sah_probe_update then
     * re-widens just the blocks of the probes it updated (<= 1024 of 32768 per frame in the reference: irradiance_cache.cpp:21-23), and
     * only sah_probe_copy into the atlas, which rewrites all of it, drops the copy. */
    uint32_t probe_generation;
} sah_gi;

/* Sky LUTs sampled by the sky fill (RenderCore/render/procedural_sky.cpp:13-42,151-172). */
typedef struct sah_sky_luts {
    sah_plane transmittance; /* RGBA16F 256x64  */
    sah_plane sky_view;      /* RGBA16F 200x200 */
} sah_sky_luts;

/* Extension (not in the reference): point lights, BASELINE configs 2/3/5.  Spec in DESIGN.md §5b. */
typedef struct sah_point_light {
    float position[3];
    float radius;
    float color[3];
    float intensity;
} sah_point_light;

typedef struct sah_light_list {
    const sah_point_light* lights; /* device pointer */
    uint32_t count;
} sah_light_list;

/* flags */
#define SAH_LIGHTING_QUIRK_SUN_BLEND (1u << 0) /* reproduce the SRC_COLOR/DST_COLOR sun blend (squares the sun term) */
#define SAH_LIGHTING_BRUTE_FORCE_LIGHTS (1u << 1) /* shade every light for every pixel (no tile culling) */
#define SAH_LIGHTING_DEFAULT_FLAGS (SAH_LIGHTING_QUIRK_SUN_BLEND)

/* One "Lighting" pass: RenderCore/render/phase/lighting_phase.cpp:34-134
 * (clear, sun [CSM], GI overlay, emissive, sky, then sun [RT mode] as a read-modify-write). */
typedef struct sah_lighting_desc {
    const sah_gbuffer* gbuffer;
    const sah_plane* ao;  /* R32_SFLOAT, consumed only by the LPV overlay; may be NULL otherwise */
    const sah_plane* lit; /* out: R16G16B16A16_SFLOAT */
    const sah_view_data* view;
    const sah_sun_light_constants* sun; /* sun->shadow_mode selects off / CSM / RT */
    const sah_volume* shadowmap;        /* CSM mode: D16_UNORM 2D array, one layer per cascade */
    const sah_plane* shadow_mask;       /* RT mode: per-pixel visibility fraction (R32_SFLOAT), NULL = 1 */
    const sah_light_list* lights;       /* may be NULL */
    const sah_gi* gi;                   /* may be NULL (= no GI) */
    const sah_sky_luts* sky;            /* NULL = no sky fill */
    uint32_t flags;
    uint32_t row_begin, row_end; /* rows [row_begin, row_end) are shaded; 0,0 = whole image */
} sah_lighting_desc;

/* ---- API ------------------------------------------------------------------------------------- */

typedef struct sah_ctx sah_ctx;

int sah_abi_version(void);
const char* sah_status_string(int status);
const char* sah_last_error(const sah_ctx* ctx);

/* comm_id: the 128-byte ncclUniqueId from sah_comm_unique_id() (the same bytes on every rank), or NULL for a context without an RCCL
 * communicator (world > 1 then needs the direct exchange below: sah_ipc_open / sah_ipc_connect).  world == 1 with a comm_id builds a one-rank communicator, so that the exchange entry points
 * below run through RCCL exactly as they do on N ranks. */
int sah_create(sah_ctx** out, int device, int rank, int world, const void* comm_id);
void sah_destroy(sah_ctx* ctx);
int sah_comm_unique_id(void* out_128_bytes);
/* Use an externally owned hipStream_t (e.g. the caller's frame stream) for all subsequent work.  Streams are not ordered against each
 * other as a whole; what the library keeps between calls (gather copies, tables, scratch buffers, the ray-tracing structure) is: a pass
 * that is moved to another stream starts behind its own last use of that state on the previous stream. */
int sah_set_stream(sah_ctx* ctx, void* hip_stream);
int sah_sync(sah_ctx* ctx);

/* LightingPhase::render — RenderCore/render/phase/lighting_phase.hpp:33-43 */
int sah_lighting(sah_ctx* ctx, const sah_lighting_desc* desc);

/* "Copy scene" (AA = None) — RenderCore/render/scene_renderer.cpp:502-527 */
int sah_copy_scene(sah_ctx* ctx, const sah_plane* lit, const sah_plane* antialiased);
/* Rows [row_begin, row_end) of the same copy (0, 0 = all); output row j reads lit rows j - 1 .. j + 1. */
int sah_copy_scene_rows(sah_ctx* ctx, const sah_plane* lit, const sah_plane* antialiased, uint32_t row_begin, uint32_t row_end);
/* "Copy scene" and the first dispatch of Bloomer::fill_bloom_tex (scene_renderer.cpp:502-527 then bloomer.cpp:50-72) as ONE pass over lit:
 * rows [aa_row_begin, aa_row_end) of `antialiased` and rows [mip_row_begin, mip_row_end) of bloom->mips[0], each exactly what
 * sah_copy_scene_rows followed by sah_bloom_mip0_rows writes ((0, 0) = all rows).  The mip rows' sources — antialiased rows
 * 2 * mip_row_begin - 3 .. 2 * mip_row_end + 2 — need NOT be among the rows asked for: they are computed from lit either way. */
int sah_copy_scene_bloom_mip0_rows(sah_ctx* ctx, const sah_plane* lit, const sah_plane* antialiased, const sah_mipchain* bloom, uint32_t aa_row_begin,
                                   uint32_t aa_row_end, uint32_t mip_row_begin, uint32_t mip_row_end);

/* Bloomer::fill_bloom_tex — RenderCore/render/bloomer.hpp:15, bloomer.cpp:38-262 */
int sah_bloom(sah_ctx* ctx, const sah_plane* scene_color, const sah_mipchain* bloom);
/* The same pyramid in two steps, for frames sharded by rows (the pyramid is the one pass that is not row-local): rows
 * [row_begin, row_end) of mip 0 from the scene — the source rows they read must be valid: sah_bloom_source_rows below — and, once every
 * rank's mip 0 rows have been exchanged (sah_allgather_rows on mips[0]), mips 1.. from mip 0.  Together they write what sah_bloom writes.
 * An EMPTY range (row_begin == row_end, (0, 0) included) writes nothing: unlike the other *_rows entry points these two have no
 * "(0, 0) = all rows" convention — a rank of a sharded frame may own no row of a small mip, and must be able to say so. */
int sah_bloom_mip0_rows(sah_ctx* ctx, const sah_plane* scene_color, const sah_mipchain* bloom, uint32_t row_begin, uint32_t row_end);
int sah_bloom_from_mip0(sah_ctx* ctx, const sah_plane* scene_color, const sah_mipchain* bloom);
/* The same split one level down, or at any level: rows [row_begin, row_end) of mip `mip` from its source (the scene for mip 0, mip - 1
 * otherwise; the source rows they read must be valid: sah_bloom_source_rows), and mips mip + 1 .. from mip `mip`.  A frame sharded by rows
 * that exchanges mip 1 instead of mip 0 moves a quarter of the bytes: every rank then computes the mip 0 rows its own mip 1 rows and its own
 * rows of the composite read (androidrenderer_amd/shard.py) — sah_bloom_mip_rows(.., 0, ..) or the fused copy above —,
 * sah_bloom_mip_rows(.., 1, ..), the exchange of mips[1], sah_bloom_from_mip(.., 1). */
int sah_bloom_mip_rows(sah_ctx* ctx, const sah_plane* scene_color, const sah_mipchain* bloom, uint32_t mip, uint32_t row_begin, uint32_t row_end);
int sah_bloom_from_mip(sah_ctx* ctx, const sah_plane* scene_color, const sah_mipchain* bloom, uint32_t mip);
/* Which source rows the downsample reads for destination rows [row_begin, row_end) (bloom_downsample.comp:16-52: 20 bilinear taps within two
 * source texels of c = (j + 0.5) * src_height / dst_height - 0.5).  That is 2j - 2 .. 2j + 3 ONLY when the source is exactly twice as high;
 * otherwise — a 75-row mip under a 37-row one — the window drifts by up to a row over the mip's height, and one more row either side is
 * allowed for the shader's fp32 coordinate.  out[0], out[1] = [first, one past last) source row, clipped to [0, src_height).  A caller that
 * shards an odd-height frame by the "2j - 2 .. 2j + 3" rule feeds rows nobody validated; this is the rule androidrenderer_amd/shard.py uses. */
int sah_bloom_source_rows(uint32_t src_height, uint32_t dst_height, uint32_t row_begin, uint32_t row_end, uint32_t out[2]);

/* UiPhase::draw_scene_image — RenderCore/render/phase/ui_phase.cpp:98-113; out = R8G8B8A8 (sRGB-encoded).
 * Rows [row_begin,row_end) of the output are written (0,0 = all). */
int sah_tonemap(sah_ctx* ctx, const sah_plane* scene_color, const sah_mipchain* bloom, const sah_plane* out_rgba8,
                uint32_t row_begin, uint32_t row_end);
/* The same pass with flags.  0 = sah_tonemap: strict, every operator of the shader individually rounded in the shader's order, codes
 * bit-identical to the oracle.  SAH_TONEMAP_TOLERANCE_1CODE: the same real-number expression evaluated in another order (the nine tent
 * taps of a mip as one column pass per tile and four row interpolations per pixel, fused multiply-adds) — every channel of every pixel
 * within ONE R8G8B8A8 code of the strict result (BASELINE.json north_star: 1 ULP of the stored format), identical on all but the few
 * pixels whose value lies within ~2^-20 relative of a code threshold; about half the time of the strict pass.  Chains of
 * more than six mips (SAH_MAX_BLOOM_MIPS allows eight; the reference makes six) take the strict kernel under this flag as well. */
#define SAH_TONEMAP_TOLERANCE_1CODE (1u << 0)
int sah_tonemap_ex(sah_ctx* ctx, const sah_plane* scene_color, const sah_mipchain* bloom, const sah_plane* out_rgba8,
                   uint32_t row_begin, uint32_t row_end, uint32_t flags);

/* LightPropagationVolume::clear_volume / propagate_lighting —
 * RenderCore/render/gi/light_propagation_volume.cpp:839-926, 970-1063.
 * a_* hold the injected light on entry and the propagated light on return when steps is even
 * (the reference ping-pongs A->B->A...; with 32 steps the result is in A). */
int sah_lpv_clear(sah_ctx* ctx, const sah_volume* red, const sah_volume* green, const sah_volume* blue,
                  const sah_volume* geometry, uint32_t num_cascades);
int sah_lpv_propagate(sah_ctx* ctx, const sah_volume a_rgb[3], const sah_volume b_rgb[3], uint32_t num_cascades,
                      uint32_t steps);

/* ProceduralSky::update_sky_luts — RenderCore/render/procedural_sky.cpp:75-149: transmittance (256x64), multiple-scattering (32x32)
 * and sky-view (200x200) LUTs, all R16G16B16A16_SFLOAT, generated in that order (shaders/sky/{transmittance,multiscattering,
 * sky_view}_lut.comp); light_vector is the sky-view push constant (the direction the sun light travels).  The transmittance and
 * sky-view LUTs are the `sky` inputs of sah_lighting. */
int sah_sky_update_luts(sah_ctx* ctx, const sah_plane* transmittance, const sah_plane* multiscattering, const sah_plane* sky_view,
                        const float light_vector[3]);

/* AmbientOcclusionPhase::generate_ao with r.AO.Mode = Off — RenderCore/render/phase/ambient_occlusion_phase.cpp:167-179: the AO
 * target (R32_SFLOAT) is cleared to 1.0.  RTAO / CACAO need the scene BVH and stay in the renderer; their output is the `ao` input
 * plane of sah_lighting. */
int sah_ao_clear(sah_ctx* ctx, const sah_plane* ao);

/* Irradiance-cache probe maintenance (a11) — RenderCore/render/gi/irradiance_cache.cpp:455-486 (copy_probes_to_new_texture)
 * and :585-724 (dispatch_probe_updates, minus the ray-tracing pass whose output `trace_results` is an input here).
 * Atlases are 2D arrays of 32 layers with one block per probe (irradiance_cache.cpp:94-183); probe grid 32 x (8 * 4 cascades) x 32;
 * block sizes are fixed by the shaders: rtgi 7x8, light cache 13x13, depth 12x12 texels (interior 5x6, 11x11, 10x10).
 *
 * The reference shaders race in two places (several invocations of one dispatch store different values to the same texel:
 * write_probe_texel_with_border for odd probe widths, gi/cache/probe_update.slangi:4-37, and init_new_probe's depth clear at
 * light-cache offsets, gi/cache/copy_cascades.comp.slang:39-45) and use WaveActiveSum on half3 (probe_finalize.comp.slang:66),
 * whose summation order the API leaves open.  This ABI fixes an order: stores of one dispatch take effect in
 * ascending linear invocation index (program order inside an invocation), the wave sum adds in lane order in fp16; out-of-range
 * image stores are dropped, out-of-range loads return 0. */
typedef struct sah_probe_atlases {
    sah_volume rtgi;        /* B10G11R11_UFLOAT_PACK32, (32*7)  x (32*8)  x 32 */
    sah_volume light_cache; /* B10G11R11_UFLOAT_PACK32, (32*13) x (32*13) x 32 */
    sah_volume depth;       /* R16G16_SFLOAT,           (32*12) x (32*12) x 32 */
    sah_volume average;     /* B10G11R11_UFLOAT_PACK32, 32 x 32 x 32 */
    sah_volume validity;    /* R8_UNORM,                32 x 32 x 32 */
} sah_probe_atlases;

/* copy_cascades.comp.slang:86-99: scroll every cascade by cascade_movement[c] (probe cells, truncated toward zero), copying probe
 * blocks src -> dst and initialising the probes that scroll in (validity 1.0 marks them new).  src and dst must not alias. */
int sah_probe_copy(sah_ctx* ctx, const sah_probe_atlases* src, const sah_probe_atlases* dst, const float cascade_movement[4][3]);

/* probe_depth_update, probe_light_cache_update, probe_rtgi_update, probe_finalize, in that order, for `num_probes` probes.
 * trace_results: R16G16B16A16_SFLOAT 20 x 20 x num_probes (rgb = radiance, a = hit distance, <= 0: miss);
 * probes_to_update: DEVICE pointer to num_probes tightly packed uint32 triples (probe x, y, layer), all distinct. */
int sah_probe_update(sah_ctx* ctx, const sah_probe_atlases* atlases, const sah_volume* trace_results, const uint32_t* probes_to_update,
                     uint32_t num_probes);

/* For a caller that updates probes of the irradiance atlas by its own means (the reference's own update shaders, say:
 * IrradianceCache::dispatch_probe_updates, irradiance_cache.cpp:644-723) while the context tracks its fp32 copy of that atlas
 * (sah_gi::probe_generation = SAH_GENERATION_TRACKED): "the blocks of these probes have changed" — the context widens those blocks again
 * (what sah_probe_update does by itself), on its stream, behind the caller's writes.  A no-op when the context holds no tracked copy of
 * `probe_irradiance`.  probes: DEVICE pointer to num_probes uint32 triples (probe x, y, layer), as for sah_probe_update. */
int sah_probe_notify_updated(sah_ctx* ctx, const sah_volume* probe_irradiance, const uint32_t* probes, uint32_t num_probes);

/* ---- producers either side of the lighting pass ----------------------------------------------------------
 * The two rasterisation passes whose outputs a1 gathers from: the sun shadow cascades (depth only, multiview) and the G-buffer.
 * The reference draws them with the hardware rasteriser; here they are compute passes (bin to 64x64 tiles, depth test in LDS)
 * that follow the Vulkan rasterisation rules as DESIGN.md §5d pins them down: fixed-point window coordinates with 8 sub-pixel
 * bits, pixel-centre sampling, top-left rule, clockwise front faces, screen-linear depth, perspective-correct attributes.
 * Everything below is a DEVICE pointer unless noted. */

/* StandardVertexData — RenderCore/shared/vertex_data.hpp:22-27 (40 bytes; positions live in their own float3 stream). */
typedef struct sah_vertex_data {
    float normal[3];
    float tangent[4];
    float texcoord[2];
    uint32_t color; /* unorm4, r in the low byte */
} sah_vertex_data;

/* BasicPbrMaterialGpu — RenderCore/shared/basic_pbr_material.hpp:6-19, with each bindless texture index replaced by the
 * texel a 1x1 texture of that slot would return; real textures are bound per material through sah_material_textures below. */
typedef struct sah_material {
    float base_color_tint[4];
    float emission_factor[4];
    float metalness_factor, roughness_factor, opacity_threshold, padding1;
    float base_color_texel[4], normal_texel[4], data_texel[4], emission_texel[4];
} sah_material;

/* ---- material textures: `textures[material.*_texture_index].SampleBias(vertex.texcoord, mip_bias)`, gltf_basic_pbr.slang:177-226 ----
 * A bindless slot is an R8G8B8A8 image with its mip chain and the sampler it is bound with (gltf_model.cpp:229-279, 520-586; the
 * default sampler of render_backend.cpp:1129-1134 is all-NEAREST / REPEAT).  Vulkan leaves the level-of-detail arithmetic, the
 * derivative quads and anisotropic footprints to the implementation; this library fixes them as follows (DESIGN.md §5d "Sampling"):
 *   texcoord   the float2 varying, perspective-correct in fp32: (l0 t0 + l1 t1) + l2 t2 with the input-triangle barycentrics l;
 *   d/dx, d/dy fine quad differences: quads are the 2x2 pixel blocks at even window coordinates, d/dx = t(x | 1, y) - t(x & ~1, y),
 *              d/dy likewise; the varying of the other pixel is the same triangle's, extrapolated when it is not covered;
 *   lod        rho2 = fmax(mx.x^2 + mx.y^2, my.x^2 + my.y^2), m = derivative * size of level 0 (every operator fp32);
 *              lambda = rho2 > 0 ? 0.5 * RN32(log2 evaluated in double) : -inf;  lambda += (sampler.mip_lod_bias + shader bias);
 *              lambda = fmin(fmax(lambda, min_lod), max_lod);
 *   anisotropy sampler.max_anisotropy A > 1 (gltf_model.cpp:581-584: 8 with a LINEAR mipmap mode; values <= 1 or NaN: off; at most 16),
 *              the example formulation of the Vulkan specification ("Texel Anisotropic Filtering") with every operator in fp32:
 *              rx = mx.x^2 + mx.y^2, ry = my.x^2 + my.y^2 (as above), rmax2 = fmax(rx, ry), rmin2 = fmin(rx, ry), major axis = x when
 *              rx > ry, else y;  eta = 1 when rmax2 is not > 0, A when rmin2 is not > 0, else fmin(sqrt(rmax2 / rmin2), A);
 *              N = ceil(eta);  lambda = 0.5 * RN32(log2 rmax2) - RN32(log2 eta) (then bias and clamps as above; -inf when rmax2 is
 *              not > 0);  N == 1: the isotropic sample.  Otherwise tap i = 1..N at texcoord + (i / (N + 1) - 0.5) * (the major
 *              axis' derivative of the texcoord), each a complete sample at that lambda (filter, level blend), summed from +0 in the
 *              order of i and divided by (float)N.  SampleLevel (the ray-tracing stages) has no derivatives: always isotropic;
 *   filter     lambda <= 0 ? mag_filter : min_filter;
 *   level      mipmap NEAREST: lambda <= 0.5 ? 0 : min(ceil(lambda + 0.5) - 1, q), q = num_mips - 1;
 *              mipmap LINEAR:  d = clamp(lambda, 0, q), hi = floor(d), lo = min(hi + 1, q), delta = d - hi,
 *                              result = (1 - delta) * tau(hi) + delta * tau(lo), every operator rounded;
 *   tau        LINEAR: the bilinear weighted sum of the other samplers of this library (u * w - 0.5, fma chain from +0 in the order
 *              (i,j) (i+1,j) (i,j+1) (i+1,j+1)); NEAREST: texel (floor(u * w), floor(v * h)); indices wrapped per address mode;
 *              a NaN coordinate yields NaN in every channel;
 *   texel      UNORM: byte / 255 in fp32; SRGB: rgb through the exact sRGB decode rounded to fp32, alpha as UNORM.
 * The shader bias is view->material_texture_mip_bias in the G-buffer pass and 0 in the shadow and RSM passes (:175-178). */
#define SAH_FILTER_NEAREST 0 /* VkFilter / VkSamplerMipmapMode values */
#define SAH_FILTER_LINEAR 1
#define SAH_ADDRESS_REPEAT 0 /* VkSamplerAddressMode values */
#define SAH_ADDRESS_MIRRORED_REPEAT 1
#define SAH_ADDRESS_CLAMP_TO_EDGE 2
#define SAH_MAX_TEXTURE_MIPS 14
#define SAH_TEXTURE_NONE 0xffffffffu

typedef struct sah_sampler {
    uint32_t mag_filter, min_filter, mipmap_mode;
    uint32_t address_u, address_v;
    float mip_lod_bias, min_lod, max_lod;
    float max_anisotropy; /* VkSamplerCreateInfo::maxAnisotropy when anisotropyEnable, else 0 */
    uint32_t reserved;
} sah_sampler;

typedef struct sah_texture {
    sah_plane mips[SAH_MAX_TEXTURE_MIPS]; /* level 0 first; R8G8B8A8_UNORM or R8G8B8A8_SRGB, every level the same format */
    uint32_t num_mips;                    /* 1 .. SAH_MAX_TEXTURE_MIPS */
    uint32_t padding;
    sah_sampler sampler;
} sah_texture;

/* The four texture indices of BasicPbrMaterialGpu (basic_pbr_material.hpp:15-18), one record per material; SAH_TEXTURE_NONE selects
 * the constant texel of sah_material for that slot. */
typedef struct sah_material_textures {
    uint32_t base_color, normal, data, emission;
} sah_material_textures;

#define SAH_PRIMITIVE_TYPE_SOLID 0  /* back faces culled (render_scene.cpp:196-197) */
#define SAH_PRIMITIVE_TYPE_CUTOUT 1 /* no culling, alpha test against opacity_threshold (render_scene.cpp:221-222) */

/* One draw: PrimitiveDataGPU::model (RenderCore/shared/primitive_data.hpp:33-50) plus the VkDrawIndexedIndirectCommand range. */
typedef struct sah_primitive {
    float model[16];
    uint32_t first_index, index_count;
    int32_t vertex_offset;
    uint32_t type;
    uint32_t material;
    uint32_t padding[3];
} sah_primitive;

typedef struct sah_scene_geometry {
    const float* vertex_positions;      /* 3 floats per vertex, tightly packed */
    const sah_vertex_data* vertex_data; /* sah_shadow_render: may be NULL when the scene has no CUTOUT primitive */
    const uint32_t* indices;
    const sah_primitive* primitives;    /* list order = draw order inside a class; all SOLID primitives are drawn before all CUTOUT ones,
                                           wherever they sit in the list (RenderScene::draw_opaque, then draw_masked) */
    const sah_material* materials;      /* sah_shadow_render: may be NULL when the scene has no CUTOUT primitive */
    uint32_t num_vertices, num_indices, num_primitives, num_materials;
    const sah_texture* textures;                    /* num_textures slots, or NULL */
    const sah_material_textures* material_textures; /* num_materials records, or NULL: every slot of every material is its constant texel */
    uint32_t num_textures, padding;
} sah_scene_geometry;

/* Written by both passes when `stats` is not NULL (device memory, 8 x uint32): [0] (view, triangle) pairs processed, [1] culled,
 * degenerate, clipped away or covering no pixel centre, [2] dropped (non-finite or out-of-range coordinates, indices outside the
 * arrays), [3] window-space triangles rasterised (after clipping and fanning), [4] (tile, triangle) bin entries, [5] further parts of bin lists
 * that were cut into pieces of 256 entries, [6] tiles whose list was cut, [7] reserved. */
#define SAH_RASTER_STATS_WORDS 8

/* DirectionalLight::render_shadows — RenderCore/render/directional_light.cpp:286-327 with the `_shadow` pipelines of
 * RenderCore/render/material_pipelines.cpp:31-62 (compare LESS, depth clamp) and the SAH_MULTIVIEW vertex stage of
 * RenderCore/shaders/materials/gltf_basic_pbr.slang:110-146.  Clears every layer of `shadowmap` (D16_UNORM 2D array) to 1.0 and
 * rasterises every primitive into each of the `num_cascades` layers with sun->cascade_matrices[layer].  CUTOUT primitives go through
 * the `_shadow_masked` fragment stage (Tools/compile_shaders.py:110-112: SAH_DEPTH_ONLY, SAH_MASKED; gltf_basic_pbr.slang:181-196):
 * a fragment whose tinted_base_color.a <= opacity_threshold writes no depth.  That needs scene->vertex_data and scene->materials; a
 * scene that has CUTOUT primitives but came without them fails with SAH_ERR_INVALID_ARGUMENT (the image is then undefined). */
int sah_shadow_render(sah_ctx* ctx, const sah_scene_geometry* scene, const sah_sun_light_constants* sun, uint32_t num_cascades,
                      const sah_volume* shadowmap, uint32_t* stats);

/* Depth pre-pass + G-buffer pass — RenderCore/render/phase/gbuffer_phase.cpp:27-97 (clear values :66-87), pipelines
 * RenderCore/render/material_pipelines.cpp:13-29,104-140 (reverse-Z GREATER, then EQUAL), shaders
 * RenderCore/shaders/materials/gltf_basic_pbr.slang:110-253 (SAH_MAIN_VIEW).  Writes all five planes of `out`.  A texture index
 * outside scene->textures, or a texture whose levels are not R8G8B8A8, fails with SAH_ERR_INVALID_ARGUMENT. */
int sah_gbuffer_render(sah_ctx* ctx, const sah_scene_geometry* scene, const sah_view_data* view, const sah_gbuffer* out, uint32_t* stats);

/* ---- LPV injection chain: RSM -> VPL list -> LPV (SURVEY.md §8-f4 and the producers of a3's volumes) -------------------------
 * LightPropagationVolume::inject_indirect_sun_light — RenderCore/render/gi/light_propagation_volume.cpp:548-697. */

/* RSM targets, 2D arrays with one layer per LPV cascade (light_propagation_volume.cpp:422-452). */
typedef struct sah_rsm_targets {
    sah_volume flux;    /* R8G8B8A8_SRGB  */
    sah_volume normals; /* R8G8B8A8_UNORM */
    sah_volume depth;   /* D16_UNORM      */
} sah_rsm_targets;

/* "Render RSM" (light_propagation_volume.cpp:566-615): every primitive into every cascade layer through cascades[i].rsm_vp with the
 * `_rsm` pipelines (material_pipelines.cpp:69-102: compare LESS, no depth clamp) and the SAH_RSM stages of
 * RenderCore/shaders/materials/gltf_basic_pbr.slang:110-253 — flux = Fd(surface, -sun direction, normal) with the metalness and
 * roughness the shader leaves at 0 in this variant, normal * 0.5 + 0.5.  Same rasteriser and rules as sah_shadow_render. */
int sah_rsm_render(sah_ctx* ctx, const sah_scene_geometry* scene, const sah_sun_light_constants* sun, const sah_lpv_cascade_matrices* cascades,
                   uint32_t num_cascades, const sah_rsm_targets* rsm, uint32_t* stats);

/* PackedVPL — RenderCore/shared/vpl.hpp:21-23: position.xy | position.z, color.r | color.gb as halfs, normal as snorm4x8. */
typedef struct sah_packed_vpl {
    uint32_t data[4];
} sah_packed_vpl;

/* "Extract VPLs" — RenderCore/shaders/gi/lpv/rsm_generate_vpls.comp:86-139, one invocation per 2x2 RSM texels of layer
 * `cascade_index`.  The shader appends with atomicAdd, so the order of its list is not a function of the input; this ABI stores the
 * lights in ascending invocation index (row-major over the (res/2)^2 invocations), which also fixes the order in which
 * sah_lpv_inject_vpls adds them.  round() and packSnorm4x8 round half to even.  vpl_list: room for (res/2)^2 entries; vpl_count:
 * one uint32; both DEVICE pointers. */
int sah_lpv_extract_vpls(sah_ctx* ctx, const sah_rsm_targets* rsm, const sah_lpv_cascade_matrices* cascades, uint32_t cascade_index,
                         float grid_cell_size, sah_packed_vpl* vpl_list, uint32_t* vpl_count);

/* "VPL Injection" — the point-list render pass of light_propagation_volume.cpp:699-760 with
 * RenderCore/shaders/gi/lpv/vpl_injection.{vert,frag} and additive blending into the three RGBA16F volumes: VPL i lands in the cell
 * (floor(x_f), floor(y_f), int(z * 32)) of its cascade, lights that fall outside the volume are dropped, and lights of one cell are
 * added in list order, each sum rounded to half (what the blend unit does in primitive order). */
int sah_lpv_inject_vpls(sah_ctx* ctx, const sah_packed_vpl* vpl_list, const uint32_t* vpl_count, uint32_t capacity,
                        const sah_lpv_cascade_matrices* cascades, uint32_t cascade_index, uint32_t num_cascades, const sah_volume rgb[3]);

/* ---- ray tracing: the acceleration structure and the two generators whose outputs sah_lighting consumes ----------
 * The reference traces against a Vulkan TLAS with one instance per primitive (RenderCore/render/raytracing_scene.cpp:15-43: transform
 * = primitive.model, SOLID -> FORCE_OPAQUE, CUTOUT -> FORCE_NO_OPAQUE, both faces hit) over per-mesh BLASes.  What a ray hits is then
 * the implementation's business (traversal order, intersection arithmetic, watertightness).  This library fixes it:
 *
 *   triangles   world space: vertex k = model * (p_k, 1), each row ((m0 x + m1 y) + m2 z) + m3 in fp32 (the rasteriser's vertex stage);
 *               a triangle with a non-finite world vertex, or an index outside the arrays, is not in the structure.
 *   pad         S * 2^-16, S = the largest |coordinate| of the world vertices in the structure (fp32 product).
 *   box test    box(T) = [min_k v_k - pad, max_k v_k + pad] per axis (fp32).  slab(ray, box): per axis inv = 1 / d (IEEE, +-inf for
 *               +-0), t0 = (lo - o) * inv, t1 = (hi - o) * inv, near = minNum(t0, t1), far = maxNum(t0, t1) (a NaN operand is ignored);
 *               passes iff maxNum(maxNum(maxNum(near_x, near_y), near_z), tmin) <= minNum(minNum(minNum(far_x, far_y), far_z), tmax).
 *   triangle    Woop / Benthin / Wald 2013, every operator an fp32 operation: kz = axis of the largest |d| (first of equals), kx, ky
 *               the next two axes cyclically, swapped when d[kz] < 0; Sx = d[kx] / d[kz], Sy = d[ky] / d[kz], Sz = 1 / d[kz];
 *               A = v0 - o (likewise B, C), Ax = A[kx] - Sx * A[kz], Ay = A[ky] - Sy * A[kz];
 *               U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax; when one of them is exactly 0 all three are
 *               recomputed in fp64 from the same fp32 operands (exact products, one rounding) and rounded to fp32;
 *               miss when (U < 0 or V < 0 or W < 0) and (U > 0 or V > 0 or W > 0); det = (U + V) + W, miss when det == 0;
 *               T = (U * (Sz * A[kz]) + V * (Sz * B[kz])) + W * (Sz * C[kz]);  t = T / det;  barycentrics (V / det, W / det) are the
 *               weights of v1 and v2 (the `barycentrics` attribute of the hit shaders).
 *   hit         candidate(ray, T) <=> slab(ray, box(T)) passes and the triangle test gives tmin < t < tmax (NaN compares false).
 *               Making the padded box part of the DEFINITION is what lets a box hierarchy cull exactly: slab() is monotone under box
 *               inclusion in fp32, so a parent box never rejects a ray its child accepts, and the result is the same as testing
 *               every triangle — which is what the oracle does.
 *   any hit     an occlusion ray asks "is there an accepted candidate": a candidate on a SOLID primitive is accepted; one on a CUTOUT
 *               primitive runs the any-hit stage of RenderCore/shaders/materials/gltf_basic_pbr.slang:291-318 — texcoord
 *               (b0 t0 + b1 t1) + b2 t2 with b0 = (1 - b1) - b2 in fp32, vertex colour through unpackUnorm4x8ToHalf / packUnorm4x8
 *               (half arithmetic, truncating pack), base-colour texture at level 0 (SampleLevel: lambda = 0 + sampler bias, clamped,
 *               then the rules of sah_texture), alpha = (texel.a * tint.a) * colour.a in fp32 — and is ignored when
 *               alpha <= opacity_threshold.  Existence does not depend on the order in which candidates are found.
 *   closest hit the accepted candidate with the smallest t; among equal t the one with the smallest (primitive index, triangle index).
 *               Facing: a triangle is FRONT facing when (v1 - v0) x (v2 - v0) points against the ray, i.e. its vertices appear
 *               counter-clockwise from the ray origin in the scene's right-handed world space.  That is the default of Vulkan and
 *               D3D12 ("clockwise in a left-handed system", the same rule on the same numbers; no FLIP_FACING bit in
 *               raytracing_scene.cpp:19-37), and it agrees with the rasteriser: what the G-buffer pass keeps as front facing is front
 *               facing for a ray from the camera.  Evaluated as det > 0 on the fp32 determinant above (the axis permutation keeps
 *               the winding).
 *               RAY_FLAG_CULL_FRONT_FACING_TRIANGLES / CULL_BACK_FACING_TRIANGLES drop candidates by that sign before anything else.
 *   a ray with a non-finite origin or direction component hits nothing.
 *   tmax        the tmax of every ray is minNum(tmax, FLT_MAX): an infinite (or NaN) distance means the largest finite float, in slab()
 *               and in the triangle test alike (sah_rtao's max_ray_distance and 4 * probe_spacing of sah_probe_trace are the two
 *               caller-given ones).
 * The structure itself (an implicit 4-wide hierarchy over Morton-sorted triangles, built on the GPU) is an implementation detail. */

/* RaytracingScene::add_primitive / commit_tlas_builds — RenderCore/render/raytracing_scene.cpp:15-170 (and the BLAS builds behind
 * mesh->blas): (re)builds the context's acceleration structure over every primitive of `scene`.  The arrays `scene` points to must stay
 * valid and unchanged while rays are traced (the any-hit stage reads vertex_data, materials and textures).  Synchronises the context's
 * stream once (the triangle count is read back).  stats: HOST pointer to 4 words or NULL — [0] triangles in the structure, [1] triangles
 * left out (non-finite, bad indices), [2] hierarchy levels, [3] reserved. */
#define SAH_RT_STATS_WORDS 4
int sah_rt_build(sah_ctx* ctx, const sah_scene_geometry* scene, uint32_t* stats);

/* AmbientOcclusionPhase::evaluate_rtao — RenderCore/render/phase/ambient_occlusion_phase.cpp:357-397 with
 * RenderCore/shaders/ao/rtao.comp.slang:54-102: one invocation per texel of ao_out (R32_SFLOAT).  Ray origin = the Slang world-space
 * position of the pixel ((pixel + 0.5) / render_resolution, every component divided by w), direction = normalize(noise[pixel % noise
 * extent].rgb * 2 - 1) flipped into the hemisphere of normalize((half3)normal), [0.01, max_ray_distance], RAY_FLAG_ACCEPT_FIRST_HIT_AND_
 * END_SEARCH | RAY_FLAG_CULL_NON_OPAQUE (CUTOUT primitives are invisible to it); ao = (spp - hits) / spp.  Kept quirks: every one of
 * the samples_per_pixel rays reads the same noise texel, so they are one ray; no depth == 0 test (a sky pixel's origin is not finite,
 * nothing is hit, ao = 1); the unused rotation matrix.  noise: R8G8B8A8_UNORM (a blue-noise layer, RenderCore/render/noise_texture.cpp),
 * extent <= 65535; samples_per_pixel <= 4096.  depth: D32_SFLOAT, normals: R16G16B16A16_SFLOAT, both of ao_out's extent. */
int sah_rtao(sah_ctx* ctx, const sah_view_data* view, const sah_plane* depth, const sah_plane* normals, const sah_plane* noise,
             uint32_t samples_per_pixel, float max_ray_distance, const sah_plane* ao_out);

/* The shadow rays of DirectionalLight::raytrace — RenderCore/render/directional_light.cpp:372-422 with
 * RenderCore/shaders/lighting/directional_light.rt.slang:91-125 and the occlusion hit groups of gltf_basic_pbr.slang:291-325 / the
 * occlusion miss shader of sky_unified.slang:210-215: writes shadow / num_shadow_samples, the factor the raygen shader multiplies the sun
 * radiance with, to mask_out (R32_SFLOAT) — the `shadow_mask` plane of sah_lighting's RT mode, which evaluates the rest of that shader.
 * For sample i the ray leaves the pixel's world-space position towards normalize(L + noise_i * tan_size), L = normalize(-direction),
 * noise_i = normalize(noise[(pixel + round(r0(i) * 128)) % 128].rgb * 2 - 1), r0(i) = (fract(2 + i / PHI), fract(3 + i / PHI)) with
 * PHI = 1.618033988749895 as fp32, round() to nearest even; [0.01, 100000]; shadow counts the rays that hit nothing.  Pixels the shader
 * does not trace for (depth == 0, or (half)clamp(dot(L, normal), 0, 1) == 0) get 1.0.  noise: R8G8B8A8_UNORM, at least 128 x 128. */
int sah_sun_shadow_mask(sah_ctx* ctx, const sah_view_data* view, const sah_sun_light_constants* sun, const sah_plane* depth,
                        const sah_plane* normals, const sah_plane* noise, const sah_plane* mask_out);

/* The GI rays (ray type RAY_TYPE_GI): closest hit over SOLID and alpha-tested CUTOUT geometry, shaded by the closest-hit stage of
 * RenderCore/shaders/materials/gltf_basic_pbr.slang:345-470 — vertex attributes interpolated with the hit's barycentrics, position
 * model * (b0 p0 + b1 p1 + b2 p2), base colour / data / emission textures at level 0, surface.normal = the interpolated OBJECT-space normal
 * (the shader does not transform it), irradiance = Fd(surface, L, normal) * sun colour * ndotl * shadow + emission in the shader's half /
 * float mix, shadow from one ray towards normalize(L + noise * tan_size) with RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | CULL_NON_OPAQUE |
 * CULL_FRONT_FACING_TRIANGLES and TMin 0.05, noise = noise[DispatchRaysIndex().xy % 128]; a back-face hit returns -t and no light;
 * remaining_bounces is 0 in both generators, so the bounce branch is dead — or, on a miss, by the GI miss shader of
 * shaders/sky/sky_unified.slang:227-230: get_sky_color(ray direction, sun direction AS STORED (not negated: quirk), ...). */

/* IrradianceCache::dispatch_probe_updates, the "probe_tracing" pass — RenderCore/render/gi/irradiance_cache.cpp:585-640 with
 * shaders/gi/cache/probe_tracing.rt.slang:39-106: 20 x 20 rays per listed probe along the octahedral directions of the trace texels, from
 * cascade.min + local probe id * spacing, [0.05, 4 x the next cascade's spacing] (8192 for the last cascade).  A miss samples the NEXT
 * cascade of the cache at the ray's end (sample_cascade, probe_sampling.slangi:6-106, on the atlases given here) or, in the last cascade,
 * keeps 10 x the sky colour, and reports the ray's full length; a back-face hit reports -t and black.  Writes
 * half4(irradiance * 0.0031415927h, distance) to trace_results (R16G16B16A16_SFLOAT, 20 x 20 x num_probes) — the input of sah_probe_update. */
typedef struct sah_probe_trace_desc {
    sah_probe_cascade cascades[4];
    const uint32_t* probes_to_update; /* DEVICE: num_probes uint32 triples (probe x, y, layer) */
    uint32_t num_probes;
    const sah_sun_light_constants* sun;
    const sah_sky_luts* sky;
    const sah_plane* noise;           /* R8G8B8A8_UNORM, at least 128 x 128 */
    sah_volume probe_irradiance;      /* the cache as sah_gi describes it: B10G11R11 224 x 256 x 32 */
    sah_volume probe_depth;           /* R16G16_SFLOAT 384 x 384 x 32 */
    sah_volume probe_validity;        /* R8_UNORM 32 x 32 x 32 */
    uint32_t probe_size[2];           /* (5, 6) */
    sah_volume trace_results;         /* out */
} sah_probe_trace_desc;
int sah_probe_trace(sah_ctx* ctx, const sah_probe_trace_desc* desc);

/* RayTracedGlobalIllumination::post_render — RenderCore/render/gi/rtgi.cpp:69-139 with shaders/gi/rtgi/rtgi.rt.slang:56-110: one GI ray
 * per pixel with depth != 0 from its world-space position along normalize(noise[pixel % 128].rgb * 2 - 1), flipped into the hemisphere of
 * the (unnormalised) G-buffer normal, [0.01, 100000].  Writes (direction, distance) to ray_buffer and (irradiance * 0.0031415927, 0) to
 * ray_irradiance (both R16G16B16A16_SFLOAT: the inputs of the RTGI overlay, sah_gi::ray_buffer / ray_irradiance); a NaN irradiance becomes
 * 0; on a miss the distance stays 0.  Pixels with depth == 0 are left untouched, as the shader leaves them. */
int sah_rtgi_trace(sah_ctx* ctx, const sah_view_data* view, const sah_sun_light_constants* sun, const sah_sky_luts* sky, const sah_plane* depth,
                   const sah_plane* normals, const sah_plane* noise, const sah_plane* ray_buffer, const sah_plane* ray_irradiance);

/* Row window of the three per-pixel ray generators (sah_rtao, sah_sun_shadow_mask, sah_rtgi_trace), for frames sharded by rows
 * (no reference counterpart): from now on they trace and write output rows [row_begin, row_end) only, clipped to the plane; the other
 * rows keep their contents.  (0, 0) = every row, the state of a new context.  A pixel's result does not depend on the window. */
int sah_rt_set_rows(sah_ctx* ctx, uint32_t row_begin, uint32_t row_end);

/* payload.remaining_bounces of the rays the two GI generators (sah_probe_trace, sah_rtgi_trace) trace from now on — the bounce branch of the
 * GI hit stage, RenderCore/shaders/materials/gltf_basic_pbr.slang:481-517: a front-face hit with bounces left traces one more GI ray from the
 * hit point along the launch index's noise direction (flipped into the hit normal's hemisphere; [0.05, 100000]; non-opaque and back-facing
 * triangles culled) with one bounce fewer, and adds ndotl * brdf * its irradiance when that is finite.  0..2; a new context holds 0, which is
 * what the reference's generators set (rtgi.rt.slang:88, probe_tracing.rt.slang:66 — they upload r.GI.NumBounces as a push constant,
 * rtgi.cpp:131, and never read it): with 0 the results are the reference's, with 1 or 2 they are what its hit stage computes once a generator
 * forwards the constant. */
int sah_rt_set_bounces(sah_ctx* ctx, uint32_t num_bounces);

/* Multi-GPU exchange step (no reference counterpart: the reference drives one device, RenderCore/render/backend/render_backend.cpp:135-153;
 * BASELINE.json north_star: "RCCL all-gather over xGMI to reassemble the final image").
 * In-place all-gather of row blocks of `image` over RCCL on the context's stream: rank r owns rows [rows_per_rank*r, rows_per_rank*(r+1))
 * clipped to image->height.  Slots are equal, so when world does not divide the height the caller pads: rows_per_rank = ceil(height / world)
 * and the buffer behind image->ptr holds `allocated_rows` >= rows_per_rank * world rows of image->row_pitch_bytes (the rows past
 * image->height are scratch).  Fails with SAH_ERR_INVALID_ARGUMENT when rows_per_rank * world < height (rows would stay ungathered) or
 * > allocated_rows.  With one rank and no communicator it is a no-op. */
int sah_allgather_rows(sah_ctx* ctx, const sah_plane* image, uint32_t rows_per_rank, uint32_t allocated_rows);
/* Optional: run the exchange on a side stream (an externally owned hipStream_t; NULL = back to the work stream).  Each gather then
 * starts when the work enqueued before it has finished and runs beside the work enqueued after it; sah_comm_wait() makes the work
 * stream wait for the last gather (call it before touching the gathered buffer again).  Without a side stream gathers are ordinary
 * entries of the work stream and sah_comm_wait() is a no-op. */
int sah_comm_set_stream(sah_ctx* ctx, void* hip_stream);
int sah_comm_wait(sah_ctx* ctx);
/* As sah_allgather_rows, with the slots in reversed rank order: rank r owns rows [rows_per_rank*(world-1-r), rows_per_rank*(world-r)).
 * The final composite samples the scene upside down (scene_upsample.frag / fullscreen.vert: v = 1 - (y + 0.5) / H), so the rank
 * that shaded scene rows near the top produces output rows near the bottom: shard the scene by rank, gather the RGBA8 rows reversed. */
int sah_allgather_rows_reversed(sah_ctx* ctx, const sah_plane* image, uint32_t rows_per_rank, uint32_t allocated_rows);
/* Same exchange on a raw DEVICE buffer of world * bytes_per_rank bytes; rank r owns [r * bytes_per_rank, (r+1) * bytes_per_rank). */
int sah_allgather_bytes(sah_ctx* ctx, void* buffer, uint64_t bytes_per_rank);

/* ---- direct exchange: the same gathers as one hop per peer over peer-mapped memory, without RCCL -------------------------------
 * The gathers of this path move 4-33 MB per frame between the 8 GPUs of one node, whose xGMI links are point to point: every rank
 * can store its rows straight into every peer's image ("a direct one-shot exchange over the 7 links").  The ranks are
 * processes of one node; buffers are made visible to each other with HIP IPC handles, which the CALLER carries between the ranks
 * over whatever channel it has (the handles are plain bytes):
 *   1. every rank: sah_ipc_open(ctx, my_handle);           the rank's mailbox (arrival counters in fine-grained device memory)
 *   2. all-gather the handles;  sah_ipc_connect(ctx, all_handles)                         — world * SAH_IPC_HANDLE_BYTES, rank order
 *   3. per gathered buffer, in the same order on every rank:
 *        sah_ipc_export(ctx, buffer, bytes, my_handle);  all-gather;  sah_ipc_register(ctx, buffer, bytes, all_handles)
 * From then on sah_allgather_rows / _rows_reversed / _bytes on memory inside a registered buffer take this path (a context may have
 * an RCCL communicator as well: unregistered buffers keep using it).  One gather = signal "my rows are ready and my copy of the buffer
 * may be overwritten" to every peer, wait for theirs, copy the own slot into every peer's buffer (hipMemcpyAsync on the exchange stream),
 * signal "done", wait for theirs — counters only ever grow, so nothing is reset between frames, and sah_comm_set_stream /
 * sah_comm_wait order it against the work stream exactly as they do the RCCL path.  A peer that does not arrive within two seconds
 * makes the waiting kernel give up (no kernel spins forever) and raise a device word that every later step of that gather, and of every
 * later gather, tests first: no copy into a peer and no "done" signal leaves a rank that has given up (the copy is a kernel of this
 * library for that reason, not hipMemcpyAsync).  sah_sync — which then also waits for the exchange stream — and sah_comm_wait return
 * SAH_ERR_COMM from then on, as does every later gather: rows gathered by a call that has not been followed by a successful sah_sync
 * are not known to be valid.  sah_ipc_open fails with SAH_ERR_UNSUPPORTED on a device without fine-grained device memory.
 * Registrations are found by address and must not overlap: before a registered buffer is freed (or its address re-used) call
 * sah_ipc_unregister(ctx, buffer) on every rank, in the same order as everything else of this protocol; it waits for the context's
 * streams, closes the peer mappings nobody else uses and frees the slot (the lowest free slot is taken by the next registration).
 * The context may be created with comm_id == NULL and world > 1 for this path.  buffer may lie inside a larger allocation (a caching
 * allocator's block): the handle carries the offset.
 * WHICH MEMORY.  Gathered buffers must be ordinary device memory (hipMalloc) of the context's own device; sah_ipc_export and
 * sah_ipc_register refuse host-pinned and managed memory (SAH_ERR_UNSUPPORTED) and memory of another device.  Only the mailbox is
 * fine-grained; the rows a peer stores land in the home device's ordinary (coarse-grained) allocation, as the rows RCCL's kernels store
 * into a user buffer do, and they are published the same way: the peer's copy kernel ENDS (release at system scope) before its "done" counter
 * is stored, and the home device reads the rows only in kernels that START behind the kernel that saw that counter (stream order on the
 * exchange stream, sah_comm_wait on the work stream) — a kernel boundary, whose acquire makes the home device's caches drop what they hold
 * of memory a peer may have written.  A consumer that polls gathered rows from INSIDE a running kernel is outside this contract.
 * STATUS: verified with several processes on ONE device (tests/test_comm_gpu.py); on two devices it has not run yet (no multi-GPU box
 * was available to the builder): RCCL is the default exchange of every entry point and of bench.py, the direct exchange is opt-in, and
 * tests/test_comm_gpu.py's two-GPU tests (skipped on one GPU) are the first thing to run on a node — README.md "On a multi-GPU node". */
#define SAH_IPC_HANDLE_BYTES 128
#define SAH_IPC_MAX_WORLD 16
#define SAH_IPC_MAX_BUFFERS 16
int sah_ipc_open(sah_ctx* ctx, void* out_handle);
int sah_ipc_connect(sah_ctx* ctx, const void* all_handles);
int sah_ipc_export(sah_ctx* ctx, const void* buffer, uint64_t bytes, void* out_handle);
int sah_ipc_register(sah_ctx* ctx, void* buffer, uint64_t bytes, const void* all_handles);
int sah_ipc_unregister(sah_ctx* ctx, const void* buffer);
/* After SAH_ERR_COMM (a gather of the direct exchange gave up: a peer stalled for more than two seconds, a rank skipped a gather).
 * The state is sticky on purpose — rows gathered behind a give-up are not valid, and a rank that has given up neither copies into its
 * peers nor is copied into by them (its "gave up" note reaches every peer's mailbox with the give-up; a peer that arrives late reads it
 * before it copies).  What the caller must NOT do on the error is free or unregister a gathered buffer at once: a peer that saw no
 * error yet may be in the middle of its copy into it.  The order out of the state, on EVERY rank:
 *     sah_sync (drain; it reports the error once more)  ->  barrier over the caller's own channel  ->  sah_ipc_reset  ->  barrier
 * after which gathers work again and buffers may be unregistered and freed as usual.  Gathers that one rank made (or merely enqueued
 * behind the one that gave up) and another skipped have left their sequence numbers apart; nothing is in flight between the two barriers,
 * so sah_ipc_reset starts the exchange over from zero: it clears this rank's whole mailbox (arrival counters, give-up notes, abort word) and
 * the sequence number of every registration slot.  A no-op on a context that has not opened the exchange. */
int sah_ipc_reset(sah_ctx* ctx);

/* ---- the row-sharded frame as a loop of this library (no reference counterpart: north_star's "shard by screen-tile rows, all-gather the
 * final image"; the reference records one frame at a time on one queue, render_backend.cpp:135-153) ---------------------------------------
 * One rank's share of the whole chain, two frames in flight, so that both exchanges travel beside compute.  A frame has three parts:
 *     L(i)  Lighting of the rank's rows (LightingPhase::render)                                                          work stream
 *     R(i)  "Copy scene" + bloom mip 0 over its band (scene_renderer.cpp:502-527, bloomer.cpp:50-72), its rows of mip 1   reduce stream
 *           -> exchange of mip 1
 *     B(i)  mips 2.. (bloomer.cpp:74-262), the composite of its rows (ui_phase.cpp:98-113)                                post stream
 *           -> exchange of the R8G8B8A8 rows, reversed rank order
 * sah_chain_submit enqueues L(i), R(i), the mip-1 gather, then B(i - 1) and its final gather.  Everything a frame writes exists twice
 * (`frames[2]`, used alternately); a part starts behind the last reader (a part of frame i - 2) of what it overwrites.  The three streams
 * may coincide (reduce_stream NULL = the work stream, post_stream NULL = the reduce stream): on one stream the order is the stream's;
 * with a post stream B(i - 1) runs beside L(i) and R(i); with a reduce stream as well the lighting of frame i + 1 starts when that of
 * frame i ends instead of behind its copy and mip rows — the shape that pays on a rank of eight, whose parts are each too small to fill the
 * chip by themselves.  The exchanges run on the context's side stream if one is set (sah_comm_set_stream), through RCCL or the direct
 * exchange as sah_allgather_rows would.
 * The row arithmetic is the caller's (androidrenderer_amd/shard.py: chain_plan; include/sah_host.hpp); this object only keeps the order
 * and the events, which is what cost the host 72 us per frame when every call crossed a language boundary.
 * All descriptors are copied (pointers into HOST memory need not outlive sah_chain_create); device memory must outlive the chain. */
typedef struct sah_chain sah_chain;
typedef struct sah_chain_plan {
    uint32_t aa_rows[2];   /* rows of `antialiased` this rank needs: [begin, end) */
    uint32_t mip0_rows[2]; /* rows of bloom mip 0 it computes for itself */
    uint32_t mip1_rows[2]; /* rows of bloom mip 1 it contributes to the first exchange (may be empty) */
    uint32_t out_rows[2];  /* rows of the final image it composites (slot world - 1 - rank; may be empty) */
    uint32_t mip1_rows_per_rank, mip1_allocated_rows; /* gather slot of mip 1 and the rows its allocation holds (>= slot * world) */
    uint32_t rows_per_rank, out_allocated_rows;       /* the same for the final image */
} sah_chain_plan;
typedef struct sah_chain_frame {
    const sah_lighting_desc* lighting[2]; /* the rank's lit rows; the row on the opposite edge that "Copy scene"'s REPEAT sampler taps, or NULL */
    sah_plane lit, antialiased;
    sah_mipchain bloom;
    sah_plane out; /* R8G8B8A8 */
} sah_chain_frame;
/* work_stream / reduce_stream / post_stream: see above (the context is left on the work stream after every call).  tonemap_flags as for
 * sah_tonemap_ex.  chain_flags: SAH_CHAIN_NO_EXCHANGE = both gathers are left out (one rank's compute of an N-rank plan on a context of
 * another world size: rehearsals and measurements of a rank's share on one GPU). */
#define SAH_CHAIN_NO_EXCHANGE (1u << 0)
/* SAH_CHAIN_CAPTURE: the parts of a frame are captured into HIP graphs the second time a buffer set is used and replayed afterwards — one
 * graph launch instead of one to three kernel launches per part (less host time per frame, MORE device time: a replay has a start-up of
 * its own; measured in tools/experiments/r5/README.md — off by default).  A replay enqueues exactly what
 * the captured calls enqueued, so it is only used while nothing those calls depend on has changed (context buffers and tables, the gather
 * copies' state: the context counts such changes) — any other use of the context in between sends the chain back to direct calls and a new
 * capture — and not for a submit that asks for the Lighting pass's events.  Frame descriptors are fixed at sah_chain_create either way; what
 * the planes and side tables CONTAIN may change from frame to frame (SAH_GENERATION_TRACKED or 0 for copies the library keeps: a non-zero
 * caller-kept change counter in a fixed descriptor could not change). */
#define SAH_CHAIN_CAPTURE (1u << 1)
int sah_chain_create(sah_ctx* ctx, const sah_chain_plan* plan, const sah_chain_frame frames[2], uint32_t tonemap_flags, uint32_t chain_flags,
                     void* work_stream, void* reduce_stream, void* post_stream, sah_chain** out);
/* lighting_begin / lighting_end: optional hipEvent_t recorded on the work stream around the frame's sah_lighting calls (NULL: none) */
int sah_chain_submit(sah_chain* chain, void* lighting_begin, void* lighting_end);
/* Completes every submitted frame; the work stream then waits for the last gathers and B halves. */
int sah_chain_flush(sah_chain* chain);
/* frames submitted / frames whose B half has been enqueued */
int sah_chain_counts(const sah_chain* chain, uint64_t* submitted, uint64_t* finished);
void sah_chain_destroy(sah_chain* chain);

#ifdef __cplusplus
}
#endif
#endif /* SAH_HIP_H */
