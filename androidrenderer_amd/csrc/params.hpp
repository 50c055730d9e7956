// Kernel-argument blocks: the reference's UBOs / push constants, flattened and pre-digested on the host
// (uniform sub-expressions are evaluated once on the CPU with the same individually rounded fp32 operators
// the per-pixel code would use — IEEE makes that bit-identical).
#pragma once
#include <stdint.h>

namespace sah {

struct PlaneArg {
    const uint8_t* ptr;
    uint32_t pitch;
};
struct VolumeArg {
    const uint8_t* ptr;
    uint32_t width, height, depth;
    uint32_t row_pitch, slice_pitch;
};

struct LpvArgs {
    VolumeArg red, green, blue;
    float world_to_cascade[4][16];
    uint32_t num_cascades;
    float num_cascades_f;
    float exposure;
};

struct ProbeAtlasArgs {  // sah_probe_atlases (probes.hip)
    VolumeArg rtgi, light_cache, depth, average, validity;
};

struct CacheArgs {
    VolumeArg irradiance, depth, validity;
    float cascade_min[4][3];
    float cascade_max[4][3];  // min + (32,8,32) * spacing
    float spacing[4];
    float inv_spacing[4];   // 1 / spacing
    uint32_t spacing_pow2;  // every spacing is a normal power of two: x / spacing == x * inv_spacing, bit for bit (an exact scaling either way)
    uint32_t probe_size[2];
    float inv_tex[2];  // RN(1 / (32 * (probe_size + 2))): the irradiance atlas extent in texels, for div_const()
    uint32_t debug_mode;
    uint32_t hot_ok;  // atlases < 4 GiB, probe grid <= 64 per axis, probe texel counts <= 30: sample_cascade_fast() applies
    // hot_ok only: the irradiance atlas widened to float4 per texel (r, g, b, 0), same layout with every pitch x 4 (k_probe_irr_unpack)
    const uint8_t* irr32;
};

struct RtgiArgs {
    PlaneArg ray_buffer, ray_irradiance, noise;
    uint32_t noise_w, noise_h;
    uint32_t num_extra_rays;
    float extra_ray_radius;
};

struct CsmArgs {
    VolumeArg shadowmap;
    uint32_t is_d16;
    uint32_t d16_recip_ok;  // host verified: the 3-flop reciprocal sequence equals v / 65535 for every 16-bit v
    float d16_recip;        // RN(1 / 65535)
    float splits[4];
    float biased[4][16];  // biasMat * cascade_matrices[i]
};

struct SkyArgs {
    PlaneArg transmittance, sky_view;
    uint32_t t_w, t_h, s_w, s_h;
    float sun_dir[3];       // -normalize(direction)
    float height;           // length(viewPos)
    float up_y;             // viewPos.y / height
    float view_pos_y;       // groundRadiusMM + 0.0002
    float horizon_angle;    // safeacos(sqrt(h*h - g*g) / h)
    float azimuth_limit;    // 0.5 * PI - .0001
    float min_sun_cos;      // cos(0.53 * PI / 180)
    float right[3];         // cross(sunDir, up)
    float forward[3];       // cross(up, right)
    float tlut_rgb[3];      // getValFromTLUT(viewPos, sunDir): uniform over the frame
    float smooth_e0;        // 0.002h as float
    uint32_t enabled;
};

// Per-context device state shared by consecutive Lighting calls (double-buffered by call parity so that no memset
// is needed between calls: the fix-up kernel of call k zeroes the slots call k+1 will use).
struct FrameState {
    // "some wave of the fast kernel has listed a pixel for the fix-up kernel": raised (a plain store of 1) by such a wave, read first thing by
    // every workgroup of k_lighting_fixup — the frames a renderer produces list nothing, and the fix-up launch is then a few hundred
    // workgroups that return at once instead of a scan of 32,400 segment counts.  Two words, used in turn by consecutive Lighting calls
    // (FastArgs::hint_slot): the fast kernel of a call clears the word of the NEXT call (which nobody reads or raises before that call),
    // so no memset, no atomic and no "last workgroup" ticket is needed (2,025 tickets on one address cost 0.1 ms).  A captured launch that
    // is replayed by itself keeps its slot: its word is then never cleared, which costs the early exit, not correctness.
    uint32_t deferred_hint[2];
    uint32_t unused;
    // 1 when the gather copy of the LPV volumes holds an inf / NaN texel: cleared (a 4-byte memset in front of the kernel) and raised by whatever
    // writes the copy — k_lpv_pack, or the emitting step of the propagation — and read by the Lighting kernels; a copy that is kept over several
    // Lighting calls keeps its verdict.  (Rounds 2-4 compared a tag with the copy's serial number, a KERNEL ARGUMENT: a captured launch replays
    // its arguments, so the verdict now lives in memory only.)
    uint32_t nonfinite;
};

// What the fast kernel may assume, established by the host (api.cpp: detect_fast_path):
//   inverse_projection separable:  vs = ((p0*X)+p12, (p5*Y)+p13, (p10*D)+p14, (p11*D)+p15)
//   inverse_view affine (row 3 = 0,0,0,1), LPV world_to_cascade = scale + translate, CSM matrices affine.
struct FastArgs {
    float p0, p12, p5, p13, p10, p14, p11, p15;
    float lpv_s[4][3], lpv_t[4][3];
    float inv_ncasc;
    uint32_t ncasc_pow2;
    uint32_t pos_div_nr;  // view-space position quotients may use the shared-reciprocal divide (lighting_fast.hpp)
    uint32_t sky_enabled;
    uint32_t sky_ratio;  // surface workgroups per sky workgroup of k_lighting_fast's grid (lighting.hip); >= 1
    uint32_t sky_first;  // 0: every (sky_ratio + 1)-th workgroup of the grid is a sky workgroup; else the number of sky workgroups, which lead the grid
    uint32_t row_magic;  // floor(2^32 / groups per row) + 1 when mulhi(gid, row_magic) == gid / groups_per_row for every thread of the call, else 0
    uint32_t repack;       // 1: this call rebuilds the gather copy first
    const float* colx_tab; // per-column view-space x numerators (k_colx_table): [0, width) the GLSL flavour, [colx_stride, ..) the Slang one; or null
    uint32_t colx_stride;
    uint32_t rowy_stride;  // per-row numerators of the view-space y behind the two column tables: [2 * colx_stride, ..) GLSL, [2 * colx_stride + rowy_stride, ..) Slang
    FrameState* state;
    // Deferred pixels, without atomics: the wave that shades thread groups [64 s, 64 s + 64) owns segment s — kSegSize(PPT) byte codes
    // (lane * PPT + pixel) at seg_list + s * seg_stride — general pixels from the front, sky pixels (depth == 0) from the back — and
    // their numbers in seg_count[s] / seg_count[num_segments + s], which it always writes (0 included), so nothing has to be cleared
    // between calls.  The fix-up and sky kernels walk 16 segments per workgroup.
    uint8_t* seg_list;
    uint16_t* seg_count;
    uint32_t num_segments, seg_stride;
    uint32_t hint_slot;  // which word of FrameState::deferred_hint this call uses (0 / 1, alternating per Lighting call of the context)
    // LPV gather copy (k_lpv_pack): texel (x,y,z) of the three volumes interleaved as 24 bytes {R[4], G[4], B[4]} (fp16) at
    // ((z+2) * pk_slice_pitch + (y+2) * pk_row_pitch + (x+2) * 24), inside a two-texel border of zeros (= CLAMP_TO_BORDER)
    const uint8_t* lpv_packed;
    uint32_t pk_row_pitch, pk_slice_pitch;
    uint32_t lpv_fast;  // tiled kernel: the LPV part of the fast-path check holds too and the gather copy is there (lpv_s / lpv_t, lpv_packed, state valid)
};
// (an fp32 copy — 48-byte texels, plain v_fma_f32 taps — was measured and loses 67 %: profiles/r3_lpv_pack32_experiment.txt)
constexpr uint32_t kLpvPackTexel = 24, kLpvPackBorder = 2;
struct LpvPackEmit {  // the gather copy as the emitting propagation step writes it (lpv.hip)
    uint8_t* packed;
    uint32_t row_pitch, slice_pitch;
    FrameState* state;
};

struct LightingArgs {
    PlaneArg color, normals, data, emission, depth, ao, shadow_mask, lit;
    uint32_t width, height, row_begin, row_end;
    uint32_t flags;
    float res[2];
    float inv_proj[16];
    float inv_view[16];
    float view_pos[3];   // -view[3].xyz
    float sun_L[3];      // normalize(-direction)
    float sun_color[3];
    uint32_t has_ao, has_mask;
    const float* luts;   // [0..255] sRGB8->linear, [256..511] UNORM8->float
    const void* lights;  // sah_point_light[count]
    uint32_t num_lights;
};

}  // namespace sah
