// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference has no tests, golden images or
// runnable build in this environment; this file is a CPU restatement that follows
// the reference shader text line by line, quirks included.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may call it.
//
// Restates the "Lighting" pass: RenderCore/render/phase/lighting_phase.cpp:99-134 and the shader bodies
// it draws, composed per pixel with the fixed-function blend/round sequence.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../include/sah_hip.h"
#include "brdf.hpp"
#include "gi.hpp"
#include "image.hpp"
#include "math.hpp"
#include "sky.hpp"

namespace orc {

Image img2d(const sah_plane& p) { return Image{(const uint8_t*)p.ptr, p.width, p.height, 1, p.row_pitch_bytes, 0, p.format}; }
Image img3d(const sah_volume& v) {
    return Image{(const uint8_t*)v.ptr, v.width, v.height, v.depth, v.row_pitch_bytes, v.slice_pitch_bytes, v.format};
}
M4 mat(const float* m) { M4 r; std::memcpy(r.m, m, 64); return r; }

// directional_light.frag:45-53, gi/lpv/overlay.frag:43-51 — gl_FragCoord already carries the +0.5, so the
// texcoord is (x + 1) / W (quirk, reproduced).
F3 viewspace_position_glsl(const sah_view_data& view, int x, int y, float depth) {
    F fx = F((float)x + 0.5f) + F(0.5f), fy = F((float)y + 0.5f) + F(0.5f);
    F tx = fx / F(view.render_resolution[0]), ty = fy / F(view.render_resolution[1]);
    F4 ndc = {tx * F(2.0f) - F(1.0f), ty * F(2.0f) - F(1.0f), F(depth), F(1.0f)};
    F4 vs = mul(mat(view.inverse_projection), ndc);
    return {vs.x / vs.w, vs.y / vs.w, vs.z / vs.w};
}

// directional_light.rt.slang:39-48, gi/cache/overlay.frag.slang:35-44 — correct pixel centre.
F3 worldspace_location_slang(const sah_view_data& view, int x, int y, float depth) {
    F tx = (F((float)x) + F(0.5f)) / F(view.render_resolution[0]);
    F ty = (F((float)y) + F(0.5f)) / F(view.render_resolution[1]);
    F4 ndc = {tx * F(2.0f) - F(1.0f), ty * F(2.0f) - F(1.0f), F(depth), F(1.0f)};
    F4 vs = mul(mat(view.inverse_projection), ndc);
    vs = {vs.x / vs.w, vs.y / vs.w, vs.z / vs.w, vs.w / vs.w};
    F4 ws = mul(mat(view.inverse_view), vs);
    return {ws.x, ws.y, ws.z};
}

static GBufferTexels fetch_gbuffer(const sah_gbuffer& g, int x, int y) {
    GBufferTexels t;
    t.depth = load_texel(img2d(g.depth), x, y, 0).c[0];
    t.color = load_texel(img2d(g.color), x, y, 0);
    t.normal = load_texel(img2d(g.normals), x, y, 0);
    t.data = load_texel(img2d(g.data), x, y, 0);
    t.emission = load_texel(img2d(g.emission), x, y, 0);
    return t;
}

// ---- a1: directional_light.frag:96-149 -------------------------------------------------------------

// directional_light.frag:62-78
static F shadow_factor(const sah_sun_light_constants& sun, const sah_volume* shadowmap, F3 ws, uint32_t cascade, F bias) {
    // biasMat * cascade_matrices[c] is a matrix product evaluated first (GLSL is left-associative).
    const float* Mc = sun.cascade_matrices[cascade];
    M4 P;
    for (int col = 0; col < 4; col++) {
        const float* c = Mc + col * 4;
        // biasMat columns: (0.5,0,0,0) (0,0.5,0,0) (0,0,1,0) (0.5,0.5,0,1)
        P.m[col * 4 + 0] = ((F(0.5f) * F(c[0]) + F(0.0f) * F(c[1])) + F(0.0f) * F(c[2]) + F(0.5f) * F(c[3])).v;
        P.m[col * 4 + 1] = ((F(0.0f) * F(c[0]) + F(0.5f) * F(c[1])) + F(0.0f) * F(c[2]) + F(0.5f) * F(c[3])).v;
        P.m[col * 4 + 2] = ((F(0.0f) * F(c[0]) + F(0.0f) * F(c[1])) + F(1.0f) * F(c[2]) + F(0.0f) * F(c[3])).v;
        P.m[col * 4 + 3] = ((F(0.0f) * F(c[0]) + F(0.0f) * F(c[1])) + F(0.0f) * F(c[2]) + F(1.0f) * F(c[3])).v;
    }
    F4 sp = mul(P, F4{ws.x, ws.y, ws.z, F(1.0f)});
    sp = {sp.x / sp.w, sp.y / sp.w, sp.z / sp.w, sp.w / sp.w};
    if (sp.x.v < 0.f || sp.y.v < 0.f || sp.z.v < 0.f || sp.x.v > 1.f || sp.y.v > 1.f || sp.z.v > 1.f) return F(1.0f);
    if (!shadowmap || !shadowmap->ptr) return F(1.0f);
    if (std::isnan(sp.x.v) || std::isnan(sp.y.v)) return F(NAN);
    float ref = (sp.z - bias).v;
    // Vulkan: D_ref is clamped to [0,1] for fixed-point depth formats before the comparison.
    if (shadowmap->format == FMT_D16_UNORM) ref = ref < 0.f ? 0.f : (ref > 1.f ? 1.f : ref);
    // sampler: linear, CLAMP_TO_EDGE, compare LESS (directional_light.cpp:338-352)
    return F(sample_shadow_pcf(img3d(*shadowmap), sp.x.v, sp.y.v, (int)cascade, ref, ADDR_CLAMP_TO_EDGE));
}

// directional_light.frag:80-94
static F sample_csm(const sah_sun_light_constants& sun, const sah_volume* shadowmap, F3 ws, F viewspace_depth, F ndotl) {
    uint32_t cascade = 0;
    for (uint32_t i = 0; i < 4; i++) {
        if (viewspace_depth.v < sun.data[i][0]) cascade = i + 1;
    }
    if (cascade > 3) return F(0.0f);  // :89-91 (the out-of-bounds matrix read before it has no effect)
    F bias = F(0.0005f) * nsqrt(F(1.0f) - ndotl * ndotl) / ndotl;
    return shadow_factor(sun, shadowmap, ws, cascade, bias);
}

// Returns false on `discard`.
static bool sun_frag(const sah_lighting_desc& d, int x, int y, const GBufferTexels& g, F out[4]) {
    if (g.depth == 0.f) return false;
    const sah_view_data& view = *d.view;
    const sah_sun_light_constants& sun = *d.sun;
    F3 base_color = {F(g.color.c[0]), F(g.color.c[1]), F(g.color.c[2])};
    F3 normal = normalize(F3{F(g.normal.c[0]), F(g.normal.c[1]), F(g.normal.c[2])});
    F3 vs = viewspace_position_glsl(view, x, y, g.depth);
    F4 ws4 = mul(mat(view.inverse_view), F4{vs.x, vs.y, vs.z, F(1.0f)});
    F3 ws = {ws4.x, ws4.y, ws4.z};
    F3 view_position = {F(-view.view[12]), F(-view.view[13]), F(-view.view[14])};  // :112 quirk
    F3 V = normalize(ws - view_position);
    F3 L = normalize(F3{F(-sun.direction_and_tan_size[0]), F(-sun.direction_and_tan_size[1]), F(-sun.direction_and_tan_size[2])});
    Surface<F> s;
    s.base_color = base_color;
    s.normal = normal;
    s.roughness = F(g.data.c[1]);
    s.metalness = F(g.data.c[2]);
    F ndotl = nclamp(dot(normal, L), F(0.f), F(1.f));
    F shadow = F(1.0f);
    if (ndotl.v > 0.f) {
        if (sun.shadow_mode == SAH_SHADOW_MODE_CSM) shadow = sample_csm(sun, d.shadowmap, ws, vs.z, ndotl);
    }
    F3 b = brdf(s, L, V);
    F3 direct = ndotl * b * F3{F(sun.color[0]), F(sun.color[1]), F(sun.color[2])} * shadow;
    if (any_nan(direct)) direct = F3(F(0.f));
    const F exposure = F(0.00031415927f);
    out[0] = direct.x * exposure;
    out[1] = direct.y * exposure;
    out[2] = direct.z * exposure;
    out[3] = F(1.0f);
    return true;
}

// ---- a1b: directional_light.rt.slang:58-139 (the ray query itself is replaced by the shadow_mask plane)
static bool sun_rt(const sah_lighting_desc& d, int x, int y, const GBufferTexels& g, float add[3]) {
    if (g.depth == 0.f) return false;
    const sah_view_data& view = *d.view;
    const sah_sun_light_constants& sun = *d.sun;
    Surface<H> s;
    s.base_color = {H(g.color.c[0]), H(g.color.c[1]), H(g.color.c[2])};  // Texture2D<half4>
    s.normal = normalize(H3{H(g.normal.c[0]), H(g.normal.c[1]), H(g.normal.c[2])});
    s.roughness = H(g.data.c[1]);
    s.metalness = H(g.data.c[2]);
    F3 location = worldspace_location_slang(view, x, y, g.depth);
    F3 L = normalize(F3{F(-sun.direction_and_tan_size[0]), F(-sun.direction_and_tan_size[1]), F(-sun.direction_and_tan_size[2])});
    H ndotl = H(nclamp(dot(L, to_f(s.normal)), F(0.f), F(1.f)).v);
    F3 view_position = {F(-view.view[12]), F(-view.view[13]), F(-view.view[14])};
    H3 V = to_h(normalize(location - view_position));
    H3 b = brdf(s, to_h(L), V);
    H3 nb = ndotl * b;
    F3 radiance = to_f(nb) * F3{F(sun.color[0]), F(sun.color[1]), F(sun.color[2])};
    if (ndotl.v > 0.f) {
        float mask = 1.0f;
        if (d.shadow_mask && d.shadow_mask->ptr) mask = load_texel(img2d(*d.shadow_mask), x, y, 0).c[0];
        radiance = radiance * F(mask);  // :125 `sun_radiance *= shadow / num_shadow_samples`
    }
    const F exposure = F(0.00031415927f);
    add[0] = (radiance.x * exposure).v;
    add[1] = (radiance.y * exposure).v;
    add[2] = (radiance.z * exposure).v;
    return true;
}

// ---- a9 (extension): point lights -----------------------------------------------------
bool point_lights_frag(const sah_lighting_desc& d, int x, int y, const GBufferTexels& g, F out[4]);

// ---- a0: per-pixel composition of the Lighting pass ----------------------------------------------
static inline H blend_add(H dst, F src) { return H(dst.v + src.v); }

static void lighting_pixel(const sah_lighting_desc& d, int x, int y, uint16_t out[4]) {
    GBufferTexels g = fetch_gbuffer(*d.gbuffer, x, y);
    H lit[4] = {H(0.f), H(0.f), H(0.f), H(0.f)};  // lighting_phase.cpp:106 LOAD_OP_CLEAR, clear value 0
    const uint32_t mode = d.sun ? d.sun->shadow_mode : SAH_SHADOW_MODE_OFF;

    // (2) sun, CSM mode only — lighting_phase.cpp:111-113; blend directional_light.cpp:70-80
    if (mode == SAH_SHADOW_MODE_CSM) {
        F s[4];
        if (sun_frag(d, x, y, g, s)) {
            if (d.flags & SAH_LIGHTING_QUIRK_SUN_BLEND) {
                // src factor SRC_COLOR, dst factor DST_COLOR; alpha factors default to ZERO
                for (int i = 0; i < 3; i++) lit[i] = H((s[i] * s[i] + F(lit[i].v) * F(lit[i].v)).v);
                lit[3] = H((s[3] * F(0.f) + F(lit[3].v) * F(0.f)).v);
            } else {
                for (int i = 0; i < 4; i++) lit[i] = blend_add(lit[i], s[i]);
            }
        }
    }
    // (2b) extension: point lights, additive
    if (d.lights && d.lights->count) {
        F s[4];
        if (point_lights_frag(d, x, y, g, s)) {
            for (int i = 0; i < 4; i++) lit[i] = blend_add(lit[i], s[i]);
        }
    }
    // (3) GI overlay, additive on rgb and a — lighting_phase.cpp:115-117
    if (d.gi && d.gi->kind != SAH_GI_NONE) {
        F s[4];
        bool drawn = false;
        if (d.gi->kind == SAH_GI_LPV) drawn = gi_lpv_frag(d, x, y, g.depth, g.color, g.normal, g.data, s);
        else if (d.gi->kind == SAH_GI_CACHE) drawn = gi_cache_frag(d, x, y, g.depth, g.color, g.normal, g.data, s);
        else if (d.gi->kind == SAH_GI_RTGI) drawn = gi_rtgi_frag(d, x, y, g.depth, g.color, g.normal, g.data, s);
        if (drawn) {
            for (int i = 0; i < 4; i++) lit[i] = blend_add(lit[i], s[i]);
        }
    }
    // (4) emissive, additive, no depth discard — emissive.frag:15-22
    {
        const F e = F(3.1415927f);
        lit[0] = blend_add(lit[0], F(g.emission.c[0]) * e);
        lit[1] = blend_add(lit[1], F(g.emission.c[1]) * e);
        lit[2] = blend_add(lit[2], F(g.emission.c[2]) * e);
        lit[3] = blend_add(lit[3], F(1.0f));
    }
    // (5) sky overwrites depth == 0 pixels — procedural_sky.cpp:151-172 (blending disabled)
    if (d.sky && g.depth == 0.f) {
        H sky[4];
        sky_frag(*d.view, *d.sun, *d.sky, x, y, sky);
        for (int i = 0; i < 4; i++) lit[i] = sky[i];
    }
    // (6) RT-mode sun: image load/store read-modify-write after the pass — lighting_phase.cpp:131-133
    if (mode == SAH_SHADOW_MODE_RT) {
        float add[3];
        if (sun_rt(d, x, y, g, add)) {
            for (int i = 0; i < 3; i++) lit[i] = H(lit[i].v + add[i]);
        }
    }
    for (int i = 0; i < 4; i++) out[i] = f32_to_f16(lit[i].v);
}

}  // namespace orc

extern "C" int orc_lighting(const sah_lighting_desc* d) {
    using namespace orc;
    if (!d || !d->gbuffer || !d->lit || !d->view) return SAH_ERR_INVALID_ARGUMENT;
    const uint32_t W = d->lit->width, Hh = d->lit->height;
    uint32_t r0 = d->row_begin, r1 = d->row_end;
    if (r0 == 0 && r1 == 0) r1 = Hh;
    if (r1 > Hh || r0 > r1) return SAH_ERR_INVALID_ARGUMENT;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = (int)r0; y < (int)r1; y++) {
        uint8_t* row = (uint8_t*)d->lit->ptr + (size_t)y * d->lit->row_pitch_bytes;
        for (uint32_t x = 0; x < W; x++) {
            uint16_t px[4];
            lighting_pixel(*d, (int)x, y, px);
            std::memcpy(row + (size_t)x * 8, px, 8);
        }
    }
    return SAH_OK;
}

// exports for the known-answer tests (SURVEY.md §8-c fixture iii: the half-pixel offset quirk)
extern "C" void orc_viewspace_position_glsl(const sah_view_data* view, int x, int y, float depth, float* out3) {
    const orc::F3 p = orc::viewspace_position_glsl(*view, x, y, depth);
    out3[0] = p.x.v; out3[1] = p.y.v; out3[2] = p.z.v;
}
extern "C" void orc_worldspace_location_slang(const sah_view_data* view, int x, int y, float depth, float* out3) {
    const orc::F3 p = orc::worldspace_location_slang(*view, x, y, depth);
    out3[0] = p.x.v; out3[1] = p.y.v; out3[2] = p.z.v;
}
