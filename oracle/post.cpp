// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// CPU restatement of the post chain and the point-light extension (a9).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../include/sah_hip.h"
#include "brdf.hpp"
#include "gi.hpp"
#include "image.hpp"
#include "math.hpp"

namespace orc {

// ---- a13 "Copy scene": RenderCore/shaders/util/copy_with_sampler.frag.slang:9-12 ------------------
// image.Sample(SV_Position.xy * inv_resolution), linear sampler with default (REPEAT) addressing
// (scene_renderer.cpp:74-79); output RGBA16F, no blending.
static void copy_scene_pixel(const Image& src, uint32_t ow, uint32_t oh, int x, int y, uint16_t out[4]) {
    F inv_w = F(1.0f) / F((float)ow), inv_h = F(1.0f) / F((float)oh);  // scene_renderer.cpp:514-516: 1.f / resolution
    F u = F((float)x + 0.5f) * inv_w, v = F((float)y + 0.5f) * inv_h;
    Texel t = sample_bilinear(src, u.v, v.v, 0, ADDR_REPEAT);
    for (int i = 0; i < 4; i++) out[i] = f32_to_f16(t.c[i]);
}

// ---- a7 bloom downsample: RenderCore/shaders/postprocessing/bloom_downsample.comp:16-52 ------------
struct C3 {
    F r, g, b;
};
static inline C3 operator+(C3 a, C3 b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
static inline C3 operator*(C3 a, F s) { return {a.r * s, a.g * s, a.b * s}; }

static inline C3 tap(const Image& src, F u, F v) {
    Texel t = sample_bilinear(src, u.v, v.v, 0, ADDR_CLAMP_TO_EDGE);  // bloomer.cpp:25-35
    return {F(t.c[0]), F(t.c[1]), F(t.c[2])};
}

// :16-25 — o = inv_source_size.xyxy * vec2(-1, 1).xxyy = (-ix, -iy, +ix, +iy)
static inline C3 box_blur(const Image& src, F u, F v, F ix, F iy) {
    F ox = ix * F(-1.0f), oy = iy * F(-1.0f), oz = ix * F(1.0f), ow = iy * F(1.0f);
    C3 s = tap(src, u + ox, v + oy) + tap(src, u + oz, v + oy) + tap(src, u + ox, v + ow) + tap(src, u + oz, v + ow);
    return s * F(0.25f);
}

// :27-36
static inline C3 cod_blur(const Image& src, F u, F v, F ix, F iy) {
    F ox = ix * F(-1.0f), oy = iy * F(-1.0f), oz = ix * F(1.0f), ow = iy * F(1.0f);
    C3 s = box_blur(src, u, v, ix, iy) * F(0.5f) + box_blur(src, u + ox, v + oy, ix, iy) * F(0.125f) +
           box_blur(src, u + oz, v + oy, ix, iy) * F(0.125f) + box_blur(src, u + ox, v + ow, ix, iy) * F(0.125f) +
           box_blur(src, u + oz, v + ow, ix, iy) * F(0.125f);
    return s;
}

static void bloom_downsample_pixel(const Image& src, uint32_t dw, uint32_t dh, int x, int y, uint16_t out[4]) {
    F ix = F(1.0f) / F((float)src.width), iy = F(1.0f) / F((float)src.height);  // :39-40
    F u = (F((float)x) + F(0.5f)) / F((float)dw), v = (F((float)y) + F(0.5f)) / F((float)dh);  // :48
    C3 c = cod_blur(src, u, v, ix, iy);
    // imageStore(vec4(bloomed, 0)) into an image that is really RGBA16F (format-mismatch quirk)
    out[0] = f32_to_f16(c.r.v);
    out[1] = f32_to_f16(c.g.v);
    out[2] = f32_to_f16(c.b.v);
    out[3] = 0;
}

// ---- a8 tonemap composite: RenderCore/shaders/ui/scene_upsample.frag:20-72 ------------------------
static inline C3 btap(const Image& mip, F u, F v) {
    Texel t = sample_bilinear(mip, u.v, v.v, 0, ADDR_CLAMP_TO_EDGE);  // ui_phase.cpp:26-36
    return {F(t.c[0]), F(t.c[1]), F(t.c[2])};
}

// :20-39
static inline C3 tent_blur(const Image& mip, F u, F v) {
    F ix = F(1.0f) / F((float)mip.width), iy = F(1.0f) / F((float)mip.height);
    F ox = ix * F(-1.0f), oy = iy * F(-1.0f), oz = ix * F(1.0f), ow = iy * F(1.0f);  // o = (-ix, -iy, ix, iy)
    C3 s = btap(mip, u, v) * F(4.0f) + btap(mip, u + ox, v + F(0.f)) * F(2.0f) + btap(mip, u + oy, v + F(0.f)) * F(2.0f) +
           btap(mip, u + F(0.f), v + oz) * F(2.0f) + btap(mip, u + F(0.f), v + ow) * F(2.0f) + btap(mip, u + ox, v + oy) * F(1.0f) +
           btap(mip, u + oz, v + oy) * F(1.0f) + btap(mip, u + ox, v + ow) * F(1.0f) + btap(mip, u + oz, v + ow) * F(1.0f);
    return {s.r / F(16.f), s.g / F(16.f), s.b / F(16.f)};
}

static inline F cr_powf(F x, double e) {
    // correctly rounded fp32 pow; pow(negative, non-integer) = NaN as in GLSL (undefined) -> NaN here
    return F((float)std::pow((double)x.v, e));
}

static void tonemap_pixel(const Image& scene, const sah_mipchain& bloom, uint32_t ow, uint32_t oh, int x, int y, uint8_t out[4]) {
    // fullscreen.vert:9-21 interpolation: u = (x + 0.5) / W, v = 1 - (y + 0.5) / H  (vertical flip, SURVEY note 1)
    F u = (F((float)x) + F(0.5f)) / F((float)ow);
    F v = F(1.0f) - (F((float)y) + F(0.5f)) / F((float)oh);
    C3 bloom_sum = {F(0.f), F(0.f), F(0.f)};
    for (uint32_t m = 0; m < 6; m++) {  // :45-49, six mips hard-coded
        if (m >= bloom.num_mips) break;
        C3 b = tent_blur(img2d(bloom.mips[m]), u, v);
        bloom_sum = bloom_sum + b;
    }
    Texel st = sample_bilinear(scene, u.v, v.v, 0, ADDR_CLAMP_TO_EDGE);
    C3 c = {F(st.c[0]) + bloom_sum.r * F(0.014159f), F(st.c[1]) + bloom_sum.g * F(0.014159f), F(st.c[2]) + bloom_sum.b * F(0.014159f)};
    F luma = c.r * F(0.2126f) + c.g * F(0.7152f) + c.b * F(0.0722f);  // :54
    F factor = luma / (luma + F(1.f));
    C3 mapped = c * factor;
    const double e = (double)(1.f / 2.2f);  // pow(mapped, vec3(1.f / 2.2f)) — the exponent is an fp32 constant
    F rgb[3] = {cr_powf(mapped.r, e), cr_powf(mapped.g, e), cr_powf(mapped.b, e)};
    // written to an sRGB swapchain: the hardware applies the OETF on top (double-gamma quirk), then UNORM8
    for (int i = 0; i < 3; i++) out[i] = float_to_unorm8(linear_to_srgb_f(rgb[i].v));
    out[3] = 255;
}

// ---- a9 extension: point lights (spec: DESIGN.md §a9; BRDF = brdf.glsl) ----------------------------
bool point_lights_frag(const sah_lighting_desc& d, int x, int y, const GBufferTexels& g, F out[4]) {
    if (g.depth == 0.f) return false;
    const sah_view_data& view = *d.view;
    Surface<F> s;
    s.base_color = {F(g.color.c[0]), F(g.color.c[1]), F(g.color.c[2])};
    s.normal = normalize(F3{F(g.normal.c[0]), F(g.normal.c[1]), F(g.normal.c[2])});
    s.roughness = F(g.data.c[1]);
    s.metalness = F(g.data.c[2]);
    F3 vs = viewspace_position_glsl(view, x, y, g.depth);
    F4 ws4 = mul(mat(view.inverse_view), F4{vs.x, vs.y, vs.z, F(1.0f)});
    F3 ws = {ws4.x, ws4.y, ws4.z};
    F3 view_position = {F(-view.view[12]), F(-view.view[13]), F(-view.view[14])};
    F3 V = normalize(ws - view_position);
    F3 sum = F3(F(0.f));
    for (uint32_t i = 0; i < d.lights->count; i++) {
        const sah_point_light& pl = d.lights->lights[i];
        F3 lv = F3{F(pl.position[0]), F(pl.position[1]), F(pl.position[2])} - ws;
        F d2 = dot(lv, lv);
        F3 L = lv * inversesqrt(d2);
        F dist = nsqrt(d2);
        F ndotl = nclamp(dot(s.normal, L), F(0.f), F(1.f));
        F xr = dist / F(pl.radius);
        F x2 = xr * xr;
        F x4 = x2 * x2;
        F w = nclamp(F(1.f) - x4, F(0.f), F(1.f));
        F att = (w * w) / nmax(d2, F(1e-4f));
        F3 b = brdf(s, L, V);
        F3 c = ndotl * b * F3{F(pl.color[0]), F(pl.color[1]), F(pl.color[2])} * (F(pl.intensity) * att);
        if (any_nan(c)) c = F3(F(0.f));
        sum = sum + c;
    }
    const F exposure = F(0.00031415927f);
    out[0] = sum.x * exposure;
    out[1] = sum.y * exposure;
    out[2] = sum.z * exposure;
    out[3] = F(1.0f);
    return true;
}

}  // namespace orc

using namespace orc;

extern "C" int orc_copy_scene(const sah_plane* lit, const sah_plane* out) {
    if (!lit || !out) return SAH_ERR_INVALID_ARGUMENT;
    Image src = img2d(*lit);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < (int)out->height; y++) {
        for (uint32_t x = 0; x < out->width; x++) {
            uint16_t px[4];
            copy_scene_pixel(src, out->width, out->height, (int)x, y, px);
            std::memcpy((uint8_t*)out->ptr + (size_t)y * out->row_pitch_bytes + (size_t)x * 8, px, 8);
        }
    }
    return SAH_OK;
}

// One downsample step (src -> dst), as each dispatch of Bloomer::fill_bloom_tex does (bloomer.cpp:50-151).
extern "C" int orc_bloom_downsample(const sah_plane* src_p, const sah_plane* dst) {
    if (!src_p || !dst) return SAH_ERR_INVALID_ARGUMENT;
    Image src = img2d(*src_p);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < (int)dst->height; y++) {
        for (uint32_t x = 0; x < dst->width; x++) {
            uint16_t px[4];
            bloom_downsample_pixel(src, dst->width, dst->height, (int)x, y, px);
            std::memcpy((uint8_t*)dst->ptr + (size_t)y * dst->row_pitch_bytes + (size_t)x * 8, px, 8);
        }
    }
    return SAH_OK;
}

extern "C" int orc_bloom(const sah_plane* scene, const sah_mipchain* bloom) {
    if (!scene || !bloom || bloom->num_mips == 0 || bloom->num_mips > SAH_MAX_BLOOM_MIPS) return SAH_ERR_INVALID_ARGUMENT;
    int rc = orc_bloom_downsample(scene, &bloom->mips[0]);
    for (uint32_t m = 1; m < bloom->num_mips && rc == SAH_OK; m++) rc = orc_bloom_downsample(&bloom->mips[m - 1], &bloom->mips[m]);
    return rc;
}

extern "C" int orc_tonemap(const sah_plane* scene_p, const sah_mipchain* bloom, const sah_plane* out, uint32_t row_begin, uint32_t row_end) {
    if (!scene_p || !bloom || !out) return SAH_ERR_INVALID_ARGUMENT;
    if (row_begin == 0 && row_end == 0) row_end = out->height;
    if (row_end > out->height || row_begin > row_end) return SAH_ERR_INVALID_ARGUMENT;
    Image scene = img2d(*scene_p);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = (int)row_begin; y < (int)row_end; y++) {
        for (uint32_t x = 0; x < out->width; x++) {
            uint8_t px[4];
            tonemap_pixel(scene, *bloom, out->width, out->height, (int)x, y, px);
            std::memcpy((uint8_t*)out->ptr + (size_t)y * out->row_pitch_bytes + (size_t)x * 4, px, 4);
        }
    }
    return SAH_OK;
}
