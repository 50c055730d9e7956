// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// Small exports used by the unit / known-answer tests (codecs, BRDF points, SH, octahedral).
#include <cstdint>

#include "../include/sah_hip.h"
#include "brdf.hpp"
#include "codec.hpp"
#include "image.hpp"
#include "math.hpp"

using namespace orc;

extern "C" {

uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
float orc_srgb8_to_linear(uint8_t v) { return srgb8_to_linear(v); }
uint8_t orc_linear_to_srgb8(float c) { return float_to_unorm8(linear_to_srgb_f(c)); }
void orc_r11g11b10_decode(uint32_t p, float* out3) { r11g11b10_decode(p, out3); }
uint32_t orc_r11g11b10_encode(const float* in3) { return r11g11b10_encode(in3); }

// brdf() at one point, fp32 flavour (brdf.glsl) and fp16 flavour (brdf.slangi). Vectors are used as given.
void orc_brdf_f32(const float* base_color, const float* normal, float roughness, float metalness, const float* l, const float* v,
                  float* out3) {
    Surface<F> s;
    s.base_color = {F(base_color[0]), F(base_color[1]), F(base_color[2])};
    s.normal = {F(normal[0]), F(normal[1]), F(normal[2])};
    s.roughness = F(roughness);
    s.metalness = F(metalness);
    F3 r = brdf(s, F3{F(l[0]), F(l[1]), F(l[2])}, F3{F(v[0]), F(v[1]), F(v[2])});
    out3[0] = r.x.v; out3[1] = r.y.v; out3[2] = r.z.v;
}
void orc_brdf_f16(const float* base_color, const float* normal, float roughness, float metalness, const float* l, const float* v,
                  float* out3) {
    Surface<H> s;
    s.base_color = {H(base_color[0]), H(base_color[1]), H(base_color[2])};
    s.normal = {H(normal[0]), H(normal[1]), H(normal[2])};
    s.roughness = H(roughness);
    s.metalness = H(metalness);
    H3 r = brdf(s, H3{H(l[0]), H(l[1]), H(l[2])}, H3{H(v[0]), H(v[1]), H(v[2])});
    out3[0] = r.x.v; out3[1] = r.y.v; out3[2] = r.z.v;
}

// Samplers, for the sampler-emulation tests.
void orc_sample_bilinear(const sah_plane* p, float u, float v, int mode, float* out4) {
    Image im{(const uint8_t*)p->ptr, p->width, p->height, 1, p->row_pitch_bytes, 0, p->format};
    Texel t = sample_bilinear(im, u, v, 0, (AddressMode)mode);
    for (int i = 0; i < 4; i++) out4[i] = t.c[i];
}
void orc_sample_trilinear(const sah_volume* p, float u, float v, float w, int mode, float* out4) {
    Image im{(const uint8_t*)p->ptr, p->width, p->height, p->depth, p->row_pitch_bytes, p->slice_pitch_bytes, p->format};
    Texel t = sample_trilinear(im, u, v, w, (AddressMode)mode);
    for (int i = 0; i < 4; i++) out4[i] = t.c[i];
}
float orc_sample_shadow(const sah_volume* p, float u, float v, int layer, float ref) {
    Image im{(const uint8_t*)p->ptr, p->width, p->height, p->depth, p->row_pitch_bytes, p->slice_pitch_bytes, p->format};
    return sample_shadow_pcf(im, u, v, layer, ref, ADDR_CLAMP_TO_EDGE);
}

}  // extern "C"
