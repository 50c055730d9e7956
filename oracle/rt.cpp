// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// CPU restatement of the ray-tracing half of SURVEY.md §8-f4:
//   RTAO               RenderCore/shaders/ao/rtao.comp.slang:54-102 (host: render/phase/ambient_occlusion_phase.cpp:357-397)
//   sun shadow rays    RenderCore/shaders/lighting/directional_light.rt.slang:91-125 (host: render/directional_light.cpp:372-422)
//   occlusion hit group / miss  RenderCore/shaders/materials/gltf_basic_pbr.slang:291-325, shaders/sky/sky_unified.slang:210-215
//   probe rays         RenderCore/shaders/gi/cache/probe_tracing.rt.slang:39-106 (host: render/gi/irradiance_cache.cpp dispatch_probe_updates)
//   RTGI rays          RenderCore/shaders/gi/rtgi/rtgi.rt.slang:56-110 (host: render/gi/rtgi.cpp post_render)
//   GI hit group / miss  RenderCore/shaders/materials/gltf_basic_pbr.slang:326-520, shaders/sky/sky_unified.slang:225-230
//   instances          RenderCore/render/raytracing_scene.cpp:15-43 (transform = model, SOLID opaque, CUTOUT non-opaque, no face culling)
// What a ray hits is the implementation's business in Vulkan; include/sah_hip.h ("ray tracing") fixes it.  This file tests EVERY
// triangle against every ray — no acceleration structure — which the definition makes equivalent to any box hierarchy.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/sah_hip.h"
#include "brdf.hpp"
#include "codec.hpp"
#include "gi.hpp"
#include "math.hpp"
#include "sky.hpp"
#include "texture.hpp"

namespace orc {
F3 sample_probe_cascade(const sah_gi& gi, F3 location, F3 direction, uint32_t cascade_index);  // gi.cpp
F3 octahedral_texel_direction(uint32_t tx, uint32_t ty, uint32_t n);                           // probes.cpp

namespace {

struct WorldTriangle {
    float v[3][3];
    uint32_t primitive, triangle;
    bool cutout;
};
struct RtStructure {
    std::vector<WorldTriangle> tris;
    float pad = 0.0f;
    uint32_t dropped = 0;
};

bool finite3(const float* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

// world = model * (p, 1), rows ((m0 x + m1 y) + m2 z) + m3 (the rasteriser's vertex stage, raster.cpp: process_triangle)
RtStructure build(const sah_scene_geometry& g) {
    RtStructure s;
    float S = 0.0f;
    for (uint32_t p = 0; p < g.num_primitives; p++) {
        const sah_primitive& prim = g.primitives[p];
        for (uint32_t tri = 0; tri < prim.index_count / 3; tri++) {
            WorldTriangle w;
            bool ok = (uint64_t)prim.first_index + 3ull * tri + 2ull < (uint64_t)g.num_indices;
            for (int k = 0; k < 3 && ok; k++) {
                const int64_t vi = (int64_t)prim.vertex_offset + (int64_t)g.indices[prim.first_index + 3 * tri + k];
                ok = vi >= 0 && vi < (int64_t)g.num_vertices;
                if (!ok) break;
                const float* pos = g.vertex_positions + 3 * vi;
                const float* m = prim.model;
                for (int c = 0; c < 3; c++) w.v[k][c] = ((F(m[c]) * F(pos[0]) + F(m[4 + c]) * F(pos[1])) + F(m[8 + c]) * F(pos[2]) + F(m[12 + c])).v;
                ok = finite3(w.v[k]);
            }
            if (!ok) { s.dropped++; continue; }
            w.primitive = p;
            w.triangle = tri;
            w.cutout = prim.type == SAH_PRIMITIVE_TYPE_CUTOUT;
            for (int k = 0; k < 3; k++)
                for (int c = 0; c < 3; c++) S = std::fmax(S, std::fabs(w.v[k][c]));
            s.tris.push_back(w);
        }
    }
    s.pad = S * 0x1p-16f;
    return s;
}

struct Ray {
    float o[3], d[3], inv[3], tmin, tmax;
    int kx, ky, kz;
    float Sx, Sy, Sz;
    bool finite;
};
Ray make_ray(const float o[3], const float d[3], float tmin, float tmax) {
    Ray r;
    for (int c = 0; c < 3; c++) { r.o[c] = o[c]; r.d[c] = d[c]; r.inv[c] = 1.0f / d[c]; }
    r.tmin = tmin; r.tmax = std::fmin(tmax, 3.402823466e+38f);  // at most the largest finite float (sah_hip.h "ray tracing")
    r.finite = finite3(o) && finite3(d);
    int kz = 0;
    float am = std::fabs(d[0]);
    if (std::fabs(d[1]) > am) { kz = 1; am = std::fabs(d[1]); }
    if (std::fabs(d[2]) > am) { kz = 2; }
    int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    if (d[kz] < 0.0f) std::swap(kx, ky);
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.Sx = d[kx] / d[kz];
    r.Sy = d[ky] / d[kz];
    r.Sz = 1.0f / d[kz];
    return r;
}

bool slab(const Ray& r, const float lo[3], const float hi[3]) {
    float tn = r.tmin, tf = r.tmax;
    for (int c = 0; c < 3; c++) {
        const float t0 = (lo[c] - r.o[c]) * r.inv[c], t1 = (hi[c] - r.o[c]) * r.inv[c];
        tn = std::fmax(tn, std::fmin(t0, t1));
        tf = std::fmin(tf, std::fmax(t0, t1));
    }
    return tn <= tf;
}

struct Hit { float t, b1, b2; bool front; };
bool woop(const Ray& r, const WorldTriangle& w, Hit& h) {
    float A[3], B[3], C[3];
    for (int c = 0; c < 3; c++) { A[c] = w.v[0][c] - r.o[c]; B[c] = w.v[1][c] - r.o[c]; C[c] = w.v[2][c] - r.o[c]; }
    const float Ax = A[r.kx] - r.Sx * A[r.kz], Ay = A[r.ky] - r.Sy * A[r.kz];
    const float Bx = B[r.kx] - r.Sx * B[r.kz], By = B[r.ky] - r.Sy * B[r.kz];
    const float Cx = C[r.kx] - r.Sx * C[r.kz], Cy = C[r.ky] - r.Sy * C[r.kz];
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = (U + V) + W;
    if (det == 0.0f) return false;
    const float Az = r.Sz * A[r.kz], Bz = r.Sz * B[r.kz], Cz = r.Sz * C[r.kz];
    const float T = (U * Az + V * Bz) + W * Cz;
    const float t = T / det;
    if (!(t > r.tmin && t < r.tmax)) return false;
    h.t = t; h.b1 = V / det; h.b2 = W / det;
    h.front = det > 0.0f;  // (v1 - v0) x (v2 - v0) against the ray: the default front face of Vulkan / D3D12 (sah_hip.h "Facing")
    return true;
}

H unpack_alpha(uint32_t packed) { return H((float)(packed >> 24)) / H::lit(255.0); }
uint32_t to_uint_sat(float f) { return f > 0.0f ? (f >= 4294967296.0f ? 0xffffffffu : (uint32_t)f) : 0u; }

// gltf_basic_pbr.slang:291-318 (SAH_RT_OCCLUSION, SAH_MASKED any-hit): true = accepted
bool cutout_accepts(const sah_scene_geometry& g, const WorldTriangle& w, const Hit& h) {
    const sah_primitive& prim = g.primitives[w.primitive];
    const uint32_t* idx = g.indices + prim.first_index + 3 * w.triangle;
    const sah_vertex_data& a = g.vertex_data[(int64_t)prim.vertex_offset + idx[0]];
    const sah_vertex_data& b = g.vertex_data[(int64_t)prim.vertex_offset + idx[1]];
    const sah_vertex_data& c = g.vertex_data[(int64_t)prim.vertex_offset + idx[2]];
    const F b0 = (F(1.0f) - F(h.b1)) - F(h.b2), b1 = F(h.b1), b2 = F(h.b2);  // float3(1.f - bary.x - bary.y, bary.x, bary.y)
    float uv[2];
    for (int k = 0; k < 2; k++) uv[k] = ((b0 * F(a.texcoord[k]) + b1 * F(b.texcoord[k])) + b2 * F(c.texcoord[k])).v;
    const F ca = (b0 * F(unpack_alpha(a.color).v) + b1 * F(unpack_alpha(b.color).v)) + b2 * F(unpack_alpha(c.color).v);
    const uint32_t alpha_byte = to_uint_sat((H(ca.v) * H::lit(255.0)).v) & 0xffu;  // packUnorm4x8: (uint)(value.w * 255.h)
    const H colour_a = H((float)alpha_byte) / H::lit(255.0);                        // unpackUnorm4x8ToHalf(v.color).w
    const sah_material& m = g.materials[prim.material];
    float texel_a = m.base_color_texel[3];
    if (g.material_textures && g.textures && g.num_textures) {
        const uint32_t ti = g.material_textures[prim.material].base_color;
        if (ti != SAH_TEXTURE_NONE) {
            float texel[4];
            sample_texture_lod(g.textures[ti], uv, 0.0f, 0.0f, texel);
            texel_a = texel[3];
        }
    }
    const F alpha = (F(texel_a) * F(m.base_color_tint[3])) * F(colour_a.v);
    return !(alpha.v <= m.opacity_threshold);
}

// candidate := slab(padded box of the triangle) and woop (sah_hip.h "ray tracing")
bool candidate(const RtStructure& s, const Ray& r, const WorldTriangle& w, Hit& h) {
    float lo[3], hi[3];
    for (int c = 0; c < 3; c++) {
        lo[c] = std::fmin(std::fmin(w.v[0][c], w.v[1][c]), w.v[2][c]) - s.pad;
        hi[c] = std::fmax(std::fmax(w.v[0][c], w.v[1][c]), w.v[2][c]) + s.pad;
    }
    return slab(r, lo, hi) && woop(r, w, h);
}

bool any_hit(const sah_scene_geometry& g, const RtStructure& s, const Ray& r, bool cull_non_opaque, bool cull_front = false) {
    if (!r.finite) return false;
    for (const WorldTriangle& w : s.tris) {
        if (cull_non_opaque && w.cutout) continue;
        Hit h;
        if (!candidate(s, r, w, h)) continue;
        if (cull_front && h.front) continue;
        if (!w.cutout || cutout_accepts(g, w, h)) return true;
    }
    return false;
}

// The accepted candidate of smallest t; equal t: smallest (primitive, triangle).  Returns the index into s.tris or -1.  RAY_FLAG_NONE for the
// generators' rays; `bounce`: RAY_FLAG_CULL_NON_OPAQUE | RAY_FLAG_CULL_BACK_FACING_TRIANGLES (gltf_basic_pbr.slang:498-506)
int closest_hit(const sah_scene_geometry& g, const RtStructure& s, const Ray& r, Hit& best, bool bounce = false) {
    int found = -1;
    if (!r.finite) return found;
    for (size_t i = 0; i < s.tris.size(); i++) {
        const WorldTriangle& w = s.tris[i];
        if (bounce && w.cutout) continue;
        Hit h;
        if (!candidate(s, r, w, h)) continue;
        if (bounce && !h.front) continue;
        if (found >= 0) {
            const WorldTriangle& b = s.tris[found];
            const bool better = h.t < best.t || (h.t == best.t && (w.primitive < b.primitive || (w.primitive == b.primitive && w.triangle < b.triangle)));
            if (!better) continue;
        }
        if (w.cutout && !cutout_accepts(g, w, h)) continue;
        found = (int)i;
        best = h;
    }
    return found;
}

bool scene_ok(const sah_scene_geometry* g) {
    if (!g) return false;
    if (g->num_primitives && (!g->primitives || !g->indices || !g->vertex_positions)) return false;
    return true;
}
bool plane_is(const sah_plane* p, uint32_t fmt) { return p && p->ptr && p->format == fmt; }

H3 load_normal(const sah_plane& p, int x, int y) {
    uint16_t h[4];
    std::memcpy(h, (const uint8_t*)p.ptr + (size_t)y * p.row_pitch_bytes + (size_t)x * 8, 8);
    return normalize(H3{H::raw(f16_to_f32(h[0])), H::raw(f16_to_f32(h[1])), H::raw(f16_to_f32(h[2]))});
}
F3 load_noise(const sah_plane& p, uint32_t x, uint32_t y) {
    const uint8_t* t = (const uint8_t*)p.ptr + (size_t)y * p.row_pitch_bytes + (size_t)x * 4;
    const F3 v = {F(unorm8_to_float(t[0])) * F(2.0f) - F(1.0f), F(unorm8_to_float(t[1])) * F(2.0f) - F(1.0f), F(unorm8_to_float(t[2])) * F(2.0f) - F(1.0f)};
    return normalize(v);
}
float load_f32(const sah_plane& p, int x, int y) {
    float f;
    std::memcpy(&f, (const uint8_t*)p.ptr + (size_t)y * p.row_pitch_bytes + (size_t)x * 4, 4);
    return f;
}

// what the GI hit and miss stages read besides the scene
struct GiInputs {
    const sah_scene_geometry* scene;
    const RtStructure* s;
    const sah_sun_light_constants* sun;
    const sah_sky_luts* sky;
    const sah_plane* noise;
};
struct GiPayload {
    F3 irradiance = F3(F(0.0f));
    F ray_distance = F(0.0f);
};

H unpack_channel(uint32_t packed, int c) { return H((float)((packed >> (8 * c)) & 0xffu)) / H::lit(255.0); }

// payload.remaining_bounces of the generators' rays: 0 in the reference (rtgi.rt.slang:88, probe_tracing.rt.slang:66), orc_rt_set_bounces
uint32_t g_num_bounces = 0;

// TraceRay(rtas, RAY_FLAG_NONE, 0xFF, RAY_TYPE_GI, ...): gltf_basic_pbr.slang:372-520 or sky_unified.slang:227-230.
// (dx, dy) = DispatchRaysIndex().xy; `bounce`: the ray is the bounce ray of a hit stage (other flags, :498-506)
GiPayload trace_gi(const GiInputs& in, const Ray& r, uint32_t dx, uint32_t dy, uint32_t remaining_bounces, bool bounce = false) {
    GiPayload pay;
    const sah_scene_geometry& g = *in.scene;
    Hit h = {0.0f, 0.0f, 0.0f, false};
    const int found = closest_hit(g, *in.s, r, h, bounce);
    if (found < 0) {
        if (r.finite) {  // miss stage: get_sky_color(WorldRayDirection(), sun_light.direction_and_tan_size.xyz, ...) — the direction as stored
            static const SkyConsts k;
            const F3 sun_dir = {F(in.sun->direction_and_tan_size[0]), F(in.sun->direction_and_tan_size[1]), F(in.sun->direction_and_tan_size[2])};
            pay.irradiance = sky_color(k, F3{F(r.d[0]), F(r.d[1]), F(r.d[2])}, sun_dir, img2d(in.sky->sky_view), img2d(in.sky->transmittance));
        }
        return pay;
    }
    const WorldTriangle& w = in.s->tris[found];
    const sah_primitive& prim = g.primitives[w.primitive];
    const uint32_t* idx = g.indices + prim.first_index + 3 * w.triangle;
    const int64_t vi[3] = {(int64_t)prim.vertex_offset + idx[0], (int64_t)prim.vertex_offset + idx[1], (int64_t)prim.vertex_offset + idx[2]};
    const sah_vertex_data &v0 = g.vertex_data[vi[0]], &v1 = g.vertex_data[vi[1]], &v2 = g.vertex_data[vi[2]];
    const F b0 = (F(1.0f) - F(h.b1)) - F(h.b2), b1 = F(h.b1), b2 = F(h.b2);
    // interpolate_vertex (:278-287)
    F3 normal;
    normal.x = (b0 * F(v0.normal[0]) + b1 * F(v1.normal[0])) + b2 * F(v2.normal[0]);
    normal.y = (b0 * F(v0.normal[1]) + b1 * F(v1.normal[1])) + b2 * F(v2.normal[1]);
    normal.z = (b0 * F(v0.normal[2]) + b1 * F(v1.normal[2])) + b2 * F(v2.normal[2]);
    float uv[2];
    for (int k = 0; k < 2; k++) uv[k] = ((b0 * F(v0.texcoord[k]) + b1 * F(v1.texcoord[k])) + b2 * F(v2.texcoord[k])).v;
    H colour[4];
    for (int k = 0; k < 4; k++) {
        const F c = (b0 * F(unpack_channel(v0.color, k).v) + b1 * F(unpack_channel(v1.color, k).v)) + b2 * F(unpack_channel(v2.color, k).v);
        const uint32_t byte = to_uint_sat((H(c.v) * H::lit(255.0)).v) & 0xffu;  // packUnorm4x8, one channel
        colour[k] = H((float)byte) / H::lit(255.0);
    }
    // surface.location = model * (b.x p0 + b.y p1 + b.z p2, 1)
    float mp[3], loc[3];
    for (int k = 0; k < 3; k++)
        mp[k] = ((b0 * F(g.vertex_positions[3 * vi[0] + k]) + b1 * F(g.vertex_positions[3 * vi[1] + k])) + b2 * F(g.vertex_positions[3 * vi[2] + k])).v;
    for (int k = 0; k < 3; k++)
        loc[k] = (((F(prim.model[k]) * F(mp[0]) + F(prim.model[4 + k]) * F(mp[1])) + F(prim.model[8 + k]) * F(mp[2])) + F(prim.model[12 + k]) * F(1.0f)).v;
    const sah_material& m = g.materials[prim.material];
    float base_t[4], data_t[4], emis_t[4];
    for (int k = 0; k < 4; k++) {
        base_t[k] = m.base_color_texel[k];
        data_t[k] = m.data_texel[k];
        emis_t[k] = m.emission_texel[k];
    }
    if (g.material_textures && g.textures && g.num_textures) {
        const sah_material_textures& mt = g.material_textures[prim.material];
        if (mt.base_color != SAH_TEXTURE_NONE) sample_texture_lod(g.textures[mt.base_color], uv, 0.0f, 0.0f, base_t);  // SampleLevel(v.texcoord, 0)
        if (mt.data != SAH_TEXTURE_NONE) sample_texture_lod(g.textures[mt.data], uv, 0.0f, 0.0f, data_t);
        if (mt.emission != SAH_TEXTURE_NONE) sample_texture_lod(g.textures[mt.emission], uv, 0.0f, 0.0f, emis_t);
    }
    Surface<H> surf;
    surf.base_color = {H(((F(base_t[0]) * F(m.base_color_tint[0])) * F(colour[0].v)).v), H(((F(base_t[1]) * F(m.base_color_tint[1])) * F(colour[1].v)).v),
                       H(((F(base_t[2]) * F(m.base_color_tint[2])) * F(colour[2].v)).v)};
    surf.normal = to_h(normal);  // no normal map in the ray-traced path ("TODO" in the shader), not normalised either
    surf.roughness = H(data_t[1]) * H(m.roughness_factor);
    surf.metalness = H(data_t[2]) * H(m.metalness_factor);
    const H3 emission = {H(emis_t[0]) * H(m.emission_factor[0]), H(emis_t[1]) * H(m.emission_factor[1]), H(emis_t[2]) * H(m.emission_factor[2])};
    const float* sd = in.sun->direction_and_tan_size;
    const H3 light = normalize(H3{H(-sd[0]), H(-sd[1]), H(-sd[2])});
    const H3 brdf_result = Fd(surf, light, surf.normal);
    const H ndotl = nclamp(dot(light, surf.normal), H::lit(0.0), H::lit(1.0));
    H shadow = H::lit(0.0);
    if (ndotl.v > 0.0f) {
        const F3 noise = load_noise(*in.noise, dx % 128u, dy % 128u);
        const F3 dir = normalize(to_f(light) + noise * F(sd[3]));
        const float d[3] = {dir.x.v, dir.y.v, dir.z.v};
        // ACCEPT_FIRST_HIT_AND_END_SEARCH | CULL_NON_OPAQUE | CULL_FRONT_FACING_TRIANGLES
        if (!any_hit(g, *in.s, make_ray(loc, d, 0.05f, 100000.0f), true, true)) shadow = H::lit(1.0);
    }
    const F3 sun_colour = {F(in.sun->color[0]), F(in.sun->color[1]), F(in.sun->color[2])};
    F3 irr = to_f(brdf_result) * sun_colour * F(ndotl.v) * F(shadow.v);
    irr = irr + to_f(emission);
    pay.irradiance = irr;
    pay.ray_distance = F(h.t);
    if (remaining_bounces > 0) {  // :481-517
        F3 ray_direction = load_noise(*in.noise, dx % 128u, dy % 128u);
        if (dot(to_f(surf.normal), ray_direction).v < 0.0f) ray_direction = ray_direction * F(-1.0f);
        const float d[3] = {ray_direction.x.v, ray_direction.y.v, ray_direction.z.v};
        const GiPayload new_payload = trace_gi(in, make_ray(loc, d, 0.05f, 100000.0f), dx, dy, remaining_bounces - 1, true);
        const H3 bounce_brdf_result = brdf(surf, to_h(ray_direction), surf.normal);  // brdf() = Fd() + Fr() here, Fd() alone for the sun above (:438)
        const H bounce_ndotl = H(nclamp(dot(ray_direction, to_f(surf.normal)), F(0.0f), F(1.0f)).v);
        const F3 bounce_radiance = to_f(bounce_brdf_result * bounce_ndotl) * new_payload.irradiance;
        const float br[3] = {bounce_radiance.x.v, bounce_radiance.y.v, bounce_radiance.z.v};
        if (std::isfinite(br[0]) && std::isfinite(br[1]) && std::isfinite(br[2])) pay.irradiance = pay.irradiance + bounce_radiance;
    }
    if (!h.front) {  // HIT_KIND_TRIANGLE_BACK_FACE
        pay.ray_distance = pay.ray_distance * F(-1.0f);
        pay.irradiance = F3(F(0.0f));
    }
    return pay;
}

bool gi_inputs_ok(const sah_sun_light_constants* sun, const sah_sky_luts* sky, const sah_plane* noise) {
    return sun && sky && plane_is(&sky->transmittance, SAH_FORMAT_R16G16B16A16_SFLOAT) && plane_is(&sky->sky_view, SAH_FORMAT_R16G16B16A16_SFLOAT) &&
           plane_is(noise, SAH_FORMAT_R8G8B8A8_UNORM) && noise->width >= 128 && noise->height >= 128;
}
void store_half4(uint8_t* p, const float v[4]) {
    const uint16_t h[4] = {f32_to_f16(v[0]), f32_to_f16(v[1]), f32_to_f16(v[2]), f32_to_f16(v[3])};
    std::memcpy(p, h, 8);
}

}  // namespace
}  // namespace orc

extern "C" {

// stats (4 words, may be null): triangles kept, left out, 0, 0
int orc_rt_stats(const sah_scene_geometry* scene, uint32_t* stats, float* pad) {
    using namespace orc;
    if (!scene_ok(scene)) return SAH_ERR_INVALID_ARGUMENT;
    const RtStructure s = build(*scene);
    if (stats) { stats[0] = (uint32_t)s.tris.size(); stats[1] = s.dropped; stats[2] = 0; stats[3] = 0; }
    if (pad) *pad = s.pad;
    return SAH_OK;
}

int orc_rtao(const sah_scene_geometry* scene, const sah_view_data* view, const sah_plane* depth, const sah_plane* normals, const sah_plane* noise,
             uint32_t samples_per_pixel, float max_ray_distance, const sah_plane* ao_out) {
    using namespace orc;
    if (!scene_ok(scene) || !view || !plane_is(ao_out, SAH_FORMAT_R32_SFLOAT) || !depth || !depth->ptr || !plane_is(normals, SAH_FORMAT_R16G16B16A16_SFLOAT) ||
        !plane_is(noise, SAH_FORMAT_R8G8B8A8_UNORM) || samples_per_pixel > 4096)
        return SAH_ERR_INVALID_ARGUMENT;
    const RtStructure s = build(*scene);
    const int W = (int)ao_out->width, Hh = (int)ao_out->height;
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < Hh; y++)
        for (int x = 0; x < W; x++) {
            const F3 pos = worldspace_location_slang(*view, x, y, load_f32(*depth, x, y));  // rtao.comp.slang:27-36
            const H3 normal = load_normal(*normals, x, y);
            F3 n = load_noise(*noise, (uint32_t)x % noise->width, (uint32_t)y % noise->height);
            if (dot(n, to_f(normal)).v < 0.0f) n = n * F(-1.0f);
            const float o[3] = {pos.x.v, pos.y.v, pos.z.v}, d[3] = {n.x.v, n.y.v, n.z.v};
            float ao = (float)samples_per_pixel;
            for (uint32_t i = 0; i < samples_per_pixel; i++)  // the same noise texel every time: the same ray
                if (any_hit(*scene, s, make_ray(o, d, 0.01f, max_ray_distance), /*cull_non_opaque=*/true)) ao -= 1.0f;
            ao /= (float)samples_per_pixel;
            std::memcpy((uint8_t*)ao_out->ptr + (size_t)y * ao_out->row_pitch_bytes + (size_t)x * 4, &ao, 4);
        }
    return SAH_OK;
}

int orc_sun_shadow_mask(const sah_scene_geometry* scene, const sah_view_data* view, const sah_sun_light_constants* sun, const sah_plane* depth,
                        const sah_plane* normals, const sah_plane* noise, const sah_plane* mask_out) {
    using namespace orc;
    if (!scene_ok(scene) || !view || !sun || !plane_is(mask_out, SAH_FORMAT_R32_SFLOAT) || !depth || !depth->ptr ||
        !plane_is(normals, SAH_FORMAT_R16G16B16A16_SFLOAT) || !plane_is(noise, SAH_FORMAT_R8G8B8A8_UNORM) || noise->width < 128 || noise->height < 128 ||
        !(sun->num_shadow_samples >= 0.0f && sun->num_shadow_samples <= 4096.0f))
        return SAH_ERR_INVALID_ARGUMENT;
    const RtStructure s = build(*scene);
    const int W = (int)mask_out->width, Hh = (int)mask_out->height;
    const F3 L = normalize(F3{-F(sun->direction_and_tan_size[0]), -F(sun->direction_and_tan_size[1]), -F(sun->direction_and_tan_size[2])});
    const F phi = F(1.618033988749895f);
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < Hh; y++)
        for (int x = 0; x < W; x++) {
            float mask = 1.0f;
            const float depth_v = load_f32(*depth, x, y);
            const H3 normal = load_normal(*normals, x, y);
            const H ndotl = H(nclamp(dot(L, to_f(normal)), F(0.0f), F(1.0f)).v);
            if (depth_v != 0.0f && ndotl.v > 0.0f) {
                const F3 pos = worldspace_location_slang(*view, x, y, depth_v);
                const float o[3] = {pos.x.v, pos.y.v, pos.z.v};
                F shadow = F(0.0f);
                for (uint32_t i = 0; (float)i < sun->num_shadow_samples; i++) {
                    const F q = F((float)i) / phi;
                    const F r0x = F(2.0f) + q, r0y = F(3.0f) + q;
                    const F fx = r0x - F(std::floor(r0x.v)), fy = r0y - F(std::floor(r0y.v));
                    const float offx = std::nearbyint((fx * F(128.0f)).v), offy = std::nearbyint((fy * F(128.0f)).v);
                    const uint32_t nx = to_uint_sat((F((float)x) + F(offx)).v) % 128u, ny = to_uint_sat((F((float)y) + F(offy)).v) % 128u;
                    const F3 n = load_noise(*noise, nx, ny);
                    const F3 dir = normalize(L + n * F(sun->direction_and_tan_size[3]));
                    const float d[3] = {dir.x.v, dir.y.v, dir.z.v};
                    shadow = shadow + F(any_hit(*scene, s, make_ray(o, d, 0.01f, 100000.0f), /*cull_non_opaque=*/false) ? 0.0f : 1.0f);
                }
                mask = (shadow / F(sun->num_shadow_samples)).v;
            }
            std::memcpy((uint8_t*)mask_out->ptr + (size_t)y * mask_out->row_pitch_bytes + (size_t)x * 4, &mask, 4);
        }
    return SAH_OK;
}

// probe_tracing.rt.slang:39-106
// remaining_bounces of the rays orc_probe_trace / orc_rtgi_trace generate from now on (process-wide; the reference's generators: 0)
int orc_rt_set_bounces(uint32_t num_bounces) {
    if (num_bounces > 8) return 1;
    orc::g_num_bounces = num_bounces;
    return 0;
}

int orc_probe_trace(const sah_scene_geometry* scene, const sah_probe_trace_desc* d) {
    using namespace orc;
    if (!scene_ok(scene) || !d || !gi_inputs_ok(d->sun, d->sky, d->noise)) return SAH_ERR_INVALID_ARGUMENT;
    if (d->num_probes == 0) return SAH_OK;
    const sah_volume& tr = d->trace_results;
    if (!d->probes_to_update || !tr.ptr || tr.format != SAH_FORMAT_R16G16B16A16_SFLOAT || tr.width != 20 || tr.height != 20 || tr.depth < d->num_probes)
        return SAH_ERR_INVALID_ARGUMENT;
    const RtStructure s = build(*scene);
    const GiInputs in = {scene, &s, d->sun, d->sky, d->noise};
    sah_gi gi;
    std::memset(&gi, 0, sizeof(gi));
    gi.kind = SAH_GI_CACHE;
    gi.probe_irradiance = d->probe_irradiance;
    gi.probe_depth = d->probe_depth;
    gi.probe_validity = d->probe_validity;
    for (int c = 0; c < 4; c++) gi.probe_cascades[c] = d->cascades[c];
    gi.probe_size[0] = d->probe_size[0];
    gi.probe_size[1] = d->probe_size[1];
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t probe = 0; probe < (int64_t)d->num_probes; probe++) {
        const uint32_t* id = d->probes_to_update + 3 * probe;
        const uint32_t cascade = id[1] / 8;
        for (uint32_t ty = 0; ty < 20; ty++)
            for (uint32_t tx = 0; tx < 20; tx++) {
                float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (cascade < 4) {  // (the shader indexes a 4-entry array; the ABI writes zeros for anything else)
                    const sah_probe_cascade& c = d->cascades[cascade];
                    const F3 local = {F((float)id[0]), F((float)(id[1] % 8)), F((float)id[2])};
                    const F3 origin = F3{F(c.min[0]), F(c.min[1]), F(c.min[2])} + local * F(c.probe_spacing);
                    const F3 dir = octahedral_texel_direction(tx, ty, 20);
                    F ray_distance = F(8192.0f);
                    if (cascade < 3) ray_distance = F(d->cascades[cascade + 1].probe_spacing) * F(4.0f);
                    const float o[3] = {origin.x.v, origin.y.v, origin.z.v}, dd[3] = {dir.x.v, dir.y.v, dir.z.v};
                    GiPayload pay = trace_gi(in, make_ray(o, dd, 0.05f, ray_distance.v), tx, ty, g_num_bounces);
                    if (pay.ray_distance.v == 0.0f) {
                        if (cascade + 1 < 4) pay.irradiance = sample_probe_cascade(gi, origin + dir * ray_distance, dir, cascade + 1);
                        else pay.irradiance = pay.irradiance * F(10.0f);
                        pay.ray_distance = ray_distance;
                    } else if (pay.ray_distance.v < 0.0f) {
                        pay.irradiance = F3(F(0.0f));
                    }
                    const H e = H::lit(0.0031415927);
                    out[0] = (H(pay.irradiance.x.v) * e).v;
                    out[1] = (H(pay.irradiance.y.v) * e).v;
                    out[2] = (H(pay.irradiance.z.v) * e).v;
                    out[3] = H(pay.ray_distance.v).v;
                }
                store_half4((uint8_t*)tr.ptr + (size_t)probe * tr.slice_pitch_bytes + (size_t)ty * tr.row_pitch_bytes + (size_t)tx * 8, out);
            }
    }
    return SAH_OK;
}

// rtgi.rt.slang:56-110.  Pixels the generator returns early from keep what the two targets held
int orc_rtgi_trace(const sah_scene_geometry* scene, const sah_view_data* view, const sah_sun_light_constants* sun, const sah_sky_luts* sky,
                   const sah_plane* depth, const sah_plane* normals, const sah_plane* noise, const sah_plane* ray_buffer, const sah_plane* ray_irradiance) {
    using namespace orc;
    if (!scene_ok(scene) || !view || !gi_inputs_ok(sun, sky, noise) || !depth || !depth->ptr || !plane_is(normals, SAH_FORMAT_R16G16B16A16_SFLOAT) ||
        !plane_is(ray_buffer, SAH_FORMAT_R16G16B16A16_SFLOAT) || !plane_is(ray_irradiance, SAH_FORMAT_R16G16B16A16_SFLOAT))
        return SAH_ERR_INVALID_ARGUMENT;
    const RtStructure s = build(*scene);
    const GiInputs in = {scene, &s, sun, sky, noise};
    const int W = (int)ray_buffer->width, Hh = (int)ray_buffer->height;
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < Hh; y++)
        for (int x = 0; x < W; x++) {
            if (!((float)x < view->render_resolution[0] && (float)y < view->render_resolution[1])) continue;
            const float depth_v = load_f32(*depth, x, y);
            if (depth_v == 0.0f) continue;
            uint16_t nh[4];
            std::memcpy(nh, (const uint8_t*)normals->ptr + (size_t)y * normals->row_pitch_bytes + (size_t)x * 8, 8);
            const F3 normal = {F(f16_to_f32(nh[0])), F(f16_to_f32(nh[1])), F(f16_to_f32(nh[2]))};  // as stored: not normalised here
            const F3 pos = worldspace_location_slang(*view, x, y, depth_v);
            F3 dir = load_noise(*noise, (uint32_t)x % 128u, (uint32_t)y % 128u);
            if (dot(normal, dir).v < 0.0f) dir = dir * F(-1.0f);
            const float o[3] = {pos.x.v, pos.y.v, pos.z.v}, dd[3] = {dir.x.v, dir.y.v, dir.z.v};
            GiPayload pay = trace_gi(in, make_ray(o, dd, 0.01f, 100000.0f), (uint32_t)x, (uint32_t)y, g_num_bounces);
            if (any_nan(pay.irradiance)) pay.irradiance = F3(F(0.0f));
            const F e = F(0.0031415927f);
            const float rb[4] = {dir.x.v, dir.y.v, dir.z.v, pay.ray_distance.v};
            const float ri[4] = {(pay.irradiance.x * e).v, (pay.irradiance.y * e).v, (pay.irradiance.z * e).v, 0.0f};
            store_half4((uint8_t*)ray_buffer->ptr + (size_t)y * ray_buffer->row_pitch_bytes + (size_t)x * 8, rb);
            store_half4((uint8_t*)ray_irradiance->ptr + (size_t)y * ray_irradiance->row_pitch_bytes + (size_t)x * 8, ri);
        }
    return SAH_OK;
}

}  // extern "C"
