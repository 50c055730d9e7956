// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// CPU restatement of the two rasterisation passes either side of the lighting pass:
//   sun shadow cascades   RenderCore/render/directional_light.cpp:286-327, pipelines RenderCore/render/material_pipelines.cpp:31-62,
//                         vertex stage RenderCore/shaders/materials/gltf_basic_pbr.slang:110-146 (SAH_MULTIVIEW)
//   depth + G-buffer      RenderCore/render/phase/gbuffer_phase.cpp:27-97, pipelines material_pipelines.cpp:13-29,104-140,
//                         shaders gltf_basic_pbr.slang:110-253 (SAH_MAIN_VIEW); cull / front face RenderCore/render/render_scene.cpp:196-222
// The reference hands these to the hardware rasteriser.  What is restated is the Vulkan 1.4 rasterisation contract with the
// implementation-defined parts fixed as DESIGN.md §5d lists them: immediate mode, one triangle after the other in draw order.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/sah_hip.h"
#include "codec.hpp"
#include "brdf.hpp"
#include "math.hpp"
#include "texture.hpp"

namespace orc {
namespace {

constexpr int kSubPixel = 256;             // 8 sub-pixel bits
constexpr float kGuardBand = 16.0f;        // |x_c|, |y_c| <= kGuardBand * w_c survives clipping
constexpr int64_t kCoordLimit = (1ll << 24) + 4096;  // snapped coordinates beyond this drop the triangle (edge functions stay below 2^52)

struct ClipVertex {
    F c[4];     // clip-space x y z w
    F bary[3];  // barycentric coordinates in the input triangle
};
struct RasterVertex {
    int64_t X, Y;
    F z, inv_w;
    F bary[3];
    bool finite;
};

ClipVertex lerp_vertex(const ClipVertex& in, const ClipVertex& out, F d_in, F d_out) {
    const F t = d_in / (d_in - d_out);
    ClipVertex r;
    for (int k = 0; k < 4; k++) r.c[k] = in.c[k] + (out.c[k] - in.c[k]) * t;
    for (int k = 0; k < 3; k++) r.bary[k] = in.bary[k] + (out.bary[k] - in.bary[k]) * t;
    return r;
}

// plane distances, in clipping order; a vertex is kept when d >= 0
F plane_distance(const ClipVertex& v, int plane) {
    switch (plane) {
        case 0: return v.c[2];                              // z >= 0
        case 1: return v.c[3] - v.c[2];                     // z <= w
        case 2: return F(kGuardBand) * v.c[3] - v.c[0];
        case 3: return F(kGuardBand) * v.c[3] + v.c[0];
        case 4: return F(kGuardBand) * v.c[3] - v.c[1];
        default: return F(kGuardBand) * v.c[3] + v.c[1];
    }
}

// Sutherland-Hodgman; the new vertex of a crossing edge is always computed from the inside vertex towards the outside one
int clip_polygon(ClipVertex* poly, int n, int first_plane) {
    ClipVertex tmp[12];
    for (int plane = first_plane; plane < 6 && n >= 3; plane++) {
        int m = 0;
        for (int i = 0; i < n; i++) {
            const ClipVertex& a = poly[i];
            const ClipVertex& b = poly[(i + 1) % n];
            const F da = plane_distance(a, plane), db = plane_distance(b, plane);
            const bool ia = da.v >= 0.0f, ib = db.v >= 0.0f;
            if (ia) tmp[m++] = a;
            if (ia != ib) tmp[m++] = ia ? lerp_vertex(a, b, da, db) : lerp_vertex(b, a, db, da);
        }
        n = m;
        for (int i = 0; i < n; i++) poly[i] = tmp[i];
    }
    return n < 3 ? 0 : n;
}

RasterVertex to_window(const ClipVertex& v, uint32_t W, uint32_t H) {
    RasterVertex r;
    const F xd = v.c[0] / v.c[3], yd = v.c[1] / v.c[3];
    r.z = v.c[2] / v.c[3];
    r.inv_w = F(1.0f) / v.c[3];
    const F hw = F((float)W * 0.5f), hh = F((float)H * 0.5f);
    const F xf = xd * hw + hw, yf = yd * hh + hh;
    const float sx = xf.v * (float)kSubPixel, sy = yf.v * (float)kSubPixel;
    r.finite = std::isfinite(sx) && std::isfinite(sy) && std::isfinite(r.z.v) && std::isfinite(r.inv_w.v) &&
               std::fabs(sx) <= (float)kCoordLimit && std::fabs(sy) <= (float)kCoordLimit;
    r.X = r.finite ? (int64_t)std::nearbyint(sx) : 0;
    r.Y = r.finite ? (int64_t)std::nearbyint(sy) : 0;
    for (int k = 0; k < 3; k++) r.bary[k] = v.bary[k];
    return r;
}

struct Fragment {
    int x, y;
    F z;
    F lambda[3];  // perspective-correct barycentrics in the input triangle
    // the same at the other pixel of the fragment's 2x2 quad row / column (quads sit at even window coordinates), extrapolated when that
    // pixel is not covered: what the derivatives of the texture coordinates are formed from
    F lambda_qx[3], lambda_qy[3];
};

struct Stats {
    uint32_t w[SAH_RASTER_STATS_WORDS] = {};
};

inline bool top_left(int64_t dx, int64_t dy) { return dy < 0 || (dy == 0 && dx > 0); }

// Rasterises one window-space triangle; `emit` is called for every covered pixel centre.
template <class Emit>
void raster_triangle(RasterVertex v0, RasterVertex v1, RasterVertex v2, bool cull_back, uint32_t W, uint32_t H, Stats& st, Emit&& emit) {
    if (!v0.finite || !v1.finite || !v2.finite) { st.w[2]++; return; }
    int64_t area = (v1.X - v0.X) * (v2.Y - v0.Y) - (v2.X - v0.X) * (v1.Y - v0.Y);
    if (area == 0 || (area < 0 && cull_back)) { st.w[1]++; return; }
    if (area < 0) { std::swap(v1, v2); area = -area; }
    const int64_t minx = std::min(v0.X, std::min(v1.X, v2.X)), maxx = std::max(v0.X, std::max(v1.X, v2.X));
    const int64_t miny = std::min(v0.Y, std::min(v1.Y, v2.Y)), maxy = std::max(v0.Y, std::max(v1.Y, v2.Y));
    // pixel p is a candidate when its centre 256 p + 128 lies in [min, max]
    auto first_px = [](int64_t lo) { const int64_t a = lo - 128; return a <= 0 ? (int64_t)0 : (a + 255) / 256; };
    auto last_px = [](int64_t hi, uint32_t size) {
        const int64_t a = hi - 128;
        if (a < 0) return (int64_t)-1;
        return std::min<int64_t>(a / 256, (int64_t)size - 1);
    };
    const int64_t x0 = first_px(minx), x1 = last_px(maxx, W), y0 = first_px(miny), y1 = last_px(maxy, H);
    if (x0 > x1 || y0 > y1) { st.w[1]++; return; }
    st.w[3]++;
    const RasterVertex* v[3] = {&v0, &v1, &v2};
    const F inv_area = F(1.0f) / F((float)area);
    // Depth is linear in window space: z(px, py) = zc + px zx + py zy, the three coefficients formed once per triangle in fp64
    // from the edge functions E_i(px, py) = c_i + a_i px + b_i py (exact integers) and the vertex depths, every operator rounded.
    double zc, zx, zy;
    {
        double ea[3], eb[3], ec[3];
        for (int i = 0; i < 3; i++) {
            const RasterVertex &a = *v[(i + 1) % 3], &b = *v[(i + 2) % 3];
            const double dx = (double)(b.X - a.X), dy = (double)(b.Y - a.Y);
            ea[i] = -256.0 * dy;
            eb[i] = 256.0 * dx;
            ec[i] = dx * (double)(128 - a.Y) - dy * (double)(128 - a.X);
        }
        const double inv = 1.0 / (double)area, z0 = (double)v0.z.v, z1 = (double)v1.z.v, z2 = (double)v2.z.v;
        zc = ((ec[0] * z0 + ec[1] * z1) + ec[2] * z2) * inv;
        zx = ((ea[0] * z0 + ea[1] * z1) + ea[2] * z2) * inv;
        zy = ((eb[0] * z0 + eb[1] * z1) + eb[2] * z2) * inv;
    }
    for (int64_t py = y0; py <= y1; py++)
        for (int64_t px = x0; px <= x1; px++) {
            const int64_t cx = px * 256 + 128, cy = py * 256 + 128;
            int64_t e[3];
            bool inside = true;
            for (int i = 0; i < 3; i++) {
                const RasterVertex &a = *v[(i + 1) % 3], &b = *v[(i + 2) % 3];
                const int64_t dx = b.X - a.X, dy = b.Y - a.Y;
                e[i] = dx * (cy - a.Y) - dy * (cx - a.X);
                inside = inside && (e[i] > 0 || (e[i] == 0 && top_left(dx, dy)));
            }
            if (!inside) continue;
            Fragment f;
            f.x = (int)px;
            f.y = (int)py;
            f.z = F((float)std::fma((double)py, zy, std::fma((double)px, zx, zc)));
            auto lambda_at = [&](int64_t qx, int64_t qy, F out[3]) {
                const int64_t ccx = qx * 256 + 128, ccy = qy * 256 + 128;
                F b[3];
                for (int i = 0; i < 3; i++) {
                    const RasterVertex &a = *v[(i + 1) % 3], &bb = *v[(i + 2) % 3];
                    b[i] = F((float)((bb.X - a.X) * (ccy - a.Y) - (bb.Y - a.Y) * (ccx - a.X))) * inv_area;
                }
                const F q0 = b[0] * v0.inv_w, q1 = b[1] * v1.inv_w, q2 = b[2] * v2.inv_w;
                const F s = q0 + q1 + q2;
                const F l0 = q0 / s, l1 = q1 / s, l2 = q2 / s;
                for (int k = 0; k < 3; k++) out[k] = l0 * v0.bary[k] + l1 * v1.bary[k] + l2 * v2.bary[k];
            };
            lambda_at(px, py, f.lambda);
            lambda_at(px ^ 1, py, f.lambda_qx);
            lambda_at(px, py ^ 1, f.lambda_qy);
            emit(f);
        }
}

M4 load_m4(const float* p) {
    M4 m;
    std::memcpy(m.m, p, 64);
    return m;
}

// Vertex fetch + clip + fan for input triangle `tri` of primitive `prim`; `to_clip` maps the world-space position to clip space.
template <class ToClip, class Emit>
void process_triangle(const sah_scene_geometry& g, const sah_primitive& prim, uint32_t tri, bool clip_depth, uint32_t W, uint32_t H, Stats& st,
                      ToClip&& to_clip, Emit&& emit) {
    st.w[0]++;
    ClipVertex poly[12];
    const M4 model = load_m4(prim.model);
    for (int k = 0; k < 3; k++) {
        const uint32_t idx = g.indices[prim.first_index + 3 * tri + k];
        const float* p = g.vertex_positions + 3 * ((int64_t)prim.vertex_offset + idx);
        const F4 world = mul(model, F4{F(p[0]), F(p[1]), F(p[2]), F(1.0f)});
        const F4 c = to_clip(world);
        poly[k].c[0] = c.x; poly[k].c[1] = c.y; poly[k].c[2] = c.z; poly[k].c[3] = c.w;
        for (int j = 0; j < 3; j++) poly[k].bary[j] = F(j == k ? 1.0f : 0.0f);
        for (int j = 0; j < 4; j++)
            if (!std::isfinite(poly[k].c[j].v)) { st.w[2]++; return; }
    }
    const int n = clip_polygon(poly, 3, clip_depth ? 0 : 2);
    if (n == 0) { st.w[1]++; return; }
    RasterVertex rv[12];
    for (int i = 0; i < n; i++) rv[i] = to_window(poly[i], W, H);
    const bool cull_back = prim.type == SAH_PRIMITIVE_TYPE_SOLID;
    for (int i = 1; i + 1 < n; i++) raster_triangle(rv[0], rv[i], rv[i + 1], cull_back, W, H, st, [&](const Fragment& f) { emit(f, (uint32_t)(i - 1)); });
}

void write_stats(uint32_t* out, const Stats& st) {
    if (out) std::memcpy(out, st.w, sizeof(st.w));
}

bool geometry_ok(const sah_scene_geometry* g, bool need_attributes) {
    if (!g || (g->num_primitives && !g->primitives)) return false;
    if (need_attributes && g->num_primitives && (!g->vertex_data || !g->materials)) return false;
    if (need_attributes && g->material_textures) {
        if (g->num_textures && !g->textures) return false;
        for (uint32_t t = 0; t < g->num_textures; t++)
            if (!texture_ok(g->textures[t])) return false;
        for (uint32_t m = 0; m < g->num_materials; m++) {
            const sah_material_textures& mt = g->material_textures[m];
            for (uint32_t idx : {mt.base_color, mt.normal, mt.data, mt.emission})
                if (idx != SAH_TEXTURE_NONE && idx >= g->num_textures) return false;
        }
    }
    for (uint32_t p = 0; p < g->num_primitives; p++) {
        const sah_primitive& pr = g->primitives[p];
        if (pr.index_count % 3 || (uint64_t)pr.first_index + pr.index_count > g->num_indices) return false;
        if (pr.type > SAH_PRIMITIVE_TYPE_CUTOUT) return false;
        if (need_attributes && pr.material >= g->num_materials) return false;
        for (uint32_t i = 0; i < pr.index_count; i++) {
            const int64_t v = (int64_t)pr.vertex_offset + g->indices[pr.first_index + i];
            if (v < 0 || v >= (int64_t)g->num_vertices) return false;
        }
    }
    return true;
}

// ---- G-buffer fragment stage (gltf_basic_pbr.slang:169-253, SAH_MAIN_VIEW) -----------------------------------------------------
struct VertexOut {  // the half-precision varyings of VertexOutput (:83-96), held as fp16-representable fp32
    H color[4];
    H normal[3];
    H tangent[4];
    F uv[2];  // float2 texcoord
};

VertexOut vertex_outputs(const sah_scene_geometry& g, const sah_primitive& prim, uint32_t index) {
    const sah_vertex_data& vd = g.vertex_data[(int64_t)prim.vertex_offset + index];
    const M4 model = load_m4(prim.model);
    VertexOut o;
    for (int k = 0; k < 4; k++) o.color[k] = H(unorm8_to_float((uint8_t)(vd.color >> (8 * k))));
    // (float3x3)model * v, then normalize in fp32, then the (half3) conversion (:139-141)
    auto rotate = [&](const float* v) {
        F3 r;
        F* out[3] = {&r.x, &r.y, &r.z};
        for (int i = 0; i < 3; i++) *out[i] = F(model.m[0 + i]) * F(v[0]) + F(model.m[4 + i]) * F(v[1]) + F(model.m[8 + i]) * F(v[2]);
        return normalize(r);
    };
    const F3 n = rotate(vd.normal), t = rotate(vd.tangent);
    o.normal[0] = H(n.x.v); o.normal[1] = H(n.y.v); o.normal[2] = H(n.z.v);
    o.tangent[0] = H(t.x.v); o.tangent[1] = H(t.y.v); o.tangent[2] = H(t.z.v);
    o.tangent[3] = H(vd.tangent[3]);
    o.uv[0] = F(vd.texcoord[0]);
    o.uv[1] = F(vd.texcoord[1]);
    return o;
}

H interpolate(const F lambda[3], H a, H b, H c) { return H((lambda[0] * F(a.v) + lambda[1] * F(b.v) + lambda[2] * F(c.v)).v); }

// `(half4)textures[index].SampleBias(vertex.texcoord, mip_bias)` of one material slot (0 base colour, 1 normal, 2 data, 3 emission), or the
// material's constant texel when the slot has no texture
struct TexEnv {
    const sah_scene_geometry* g;
    uint32_t material;
    float shader_bias;
};
void material_texel(const TexEnv& env, int slot, const float constant[4], const VertexOut vo[3], const Fragment& f, H out[4]) {
    uint32_t index = SAH_TEXTURE_NONE;
    if (env.g->material_textures) {
        const sah_material_textures& mt = env.g->material_textures[env.material];
        index = slot == 0 ? mt.base_color : slot == 1 ? mt.normal : slot == 2 ? mt.data : mt.emission;
    }
    if (index == SAH_TEXTURE_NONE) {
        for (int k = 0; k < 4; k++) out[k] = H(constant[k]);
        return;
    }
    auto texcoord = [&](const F lambda[3], float t[2]) {
        for (int c = 0; c < 2; c++) t[c] = (lambda[0] * vo[0].uv[c] + lambda[1] * vo[1].uv[c] + lambda[2] * vo[2].uv[c]).v;
    };
    float t[2], tx[2], ty[2], ddx[2], ddy[2];
    texcoord(f.lambda, t);
    texcoord(f.lambda_qx, tx);
    texcoord(f.lambda_qy, ty);
    for (int c = 0; c < 2; c++) {  // fine derivatives: odd pixel minus even pixel of the quad
        ddx[c] = (f.x & 1) ? (F(t[c]) - F(tx[c])).v : (F(tx[c]) - F(t[c])).v;
        ddy[c] = (f.y & 1) ? (F(t[c]) - F(ty[c])).v : (F(ty[c]) - F(t[c])).v;
    }
    float texel[4];
    sample_texture(env.g->textures[index], t, ddx, ddy, env.shader_bias, texel);
    for (int k = 0; k < 4; k++) out[k] = H(texel[k]);
}

struct GbufferTexel {
    uint8_t color[4];
    uint16_t normal[4];
    uint8_t data[4];
    uint8_t emission[4];
    bool discarded;
};

uint8_t srgb8(H v) { return float_to_unorm8(linear_to_srgb_f(v.v)); }

GbufferTexel shade_fragment(const TexEnv& env, const sah_material& m, const VertexOut vo[3], const Fragment& f) {
    GbufferTexel out{};
    const F* lambda = f.lambda;
    H base_texel[4], normal_texel[4], data_texel[4], emission_texel[4];
    material_texel(env, 0, m.base_color_texel, vo, f, base_texel);
    material_texel(env, 1, m.normal_texel, vo, f, normal_texel);
    material_texel(env, 2, m.data_texel, vo, f, data_texel);
    material_texel(env, 3, m.emission_texel, vo, f, emission_texel);
    H color[4], normal[3], tangent[4];
    for (int k = 0; k < 4; k++) color[k] = interpolate(lambda, vo[0].color[k], vo[1].color[k], vo[2].color[k]);
    for (int k = 0; k < 3; k++) normal[k] = interpolate(lambda, vo[0].normal[k], vo[1].normal[k], vo[2].normal[k]);
    for (int k = 0; k < 4; k++) tangent[k] = interpolate(lambda, vo[0].tangent[k], vo[1].tangent[k], vo[2].tangent[k]);
    // base colour (:181-189)
    H tinted[4];
    for (int k = 0; k < 4; k++) tinted[k] = base_texel[k] * color[k] * H(m.base_color_tint[k]);
    out.discarded = tinted[3].v <= m.opacity_threshold;
    // normals (:197-207)
    const H3 N{normal[0], normal[1], normal[2]}, T{tangent[0], tangent[1], tangent[2]};
    const H3 B = cross(N, T) * tangent[3];
    H ns[3];
    for (int k = 0; k < 3; k++) ns[k] = normal_texel[k] * H(2.0f) - H(1.0f);
    const H nx = ns[0] * T.x + ns[1] * B.x + ns[2] * N.x;
    const H ny = ns[0] * T.y + ns[1] * B.y + ns[2] * N.y;
    const H nz = ns[0] * T.z + ns[1] * B.z + ns[2] * N.z;
    // data (:213-218) and emission (:221-226)
    const H factor[4] = {H(0.0f), H(m.roughness_factor), H(m.metalness_factor), H(0.0f)};
    for (int k = 0; k < 4; k++) {
        const H d = data_texel[k] * factor[k];
        const H e = emission_texel[k] * H(m.emission_factor[k]);
        out.data[k] = float_to_unorm8(d.v);
        out.emission[k] = k < 3 ? srgb8(e) : float_to_unorm8(e.v);
        out.color[k] = k < 3 ? srgb8(tinted[k]) : float_to_unorm8(tinted[k].v);
    }
    out.normal[0] = f32_to_f16(nx.v); out.normal[1] = f32_to_f16(ny.v); out.normal[2] = f32_to_f16(nz.v); out.normal[3] = 0;
    return out;
}

// Draw order of a pass: every SOLID primitive, then every CUTOUT one, each class in list order (RenderScene::draw_opaque followed by
// draw_masked: gbuffer_phase.cpp:91-93, depth_culling_phase.cpp:473-476, directional_light.cpp:318-321, light_propagation_volume.cpp:611-613)
std::vector<uint32_t> draw_order(const sah_scene_geometry& g) {
    std::vector<uint32_t> order;
    for (uint32_t type : {(uint32_t)SAH_PRIMITIVE_TYPE_SOLID, (uint32_t)SAH_PRIMITIVE_TYPE_CUTOUT})
        for (uint32_t p = 0; p < g.num_primitives; p++)
            if (g.primitives[p].type == type) order.push_back(p);
    return order;
}

// tinted_base_color.a of the *_masked fragment stages (gltf_basic_pbr.slang:181-196): texel.a * vertex colour.a * tint.a in half
bool alpha_discarded(const TexEnv& env, const sah_material& m, const VertexOut vo[3], const Fragment& f) {
    H base_texel[4];
    material_texel(env, 0, m.base_color_texel, vo, f, base_texel);
    const H a = base_texel[3] * interpolate(f.lambda, vo[0].color[3], vo[1].color[3], vo[2].color[3]) * H(m.base_color_tint[3]);
    return a.v <= m.opacity_threshold;
}

bool plane_is(const sah_plane& p, uint32_t fmt, uint32_t w, uint32_t h) { return p.ptr && p.format == fmt && p.width == w && p.height == h; }

// ---- RSM fragment stage (gltf_basic_pbr.slang:169-253, SAH_RSM) -------------------------------------------------------------------
// gbuffer.normal = vertex.normal (no normal map outside SAH_MAIN_VIEW); gbuffer.data stays 0 in this variant (its block is compiled
// out, :210-227), so the surface has metalness 0 and roughness 0; flux = Fd(surface, -sun direction, surface.normal).
struct RsmTexel {
    uint8_t flux[4], normal[4];
    bool discarded;
};
RsmTexel shade_rsm_fragment(const TexEnv& env, const sah_material& m, const VertexOut vo[3], const Fragment& f, const float sun_direction[3]) {
    RsmTexel out{};
    const F* lambda = f.lambda;
    H base_texel[4];
    material_texel(env, 0, m.base_color_texel, vo, f, base_texel);
    H color[4], normal[3];
    for (int k = 0; k < 4; k++) color[k] = interpolate(lambda, vo[0].color[k], vo[1].color[k], vo[2].color[k]);
    for (int k = 0; k < 3; k++) normal[k] = interpolate(lambda, vo[0].normal[k], vo[1].normal[k], vo[2].normal[k]);
    H tinted[4];
    for (int k = 0; k < 4; k++) tinted[k] = base_texel[k] * color[k] * H(m.base_color_tint[k]);
    out.discarded = tinted[3].v <= m.opacity_threshold;
    Surface<H> s;
    s.base_color = {tinted[0], tinted[1], tinted[2]};
    s.normal = {normal[0], normal[1], normal[2]};
    s.metalness = H(0.0f);
    s.roughness = H(0.0f);
    const H3 l{-H(sun_direction[0]), -H(sun_direction[1]), -H(sun_direction[2])};
    const H3 flux = Fd(s, l, s.normal);
    out.flux[0] = srgb8(flux.x); out.flux[1] = srgb8(flux.y); out.flux[2] = srgb8(flux.z); out.flux[3] = 255;
    for (int k = 0; k < 3; k++) out.normal[k] = float_to_unorm8((normal[k] * H(0.5f) + H(0.5f)).v);
    out.normal[3] = 255;
    return out;
}

}  // namespace
}  // namespace orc

extern "C" {

int orc_shadow_render(const sah_scene_geometry* scene, const sah_sun_light_constants* sun, uint32_t num_cascades, const sah_volume* shadowmap,
                      uint32_t* stats) {
    using namespace orc;
    if (!geometry_ok(scene, false) || !sun || !shadowmap || !shadowmap->ptr || shadowmap->format != SAH_FORMAT_D16_UNORM || num_cascades == 0 ||
        num_cascades > 4 || shadowmap->depth < num_cascades || shadowmap->width > 8192 || shadowmap->height > 8192)
        return SAH_ERR_INVALID_ARGUMENT;
    for (uint32_t p = 0; p < scene->num_primitives; p++)  // masked geometry needs its vertex colours and material for the alpha test
        if (scene->primitives[p].type == SAH_PRIMITIVE_TYPE_CUTOUT && !geometry_ok(scene, true)) return SAH_ERR_INVALID_ARGUMENT;
    const uint32_t W = shadowmap->width, H = shadowmap->height;
    Stats st;
    for (uint32_t layer = 0; layer < num_cascades; layer++) {
        uint8_t* base = (uint8_t*)shadowmap->ptr + (size_t)layer * shadowmap->slice_pitch_bytes;
        for (uint32_t y = 0; y < H; y++) {
            uint16_t* row = (uint16_t*)(base + (size_t)y * shadowmap->row_pitch_bytes);
            for (uint32_t x = 0; x < W; x++) row[x] = 0xffffu;  // clear value 1.0 (directional_light.cpp:311)
        }
        const M4 world_to_ndc = load_m4(sun->cascade_matrices[layer]);
        for (uint32_t p = 0; p < scene->num_primitives; p++) {  // depth only, compare LESS: the image does not depend on the draw order
            const sah_primitive& prim = scene->primitives[p];
            const bool masked = prim.type == SAH_PRIMITIVE_TYPE_CUTOUT;  // shadow_masked_pso: SAH_DEPTH_ONLY + SAH_MASKED fragment stage
            for (uint32_t tri = 0; tri < prim.index_count / 3; tri++) {
                VertexOut vo[3];
                if (masked)
                    for (int k = 0; k < 3; k++) vo[k] = vertex_outputs(*scene, prim, scene->indices[prim.first_index + 3 * tri + k]);
                process_triangle(*scene, prim, tri, /*clip_depth=*/false, W, H, st, [&](F4 world) { return mul(world_to_ndc, world); },
                                 [&](const Fragment& f, uint32_t) {
                                     if (masked && alpha_discarded(TexEnv{scene, prim.material, 0.0f}, scene->materials[prim.material], vo, f)) return;
                                     const F z = nclamp(f.z, F(0.0f), F(1.0f));  // depth clamp; NaN -> 0
                                     const uint32_t code = (uint32_t)std::nearbyint(z.v * 65535.0f);
                                     uint16_t* texel = (uint16_t*)(base + (size_t)f.y * shadowmap->row_pitch_bytes) + f.x;
                                     if (code < *texel) *texel = (uint16_t)code;  // VK_COMPARE_OP_LESS
                                 });
            }
        }
    }
    write_stats(stats, st);
    return SAH_OK;
}

int orc_gbuffer_render(const sah_scene_geometry* scene, const sah_view_data* view, const sah_gbuffer* out, uint32_t* stats) {
    using namespace orc;
    if (!geometry_ok(scene, true) || !view || !out) return SAH_ERR_INVALID_ARGUMENT;
    const uint32_t W = out->depth.width, H = out->depth.height;
    if (W == 0 || H == 0 || W > 8192 || H > 8192 || !plane_is(out->depth, SAH_FORMAT_D32_SFLOAT, W, H) || !plane_is(out->color, SAH_FORMAT_R8G8B8A8_SRGB, W, H) ||
        !plane_is(out->normals, SAH_FORMAT_R16G16B16A16_SFLOAT, W, H) || !plane_is(out->data, SAH_FORMAT_R8G8B8A8_UNORM, W, H) ||
        !plane_is(out->emission, SAH_FORMAT_R8G8B8A8_SRGB, W, H))
        return SAH_ERR_INVALID_ARGUMENT;
    auto at = [&](const sah_plane& p, int x, int y, int bpp) { return (uint8_t*)p.ptr + (size_t)y * p.row_pitch_bytes + (size_t)x * bpp; };
    // clear values: gbuffer_phase.cpp:66-87; depth 0 = far plane of the reversed-Z projection
    const uint16_t clear_normal[4] = {f32_to_f16(0.5f), f32_to_f16(0.5f), f32_to_f16(1.0f), 0};
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            std::memset(at(out->color, x, y, 4), 0, 4);
            std::memcpy(at(out->normals, x, y, 8), clear_normal, 8);
            std::memset(at(out->data, x, y, 4), 0, 4);
            std::memset(at(out->emission, x, y, 4), 0, 4);
            std::memset(at(out->depth, x, y, 4), 0, 4);
        }
    const M4 V = load_m4(view->view), P = load_m4(view->projection);
    Stats st, st_second;
    const std::vector<uint32_t> order = draw_order(*scene);
    // Two passes over the same draws, as the reference makes them: the depth pre-pass (depth_culling_phase.cpp:455-481: compare GREATER,
    // depth writes on, masked geometry alpha-tested) and the G-buffer pass (gbuffer_phase.cpp:27-97 with material_pipelines.cpp
    // gbuffer_pso / gbuffer_masked_pso: compare EQUAL, depth writes off), in which EVERY fragment at the settled depth overwrites the
    // colour targets — the last one in draw order stays.
    for (int pass = 0; pass < 2; pass++)
        for (uint32_t p : order) {
            const sah_primitive& prim = scene->primitives[p];
            const sah_material& mat = scene->materials[prim.material];
            for (uint32_t tri = 0; tri < prim.index_count / 3; tri++) {
                VertexOut vo[3];
                for (int k = 0; k < 3; k++) vo[k] = vertex_outputs(*scene, prim, scene->indices[prim.first_index + 3 * tri + k]);
                process_triangle(*scene, prim, tri, /*clip_depth=*/true, W, H, pass == 0 ? st : st_second, [&](F4 world) { return mul(P, mul(V, world)); },
                                 [&](const Fragment& f, uint32_t) {
                                     const F z = nclamp(f.z, F(0.0f), F(1.0f));
                                     float stored;
                                     std::memcpy(&stored, at(out->depth, f.x, f.y, 4), 4);
                                     if (pass == 0 ? !(z.v > stored) : !(z.v == stored && z.v > 0.0f)) return;  // GREATER / EQUAL (never the cleared 0)
                                     const TexEnv env{scene, prim.material, view->material_texture_mip_bias};
                                     if (prim.type == SAH_PRIMITIVE_TYPE_CUTOUT && alpha_discarded(env, mat, vo, f)) return;
                                     if (pass == 0) {
                                         std::memcpy(at(out->depth, f.x, f.y, 4), &z.v, 4);
                                         return;
                                     }
                                     const GbufferTexel t = shade_fragment(env, mat, vo, f);
                                     std::memcpy(at(out->color, f.x, f.y, 4), t.color, 4);
                                     std::memcpy(at(out->normals, f.x, f.y, 8), t.normal, 8);
                                     std::memcpy(at(out->data, f.x, f.y, 4), t.data, 4);
                                     std::memcpy(at(out->emission, f.x, f.y, 4), t.emission, 4);
                                 });
            }
        }
    write_stats(stats, st);
    return SAH_OK;
}

int orc_rsm_render(const sah_scene_geometry* scene, const sah_sun_light_constants* sun, const sah_lpv_cascade_matrices* cascades, uint32_t num_cascades,
                   const sah_rsm_targets* rsm, uint32_t* stats) {
    using namespace orc;
    if (!geometry_ok(scene, true) || !sun || !cascades || !rsm || num_cascades == 0 || num_cascades > 4) return SAH_ERR_INVALID_ARGUMENT;
    const uint32_t W = rsm->depth.width, H = rsm->depth.height;
    if (W == 0 || H == 0 || W > 8192 || H > 8192 || rsm->depth.format != SAH_FORMAT_D16_UNORM || rsm->flux.format != SAH_FORMAT_R8G8B8A8_SRGB ||
        rsm->normals.format != SAH_FORMAT_R8G8B8A8_UNORM || rsm->flux.width != W || rsm->flux.height != H || rsm->normals.width != W ||
        rsm->normals.height != H || rsm->depth.depth < num_cascades || rsm->flux.depth < num_cascades || rsm->normals.depth < num_cascades)
        return SAH_ERR_INVALID_ARGUMENT;
    auto at = [&](const sah_volume& v, uint32_t layer, int x, int y, int bpp) {
        return (uint8_t*)v.ptr + (size_t)layer * v.slice_pitch_bytes + (size_t)y * v.row_pitch_bytes + (size_t)x * bpp;
    };
    Stats st;
    for (uint32_t layer = 0; layer < num_cascades; layer++) {
        const uint8_t clear_normal[4] = {128, 128, 255, 0};  // (0.5, 0.5, 1, 0), light_propagation_volume.cpp:596-600
        for (uint32_t y = 0; y < H; y++)
            for (uint32_t x = 0; x < W; x++) {
                std::memset(at(rsm->flux, layer, x, y, 4), 0, 4);
                std::memcpy(at(rsm->normals, layer, x, y, 4), clear_normal, 4);
                const uint16_t one = 0xffffu;
                std::memcpy(at(rsm->depth, layer, x, y, 2), &one, 2);
            }
        const M4 rsm_vp = load_m4(cascades[layer].rsm_vp);
        for (uint32_t p : draw_order(*scene)) {  // one pass, LESS with depth writes: the first of equal codes in draw order stays
            const sah_primitive& prim = scene->primitives[p];
            const sah_material& mat = scene->materials[prim.material];
            for (uint32_t tri = 0; tri < prim.index_count / 3; tri++) {
                VertexOut vo[3];
                for (int k = 0; k < 3; k++) vo[k] = vertex_outputs(*scene, prim, scene->indices[prim.first_index + 3 * tri + k]);
                process_triangle(*scene, prim, tri, /*clip_depth=*/true, W, H, st, [&](F4 world) { return mul(rsm_vp, world); },
                                 [&](const Fragment& f, uint32_t) {
                                     const F z = nclamp(f.z, F(0.0f), F(1.0f));
                                     const uint32_t code = (uint32_t)std::nearbyint(z.v * 65535.0f);
                                     uint16_t stored;
                                     std::memcpy(&stored, at(rsm->depth, layer, f.x, f.y, 2), 2);
                                     if (!(code < stored)) return;  // VK_COMPARE_OP_LESS on the D16 code: the first of equal codes stays
                                     const RsmTexel t = shade_rsm_fragment(TexEnv{scene, prim.material, 0.0f}, mat, vo, f, sun->direction_and_tan_size);
                                     if (prim.type == SAH_PRIMITIVE_TYPE_CUTOUT && t.discarded) return;
                                     const uint16_t c16 = (uint16_t)code;
                                     std::memcpy(at(rsm->depth, layer, f.x, f.y, 2), &c16, 2);
                                     std::memcpy(at(rsm->flux, layer, f.x, f.y, 4), t.flux, 4);
                                     std::memcpy(at(rsm->normals, layer, f.x, f.y, 4), t.normal, 4);
                                 });
            }
        }
    }
    write_stats(stats, st);
    return SAH_OK;
}

}  // extern "C"
