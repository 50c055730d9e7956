// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// CPU restatement of the three GI overlays.
#include "gi.hpp"

#include <cmath>

#include "brdf.hpp"

namespace orc {

// =====================================================================================================
// a3 — LPV gather: RenderCore/shaders/gi/lpv/overlay.frag:70-164
// =====================================================================================================

// common/spherical_harmonics.glsl:28-35,73-82 (dir_to_sh == dir_to_sh_ericpolman)
static inline void dir_to_sh(F3 dir, F out[4]) {
    const F c0 = F(0.282094792f), c1 = F(0.488602512f);
    out[0] = c0;
    out[1] = -c1 * dir.y;
    out[2] = c1 * dir.z;
    out[3] = -c1 * dir.x;
}

static inline F dot4(const Texel& t, const F n[4]) { return F(t.c[0]) * n[0] + F(t.c[1]) * n[1] + F(t.c[2]) * n[2] + F(t.c[3]) * n[3]; }

// sampler: linear, CLAMP_TO_BORDER, transparent black (light_propagation_volume.cpp:99-113)
static inline void lpv_fetch(const sah_gi& gi, F4 p, const F n[4], F out[3]) {
    Texel r = sample_trilinear(img3d(gi.lpv_red), p.x.v, p.y.v, p.z.v, ADDR_CLAMP_TO_BORDER);
    Texel g = sample_trilinear(img3d(gi.lpv_green), p.x.v, p.y.v, p.z.v, ADDR_CLAMP_TO_BORDER);
    Texel b = sample_trilinear(img3d(gi.lpv_blue), p.x.v, p.y.v, p.z.v, ADDR_CLAMP_TO_BORDER);
    out[0] = dot4(r, n);
    out[1] = dot4(g, n);
    out[2] = dot4(b, n);
}

bool gi_lpv_frag(const sah_lighting_desc& d, int x, int y, float depth, const Texel& color, const Texel& normal_t, const Texel& data,
                 F out[4]) {
    if (depth == 0.f) return false;  // overlay.frag:73-76
    const sah_gi& gi = *d.gi;
    const sah_view_data& view = *d.view;
    const int num_cascades = (int)gi.lpv_num_cascades;

    Surface<F> s;
    s.base_color = {F(color.c[0]), F(color.c[1]), F(color.c[2])};
    s.normal = normalize(F3{F(normal_t.c[0]), F(normal_t.c[1]), F(normal_t.c[2])});
    s.metalness = F(data.c[2]);
    s.roughness = F(data.c[1]);

    F3 vs = viewspace_position_glsl(view, x, y, depth);
    F4 ws = mul(mat(view.inverse_view), F4{vs.x, vs.y, vs.z, F(1.0f)});
    F3 view_position = {F(-view.view[12]), F(-view.view[13]), F(-view.view[14])};
    F3 V = normalize(F3{ws.x, ws.y, ws.z} - view_position);

    // overlay.frag:97-103
    uint32_t selected = 0;
    for (int i = num_cascades - 1; i >= 0; i--) {
        F4 cp = mul(mat(gi.lpv_cascades[i].world_to_cascade), ws);
        if (cp.x.v > 0.f && cp.y.v > 0.f && cp.z.v > 0.f && cp.x.v < 1.f && cp.y.v < 1.f && cp.z.v < 1.f) selected = (uint32_t)i;
    }

    // :105-107
    F3 lpv_normal = -s.normal;
    lpv_normal.x = lpv_normal.x * F(-1.0f);
    F nc[4];
    dir_to_sh(lpv_normal, nc);

    // :110-111 + sample_light_from_cascade :53-68
    F indirect[3];
    {
        F4 pos = {ws.x + s.normal.x, ws.y + s.normal.y, ws.z + s.normal.z, ws.w + F(0.f)};
        F4 cp = mul(mat(gi.lpv_cascades[selected].world_to_cascade), pos);
        cp.x = cp.x + F((float)selected);
        cp.x = cp.x / F((float)gi.lpv_num_cascades);
        lpv_fetch(gi, cp, nc, indirect);
    }

    // :113-147
    F3 I = -V;
    F3 refl = I - s.normal * (F(2.0f) * dot(s.normal, I));  // reflect(I, N) = I - 2 * dot(N, I) * N
    F spec[3] = {F(0.f), F(0.f), F(0.f)};
    if (selected == 0) {
        F4 cp = mul(mat(gi.lpv_cascades[0].world_to_cascade), ws);
        F rc[4];
        dir_to_sh(refl, rc);
        lpv_fetch(gi, cp, rc, spec);  // note: no cascade-atlas remap of x here (quirk, :116-120)
        {
            F3 loc = F3{ws.x, ws.y, ws.z} + refl * F(1.0f);
            F4 cp1 = mul(mat(gi.lpv_cascades[0].world_to_cascade), F4{loc.x, loc.y, loc.z, F(1.f)});
            F more[3];
            lpv_fetch(gi, cp1, rc, more);
            for (int i = 0; i < 3; i++) spec[i] = spec[i] + more[i];
        }
        for (int i = 0; i < 3; i++) spec[i] = spec[i] / F(2.0f);
    }

    // :149-151
    F3 diffuse_factor = Fd(s, s.normal, s.normal);
    F3 fr = Fr(s, s.normal, refl);
    F3 specular_factor = {fr.x * F(0.f), fr.y * F(0.f), fr.z * F(0.f)};

    // :153-155
    F ao = F(1.0f);
    if (d.ao && d.ao->ptr) ao = F(load_texel(img2d(*d.ao), x, y, 0).c[0]);
    F3 total = {indirect[0] * diffuse_factor.x * ao + spec[0] * specular_factor.x, indirect[1] * diffuse_factor.y * ao + spec[1] * specular_factor.y,
                indirect[2] * diffuse_factor.z * ao + spec[2] * specular_factor.z};
    if (any_nan(total)) total = F3(F(0.f));  // :159-161
    F exposure = F(gi.lpv_exposure);
    out[0] = total.x * exposure;
    out[1] = total.y * exposure;
    out[2] = total.z * exposure;
    out[3] = F(1.0f);
    return true;
}

// =====================================================================================================
// a4 — irradiance-cache gather: RenderCore/shaders/gi/cache/overlay.frag.slang:46-118,
//      probe_sampling.slangi:6-106, common/octahedral.slangi:56-74
// =====================================================================================================

struct F2 {
    F x, y;
};

// octahedral.slangi:56-63
static inline F2 octahedral_coordinates(F3 dir) {
    F l1 = nabs(dir.x) + nabs(dir.y) + nabs(dir.z);
    F inv = F(1.f) / l1;
    F2 uv = {dir.x * inv, dir.y * inv};
    if (dir.z.v < 0.f) {
        F sx = F(uv.x.v >= 0.f ? 1.f : -1.f), sy = F(uv.y.v >= 0.f ? 1.f : -1.f);
        F2 r = {(F(1.f) - nabs(uv.y)) * sx, (F(1.f) - nabs(uv.x)) * sy};
        uv = r;
    }
    return uv;
}

// octahedral.slangi:65-74
static inline void probe_uv(const uint32_t idx[3], F2 oct, const uint32_t n[2], F uv[2]) {
    F total[2] = {F((float)n[0]) + F(2.f), F((float)n[1]) + F(2.f)};
    F tex_size[2] = {total[0] * F(32.f), total[1] * F(32.f)};
    F o[2] = {oct.x, oct.y};
    for (int i = 0; i < 2; i++) {
        F u = F((float)idx[i]) * total[i] + total[i] * F(0.5f);
        u = u + o[i] * (F((float)n[i]) * F(0.5f));
        uv[i] = u / tex_size[i];
    }
}

// float -> uint conversion as the hardware does it: NaN and negatives -> 0, saturating.
static inline uint32_t f2uint(float f) {
    if (!(f > 0.f)) return 0u;
    if (f >= 4294967296.f) return 0xffffffffu;
    return (uint32_t)f;
}

// array layer selection: round to nearest even, clamp to [0, layers-1]
static inline int array_layer(float l, uint32_t layers) {
    float r = std::nearbyint(l);
    if (!(r > 0.f)) return 0;
    if (r > (float)(layers - 1)) return (int)layers - 1;
    return (int)r;
}

// probe_sampling.slangi:6-106; atlases sampled with linear + REPEAT (irradiance_cache.cpp:205-217)
static F3 sample_cascade(const sah_gi& gi, F3 location, F3 direction, uint32_t cascade_index) {
    const sah_probe_cascade& c = gi.probe_cascades[cascade_index];
    const F spacing = F(c.probe_spacing);
    const F3 rel = location - F3{F(c.min[0]), F(c.min[1]), F(c.min[2])};
    const F3 ps = rel / spacing;
    const F3 min_probe = {F(std::floor(ps.x.v)), F(std::floor(ps.y.v)), F(std::floor(ps.z.v))};
    const F3 alpha = {nclamp(ps.x - min_probe.x, F(0.f), F(1.f)), nclamp(ps.y - min_probe.y, F(0.f), F(1.f)),
                      nclamp(ps.z - min_probe.z, F(0.f), F(1.f))};
    const Image irr_img = img3d(gi.probe_irradiance), depth_img = img3d(gi.probe_depth), val_img = img3d(gi.probe_validity);

    F3 irradiance = F3(F(0.f));
    F weight = F(0.f);
    for (uint32_t i = 0; i < 8; i++) {
        const F3 off = {F((float)(i & 1)), F((float)((i >> 1) & 1)), F((float)((i >> 2) & 1))};
        const F3 probe_location = min_probe + off;
        const F3 dir_to_probe = probe_location - ps;
        const F dist = length(dir_to_probe) * spacing;
        const F3 pidx_f = probe_location + F3{F(0.f), F((float)cascade_index) * F(8.f), F(0.f)};
        const uint32_t pidx[3] = {f2uint(pidx_f.x.v), f2uint(pidx_f.y.v), f2uint(pidx_f.z.v)};

        // Texture2DArray<half> probe_validity[uint3(x, y, layer)] — out of range loads return 0
        float validity = 0.f;
        if (pidx[0] < val_img.width && pidx[1] < val_img.height && pidx[2] < val_img.depth)
            validity = rh(load_texel(val_img, (int)pidx[0], (int)pidx[1], (int)pidx[2]).c[0]);
        if (validity == 0.f) continue;

        const F3 tri = {nmax(F(0.001f), mix(F(1.f) - alpha.x, alpha.x, off.x)), nmax(F(0.001f), mix(F(1.f) - alpha.y, alpha.y, off.y)),
                        nmax(F(0.001f), mix(F(1.f) - alpha.z, alpha.z, off.z))};
        const F trilinear_weight = tri.x * tri.y * tri.z;
        F probe_weight = F(1.f);

        const F2 depth_oct = octahedral_coordinates(-dir_to_probe);
        const uint32_t ten[2] = {10, 10};
        F duv[2];
        probe_uv(pidx, depth_oct, ten, duv);
        Texel dt = sample_bilinear(depth_img, duv[0].v, duv[1].v, array_layer((float)pidx[2], depth_img.depth), ADDR_REPEAT);
        const H dx = H(dt.c[0]), dy = H(dt.c[1]);  // Sampler2DArray<half2>
        const F variance = F(nabs(dx * dx - dy).v);

        F cheb = F(1.f);
        if (dist.v > dx.v) {
            const F v = dist - F(dx.v);
            cheb = variance / (variance + (v * v));
            cheb = nmax(cheb * cheb * cheb, F(0.f));
        }
        probe_weight = probe_weight * nmax(F(0.05f), cheb);
        probe_weight = nmax(F(0.000001f), probe_weight);
        const F crush = F(0.2f);
        if (probe_weight.v < crush.v) probe_weight = probe_weight * ((probe_weight * probe_weight) * (F(1.f) / (crush * crush)));
        probe_weight = probe_weight * trilinear_weight;

        const F2 irr_oct = octahedral_coordinates(direction);
        F iuv[2];
        probe_uv(pidx, irr_oct, gi.probe_size, iuv);
        Texel it = sample_bilinear(irr_img, iuv[0].v, iuv[1].v, array_layer((float)pidx[2], irr_img.depth), ADDR_REPEAT);
        const H3 pi = {H(it.c[0]), H(it.c[1]), H(it.c[2])};  // Sampler2DArray<half3>
        irradiance = irradiance + to_f(pi) * probe_weight;
        weight = weight + probe_weight;
    }
    if (weight.v == 0.f) return F3(F(0.f));
    irradiance = irradiance / weight;
    // `irradiance * 2 * PI` with PI = 3.1415927h (brdf.slangi:4-6 wins the #ifndef race) promoted to float
    return irradiance * F(2.f) * F(rh(3.1415927f));
}

bool gi_cache_frag(const sah_lighting_desc& d, int x, int y, float depth, const Texel& color, const Texel& normal_t,
                   const Texel& data, F out[4]) {
    if (depth == 0.f) return false;  // overlay.frag.slang:50-53
    const sah_gi& gi = *d.gi;
    const sah_view_data& view = *d.view;
    Surface<H> s;
    s.base_color = {H(color.c[0]), H(color.c[1]), H(color.c[2])};
    s.normal = normalize(H3{H(normal_t.c[0]), H(normal_t.c[1]), H(normal_t.c[2])});
    s.roughness = H(data.c[1]);
    s.metalness = H(data.c[2]);
    const F3 location = worldspace_location_slang(view, x, y, depth);
    const F3 view_position = {F(-view.view[12]), F(-view.view[13]), F(-view.view[14])};
    const H3 V = to_h(normalize(location - view_position));

    // :69-77
    uint32_t cascade_index = 5;
    for (uint32_t i = 0; i < 4; i++) {
        const sah_probe_cascade& c = gi.probe_cascades[i];
        const F3 cmin = {F(c.min[0]), F(c.min[1]), F(c.min[2])};
        const F3 cmax = cmin + F3{F(32.f), F(8.f), F(32.f)} * F(c.probe_spacing);
        if (location.x.v > cmin.x.v && location.y.v > cmin.y.v && location.z.v > cmin.z.v && location.x.v < cmax.x.v &&
            location.y.v < cmax.y.v && location.z.v < cmax.z.v) {
            cascade_index = i;
            break;
        }
    }
    if (cascade_index > 3) {  // :79-81 — returns (half4)0 and is still blended
        for (int i = 0; i < 4; i++) out[i] = F(0.f);
        return true;
    }
    const H3 irradiance = to_h(sample_cascade(gi, location, to_f(s.normal), cascade_index));
    const H3 b = brdf(s, s.normal, V);
    const H exposure = H::lit(0.314159);
    H3 c = b * irradiance * exposure;
    if (gi.cache_debug_mode == 1) {  // :100-112
        static const float dbg[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}};
        c = {H(dbg[cascade_index][0]), H(dbg[cascade_index][1]), H(dbg[cascade_index][2])};
    }
    if (any_nan(c)) c = H3(H(0.f));
    out[0] = F(c.x.v);
    out[1] = F(c.y.v);
    out[2] = F(c.z.v);
    out[3] = F(1.f);
    return true;
}

// =====================================================================================================
// a5 — RTGI reconstruction: RenderCore/shaders/gi/rtgi/overlay.frag.slang:68-117
// =====================================================================================================

// Out-of-range image loads return 0 (robustBufferAccess2 / robustImageAccess semantics).
static inline Texel load_or_zero(const Image& im, uint32_t x, uint32_t y) {
    if (x >= im.width || y >= im.height) return Texel{{0.f, 0.f, 0.f, 0.f}};
    return load_texel(im, (int)x, (int)y, 0);
}

static H3 rtgi_contribution(const Surface<H>& s, H3 V, H3 dir, H3 irr) {
    // overlay.frag.slang:62-66
    const H3 b = brdf(s, dir, V);
    const H ndotl = H(nclamp(F(dot(dir, s.normal).v), F(0.f), F(1.f)).v);  // dot in half, clamp(.,0,1) with int literals
    return b * irr * ndotl;
}

bool gi_rtgi_frag(const sah_lighting_desc& d, int x, int y, float depth, const Texel& color, const Texel& normal_t, const Texel& data,
                  F out[4]) {
    if (depth == 0.f) return false;
    const sah_gi& gi = *d.gi;
    const sah_view_data& view = *d.view;
    const Image depth_img = img2d(d.gbuffer->depth), ray_img = img2d(gi.ray_buffer), irr_img = img2d(gi.ray_irradiance);
    Surface<H> s;
    s.base_color = {H(color.c[0]), H(color.c[1]), H(color.c[2])};
    s.normal = normalize(H3{H(normal_t.c[0]), H(normal_t.c[1]), H(normal_t.c[2])});
    s.roughness = H(data.c[1]);
    s.metalness = H(data.c[2]);
    const F3 location = worldspace_location_slang(view, x, y, depth);
    const F3 view_position = {F(-view.view[12]), F(-view.view[13]), F(-view.view[14])};
    const H3 V = to_h(normalize(location - view_position));

    auto path = [&](uint32_t px, uint32_t py, H3& dir, H3& irr) {
        Texel r = load_or_zero(ray_img, px, py), i = load_or_zero(irr_img, px, py);
        dir = {H(r.c[0]), H(r.c[1]), H(r.c[2])};
        irr = {H(i.c[0]), H(i.c[1]), H(i.c[2])};
    };
    H3 dir, irr;
    path((uint32_t)x, (uint32_t)y, dir, irr);
    H3 radiance = rtgi_contribution(s, V, dir, irr);
    uint32_t num_samples = 1;

    for (uint32_t ray = 0; ray < gi.num_extra_rays; ray++) {
        // r1(n) :30-36 — PHI literal is a float in Slang
        const F phi = F(1.618033988749895f);
        const F q = F((float)ray) / phi;
        F r1x = F(2.f) + q, r1y = F(3.f) + q;
        r1x = r1x - F(std::floor(r1x.v));
        r1y = r1y - F(std::floor(r1y.v));
        const uint32_t nox = f2uint((r1x * F(128.f)).v), noy = f2uint((r1y * F(128.f)).v);
        // noise[(pixel + noise_offset) % 128] * 2.h - 1.h   (Texture2D<half2>)
        Texel nt = {{0.f, 0.f, 0.f, 0.f}};
        if (gi.noise.ptr) nt = load_or_zero(img2d(gi.noise), ((uint32_t)x + nox) % 128u, ((uint32_t)y + noy) % 128u);
        const H nsx = H(nt.c[0]) * H(2.f) - H(1.f), nsy = H(nt.c[1]) * H(2.f) - H(1.f);
        // (uint2)round(pixel + noise_sample * extra_ray_radius); round = round-half-even (documented choice)
        const F ox = F((float)x) + F(nsx.v) * F(gi.extra_ray_radius), oy = F((float)y) + F(nsy.v) * F(gi.extra_ray_radius);
        const uint32_t opx = f2uint(std::nearbyint(ox.v)), opy = f2uint(std::nearbyint(oy.v));
        const float odepth = load_or_zero(depth_img, opx, opy).c[0];
        // get_worldspace_location(offset_pixel) :38-47 (the offset pixel may be out of range: (float2)uint2)
        F tx = (F((float)opx) + F(0.5f)) / F(view.render_resolution[0]);
        F ty = (F((float)opy) + F(0.5f)) / F(view.render_resolution[1]);
        F4 ndc = {tx * F(2.0f) - F(1.0f), ty * F(2.0f) - F(1.0f), F(odepth), F(1.0f)};
        F4 vs = mul(mat(view.inverse_projection), ndc);
        vs = {vs.x / vs.w, vs.y / vs.w, vs.z / vs.w, vs.w / vs.w};
        F4 ws = mul(mat(view.inverse_view), vs);
        const F3 other_location = {ws.x, ws.y, ws.z};
        if (length(location - other_location).v > 2.f) continue;  // NaN compares false: not skipped (as in the shader)
        H3 d2, i2;
        path(opx, opy, d2, i2);
        radiance = radiance + rtgi_contribution(s, V, d2, i2);
        num_samples++;
    }
    if (any_nan(radiance)) radiance = H3(H(0.f));
    const H n = H((float)num_samples);
    out[0] = F((radiance.x / n).v);
    out[1] = F((radiance.y / n).v);
    out[2] = F((radiance.z / n).v);
    out[3] = F(1.f);
    return true;
}

// for the probe ray generator's misses (rt.cpp)
F3 sample_probe_cascade(const sah_gi& gi, F3 location, F3 direction, uint32_t cascade_index) { return sample_cascade(gi, location, direction, cascade_index); }

}  // namespace orc

// exports for the known-answer tests (SURVEY.md §8-c fixture i)
extern "C" void orc_dir_to_sh(const float* d3, float* out4) {
    orc::F o[4];
    orc::dir_to_sh(orc::F3{orc::F(d3[0]), orc::F(d3[1]), orc::F(d3[2])}, o);
    for (int i = 0; i < 4; i++) out4[i] = o[i].v;
}
extern "C" void orc_octahedral_coordinates(const float* d3, float* out2) {
    const orc::F2 uv = orc::octahedral_coordinates(orc::F3{orc::F(d3[0]), orc::F(d3[1]), orc::F(d3[2])});
    out2[0] = uv.x.v;
    out2[1] = uv.y.v;
}
extern "C" void orc_probe_uv(const uint32_t* idx3, const float* oct2, uint32_t n0, uint32_t n1, float* out2) {
    const uint32_t n[2] = {n0, n1};
    orc::F uv[2];
    orc::probe_uv(idx3, orc::F2{orc::F(oct2[0]), orc::F(oct2[1])}, n, uv);
    out2[0] = uv[0].v;
    out2[1] = uv[1].v;
}
