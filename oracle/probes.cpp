// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// CPU restatement of the irradiance-cache probe maintenance compute:
//   RenderCore/shaders/gi/cache/copy_cascades.comp.slang:22-99        (host render/gi/irradiance_cache.cpp:455-486)
//   RenderCore/shaders/gi/cache/probe_update.slangi:4-37              write_probe_texel_with_border
//   RenderCore/shaders/gi/cache/probe_depth_update.comp.slang:11-49
//   RenderCore/shaders/gi/cache/probe_light_cache_update.comp.slang:13-53
//   RenderCore/shaders/gi/cache/probe_rtgi_update.comp.slang:13-53
//   RenderCore/shaders/gi/cache/probe_finalize.comp.slang:13-74       (host irradiance_cache.cpp:585-724)
//   RenderCore/shaders/common/octahedral.slangi:18-54
// Quirks kept: interior texels are written WITHOUT the +1 border offset (probe_update.slangi:12) while every reader adds it;
// init_new_probe clears the depth atlas at light-cache offsets (copy_cascades.comp.slang:39-45); probe_finalize tests only texels
// 0 and 64 on every lane (:31-47) and averages with x = lane % 5, y = lane / 6 (:59-60); cosine weights are computed but unused
// in the depth and rtgi updates.
// Where the reference races (several invocations of one dispatch store to one texel) or leaves an order open (WaveActiveSum on
// half3) this file IS the definition (include/sah_hip.h): invocations run in ascending linear index, stores in
// program order, the wave sum adds in lane order in fp16; out-of-range stores are dropped, out-of-range loads return 0.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../include/sah_hip.h"
#include "image.hpp"
#include "math.hpp"

namespace orc {

Image img3d(const sah_volume& v);

namespace {

bool inside(const Image& im, int x, int y, int z) {
    return x >= 0 && y >= 0 && z >= 0 && (uint32_t)x < im.width && (uint32_t)y < im.height && (uint32_t)z < im.depth;
}
Texel load_or_zero(const Image& im, int x, int y, int z) {
    if (!inside(im, x, y, z)) return Texel{{0.f, 0.f, 0.f, 0.f}};
    return load_texel(im, x, y, z);
}
uint8_t* texel_ptr(const Image& im, int x, int y, int z) {
    return const_cast<uint8_t*>(im.ptr) + (size_t)z * im.slice_pitch + (size_t)y * im.row_pitch + (size_t)x * format_bpp(im.format);
}
void store_half3(const Image& im, int x, int y, int z, H3 v) {  // RWTexture2DArray<half3> on B10G11R11
    if (!inside(im, x, y, z)) return;
    const float f[3] = {v.x.v, v.y.v, v.z.v};
    const uint32_t p = r11g11b10_encode(f);
    std::memcpy(texel_ptr(im, x, y, z), &p, 4);
}
void store_half2(const Image& im, int x, int y, int z, H a, H b) {  // RWTexture2DArray<half2> on R16G16_SFLOAT
    if (!inside(im, x, y, z)) return;
    const uint16_t h[2] = {f32_to_f16(a.v), f32_to_f16(b.v)};
    std::memcpy(texel_ptr(im, x, y, z), h, 4);
}
void store_half1_unorm8(const Image& im, int x, int y, int z, H v) {  // RWTexture2DArray<half> on R8_UNORM
    if (!inside(im, x, y, z)) return;
    *texel_ptr(im, x, y, z) = float_to_unorm8(v.v);
}

int isign(int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }

// probe_update.slangi:4-37.  `store(x, y, z)` writes the invocation's value.
template <class Store> void write_probe_texel_with_border(int rx, int ry, const uint32_t probe_id[3], int tx, int ty, Store store) {
    const int bx = (int)probe_id[0] * (rx + 2), by = (int)probe_id[1] * (ry + 2), bz = (int)probe_id[2];
    const bool edge_x = tx == 0 || tx == rx - 1, edge_y = ty == 0 || ty == ry - 1;
    int mx = tx - rx / 2, my = ty - ry / 2;
    mx += mx >= 0 ? 1 : 0;
    my += my >= 0 ? 1 : 0;
    store(tx + bx, ty + by, bz);  // no +1: the interior lands on cells [0, res) of the (res + 2)-wide block
    if (edge_x && edge_y) {
        int dx = -mx, dy = -my;
        dx += mx >= 0 ? -1 : 0;
        dy += my >= 0 ? -1 : 0;
        dx += rx / 2;
        dy += ry / 2;
        store(dx + bx, dy + by, bz);
    }
    if (edge_x) {
        int ex = mx + isign(mx), ey = -my;
        ex += ex >= 0 ? -1 : 0;
        ey += ey >= 0 ? -1 : 0;
        ex += rx / 2;
        ey += ry / 2;
        store(ex + bx, ey + by, bz);
    }
    if (edge_y) {
        int ex = -mx, ey = my + isign(my);
        ex += ex >= 0 ? -1 : 0;
        ey += ey >= 0 ? -1 : 0;
        ex += rx / 2;
        ey += ry / 2;
        store(ex + bx, ey + by, bz);
    }
}

struct F2 {
    F x, y;
};
// octahedral.slangi:25-39
F2 normalized_octahedral_coordinates(uint32_t tx, uint32_t ty, uint32_t nx, uint32_t ny) {
    F cx = F((float)(tx % nx)), cy = F((float)(ty % ny));
    cx = cx + F(0.5f);
    cy = cy + F(0.5f);
    cx = cx / F((float)nx);
    cy = cy / F((float)ny);
    cx = cx * F(2.f);
    cy = cy * F(2.f);
    return {cx - F(1.f), cy - F(1.f)};
}
F sign_not_zero(F v) { return F(v.v >= 0.f ? 1.f : -1.f); }
// octahedral.slangi:44-50
F3 octahedral_direction(F2 c) {
    F3 d = {c.x, c.y, F(1.f) - nabs(c.x) - nabs(c.y)};
    if (d.z.v < 0.f) {
        const F nx = (F(1.f) - nabs(d.y)) * sign_not_zero(d.x);
        const F ny = (F(1.f) - nabs(d.x)) * sign_not_zero(d.y);
        d.x = nx;
        d.y = ny;
    }
    return normalize(d);
}

struct Atlases {
    Image rtgi, light_cache, depth, average, validity;
};
bool atlases_ok(const sah_probe_atlases* a, Atlases* out) {
    if (!a || !a->rtgi.ptr || !a->light_cache.ptr || !a->depth.ptr || !a->average.ptr || !a->validity.ptr) return false;
    if (a->rtgi.format != FMT_B10G11R11_UFLOAT || a->light_cache.format != FMT_B10G11R11_UFLOAT || a->average.format != FMT_B10G11R11_UFLOAT ||
        a->depth.format != FMT_R16G16_SFLOAT || a->validity.format != FMT_R8_UNORM)
        return false;
    out->rtgi = img3d(a->rtgi);
    out->light_cache = img3d(a->light_cache);
    out->depth = img3d(a->depth);
    out->average = img3d(a->average);
    out->validity = img3d(a->validity);
    return true;
}

}  // namespace

// direction of trace texel (tx, ty) of an n x n octahedral map, for the probe ray generator (rt.cpp)
F3 octahedral_texel_direction(uint32_t tx, uint32_t ty, uint32_t n) { return octahedral_direction(normalized_octahedral_coordinates(tx, ty, n, n)); }

}  // namespace orc

using namespace orc;

// copy_cascades.comp.slang:86-99, dispatch (8,8,8) x numthreads (4,4,4) = 32 x 32 x 32 cells
extern "C" int orc_probe_copy(const sah_probe_atlases* src_d, const sah_probe_atlases* dst_d, const float cascade_movement[4][3]) {
    Atlases s, d;
    if (!atlases_ok(src_d, &s) || !atlases_ok(dst_d, &d) || !cascade_movement) return SAH_ERR_INVALID_ARGUMENT;
    for (int z = 0; z < 32; z++) {
        for (int y = 0; y < 32; y++) {
            for (int x = 0; x < 32; x++) {
                const int cascade = y / 8;
                // (int3)movement truncates toward zero; a movement beyond the grid (or NaN, whose conversion is undefined) scrolls
                // the whole cascade out, which any value >= 32 expresses
                auto cells = [](float m) { return (m >= -64.f && m <= 64.f) ? (int)m : 64; };
                const int sx = x - cells(cascade_movement[cascade][0]), sy = y - cells(cascade_movement[cascade][1]),
                          sz = z - cells(cascade_movement[cascade][2]);
                const bool copy = sx >= 0 && sy >= 8 * cascade && sz >= 0 && sx < 32 && sy < 8 * (cascade + 1) && sz < 32;
                if (copy) {  // copy_from_cell :54-84
                    for (int j = 0; j < 8; j++)
                        for (int i = 0; i < 7; i++) {
                            const Texel t = load_or_zero(s.rtgi, sx * 7 + i, sy * 8 + j, sz);
                            store_half3(d.rtgi, x * 7 + i, y * 8 + j, z, H3{H(t.c[0]), H(t.c[1]), H(t.c[2])});
                        }
                    for (int j = 0; j < 13; j++)
                        for (int i = 0; i < 13; i++) {
                            const Texel t = load_or_zero(s.light_cache, sx * 13 + i, sy * 13 + j, sz);
                            store_half3(d.light_cache, x * 13 + i, y * 13 + j, z, H3{H(t.c[0]), H(t.c[1]), H(t.c[2])});
                        }
                    for (int j = 0; j < 12; j++)
                        for (int i = 0; i < 12; i++) {
                            const Texel t = load_or_zero(s.depth, sx * 12 + i, sy * 12 + j, sz);
                            store_half2(d.depth, x * 12 + i, y * 12 + j, z, H(t.c[0]), H(t.c[1]));
                        }
                    const Texel a = load_or_zero(s.average, sx, sy, sz);
                    store_half3(d.average, x, y, z, H3{H(a.c[0]), H(a.c[1]), H(a.c[2])});
                    const Texel v = load_or_zero(s.validity, sx, sy, sz);
                    store_half1_unorm8(d.validity, x, y, z, H(v.c[0]));
                } else {  // init_new_probe :22-52
                    const H3 zero3 = {H(0.f), H(0.f), H(0.f)};
                    for (int j = 0; j < 8; j++)
                        for (int i = 0; i < 7; i++) store_half3(d.rtgi, x * 7 + i, y * 8 + j, z, zero3);
                    for (int j = 0; j < 13; j++)
                        for (int i = 0; i < 13; i++) store_half3(d.light_cache, x * 13 + i, y * 13 + j, z, zero3);
                    for (int j = 0; j < 12; j++)  // depth_dest[light_cache_pixel + ...]: light-cache offsets (:43)
                        for (int i = 0; i < 12; i++) store_half2(d.depth, x * 13 + i, y * 13 + j, z, H(0.f), H(0.f));
                    store_half3(d.average, x, y, z, zero3);
                    store_half1_unorm8(d.validity, x, y, z, H(255.f));  // 0xff marker, saturates in R8_UNORM
                }
            }
        }
    }
    return SAH_OK;
}

// probes_to_update: host pointer here (the oracle runs on host memory)
extern "C" int orc_probe_update(const sah_probe_atlases* atl, const sah_volume* trace_results, const uint32_t* probes_to_update, uint32_t num_probes) {
    Atlases a;
    if (!atlases_ok(atl, &a) || !trace_results || !trace_results->ptr || trace_results->format != FMT_R16G16B16A16_SFLOAT) return SAH_ERR_INVALID_ARGUMENT;
    if (num_probes && !probes_to_update) return SAH_ERR_INVALID_ARGUMENT;
    const Image tr = img3d(*trace_results);

    // probe_depth_update.comp.slang:11-49, numthreads (10,10,1), one group per probe
    for (uint32_t p = 0; p < num_probes; p++) {
        const uint32_t* id = probes_to_update + 3 * p;
        for (int ty = 0; ty < 10; ty++)
            for (int tx = 0; tx < 10; tx++) {
                H depth = H(0.f), n = H(0.f);
                for (uint32_t i = 0; i < 4; i++) {
                    const int rx = tx * 2 + (int)(i % 2), ry = ty * 2 + (int)(i / 2);
                    const H ray_depth = H(load_or_zero(tr, rx, ry, (int)p).c[3]);
                    if (ray_depth.v > 0.f) {
                        depth = depth + ray_depth;  // "* weight" is commented out in the shader
                        n = n + H(1.f);
                    }
                }
                depth = n.v > 0.f ? depth / n : H(0.f);
                const H d2 = depth * depth;
                write_probe_texel_with_border(10, 10, id, tx, ty, [&](int x, int y, int z) { store_half2(a.depth, x, y, z, depth, d2); });
            }
    }
    // probe_light_cache_update.comp.slang:13-53, numthreads (11,11,1)
    for (uint32_t p = 0; p < num_probes; p++) {
        const uint32_t* id = probes_to_update + 3 * p;
        for (int ty = 0; ty < 11; ty++)
            for (int tx = 0; tx < 11; tx++) {
                const H3 direction = to_h(octahedral_direction(normalized_octahedral_coordinates((uint32_t)tx, (uint32_t)ty, 11, 11)));
                const uint32_t filter = (uint32_t)std::ceil(20.0f / 11.0f);
                const uint32_t bx = (uint32_t)std::floor((float)tx * (float)filter), by = (uint32_t)std::floor((float)ty * (float)filter);
                H3 light = {H(0.f), H(0.f), H(0.f)};
                H n = H(0.f);
                for (uint32_t i = 0; i < filter * filter; i++) {
                    const uint32_t rx = bx + i % filter, ry = by + i / filter;
                    const Texel t = load_or_zero(tr, (int)rx, (int)ry, (int)p);
                    if (H(t.c[3]).v > 0.f) {
                        const F3 ray_dir = octahedral_direction(normalized_octahedral_coordinates(rx, ry, 20, 20));
                        const H weight = H(dot(to_f(direction), ray_dir).v);  // dot(half3, float3) evaluates in float
                        light = light + H3{H(t.c[0]), H(t.c[1]), H(t.c[2])} * weight;
                        n = n + H(1.f);
                    }
                }
                if (n.v > 0.f) light = light / n;
                else light = {H(0.f), H(0.f), H(0.f)};
                write_probe_texel_with_border(11, 11, id, tx, ty, [&](int x, int y, int z) { store_half3(a.light_cache, x, y, z, light); });
            }
    }
    // probe_rtgi_update.comp.slang:13-53, numthreads (5,6,1)
    for (uint32_t p = 0; p < num_probes; p++) {
        const uint32_t* id = probes_to_update + 3 * p;
        for (int ty = 0; ty < 6; ty++)
            for (int tx = 0; tx < 5; tx++) {
                const uint32_t filter = 20u / 5u;
                const uint32_t bx = (uint32_t)tx * filter, by = (uint32_t)ty * filter;
                H3 light = {H(0.f), H(0.f), H(0.f)};
                H n = H(0.f);
                for (uint32_t i = 0; i < filter * filter; i++) {
                    const Texel t = load_or_zero(tr, (int)(bx + i % filter), (int)(by + i / filter), (int)p);
                    if (H(t.c[3]).v > 0.f) {
                        light = light + H3{H(t.c[0]), H(t.c[1]), H(t.c[2])};  // the cosine weight is computed but not used
                        n = n + H(1.f);
                    }
                }
                if (n.v > 0.f) light = light / n;
                else light = {H(0.f), H(0.f), H(0.f)};
                write_probe_texel_with_border(5, 6, id, tx, ty, [&](int x, int y, int z) { store_half3(a.rtgi, x, y, z, light); });
            }
    }
    // probe_finalize.comp.slang:13-74, numthreads (64,1,1), one group per probe
    for (uint32_t p = 0; p < num_probes; p++) {
        const uint32_t* id = probes_to_update + 3 * p;
        uint32_t num_valid = 0;
        for (uint32_t idx = 0; idx < 100; idx += 64) {  // idx, not idx + lane: all 64 lanes test the same texel
            const uint32_t x = idx % 10, y = idx / 10;
            if (y >= 10) continue;
            const Texel d = load_or_zero(a.depth, (int)(id[0] * 12 + x + 1), (int)(id[1] * 12 + y + 1), (int)id[2]);
            if (H(d.c[0]).v > 0.f) num_valid += 64;
        }
        store_half1_unorm8(a.validity, (int)id[0], (int)id[1], (int)id[2], H((float)num_valid) / H(100.f));
        H3 sum = {H(0.f), H(0.f), H(0.f)};
        for (uint32_t lane = 0; lane < 30; lane++) {
            const uint32_t x = lane % 5, y = lane / 6;
            const Texel t = load_or_zero(a.rtgi, (int)(id[0] * 7 + x + 1), (int)(id[1] * 8 + y + 1), (int)id[2]);
            const H3 v = {H(t.c[0]), H(t.c[1]), H(t.c[2])};
            sum = lane == 0 ? v : sum + v;  // WaveActiveSum: defined here as the fp16 sum in lane order
        }
        store_half3(a.average, (int)id[0], (int)id[1], (int)id[2], sum / H(30.f));
    }
    return SAH_OK;
}

// exports for the known-answer tests (SURVEY.md §8-c fixture i): texel -> octahedral coordinates -> direction
extern "C" void orc_octahedral_direction_of_texel(uint32_t tx, uint32_t ty, uint32_t nx, uint32_t ny, float* coords2, float* dir3) {
    const F2 c = normalized_octahedral_coordinates(tx, ty, nx, ny);
    const F3 d = octahedral_direction(c);
    coords2[0] = c.x.v;
    coords2[1] = c.y.v;
    dir3[0] = d.x.v;
    dir3[1] = d.y.v;
    dir3[2] = d.z.v;
}
