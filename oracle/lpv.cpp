// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// CPU restatement of LPV propagate / clear:
//   RenderCore/shaders/gi/lpv/lpv_propagate.comp.slang:76-156 (half arithmetic, use_gv hard-wired false by
//   RenderCore/render/gi/light_propagation_volume.cpp:975), RenderCore/shaders/gi/lpv/clear_lpv.comp:22-29.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/sah_hip.h"
#include "image.hpp"
#include "math.hpp"

namespace orc {

Image img3d(const sah_volume& v);

struct H4 {
    H x, y, z, w;
};
static inline H4 operator*(H s, H4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
static inline H4 operator*(H4 a, H s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline H4 operator+(H4 a, H4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline H dot4h(H4 a, H4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// spherical_harmonics.slangi:14-32 — constants are float literals, so each product is float*half -> float,
// rounded to half by the half4 constructor.
static inline H4 dir_to_sh_h(H3 d) {
    return {H(0.282094792f), H(-0.488602512f * d.y.v), H(0.488602512f * d.z.v), H(-0.488602512f * d.x.v)};
}
static inline H4 dir_to_cosine_lobe_h(H3 d) {
    return {H(0.886226925f), H(-1.02332671f * d.y.v), H(1.02332671f * d.z.v), H(-1.02332671f * d.x.v)};
}

// lpv_propagate.comp.slang:35-48 (rows), :50-56, :59
static const int kOrient[6][3][3] = {
    {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}},   {{-1, 0, 0}, {0, 1, 0}, {0, 0, -1}}, {{0, 0, 1}, {0, 1, 0}, {-1, 0, 0}},
    {{0, 0, -1}, {0, 1, 0}, {1, 0, 0}},  {{1, 0, 0}, {0, 0, 1}, {0, -1, 0}},  {{1, 0, 0}, {0, 0, -1}, {0, 1, 0}},
};
static const int kDir[6][3] = {{0, 0, 1}, {0, 0, -1}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}};
static const int kSide[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};

static inline H3 mul33(const int M[3][3], H3 v) {
    H3 r;
    H* o[3] = {&r.x, &r.y, &r.z};
    for (int i = 0; i < 3; i++) *o[i] = H((float)M[i][0]) * v.x + H((float)M[i][1]) * v.y + H((float)M[i][2]) * v.z;
    return r;
}

struct PropTables {
    H4 eval_sh[6][4], reproj_lobe[6][4], cur_lobe[6], cur_sh[6];
    H direct_sa, side_sa;
    PropTables() {
        const H small = H::lit(0.4472135), big = H::lit(0.894427);  // :63-64
        for (int n = 0; n < 6; n++) {
            for (int s = 0; s < 4; s++) {
                H3 e = mul33(kOrient[n], H3{H((float)kSide[s][0]) * small, H((float)kSide[s][1]) * small, big});  // :62-69
                H3 r = mul33(kOrient[n], H3{H((float)kSide[s][0]), H((float)kSide[s][1]), H(0.f)});              // :71-74
                eval_sh[n][s] = dir_to_sh_h(e);
                reproj_lobe[n][s] = dir_to_cosine_lobe_h(r);
            }
            H3 c = {H((float)kDir[n][0]), H((float)kDir[n][1]), H((float)kDir[n][2])};
            cur_lobe[n] = dir_to_cosine_lobe_h(c);
            cur_sh[n] = dir_to_sh_h(c);
        }
        // :125-126 — `0.4006696846h / PI` with PI = 3.1415927 (float, prelude.h): half/float -> float -> half
        direct_sa = H(H::lit(0.4006696846).v / 3.1415927f);
        side_sa = H(H::lit(0.4234413544).v / 3.1415927f);
    }
};

static inline H4 load_h4(const Image& im, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= (int)im.width || y >= (int)im.height || z >= (int)im.depth) return {H(0.f), H(0.f), H(0.f), H(0.f)};
    Texel t = load_texel(im, x, y, z);
    return {H::raw(t.c[0]), H::raw(t.c[1]), H::raw(t.c[2]), H::raw(t.c[3])};
}
static inline void store_h4(const sah_volume& v, int x, int y, int z, H4 c) {
    uint16_t h[4] = {f32_to_f16(c.x.v), f32_to_f16(c.y.v), f32_to_f16(c.z.v), f32_to_f16(c.w.v)};
    std::memcpy((uint8_t*)v.ptr + (size_t)z * v.slice_pitch_bytes + (size_t)y * v.row_pitch_bytes + (size_t)x * 8, h, 8);
}

static void propagate_step(const sah_volume src[3], const sah_volume dst[3], uint32_t num_cascades) {
    static const PropTables T;
    const Image im[3] = {img3d(src[0]), img3d(src[1]), img3d(src[2])};
    const int total = (int)num_cascades * 32 * 32 * 32;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < total; idx++) {
        const int cx = idx & 31, cy = (idx >> 5) & 31, cz = (idx >> 10) & 31, cascade = idx >> 15;
        const int xoff = cascade * 32;
        H4 acc[3];
        for (int c = 0; c < 3; c++) acc[c] = {H(0.f), H(0.f), H(0.f), H(0.f)};  // :89-91 — no accumulation of the old value
        for (int n = 0; n < 6; n++) {
            const int nx = cx - kDir[n][0], ny = cy - kDir[n][1], nz = cz - kDir[n][2];
            if (nx < -1 || ny < -1 || nz < -1 || nx > 31 || ny > 31 || nz > 31) continue;  // :99-102 (allows -1, rejects 32)
            H4 coef[3];
            for (int c = 0; c < 3; c++) coef[c] = load_h4(im[c], nx + xoff, ny, nz);  // :121-123, OOB loads return 0
            for (int s = 0; s < 4; s++) {  // :128-140 (geo_volume_factor == 1 exactly when use_gv == 0)
                for (int c = 0; c < 3; c++) {
                    H m = nmax(H(0.f), dot4h(coef[c], T.eval_sh[n][s]));
                    acc[c] = acc[c] + (T.side_sa * m) * T.reproj_lobe[n][s] * H(1.f);
                }
            }
            for (int c = 0; c < 3; c++) {  // :142-150
                H m = nmax(H(0.f), dot4h(coef[c], T.cur_sh[n]));
                acc[c] = acc[c] + (T.direct_sa * m) * T.cur_lobe[n] * H(1.f);
            }
        }
        for (int c = 0; c < 3; c++) store_h4(dst[c], cx + xoff, cy, cz, acc[c]);
    }
}

}  // namespace orc

using namespace orc;

extern "C" int orc_lpv_clear(const sah_volume* red, const sah_volume* green, const sah_volume* blue, const sah_volume* geometry,
                             uint32_t num_cascades) {
    const sah_volume* vols[4] = {red, green, blue, geometry};
    for (const sah_volume* v : vols) {
        if (!v || !v->ptr) continue;
        // dispatch(num_cascades, 32, 32) x local_size 32: texels [0, 32*num_cascades) x 32 x 32
        for (uint32_t z = 0; z < 32 && z < v->depth; z++)
            for (uint32_t y = 0; y < 32 && y < v->height; y++)
                std::memset((uint8_t*)v->ptr + (size_t)z * v->slice_pitch_bytes + (size_t)y * v->row_pitch_bytes, 0,
                            (size_t)std::min(32u * num_cascades, v->width) * 8);
    }
    return SAH_OK;
}

extern "C" int orc_lpv_propagate(const sah_volume a[3], const sah_volume b[3], uint32_t num_cascades, uint32_t steps) {
    if (!a || !b || num_cascades == 0 || num_cascades > 4) return SAH_ERR_INVALID_ARGUMENT;
    // light_propagation_volume.cpp:1016-1034: A->B then B->A per pair of steps
    for (uint32_t s = 0; s < steps; s++) {
        if ((s & 1) == 0) propagate_step(a, b, num_cascades);
        else propagate_step(b, a, num_cascades);
    }
    return SAH_OK;
}
