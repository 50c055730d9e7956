// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// CPU restatement of the two passes between the RSM and the LPV:
//   "Extract VPLs"   RenderCore/shaders/gi/lpv/rsm_generate_vpls.comp:44-139, dispatch RenderCore/render/gi/light_propagation_volume.cpp:636-686
//   "VPL Injection"  RenderCore/shaders/gi/lpv/vpl_injection.vert:27-66, vpl_injection.frag:13-52, render pass light_propagation_volume.cpp:699-760
// GLSL fp32 (mediump = RelaxedPrecision evaluates in fp32), every operator rounded.  The order of the VPL list and the
// roundings the API leaves open are the ones include/sah_hip.h fixes: ascending invocation index; round() and packSnorm4x8 to even.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../include/sah_hip.h"
#include "codec.hpp"
#include "math.hpp"

namespace orc {
namespace {

struct Vpl {
    F3 position, color, normal;
};

const uint8_t* texel(const sah_volume& v, uint32_t layer, int x, int y, int bpp) {
    return (const uint8_t*)v.ptr + (size_t)layer * v.slice_pitch_bytes + (size_t)y * v.row_pitch_bytes + (size_t)x * bpp;
}
M4 load_m4(const float* p) {
    M4 m;
    std::memcpy(m.m, p, 64);
    return m;
}
F rint_f(F x) { return F(std::nearbyint(x.v)); }

// rsm_generate_vpls.comp:44-53, 71-78
Vpl load_rsm_vpl(const sah_rsm_targets& rsm, const sah_lpv_cascade_matrices& c, uint32_t cascade, int x, int y) {
    uint16_t d16;
    std::memcpy(&d16, texel(rsm.depth, cascade, x, y, 2), 2);
    const F depth = F(unorm16_to_float(d16));
    const F res = F((float)rsm.depth.width);
    const F tx = (F((float)x) + F(0.5f)) / res, ty = (F((float)y) + F(0.5f)) / res;
    F4 ws = mul(load_m4(c.inverse_rsm_vp), F4{tx * F(2.0f) - F(1.0f), ty * F(2.0f) - F(1.0f), depth, F(1.0f)});
    Vpl l;
    l.position = {ws.x / ws.w, ws.y / ws.w, ws.z / ws.w};
    const uint8_t* f = texel(rsm.flux, cascade, x, y, 4);
    l.color = {F(srgb8_to_linear(f[0])), F(srgb8_to_linear(f[1])), F(srgb8_to_linear(f[2]))};
    const uint8_t* n = texel(rsm.normals, cascade, x, y, 4);
    l.normal = {F(unorm8_to_float(n[0])) * F(2.0f) - F(1.0f), F(unorm8_to_float(n[1])) * F(2.0f) - F(1.0f), F(unorm8_to_float(n[2])) * F(2.0f) - F(1.0f)};
    return l;
}
// :80-87
F3 position_to_grid_cell(const sah_lpv_cascade_matrices& c, uint32_t cascade, float grid_cell_size, F3 p) {
    const F4 cp = mul(load_m4(c.world_to_cascade), F4{p.x, p.y, p.z, F(1.0f)});
    const F side = F(grid_cell_size) * F(32.0f);
    return {rint_f((cp.x + F((float)cascade)) * side), rint_f(cp.y * side), rint_f(cp.z * side)};
}
uint32_t pack_half2(F a, F b) { return (uint32_t)f32_to_f16(a.v) | ((uint32_t)f32_to_f16(b.v) << 16); }
uint32_t pack_snorm4(F x, F y, F z, F w) {
    const F v[4] = {x, y, z, w};
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const int q = (int)std::nearbyint((nclamp(v[i], F(-1.0f), F(1.0f)) * F(127.0f)).v);  // NaN clamps to -1 by nmax/nmin order
        r |= ((uint32_t)(q & 0xff)) << (8 * i);
    }
    return r;
}

// GLSL built-ins of vpl_injection.frag
F fract_f(F x) { return x - F(std::floor(x.v)); }
F step_f(F edge, F x) { return F(x.v < edge.v ? 0.0f : 1.0f); }
F mixf(F x, F y, F a) { return x * (F(1.0f) - a) + y * a; }

}  // namespace
}  // namespace orc

extern "C" {

int orc_lpv_extract_vpls(const sah_rsm_targets* rsm, const sah_lpv_cascade_matrices* cascades, uint32_t cascade_index, float grid_cell_size,
                         sah_packed_vpl* vpl_list, uint32_t* vpl_count) {
    using namespace orc;
    if (!rsm || !cascades || !vpl_list || !vpl_count || cascade_index >= 4 || cascade_index >= rsm->depth.depth) return SAH_ERR_INVALID_ARGUMENT;
    const uint32_t res = rsm->depth.width;
    if (res == 0 || res % 2 || rsm->depth.height != res) return SAH_ERR_INVALID_ARGUMENT;
    const sah_lpv_cascade_matrices& c = cascades[cascade_index];
    uint32_t count = 0;
    for (uint32_t gy = 0; gy < res / 2; gy++)
        for (uint32_t gx = 0; gx < res / 2; gx++) {
            const int x0 = (int)gx * 2, y0 = (int)gy * 2;
            // brightest of the 2x2 texels (:97-110); chosen_cell is uninitialised in the shader when no texel is brighter than 0 —
            // then every colour is 0 and nothing is stored whatever it holds
            F brightest = F(0.0f);
            F3 chosen{F(0.0f), F(0.0f), F(0.0f)};
            for (int y = 0; y < 2; y++)
                for (int x = 0; x < 2; x++) {
                    const Vpl v = load_rsm_vpl(*rsm, c, cascade_index, x0 + x, y0 + y);
                    const F luma = v.color.x * F(0.2126f) + v.color.y * F(0.7152f) + v.color.z * F(0.0722f);
                    if (luma.v > brightest.v) {
                        brightest = luma;
                        chosen = position_to_grid_cell(c, cascade_index, grid_cell_size, v.position);
                    }
                }
            // texels within sqrt(3) cells of it (:112-133)
            Vpl r{{F(0.f), F(0.f), F(0.f)}, {F(0.f), F(0.f), F(0.f)}, {F(0.f), F(0.f), F(0.f)}};
            F n = F(0.0f);
            for (int y = 0; y < 2; y++)
                for (int x = 0; x < 2; x++) {
                    const Vpl v = load_rsm_vpl(*rsm, c, cascade_index, x0 + x, y0 + y);
                    const F3 d = position_to_grid_cell(c, cascade_index, grid_cell_size, v.position) - chosen;
                    if (dot(d, d).v < 3.0f) {
                        r.position = r.position + v.position;
                        r.color = r.color + v.color;
                        r.normal = r.normal + v.normal;
                        n = n + F(1.0f);
                    }
                }
            if (n.v > 0.0f) {
                r.position = r.position / n;
                r.color = r.color / n;
                r.normal = normalize(r.normal / n);
            }
            if (length(r.color).v > 0.0f && length(r.normal).v > 0.0f) {  // NaN lengths compare false
                sah_packed_vpl p;
                p.data[0] = pack_half2(r.position.x, r.position.y);
                p.data[1] = pack_half2(r.position.z, r.color.x);
                p.data[2] = pack_half2(r.color.y, r.color.z);
                p.data[3] = pack_snorm4(r.normal.x, r.normal.y, r.normal.z, F(0.0f));
                vpl_list[count++] = p;
            }
        }
    *vpl_count = count;
    return SAH_OK;
}

int orc_lpv_inject_vpls(const sah_packed_vpl* vpl_list, const uint32_t* vpl_count, uint32_t capacity, const sah_lpv_cascade_matrices* cascades,
                        uint32_t cascade_index, uint32_t num_cascades, const sah_volume rgb[3]) {
    using namespace orc;
    if (!vpl_list || !vpl_count || !cascades || !rgb || cascade_index >= num_cascades || num_cascades > 4) return SAH_ERR_INVALID_ARGUMENT;
    for (int c = 0; c < 3; c++)
        if (!rgb[c].ptr || rgb[c].format != SAH_FORMAT_R16G16B16A16_SFLOAT || rgb[c].width != rgb[0].width || rgb[c].height != rgb[0].height || rgb[c].depth != rgb[0].depth)
            return SAH_ERR_INVALID_ARGUMENT;
    const uint32_t W = rgb[0].width, H = rgb[0].height, D = rgb[0].depth;
    const uint32_t count = *vpl_count < capacity ? *vpl_count : capacity;
    const M4 world_to_cascade = load_m4(cascades[cascade_index].world_to_cascade);
    for (uint32_t i = 0; i < count; i++) {
        const sah_packed_vpl& p = vpl_list[i];
        // vpl_injection.vert:27-43
        const F3 position{F(f16_to_f32((uint16_t)p.data[0])), F(f16_to_f32((uint16_t)(p.data[0] >> 16))), F(f16_to_f32((uint16_t)p.data[1]))};
        const F3 color{F(f16_to_f32((uint16_t)(p.data[1] >> 16))), F(f16_to_f32((uint16_t)p.data[2])), F(f16_to_f32((uint16_t)(p.data[2] >> 16)))};
        auto snorm = [](uint32_t b) { const float v = (float)(int8_t)(uint8_t)b / 127.0f; return F(v < -1.0f ? -1.0f : v); };
        const F3 normal = normalize(F3{snorm(p.data[3]), snorm(p.data[3] >> 8), snorm(p.data[3] >> 16)});
        // :45-66
        const F4 cp = mul(world_to_cascade, F4{position.x, position.y, position.z, F(1.0f)});
        const F px = (cp.x + F((float)cascade_index)) / F((float)num_cascades);
        const F ndc_x = px * F(2.0f) - F(1.0f), ndc_y = cp.y * F(2.0f) - F(1.0f);
        const F layer_f = cp.z * F(32.0f);
        if (length(normal).v < 1.0f || length(color).v == 0.0f) continue;  // gl_Position = NaN: culled.  NaN lengths fall through, as in GLSL
        // point of size 1 at window position (x_f, y_f): the pixel that contains it; outside the viewport or the layers: dropped
        const F xf = ndc_x * F((float)W * 0.5f) + F((float)W * 0.5f), yf = ndc_y * F((float)H * 0.5f) + F((float)H * 0.5f);
        if (!(xf.v >= 0.0f && xf.v < (float)W && yf.v >= 0.0f && yf.v < (float)H)) continue;
        if (!(layer_f.v > -1.0f && layer_f.v < (float)D)) continue;  // int() truncates toward zero: (-1, 0) is layer 0
        const int cx = (int)std::floor(xf.v), cy = (int)std::floor(yf.v), cz = (int)layer_f.v;
        // vpl_injection.frag:36-51
        const F3 scaled{color.x * F(1024.0f) / F(16384.0f), color.y * F(1024.0f) / F(16384.0f), color.z * F(1024.0f) / F(16384.0f)};
        // rgb2hsv (:13-22)
        const F Kx = F(0.0f), Ky = F(-1.0f) / F(3.0f), Kz = F(2.0f) / F(3.0f), Kw = F(-1.0f);
        const F s1 = step_f(scaled.z, scaled.y);
        const F p4[4] = {mixf(scaled.z, scaled.y, s1), mixf(scaled.y, scaled.z, s1), mixf(Kw, Kx, s1), mixf(Kz, Ky, s1)};
        const F s2 = step_f(p4[0], scaled.x);
        const F q4[4] = {mixf(p4[0], scaled.x, s2), mixf(p4[1], p4[1], s2), mixf(p4[3], p4[2], s2), mixf(scaled.x, p4[0], s2)};
        const F d = q4[0] - nmin(q4[3], q4[1]);
        const F e = F(1.0e-10f);
        F hsv[3] = {nabs(q4[2] + (q4[3] - q4[1]) / (F(6.0f) * d + e)), d / (q4[0] + e), q4[0]};
        hsv[1] = hsv[1] * F(2.0f);  // "Boost saturation because yolo"
        // hsv2rgb (:24-29)
        const F k4[4] = {F(1.0f), F(2.0f) / F(3.0f), F(1.0f) / F(3.0f), F(3.0f)};
        F corrected[3];
        for (int k = 0; k < 3; k++) {
            const F pk = nabs(fract_f(hsv[0] + k4[k]) * F(6.0f) - k4[3]);
            corrected[k] = hsv[2] * mixf(k4[0], nclamp(pk - k4[0], F(0.0f), F(1.0f)), hsv[1]);
        }
        // dir_to_cosine_lobe (spherical_harmonics.glsl:28-30,73-76)
        const F c0 = F(0.886226925f), c1 = F(1.02332671f);
        const F sh[4] = {c0, -c1 * normal.y, c1 * normal.z, -c1 * normal.x};
        const F pi = F(3.1415927f);
        for (int ch = 0; ch < 3; ch++) {
            uint16_t* dst = (uint16_t*)((uint8_t*)rgb[ch].ptr + (size_t)cz * rgb[ch].slice_pitch_bytes + (size_t)cy * rgb[ch].row_pitch_bytes + (size_t)cx * 8);
            for (int k = 0; k < 4; k++) {
                const F src = sh[k] * corrected[ch] / pi;
                dst[k] = f32_to_f16((F(f16_to_f32(dst[k])) + src).v);  // additive blend ONE / ONE, one rounding to half
            }
        }
    }
    return SAH_OK;
}

}  // extern "C"
