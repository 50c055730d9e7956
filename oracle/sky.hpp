// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// Restates the sky fill: RenderCore/shaders/sky/sky_unified.slang:54-206 (main_fs and helpers).
// Transcendentals (acos, atan, exp, cos) are the correctly rounded fp32 value (evaluated in double).
#pragma once
#include <cmath>

#include "../include/sah_hip.h"
#include "image.hpp"
#include "math.hpp"

namespace orc {

Image img2d(const sah_plane& p);
M4 mat(const float* m);

static inline F cr_acos(F x) { return F((float)std::acos((double)x.v)); }
static inline F cr_atan(F x) { return F((float)std::atan((double)x.v)); }
static inline F cr_exp(F x) { return F((float)std::exp((double)x.v)); }
static inline F cr_cos(F x) { return F((float)std::cos((double)x.v)); }

struct SkyConsts {
    F sky_pi = F(3.14159265358f);         // sky_unified.slang:18-20 (defined before any include)
    F ground = F(6.360f);                 // :27
    F atmosphere = F(6.460f);             // :28
    F3 view_pos;                          // :31
    SkyConsts() { view_pos = {F(0.0f), F(6.360f) + F(0.0002f), F(0.0f)}; }
};

// sky_unified.slang:54-56
static inline F safeacos(F x) { return cr_acos(nclamp(x, F(-1.0f), F(1.0f))); }

// sky_unified.slang:59-74
static inline F ray_intersect_sphere(F3 ro, F3 rd, F rad) {
    F b = dot(ro, rd);
    F c = dot(ro, ro) - rad * rad;
    if (c.v > 0.0f && b.v > 0.0f) return F(-1.0f);
    F discr = b * b - c;
    if (discr.v < 0.0f) return F(-1.0f);
    if (discr.v > (b * b).v) return (-b + nsqrt(discr));
    return -b - nsqrt(discr);
}

static inline F fsign(F x) { return F(x.v > 0.f ? 1.0f : (x.v < 0.f ? -1.0f : 0.0f)); }

// sky_unified.slang:80-109; sampler = linear, REPEAT (procedural_sky.cpp:62-68)
static inline F3 sky_lut_value(const SkyConsts& k, F3 rayDir, F3 sunDir, const Image& sky_view) {
    F height = length(k.view_pos);
    F3 up = k.view_pos / height;
    F horizonAngle = safeacos(nsqrt(height * height - k.ground * k.ground) / height);
    F altitudeAngle = horizonAngle - cr_acos(dot(rayDir, up));
    F azimuthAngle;
    if (std::fabs(altitudeAngle.v) > (F(0.5f) * k.sky_pi - F(.0001f)).v) {
        azimuthAngle = F(0.0f);
    } else {
        F3 right = cross(sunDir, up);
        F3 forward = cross(up, right);
        F3 projectedDir = normalize(rayDir - up * (dot(rayDir, up)));
        F sinTheta = dot(projectedDir, right);
        F cosTheta = dot(projectedDir, forward);
        azimuthAngle = cr_atan(cosTheta / sinTheta) + k.sky_pi;
    }
    F v = F(0.5f) + F(0.5f) * fsign(altitudeAngle) * nsqrt(nabs(altitudeAngle) * F(2.0f) / k.sky_pi);
    F u = azimuthAngle / (F(2.0f) * k.sky_pi);
    Texel t = sample_bilinear(sky_view, u.v, v.v, 0, ADDR_REPEAT);
    return {F(t.c[0]), F(t.c[1]), F(t.c[2])};
}

// sky_unified.slang:111-118
static inline F3 tlut_value(const SkyConsts& k, const Image& tex, F3 pos, F3 sunDir) {
    F height = length(pos);
    F3 up = pos / height;
    F sunCosZenithAngle = dot(sunDir, up);
    F u = nclamp(F(0.5f) + F(0.5f) * sunCosZenithAngle, F(0.0f), F(1.0f));
    F v = nmax(F(0.0f), nmin(F(1.0f), (height - k.ground) / (k.atmosphere - k.ground)));
    Texel t = sample_bilinear(tex, u.v, v.v, 0, ADDR_REPEAT);
    return {F(t.c[0]), F(t.c[1]), F(t.c[2])};
}

// sky_unified.slang:120-135
static inline F sun_with_bloom(const SkyConsts& k, F3 rayDir, F3 sunDir) {
    const F sunSolidAngle = F(0.53f) * k.sky_pi / F(180.0f);
    const F minSunCosTheta = cr_cos(sunSolidAngle);
    F cosTheta = dot(rayDir, sunDir);
    if (cosTheta.v >= minSunCosTheta.v) return F(1.f);
    F offset = minSunCosTheta - cosTheta;
    F gaussianBloom = cr_exp(-offset * F(50000.0f)) * F(0.5f);
    F invBloom = F(1.0f) / (F(0.02f) + offset * F(300.0f)) * F(0.01f);
    return gaussianBloom + invBloom;
}

static inline F smoothstep(F e0, F e1, F x) {
    F t = nclamp((x - e0) / (e1 - e0), F(0.0f), F(1.0f));
    return t * t * (F(3.0f) - F(2.0f) * t);
}

// sky_unified.slang:137-166
static inline F3 sky_color(const SkyConsts& k, F3 view_vector, F3 sunDir, const Image& sky_view, const Image& transmittance) {
    F3 lum = sky_lut_value(k, view_vector, sunDir, sky_view);
    F3 sunLum = F3(sun_with_bloom(k, view_vector, sunDir));
    F s = smoothstep(F(rh(0.002f)), F(1.0f), sunLum.x);  // 0.002h, 1.0h promoted to float
    sunLum = F3(s);
    if (length(sunLum).v > 0.0f) {
        if (ray_intersect_sphere(k.view_pos, view_vector, k.ground).v >= 0.0f) {
            sunLum = F3(F(0.f));
        } else {
            sunLum = sunLum * tlut_value(k, transmittance, k.view_pos, sunDir);
        }
    }
    lum = lum + sunLum;
    lum = lum * F(20.0f);
    lum = lum * F(1.0f);  // exposure_factor = 1.h
    return lum;
}

// sky_unified.slang:185-206.  SV_Position carries the +0.5 already: location_screen = (x + 1) / W (quirk);
// the clip-space xy is that [0,1] value, not remapped to [-1,1] (quirk).  Output half4(half3(sky), 1).
static inline void sky_frag(const sah_view_data& view, const sah_sun_light_constants& sun, const sah_sky_luts& luts, int x, int y,
                            H out[4]) {
    static const SkyConsts k;
    F sx = (F((float)x + 0.5f) + F(0.5f)) / F(view.render_resolution[0]);
    F sy = (F((float)y + 0.5f) + F(0.5f)) / F(view.render_resolution[1]);
    F4 clip = {sx, sy, F(1.f), F(1.f)};
    F4 vs = mul(mat(view.inverse_projection), clip);
    vs = {vs.x / vs.w, vs.y / vs.w, vs.z / vs.w, vs.w / vs.w};
    F4 wv = mul(mat(view.inverse_view), F4{vs.x, vs.y, vs.z, F(0.f)});
    F3 vv = -normalize(F3{wv.x, wv.y, wv.z});
    vv.y = vv.y * F(-1.0f);
    F3 sunDir = -normalize(F3{F(sun.direction_and_tan_size[0]), F(sun.direction_and_tan_size[1]), F(sun.direction_and_tan_size[2])});
    F3 c = sky_color(k, vv, sunDir, img2d(luts.sky_view), img2d(luts.transmittance));
    out[0] = H(c.x.v);
    out[1] = H(c.y.v);
    out[2] = H(c.z.v);
    out[3] = H(1.0f);
}

}  // namespace orc
