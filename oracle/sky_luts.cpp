// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// CPU restatement of the sky LUT generators (the producer of a6's inputs):
//   RenderCore/shaders/sky/common.glsl:8-110
//   RenderCore/shaders/sky/transmittance_lut.comp:9-65        256 x 64
//   RenderCore/shaders/sky/multiscattering_lut.comp:9-137     32 x 32, reads the transmittance LUT (linear, REPEAT)
//   RenderCore/shaders/sky/sky_view_lut.comp:9-106            200 x 200, reads both (host: render/procedural_sky.cpp:75-149)
// GLSL fp32: every operator individually rounded, constant expressions included (evaluated here in fp32, operator by operator);
// exp / sin / cos / acos / pow are the fp64 libm value rounded to fp32; stores round to fp16 (rgba16f images).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../include/sah_hip.h"
#include "sky.hpp"

namespace orc {
namespace {

F cr_sin(F x) { return F((float)std::sin((double)x.v)); }
F cr_pow(F x, F y) { return F((float)std::pow((double)x.v, (double)y.v)); }
F3 exp3(F3 v) { return {cr_exp(v.x), cr_exp(v.y), cr_exp(v.z)}; }
F3 operator+(F3 a, F s) { return {a.x + s, a.y + s, a.z + s}; }
F3 operator/(F3 a, F3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }

const F kPi = F(3.14159265358f);
const F kGround = F(6.360f), kAtmosphere = F(6.460f);

// common.glsl:36-44
F mie_phase(F cosTheta) {
    const F g = F(0.8f);
    const F scale = F(3.0f) / (F(8.0f) * kPi);
    const F num = (F(1.0f) - g * g) * (F(1.0f) + cosTheta * cosTheta);
    const F denom = (F(2.0f) + g * g) * cr_pow(F(1.0f) + g * g - F(2.0f) * g * cosTheta, F(1.5f));
    return scale * num / denom;
}
// common.glsl:46-49
F rayleigh_phase(F cosTheta) {
    const F k = F(3.0f) / (F(16.0f) * kPi);
    return k * (F(1.0f) + cosTheta * cosTheta);
}
// common.glsl:51-70
void scattering_values(F3 pos, F3& rayleighScattering, F& mieScattering, F3& extinction) {
    const F altitudeKM = nmax(F(0.f), length(pos) - kGround) * F(1000.0f);
    const F rayleighDensity = cr_exp(-altitudeKM / F(8.0f));
    const F mieDensity = cr_exp(-altitudeKM / F(1.2f));
    rayleighScattering = F3{F(6.6f), F(12.3f), F(29.4f)} * rayleighDensity;
    const F rayleighAbsorption = F(0.0f) * rayleighDensity;
    mieScattering = F(3.996f) * mieDensity;
    const F mieAbsorption = F(4.4f) * mieDensity;
    const F3 ozoneAbsorption = F3{F(2.26f), F(1.54f), F(0.f)} * nmax(F(0.0f), F(1.0f) - nabs(altitudeKM - F(25.0f)) / F(15.0f));
    extinction = rayleighScattering + rayleighAbsorption + mieScattering + mieAbsorption + ozoneAbsorption;
}
// common.glsl:94-110 (both LUT lookups share the parameterisation)
F3 lut_value(const Image& lut, F3 pos, F3 sunDir) {
    const F height = length(pos);
    const F3 up = pos / height;
    const F sunCosZenithAngle = dot(sunDir, up);
    const F u = nclamp(F(0.5f) + F(0.5f) * sunCosZenithAngle, F(0.0f), F(1.0f));
    const F v = nmax(F(0.0f), nmin(F(1.0f), (height - kGround) / (kAtmosphere - kGround)));
    const Texel t = sample_bilinear(lut, u.v, v.v, 0, ADDR_REPEAT);
    return {F(t.c[0]), F(t.c[1]), F(t.c[2])};
}
void store_rgba16f(const sah_plane& p, uint32_t x, uint32_t y, F3 rgb) {
    if (x >= p.width || y >= p.height) return;  // imageStore outside the image is dropped
    uint16_t h[4] = {f32_to_f16(rgb.x.v), f32_to_f16(rgb.y.v), f32_to_f16(rgb.z.v), f32_to_f16(1.0f)};
    std::memcpy((uint8_t*)p.ptr + (size_t)y * p.row_pitch_bytes + (size_t)x * 8, h, 8);
}
// the (u, v) -> (pos, sunDir) parameterisation shared by the first two LUTs: transmittance_lut.comp:53-61
void lut_frame(uint32_t x, uint32_t y, uint32_t W, uint32_t H, F3& pos, F3& sunDir) {
    const F u = F((float)x) / F((float)W), v = F((float)y) / F((float)H);
    const F sunCosTheta = F(2.0f) * u - F(1.0f);
    const F sunTheta = safeacos(sunCosTheta);
    const F height = mix(kGround, kAtmosphere, v);
    pos = {F(0.0f), height, F(0.0f)};
    sunDir = normalize(F3{F(0.0f), sunCosTheta, -cr_sin(sunTheta)});
}

// transmittance_lut.comp:15-41
F3 sun_transmittance(F3 pos, F3 sunDir) {
    if (ray_intersect_sphere(pos, sunDir, kGround).v > 0.0f) return F3(F(0.0f));
    const F atmoDist = ray_intersect_sphere(pos, sunDir, kAtmosphere);
    F t = F(0.0f);
    F3 transmittance = F3(F(1.0f));
    for (float i = 0.0f; i < 40.0f; i += 1.0f) {
        const F newT = ((F(i) + F(0.3f)) / F(40.0f)) * atmoDist;
        const F dt = newT - t;
        t = newT;
        const F3 newPos = pos + t * sunDir;
        F3 rayleighScattering, extinction;
        F mieScattering;
        scattering_values(newPos, rayleighScattering, mieScattering, extinction);
        transmittance = transmittance * exp3(-dt * extinction);
    }
    return transmittance;
}

// multiscattering_lut.comp:16-23
F3 spherical_dir(F theta, F phi) {
    const F cosPhi = cr_cos(phi), sinPhi = cr_sin(phi), cosTheta = cr_cos(theta), sinTheta = cr_sin(theta);
    return {sinPhi * sinTheta, cosPhi, sinPhi * cosTheta};
}
// multiscattering_lut.comp:26-108
void mul_scatt_values(const Image& tlut, F3 pos, F3 sunDir, F3& lumTotal, F3& fms) {
    lumTotal = F3(F(0.0f));
    fms = F3(F(0.0f));
    const int sqrtSamples = 8;
    const F invSamples = F(1.0f) / F((float)(sqrtSamples * sqrtSamples));
    for (int i = 0; i < sqrtSamples; i++) {
        for (int j = 0; j < sqrtSamples; j++) {
            const F theta = kPi * (F((float)i) + F(0.5f)) / F((float)sqrtSamples);
            const F phi = safeacos(F(1.0f) - F(2.0f) * (F((float)j) + F(0.5f)) / F((float)sqrtSamples));
            const F3 rayDir = spherical_dir(theta, phi);
            const F atmoDist = ray_intersect_sphere(pos, rayDir, kAtmosphere);
            const F groundDist = ray_intersect_sphere(pos, rayDir, kGround);
            F tMax = atmoDist;
            if (groundDist.v > 0.0f) tMax = groundDist;
            const F cosTheta = dot(rayDir, sunDir);
            const F miePhaseValue = mie_phase(cosTheta);
            const F rayleighPhaseValue = rayleigh_phase(-cosTheta);
            F3 lum = F3(F(0.0f)), lumFactor = F3(F(0.0f)), transmittance = F3(F(1.0f));
            F t = F(0.0f);
            for (float stepI = 0.0f; stepI < 20.0f; stepI += 1.0f) {
                const F newT = ((F(stepI) + F(0.3f)) / F(20.0f)) * tMax;
                const F dt = newT - t;
                t = newT;
                const F3 newPos = pos + t * rayDir;
                F3 rayleighScattering, extinction;
                F mieScattering;
                scattering_values(newPos, rayleighScattering, mieScattering, extinction);
                const F3 sampleTransmittance = exp3(-dt * extinction);
                const F3 scatteringNoPhase = rayleighScattering + mieScattering;
                const F3 scatteringF = (scatteringNoPhase - scatteringNoPhase * sampleTransmittance) / extinction;
                lumFactor = lumFactor + transmittance * scatteringF;
                const F3 sunTransmittance = lut_value(tlut, newPos, sunDir);
                const F3 rayleighInScattering = rayleighScattering * rayleighPhaseValue;
                const F mieInScattering = mieScattering * miePhaseValue;
                const F3 inScattering = (rayleighInScattering + mieInScattering) * sunTransmittance;
                const F3 scatteringIntegral = (inScattering - inScattering * sampleTransmittance) / extinction;
                lum = lum + scatteringIntegral * transmittance;
                transmittance = transmittance * sampleTransmittance;
            }
            if (groundDist.v > 0.0f) {
                F3 hitPos = pos + groundDist * rayDir;
                if (dot(pos, sunDir).v > 0.0f) {
                    hitPos = normalize(hitPos) * kGround;
                    lum = lum + transmittance * F3(F(0.3f)) * lut_value(tlut, hitPos, sunDir);
                }
            }
            fms = fms + lumFactor * invSamples;
            lumTotal = lumTotal + lum * invSamples;
        }
    }
}

// sky_view_lut.comp:21-60
F3 raymarch_scattering(const Image& tlut, const Image& mslut, F3 pos, F3 rayDir, F3 sunDir, F tMax, F numSteps) {
    const F cosTheta = dot(rayDir, sunDir);
    const F miePhaseValue = mie_phase(cosTheta);
    const F rayleighPhaseValue = rayleigh_phase(-cosTheta);
    F3 lum = F3(F(0.0f)), transmittance = F3(F(1.0f));
    F t = F(0.0f);
    for (float i = 0.0f; i < numSteps.v; i += 1.0f) {
        const F newT = ((F(i) + F(0.3f)) / numSteps) * tMax;
        const F dt = newT - t;
        t = newT;
        const F3 newPos = pos + t * rayDir;
        F3 rayleighScattering, extinction;
        F mieScattering;
        scattering_values(newPos, rayleighScattering, mieScattering, extinction);
        const F3 sampleTransmittance = exp3(-dt * extinction);
        const F3 sunTransmittance = lut_value(tlut, newPos, sunDir);
        const F3 psiMS = lut_value(mslut, newPos, sunDir);
        const F3 rayleighInScattering = rayleighScattering * (rayleighPhaseValue * sunTransmittance + psiMS);
        const F3 mieInScattering = mieScattering * (miePhaseValue * sunTransmittance + psiMS);
        const F3 inScattering = rayleighInScattering + mieInScattering;
        const F3 scatteringIntegral = (inScattering - inScattering * sampleTransmittance) / extinction;
        lum = lum + scatteringIntegral * transmittance;
        transmittance = transmittance * sampleTransmittance;
    }
    return lum;
}

bool lut_ok(const sah_plane* p, uint32_t w, uint32_t h) {
    return p && p->ptr && p->format == FMT_R16G16B16A16_SFLOAT && p->width == w && p->height == h && p->row_pitch_bytes >= w * 8;
}

}  // namespace
}  // namespace orc

using namespace orc;

// ProceduralSky::update_sky_luts: the three dispatches in order
extern "C" int orc_sky_update_luts(const sah_plane* transmittance, const sah_plane* multiscattering, const sah_plane* sky_view, const float* light_vector) {
    if (!lut_ok(transmittance, 256, 64) || !lut_ok(multiscattering, 32, 32) || !lut_ok(sky_view, 200, 200) || !light_vector) return SAH_ERR_INVALID_ARGUMENT;
    // 1. transmittance
#pragma omp parallel for schedule(dynamic)
    for (int y = 0; y < 64; y++)
        for (uint32_t x = 0; x < 256; x++) {
            F3 pos, sunDir;
            lut_frame(x, (uint32_t)y, 256, 64, pos, sunDir);
            store_rgba16f(*transmittance, x, (uint32_t)y, sun_transmittance(pos, sunDir));
        }
    const Image tlut = img2d(*transmittance);
    // 2. multiple scattering
#pragma omp parallel for schedule(dynamic)
    for (int y = 0; y < 32; y++)
        for (uint32_t x = 0; x < 32; x++) {
            F3 pos, sunDir, lum, fms;
            lut_frame(x, (uint32_t)y, 32, 32, pos, sunDir);
            mul_scatt_values(tlut, pos, sunDir, lum, fms);
            store_rgba16f(*multiscattering, x, (uint32_t)y, lum / (F3(F(1.0f)) - fms));
        }
    const Image mslut = img2d(*multiscattering);
    // 3. sky view: 26 x 26 groups of 8 x 8, guard `id > size` (sky_view_lut.comp:67-70): x == 200 computes and its store is dropped
    const F3 viewPos = {F(0.0f), kGround + F(0.0002f), F(0.0f)};
    const F3 lightDir = {F(light_vector[0]), F(light_vector[1]), F(light_vector[2])};
#pragma omp parallel for schedule(dynamic)
    for (int y = 0; y < 200; y++)
        for (uint32_t x = 0; x < 200; x++) {
            const F u = F((float)x) / F(200.0f), v = F((float)y) / F(200.0f);
            const F azimuthAngle = (u - F(0.5f)) * F(2.0f) * kPi;
            F adjV;
            if (v.v < 0.5f) {
                const F coord = F(1.0f) - F(2.0f) * v;
                adjV = -coord * coord;
            } else {
                const F coord = v * F(2.0f) - F(1.0f);
                adjV = coord * coord;
            }
            const F height = length(viewPos);
            const F3 up = viewPos / height;
            const F horizonAngle = safeacos(nsqrt(height * height - kGround * kGround) / height) - F(0.5f) * kPi;
            const F altitudeAngle = adjV * F(0.5f) * kPi - horizonAngle;
            const F cosAltitude = cr_cos(altitudeAngle);
            const F3 rayDir = {cosAltitude * cr_sin(azimuthAngle), cr_sin(altitudeAngle), -cosAltitude * cr_cos(azimuthAngle)};
            const F sunAltitude = (F(0.5f) * kPi) - cr_acos(dot(-lightDir, up));
            const F3 sunDir = {F(0.0f), cr_sin(sunAltitude), -cr_cos(sunAltitude)};
            const F atmoDist = ray_intersect_sphere(viewPos, rayDir, kAtmosphere);
            const F groundDist = ray_intersect_sphere(viewPos, rayDir, kGround);
            const F tMax = groundDist.v < 0.0f ? atmoDist : groundDist;
            store_rgba16f(*sky_view, x, (uint32_t)y, raymarch_scattering(tlut, mslut, viewPos, rayDir, sunDir, tMax, F(32.0f)));
        }
    return SAH_OK;
}
