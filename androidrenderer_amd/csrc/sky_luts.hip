// Sky LUT generators for gfx950 (the producer of the sky fill's inputs):
//   RenderCore/shaders/sky/common.glsl:8-110, transmittance_lut.comp:9-65 (256 x 64), multiscattering_lut.comp:9-137 (32 x 32),
//   sky_view_lut.comp:9-106 (200 x 200); host RenderCore/render/procedural_sky.cpp:75-149 (three dispatches per frame).
// GLSL fp32 with every operator individually rounded; exp / sin / cos / acos / pow are the fp64 libm value rounded to fp32
// (DESIGN.md §3).  The work is small (a few hundred thousand ray-march steps) and transcendental bound; the only structure worth
// having is parallelism: the multiple-scattering LUT runs its 64 sphere directions on 64 threads and adds them up in the shader's
// (i, j) order.
#include <hip/hip_runtime.h>

#include "../../include/sah_hip.h"
#include "lighting_common.hpp"

namespace sah {

SAH_DEV Fn cr_sin(Fn x) { return Fn((float)sin((double)x.v)); }
SAH_DEV Fn cr_cos(Fn x) { return Fn((float)cos((double)x.v)); }
SAH_DEV Fn cr_pow(Fn x, Fn y) { return Fn((float)pow((double)x.v, (double)y.v)); }
SAH_DEV Fn safeacos(Fn x) { return cr_acos(nclamp(x, Fn(-1.0f), Fn(1.0f))); }
SAH_DEV F3 exp3(F3 v) { return {cr_exp(v.x), cr_exp(v.y), cr_exp(v.z)}; }
SAH_DEV F3 add_s(F3 a, Fn s) { return {a.x + s, a.y + s, a.z + s}; }
SAH_DEV F3 div3(F3 a, F3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }

constexpr float kSkyPi = 3.14159265358f, kGroundMM = 6.360f, kAtmosphereMM = 6.460f;

// common.glsl:36-44
SAH_DEV Fn mie_phase(Fn cosTheta) {
    const Fn g = Fn(0.8f);
    const Fn scale = Fn(3.0f) / (Fn(8.0f) * Fn(kSkyPi));
    const Fn num = (Fn(1.0f) - g * g) * (Fn(1.0f) + cosTheta * cosTheta);
    const Fn denom = (Fn(2.0f) + g * g) * cr_pow(Fn(1.0f) + g * g - Fn(2.0f) * g * cosTheta, Fn(1.5f));
    return scale * num / denom;
}
// common.glsl:46-49
SAH_DEV Fn rayleigh_phase(Fn cosTheta) {
    const Fn k = Fn(3.0f) / (Fn(16.0f) * Fn(kSkyPi));
    return k * (Fn(1.0f) + cosTheta * cosTheta);
}
// common.glsl:51-70
SAH_DEV void scattering_values(F3 pos, F3& rayleighScattering, Fn& mieScattering, F3& extinction) {
    const Fn altitudeKM = nmax(Fn(0.f), length(pos) - Fn(kGroundMM)) * Fn(1000.0f);
    const Fn rayleighDensity = cr_exp(-altitudeKM / Fn(8.0f));
    const Fn mieDensity = cr_exp(-altitudeKM / Fn(1.2f));
    rayleighScattering = F3{Fn(6.6f), Fn(12.3f), Fn(29.4f)} * rayleighDensity;
    const Fn rayleighAbsorption = Fn(0.0f) * rayleighDensity;
    mieScattering = Fn(3.996f) * mieDensity;
    const Fn mieAbsorption = Fn(4.4f) * mieDensity;
    const F3 ozoneAbsorption = F3{Fn(2.26f), Fn(1.54f), Fn(0.f)} * nmax(Fn(0.0f), Fn(1.0f) - nabs(altitudeKM - Fn(25.0f)) / Fn(15.0f));
    extinction = add_s(add_s(add_s(rayleighScattering, rayleighAbsorption), mieScattering), mieAbsorption) + ozoneAbsorption;
}
struct LutArg {
    PlaneArg p;
    uint32_t w, h;
};
// common.glsl:94-110 (both LUT lookups share the parameterisation); sampler: linear, REPEAT (procedural_sky.cpp:62-68)
SAH_DEV F3 lut_value(const LutArg& lut, F3 pos, F3 sunDir) {
    const Fn height = length(pos);
    const F3 up = pos / height;
    const Fn sunCosZenithAngle = dot(sunDir, up);
    const Fn u = nclamp(Fn(0.5f) + Fn(0.5f) * sunCosZenithAngle, Fn(0.0f), Fn(1.0f));
    const Fn v = nmax(Fn(0.0f), nmin(Fn(1.0f), (height - Fn(kGroundMM)) / (Fn(kAtmosphereMM) - Fn(kGroundMM))));
    float t[4];
    sample_bilinear_repeat_rgba16f(lut.p, lut.w, lut.h, u.v, v.v, t);
    return {Fn(t[0]), Fn(t[1]), Fn(t[2])};
}
SAH_DEV void store_lut(const LutArg& lut, uint32_t x, uint32_t y, F3 rgb) {
    if (x >= lut.w || y >= lut.h) return;  // imageStore outside the image is dropped
    uint2 q;
    q.x = (uint32_t)f2h(rgb.x.v) | ((uint32_t)f2h(rgb.y.v) << 16);
    q.y = (uint32_t)f2h(rgb.z.v) | ((uint32_t)f2h(1.0f) << 16);
    *reinterpret_cast<uint2*>(const_cast<uint8_t*>(lut.p.ptr) + (size_t)y * lut.p.pitch + (size_t)x * 8) = q;
}
// transmittance_lut.comp:53-61: texel -> (pos, sunDir)
SAH_DEV void lut_frame(uint32_t x, uint32_t y, uint32_t W, uint32_t H, F3& pos, F3& sunDir) {
    const Fn u = Fn((float)x) / Fn((float)W), v = Fn((float)y) / Fn((float)H);
    const Fn sunCosTheta = Fn(2.0f) * u - Fn(1.0f);
    const Fn sunTheta = safeacos(sunCosTheta);
    const Fn height = mix(Fn(kGroundMM), Fn(kAtmosphereMM), v);
    pos = {Fn(0.0f), height, Fn(0.0f)};
    sunDir = normalize(F3{Fn(0.0f), sunCosTheta, -cr_sin(sunTheta)});
}

// ---- transmittance_lut.comp:15-65 -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_sky_transmittance(LutArg out) {
    const uint32_t x = blockIdx.x * 8 + (threadIdx.x & 7), y = blockIdx.y * 8 + (threadIdx.x >> 3);
    if (x > out.w || y > out.h) return;  // the shader's guard is `>`
    F3 pos, sunDir;
    lut_frame(x, y, out.w, out.h, pos, sunDir);
    F3 transmittance = F3(Fn(1.0f));
    if (ray_intersect_sphere(pos, sunDir, Fn(kGroundMM)).v > 0.0f) {
        transmittance = F3(Fn(0.0f));
    } else {
        const Fn atmoDist = ray_intersect_sphere(pos, sunDir, Fn(kAtmosphereMM));
        Fn t = Fn(0.0f);
        for (float i = 0.0f; i < 40.0f; i += 1.0f) {
            const Fn newT = ((Fn(i) + Fn(0.3f)) / Fn(40.0f)) * atmoDist;
            const Fn dt = newT - t;
            t = newT;
            const F3 newPos = pos + t * sunDir;
            F3 rayleighScattering, extinction;
            Fn mieScattering;
            scattering_values(newPos, rayleighScattering, mieScattering, extinction);
            transmittance = transmittance * exp3(-dt * extinction);
        }
    }
    store_lut(out, x, y, transmittance);
}

// ---- multiscattering_lut.comp:16-137: one workgroup per texel, one thread per sphere direction (i, j) ----------------------------
__global__ void __launch_bounds__(64) k_sky_multiscattering(LutArg tlut, LutArg out) {
    __shared__ float s_lum[64][3], s_fac[64][3];
    const uint32_t x = blockIdx.x, y = blockIdx.y;
    F3 pos, sunDir;
    lut_frame(x, y, out.w, out.h, pos, sunDir);
    {
        const int i = threadIdx.x >> 3, j = threadIdx.x & 7;  // the shader's loops: i outer, j inner
        const Fn theta = Fn(kSkyPi) * (Fn((float)i) + Fn(0.5f)) / Fn(8.0f);
        const Fn phi = safeacos(Fn(1.0f) - Fn(2.0f) * (Fn((float)j) + Fn(0.5f)) / Fn(8.0f));
        const Fn cosPhi = cr_cos(phi), sinPhi = cr_sin(phi), cosTheta_ = cr_cos(theta), sinTheta = cr_sin(theta);
        const F3 rayDir = {sinPhi * sinTheta, cosPhi, sinPhi * cosTheta_};
        const Fn atmoDist = ray_intersect_sphere(pos, rayDir, Fn(kAtmosphereMM));
        const Fn groundDist = ray_intersect_sphere(pos, rayDir, Fn(kGroundMM));
        const Fn tMax = groundDist.v > 0.0f ? groundDist : atmoDist;
        const Fn cosTheta = dot(rayDir, sunDir);
        const Fn miePhaseValue = mie_phase(cosTheta);
        const Fn rayleighPhaseValue = rayleigh_phase(-cosTheta);
        F3 lum = F3(Fn(0.0f)), lumFactor = F3(Fn(0.0f)), transmittance = F3(Fn(1.0f));
        Fn t = Fn(0.0f);
        for (float stepI = 0.0f; stepI < 20.0f; stepI += 1.0f) {
            const Fn newT = ((Fn(stepI) + Fn(0.3f)) / Fn(20.0f)) * tMax;
            const Fn dt = newT - t;
            t = newT;
            const F3 newPos = pos + t * rayDir;
            F3 rayleighScattering, extinction;
            Fn mieScattering;
            scattering_values(newPos, rayleighScattering, mieScattering, extinction);
            const F3 sampleTransmittance = exp3(-dt * extinction);
            const F3 scatteringNoPhase = add_s(rayleighScattering, mieScattering);
            const F3 scatteringF = div3(scatteringNoPhase - scatteringNoPhase * sampleTransmittance, extinction);
            lumFactor = lumFactor + transmittance * scatteringF;
            const F3 sunTransmittance = lut_value(tlut, newPos, sunDir);
            const F3 rayleighInScattering = rayleighScattering * rayleighPhaseValue;
            const Fn mieInScattering = mieScattering * miePhaseValue;
            const F3 inScattering = add_s(rayleighInScattering, mieInScattering) * sunTransmittance;
            const F3 scatteringIntegral = div3(inScattering - inScattering * sampleTransmittance, extinction);
            lum = lum + scatteringIntegral * transmittance;
            transmittance = transmittance * sampleTransmittance;
        }
        if (groundDist.v > 0.0f) {
            F3 hitPos = pos + groundDist * rayDir;
            if (dot(pos, sunDir).v > 0.0f) {
                hitPos = normalize(hitPos) * Fn(kGroundMM);
                lum = lum + transmittance * F3(Fn(0.3f)) * lut_value(tlut, hitPos, sunDir);
            }
        }
        const Fn invSamples = Fn(1.0f) / Fn(64.0f);
        const F3 f = lumFactor * invSamples, l = lum * invSamples;
        s_fac[threadIdx.x][0] = f.x.v; s_fac[threadIdx.x][1] = f.y.v; s_fac[threadIdx.x][2] = f.z.v;
        s_lum[threadIdx.x][0] = l.x.v; s_lum[threadIdx.x][1] = l.y.v; s_lum[threadIdx.x][2] = l.z.v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // fms += ..., lumTotal += ... in (i, j) order
        F3 fms = F3(Fn(0.0f)), lumTotal = F3(Fn(0.0f));
        for (int k = 0; k < 64; k++) {
            fms = fms + F3{Fn(s_fac[k][0]), Fn(s_fac[k][1]), Fn(s_fac[k][2])};
            lumTotal = lumTotal + F3{Fn(s_lum[k][0]), Fn(s_lum[k][1]), Fn(s_lum[k][2])};
        }
        store_lut(out, x, y, div3(lumTotal, F3(Fn(1.0f)) - fms));
    }
}

// ---- sky_view_lut.comp:21-106 -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_sky_view(LutArg tlut, LutArg mslut, LutArg out, float lx, float ly, float lz) {
    const uint32_t x = blockIdx.x * 8 + (threadIdx.x & 7), y = blockIdx.y * 8 + (threadIdx.x >> 3);
    if (x >= out.w || y >= out.h) return;  // the shader lets x == width compute and drops its store: nothing observable
    const F3 viewPos = {Fn(0.0f), Fn(kGroundMM) + Fn(0.0002f), Fn(0.0f)};
    const Fn u = Fn((float)x) / Fn((float)out.w), v = Fn((float)y) / Fn((float)out.h);
    const Fn azimuthAngle = (u - Fn(0.5f)) * Fn(2.0f) * Fn(kSkyPi);
    Fn adjV;
    if (v.v < 0.5f) {
        const Fn coord = Fn(1.0f) - Fn(2.0f) * v;
        adjV = -coord * coord;
    } else {
        const Fn coord = v * Fn(2.0f) - Fn(1.0f);
        adjV = coord * coord;
    }
    const Fn height = length(viewPos);
    const F3 up = viewPos / height;
    const Fn horizonAngle = safeacos(nsqrt(height * height - Fn(kGroundMM) * Fn(kGroundMM)) / height) - Fn(0.5f) * Fn(kSkyPi);
    const Fn altitudeAngle = adjV * Fn(0.5f) * Fn(kSkyPi) - horizonAngle;
    const Fn cosAltitude = cr_cos(altitudeAngle);
    const F3 rayDir = {cosAltitude * cr_sin(azimuthAngle), cr_sin(altitudeAngle), -cosAltitude * cr_cos(azimuthAngle)};
    const F3 lightDir = {Fn(lx), Fn(ly), Fn(lz)};
    const Fn sunAltitude = (Fn(0.5f) * Fn(kSkyPi)) - cr_acos(dot(-lightDir, up));
    const F3 sunDir = {Fn(0.0f), cr_sin(sunAltitude), -cr_cos(sunAltitude)};
    const Fn atmoDist = ray_intersect_sphere(viewPos, rayDir, Fn(kAtmosphereMM));
    const Fn groundDist = ray_intersect_sphere(viewPos, rayDir, Fn(kGroundMM));
    const Fn tMax = groundDist.v < 0.0f ? atmoDist : groundDist;
    // raymarchScattering :21-60
    const Fn cosTheta = dot(rayDir, sunDir);
    const Fn miePhaseValue = mie_phase(cosTheta);
    const Fn rayleighPhaseValue = rayleigh_phase(-cosTheta);
    F3 lum = F3(Fn(0.0f)), transmittance = F3(Fn(1.0f));
    Fn t = Fn(0.0f);
    for (float i = 0.0f; i < 32.0f; i += 1.0f) {
        const Fn newT = ((Fn(i) + Fn(0.3f)) / Fn(32.0f)) * tMax;
        const Fn dt = newT - t;
        t = newT;
        const F3 newPos = viewPos + t * rayDir;
        F3 rayleighScattering, extinction;
        Fn mieScattering;
        scattering_values(newPos, rayleighScattering, mieScattering, extinction);
        const F3 sampleTransmittance = exp3(-dt * extinction);
        const F3 sunTransmittance = lut_value(tlut, newPos, sunDir);
        const F3 psiMS = lut_value(mslut, newPos, sunDir);
        const F3 rayleighInScattering = rayleighScattering * (rayleighPhaseValue * sunTransmittance + psiMS);
        const F3 mieInScattering = mieScattering * (miePhaseValue * sunTransmittance + psiMS);
        const F3 inScattering = rayleighInScattering + mieInScattering;
        const F3 scatteringIntegral = div3(inScattering - inScattering * sampleTransmittance, extinction);
        lum = lum + scatteringIntegral * transmittance;
        transmittance = transmittance * sampleTransmittance;
    }
    store_lut(out, x, y, lum);
}

hipError_t launch_sky_luts(const PlaneArg& transmittance, const PlaneArg& multiscattering, const PlaneArg& sky_view, const float light_vector[3], hipStream_t st) {
    const LutArg t = {transmittance, 256, 64}, m = {multiscattering, 32, 32}, s = {sky_view, 200, 200};
    hipLaunchKernelGGL(k_sky_transmittance, dim3(256 / 8, 64 / 8), dim3(64), 0, st, t);
    hipLaunchKernelGGL(k_sky_multiscattering, dim3(32, 32), dim3(64), 0, st, t, m);
    hipLaunchKernelGGL(k_sky_view, dim3(200 / 8 + 1, 200 / 8 + 1), dim3(64), 0, st, t, m, s, light_vector[0], light_vector[1], light_vector[2]);
    return hipGetLastError();
}

}  // namespace sah
