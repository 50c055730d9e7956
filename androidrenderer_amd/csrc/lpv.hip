// LPV maintenance kernels for gfx950:
//   clear      RenderCore/shaders/gi/lpv/clear_lpv.comp:22-29        (host light_propagation_volume.cpp:839-926)
//   propagate  RenderCore/shaders/gi/lpv/lpv_propagate.comp.slang:76-156 (host :970-1063), fp16 arithmetic,
//              use_gv hard-wired to false by the host (:975) => geo_volume_factor == 1.
// One thread per cell; the three colour volumes are 1 MiB each and stay in the per-XCD L2 between steps.
#include <hip/hip_runtime.h>

#include "../../include/sah_hip.h"
#include "numerics.hpp"
#include "params.hpp"

namespace sah {

struct H4 {
    Hn x, y, z, w;
};
SAH_DEV H4 operator*(Hn s, H4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
SAH_DEV H4 operator*(H4 a, Hn s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
SAH_DEV H4 operator+(H4 a, H4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
SAH_DEV Hn dot4h(H4 a, H4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// spherical_harmonics.slangi:14-32: float literal * half -> float, rounded by the half4 constructor
SAH_DEV H4 dir_to_sh_h(H3 d) {
    return {Hn(0.282094792f), Hn(-0.488602512f * tof(d.y)), Hn(0.488602512f * tof(d.z)), Hn(-0.488602512f * tof(d.x))};
}
SAH_DEV H4 dir_to_cosine_lobe_h(H3 d) {
    return {Hn(0.886226925f), Hn(-1.02332671f * tof(d.y)), Hn(1.02332671f * tof(d.z)), Hn(-1.02332671f * tof(d.x))};
}

__constant__ int8_t kOrient[6][9] = {
    {1, 0, 0, 0, 1, 0, 0, 0, 1},  {-1, 0, 0, 0, 1, 0, 0, 0, -1}, {0, 0, 1, 0, 1, 0, -1, 0, 0},
    {0, 0, -1, 0, 1, 0, 1, 0, 0}, {1, 0, 0, 0, 0, 1, 0, -1, 0},  {1, 0, 0, 0, 0, -1, 0, 1, 0},
};
__constant__ int8_t kDir[6][3] = {{0, 0, 1}, {0, 0, -1}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}};
__constant__ int8_t kSide[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};

SAH_DEV H3 mul33(const int8_t* M, H3 v) {
    H3 r;
    r.x = Hn((float)M[0]) * v.x + Hn((float)M[1]) * v.y + Hn((float)M[2]) * v.z;
    r.y = Hn((float)M[3]) * v.x + Hn((float)M[4]) * v.y + Hn((float)M[5]) * v.z;
    r.z = Hn((float)M[6]) * v.x + Hn((float)M[7]) * v.y + Hn((float)M[8]) * v.z;
    return r;
}

struct Q4 {  // trivially constructible LDS image of an H4
    _Float16 v[4];
};
SAH_DEV Q4 to_q(H4 a) { return Q4{{a.x.v, a.y.v, a.z.v, a.w.v}}; }
SAH_DEV H4 from_q(const Q4& q) { return {Hn::raw(q.v[0]), Hn::raw(q.v[1]), Hn::raw(q.v[2]), Hn::raw(q.v[3])}; }
struct PropTables {  // 30 direction pairs, built per block into LDS
    Q4 eval_sh[6][4], reproj_lobe[6][4], cur_lobe[6], cur_sh[6];
};

SAH_DEV H4 load_h4(const VolumeArg& v, int x, int y, int z) {
    if ((unsigned)x < v.width && (unsigned)y < v.height && (unsigned)z < v.depth) {
        const uint2 q = *reinterpret_cast<const uint2*>(v.ptr + (size_t)z * v.slice_pitch + (size_t)y * v.row_pitch + (size_t)x * 8);
        H4 r;
        r.x = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.x & 0xffffu)));
        r.y = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.x >> 16)));
        r.z = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.y & 0xffffu)));
        r.w = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.y >> 16)));
        return r;
    }
    return {Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f)};
}
SAH_DEV void store_h4(const VolumeArg& v, int x, int y, int z, H4 c) {
    uint2 q;
    q.x = (uint32_t)__builtin_bit_cast(uint16_t, c.x.v) | ((uint32_t)__builtin_bit_cast(uint16_t, c.y.v) << 16);
    q.y = (uint32_t)__builtin_bit_cast(uint16_t, c.z.v) | ((uint32_t)__builtin_bit_cast(uint16_t, c.w.v) << 16);
    *reinterpret_cast<uint2*>(const_cast<uint8_t*>(v.ptr) + (size_t)z * v.slice_pitch + (size_t)y * v.row_pitch + (size_t)x * 8) = q;
}

struct PropArgs {
    VolumeArg src[3], dst[3];
    uint32_t num_cascades;
    // the emitting step (the last one of a sah_lpv_propagate call, when the context keeps the Lighting pass's gather copy current:
    // SAH_GENERATION_TRACKED): every texel is also stored into the interleaved copy — colour c of texel (x, y, z) at
    // (z + 2) * pk_slice_pitch + (y + 2) * pk_row_pitch + (x + 2) * 24 + 8 c, inside a border of zeros that is already there
    // (params.hpp: FastArgs::lpv_packed) — and an inf / NaN texel raises the copy's flag as k_lpv_pack does
    uint8_t* packed;
    uint32_t pk_row_pitch, pk_slice_pitch;
    FrameState* state;
};

// tables of the 30 direction pairs: built once per context into device memory (k_build_prop_tables) and read by the propagate kernels
// through uniform (scalar) loads — as LDS tables they cost every cell 45 16-byte LDS reads and every workgroup a build + barrier
SAH_DEV void build_prop_tables(PropTables& T) {
    if (threadIdx.x < 24) {
        const int n = threadIdx.x >> 2, s = threadIdx.x & 3;
        const Hn small = Hn::lit(0.4472135f), big = Hn::lit(0.894427f);
        const H3 e = mul33(kOrient[n], H3{Hn((float)kSide[s][0]) * small, Hn((float)kSide[s][1]) * small, big});
        const H3 r = mul33(kOrient[n], H3{Hn((float)kSide[s][0]), Hn((float)kSide[s][1]), Hn(0.f)});
        T.eval_sh[n][s] = to_q(dir_to_sh_h(e));
        T.reproj_lobe[n][s] = to_q(dir_to_cosine_lobe_h(r));
    } else if (threadIdx.x < 30) {
        const int n = threadIdx.x - 24;
        const H3 c = {Hn((float)kDir[n][0]), Hn((float)kDir[n][1]), Hn((float)kDir[n][2])};
        T.cur_lobe[n] = to_q(dir_to_cosine_lobe_h(c));
        T.cur_sh[n] = to_q(dir_to_sh_h(c));
    }
}

__global__ void k_build_prop_tables(PropTables* out) { build_prop_tables(*out); }
// filled once per context by k_build_prop_tables through the symbol's address; constant address space: uniform reads are scalar loads
__constant__ PropTables c_prop_tables;

// the 30 direction pairs of one cell from its six neighbours' coefficients: lpv_propagate.comp.slang:96-152
SAH_DEV H4 propagate_from(const PropTables& T, const H4 (&coef)[6]) {
    const Hn direct_sa = Hn(tof(Hn::lit(0.4006696846f)) / 3.1415927f);
    const Hn side_sa = Hn(tof(Hn::lit(0.4234413544f)) / 3.1415927f);
    // (Hn::lit: compile-time constants.  geo_volume_factor == 1: x * 1.0h is x for every x, the compiler folds it.)
    const Hn zero = Hn::lit(0.f), geo_volume_factor = Hn::lit(1.f);
    H4 acc = {zero, zero, zero, zero};
#pragma unroll
    for (int n = 0; n < 6; n++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const Hn m = nmax(zero, dot4h(coef[n], from_q(T.eval_sh[n][s])));
            acc = acc + (side_sa * m) * from_q(T.reproj_lobe[n][s]) * geo_volume_factor;
        }
        const Hn m = nmax(zero, dot4h(coef[n], from_q(T.cur_sh[n])));
        acc = acc + (direct_sa * m) * from_q(T.cur_lobe[n]) * geo_volume_factor;
    }
    return acc;
}

// one propagation step of one cell: lpv_propagate.comp.slang:76-156
// (one colour volume per call: the three channels are independent and run as separate workgroups, blockIdx.y, which triples the
// number of waves in flight — with one thread per cell doing all three the step was bound by its own dependency chains)
SAH_DEV H4 propagate_cell(const PropTables& T, const VolumeArg& src, const VolumeArg& dst, uint32_t idx) {
    const int cx = idx & 31, cy = (idx >> 5) & 31, cz = (idx >> 10) & 31, cascade = idx >> 15;
    const int xoff = cascade * 32;
    // All 18 neighbour texels are fetched before any arithmetic (one latency phase instead of six: with two waves per SIMD the
    // step was latency bound).  A neighbour the shader skips (`continue`, the asymmetric [-1, 31] test) is given zero coefficients
    // instead: max(0, dot(0, sh)) == +0 and acc + (sa * 0) * lobe == acc + (+-0) == acc for every acc this loop can hold (acc starts
    // at +0 and +0 + -0 == +0, so it is never -0).
    H4 coef[6];
#pragma unroll
    for (int n = 0; n < 6; n++) {
        const int nx = cx - kDir[n][0], ny = cy - kDir[n][1], nz = cz - kDir[n][2];
        const bool skipped = nx < -1 || ny < -1 || nz < -1 || nx > 31 || ny > 31 || nz > 31;
        coef[n] = load_h4(src, skipped ? -1 : nx + xoff, ny, nz);
    }
    const H4 out = propagate_from(T, coef);
    store_h4(dst, cx + xoff, cy, cz, out);
    return out;
}

template <bool EMIT>
__global__ void __launch_bounds__(256) k_lpv_propagate(PropArgs a) {
    const PropTables& T = c_prop_tables;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= a.num_cascades * 32768u) return;
    const uint32_t c = blockIdx.y;  // colour volume
    const H4 out = propagate_cell(T, a.src[c], a.dst[c], idx);
    if constexpr (EMIT) {
        const uint32_t x = (idx & 31u) + (idx >> 15) * 32u, y = (idx >> 5) & 31u, z = (idx >> 10) & 31u;
        uint2 q;
        q.x = (uint32_t)__builtin_bit_cast(uint16_t, out.x.v) | ((uint32_t)__builtin_bit_cast(uint16_t, out.y.v) << 16);
        q.y = (uint32_t)__builtin_bit_cast(uint16_t, out.z.v) | ((uint32_t)__builtin_bit_cast(uint16_t, out.w.v) << 16);
        *reinterpret_cast<uint2*>(a.packed + (size_t)(z + kLpvPackBorder) * a.pk_slice_pitch + (size_t)(y + kLpvPackBorder) * a.pk_row_pitch +
                                  (size_t)(x + kLpvPackBorder) * kLpvPackTexel + 8u * c) = q;
        const bool bad = ((q.x & 0x7c00u) == 0x7c00u) | ((q.x & 0x7c000000u) == 0x7c000000u) | ((q.y & 0x7c00u) == 0x7c00u) | ((q.y & 0x7c000000u) == 0x7c000000u);
        if (wave_any(bad) && (threadIdx.x & 63u) == 0u) atomicMax(&a.state->nonfinite, 1u);
    }
}

struct ClearArgs {
    VolumeArg v[4];
    int n;
    uint32_t num_cascades;
};

__global__ void __launch_bounds__(256) k_lpv_clear(ClearArgs a) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= a.num_cascades * 32768u) return;
    const uint32_t x = idx % (32u * a.num_cascades), y = (idx / (32u * a.num_cascades)) & 31u, z = idx / (32u * a.num_cascades * 32u);
    for (int i = 0; i < a.n; i++) {
        const VolumeArg& v = a.v[i];
        if (x < v.width && y < v.height && z < v.depth)
            *reinterpret_cast<uint2*>(const_cast<uint8_t*>(v.ptr) + (size_t)z * v.slice_pitch + (size_t)y * v.row_pitch + (size_t)x * 8) = make_uint2(0u, 0u);
    }
}

hipError_t launch_lpv_clear(const VolumeArg* vols, int n, uint32_t num_cascades, hipStream_t st) {
    ClearArgs a;
    a.n = n;
    a.num_cascades = num_cascades;
    for (int i = 0; i < n; i++) a.v[i] = vols[i];
    hipLaunchKernelGGL(k_lpv_clear, dim3(num_cascades * 128), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_lpv_build_tables(hipStream_t st) {  // on the current device
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(c_prop_tables));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_build_prop_tables, dim3(1), dim3(64), 0, st, (PropTables*)sym);
    return hipGetLastError();
}

// `emit` (or null): where the step also writes the Lighting pass's gather copy of `dst` (PropArgs)
hipError_t launch_lpv_propagate(const VolumeArg src[3], const VolumeArg dst[3], uint32_t num_cascades, const LpvPackEmit* emit, hipStream_t st) {
    PropArgs a = {};
    for (int i = 0; i < 3; i++) { a.src[i] = src[i]; a.dst[i] = dst[i]; }
    a.num_cascades = num_cascades;
    if (emit) {
        a.packed = emit->packed;
        a.pk_row_pitch = emit->row_pitch;
        a.pk_slice_pitch = emit->slice_pitch;
        a.state = emit->state;
        const hipError_t me = hipMemsetAsync(&emit->state->nonfinite, 0, sizeof(uint32_t), st);  // the copy's verdict starts at "finite"
        if (me != hipSuccess) return me;
        hipLaunchKernelGGL(k_lpv_propagate<true>, dim3(num_cascades * 128, 3), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(k_lpv_propagate<false>, dim3(num_cascades * 128, 3), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace sah
