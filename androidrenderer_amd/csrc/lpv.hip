// LPV maintenance kernels for gfx950:
//   clear      RenderCore/shaders/gi/lpv/clear_lpv.comp:22-29        (host light_propagation_volume.cpp:839-926)
//   propagate  RenderCore/shaders/gi/lpv/lpv_propagate.comp.slang:76-156 (host :970-1063), fp16 arithmetic,
//              use_gv hard-wired to false by the host (:975) => geo_volume_factor == 1.
// One thread per cell; the three colour volumes are 1 MiB each and stay in the per-XCD L2 between steps.
#include <hip/hip_runtime.h>

#include <type_traits>

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
    uint32_t hot;  // the tables have the structure propagate_from_hot relies on: waves with finite coefficients take it
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

// ---- the same 30 direction pairs for FINITE coefficients, at half the arithmetic (round 6) ------------------------------------------------
// Of the 4 + 4 table entries of a direction pair most are zeros BY CONSTRUCTION: kOrient[n] is a signed permutation, so the evaluation
// direction of side s of neighbour n is +-0.894 along one world axis (w: the same for all four sides), +-0.447 along a second (u for s = 0, 2;
// v for s = 1, 3) and 0 along the third, its re-projection direction is +-1 along u or v alone, and the neighbour's own direction +-1 along w
// alone.  With the SH / lobe layouts (c0, -k y, k z, -k x) a pair therefore has three non-zero SH entries and two non-zero lobe entries; the
// first SH entry is one constant for all 30 pairs, the w entry one value for a neighbour's four sides, and sides s and s + 2 hold negated
// u / v entries.  For finite coefficients
//   * a product with a zero entry is +-0, and  x + (+-0) == x  for every x but a zero, whose SIGN may change: a dot product evaluated
//     without those terms differs from the shader's at most in the sign of a zero result;
//   * max(0, +-0) is a zero either way, (sa * +-0) * lobe a vector of zeros, and  acc + (+-0) == acc  because acc is never -0 (it starts at +0,
//     and a sum is -0 only if both operands are): the sign of that zero never reaches acc;
//   * no dot product overflows (|dot| <= 0.94 max|c| < 65504), so sa * m is finite and its products with zero lobe entries are zeros, not NaN;
//   * c * (-S) == -(c * S) bit for bit.
// So the accumulated vector is the shader's bit for bit from 44 instead of 85 fp16 operations per neighbour.  A wave with an inf / NaN among its
// coefficients takes propagate_from above.  The structure is not taken on faith: launch_lpv_build_tables reads the tables back and checks every
// identity used here (prop_tables_have_hot_structure); a context whose tables fail it propagates with the general form only.
constexpr int8_t kOrientC[6][9] = {
    {1, 0, 0, 0, 1, 0, 0, 0, 1},  {-1, 0, 0, 0, 1, 0, 0, 0, -1}, {0, 0, 1, 0, 1, 0, -1, 0, 0},
    {0, 0, -1, 0, 1, 0, 1, 0, 0}, {1, 0, 0, 0, 0, 1, 0, -1, 0},  {1, 0, 0, 0, 0, -1, 0, 1, 0},
};
// SH / lobe component that carries world axis i (x -> 3, y -> 1, z -> 2), and the component of column j (0: u, 1: v, 2: w) of kOrient[n]
constexpr int sh_comp_of_axis(int i) { return i == 0 ? 3 : (i == 1 ? 1 : 2); }
constexpr int sh_comp_of_col(int n, int j) {
    for (int i = 0; i < 3; i++)
        if (kOrientC[n][i * 3 + j] != 0) return sh_comp_of_axis(i);
    return 0;
}
template <int K> SAH_DEV Hn h4_get(const H4& a) {
    if constexpr (K == 0) return a.x;
    else if constexpr (K == 1) return a.y;
    else if constexpr (K == 2) return a.z;
    else return a.w;
}
template <int K> SAH_DEV void h4_add(H4& a, Hn t) {
    if constexpr (K == 0) a.x = a.x + t;
    else if constexpr (K == 1) a.y = a.y + t;
    else if constexpr (K == 2) a.z = a.z + t;
    else a.w = a.w + t;
}
SAH_DEV Hn hneg(Hn a) { return Hn::raw(-a.v); }

template <int N> SAH_DEV void propagate_from_neighbour_hot(const PropTables& T, const H4& c, H4& acc) {
    constexpr int CU = sh_comp_of_col(N, 0), CV = sh_comp_of_col(N, 1), CW = sh_comp_of_col(N, 2);
    const Hn direct_sa = Hn(tof(Hn::lit(0.4006696846f)) / 3.1415927f);
    const Hn side_sa = Hn(tof(Hn::lit(0.4234413544f)) / 3.1415927f);
    const Hn zero = Hn::lit(0.f);
    const H4 sh0 = from_q(T.eval_sh[N][0]), sh1 = from_q(T.eval_sh[N][1]), shd = from_q(T.cur_sh[N]);
    const Hn p0 = c.x * shd.x;
    const Hn pw = h4_get<CW>(c) * h4_get<CW>(sh0);
    const Hn pu = h4_get<CU>(c) * h4_get<CU>(sh0);
    const Hn pv = h4_get<CV>(c) * h4_get<CV>(sh1);
    const Hn pd = h4_get<CW>(c) * h4_get<CW>(shd);
    auto side = [&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        constexpr int CS = (S & 1) ? CV : CU;
        const Hn ps = S == 0 ? pu : (S == 2 ? hneg(pu) : (S == 1 ? pv : hneg(pv)));
        Hn dot = p0;  // components in the order of dot4h: 1, 2, 3 (the zero one dropped)
#pragma unroll
        for (int k = 1; k <= 3; k++) {
            if (k == CW) dot = dot + pw;
            else if (k == CS) dot = dot + ps;
        }
        const Hn t = side_sa * nmax(zero, dot);
        const H4 lobe = from_q(T.reproj_lobe[N][S]);
        acc.x = acc.x + t * lobe.x;
        h4_add<CS>(acc, t * h4_get<CS>(lobe));
    };
    side(std::integral_constant<int, 0>{});
    side(std::integral_constant<int, 1>{});
    side(std::integral_constant<int, 2>{});
    side(std::integral_constant<int, 3>{});
    const Hn t = direct_sa * nmax(zero, p0 + pd);
    const H4 lobe = from_q(T.cur_lobe[N]);
    acc.x = acc.x + t * lobe.x;
    h4_add<CW>(acc, t * h4_get<CW>(lobe));
}
SAH_DEV H4 propagate_from_hot(const PropTables& T, const H4 (&coef)[6]) {
    const Hn zero = Hn::lit(0.f);
    H4 acc = {zero, zero, zero, zero};
    propagate_from_neighbour_hot<0>(T, coef[0], acc);
    propagate_from_neighbour_hot<1>(T, coef[1], acc);
    propagate_from_neighbour_hot<2>(T, coef[2], acc);
    propagate_from_neighbour_hot<3>(T, coef[3], acc);
    propagate_from_neighbour_hot<4>(T, coef[4], acc);
    propagate_from_neighbour_hot<5>(T, coef[5], acc);
    return acc;
}

// the identities propagate_from_hot relies on, checked on the tables the device built (bit patterns; host side)
static bool prop_tables_have_hot_structure(const PropTables& T) {
    auto bits = [](const Q4& q, int k) { return __builtin_bit_cast(uint16_t, q.v[k]); };
    auto is_zero = [&](const Q4& q, int k) { return (bits(q, k) & 0x7fffu) == 0u; };
    auto finite_nonzero = [&](const Q4& q, int k) { return !is_zero(q, k) && (bits(q, k) & 0x7c00u) != 0x7c00u; };
    bool ok = true;
    for (int n = 0; n < 6; n++) {
        const int cu = sh_comp_of_col(n, 0), cv = sh_comp_of_col(n, 1), cw = sh_comp_of_col(n, 2);
        ok = ok && cu != cv && cv != cw && cu != cw && cu >= 1 && cv >= 1 && cw >= 1;
        for (int s = 0; s < 4; s++) {
            const int cs = (s & 1) ? cv : cu;
            for (int k = 0; k < 4; k++) {
                const bool sh_nz = k == 0 || k == cw || k == cs, lobe_nz = k == 0 || k == cs;
                ok = ok && (sh_nz ? finite_nonzero(T.eval_sh[n][s], k) : is_zero(T.eval_sh[n][s], k));
                ok = ok && (lobe_nz ? finite_nonzero(T.reproj_lobe[n][s], k) : is_zero(T.reproj_lobe[n][s], k));
            }
            ok = ok && bits(T.eval_sh[n][s], 0) == bits(T.cur_sh[n], 0) && bits(T.eval_sh[n][s], cw) == bits(T.eval_sh[n][0], cw);
        }
        ok = ok && bits(T.eval_sh[n][2], cu) == (bits(T.eval_sh[n][0], cu) ^ 0x8000u) && bits(T.eval_sh[n][3], cv) == (bits(T.eval_sh[n][1], cv) ^ 0x8000u);
        for (int k = 0; k < 4; k++) {
            const bool nz = k == 0 || k == cw;
            ok = ok && (nz ? finite_nonzero(T.cur_sh[n], k) : is_zero(T.cur_sh[n], k)) && (nz ? finite_nonzero(T.cur_lobe[n], k) : is_zero(T.cur_lobe[n], k));
        }
    }
    return ok;
}

// one propagation step of one cell: lpv_propagate.comp.slang:76-156
// (one colour volume per call: the three channels are independent and run as separate workgroups, blockIdx.y, which triples the
// number of waves in flight — with one thread per cell doing all three the step was bound by its own dependency chains)
SAH_DEV uint2 load_q(const VolumeArg& v, int x, int y, int z) {
    if ((unsigned)x < v.width && (unsigned)y < v.height && (unsigned)z < v.depth)
        return *reinterpret_cast<const uint2*>(v.ptr + (size_t)z * v.slice_pitch + (size_t)y * v.row_pitch + (size_t)x * 8);
    return make_uint2(0u, 0u);
}
SAH_DEV H4 h4_of(uint2 q) {
    H4 r;
    r.x = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.x & 0xffffu)));
    r.y = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.x >> 16)));
    r.z = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.y & 0xffffu)));
    r.w = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.y >> 16)));
    return r;
}
// bit 15 / bit 31 set <=> the low / high half of w has an all-ones exponent (inf or NaN): 0x7c00 + 0x0400 carries into the sign position
SAH_DEV uint32_t nonfinite_halves(uint32_t w) { return (w & 0x7c007c00u) + 0x04000400u; }

// NC colour volumes per thread (1: blockIdx.y picks the colour; 3: all of them — the neighbour tests, addresses and the scalar table reads are
// shared, a third as many waves each three times as long).
// The neighbour of direction n is the cell at c - kDir[n]; which of them exist is one compare each, because c is in [0, 32)^3 and the host has
// checked the extents ((32 * cascades) x 32 x 32 at least, api_post.cpp):  c - 1 is outside the volume only below 0 — for x that is x == 0 of
// cascade 0: a cell of column 0 of a later cascade reads the last column of the cascade before it, the shader's own quirk —, and c + 1 is the
// shader's skipped neighbour exactly when c == 31 (the asymmetric [-1, 31] test: column 31 never reads the next cascade).  Offsets are 32-bit
// (host-checked) from the centre cell's: one add per neighbour, no 64-bit multiply-adds; directions are compile-time constants — as
// __constant__ data (rounds 1-5) every neighbour's address waited for a load of its direction, and the waits (vmcnt counts in order) also
// waited for the neighbour texels requested before: six round trips in series, most of a step's 9.3 us.
constexpr int8_t kDirC[6][3] = {{0, 0, 1}, {0, 0, -1}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}};
template <int NC, bool EMIT>
SAH_DEV void propagate_cell_hot(const PropTables& T, const PropArgs& a, uint32_t idx, uint32_t c0) {
    const uint32_t cx = idx & 31u, cy = (idx >> 5) & 31u, cz = (idx >> 10) & 31u, x = cx + (idx >> 15) * 32u;
    uint2 q[NC][6];
    uint32_t off_c[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const VolumeArg& v = a.src[c0 + c];
        off_c[c] = cz * v.slice_pitch + cy * v.row_pitch + x * 8u;
    }
#pragma unroll
    for (int n = 0; n < 6; n++) {
        constexpr int8_t zero8 = 0;
        const int axis = kDirC[n][0] != zero8 ? 0 : (kDirC[n][1] != zero8 ? 1 : 2);
        const bool minus = kDirC[n][axis] > 0;  // the neighbour at c - 1 along the axis
        const uint32_t ca = axis == 0 ? cx : (axis == 1 ? cy : cz);
        const bool valid = minus ? (axis == 0 ? x >= 1u : ca >= 1u) : ca <= 30u;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const VolumeArg& v = a.src[c0 + c];
            const uint32_t step = axis == 0 ? 8u : (axis == 1 ? v.row_pitch : v.slice_pitch);
            const uint32_t off = minus ? off_c[c] - step : off_c[c] + step;
            q[c][n] = make_uint2(0u, 0u);
            if (valid) q[c][n] = *reinterpret_cast<const uint2*>(v.ptr + off);
        }
    }
    uint32_t bad = 0;
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int n = 0; n < 6; n++) bad |= nonfinite_halves(q[c][n].x) | nonfinite_halves(q[c][n].y);
    const bool general = !a.hot || wave_any((bad & 0x80008000u) != 0u);
    uint32_t bad_out = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        H4 coef[6];
#pragma unroll
        for (int n = 0; n < 6; n++) coef[n] = h4_of(q[c][n]);
        const H4 out = general ? propagate_from(T, coef) : propagate_from_hot(T, coef);
        uint2 o;
        o.x = (uint32_t)__builtin_bit_cast(uint16_t, out.x.v) | ((uint32_t)__builtin_bit_cast(uint16_t, out.y.v) << 16);
        o.y = (uint32_t)__builtin_bit_cast(uint16_t, out.z.v) | ((uint32_t)__builtin_bit_cast(uint16_t, out.w.v) << 16);
        const VolumeArg& d = a.dst[c0 + c];
        *reinterpret_cast<uint2*>(const_cast<uint8_t*>(d.ptr) + (cz * d.slice_pitch + cy * d.row_pitch + x * 8u)) = o;
        if constexpr (EMIT) {
            *reinterpret_cast<uint2*>(a.packed + ((cz + kLpvPackBorder) * a.pk_slice_pitch + (cy + kLpvPackBorder) * a.pk_row_pitch +
                                                  (x + kLpvPackBorder) * kLpvPackTexel + 8u * (c0 + (uint32_t)c))) = o;
            bad_out |= nonfinite_halves(o.x) | nonfinite_halves(o.y);
        }
    }
    if constexpr (EMIT) {
        if (wave_any((bad_out & 0x80008000u) != 0u) && (threadIdx.x & 63u) == 0u) atomicMax(&a.state->nonfinite, 1u);
    }
}

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

template <int NC, bool EMIT>
__global__ void __launch_bounds__(256) k_lpv_propagate_hot(PropArgs a) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= a.num_cascades * 32768u) return;
    propagate_cell_hot<NC, EMIT>(c_prop_tables, a, idx, NC == 1 ? blockIdx.y : 0u);
}

hipError_t launch_lpv_clear(const VolumeArg* vols, int n, uint32_t num_cascades, hipStream_t st) {
    ClearArgs a;
    a.n = n;
    a.num_cascades = num_cascades;
    for (int i = 0; i < n; i++) a.v[i] = vols[i];
    hipLaunchKernelGGL(k_lpv_clear, dim3(num_cascades * 128), dim3(256), 0, st, a);
    return hipGetLastError();
}

// on the current device.  `hot_structure`: the tables as built have the structure propagate_from_hot relies on (read back once and checked)
hipError_t launch_lpv_build_tables(hipStream_t st, bool* hot_structure) {
    void* sym = nullptr;
    hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(c_prop_tables));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_build_prop_tables, dim3(1), dim3(64), 0, st, (PropTables*)sym);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    PropTables host;
    if ((e = hipMemcpyAsync(&host, sym, sizeof(host), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    *hot_structure = prop_tables_have_hot_structure(host);
    return hipSuccess;
}

// `emit` (or null): where the step also writes the Lighting pass's gather copy of `dst` (PropArgs)
// `mode`: 0 the general form only (rounds 1-5's kernel), 1 hot form, one colour volume per thread, 3 hot form, the three colours of a cell in one thread
hipError_t launch_lpv_propagate(const VolumeArg src[3], const VolumeArg dst[3], uint32_t num_cascades, const LpvPackEmit* emit, int mode, hipStream_t st) {
    PropArgs a = {};
    for (int i = 0; i < 3; i++) { a.src[i] = src[i]; a.dst[i] = dst[i]; }
    a.num_cascades = num_cascades;
    a.hot = mode != 0 ? 1u : 0u;
    if (emit) {
        a.packed = emit->packed;
        a.pk_row_pitch = emit->row_pitch;
        a.pk_slice_pitch = emit->slice_pitch;
        a.state = emit->state;
    }
    if (mode != 0) {
        if (emit) {
            const hipError_t me = hipMemsetAsync(&emit->state->nonfinite, 0, sizeof(uint32_t), st);  // the copy's verdict starts at "finite"
            if (me != hipSuccess) return me;
        }
        const dim3 grid(num_cascades * 128, mode == 3 ? 1 : 3);
        if (mode == 3) {
            if (emit) hipLaunchKernelGGL((k_lpv_propagate_hot<3, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_lpv_propagate_hot<3, false>), grid, dim3(256), 0, st, a);
        } else {
            if (emit) hipLaunchKernelGGL((k_lpv_propagate_hot<1, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_lpv_propagate_hot<1, false>), grid, dim3(256), 0, st, a);
        }
        return hipGetLastError();
    }
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
