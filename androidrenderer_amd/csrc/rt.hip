// Ray tracing: acceleration structure + the two generators whose outputs sah_lighting consumes.
//   reference: RenderCore/render/raytracing_scene.cpp:15-170 (one TLAS instance per primitive, SOLID opaque / CUTOUT non-opaque),
//              RenderCore/shaders/ao/rtao.comp.slang:54-102, RenderCore/shaders/lighting/directional_light.rt.slang:91-125,
//              RenderCore/shaders/materials/gltf_basic_pbr.slang:291-325 (occlusion any-hit / closest-hit), sky_unified.slang:210-215 (miss)
// What a ray hits is defined by include/sah_hip.h ("ray tracing"): a candidate is a triangle whose PADDED BOX the ray passes (fp32 slab
// test) and whose watertight fp32 test gives tmin < t < tmax.  Because the box test is part of the definition and is monotone under box
// inclusion, any hierarchy of enclosing boxes culls exactly; the one built here is the simplest that needs no cross-workgroup
// synchronisation: triangles sorted by the Morton code of their box centre, groups of four consecutive triangles under a level-0 node,
// groups of four consecutive nodes under a node of the next level, one launch per level (the per-XCD L2s are not coherent, so a
// bottom-up build with arrival counters would need an agent-scope fence per node).  Pointerless: children of node i are 4i .. 4i + 3.
// Not tuned ("do not tune it this round"): one ray per lane, private stack, divergent loops.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "numerics.hpp"
#include "rt_args.hpp"
#include "texture_sample.hpp"

namespace sah {
namespace {

SAH_DEV bool finite3(const float v[3]) {
    return __builtin_fabsf(v[0]) < __builtin_inff() && __builtin_fabsf(v[1]) < __builtin_inff() && __builtin_fabsf(v[2]) < __builtin_inff();
}
SAH_DEV float pick(const float v[3], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }
// order-preserving float <-> uint (for atomicMin / atomicMax on floats of either sign)
SAH_DEV uint32_t ordered(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
SAH_DEV float unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// ---- build ------------------------------------------------------------------------------------------------------------------------
// triangles per primitive -> exclusive offsets (one workgroup; a scene has thousands of primitives at most)
__global__ __launch_bounds__(1024) void k_rt_scan(const sah_primitive* prims, uint32_t n, uint32_t* tri_base, RtBuildState* st) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < n ? prims[i].index_count / 3u : 0u;
        uint32_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_base = 0;
        for (uint32_t w = 0; w < wave; w++) wave_base += s_wave[w];
        const uint32_t carry = s_carry;
        if (i < n) tri_base[i] = carry + wave_base + incl - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wave_base + incl;
        __syncthreads();
    }
    if (tid == 0) {
        st->total = s_carry;
        st->kept = 0;
        st->dropped = 0;
        st->max_abs_bits = 0;
        for (int k = 0; k < 3; k++) {
            st->cmin[k] = 0xffffffffu;
            st->cmax[k] = 0u;
        }
    }
}

SAH_DEV uint32_t find_primitive(const uint32_t* tri_base, uint32_t n, uint32_t t) {  // last p with tri_base[p] <= t
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tri_base[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

SAH_DEV float mat_row3(const float* m, int r, float x, float y, float z) { return ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r]; }

// one thread per triangle of every primitive: world-space vertices (the rasteriser's vertex stage), validity, scene extents
__global__ __launch_bounds__(256) void k_rt_world(const RtScene sc, const uint32_t* tri_base, RtTriangle* out, RtBuildState* st) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = st->total;
    bool valid = false;
    RtTriangle r;
    float amax = 0.f, centre[3] = {0.f, 0.f, 0.f};
    if (t < total) {
        const uint32_t p = find_primitive(tri_base, sc.num_primitives, t);
        const sah_primitive prim = sc.primitives[p];
        const uint32_t tri = t - tri_base[p];
        float v[3][3];
        valid = (uint64_t)prim.first_index + 3ull * tri + 2ull < (uint64_t)sc.num_indices;
        for (int k = 0; k < 3 && valid; k++) {
            const int64_t vi = (int64_t)prim.vertex_offset + (int64_t)sc.indices[prim.first_index + 3u * tri + (uint32_t)k];
            valid = vi >= 0 && vi < (int64_t)sc.num_vertices;
            if (!valid) break;
            const float* pos = sc.positions + 3 * vi;
            for (int c = 0; c < 3; c++) v[k][c] = mat_row3(prim.model, c, pos[0], pos[1], pos[2]);
            valid = finite3(v[k]);
        }
        if (valid) {
            for (int c = 0; c < 3; c++) {
                r.v0[c] = v[0][c];
                r.v1[c] = v[1][c];
                r.v2[c] = v[2][c];
                const float lo = __builtin_fminf(__builtin_fminf(v[0][c], v[1][c]), v[2][c]), hi = __builtin_fmaxf(__builtin_fmaxf(v[0][c], v[1][c]), v[2][c]);
                amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(lo), __builtin_fabsf(hi)));
                centre[c] = lo * 0.5f + hi * 0.5f;
            }
            r.primitive = p;
            r.triangle = tri;
            r.flags = prim.type == SAH_PRIMITIVE_TYPE_CUTOUT ? 1u : 0u;
        } else {
            r.primitive = 0xffffffffu;  // marks the slot as left out
            r.triangle = 0;
            r.flags = 0;
            for (int c = 0; c < 3; c++) r.v0[c] = r.v1[c] = r.v2[c] = 0.f;
        }
        out[t] = r;
    }
    // wave-aggregated statistics
    const uint64_t mv = __ballot(valid), ma = __ballot(t < total);
    float wmax = amax;
    uint32_t cmn[3], cmx[3];
    for (int c = 0; c < 3; c++) {
        cmn[c] = valid ? ordered(centre[c]) : 0xffffffffu;
        cmx[c] = valid ? ordered(centre[c]) : 0u;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        wmax = __builtin_fmaxf(wmax, __shfl_xor(wmax, d, 64));
        for (int c = 0; c < 3; c++) {
            cmn[c] = min(cmn[c], (uint32_t)__shfl_xor((int)cmn[c], d, 64));
            cmx[c] = max(cmx[c], (uint32_t)__shfl_xor((int)cmx[c], d, 64));
        }
    }
    if ((threadIdx.x & 63u) == 0 && ma) {
        const uint32_t nv = (uint32_t)__builtin_popcountll(mv), na = (uint32_t)__builtin_popcountll(ma);
        if (nv) {
            atomicAdd(&st->kept, nv);
            atomicMax(&st->max_abs_bits, __float_as_uint(wmax));
            for (int c = 0; c < 3; c++) {
                atomicMin(&st->cmin[c], cmn[c]);
                atomicMax(&st->cmax[c], cmx[c]);
            }
        }
        if (na - nv) atomicAdd(&st->dropped, na - nv);
    }
}

SAH_DEV uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// sort keys: (30-bit Morton code of the box centre) << 32 | running triangle; left-out and padding slots sort to the end
__global__ __launch_bounds__(256) void k_rt_keys(const RtTriangle* tris, const RtBuildState* st, unsigned long long* keys, uint32_t padded) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= padded) return;
    unsigned long long key = ~0ull;
    if (t < st->total && tris[t].primitive != 0xffffffffu) {
        const RtTriangle r = tris[t];
        uint32_t q[3];
        for (int c = 0; c < 3; c++) {
            const float lo = __builtin_fminf(__builtin_fminf(r.v0[c], r.v1[c]), r.v2[c]), hi = __builtin_fmaxf(__builtin_fmaxf(r.v0[c], r.v1[c]), r.v2[c]);
            const float centre = lo * 0.5f + hi * 0.5f;
            const float bmin = unordered(st->cmin[c]), ext = unordered(st->cmax[c]) - bmin;
            float f = ext > 0.f ? (centre - bmin) / ext * 1023.0f : 0.f;
            f = __builtin_fminf(__builtin_fmaxf(f, 0.f), 1023.0f);  // (NaN -> 0)
            q[c] = (uint32_t)f;
        }
        const uint32_t code = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
        key = ((unsigned long long)code << 32) | t;
    }
    keys[t] = key;
}

// Bitonic network on 64-bit keys (all distinct except the ~0 padding): chunks of kRtSortChunk keys in LDS for partner distances below
// the chunk size, one global compare-exchange pass per larger distance.
SAH_DEV void cmpx(unsigned long long& a, unsigned long long& b, bool ascending) {
    if ((a > b) == ascending) {
        const unsigned long long t = a;
        a = b;
        b = t;
    }
}
__global__ __launch_bounds__(256) void k_rt_sort_local(unsigned long long* keys, uint32_t k_first, uint32_t k_last) {
    __shared__ unsigned long long s[kRtSortChunk];
    const uint32_t base = blockIdx.x * kRtSortChunk;
    for (uint32_t i = threadIdx.x; i < kRtSortChunk; i += 256u) s[i] = keys[base + i];
    __syncthreads();
    for (uint32_t k = k_first; k <= k_last; k <<= 1) {
        for (uint32_t j = min(k >> 1, kRtSortChunk >> 1); j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < kRtSortChunk / 2; t += 256u) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
                cmpx(s[i], s[i | j], ((base + i) & k) == 0u);
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < kRtSortChunk; i += 256u) keys[base + i] = s[i];
}
__global__ __launch_bounds__(256) void k_rt_sort_global(unsigned long long* keys, uint32_t pairs, uint32_t j, uint32_t k) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= pairs) return;
    const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
    unsigned long long a = keys[i], b = keys[i | j];
    const unsigned long long a0 = a;
    cmpx(a, b, (i & k) == 0u);
    if (a != a0) {
        keys[i] = a;
        keys[i | j] = b;
    }
}

SAH_DEV void tri_box(const RtTriangle& r, float pad, float lo[3], float hi[3]) {
    for (int c = 0; c < 3; c++) {
        lo[c] = __builtin_fminf(__builtin_fminf(r.v0[c], r.v1[c]), r.v2[c]) - pad;
        hi[c] = __builtin_fmaxf(__builtin_fmaxf(r.v0[c], r.v1[c]), r.v2[c]) + pad;
    }
}

// one thread per level-0 node: gathers its (up to) four triangles into Morton order and writes the union of their padded boxes
__global__ __launch_bounds__(256) void k_rt_leaves(const RtTriangle* unsorted, const unsigned long long* keys, uint32_t num_tris, float pad,
                                                   RtTriangle* sorted, RtNode* nodes, uint32_t num_nodes) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= num_nodes) return;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t k = 0; k < kRtFanout; k++) {
        const uint32_t i = n * kRtFanout + k;
        if (i >= num_tris) break;
        const RtTriangle r = unsorted[(uint32_t)keys[i]];
        sorted[i] = r;
        float l[3], h[3];
        tri_box(r, pad, l, h);
        for (int c = 0; c < 3; c++) {
            lo[c] = __builtin_fminf(lo[c], l[c]);
            hi[c] = __builtin_fmaxf(hi[c], h[c]);
        }
    }
    nodes[n] = RtNode{{lo[0], lo[1], lo[2]}, hi[0], hi[1], hi[2], 0.f, 0.f};
}
__global__ __launch_bounds__(256) void k_rt_level(const RtNode* children, uint32_t num_children, RtNode* nodes, uint32_t num_nodes) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= num_nodes) return;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t k = 0; k < kRtFanout; k++) {
        const uint32_t i = n * kRtFanout + k;
        if (i >= num_children) break;
        const RtNode c = children[i];
        lo[0] = __builtin_fminf(lo[0], c.lo[0]); lo[1] = __builtin_fminf(lo[1], c.lo[1]); lo[2] = __builtin_fminf(lo[2], c.lo[2]);
        hi[0] = __builtin_fmaxf(hi[0], c.hi0); hi[1] = __builtin_fmaxf(hi[1], c.hi1); hi[2] = __builtin_fmaxf(hi[2], c.hi2);
    }
    nodes[n] = RtNode{{lo[0], lo[1], lo[2]}, hi[0], hi[1], hi[2], 0.f, 0.f};
}

// ---- traversal --------------------------------------------------------------------------------------------------------------------
struct Ray {
    float o[3], d[3], inv[3];
    float tmin, tmax;
    int kx, ky, kz;
    float Sx, Sy, Sz;
    bool finite;
};
SAH_DEV Ray make_ray(const float o[3], const float d[3], float tmin, float tmax) {
    Ray r;
    for (int c = 0; c < 3; c++) {
        r.o[c] = o[c];
        r.d[c] = d[c];
        r.inv[c] = 1.0f / d[c];
    }
    r.tmin = tmin;
    r.tmax = tmax;
    r.finite = finite3(o) && finite3(d);  // a ray with a non-finite origin or direction hits nothing (sah_hip.h)
    const float ax = __builtin_fabsf(d[0]), ay = __builtin_fabsf(d[1]), az = __builtin_fabsf(d[2]);
    int kz = 0;
    float am = ax;
    if (ay > am) { kz = 1; am = ay; }
    if (az > am) { kz = 2; }
    int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
    if (pick(d, kz) < 0.0f) { const int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    const float dz = pick(d, kz);
    r.Sx = pick(d, kx) / dz;
    r.Sy = pick(d, ky) / dz;
    r.Sz = 1.0f / dz;
    return r;
}
SAH_DEV bool slab(const Ray& r, const float lo[3], const float hi[3]) {
    float tn = r.tmin, tf = r.tmax;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t0 = (lo[c] - r.o[c]) * r.inv[c], t1 = (hi[c] - r.o[c]) * r.inv[c];
        tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1));
        tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
    }
    return tn <= tf;
}
struct Hit {
    float t, b1, b2;
};
SAH_DEV bool woop(const Ray& r, const RtTriangle& tr, Hit& h) {
    float A[3], B[3], C[3];
    for (int c = 0; c < 3; c++) {
        A[c] = tr.v0[c] - r.o[c];
        B[c] = tr.v1[c] - r.o[c];
        C[c] = tr.v2[c] - r.o[c];
    }
    const float Akz = pick(A, r.kz), Bkz = pick(B, r.kz), Ckz = pick(C, r.kz);
    const float Ax = pick(A, r.kx) - r.Sx * Akz, Ay = pick(A, r.ky) - r.Sy * Akz;
    const float Bx = pick(B, r.kx) - r.Sx * Bkz, By = pick(B, r.ky) - r.Sy * Bkz;
    const float Cx = pick(C, r.kx) - r.Sx * Ckz, Cy = pick(C, r.ky) - r.Sy * Ckz;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = (U + V) + W;
    if (det == 0.0f) return false;
    const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
    const float T = (U * Az + V * Bz) + W * Cz;
    const float t = T / det;
    if (!(t > r.tmin && t < r.tmax)) return false;
    h.t = t;
    h.b1 = V / det;
    h.b2 = W / det;
    return true;
}

// unpackUnorm4x8ToHalf / packUnorm4x8 of gltf_basic_pbr.slang:257-276, alpha channel only (the any-hit stage uses nothing else of v.color)
SAH_DEV Hn unpack_alpha(uint32_t packed) { return Hn((float)(packed >> 24)) / Hn::lit(255.0f); }
SAH_DEV uint32_t to_uint_sat(float f) { return f > 0.0f ? (f >= 4294967296.0f ? 0xffffffffu : (uint32_t)f) : 0u; }  // negative, NaN -> 0

// any-hit stage of the occlusion hit group of a CUTOUT primitive (gltf_basic_pbr.slang:291-318): true = the hit is accepted
SAH_DEV bool cutout_accepts(const RtScene& sc, const RtTriangle& tr, const Hit& h) {
    const sah_primitive& prim = sc.primitives[tr.primitive];
    const uint32_t* idx = sc.indices + prim.first_index + 3u * tr.triangle;
    const sah_vertex_data& a = sc.vertex_data[(int64_t)prim.vertex_offset + idx[0]];
    const sah_vertex_data& b = sc.vertex_data[(int64_t)prim.vertex_offset + idx[1]];
    const sah_vertex_data& c = sc.vertex_data[(int64_t)prim.vertex_offset + idx[2]];
    const float b0 = (1.0f - h.b1) - h.b2;
    float uv[2];
    for (int k = 0; k < 2; k++) uv[k] = (b0 * a.texcoord[k] + h.b1 * b.texcoord[k]) + h.b2 * c.texcoord[k];
    // v.color = packUnorm4x8(bary.x * unpack(v0.color) + ...): float * half4 sums in fp32, the sum converted to half, * 255.h in half, truncated
    const float ca = (b0 * tof(unpack_alpha(a.color)) + h.b1 * tof(unpack_alpha(b.color))) + h.b2 * tof(unpack_alpha(c.color));
    const uint32_t alpha_byte = to_uint_sat(tof(Hn(ca) * Hn::lit(255.0f))) & 0xffu;
    const Hn colour_a = Hn((float)alpha_byte) / Hn::lit(255.0f);
    const sah_material& m = sc.materials[prim.material];
    float texel_a = m.base_color_texel[3];
    if (sc.material_textures) {
        const uint32_t ti = sc.material_textures[prim.material].base_color;
        if (ti != SAH_TEXTURE_NONE) {
            float texel[4];
            sample_texture_lod(sc.luts, sc.textures[ti], uv, 0.0f, 0.0f, texel);
            texel_a = texel[3];
        }
    }
    const float alpha = (texel_a * m.base_color_tint[3]) * tof(colour_a);
    return !(alpha <= m.opacity_threshold);
}

// "is there an accepted candidate" (RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH); CULL_NON_OPAQUE: CUTOUT primitives do not exist for this ray
template <bool CULL_NON_OPAQUE>
SAH_DEV bool any_hit(const RtBvh& bvh, const RtScene& sc, const Ray& r) {
    if (!r.finite || bvh.num_tris == 0) return false;
    uint32_t stack[3 * kRtMaxLevels + 4];
    int sp = 0;
    stack[sp++] = ((bvh.num_levels - 1u) << 28);  // the top level's single node
    while (sp > 0) {
        const uint32_t e = stack[--sp];
        const uint32_t level = e >> 28, node = e & 0x0fffffffu;
        if (level == 0) {
            for (uint32_t k = 0; k < kRtFanout; k++) {
                const uint32_t i = node * kRtFanout + k;
                if (i >= bvh.num_tris) break;
                const RtTriangle tr = bvh.tris[i];
                if (CULL_NON_OPAQUE && (tr.flags & 1u)) continue;
                float lo[3], hi[3];
                tri_box(tr, bvh.pad, lo, hi);
                if (!slab(r, lo, hi)) continue;
                Hit h;
                if (!woop(r, tr, h)) continue;
                if (!(tr.flags & 1u) || cutout_accepts(sc, tr, h)) return true;
            }
        } else {
            const uint32_t count = bvh.level_count[level - 1u];
            const RtNode* ch = bvh.nodes + bvh.level_offset[level - 1u];
            for (uint32_t k = 0; k < kRtFanout; k++) {
                const uint32_t i = node * kRtFanout + k;
                if (i >= count) break;
                const RtNode n = ch[i];
                const float lo[3] = {n.lo[0], n.lo[1], n.lo[2]}, hi[3] = {n.hi0, n.hi1, n.hi2};
                if (slab(r, lo, hi)) stack[sp++] = ((level - 1u) << 28) | i;
            }
        }
    }
    return false;
}

// ---- generators -------------------------------------------------------------------------------------------------------------------
// get_worldspace_position / get_worldspace_location of rtao.comp.slang:27-36 and directional_light.rt.slang:39-48
SAH_DEV void world_position(const float* inv_proj, const float* inv_view, const float res[2], uint32_t x, uint32_t y, float depth, float out[3]) {
    const Fn tx = (Fn((float)x) + Fn(0.5f)) / Fn(res[0]);
    const Fn ty = (Fn((float)y) + Fn(0.5f)) / Fn(res[1]);
    const F4 ndc = {tx * Fn(2.0f) - Fn(1.0f), ty * Fn(2.0f) - Fn(1.0f), Fn(depth), Fn(1.0f)};
    F4 vs = mul44(inv_proj, ndc);
    vs = {vs.x / vs.w, vs.y / vs.w, vs.z / vs.w, vs.w / vs.w};
    const F4 ws = mul44(inv_view, vs);
    out[0] = ws.x.v;
    out[1] = ws.y.v;
    out[2] = ws.z.v;
}
SAH_DEV H3 load_normal_h(const PlaneArg& p, uint32_t x, uint32_t y) {
    const uint2 w = *reinterpret_cast<const uint2*>(p.ptr + (size_t)y * p.pitch + (size_t)x * 8);
    const H3 n = {Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(w.x & 0xffffu))), Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(w.x >> 16))),
                  Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(w.y & 0xffffu)))};
    return normalize(n);
}
// normalize(noisetex[pixel].rgb * 2 - 1) in fp32
SAH_DEV F3 load_noise(const PlaneArg& p, const float* luts, uint32_t x, uint32_t y) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(p.ptr + (size_t)y * p.pitch + (size_t)x * 4);
    const F3 v = {Fn(luts[256u + (w & 0xffu)]) * Fn(2.0f) - Fn(1.0f), Fn(luts[256u + ((w >> 8) & 0xffu)]) * Fn(2.0f) - Fn(1.0f),
                  Fn(luts[256u + ((w >> 16) & 0xffu)]) * Fn(2.0f) - Fn(1.0f)};
    return normalize(v);
}

__global__ __launch_bounds__(256) void k_rtao(const RtaoArgs a, const RtBvh bvh, const RtScene sc) {
    const uint32_t x = blockIdx.x * 16u + (threadIdx.x & 15u), y = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (x >= a.width || y >= a.height) return;
    const float depth = *reinterpret_cast<const float*>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x * 4);
    float o[3];
    world_position(a.inv_proj, a.inv_view, a.res, x, y, depth, o);
    const H3 normal = load_normal_h(a.normals, x, y);
    F3 noise = load_noise(a.noise, sc.luts, x % a.noise_w, y % a.noise_h);
    if (dot(noise, to_f(normal)).v < 0.0f) noise = noise * Fn(-1.0f);
    const float d[3] = {noise.x.v, noise.y.v, noise.z.v};
    const Ray r = make_ray(o, d, 0.01f, a.max_distance);
    const bool hit = any_hit<true>(bvh, sc, r);
    // every one of the spp rays is this ray (the shader reads the same noise texel for each): ao = spp - spp or spp, exact for spp <= 4096
    const float spp = (float)a.samples;
    const float ao = (hit ? spp - spp : spp) / spp;
    *reinterpret_cast<float*>(const_cast<uint8_t*>(a.out.ptr) + (size_t)y * a.out.pitch + (size_t)x * 4) = ao;
}

__global__ __launch_bounds__(256) void k_sun_shadow_mask(const ShadowMaskArgs a, const RtBvh bvh, const RtScene sc) {
    const uint32_t x = blockIdx.x * 16u + (threadIdx.x & 15u), y = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (x >= a.width || y >= a.height) return;
    float* dst = reinterpret_cast<float*>(const_cast<uint8_t*>(a.out.ptr) + (size_t)y * a.out.pitch + (size_t)x * 4);
    const float depth = *reinterpret_cast<const float*>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x * 4);
    const F3 L = {Fn(a.L[0]), Fn(a.L[1]), Fn(a.L[2])};
    const H3 normal = load_normal_h(a.normals, x, y);
    const Hn ndotl = Hn(nclamp(dot(L, to_f(normal)), Fn(0.f), Fn(1.f)).v);
    if (depth == 0.0f || !(tof(ndotl) > 0.0f)) {
        *dst = 1.0f;
        return;
    }
    float o[3];
    world_position(a.inv_proj, a.inv_view, a.res, x, y, depth, o);
    const Fn phi = Fn(1.618033988749895f);
    Fn shadow = Fn(0.0f);
    for (uint32_t i = 0; (float)i < a.num_samples; i++) {
        const Fn q = Fn((float)i) / phi;
        const Fn r0x = Fn(2.0f) + q, r0y = Fn(3.0f) + q;
        const Fn fx = r0x - Fn(__builtin_floorf(r0x.v)), fy = r0y - Fn(__builtin_floorf(r0y.v));
        const float offx = __builtin_rintf((fx * Fn(128.0f)).v), offy = __builtin_rintf((fy * Fn(128.0f)).v);
        const uint32_t nx = to_uint_sat((Fn((float)x) + Fn(offx)).v) % 128u, ny = to_uint_sat((Fn((float)y) + Fn(offy)).v) % 128u;
        const F3 noise = load_noise(a.noise, sc.luts, nx, ny);
        const F3 dir = normalize(L + noise * Fn(a.tan_size));
        const float d[3] = {dir.x.v, dir.y.v, dir.z.v};
        const Ray r = make_ray(o, d, 0.01f, 100000.0f);
        shadow = shadow + Fn(any_hit<false>(bvh, sc, r) ? 0.0f : 1.0f);
    }
    *dst = (shadow / Fn(a.num_samples)).v;
}

}  // namespace

// ---- launchers ----------------------------------------------------------------------------------------------------------------------
hipError_t launch_rt_scan(const sah_primitive* prims, uint32_t n, uint32_t* tri_base, RtBuildState* st, hipStream_t s) {
    hipLaunchKernelGGL(k_rt_scan, dim3(1), dim3(1024), 0, s, prims, n, tri_base, st);
    return hipGetLastError();
}
hipError_t launch_rt_world(const RtScene& sc, const uint32_t* tri_base, uint32_t total, RtTriangle* out, RtBuildState* st, hipStream_t s) {
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rt_world, dim3((total + 255u) / 256u), dim3(256), 0, s, sc, tri_base, out, st);
    return hipGetLastError();
}
// keys of `total` slots padded to `padded` (a power of two, >= kRtSortChunk), sorted ascending
hipError_t launch_rt_sort(const RtTriangle* tris, const RtBuildState* st, unsigned long long* keys, uint32_t padded, hipStream_t s) {
    hipLaunchKernelGGL(k_rt_keys, dim3((padded + 255u) / 256u), dim3(256), 0, s, tris, st, keys, padded);
    const uint32_t chunks = padded / kRtSortChunk;
    hipLaunchKernelGGL(k_rt_sort_local, dim3(chunks), dim3(256), 0, s, keys, 2u, kRtSortChunk);
    for (uint32_t k = kRtSortChunk * 2u; k <= padded; k <<= 1) {
        for (uint32_t j = k >> 1; j >= kRtSortChunk; j >>= 1)
            hipLaunchKernelGGL(k_rt_sort_global, dim3((padded / 2u + 255u) / 256u), dim3(256), 0, s, keys, padded / 2u, j, k);
        hipLaunchKernelGGL(k_rt_sort_local, dim3(chunks), dim3(256), 0, s, keys, k, k);
    }
    return hipGetLastError();
}
hipError_t launch_rt_nodes(const RtTriangle* unsorted, const unsigned long long* keys, RtTriangle* sorted, RtNode* nodes, const RtBvh& bvh, hipStream_t s) {
    if (bvh.num_tris == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rt_leaves, dim3((bvh.level_count[0] + 255u) / 256u), dim3(256), 0, s, unsorted, keys, bvh.num_tris, bvh.pad, sorted, nodes,
                       bvh.level_count[0]);
    for (uint32_t l = 1; l < bvh.num_levels; l++)
        hipLaunchKernelGGL(k_rt_level, dim3((bvh.level_count[l] + 255u) / 256u), dim3(256), 0, s, nodes + bvh.level_offset[l - 1], bvh.level_count[l - 1],
                           nodes + bvh.level_offset[l], bvh.level_count[l]);
    return hipGetLastError();
}
hipError_t launch_rtao(const RtaoArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s) {
    hipLaunchKernelGGL(k_rtao, dim3((a.width + 15u) / 16u, (a.height + 15u) / 16u), dim3(256), 0, s, a, bvh, sc);
    return hipGetLastError();
}
hipError_t launch_sun_shadow_mask(const ShadowMaskArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s) {
    hipLaunchKernelGGL(k_sun_shadow_mask, dim3((a.width + 15u) / 16u, (a.height + 15u) / 16u), dim3(256), 0, s, a, bvh, sc);
    return hipGetLastError();
}

}  // namespace sah
