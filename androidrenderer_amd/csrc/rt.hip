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

#include "lighting_common.hpp"
#include "lighting_gi_ext.hpp"
#include "numerics.hpp"
#include "octahedral.hpp"
#include "rt_args.hpp"
#include "texture_sample.hpp"

namespace sah {
namespace {

SAH_DEV bool finite3r(float a, float b, float c) {
    const float inf = __builtin_inff();
    return __builtin_fabsf(a) < inf && __builtin_fabsf(b) < inf && __builtin_fabsf(c) < inf;
}
SAH_DEV bool finite3(const float v[3]) {
    return __builtin_fabsf(v[0]) < __builtin_inff() && __builtin_fabsf(v[1]) < __builtin_inff() && __builtin_fabsf(v[2]) < __builtin_inff();
}
// v[k], k in {0, 1, 2}, as two selects on VALUES: written as a conditional expression on array elements it compiled to nested exec-mask
// regions around single moves (eighteen per triangle test), and as selects of array elements to a dynamically indexed private array.
SAH_DEV float pick3(float v0, float v1, float v2, int k) {
    float r = v0;
    r = k == 1 ? v1 : r;
    r = k == 2 ? v2 : r;
    return r;
}
SAH_DEV float pick(const float v[3], int k) { return pick3(v[0], v[1], v[2], k); }
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

// sort keys: (30-bit Hilbert index of the box centre) << 32 | running triangle; left-out and padding slots sort to the end
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
        {  // position along the Hilbert curve through the 1024^3 grid (Skilling's transpose form, then bit-interleaved): groups of four
           // consecutive triangles, and of four consecutive groups, are tighter than along the Z curve — RTAO 2.26 -> 1.68 ms
            uint32_t X[3] = {q[0], q[1], q[2]};
            const uint32_t M = 1u << 9;
            for (uint32_t Q = M; Q > 1u; Q >>= 1) {
                const uint32_t P = Q - 1u;
                for (int i = 0; i < 3; i++) {
                    if (X[i] & Q) X[0] ^= P;
                    else {
                        const uint32_t tt = (X[0] ^ X[i]) & P;
                        X[0] ^= tt;
                        X[i] ^= tt;
                    }
                }
            }
            for (int i = 1; i < 3; i++) X[i] ^= X[i - 1];
            uint32_t tt = 0;
            for (uint32_t Q = M; Q > 1u; Q >>= 1)
                if (X[2] & Q) tt ^= Q - 1u;
            for (int i = 0; i < 3; i++) X[i] ^= tt;
            q[0] = X[0]; q[1] = X[1]; q[2] = X[2];
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

// Refinement of the curve order, guided by surface area.  The curve knows only box centres: at the seams of a mesh it puts triangles of
// different planes into one group of four, whose box then has a volume where each plane's group would be flat.  Inside every aligned
// window of 1024 consecutive triangles (one node of level 5 and everything under it) a workgroup therefore re-partitions the triangles
// top-down: a segment of S slots is split into its first and second S / 2 along one of nine orders — by box centre, lower or upper corner
// on x, y or z — whichever has the smallest area(left) x count(left) + area(right) x count(right); each half again, down to the groups of
// four.  Every order is a bitonic sort of (22-bit quantised coordinate, item) words inside the level's segments in LDS — the three axes of a kind side by side, a team of 256 threads each —; the halves' boxes
// are reduced with shuffles.  Triangle sets of the levels above the window stay what the curve made them.  Slots behind the window's last
// triangle sort last in every order, so the triangles stay a prefix of every segment (the hierarchy is complete: node n exists iff
// n < count).  Which order wins changes no result (sah_hip.h: any hierarchy culls exactly), only how many boxes a ray meets:
// sum of node areas / root area of the atrium's 23 808 triangles 31.3 -> 26.6 (tools/experiments/tree_cost.py).
constexpr uint32_t kRefineWindow = 1024u;
SAH_DEV float half_area(const float lo[3], const float hi[3]) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx * dy + dy * dz) + dz * dx;
}
__global__ __launch_bounds__(768) void k_rt_refine(const RtTriangle* unsorted, unsigned long long* keys, uint32_t num_tris) {
    // 768 threads: three teams of 256 — team g sorts along axis g — so that the three orders of a kind advance between the same barriers
    // (the kernel is bound by the latency of its ~650 barrier-separated sort stages per window, not by their work)
    __shared__ float s_box[6][kRefineWindow];  // by item = slot in the window on entry; empty items hold (+inf, -inf)
    __shared__ uint32_t s_idx[kRefineWindow], s_srt[3][kRefineWindow];
    __shared__ uint16_t s_perm[kRefineWindow], s_best_perm[kRefineWindow];
    __shared__ float s_hbox[3][6][kRefineWindow / 4u];
    __shared__ uint32_t s_hcnt[3][kRefineWindow / 4u];
    __shared__ float s_best[kRefineWindow / 8u], s_cost[3][kRefineWindow / 8u];
    __shared__ uint32_t s_swap[3][kRefineWindow / 8u], s_take[kRefineWindow / 8u];
    __shared__ uint32_t s_bounds[6];
    const uint32_t base = blockIdx.x * kRefineWindow, g = threadIdx.x / 256u, tid = threadIdx.x % 256u, lane = tid & 63u;
    if (base >= num_tris) return;
    const uint32_t n_real = min(kRefineWindow, num_tris - base);
    if (n_real <= 4u) return;
    if (threadIdx.x < 3u) {
        s_bounds[threadIdx.x] = 0xffffffffu;
        s_bounds[3u + threadIdx.x] = 0u;
    }
    __syncthreads();
    if (g == 0u) {
        float wlo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, whi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        for (uint32_t j = 0; j < 4u; j++) {
            const uint32_t pos = 4u * tid + j;
            float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            uint32_t idx = 0;
            if (pos < n_real) {
                idx = (uint32_t)(keys[base + pos] & 0xffffffffull);
                tri_box(unsorted[idx], 0.0f, lo, hi);
            }
            s_idx[pos] = idx;
            s_perm[pos] = (uint16_t)pos;
            for (int c = 0; c < 3; c++) {
                s_box[c][pos] = lo[c];
                s_box[3 + c][pos] = hi[c];
                wlo[c] = __builtin_fminf(wlo[c], lo[c]);
                whi[c] = __builtin_fmaxf(whi[c], hi[c]);
            }
        }
        for (uint32_t m = 1u; m < 64u; m <<= 1)
            for (int c = 0; c < 3; c++) {
                wlo[c] = __builtin_fminf(wlo[c], __shfl_xor(wlo[c], (int)m));
                whi[c] = __builtin_fmaxf(whi[c], __shfl_xor(whi[c], (int)m));
            }
        if (lane == 0u)
            for (int c = 0; c < 3; c++) {
                atomicMin(&s_bounds[c], ordered(wlo[c]));
                atomicMax(&s_bounds[3 + c], ordered(whi[c]));
            }
    }
    __syncthreads();
    // quantisation of this team's coordinate to 22 bits over the window's extent (monotone; ties go by item)
    const float wmin = unordered(s_bounds[g]), wext = unordered(s_bounds[3u + g]) - wmin, wscale = wext > 0.f ? 4194303.0f / wext : 0.f;
    uint32_t* const srt = s_srt[g];
    for (uint32_t S = kRefineWindow; S >= 8u; S >>= 1) {
        const uint32_t H = S >> 1, nseg = kRefineWindow / S, group = min(H >> 2, 64u);  // threads whose slots lie in one half (4 slots each)
        for (uint32_t kind = 0; kind < 3u; kind++) {
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t pos = 4u * tid + j, item = s_perm[pos];
                const float lo = s_box[g][item], hi = s_box[3u + g][item];
                uint32_t q = 0x3fffffu;
                if (lo <= hi) {
                    const float v = kind == 0u ? lo * 0.5f + hi * 0.5f : (kind == 1u ? lo : hi);
                    const float f = __builtin_fminf(__builtin_fmaxf((v - wmin) * wscale, 0.f), 4194303.0f);  // (NaN -> 0)
                    q = (uint32_t)f;
                }
                srt[pos] = (q << 10) | item;
            }
            __syncthreads();
            for (uint32_t kk = 2u; kk <= S; kk <<= 1)
                for (uint32_t j = kk >> 1; j > 0u; j >>= 1) {
                    for (uint32_t t = tid; t < kRefineWindow / 2u; t += 256u) {
                        const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
                        const uint32_t x = srt[i], y = srt[i | j];
                        const bool ascending = kk == S || (i & kk) == 0u;
                        if ((x > y) == ascending) {
                            srt[i] = y;
                            srt[i | j] = x;
                        }
                    }
                    __syncthreads();
                }
            // the box and the triangle count of every half: a thread's four slots lie in one half, and so do `group` neighbouring threads
            float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t item = srt[4u * tid + j] & 1023u;
                cnt += s_box[0][item] <= s_box[3][item] ? 1u : 0u;
                for (int c = 0; c < 3; c++) {
                    lo[c] = __builtin_fminf(lo[c], s_box[c][item]);
                    hi[c] = __builtin_fmaxf(hi[c], s_box[3 + c][item]);
                }
            }
            for (uint32_t m = 1u; m < group; m <<= 1) {
                for (int c = 0; c < 3; c++) {
                    lo[c] = __builtin_fminf(lo[c], __shfl_xor(lo[c], (int)m));
                    hi[c] = __builtin_fmaxf(hi[c], __shfl_xor(hi[c], (int)m));
                }
                cnt += (uint32_t)__shfl_xor((int)cnt, (int)m);
            }
            // one record per `group` threads: record r covers slots [4 * group * r, 4 * group * (r + 1)); a half is H / (4 * group) records
            if ((tid & (group - 1u)) == 0u) {
                const uint32_t r = tid / group;
                for (int c = 0; c < 3; c++) {
                    s_hbox[g][c][r] = lo[c];
                    s_hbox[g][3 + c][r] = hi[c];
                }
                s_hcnt[g][r] = cnt;
            }
            __syncthreads();
            if (tid < nseg) {  // this team's cost of segment `tid`
                const uint32_t per_half = H / (4u * group);  // 1, or 2 for the 512-slot halves (two waves each)
                float cost = 0.f, area[2] = {0.f, 0.f};
                uint32_t total = 0;
                for (uint32_t side = 0; side < 2u; side++) {
                    float blo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, bhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
                    uint32_t bc = 0;
                    for (uint32_t r = (2u * tid + side) * per_half; r < (2u * tid + side + 1u) * per_half; r++) {
                        bc += s_hcnt[g][r];
                        for (int c = 0; c < 3; c++) {
                            blo[c] = __builtin_fminf(blo[c], s_hbox[g][c][r]);
                            bhi[c] = __builtin_fmaxf(bhi[c], s_hbox[g][3 + c][r]);
                        }
                    }
                    if (bc != 0u) {
                        area[side] = half_area(blo, bhi);
                        cost += area[side] * (float)bc;
                    }
                    total += bc;
                }
                s_cost[g][tid] = cost;
                // of a full segment's halves the one with the larger box goes second, i.e. is entered first by the any-hit walk (descending
                // child order): a ray is likelier to meet an occluder there (shadow mask 1.65 -> 1.59 ms; the smaller one second: 1.90)
                s_swap[g][tid] = total == S && area[0] > area[1] ? 1u : 0u;
            }
            __syncthreads();
            if (g == 0u && tid < nseg) {  // the cheapest of the three, against the best of the kinds before: 0 = keep, else 1 + 2 * team + swap
                uint32_t take = 0;
                float best = kind == 0u ? __builtin_inff() : s_best[tid];
                for (uint32_t t = 0; t < 3u; t++) {
                    const float c = s_cost[t][tid];
                    if ((kind == 0u && t == 0u) || c < best) {
                        best = c;
                        take = 1u + 2u * t + s_swap[t][tid];
                    }
                }
                s_best[tid] = best;
                s_take[tid] = take;
            }
            __syncthreads();
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t pos = 4u * tid + j;
                const uint32_t tk = s_take[pos / S];
                if (tk != 0u && (tk - 1u) / 2u == g) s_best_perm[((tk - 1u) & 1u) ? pos ^ H : pos] = (uint16_t)(srt[pos] & 1023u);
            }
            __syncthreads();
        }
        if (g == 0u)
            for (uint32_t j = 0; j < 4u; j++) s_perm[4u * tid + j] = s_best_perm[4u * tid + j];
        __syncthreads();
    }
    if (g == 0u)
        for (uint32_t pos = tid; pos < n_real; pos += 256u) keys[base + pos] = (keys[base + pos] & 0xffffffff00000000ull) | s_idx[s_perm[pos]];
}

// The lanes of a level's last group that stand for no node hold the box [+inf, +inf]^3, which no ray passes: RN((+inf - o) * inv) is
// +inf on both planes of an axis where inv > 0 (entry = +inf > exit = min(tmax, +inf): make_ray keeps tmax FINITE, and the closest-hit
// walk only ever lowers it) and -inf where inv < 0 (exit = -inf < entry); inv is never 0 or NaN for a ray that walks (non-finite rays do
// not).  The walk then needs no "does this child exist" test.
SAH_DEV void fill_absent(RtNodeGroup& g, uint32_t first_absent) {
    if (first_absent == 0u) return;  // the group is full
    for (uint32_t k = first_absent; k < kRtFanout; k++)
        for (int c = 0; c < 3; c++) g.lo[c][k] = g.hi[c][k] = __builtin_inff();
}
// one thread per triangle: moves it into curve order and writes its padded box — the level-0 "node" of the hierarchy, so that the walk
// meets a triangle's own box (part of the hit definition, sah_hip.h) like any other box
__global__ __launch_bounds__(256) void k_rt_leaves(const RtTriangle* unsorted, const unsigned long long* keys, uint32_t num_tris, float pad,
                                                   RtTriangle* sorted, RtNodeGroup* nodes) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= num_tris) return;
    const RtTriangle r = unsorted[(uint32_t)(keys[i] & 0xffffffffull)];
    sorted[i] = r;
    float lo[3], hi[3];
    tri_box(r, pad, lo, hi);
    RtNodeGroup& g = nodes[i / kRtFanout];
    for (int c = 0; c < 3; c++) {
        g.lo[c][i % kRtFanout] = lo[c];
        g.hi[c][i % kRtFanout] = hi[c];
    }
    if (i + 1u == num_tris) fill_absent(g, num_tris % kRtFanout);
}
__global__ __launch_bounds__(256) void k_rt_level(const RtNodeGroup* children, uint32_t num_children, RtNodeGroup* nodes, uint32_t num_nodes) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= num_nodes) return;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    const RtNodeGroup c = children[n];  // (the lanes of a level's last group beyond its node count hold nothing)
    for (uint32_t k = 0; k < kRtFanout; k++) {
        if (n * kRtFanout + k >= num_children) break;
        for (int a = 0; a < 3; a++) {
            lo[a] = __builtin_fminf(lo[a], c.lo[a][k]);
            hi[a] = __builtin_fmaxf(hi[a], c.hi[a][k]);
        }
    }
    RtNodeGroup& g = nodes[n / kRtFanout];
    for (int a = 0; a < 3; a++) {
        g.lo[a][n % kRtFanout] = lo[a];
        g.hi[a][n % kRtFanout] = hi[a];
    }
    if (n + 1u == num_nodes) fill_absent(g, num_nodes % kRtFanout);
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
    // tmax is at most the largest finite float (sah_hip.h "ray tracing": an infinite or NaN distance means that).  fill_absent()'s boxes
    // rely on it: with tmax = +inf a ray whose direction has no negative component would pass them and walk into nodes that do not exist
    r.tmax = __builtin_fminf(tmax, 3.402823466e+38f);
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
    r.Sz = pick(r.inv, kz);  // 1.0f / dz, already there
    return r;
}
SAH_DEV bool slab(const Ray& r, const float lo[3], const float hi[3], float* entry = nullptr) {
    float tn = r.tmin, tf = r.tmax;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t0 = (lo[c] - r.o[c]) * r.inv[c], t1 = (hi[c] - r.o[c]) * r.inv[c];
        tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1));
        tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
    }
    if (entry) *entry = tn;
    return tn <= tf;
}
struct Hit {
    float t, b1, b2;
    bool front;  // (v1 - v0) x (v2 - v0) against the ray (sah_hip.h "Facing"): det > 0
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
    h.front = det > 0.0f;
    return true;
}

// unpackUnorm4x8ToHalf / packUnorm4x8 of gltf_basic_pbr.slang:257-276, alpha channel only (the any-hit stage uses nothing else of v.color)
SAH_DEV Hn unpack_alpha(uint32_t packed) { return Hn((float)(packed >> 24)) / Hn::lit(255.0f); }
SAH_DEV uint32_t to_uint_sat(float f) { return cvt_u32_sat(f); }  // negative, NaN -> 0; saturating

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

// ---- walking the hierarchy ---------------------------------------------------------------------------------------------------------
// The hierarchy is implicit and complete (node n of level L covers nodes 4n .. 4n + 3 of level L - 1; level 0 nodes cover four triangles),
// so a depth-first walk needs no stack in memory: the position is (level, node), and what is left to visit on the way back up is four
// bits per level — the children of the path's node at that level that the ray's slab test passed and that have not been entered yet.
// Per-level offsets sit in LDS because the level is a per-lane value (the counts are not needed: fill_absent).
struct Trav {
    const uint32_t* off;
};
SAH_DEV Trav trav_init(const RtBvh& bvh, uint32_t* smem /* kRtMaxLevels words of LDS; every thread of the workgroup calls this */) {
    // (a chain of selects on the kernel argument's words, then one store: `if (threadIdx.x == l) smem[l] = ...` fifteen times compiled to a
    //  decision tree of sixty exec-mask regions at the head of every workgroup)
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t l = 0; l < kRtMaxLevels; l++) mine = threadIdx.x == l ? bvh.level_offset[l] : mine;
    if (threadIdx.x < kRtMaxLevels) smem[threadIdx.x] = mine;
    __syncthreads();
    return {smem};
}
// bit k: child 4 * node + k of (level, node) exists and the ray's slab test passes its box.  The four boxes are one 96-byte group
// `nearest` (optional): among the passing children, the one the ray enters first (the largest index among equals)
SAH_DEV uint32_t children_hit(const RtBvh& bvh, const Trav& tv, const Ray& r, uint32_t level, uint32_t node, uint32_t* nearest = nullptr) {
    const float4* p = reinterpret_cast<const float4*>(bvh.nodes + tv.off[level - 1u] + node);
    float q[6][4];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const float4 v = p[k];
        q[k][0] = v.x; q[k][1] = v.y; q[k][2] = v.z; q[k][3] = v.w;
    }
    uint32_t m = 0, best = 0;
    float best_t = __builtin_inff();
#pragma unroll
    for (uint32_t k = 0; k < kRtFanout; k++) {
        const float lo[3] = {q[0][k], q[1][k], q[2][k]}, hi[3] = {q[3][k], q[4][k], q[5][k]};
        float tn;
        const bool pass = slab(r, lo, hi, &tn);  // (a child that does not exist holds a box nothing passes: fill_absent)
        if (pass) m |= 1u << k;
        if (nearest && pass && tn <= best_t) {
            best_t = tn;
            best = k;
        }
    }
    if (nearest) *nearest = best;
    return m;
}
SAH_DEV RtTriangle load_triangle(const RtTriangle* tris, uint32_t i) {  // three 16-byte loads
    const float4* p = reinterpret_cast<const float4*>(tris + i);
    float4 a = p[0], b = p[1], c = p[2];
    // all three in flight together: left alone the compiler fetches the flags word first and the vertices only where the flags let the
    // triangle be tested — two memory round trips, one behind the other, per candidate
    asm volatile("" : "+v"(a.x), "+v"(b.x), "+v"(c.w));
    RtTriangle t;
    t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z; t.primitive = __builtin_bit_cast(uint32_t, a.w);
    t.v1[0] = b.x; t.v1[1] = b.y; t.v1[2] = b.z; t.triangle = __builtin_bit_cast(uint32_t, b.w);
    t.v2[0] = c.x; t.v2[1] = c.y; t.v2[2] = c.z; t.flags = __builtin_bit_cast(uint32_t, c.w);
    return t;
}
// next position of the walk after (level, node) left `m` of its children to visit; false when the walk is over
// `prefer`: the child to enter first when (level, node) has just been tested (children_hit's nearest); 4 = none.  Children left for
// later are entered in descending index order (measured: DESIGN.md §5f)
SAH_DEV bool trav_next(uint32_t top, uint32_t& level, uint32_t& node, unsigned long long& pending, uint32_t m, uint32_t prefer = 4u) {
    const bool fresh = m != 0u && prefer < 4u;
    if (m == 0u) {  // up to the nearest level that has children left: the first non-zero nibble of `pending` above this level's (no loop:
                    // as one, the lanes of a wave climbed one level per iteration, all waiting for the one that climbs furthest)
        const unsigned long long above = pending >> (4u * level + 4u);
        if (above == 0ull) return false;
        const uint32_t up = (uint32_t)__builtin_ctzll(above) >> 2;
        level += 1u + up;
        node >>= 2u * (1u + up);
        m = (uint32_t)(above >> (4u * up)) & 15u;
    }
    const uint32_t k = fresh ? prefer : 31u - (uint32_t)__builtin_clz(m);
    m &= ~(1u << k);
    pending = (pending & ~(15ull << (4u * level))) | ((unsigned long long)m << (4u * level));
    node = node * kRtFanout + k;
    level--;
    return true;
}

// "is there an accepted candidate" (RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH); CULL_NON_OPAQUE: CUTOUT primitives do not exist for this ray.
// Two nested loops: every lane walks boxes until it stands on a triangle whose padded box the ray passes (or is done), then the lanes
// that have one test it together.
// is triangle `i` an accepted candidate of the ray: its own padded box, the watertight test, the flags' culling, the any-hit stage
template <bool CULL_NON_OPAQUE, bool CULL_FRONT = false>
SAH_DEV bool accepts(const RtBvh& bvh, const RtScene& sc, const Ray& r, uint32_t i) {
    if (!r.finite || i >= bvh.num_tris) return false;
    const RtNodeGroup& g = bvh.nodes[i / kRtFanout];  // level 0: one box per triangle
    const uint32_t k = i % kRtFanout;
    const float lo[3] = {g.lo[0][k], g.lo[1][k], g.lo[2][k]}, hi[3] = {g.hi[0][k], g.hi[1][k], g.hi[2][k]};
    if (!slab(r, lo, hi)) return false;
    const RtTriangle tr = load_triangle(bvh.tris, i);
    if (CULL_NON_OPAQUE && (tr.flags & 1u)) return false;
    Hit h;
    return woop(r, tr, h) && !(CULL_FRONT && h.front) && (!(tr.flags & 1u) || cutout_accepts(sc, tr, h));
}

// `occluder` (optional): the triangle that ended the search
template <bool CULL_NON_OPAQUE, bool CULL_FRONT = false>
SAH_DEV bool any_hit(const RtBvh& bvh, const RtScene& sc, const Trav& tv, const Ray& r, uint32_t* occluder = nullptr) {
    if (!r.finite || bvh.num_tris == 0) return false;
    const uint32_t top = bvh.num_levels - 1u;
    uint32_t level = top, node = 0;  // the top level's single node
    unsigned long long pending = 0;
    bool alive = true;
    if (top == 0) {  // a single triangle: its box is the top node, which nothing has tested
        const RtNodeGroup& n = bvh.nodes[0];
        const float lo[3] = {n.lo[0][0], n.lo[1][0], n.lo[2][0]}, hi[3] = {n.hi[0][0], n.hi[1][0], n.hi[2][0]};
        alive = slab(r, lo, hi);
    }
    while (alive) {
        while (alive && level != 0) {
            // (entering the nearest child first, as the closest-hit walk does, does not pay here: RTAO + 4 %, shadow mask + 17 %; ascending
            //  index: shadow mask + 38 %)
            alive = trav_next(top, level, node, pending, children_hit(bvh, tv, r, level, node));
        }
        if (alive) {
            const RtTriangle tr = load_triangle(bvh.tris, node);
            if (!(CULL_NON_OPAQUE && (tr.flags & 1u))) {
                Hit h;
                if (woop(r, tr, h) && !(CULL_FRONT && h.front) && (!(tr.flags & 1u) || cutout_accepts(sc, tr, h))) {
                    if (occluder) *occluder = node;
                    return true;
                }
            }
            alive = trav_next(top, level, node, pending, 0u);
        }
    }
    return false;
}

// ---- beams -------------------------------------------------------------------------------------------------------------------------
// Rays that leave ONE point in nearly one direction (a pixel's sun samples) can share a walk.  A beam holds, per axis, the interval
// [inv_lo, inv_hi] of the rays' 1 / d — all of one sign, finite — and its slab test evaluates the ray test's own operators on the
// interval ends: x -> RN(a * x) is monotone for a fixed a, so a ray's RN((lo - o) * inv) lies between the products at the two ends, its
// entry distance is >= the beam's and its exit distance <= the beam's: every box a ray of the beam passes, the beam passes.  The beam's
// walk therefore reaches every triangle whose own box any of its rays passes, and there each ray is tested by itself (accepts(): the
// full hit definition).  No result depends on the beam; it only decides which triangles are looked at.
struct Beam {
    float o[3], inv_lo[3], inv_hi[3];
    float tmin, tmax;
};
SAH_DEV bool beam_slab(const Beam& b, const float lo[3], const float hi[3]) {
    float tn = b.tmin, tf = b.tmax;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float a0 = lo[c] - b.o[c], a1 = hi[c] - b.o[c];
        const float p0 = a0 * b.inv_lo[c], p1 = a0 * b.inv_hi[c], p2 = a1 * b.inv_lo[c], p3 = a1 * b.inv_hi[c];
        tn = __builtin_fmaxf(tn, __builtin_fminf(__builtin_fminf(p0, p1), __builtin_fminf(p2, p3)));
        tf = __builtin_fminf(tf, __builtin_fmaxf(__builtin_fmaxf(p0, p1), __builtin_fmaxf(p2, p3)));
    }
    return tn <= tf;
}
SAH_DEV uint32_t children_hit_beam(const RtBvh& bvh, const Trav& tv, const Beam& b, uint32_t level, uint32_t node) {
    const float4* p = reinterpret_cast<const float4*>(bvh.nodes + tv.off[level - 1u] + node);
    float q[6][4];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const float4 v = p[k];
        q[k][0] = v.x; q[k][1] = v.y; q[k][2] = v.z; q[k][3] = v.w;
    }
    uint32_t m = 0;
#pragma unroll
    for (uint32_t k = 0; k < kRtFanout; k++) {
        const float lo[3] = {q[0][k], q[1][k], q[2][k]}, hi[3] = {q[3][k], q[4][k], q[5][k]};
        if (beam_slab(b, lo, hi)) m |= 1u << k;
    }
    return m;
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

// closest accepted candidate (RAY_FLAG_NONE: CUTOUT candidates run the any-hit stage); ties: smallest (primitive, triangle)
struct Closest {
    bool hit;
    Hit h;
    uint32_t tri;  // index into bvh.tris
};
// (CULL_NON_OPAQUE / CULL_BACK: the flags of the bounce rays of the GI hit stage, gltf_basic_pbr.slang:498-506)
template <bool CULL_NON_OPAQUE = false, bool CULL_BACK = false>
SAH_DEV Closest closest_hit(const RtBvh& bvh, const RtScene& sc, const Trav& tv, Ray r) {
    Closest best;
    best.hit = false;
    best.tri = 0;
    best.h = {0.f, 0.f, 0.f, false};
    if (!r.finite || bvh.num_tris == 0) return best;
    uint32_t best_prim = 0xffffffffu, best_tri = 0xffffffffu;
    // boxes are culled against the best t so far (a hair beyond it: the entry into a padded box of a candidate at t <= limit is computed
    // with its own rounding, and visiting a box too many changes nothing); the triangle test keeps the ray's own tmax
    const float tmax0 = r.tmax;
    const uint32_t top = bvh.num_levels - 1u;
    uint32_t level = top, node = 0;
    unsigned long long pending = 0;
    bool alive = true;
    if (top == 0) {
        const RtNodeGroup& n = bvh.nodes[0];
        const float lo[3] = {n.lo[0][0], n.lo[1][0], n.lo[2][0]}, hi[3] = {n.hi[0][0], n.hi[1][0], n.hi[2][0]};
        alive = slab(r, lo, hi);
    }
    while (alive) {
        while (alive && level != 0) {
            uint32_t nearest;
            const uint32_t m = children_hit(bvh, tv, r, level, node, &nearest);
            alive = trav_next(top, level, node, pending, m, nearest);
        }
        if (alive) {
            const RtTriangle tr = load_triangle(bvh.tris, node);
            Ray full = r;
            full.tmax = tmax0;
            Hit h;
            if (woop(full, tr, h)) {
                const bool better = !best.hit || h.t < best.h.t ||
                                    (h.t == best.h.t && (tr.primitive < best_prim || (tr.primitive == best_prim && tr.triangle < best_tri)));
                const bool culled = (CULL_NON_OPAQUE && (tr.flags & 1u)) || (CULL_BACK && !h.front);
                if (better && !culled && (!(tr.flags & 1u) || cutout_accepts(sc, tr, h))) {
                    best.hit = true;
                    best.h = h;
                    best.tri = node;
                    best_prim = tr.primitive;
                    best_tri = tr.triangle;
                    r.tmax = __builtin_fminf(h.t * 1.000244140625f, tmax0);  // (never above the ray's own, finite, tmax)
                }
            }
            alive = trav_next(top, level, node, pending, 0u);
        }
    }
    return best;
}

// unpackUnorm4x8ToHalf / packUnorm4x8 (gltf_basic_pbr.slang:257-276) of one channel
SAH_DEV Hn unpack_channel(uint32_t packed, int c) { return Hn((float)((packed >> (8 * c)) & 0xffu)) / Hn::lit(255.0f); }

struct GiPayload {
    F3 irradiance;
    Fn ray_distance;
};

// TraceRay(rtas, RAY_FLAG_NONE, 0xFF, RAY_TYPE_GI, ...) with remaining_bounces == 0: closest-hit stage of gltf_basic_pbr.slang:345-470 or the
// GI miss stage of sky_unified.slang:227-230.  (dx, dy) = DispatchRaysIndex().xy
// MAXB: how deep the bounce branch of the hit stage (gltf_basic_pbr.slang:481-517) is instantiated; `remaining` = payload.remaining_bounces
// (<= MAXB).  The reference's generators trace with 0 (they never forward their num_bounces push constant: the branch is dead there); a
// context may ask for up to 2 (sah_rt_set_bounces).  BOUNCE: this ray IS a bounce ray (RAY_FLAG_CULL_NON_OPAQUE | CULL_BACK_FACING_TRIANGLES).
template <int MAXB = 0, bool BOUNCE = false>
SAH_DEV GiPayload trace_gi(const RtBvh& bvh, const RtScene& sc, const Trav& tv, const GiArgs& g, const Ray& r, uint32_t dx, uint32_t dy, uint32_t remaining = 0u) {
    GiPayload pay;
    pay.irradiance = F3(Fn(0.f));
    pay.ray_distance = Fn(0.f);
    const Closest c = closest_hit<BOUNCE, BOUNCE>(bvh, sc, tv, r);
    if (!c.hit) {
        // (a ray with a non-finite component reaches no stage at all: the payload stays zero)
        if (r.finite) pay.irradiance = sky_color(g.sky, F3{Fn(r.d[0]), Fn(r.d[1]), Fn(r.d[2])});
        return pay;
    }
    const RtTriangle tr = bvh.tris[c.tri];
    const sah_primitive& prim = sc.primitives[tr.primitive];
    const uint32_t* idx = sc.indices + prim.first_index + 3u * tr.triangle;
    const int64_t i0 = (int64_t)prim.vertex_offset + idx[0], i1 = (int64_t)prim.vertex_offset + idx[1], i2 = (int64_t)prim.vertex_offset + idx[2];
    const sah_vertex_data &a = sc.vertex_data[i0], &b = sc.vertex_data[i1], &cc = sc.vertex_data[i2];
    const Fn b1 = Fn(c.h.b1), b2 = Fn(c.h.b2), b0 = (Fn(1.0f) - b1) - b2;
    // interpolate_vertex (:278-287): normal, texcoord in fp32; colour through half and the truncating pack
    F3 normal_f;
    normal_f.x = b0 * Fn(a.normal[0]) + b1 * Fn(b.normal[0]) + b2 * Fn(cc.normal[0]);
    normal_f.y = b0 * Fn(a.normal[1]) + b1 * Fn(b.normal[1]) + b2 * Fn(cc.normal[1]);
    normal_f.z = b0 * Fn(a.normal[2]) + b1 * Fn(b.normal[2]) + b2 * Fn(cc.normal[2]);
    float uv[2];
    for (int k = 0; k < 2; k++) uv[k] = (b0 * Fn(a.texcoord[k]) + b1 * Fn(b.texcoord[k]) + b2 * Fn(cc.texcoord[k])).v;
    Hn colour[4];
    for (int k = 0; k < 4; k++) {
        const Fn v = b0 * Fn(tof(unpack_channel(a.color, k))) + b1 * Fn(tof(unpack_channel(b.color, k))) + b2 * Fn(tof(unpack_channel(cc.color, k)));
        const uint32_t byte = to_uint_sat(tof(Hn(v.v) * Hn::lit(255.0f))) & 0xffu;
        colour[k] = Hn((float)byte) / Hn::lit(255.0f);
    }
    // position: model * (b.x p0 + b.y p1 + b.z p2, 1)
    const float *p0 = sc.positions + 3 * i0, *p1 = sc.positions + 3 * i1, *p2 = sc.positions + 3 * i2;
    float mp[3];
    for (int k = 0; k < 3; k++) mp[k] = (b0 * Fn(p0[k]) + b1 * Fn(p1[k]) + b2 * Fn(p2[k])).v;
    float loc[3];
    for (int k = 0; k < 3; k++) loc[k] = ((prim.model[k] * mp[0] + prim.model[4 + k] * mp[1]) + prim.model[8 + k] * mp[2]) + prim.model[12 + k] * 1.0f;
    const sah_material& m = sc.materials[prim.material];
    // the three material slots at level 0 (SampleLevel), or the material's constant texels
    float base_t[4], data_t[4], emis_t[4];
    for (int k = 0; k < 4; k++) {
        base_t[k] = m.base_color_texel[k];
        data_t[k] = m.data_texel[k];
        emis_t[k] = m.emission_texel[k];
    }
    if (sc.material_textures) {
        const sah_material_textures mt = sc.material_textures[prim.material];
        if (mt.base_color != SAH_TEXTURE_NONE) sample_texture_lod(sc.luts, sc.textures[mt.base_color], uv, 0.0f, 0.0f, base_t);
        if (mt.data != SAH_TEXTURE_NONE) sample_texture_lod(sc.luts, sc.textures[mt.data], uv, 0.0f, 0.0f, data_t);
        if (mt.emission != SAH_TEXTURE_NONE) sample_texture_lod(sc.luts, sc.textures[mt.emission], uv, 0.0f, 0.0f, emis_t);
    }
    Surface<Hn> s;
    s.base_color = {Hn(((Fn(base_t[0]) * Fn(m.base_color_tint[0])) * Fn(tof(colour[0]))).v), Hn(((Fn(base_t[1]) * Fn(m.base_color_tint[1])) * Fn(tof(colour[1]))).v),
                    Hn(((Fn(base_t[2]) * Fn(m.base_color_tint[2])) * Fn(tof(colour[2]))).v)};
    s.normal = {Hn(normal_f.x.v), Hn(normal_f.y.v), Hn(normal_f.z.v)};
    s.roughness = Hn(data_t[1]) * Hn(m.roughness_factor);  // tinted_data = data_sample * half4(0, roughness_factor, metalness_factor, 0)
    s.metalness = Hn(data_t[2]) * Hn(m.metalness_factor);
    const H3 emission = {Hn(emis_t[0]) * Hn(m.emission_factor[0]), Hn(emis_t[1]) * Hn(m.emission_factor[1]), Hn(emis_t[2]) * Hn(m.emission_factor[2])};
    const H3 Lh = normalize(H3{Hn(-g.sun_dir[0]), Hn(-g.sun_dir[1]), Hn(-g.sun_dir[2])});  // normalize((half3)-direction)
    const H3 brdf_result = Fd(s, Lh, s.normal);
    const Hn ndotl = nclamp(dot(Lh, s.normal), Hn::lit(0.f), Hn::lit(1.f));
    Hn shadow = Hn::lit(0.f);
    if (tof(ndotl) > 0.f && c.h.front) {  // (a back-face hit is black whatever its shadow ray says: none is traced)
        const F3 noise = load_noise(g.noise, sc.luts, dx % 128u, dy % 128u);
        const F3 dir = normalize(to_f(Lh) + noise * Fn(g.tan_size));
        const float d[3] = {dir.x.v, dir.y.v, dir.z.v};
        const Ray sr = make_ray(loc, d, 0.05f, 100000.0f);
        if (!any_hit<true, true>(bvh, sc, tv, sr)) shadow = Hn::lit(1.f);
    }
    // payload.irradiance = brdf_result * sun_light.color.rgb * ndotl * shadow (half3 * float3 -> float3, then * half, * half); += emission
    const F3 sun = {Fn(g.sun_color[0]), Fn(g.sun_color[1]), Fn(g.sun_color[2])};
    F3 irr = to_f(brdf_result) * sun * Fn(tof(ndotl)) * Fn(tof(shadow));
    irr = irr + to_f(emission);
    pay.irradiance = irr;
    pay.ray_distance = Fn(c.h.t);
    if constexpr (MAXB > 0) {
        // (a back-face hit is zeroed below whatever the bounce brings: its bounce ray is not traced)
        if (remaining > 0u && c.h.front) {
            F3 dir = load_noise(g.noise, sc.luts, dx % 128u, dy % 128u);  // the launch index's noise texel, at every depth
            if (dot(to_f(s.normal), dir).v < 0.f) dir = dir * Fn(-1.0f);
            const float d[3] = {dir.x.v, dir.y.v, dir.z.v};
            const GiPayload next = trace_gi<MAXB - 1, true>(bvh, sc, tv, g, make_ray(loc, d, 0.05f, 100000.0f), dx, dy, remaining - 1u);
            const H3 dh = {Hn(dir.x.v), Hn(dir.y.v), Hn(dir.z.v)};
            const H3 bounce_brdf = brdf_sl(s, dh, s.normal);  // brdf(surface, bounce_ray.Direction, surface.normal) = Fd + Fr (the sun term above is Fd alone)
            const Hn bounce_ndotl = Hn(nclamp(dot(dir, to_f(s.normal)), Fn(0.f), Fn(1.f)).v);
            const F3 radiance = to_f(bounce_brdf * bounce_ndotl) * next.irradiance;  // half * half3, then * float3
            const bool finite = finite3r(radiance.x.v, radiance.y.v, radiance.z.v);  // !any(isnan) && !any(isinf)
            if (finite) pay.irradiance = pay.irradiance + radiance;
        }
    }
    if (!c.h.front) {  // HIT_KIND_TRIANGLE_BACK_FACE
        pay.ray_distance = pay.ray_distance * Fn(-1.0f);
        pay.irradiance = F3(Fn(0.f));
    }
    return pay;
}

// Rays that leave neighbouring pixels in unrelated directions (one noise texel each: RTAO, the RTGI rays) would make the lanes of a wave
// walk different parts of the hierarchy.  The 256 rays of a workgroup are therefore re-dealt before the walk: sorted by the octant and
// the dominant axis of their direction (a counting sort through LDS; which lane ends up with which ray of a bin is not defined and
// does not matter — a ray's result goes to the ray's own pixel), so that a wave holds rays that travel the same way.
constexpr uint32_t kDealBins = 25u;  // 8 octants x 3 dominant axes, and one for the threads without a ray
struct DealLds {
    uint32_t count[kDealBins], base[kDealBins];
    float ray[6][256];
    uint16_t src[256];
};
// every thread of the 256-thread workgroup calls these two, in this order, with barriers of its own in between (deal_clear before
// trav_init, whose barrier publishes the zeros)
SAH_DEV void deal_clear(DealLds& L) {
    if (threadIdx.x < kDealBins) L.count[threadIdx.x] = 0u;
}
// in: this thread's ray (has_ray false: none).  out: the ray this thread walks and the thread that made it; false: nothing to walk
SAH_DEV bool deal_rays(DealLds& L, bool has_ray, float (&o)[3], float (&d)[3], uint32_t& src) {
    uint32_t key = kDealBins - 1u;
    if (has_ray) {
        const float ax = __builtin_fabsf(d[0]), ay = __builtin_fabsf(d[1]), az = __builtin_fabsf(d[2]);
        const uint32_t major = az > __builtin_fmaxf(ax, ay) ? 2u : (ay > ax ? 1u : 0u);
        key = ((d[0] < 0.f ? 1u : 0u) | (d[1] < 0.f ? 2u : 0u) | (d[2] < 0.f ? 4u : 0u)) * 3u + major;
    }
    const uint32_t rank = atomicAdd(&L.count[key], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (uint32_t k = 0; k < kDealBins; k++) {
            L.base[k] = sum;
            sum += L.count[k];
        }
    }
    __syncthreads();
    const uint32_t slot = L.base[key] + rank;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        L.ray[c][slot] = o[c];
        L.ray[3 + c][slot] = d[c];
    }
    L.src[slot] = (uint16_t)threadIdx.x;
    __syncthreads();
    if (threadIdx.x >= L.base[kDealBins - 1u]) return false;  // the slots behind the last ray
#pragma unroll
    for (int c = 0; c < 3; c++) {
        o[c] = L.ray[c][threadIdx.x];
        d[c] = L.ray[3 + c][threadIdx.x];
    }
    src = L.src[threadIdx.x];
    return true;
}
// thread index -> pixel: a wave is an 8 x 8 pixel square, a workgroup 16 x 16; the grid's first row of tiles holds row `row_begin`
SAH_DEV void tile_pixel(uint32_t t, uint32_t row_begin, uint32_t& x, uint32_t& y) {
    x = blockIdx.x * 16u + (t & 7u) + ((t >> 3) & 8u);
    y = (row_begin & ~15u) + blockIdx.y * 16u + ((t >> 3) & 7u) + ((t >> 4) & 8u);
}

// probe_tracing.rt.slang:39-106: thread (tx, ty, probe) of dispatch_rays({20, 20, num_probes}); a workgroup takes 256 consecutive rays
// of the dispatch and re-deals them by direction (a probe's 400 rays cover the sphere)
template <int MAXB>
__global__ __launch_bounds__(256) void k_probe_trace(const ProbeTraceArgs a, const RtBvh bvh, const RtScene sc) {
    __shared__ uint32_t s_levels[kRtMaxLevels];
    __shared__ DealLds s_deal;
    deal_clear(s_deal);
    const Trav tv = trav_init(bvh, s_levels);
    const CacheArgs& c = a.cache;
    auto probe_ray = [&](uint32_t t, uint32_t& probe, uint32_t& tx, uint32_t& ty, uint32_t& cascade, F3& origin) {
        probe = t / 400u;
        tx = (t % 400u) % 20u;
        ty = (t % 400u) / 20u;
        const uint32_t px = a.probes[3u * probe], py = a.probes[3u * probe + 1u], pz = a.probes[3u * probe + 2u];
        cascade = py / 8u;
        if (cascade < 4u) {
            const F3 local = {Fn((float)px), Fn((float)(py % 8u)), Fn((float)pz)};
            origin = F3{Fn(c.cascade_min[cascade][0]), Fn(c.cascade_min[cascade][1]), Fn(c.cascade_min[cascade][2])} + local * Fn(c.spacing[cascade]);
        }
    };
    uint32_t t = blockIdx.x * 256u + threadIdx.x, probe = 0, tx = 0, ty = 0, cascade = 4;
    F3 origin = F3(Fn(0.f));
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    bool has_ray = false;
    if (t < 400u * a.num_probes) {
        probe_ray(t, probe, tx, ty, cascade, origin);
        has_ray = cascade < 4u;
        if (has_ray) {
            const F3 dir = octahedral_direction(normalized_octahedral_coordinates(tx, ty, 20u, 20u));
            d[0] = dir.x.v; d[1] = dir.y.v; d[2] = dir.z.v;
            o[0] = origin.x.v; o[1] = origin.y.v; o[2] = origin.z.v;
        } else {  // (cascades[cascade_index] of a 4-entry array: anything else is out of bounds in the shader; the ABI writes zeros)
            *reinterpret_cast<uint2*>(const_cast<uint8_t*>(a.out.ptr) + (size_t)probe * a.out.slice_pitch + (size_t)ty * a.out.row_pitch + (size_t)tx * 8) = make_uint2(0u, 0u);
        }
    }
    uint32_t src;
    if (!deal_rays(s_deal, has_ray, o, d, src)) return;
    t = blockIdx.x * 256u + src;
    probe_ray(t, probe, tx, ty, cascade, origin);
    const F3 dir = {Fn(d[0]), Fn(d[1]), Fn(d[2])};
    Fn ray_distance = Fn(8192.f);
    if (cascade < 3u) ray_distance = Fn(c.spacing[cascade + 1u]) * Fn(4.f);
    const Ray r = make_ray(o, d, 0.05f, ray_distance.v);
    GiPayload pay = trace_gi<MAXB>(bvh, sc, tv, a.gi, r, tx, ty, MAXB ? a.gi.num_bounces : 0u);
    if (pay.ray_distance.v == 0.f) {
        if (cascade + 1u < 4u) pay.irradiance = sample_cascade(c, origin + dir * ray_distance, dir, cascade + 1u);
        else pay.irradiance = pay.irradiance * Fn(10.f);
        pay.ray_distance = ray_distance;
    } else if (pay.ray_distance.v < 0.f) {
        pay.irradiance = F3(Fn(0.f));
    }
    const Hn e = Hn::lit(0.0031415927f);
    const Hn out[4] = {Hn(pay.irradiance.x.v) * e, Hn(pay.irradiance.y.v) * e, Hn(pay.irradiance.z.v) * e, Hn(pay.ray_distance.v)};
    uint2 w;
    w.x = (uint32_t)__builtin_bit_cast(uint16_t, out[0].v) | ((uint32_t)__builtin_bit_cast(uint16_t, out[1].v) << 16);
    w.y = (uint32_t)__builtin_bit_cast(uint16_t, out[2].v) | ((uint32_t)__builtin_bit_cast(uint16_t, out[3].v) << 16);
    *reinterpret_cast<uint2*>(const_cast<uint8_t*>(a.out.ptr) + (size_t)probe * a.out.slice_pitch + (size_t)ty * a.out.row_pitch + (size_t)tx * 8) = w;
}

// rtgi.rt.slang:56-110
template <int MAXB>
__global__ __launch_bounds__(256) void k_rtgi_trace(const RtgiTraceArgs a, const RtBvh bvh, const RtScene sc) {
    __shared__ uint32_t s_levels[kRtMaxLevels];
    __shared__ DealLds s_deal;
    deal_clear(s_deal);
    const Trav tv = trav_init(bvh, s_levels);
    uint32_t x, y;
    tile_pixel(threadIdx.x, a.row_begin, x, y);
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    bool has_ray = x < a.width && y >= a.row_begin && y < a.row_end && (float)x < a.res[0] && (float)y < a.res[1];  // any(thread_id >= render_resolution): uint against float
    if (has_ray) {
        const float depth = *reinterpret_cast<const float*>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x * 4);
        has_ray = depth != 0.f;
        if (has_ray) {
            const uint2 nw = *reinterpret_cast<const uint2*>(a.normals.ptr + (size_t)y * a.normals.pitch + (size_t)x * 8);
            const H3 normal = {Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(nw.x & 0xffffu))), Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(nw.x >> 16))),
                               Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(nw.y & 0xffffu)))};  // not normalised (quirk)
            world_position(a.inv_proj, a.inv_view, a.res, x, y, depth, o);
            F3 dir = load_noise(a.gi.noise, sc.luts, x % 128u, y % 128u);
            if (dot(to_f(normal), dir).v < 0.f) dir = dir * Fn(-1.0f);
            d[0] = dir.x.v; d[1] = dir.y.v; d[2] = dir.z.v;
        }
    }
    uint32_t src;
    if (!deal_rays(s_deal, has_ray, o, d, src)) return;
    tile_pixel(src, a.row_begin, x, y);  // DispatchRaysIndex of the ray this thread walks now
    const Ray r = make_ray(o, d, 0.01f, 100000.0f);
    GiPayload pay = trace_gi<MAXB>(bvh, sc, tv, a.gi, r, x, y, MAXB ? a.gi.num_bounces : 0u);
    if (any_nan(pay.irradiance)) pay.irradiance = F3(Fn(0.f));
    const Fn e = Fn(0.0031415927f);
    auto store = [](const PlaneArg& p, uint32_t px, uint32_t py, float c0, float c1, float c2, float c3) {
        uint2 w;
        w.x = (uint32_t)f2h(c0) | ((uint32_t)f2h(c1) << 16);
        w.y = (uint32_t)f2h(c2) | ((uint32_t)f2h(c3) << 16);
        *reinterpret_cast<uint2*>(const_cast<uint8_t*>(p.ptr) + (size_t)py * p.pitch + (size_t)px * 8) = w;
    };
    store(a.ray_buffer, x, y, d[0], d[1], d[2], pay.ray_distance.v);
    store(a.ray_irradiance, x, y, (pay.irradiance.x * e).v, (pay.irradiance.y * e).v, (pay.irradiance.z * e).v, 0.f);
}

// rtao.comp.slang:54-102
__global__ __launch_bounds__(256) void k_rtao(const RtaoArgs a, const RtBvh bvh, const RtScene sc) {
    __shared__ uint32_t s_levels[kRtMaxLevels];
    __shared__ DealLds s_deal;
    deal_clear(s_deal);
    const Trav tv = trav_init(bvh, s_levels);
    uint32_t x, y;
    tile_pixel(threadIdx.x, a.row_begin, x, y);
    const bool inside = x < a.width && y >= a.row_begin && y < a.row_end;
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    if (inside) {
        const float depth = *reinterpret_cast<const float*>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x * 4);
        world_position(a.inv_proj, a.inv_view, a.res, x, y, depth, o);
        const H3 normal = load_normal_h(a.normals, x, y);
        F3 noise = load_noise(a.noise, sc.luts, x % a.noise_w, y % a.noise_h);
        if (dot(noise, to_f(normal)).v < 0.0f) noise = noise * Fn(-1.0f);
        d[0] = noise.x.v; d[1] = noise.y.v; d[2] = noise.z.v;
    }
    uint32_t src;
    if (!deal_rays(s_deal, inside, o, d, src)) return;
    tile_pixel(src, a.row_begin, x, y);
    const Ray r = make_ray(o, d, 0.01f, a.max_distance);
    const bool hit = any_hit<true>(bvh, sc, tv, r);
    // every one of the spp rays is this ray (the shader reads the same noise texel for each): ao = spp - spp or spp, exact for spp <= 4096
    const float spp = (float)a.samples;
    const float ao = (hit ? spp - spp : spp) / spp;
    *reinterpret_cast<float*>(const_cast<uint8_t*>(a.out.ptr) + (size_t)y * a.out.pitch + (size_t)x * 4) = ao;
}

// load_noise() of the 128 x 128 texels the sun's shadow samples index, once per call
__global__ __launch_bounds__(256) void k_noise_dirs(const PlaneArg noise, const float* luts, float* out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, x = t % 128u, y = t / 128u;
    const F3 v = load_noise(noise, luts, x, y);
    reinterpret_cast<float4*>(out)[t] = make_float4(v.x.v, v.y.v, v.z.v, 0.f);
}

// directional_light.rt.slang:91-125.  Only the pixels that face the light trace rays (half of a typical frame), each of them
// num_shadow_samples of them, and an occluded ray ends early: one pixel per lane leaves most lanes idle most of the time.  The traced
// pixels of a workgroup are therefore compacted into LDS, and a pixel's unoccluded rays are counted with an LDS atomic: `shadow` is a
// sum of 1.0s, an exact integer whatever the order.  Sample 0 of every traced pixel is walked first, one pixel per lane.  Then
//   * a pixel whose sample 0 was occluded deals its other samples to the lanes as (pixel, sample) pairs, sample-major — a wave holds
//     neighbouring pixels with the same sample's noise offset — and each is first tried against the triangle that last occluded a ray of
//     its pixel (the occluder cache below);
//   * a pixel whose sample 0 was not — most of them have no occluder at all, and per-ray walks would cross the whole scene once per
//     sample — walks its other samples as ONE beam (above): what the beam meets is tested ray by ray, and the walk ends early when no
//     ray of the pixel is left unoccluded.
__global__ __launch_bounds__(256) void k_sun_shadow_mask(const ShadowMaskArgs a, const RtBvh bvh, const RtScene sc) {
    __shared__ uint32_t s_levels[kRtMaxLevels];
    __shared__ float s_origin[3][256];
    __shared__ uint16_t s_pixel[256], s_list[2][256];
    __shared__ uint32_t s_unoccluded[256], s_wave_count[4], s_list_n[2];
    // the triangle that last occluded a ray of the pixel: the next sample's ray leaves the same point in almost the same direction and is
    // tried against it before anything else.  Whether a ray is occluded does not depend on which occluder is found, so neither a
    // stale entry nor the order in which the lanes get here can change a result
    __shared__ uint32_t s_occluder[256];
    constexpr uint32_t kMaskChunk = 2048u;
    __shared__ uint16_t s_miss[kMaskChunk], s_off[64];
    __shared__ uint32_t s_miss_n;
    if (threadIdx.x < 2u) s_list_n[threadIdx.x] = 0u;
    const Trav tv = trav_init(bvh, s_levels);
    uint32_t x, y;
    tile_pixel(threadIdx.x, a.row_begin, x, y);
    const F3 L = {Fn(a.L[0]), Fn(a.L[1]), Fn(a.L[2])};
    bool traced = false;
    float o[3] = {0.f, 0.f, 0.f};
    if (x < a.width && y >= a.row_begin && y < a.row_end) {
        const float depth = *reinterpret_cast<const float*>(a.depth.ptr + (size_t)y * a.depth.pitch + (size_t)x * 4);
        const H3 normal = load_normal_h(a.normals, x, y);
        const Hn ndotl = Hn(nclamp(dot(L, to_f(normal)), Fn(0.f), Fn(1.f)).v);
        traced = !(depth == 0.0f || !(tof(ndotl) > 0.0f));
        if (traced) world_position(a.inv_proj, a.inv_view, a.res, x, y, depth, o);
        else *reinterpret_cast<float*>(const_cast<uint8_t*>(a.out.ptr) + (size_t)y * a.out.pitch + (size_t)x * 4) = 1.0f;
    }
    // compaction, pixel order kept: ballot + prefix inside the wave, wave totals through LDS
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long ballot = __ballot(traced);
    if (lane == 0) s_wave_count[wave] = (uint32_t)__builtin_popcountll(ballot);
    __syncthreads();
    uint32_t slot = (uint32_t)__builtin_popcountll(ballot & ((1ull << lane) - 1ull)), num_traced = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t c = s_wave_count[w];
        slot += w < wave ? c : 0u;
        num_traced += c;
    }
    if (traced) {
#pragma unroll
        for (int c = 0; c < 3; c++) s_origin[c][slot] = o[c];
        s_pixel[slot] = (uint16_t)threadIdx.x;
        s_unoccluded[slot] = 0u;
        s_occluder[slot] = 0xffffffffu;
    }
    __syncthreads();
    // the shader's loop `for (i = 0; i < num_shadow_samples; i++)` with a float bound in [0, 4096] (host check) runs ceil(bound) times
    const uint32_t num_samples = a.num_samples > 0.f ? (uint32_t)__builtin_ceilf(a.num_samples) : 0u;
    // direction of sample i of the pixel (px, py); the noise offsets of the first 64 samples are tabulated (one divide each)
    auto sample_offset = [](uint32_t i, uint32_t& offx, uint32_t& offy) {
        const Fn phi = Fn(1.618033988749895f);
        const Fn q = Fn((float)i) / phi;
        const Fn r0x = Fn(2.0f) + q, r0y = Fn(3.0f) + q;
        const Fn fx = r0x - Fn(__builtin_floorf(r0x.v)), fy = r0y - Fn(__builtin_floorf(r0y.v));
        offx = (uint32_t)__builtin_rintf((fx * Fn(128.0f)).v);  // in [0, 128]: uint(float(px) + off) % 128 is (px + off) % 128, exactly
        offy = (uint32_t)__builtin_rintf((fy * Fn(128.0f)).v);
    };
    if (threadIdx.x < 64u) {
        uint32_t ox, oy;
        sample_offset(threadIdx.x, ox, oy);
        s_off[threadIdx.x] = (uint16_t)(ox | (oy << 8));
    }
    __syncthreads();
    auto sample_dir = [&](uint32_t px, uint32_t py, uint32_t i, float (&d)[3]) {
        uint32_t offx, offy;
        if (i < 64u) {
            const uint32_t w = s_off[i];
            offx = w & 255u;
            offy = w >> 8;
        } else {
            sample_offset(i, offx, offy);
        }
        const uint32_t nx = (px + offx) % 128u, ny = (py + offy) % 128u;
        const float4 nv = reinterpret_cast<const float4*>(a.noise_dirs)[ny * 128u + nx];
        const F3 noise = {Fn(nv.x), Fn(nv.y), Fn(nv.z)};
        const F3 dir = normalize(L + noise * Fn(a.tan_size));
        d[0] = dir.x.v; d[1] = dir.y.v; d[2] = dir.z.v;
    };
    const bool beams = num_samples >= 2u && num_samples <= 33u;  // samples 1 .. n - 1 of a pixel in one 32-bit mask
    // ---- sample 0, one traced pixel per lane
    if (threadIdx.x < num_traced && num_samples != 0u) {
        const uint32_t p = threadIdx.x;
        uint32_t px, py;
        tile_pixel(s_pixel[p], a.row_begin, px, py);
        const float po[3] = {s_origin[0][p], s_origin[1][p], s_origin[2][p]};
        float d[3];
        sample_dir(px, py, 0u, d);
        const Ray r = make_ray(po, d, 0.01f, 100000.0f);
        uint32_t occluder;
        const bool hit = any_hit<false>(bvh, sc, tv, r, &occluder);
        if (hit) s_occluder[p] = occluder;
        else s_unoccluded[p] = 1u;
        const uint32_t cls = !hit && beams ? 1u : 0u;
        s_list[cls][atomicAdd(&s_list_n[cls], 1u)] = (uint16_t)p;
    }
    __syncthreads();
    // ---- pixels whose sample 0 was unoccluded: the other samples as one beam
    const uint32_t top = bvh.num_levels - 1u;
    for (uint32_t qi = threadIdx.x; qi < s_list_n[1]; qi += 256u) {
        const uint32_t p = s_list[1][qi];
        uint32_t px, py;
        tile_pixel(s_pixel[p], a.row_begin, px, py);
        const float po[3] = {s_origin[0][p], s_origin[1][p], s_origin[2][p]};
        Beam b;
        bool ok = finite3(po) && bvh.num_tris != 0u;
        for (int c = 0; c < 3; c++) {
            b.o[c] = po[c];
            b.inv_lo[c] = __builtin_inff();
            b.inv_hi[c] = -__builtin_inff();
        }
        b.tmin = 0.01f;
        b.tmax = 100000.0f;
        for (uint32_t i = 1; i < num_samples; i++) {
            float d[3];
            sample_dir(px, py, i, d);
            for (int c = 0; c < 3; c++) {
                const float inv = 1.0f / d[c];  // make_ray's
                ok = ok && __builtin_fabsf(inv) < __builtin_inff() && inv != 0.0f && d[c] == d[c];
                b.inv_lo[c] = __builtin_fminf(b.inv_lo[c], inv);
                b.inv_hi[c] = __builtin_fmaxf(b.inv_hi[c], inv);
            }
        }
        for (int c = 0; c < 3; c++) ok = ok && (b.inv_lo[c] > 0.0f) == (b.inv_hi[c] > 0.0f);
        uint32_t left = 0xffffffffu >> (33u - num_samples);  // bit i - 1: sample i has met no occluder yet (2 <= num_samples <= 33)
        if (ok) {
            uint32_t level = top, node = 0;
            unsigned long long pending = 0;
            bool walking = true;
            if (top == 0u) {
                const RtNodeGroup& g = bvh.nodes[0];
                const float lo[3] = {g.lo[0][0], g.lo[1][0], g.lo[2][0]}, hi[3] = {g.hi[0][0], g.hi[1][0], g.hi[2][0]};
                walking = beam_slab(b, lo, hi);
            }
            while (walking && left != 0u) {
                while (walking && level != 0u) walking = trav_next(top, level, node, pending, children_hit_beam(bvh, tv, b, level, node));
                if (walking) {
                    for (uint32_t m = left; m != 0u; m &= m - 1u) {
                        const uint32_t i = (uint32_t)__builtin_ctz(m) + 1u;
                        float d[3];
                        sample_dir(px, py, i, d);
                        const Ray r = make_ray(po, d, 0.01f, 100000.0f);
                        if (accepts<false>(bvh, sc, r, node)) left &= ~(1u << (i - 1u));
                    }
                    walking = trav_next(top, level, node, pending, 0u);
                }
            }
        } else {  // (a direction with a zero or a sign change between the samples, a non-finite origin: ray by ray)
            for (uint32_t i = 1; i < num_samples; i++) {
                float d[3];
                sample_dir(px, py, i, d);
                const Ray r = make_ray(po, d, 0.01f, 100000.0f);
                if (any_hit<false>(bvh, sc, tv, r)) left &= ~(1u << (i - 1u));
            }
        }
        atomicAdd(&s_unoccluded[p], (uint32_t)__builtin_popcount(left));
    }
    // ---- the others: (pixel, sample) pairs, sample-major, in chunks of 2048.  Pass 1 tries every pair against its pixel's cached
    // occluder; the few that miss are listed in LDS and walked in pass 2, densely packed — walked where they stand, one missing lane
    // would take its whole wave through a walk (at a 2 % miss rate three waves in four).
    const uint32_t n0 = s_list_n[0], items = num_samples > 1u ? n0 * (num_samples - 1u) : 0u;
    auto pair_ray = [&](uint32_t item, uint32_t& p) {
        const uint32_t i = item / n0 + 1u;
        p = s_list[0][item % n0];
        uint32_t px, py;
        tile_pixel(s_pixel[p], a.row_begin, px, py);
        const float po[3] = {s_origin[0][p], s_origin[1][p], s_origin[2][p]};
        float d[3];
        sample_dir(px, py, i, d);
        return make_ray(po, d, 0.01f, 100000.0f);
    };
    for (uint32_t chunk = 0; chunk < items; chunk += kMaskChunk) {
        if (threadIdx.x == 0) s_miss_n = 0u;
        __syncthreads();
        for (uint32_t item = chunk + threadIdx.x; item < min(items, chunk + kMaskChunk); item += 256u) {
            uint32_t p;
            const Ray r = pair_ray(item, p);
            const uint32_t last = s_occluder[p];
            if (!(last != 0xffffffffu && accepts<false>(bvh, sc, r, last))) s_miss[atomicAdd(&s_miss_n, 1u)] = (uint16_t)(item - chunk);
        }
        __syncthreads();
        for (uint32_t m = threadIdx.x; m < s_miss_n; m += 256u) {
            uint32_t p, occluder;
            const Ray r = pair_ray(chunk + s_miss[m], p);
            if (any_hit<false>(bvh, sc, tv, r, &occluder)) s_occluder[p] = occluder;
            else atomicAdd(&s_unoccluded[p], 1u);
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x < num_traced) {
        uint32_t px, py;
        tile_pixel(s_pixel[threadIdx.x], a.row_begin, px, py);
        const Fn shadow = Fn((float)s_unoccluded[threadIdx.x]);
        *reinterpret_cast<float*>(const_cast<uint8_t*>(a.out.ptr) + (size_t)py * a.out.pitch + (size_t)px * 4) = (shadow / Fn(a.num_samples)).v;
    }
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
hipError_t launch_rt_nodes(const RtTriangle* unsorted, const unsigned long long* keys, RtTriangle* sorted, RtNodeGroup* nodes, const RtBvh& bvh, hipStream_t s) {
    if (bvh.num_tris == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rt_refine, dim3((bvh.num_tris + kRefineWindow - 1u) / kRefineWindow), dim3(768), 0, s, unsorted, const_cast<unsigned long long*>(keys), bvh.num_tris);
    hipLaunchKernelGGL(k_rt_leaves, dim3((bvh.num_tris + 255u) / 256u), dim3(256), 0, s, unsorted, keys, bvh.num_tris, bvh.pad, sorted, nodes);
    for (uint32_t l = 1; l < bvh.num_levels; l++)
        hipLaunchKernelGGL(k_rt_level, dim3((bvh.level_count[l] + 255u) / 256u), dim3(256), 0, s, nodes + bvh.level_offset[l - 1], bvh.level_count[l - 1],
                           nodes + bvh.level_offset[l], bvh.level_count[l]);
    return hipGetLastError();
}
hipError_t launch_probe_trace(const ProbeTraceArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s) {
    if (a.num_probes == 0) return hipSuccess;
    // (the bounce branch is a second instantiation: the reference's configuration, no bounces, runs the kernel without it)
    if (a.gi.num_bounces == 0u) hipLaunchKernelGGL(k_probe_trace<0>, dim3((400u * a.num_probes + 255u) / 256u), dim3(256), 0, s, a, bvh, sc);
    else hipLaunchKernelGGL(k_probe_trace<kRtMaxBounces>, dim3((400u * a.num_probes + 255u) / 256u), dim3(256), 0, s, a, bvh, sc);
    return hipGetLastError();
}
hipError_t launch_rtgi_trace(const RtgiTraceArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s) {
    if (a.row_end <= a.row_begin) return hipSuccess;
    const dim3 grid((a.width + 15u) / 16u, (a.row_end - (a.row_begin & ~15u) + 15u) / 16u);
    if (a.gi.num_bounces == 0u) hipLaunchKernelGGL(k_rtgi_trace<0>, grid, dim3(256), 0, s, a, bvh, sc);
    else hipLaunchKernelGGL(k_rtgi_trace<kRtMaxBounces>, grid, dim3(256), 0, s, a, bvh, sc);
    return hipGetLastError();
}
hipError_t launch_rtao(const RtaoArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s) {
    if (a.row_end <= a.row_begin) return hipSuccess;
    hipLaunchKernelGGL(k_rtao, dim3((a.width + 15u) / 16u, (a.row_end - (a.row_begin & ~15u) + 15u) / 16u), dim3(256), 0, s, a, bvh, sc);
    return hipGetLastError();
}
hipError_t launch_noise_dirs(const PlaneArg& noise, const float* luts, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_noise_dirs, dim3(64), dim3(256), 0, s, noise, luts, out);
    return hipGetLastError();
}
hipError_t launch_sun_shadow_mask(const ShadowMaskArgs& a, const RtBvh& bvh, const RtScene& sc, hipStream_t s) {
    if (a.row_end <= a.row_begin) return hipSuccess;
    hipLaunchKernelGGL(k_sun_shadow_mask, dim3((a.width + 15u) / 16u, (a.row_end - (a.row_begin & ~15u) + 15u) / 16u), dim3(256), 0, s, a, bvh, sc);
    return hipGetLastError();
}

}  // namespace sah
