// Scene rasterisation as compute: the sun shadow cascades, the depth + G-buffer pass and the LPV's RSM.
//   reference: RenderCore/render/directional_light.cpp:286-327, RenderCore/render/phase/gbuffer_phase.cpp:27-97,
//              RenderCore/render/gi/light_propagation_volume.cpp:566-615 (RSM), RenderCore/render/material_pipelines.cpp:13-140,
//              RenderCore/shaders/materials/gltf_basic_pbr.slang:110-253, RenderCore/render/render_scene.cpp:196-222 (cull mode / front face)
// The reference uses the fixed-function rasteriser.  Here: a set-up kernel turns every (view, triangle) into window-space records
// (vertex stage, trivial accept or a queue for the clip kernel — Sutherland-Hodgman against the depth planes and a guard band, fan —
// then 24.8 snapping and culling), the records are binned to 64x64-pixel tiles (count, scan, fill), and one workgroup per tile
// resolves visibility in LDS with ds_min_u32 (D16 shadow maps) or ds_max_u64 on (depth, ~draw order) keys (G-buffer, RSM), sweeping
// 8x8 pixel blocks with exact fp64 edge functions, then shades and writes its tile once, coalesced.  Depth tests are
// order-independent by construction, so the images do not depend on the (nondeterministic) order of the bin lists.  The
// rasterisation rules — the part the API leaves to the implementation — are DESIGN.md §5d; arithmetic follows §3 (every fp32
// operator individually rounded; half expressions rounded after every operator).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "numerics.hpp"
#include "raster_args.hpp"
#include "texture_sample.hpp"

namespace sah {
namespace {

constexpr int kTile = (int)kRasterTile;  // pixels per tile edge
constexpr float kGuardBand = 16.0f;     // |x_c|, |y_c| <= kGuardBand * w_c survives clipping
constexpr float kCoordLimit = 0x1p24f + 4096.0f;  // snapped coordinates beyond this drop the triangle: edge functions stay below 2^52
#ifndef SAH_RASTER_SMALL_AREA
#define SAH_RASTER_SMALL_AREA 64  // 4 / 16 / 64 measured: 64 is best for dense meshes (-8 %), neutral elsewhere
#endif
#ifndef SAH_RASTER_MEDIUM_AREA
#define SAH_RASTER_MEDIUM_AREA 1024  // 256 / 1024 / 4096 measured
#endif
constexpr uint32_t kSmallArea = SAH_RASTER_SMALL_AREA;    // (bbox ∩ tile) pixel count up to which one lane walks a triangle alone
constexpr uint32_t kMediumArea = SAH_RASTER_MEDIUM_AREA;  // ... up to which one wave does; above, the whole workgroup

enum Counter { C_TRIS = 0, C_RECORDS = 1, C_PAIRS = 2, C_CLIPPED = 3, C_STATS = 4, C_EXTRA = 9, C_HEAVY = 10, C_CUTOUT_NO_ATTR = 12, C_BAD_TEXTURE = 13 };  // the last two are statistics words 5 and 6  // C_STATS .. C_STATS+7 mirror SAH_RASTER_STATS_WORDS

struct ClipVertex {
    float c[4];
    float bary[3];
};

SAH_DEV float mat_row(const float* m, int r, float x, float y, float z, float w) { return ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w; }
SAH_DEV void mat_vec(const float* m, const float v[4], float out[4]) {
    for (int r = 0; r < 4; r++) out[r] = mat_row(m, r, v[0], v[1], v[2], v[3]);
}

SAH_DEV float plane_distance(const ClipVertex& v, int plane) {
    switch (plane) {
        case 0: return v.c[2];
        case 1: return v.c[3] - v.c[2];
        case 2: return kGuardBand * v.c[3] - v.c[0];
        case 3: return kGuardBand * v.c[3] + v.c[0];
        case 4: return kGuardBand * v.c[3] - v.c[1];
        default: return kGuardBand * v.c[3] + v.c[1];
    }
}
SAH_DEV ClipVertex lerp_vertex(const ClipVertex& in, const ClipVertex& out, float d_in, float d_out) {
    const float t = d_in / (d_in - d_out);
    ClipVertex r;
    for (int k = 0; k < 4; k++) r.c[k] = in.c[k] + (out.c[k] - in.c[k]) * t;
    for (int k = 0; k < 3; k++) r.bary[k] = in.bary[k] + (out.bary[k] - in.bary[k]) * t;
    return r;
}
// Sutherland-Hodgman in place (rare path: most triangles are inside every plane and skip it)
SAH_DEV int clip_polygon(ClipVertex* poly, ClipVertex* tmp, int n, int first_plane) {
    for (int plane = first_plane; plane < 6 && n >= 3; plane++) {
        int m = 0;
        for (int i = 0; i < n; i++) {
            const ClipVertex a = poly[i];
            const ClipVertex b = poly[i + 1 == n ? 0 : i + 1];
            const float da = plane_distance(a, plane), db = plane_distance(b, plane);
            const bool ia = da >= 0.0f, ib = db >= 0.0f;
            if (ia) tmp[m++] = a;
            if (ia != ib) tmp[m++] = ia ? lerp_vertex(a, b, da, db) : lerp_vertex(b, a, db, da);
        }
        n = m;
        for (int i = 0; i < n; i++) poly[i] = tmp[i];
    }
    return n < 3 ? 0 : n;
}

struct WindowVertex {
    int32_t X, Y;
    float z, inv_w;
    float bary[3];
    bool finite;
};
SAH_DEV bool is_finite(float x) { return __builtin_fabsf(x) < __builtin_inff(); }
SAH_DEV WindowVertex to_window(const ClipVertex& v, float half_w, float half_h) {
    WindowVertex r;
    const float xd = v.c[0] / v.c[3], yd = v.c[1] / v.c[3];
    r.z = v.c[2] / v.c[3];
    r.inv_w = 1.0f / v.c[3];
    const float xf = xd * half_w + half_w, yf = yd * half_h + half_h;
    const float sx = xf * 256.0f, sy = yf * 256.0f;
    r.finite = is_finite(sx) && is_finite(sy) && is_finite(r.z) && is_finite(r.inv_w) && __builtin_fabsf(sx) <= kCoordLimit && __builtin_fabsf(sy) <= kCoordLimit;
    r.X = r.finite ? (int32_t)__builtin_rintf(sx) : 0;
    r.Y = r.finite ? (int32_t)__builtin_rintf(sy) : 0;
    for (int k = 0; k < 3; k++) r.bary[k] = v.bary[k];
    return r;
}

SAH_DEV int32_t first_px(int32_t lo) { const int32_t a = lo - 128; return a <= 0 ? 0 : (a + 255) >> 8; }
SAH_DEV int32_t last_px(int32_t hi, uint32_t size) {
    const int32_t a = hi - 128;
    if (a < 0) return -1;
    const int32_t p = a >> 8;
    return p < (int32_t)size - 1 ? p : (int32_t)size - 1;
}

// ---- K0: exclusive scan (single workgroup, chunked) ------------------------------------------------------------------------------
// mode 0: in[i] = primitives[i].index_count / 3; mode 1: in[i] = values[i].  Eight consecutive elements per thread and round (the
// tile counts of four 4096^2 cascades are 16 K elements: 2 rounds instead of 16, 18 -> 5 us).
__global__ __launch_bounds__(1024) void k_exclusive_scan(const sah_primitive* prims, const uint32_t* values, uint32_t n, uint32_t* out, uint32_t* total) {
    constexpr uint32_t kPer = 8;
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024 * kPer) {
        const uint32_t i0 = base + tid * kPer;
        uint32_t v[kPer], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            const uint32_t i = i0 + k;
            v[k] = i < n ? (prims ? prims[i].index_count / 3u : values[i]) : 0u;
            sum += v[k];
        }
        uint32_t incl = sum;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_base = 0;
        for (uint32_t w = 0; w < wave; w++) wave_base += s_wave[w];
        const uint32_t carry = s_carry;
        uint32_t run = carry + wave_base + incl - sum;
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            if (i0 + k < n) out[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) s_carry = run;
        __syncthreads();
    }
    if (tid == 0) *total = s_carry;
}

// ---- K1: vertex stage, clipping, fan, snapping, culling -> records ----------------------------------------------------------------
SAH_DEV uint32_t find_primitive(const uint32_t* tri_base, uint32_t n, uint32_t t) {  // last p with tri_base[p] <= t
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tri_base[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// One atomic per wave instead of one per lane (the counters are single addresses: per-lane atomics serialise in L2).
SAH_DEV uint32_t wave_sum(uint32_t v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// Adds the workgroup's sum of `local[i]` to counter[i], one global atomic per counter and workgroup (every thread must call this;
// `s_acc` is N words of LDS).  The counters are single addresses and such atomics complete at a few tens of nanoseconds each, device
// wide: per-wave flushes of four counters cost more than the set-up work itself.
template <int N>
SAH_DEV void block_flush(uint32_t* counter, const uint32_t (&local)[N], uint32_t* s_acc) {
    if (threadIdx.x < N) s_acc[threadIdx.x] = 0;
    __syncthreads();
    for (int i = 0; i < N; i++) {
        const uint32_t total = wave_sum(local[i]);
        if ((threadIdx.x & 63u) == 0 && total) atomicAdd(&s_acc[i], total);
    }
    __syncthreads();
    if (threadIdx.x < N && s_acc[threadIdx.x]) atomicAdd(&counter[threadIdx.x], s_acc[threadIdx.x]);
}
// slot for the lanes that `want` one: the first of them adds the count, the rest take consecutive slots
SAH_DEV uint32_t wave_alloc(uint32_t* counter, bool want) {
    const uint64_t mask = __ballot(want);
    uint32_t slot = 0;
    if (want) {
        const uint32_t lane = threadIdx.x & 63u;
        const int leader = __builtin_ctzll(mask);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(mask));
        base = __shfl(base, leader, 64);
        slot = base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
    }
    return slot;
}

// the half-precision varyings of one vertex (gltf_basic_pbr.slang:134-143): colour, normalize(model3x3 * normal), tangent
SAH_DEV void rotate_normalize(const float* m, const float v[3], uint16_t out[3]) {
    float r[3];
    for (int i = 0; i < 3; i++) r[i] = (m[i] * v[0] + m[4 + i] * v[1]) + m[8 + i] * v[2];
    const float inv = 1.0f / __builtin_sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    for (int i = 0; i < 3; i++) out[i] = f2h(r[i] * inv);
}
SAH_DEV void vertex_outputs(const sah_primitive& prim, const sah_vertex_data& vd, uint16_t out[12]) {
    for (int c = 0; c < 4; c++) out[c] = f2h((float)((vd.color >> (8 * c)) & 0xffu) / 255.0f);
    rotate_normalize(prim.model, vd.normal, out + 4);
    rotate_normalize(prim.model, vd.tangent, out + 7);
    out[10] = f2h(vd.tangent[3]);
    out[11] = 0;
}

constexpr uint32_t kAppend = 0xffffffffu;
SAH_DEV void mark_empty(RasterRecord& r) { r.x0 = 1; r.x1 = 0; r.y0 = 1; r.y1 = 0; }
SAH_DEV bool is_empty(const RasterRecord& r) { return r.x0 > r.x1; }
SAH_DEV uint32_t record_count(const RasterArgs& a) {  // direct slots + appended fans, clamped to the buffer
    const uint64_t n = (uint64_t)a.counters[C_TRIS] * a.num_views + a.counters[C_RECORDS];
    return n < a.record_capacity ? (uint32_t)n : a.record_capacity;
}

struct SetupStats {
    uint32_t in = 0, culled = 0, dropped = 0, raster = 0;
};

// One window-space triangle of the fan: facing, bounding box, record.
template <bool GBUFFER>
SAH_DEV void emit_triangle(const RasterArgs& a, SetupStats& st, uint32_t view, uint32_t p, const sah_primitive& prim, uint32_t tri, uint32_t seq,
                           const WindowVertex& v0, WindowVertex v1, WindowVertex v2, uint32_t slot) {
    if (!v0.finite || !v1.finite || !v2.finite) { st.dropped++; return; }
    const int64_t area = (int64_t)(v1.X - v0.X) * (v2.Y - v0.Y) - (int64_t)(v2.X - v0.X) * (v1.Y - v0.Y);
    if (area == 0 || (area < 0 && prim.type == SAH_PRIMITIVE_TYPE_SOLID)) { st.culled++; return; }
    if (area < 0) { const WindowVertex s = v1; v1 = v2; v2 = s; }
    const int32_t minx = min(v0.X, min(v1.X, v2.X)), maxx = max(v0.X, max(v1.X, v2.X));
    const int32_t miny = min(v0.Y, min(v1.Y, v2.Y)), maxy = max(v0.Y, max(v1.Y, v2.Y));
    const int32_t x0 = first_px(minx), x1 = last_px(maxx, a.width), y0 = first_px(miny), y1 = last_px(maxy, a.height);
    if (x0 > x1 || y0 > y1) { st.culled++; return; }
    st.raster++;
    // Unclipped triangles own the slot of their work item (no allocation: a single-address atomic per wave was the bottleneck of this
    // kernel); the fans of clipped ones are appended behind those.
    const uint32_t r = slot != kAppend ? slot : a.counters[C_TRIS] * a.num_views + wave_alloc(&a.counters[C_RECORDS], true);
    if (r >= a.record_capacity) return;  // the host sees the counts, grows the buffer and runs the pass again
    RasterRecord rec;
    rec.X[0] = v0.X; rec.X[1] = v1.X; rec.X[2] = v2.X;
    rec.Y[0] = v0.Y; rec.Y[1] = v1.Y; rec.Y[2] = v2.Y;
    rec.z[0] = v0.z; rec.z[1] = v1.z; rec.z[2] = v2.z;
    rec.view = view;
    rec.x0 = (uint16_t)x0; rec.x1 = (uint16_t)x1; rec.y0 = (uint16_t)y0; rec.y1 = (uint16_t)y1;
    rec.seq = seq;
    // masked geometry is alpha-tested in every pass (shadow_masked_pso / rsm_masked_pso / gbuffer_masked_pso, material_pipelines.cpp:47-140)
    rec.cutout = prim.type == SAH_PRIMITIVE_TYPE_CUTOUT && (GBUFFER || a.shadow_attrs != nullptr);
    a.records[r] = rec;
    if (!GBUFFER && prim.type == SAH_PRIMITIVE_TYPE_CUTOUT) {
        if (a.shadow_attrs) {
            ShadowAttr sa;
            sa.inv_w[0] = v0.inv_w; sa.inv_w[1] = v1.inv_w; sa.inv_w[2] = v2.inv_w;
            for (int k = 0; k < 3; k++) { sa.bary[0][k] = v0.bary[k]; sa.bary[1][k] = v1.bary[k]; sa.bary[2][k] = v2.bary[k]; }
            for (int k = 0; k < 3; k++) {
                const sah_vertex_data& vd = a.vertex_data[(int64_t)prim.vertex_offset + a.indices[prim.first_index + 3 * tri + k]];
                sa.alpha[k] = f2h((float)((vd.color >> 24) & 0xffu) / 255.0f);
            }
            sa.pad = 0;
            sa.material = prim.material;
            sa.pad2 = 0;
            for (int k = 0; k < 3; k++) {
                const sah_vertex_data& vd = a.vertex_data[(int64_t)prim.vertex_offset + a.indices[prim.first_index + 3 * tri + k]];
                sa.uv[k][0] = vd.texcoord[0];
                sa.uv[k][1] = vd.texcoord[1];
            }
            sa.pad3[0] = sa.pad3[1] = 0;
            a.shadow_attrs[r] = sa;
        } else {
            atomicAdd(&a.counters[C_CUTOUT_NO_ATTR], 1u);  // the host turns this into SAH_ERR_INVALID_ARGUMENT (api_raster.cpp)
        }
    }
    if (GBUFFER) {
        RasterAttr at;
        at.inv_w[0] = v0.inv_w; at.inv_w[1] = v1.inv_w; at.inv_w[2] = v2.inv_w;
        for (int k = 0; k < 3; k++) { at.bary[0][k] = v0.bary[k]; at.bary[1][k] = v1.bary[k]; at.bary[2][k] = v2.bary[k]; }
        at.primitive = p;
        at.material = prim.material;
        at.seq = seq;
        at.cutout = prim.type == SAH_PRIMITIVE_TYPE_CUTOUT;
        for (int k = 0; k < 3; k++) {
            const sah_vertex_data& vd = a.vertex_data[(int64_t)prim.vertex_offset + a.indices[prim.first_index + 3 * tri + k]];
            vertex_outputs(prim, vd, at.vout[k]);
            at.uv[k][0] = vd.texcoord[0];
            at.uv[k][1] = vd.texcoord[1];
        }
        a.attrs[r] = at;
    }
}

// vertex stage of corner k of input triangle `tri` (gltf_basic_pbr.slang:126-133)
template <bool GBUFFER>
SAH_DEV ClipVertex clip_vertex(const RasterArgs& a, const sah_primitive& prim, uint32_t view, uint32_t tri, int k) {
    const uint32_t idx = a.indices[prim.first_index + 3 * tri + k];
    const float* pos = a.positions + 3 * ((int64_t)prim.vertex_offset + idx);
    const float local[4] = {pos[0], pos[1], pos[2], 1.0f};
    float world[4], clip[4];
    mat_vec(prim.model, local, world);
    if (GBUFFER && !a.rsm) {
        float vs[4];
        mat_vec(a.view_matrix, world, vs);
        mat_vec(a.clip_matrix[0], vs, clip);
    } else {  // shadow cascades and RSM layers: one world -> clip matrix per view
        mat_vec(a.clip_matrix[view], world, clip);
    }
    ClipVertex c;
    for (int j = 0; j < 4; j++) c.c[j] = clip[j];
    for (int j = 0; j < 3; j++) c.bary[j] = j == k ? 1.0f : 0.0f;
    return c;
}

// Rare path: the triangle crosses a clipping plane.  k_setup queues it and this kernel, launched right after, clips and fans it, so
// that the polygon arrays (scratch memory) and their registers burden only the triangles that need them.
template <bool GBUFFER>
__global__ __launch_bounds__(64) void k_setup_clipped(const RasterArgs a) {
    // the polygons live in LDS, 12 vertices per lane and buffer: dynamically indexed private arrays would sit in scratch memory,
    // and the clipping loop is one long chain of dependent accesses to them
    __shared__ ClipVertex s_poly[64 * 12], s_tmp[64 * 12];
    const uint32_t queued = min(a.counters[C_CLIPPED], a.clip_capacity);
    SetupStats st;
    ClipVertex* poly = s_poly + threadIdx.x * 12;
    for (uint32_t q = blockIdx.x * 64 + threadIdx.x; q < queued; q += gridDim.x * 64) {
        const uint32_t view = a.clip_queue[q].x, t = a.clip_queue[q].y;
        const uint32_t p = find_primitive(a.tri_base, a.num_primitives, t);
        const sah_primitive& prim = a.primitives[p];
        const uint32_t tri = t - a.tri_base[p];
        for (int k = 0; k < 3; k++) poly[k] = clip_vertex<GBUFFER>(a, prim, view, tri, k);
        const int n = clip_polygon(poly, s_tmp + threadIdx.x * 12, 3, GBUFFER ? 0 : 2);
        if (n == 0) { st.culled++; continue; }
        const WindowVertex v0 = to_window(poly[0], a.half_w, a.half_h);
        WindowVertex prev = to_window(poly[1], a.half_w, a.half_h);
        for (int i = 1; i + 1 < n; i++) {
            const WindowVertex next = to_window(poly[i + 1], a.half_w, a.half_h);
            emit_triangle<GBUFFER>(a, st, view, p, prim, tri, t * 8u + (uint32_t)(i - 1), v0, prev, next, kAppend);
            prev = next;
        }
    }
    __shared__ uint32_t s_acc[4];
    const uint32_t local[4] = {0u, st.culled, st.dropped, st.raster};
    block_flush<4>(&a.counters[C_STATS], local, s_acc);
}

template <bool GBUFFER>
__global__ __launch_bounds__(256) void k_setup(const RasterArgs a) {
    const uint32_t total = a.counters[C_TRIS];
    const uint64_t work = (uint64_t)total * a.num_views;
    SetupStats st;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < work; w += (uint64_t)gridDim.x * 256) {
        const uint32_t view = (uint32_t)(w / total), t = (uint32_t)(w % total);
        const uint32_t p = find_primitive(a.tri_base, a.num_primitives, t);
        const sah_primitive& prim = a.primitives[p];
        const uint32_t tri = t - a.tri_base[p];
        st.in++;
        if (w < a.record_capacity) mark_empty(a.records[w]);  // overwritten below if the triangle survives unclipped
        // a draw that points outside the index / vertex / material arrays is dropped, never dereferenced
        bool in_range = (uint64_t)prim.first_index + 3ull * tri + 3ull <= a.num_indices &&
                        (!(GBUFFER || (a.shadow_attrs && prim.type == SAH_PRIMITIVE_TYPE_CUTOUT)) || prim.material < a.num_materials);
        for (int k = 0; k < 3 && in_range; k++) {
            const int64_t v = (int64_t)prim.vertex_offset + a.indices[prim.first_index + 3 * tri + k];
            in_range = v >= 0 && v < (int64_t)a.num_vertices;
        }
        if (!in_range) { st.dropped++; continue; }
        const ClipVertex c0 = clip_vertex<GBUFFER>(a, prim, view, tri, 0), c1 = clip_vertex<GBUFFER>(a, prim, view, tri, 1),
                         c2 = clip_vertex<GBUFFER>(a, prim, view, tri, 2);
        bool finite = true, inside = true;
        for (int j = 0; j < 4; j++) finite = finite && is_finite(c0.c[j]) && is_finite(c1.c[j]) && is_finite(c2.c[j]);
        for (int plane = GBUFFER ? 0 : 2; plane < 6; plane++)
            inside = inside && plane_distance(c0, plane) >= 0.0f && plane_distance(c1, plane) >= 0.0f && plane_distance(c2, plane) >= 0.0f;
        if (!finite) { st.dropped++; continue; }
        if (inside) {
            emit_triangle<GBUFFER>(a, st, view, p, prim, tri, t * 8u, to_window(c0, a.half_w, a.half_h), to_window(c1, a.half_w, a.half_h),
                                   to_window(c2, a.half_w, a.half_h), w < a.record_capacity ? (uint32_t)w : a.record_capacity);
        } else {
            const uint32_t q = wave_alloc(&a.counters[C_CLIPPED], true);
            if (q < a.clip_capacity) a.clip_queue[q] = make_uint2(view, t);  // overflow: the host sees the count and runs the pass again
        }
    }
    __shared__ uint32_t s_acc[4];
    const uint32_t local[4] = {st.in, st.culled, st.dropped, st.raster};
    block_flush<4>(&a.counters[C_STATS], local, s_acc);
}

// No pixel centre of the 64x64 tile at pixel (x0, y0) is inside the record's triangle: some edge function is negative even at the
// tile corner that favours it.  Exact (fp64 on integers below 2^52, see EdgeSetup further down).
SAH_DEV bool tile_outside(const RasterRecord& rec, int32_t x0, int32_t y0) {
    for (int i = 0; i < 3; i++) {
        const int va = (i + 1) % 3, vb = (i + 2) % 3;
        const int32_t dx = rec.X[vb] - rec.X[va], dy = rec.Y[vb] - rec.Y[va];
        // E(px, py) = dx ((256 py + 128) - Ya) - dy ((256 px + 128) - Xa): largest where py is at the dx > 0 end and px at the dy < 0 end
        const int32_t py = dx > 0 ? y0 + kTile - 1 : y0, px = dy < 0 ? x0 + kTile - 1 : x0;
        const double e = (double)dx * (double)(256 * py + 128 - rec.Y[va]) - (double)dy * (double)(256 * px + 128 - rec.X[va]);
        if (e < 0.0) return true;
    }
    return false;
}

// ---- K2 / K4: binning -------------------------------------------------------------------------------------------------------------
// One wave per 64 records.  A record that touches up to 4 tiles is binned by its own lane; wider ones are taken one at a time by the
// whole wave (ballot + readlane), lanes striding over the tiles of the bounding box.
template <bool FILL>
__global__ __launch_bounds__(256) void k_bin(const RasterArgs a) {
    const uint32_t nrec = record_count(a);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * 4u;
    uint32_t st_pairs = 0;
    // records per wave: 64 when there are plenty, fewer when the scene is a handful of screen-filling triangles (each of which is
    // a long loop over tiles that should not queue up behind 63 others in one wave)
    const uint32_t per_wave = min(64u, max(1u, (nrec + waves - 1) / waves));
    for (uint32_t base = (blockIdx.x * 4u + (threadIdx.x >> 6)) * per_wave; base < nrec; base += waves * per_wave) {
        const uint32_t r = base + lane;
        uint32_t tx0 = 1, tx1 = 0, ty0 = 1, ty1 = 0, view = 0;
        bool live = false;
        if (lane < per_wave && r < nrec) {
            const RasterRecord& rec = a.records[r];
            live = !is_empty(rec);
            if (live) { tx0 = rec.x0 / kTile; tx1 = rec.x1 / kTile; ty0 = rec.y0 / kTile; ty1 = rec.y1 / kTile; view = rec.view; }
        }
        const uint32_t ntiles = live ? (tx1 - tx0 + 1) * (ty1 - ty0 + 1) : 0u;
        auto visit = [&](uint32_t tile, uint32_t rec_index) {
            if (FILL) {
                const uint32_t pos = atomicAdd(&a.tile_cursor[tile], 1u);
                const uint32_t at = a.tile_offset[tile] + pos;
                if (at < a.pairs_capacity) a.pairs[at] = rec_index;  // a short list is noticed by the host, which grows it and repeats the pass
            } else {
                atomicAdd(&a.tile_count[tile], 1u);
                st_pairs++;
            }
        };
        // single-tile records (most of a dense mesh): neighbouring triangles land in the same few tiles, so the lanes that share a tile
        // share one atomic — per-address atomic throughput is what bounds this kernel
        uint64_t single = __ballot(ntiles == 1);
        const uint32_t my_tile = (view * a.tiles_y + ty0) * a.tiles_x + tx0;
        while (single) {
            const int leader = __builtin_ctzll(single);
            const uint32_t tile = __shfl(my_tile, leader, 64);
            const uint64_t same = __ballot(ntiles == 1 && my_tile == tile) & single;
            single &= ~same;
            const uint32_t n = (uint32_t)__builtin_popcountll(same);
            const bool mine = (same >> lane) & 1ull;
            if (FILL) {
                uint32_t first = 0;
                if ((int)lane == leader) first = atomicAdd(&a.tile_cursor[tile], n);
                first = __shfl(first, leader, 64);
                const uint32_t at = a.tile_offset[tile] + first + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
                if (mine && at < a.pairs_capacity) a.pairs[at] = r;
            } else {
                if ((int)lane == leader) atomicAdd(&a.tile_count[tile], n);
                st_pairs += mine ? 1u : 0u;
            }
        }
        if (ntiles > 1 && ntiles <= 4)
            for (uint32_t ty = ty0; ty <= ty1; ty++)
                for (uint32_t tx = tx0; tx <= tx1; tx++) visit((view * a.tiles_y + ty) * a.tiles_x + tx, r);
        uint64_t wide = __ballot(ntiles > 4);
        while (wide) {
            const int src = __builtin_ctzll(wide);
            wide &= wide - 1;
            const uint32_t bx0 = __shfl(tx0, src, 64), bx1 = __shfl(tx1, src, 64), by0 = __shfl(ty0, src, 64), by1 = __shfl(ty1, src, 64);
            const uint32_t bview = __shfl(view, src, 64), bw = bx1 - bx0 + 1, count = bw * (by1 - by0 + 1);
            // a tile of the bounding box that lies wholly outside one edge gets no entry (about half the tiles of a large triangle);
            // both binning passes run the same exact test, so count and fill agree
            const RasterRecord& wrec = a.records[base + (uint32_t)src];
            for (uint32_t i = lane; i < count; i += 64) {
                const uint32_t tx = bx0 + i % bw, ty = by0 + i / bw;
                if (count < 16 || !tile_outside(wrec, (int32_t)(tx * kTile), (int32_t)(ty * kTile))) visit((bview * a.tiles_y + ty) * a.tiles_x + tx, base + (uint32_t)src);
            }
        }
    }
    __shared__ uint32_t s_acc[1];
    const uint32_t local[1] = {FILL ? 0u : st_pairs};
    block_flush<1>(&a.counters[C_STATS + 4], local, s_acc);
}

// ---- K5: one workgroup per tile -----------------------------------------------------------------------------------------------------
// Edge functions in fp64.  Window coordinates are integers below 2^24.1 (guard band 16 half-viewports of at most 8192 pixels, 8
// sub-pixel bits), so every product and sum below is an integer of magnitude < 2^52: fp64 evaluates it EXACTLY, and (float)E is the
// single correctly rounded conversion of the integer the specification talks about.  E_i(px, py) = c_i + px a_i + py b_i.
struct EdgeSetup {
    double a[3], b[3], c[3];
    uint32_t tl;        // bit i: edge i is a top or left edge
    double zc, zx, zy;  // depth plane z(px, py) = zc + px zx + py zy
    float inv_area;
    uint32_t seq, cutout;
};
SAH_DEV uint32_t readlane(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
SAH_DEV float readlane(float v, int src) { return __uint_as_float(readlane(__float_as_uint(v), src)); }
SAH_DEV double readlane(double v, int src) {
    const uint64_t bits = __builtin_bit_cast(uint64_t, v);
    return __builtin_bit_cast(double, (uint64_t)readlane((uint32_t)bits, src) | ((uint64_t)readlane((uint32_t)(bits >> 32), src) << 32));
}
// lane `src`'s set-up, in scalar registers of every lane of the wave
SAH_DEV EdgeSetup broadcast(const EdgeSetup& e, int src) {
    EdgeSetup r;
    for (int i = 0; i < 3; i++) { r.a[i] = readlane(e.a[i], src); r.b[i] = readlane(e.b[i], src); r.c[i] = readlane(e.c[i], src); }
    r.zc = readlane(e.zc, src); r.zx = readlane(e.zx, src); r.zy = readlane(e.zy, src);
    r.tl = readlane(e.tl, src);
    r.inv_area = readlane(e.inv_area, src);
    r.seq = readlane(e.seq, src);
    r.cutout = readlane(e.cutout, src);
    return r;
}
SAH_DEV EdgeSetup edge_setup(const RasterRecord& rec) {
    EdgeSetup e;
    e.tl = 0;
    e.seq = rec.seq;
    e.cutout = rec.cutout;
    for (int i = 0; i < 3; i++) {
        const int va = (i + 1) % 3, vb = (i + 2) % 3;  // edge i runs from vertex i+1 to vertex i+2
        const int32_t dx = rec.X[vb] - rec.X[va], dy = rec.Y[vb] - rec.Y[va];
        e.tl |= ((dy < 0) | ((dy == 0) & (dx > 0))) ? 1u << i : 0u;
        e.a[i] = -256.0 * (double)dy;
        e.b[i] = 256.0 * (double)dx;
        e.c[i] = (double)dx * (double)(128 - rec.Y[va]) - (double)dy * (double)(128 - rec.X[va]);
    }
    const double area = (double)(rec.X[1] - rec.X[0]) * (double)(rec.Y[2] - rec.Y[0]) - (double)(rec.X[2] - rec.X[0]) * (double)(rec.Y[1] - rec.Y[0]);
    e.inv_area = 1.0f / (float)area;
    // depth plane: sum_i E_i(px, py) z_i / area, coefficient by coefficient in fp64 (every operator rounded)
    const double inv = 1.0 / area, z0 = (double)rec.z[0], z1 = (double)rec.z[1], z2 = (double)rec.z[2];
    e.zc = ((e.c[0] * z0 + e.c[1] * z1) + e.c[2] * z2) * inv;
    e.zx = ((e.a[0] * z0 + e.a[1] * z1) + e.a[2] * z2) * inv;
    e.zy = ((e.b[0] * z0 + e.b[1] * z1) + e.b[2] * z2) * inv;
    return e;
}
// coverage of pixel (px, py); v = the three edge functions
SAH_DEV bool cover(const EdgeSetup& e, int32_t px, int32_t py, double v[3]) {
    const double x = (double)px, y = (double)py;
    bool inside = true;
    for (int i = 0; i < 3; i++) {
        v[i] = __builtin_fma(x, e.a[i], __builtin_fma(y, e.b[i], e.c[i]));
        // (bitwise, here and in the block sweep: the short-circuit forms compiled to three nested exec-mask regions per pixel in the
        //  innermost loops)
        inside = inside & ((v[i] > 0.0) | ((v[i] == 0.0) & (((e.tl >> i) & 1u) != 0u)));
    }
    return inside;
}
// screen-space barycentrics from the edge functions
SAH_DEV void barycentrics(const EdgeSetup& e, const double v[3], float b[3]) {
    for (int i = 0; i < 3; i++) b[i] = (float)v[i] * e.inv_area;
}
SAH_DEV float fragment_depth(const EdgeSetup& e, int32_t px, int32_t py) {
    const float z = (float)__builtin_fma((double)py, e.zy, __builtin_fma((double)px, e.zx, e.zc));
    return __builtin_fminf(__builtin_fmaxf(z, 0.0f), 1.0f);  // depth clamp (shadow PSO) / [0,1] viewport range; NaN -> 0
}

// perspective-correct barycentrics in the INPUT triangle
SAH_DEV void input_barycentrics(const RasterAttr& at, const float b[3], float lambda[3]) {
    const float q0 = b[0] * at.inv_w[0], q1 = b[1] * at.inv_w[1], q2 = b[2] * at.inv_w[2];
    const float s = (q0 + q1) + q2;
    const float l0 = q0 / s, l1 = q1 / s, l2 = q2 / s;
    for (int k = 0; k < 3; k++) lambda[k] = (l0 * at.bary[0][k] + l1 * at.bary[1][k]) + l2 * at.bary[2][k];
}
// varying `c` of the three vertices, interpolated in fp32 and rounded to half
SAH_DEV Hn interp_h(const RasterAttr& at, const float lambda[3], int c) {
    return Hn((lambda[0] * h2f(at.vout[0][c]) + lambda[1] * h2f(at.vout[1][c])) + lambda[2] * h2f(at.vout[2][c]));
}

SAH_DEV uint32_t unorm8_of(float c) {  // floor(c * 255 + 0.5) in fp32, clamped, NaN -> 0
    if (!(c > 0.0f)) return 0u;
    if (c >= 1.0f) return 255u;
    return (uint32_t)(c * 255.0f + 0.5f);
}

// perspective-correct barycentrics of pixel (px, py) in the input triangle — also for a pixel the triangle does not cover (the other
// pixels of a fragment's quad)
SAH_DEV void lambda_at(const EdgeSetup& e, const float inv_w[3], const float (&bary)[3][3], int32_t px, int32_t py, float lambda[3]) {
    double v[3];
    float b[3];
    cover(e, px, py, v);
    barycentrics(e, v, b);
    const float q0 = b[0] * inv_w[0], q1 = b[1] * inv_w[1], q2 = b[2] * inv_w[2];
    const float s = (q0 + q1) + q2;
    const float l0 = q0 / s, l1 = q1 / s, l2 = q2 / s;
    for (int k = 0; k < 3; k++) lambda[k] = (l0 * bary[0][k] + l1 * bary[1][k]) + l2 * bary[2][k];
}
struct TexCoord {
    float t[2], ddx[2], ddy[2];
};
// the texcoord varying at the fragment and its fine derivatives over the 2x2 quad at even window coordinates
SAH_DEV TexCoord texcoord_of(const EdgeSetup& e, const float inv_w[3], const float (&bary)[3][3], const float (&uv)[3][2], int32_t px, int32_t py,
                             const float lambda[3]) {
    float lx[3], ly[3];
    lambda_at(e, inv_w, bary, px ^ 1, py, lx);
    lambda_at(e, inv_w, bary, px, py ^ 1, ly);
    TexCoord r;
    for (int c = 0; c < 2; c++) {
        const float t = (lambda[0] * uv[0][c] + lambda[1] * uv[1][c]) + lambda[2] * uv[2][c];
        const float tx = (lx[0] * uv[0][c] + lx[1] * uv[1][c]) + lx[2] * uv[2][c];
        const float ty = (ly[0] * uv[0][c] + ly[1] * uv[1][c]) + ly[2] * uv[2][c];
        r.t[c] = t;
        r.ddx[c] = (px & 1) ? t - tx : tx - t;
        r.ddy[c] = (py & 1) ? t - ty : ty - t;
    }
    return r;
}
SAH_DEV sah_material_textures textures_of(const RasterArgs& a, uint32_t material) {
    if (!a.material_textures) return {SAH_TEXTURE_NONE, SAH_TEXTURE_NONE, SAH_TEXTURE_NONE, SAH_TEXTURE_NONE};
    return a.material_textures[min(material, a.num_materials - 1u)];
}
SAH_DEV bool any_texture(const RasterArgs& a, const sah_material_textures& mt) {
    return mt.base_color < a.num_textures || mt.normal < a.num_textures || mt.data < a.num_textures || mt.emission < a.num_textures;
}
// (half4) of one material slot: the sampled texture, or the material's constant texel (an index beyond the table — reported through
// C_BAD_TEXTURE by k_check_textures — is never dereferenced)
SAH_DEV void material_texel(const RasterArgs& a, uint32_t index, const float (&constant)[4], const TexCoord& tc, Hn out[4]) {
    if (index >= a.num_textures || a.counters[C_BAD_TEXTURE] != 0u) {  // (a bad table fails the call: nothing of it is dereferenced)
        for (int c = 0; c < 4; c++) out[c] = Hn(constant[c]);
        return;
    }
    float texel[4];
    sample_texture(a.luts, a.textures[index], tc.t, tc.ddx, tc.ddy, a.shader_mip_bias, texel);
    for (int c = 0; c < 4; c++) out[c] = Hn(texel[c]);
}

// one thread per texture slot and per material: anything the fragment stages could not sample safely is counted, and the host fails the call
__global__ __launch_bounds__(256) void k_check_textures(const RasterArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (i < a.num_textures) {
        const sah_texture& T = a.textures[i];
        bad = T.num_mips < 1 || T.num_mips > SAH_MAX_TEXTURE_MIPS || T.sampler.mag_filter > 1 || T.sampler.min_filter > 1 || T.sampler.mipmap_mode > 1 ||
              T.sampler.address_u > 2 || T.sampler.address_v > 2 || T.sampler.max_anisotropy > 16.0f;
        for (uint32_t l = 0; !bad && l < T.num_mips; l++) {
            const sah_plane& p = T.mips[l];
            bad = !p.ptr || p.width == 0 || p.height == 0 || p.width > 16384 || p.height > 16384 || p.format != T.mips[0].format ||
                  (p.format != SAH_FORMAT_R8G8B8A8_UNORM && p.format != SAH_FORMAT_R8G8B8A8_SRGB) || p.row_pitch_bytes < p.width * 4u;
        }
    } else if (i - a.num_textures < a.num_materials) {
        const sah_material_textures mt = a.material_textures[i - a.num_textures];
        const uint32_t idx[4] = {mt.base_color, mt.normal, mt.data, mt.emission};
        for (int k = 0; k < 4; k++) bad = bad || (idx[k] != SAH_TEXTURE_NONE && idx[k] >= a.num_textures);
    }
    if (bad) atomicAdd(&a.counters[C_BAD_TEXTURE], 1u);
}

// depth test of a covered pixel; v = its edge functions (read by cutout fragments only)
template <bool GBUFFER, bool TEX>
SAH_DEV void emit_fragment(const RasterArgs& a, const EdgeSetup& e, uint32_t rec_index, int32_t px, int32_t py, const double v[3], int32_t tile_x, int32_t tile_y,
                           uint32_t* s_depth, unsigned long long* s_key) {
    const float z = fragment_depth(e, px, py);
    const uint32_t slot = (uint32_t)(py - tile_y) * kTile + (uint32_t)(px - tile_x);
    if (!GBUFFER) {
        if (e.cutout) {  // shadow_masked fragment stage: discard when tinted_base_color.a <= opacity_threshold
            const ShadowAttr& sa = a.shadow_attrs[rec_index];
            float b[3];
            barycentrics(e, v, b);
            const float q0 = b[0] * sa.inv_w[0], q1 = b[1] * sa.inv_w[1], q2 = b[2] * sa.inv_w[2];
            const float sum = (q0 + q1) + q2;
            const float l0 = q0 / sum, l1 = q1 / sum, l2 = q2 / sum;
            float lambda[3];
            for (int k = 0; k < 3; k++) lambda[k] = (l0 * sa.bary[0][k] + l1 * sa.bary[1][k]) + l2 * sa.bary[2][k];
            const Hn va = Hn((lambda[0] * h2f(sa.alpha[0]) + lambda[1] * h2f(sa.alpha[1])) + lambda[2] * h2f(sa.alpha[2]));
            const sah_material& m = a.materials[min(sa.material, a.num_materials - 1u)];
            Hn texel[4] = {Hn(0.f), Hn(0.f), Hn(0.f), Hn(m.base_color_texel[3])};
            const uint32_t tex = TEX ? textures_of(a, sa.material).base_color : SAH_TEXTURE_NONE;
            if (TEX && tex < a.num_textures) material_texel(a, tex, m.base_color_texel, texcoord_of(e, sa.inv_w, sa.bary, sa.uv, px, py, lambda), texel);
            const Hn alpha = texel[3] * va * Hn(m.base_color_tint[3]);
            if (tof(alpha) <= m.opacity_threshold) return;
        }
        atomicMin(&s_depth[slot], (uint32_t)__builtin_rintf(z * 65535.0f));
    } else {
        // key: the larger wins.  G-buffer: reverse-Z depth bits (GREATER against the cleared 0).  RSM: D16 compare LESS against the
        // cleared 1.0, so the key holds 0xffff - code and a fragment at code 0xffff cannot pass.
        const uint32_t depth_key = a.rsm ? 0xffffu - (uint32_t)__builtin_rintf(z * 65535.0f) : __float_as_uint(z);
        if (a.rsm ? depth_key == 0u : !(z > 0.0f)) return;
        if (e.cutout) {  // alpha of tinted_base_color against the threshold (gltf_basic_pbr.slang:181-189)
            const RasterAttr& at = a.attrs[rec_index];
            float b[3], lambda[3];
            barycentrics(e, v, b);
            input_barycentrics(at, b, lambda);
            const sah_material& m = a.materials[min(at.material, a.num_materials - 1u)];  // k_setup validated it; the clamp only matters for stale slots of a pass that is being repeated
            Hn texel[4] = {Hn(0.f), Hn(0.f), Hn(0.f), Hn(m.base_color_texel[3])};
            const uint32_t tex = TEX ? textures_of(a, at.material).base_color : SAH_TEXTURE_NONE;
            if (TEX && tex < a.num_textures) material_texel(a, tex, m.base_color_texel, texcoord_of(e, at.inv_w, at.bary, at.uv, px, py, lambda), texel);
            const Hn alpha = texel[3] * interp_h(at, lambda, 3) * Hn(m.base_color_tint[3]);
            if (tof(alpha) <= m.opacity_threshold) return;
        }
        // low word: who wins among equal depths.  Draw order is all SOLID primitives, then all CUTOUT ones (draw_opaque, draw_masked:
        // gbuffer_phase.cpp:91-93, light_propagation_volume.cpp:611-613), triangles in list order inside a class: order = (class, seq).
        // G-buffer: a depth pre-pass (GREATER) settles the depth, the colour pass runs with compare EQUAL and depth writes off
        // (material_pipelines.cpp gbuffer_pso / gbuffer_masked_pso), so every fragment at the final depth overwrites the targets and
        // the LAST in draw order stays.  RSM: one pass, LESS with depth writes: the FIRST of equal codes stays.
        const uint32_t order = (e.cutout << 31) | e.seq;
        atomicMax(&s_key[slot], ((unsigned long long)depth_key << 32) | (unsigned long long)(a.rsm ? ~order : order));
    }
}
template <bool GBUFFER, bool TEX>
SAH_DEV void test_pixel(const RasterArgs& a, const EdgeSetup& e, uint32_t rec_index, int32_t px, int32_t py, int32_t tile_x, int32_t tile_y, uint32_t* s_depth,
                        unsigned long long* s_key) {
    double v[3];
    if (cover(e, px, py, v)) emit_fragment<GBUFFER, TEX>(a, e, rec_index, px, py, v, tile_x, tile_y, s_depth, s_key);
}

// One wave sweeps rows first_row, first_row + row_step, ... of 8x8 pixel blocks over the clipped bounding box, lanes as the pixels of
// a block.  Per block the edge functions advance by one fp64 add each (exact: integers below 2^52); a block whose most favourable
// corner is outside an edge is skipped, one whose least favourable corner is inside all three needs no per-pixel coverage test.
template <bool GBUFFER, bool TEX>
SAH_DEV void sweep(const RasterArgs& a, const EdgeSetup& e, uint32_t rec_index, int32_t sx0, int32_t sx1, int32_t bx1, int32_t by1, int32_t first_row,
                   int32_t row_step, uint32_t lane, int32_t tile_x, int32_t tile_y, uint32_t* s_depth, unsigned long long* s_key) {
    // blocks start at x = sx0, sx0 + 8, ... <= sx1; pixels beyond (bx1, by1) are outside the record's clipped bounding box
    const int32_t bx0 = sx0;
    const int32_t lx = (int32_t)(lane & 7u), ly = (int32_t)(lane >> 3);
    double kmax[3], kmin[3], lane_off[3], step_x[3];
    for (int i = 0; i < 3; i++) {
        kmax[i] = 7.0 * (__builtin_fmax(e.a[i], 0.0) + __builtin_fmax(e.b[i], 0.0));
        kmin[i] = 7.0 * (__builtin_fmin(e.a[i], 0.0) + __builtin_fmin(e.b[i], 0.0));
        lane_off[i] = __builtin_fma((double)lx, e.a[i], (double)ly * e.b[i]);
        step_x[i] = 8.0 * e.a[i];
    }
    for (int32_t oy = first_row; oy <= by1; oy += row_step) {
        double base[3];
        for (int i = 0; i < 3; i++) base[i] = __builtin_fma((double)oy, e.b[i], __builtin_fma((double)bx0, e.a[i], e.c[i]));
        for (int32_t ox = bx0; ox <= sx1; ox += 8) {
            bool outside = false, all_in = true;
            for (int i = 0; i < 3; i++) {
                outside = outside | (base[i] + kmax[i] < 0.0);
                all_in = all_in & (base[i] + kmin[i] > 0.0);
            }
            const int32_t px = ox + lx, py = oy + ly;
            if (!outside && px <= bx1 && py <= by1) {
                double v[3] = {0.0, 0.0, 0.0};
                bool covered = all_in;
                if (!all_in || e.cutout) {
                    covered = true;
                    for (int i = 0; i < 3; i++) {
                        v[i] = base[i] + lane_off[i];
                        covered = covered & ((v[i] > 0.0) | ((v[i] == 0.0) & (((e.tl >> i) & 1u) != 0u)));
                    }
                }
                if (covered) emit_fragment<GBUFFER, TEX>(a, e, rec_index, px, py, v, tile_x, tile_y, s_depth, s_key);
            }
            for (int i = 0; i < 3; i++) base[i] += step_x[i];
        }
    }
}

// fragment stage of the winning triangle (gltf_basic_pbr.slang:169-253, SAH_MAIN_VIEW, constant textures)
template <bool TEX>
SAH_DEV void shade_and_store(const RasterArgs& a, const EdgeSetup& e, const RasterAttr& at, const sah_material& m, const sah_material_textures& mt, int32_t px,
                             int32_t py, float z) {
    double v[3];
    float b[3], lambda[3];
    cover(e, px, py, v);
    barycentrics(e, v, b);
    input_barycentrics(at, b, lambda);
    Hn base_texel[4], normal_texel[4], data_texel[4], emission_texel[4];
    if (TEX) {
        TexCoord tc{};
        if (any_texture(a, mt)) tc = texcoord_of(e, at.inv_w, at.bary, at.uv, px, py, lambda);
        material_texel(a, mt.base_color, m.base_color_texel, tc, base_texel);
        material_texel(a, mt.normal, m.normal_texel, tc, normal_texel);
        material_texel(a, mt.data, m.data_texel, tc, data_texel);
        material_texel(a, mt.emission, m.emission_texel, tc, emission_texel);
    } else {
        for (int c = 0; c < 4; c++) {
            base_texel[c] = Hn(m.base_color_texel[c]);
            normal_texel[c] = Hn(m.normal_texel[c]);
            data_texel[c] = Hn(m.data_texel[c]);
            emission_texel[c] = Hn(m.emission_texel[c]);
        }
    }
    Hn col[4], N[3], T[4];
    for (int c = 0; c < 4; c++) col[c] = interp_h(at, lambda, c);
    for (int c = 0; c < 3; c++) N[c] = interp_h(at, lambda, 4 + c);
    for (int c = 0; c < 4; c++) T[c] = interp_h(at, lambda, 7 + c);
    Hn tinted[4];
    for (int c = 0; c < 4; c++) tinted[c] = base_texel[c] * col[c] * Hn(m.base_color_tint[c]);
    // bitangent = cross(normal, tangent.xyz) * tangent.w; normal = normal_sample * TBN (:197-207)
    const Hn B[3] = {(N[1] * T[2] - N[2] * T[1]) * T[3], (N[2] * T[0] - N[0] * T[2]) * T[3], (N[0] * T[1] - N[1] * T[0]) * T[3]};
    Hn ns[3], n_out[3];
    for (int c = 0; c < 3; c++) ns[c] = normal_texel[c] * Hn::lit(2.0f) - Hn::lit(1.0f);
    for (int c = 0; c < 3; c++) n_out[c] = ns[0] * T[c] + ns[1] * B[c] + ns[2] * N[c];
    const float factor[4] = {0.0f, m.roughness_factor, m.metalness_factor, 0.0f};
    uint32_t color_bits = 0, data_bits = 0, emission_bits = 0;
    for (int c = 0; c < 4; c++) {
        const Hn d = data_texel[c] * Hn(factor[c]);
        const Hn em = emission_texel[c] * Hn(m.emission_factor[c]);
        data_bits |= unorm8_of(tof(d)) << (8 * c);
        emission_bits |= (c < 3 ? (uint32_t)a.half_to_srgb8[__builtin_bit_cast(uint16_t, em.v)] : unorm8_of(tof(em))) << (8 * c);
        color_bits |= (c < 3 ? (uint32_t)a.half_to_srgb8[__builtin_bit_cast(uint16_t, tinted[c].v)] : unorm8_of(tof(tinted[c]))) << (8 * c);
    }
    uint2 nbits;
    nbits.x = (uint32_t)__builtin_bit_cast(uint16_t, n_out[0].v) | ((uint32_t)__builtin_bit_cast(uint16_t, n_out[1].v) << 16);
    nbits.y = (uint32_t)__builtin_bit_cast(uint16_t, n_out[2].v);
    *(uint32_t*)(a.out_color.ptr + (size_t)py * a.out_color.pitch + (size_t)px * 4) = color_bits;
    *(uint2*)(a.out_normals.ptr + (size_t)py * a.out_normals.pitch + (size_t)px * 8) = nbits;
    *(uint32_t*)(a.out_data.ptr + (size_t)py * a.out_data.pitch + (size_t)px * 4) = data_bits;
    *(uint32_t*)(a.out_emission.ptr + (size_t)py * a.out_emission.pitch + (size_t)px * 4) = emission_bits;
    *(float*)(a.out_depth.ptr + (size_t)py * a.out_depth.pitch + (size_t)px * 4) = z;
}

// RSM fragment stage of the winning triangle (gltf_basic_pbr.slang:169-253, SAH_RSM): flux = Fd(surface, -sun direction, normal) with
// the metalness / roughness this variant leaves at 0, normal * 0.5 + 0.5; the D16 code goes to the depth layer
template <bool TEX>
SAH_DEV void shade_rsm_and_store(const RasterArgs& a, const EdgeSetup& e, const RasterAttr& at, const sah_material& m, const sah_material_textures& mt, uint32_t layer,
                                 int32_t px, int32_t py, uint32_t depth_code) {
    double v[3];
    float b[3], lambda[3];
    cover(e, px, py, v);
    barycentrics(e, v, b);
    input_barycentrics(at, b, lambda);
    Hn base_texel[4];
    if (TEX) {
        TexCoord tc{};
        if (mt.base_color < a.num_textures) tc = texcoord_of(e, at.inv_w, at.bary, at.uv, px, py, lambda);
        material_texel(a, mt.base_color, m.base_color_texel, tc, base_texel);
    } else {
        for (int c = 0; c < 4; c++) base_texel[c] = Hn(m.base_color_texel[c]);
    }
    Hn tinted[3], N[3];
    for (int c = 0; c < 3; c++) tinted[c] = base_texel[c] * interp_h(at, lambda, c) * Hn(m.base_color_tint[c]);
    for (int c = 0; c < 3; c++) N[c] = interp_h(at, lambda, 4 + c);
    Surface<Hn> s;
    s.base_color = {tinted[0], tinted[1], tinted[2]};
    s.normal = {N[0], N[1], N[2]};
    s.metalness = Hn::lit(0.0f);
    s.roughness = Hn::lit(0.0f);
    const V3<Hn> l = {-Hn(a.sun_direction[0]), -Hn(a.sun_direction[1]), -Hn(a.sun_direction[2])};
    const V3<Hn> flux = Fd(s, l, s.normal);
    const uint32_t flux_bits = (uint32_t)a.half_to_srgb8[__builtin_bit_cast(uint16_t, flux.x.v)] | ((uint32_t)a.half_to_srgb8[__builtin_bit_cast(uint16_t, flux.y.v)] << 8) |
                               ((uint32_t)a.half_to_srgb8[__builtin_bit_cast(uint16_t, flux.z.v)] << 16) | 0xff000000u;
    uint32_t normal_bits = 0xff000000u;
    for (int c = 0; c < 3; c++) normal_bits |= unorm8_of(tof(N[c] * Hn::lit(0.5f) + Hn::lit(0.5f))) << (8 * c);
    *(uint32_t*)(a.rsm_flux.ptr + (size_t)layer * a.rsm_flux.slice_pitch + (size_t)py * a.rsm_flux.row_pitch + (size_t)px * 4) = flux_bits;
    *(uint32_t*)(a.rsm_normals.ptr + (size_t)layer * a.rsm_normals.slice_pitch + (size_t)py * a.rsm_normals.row_pitch + (size_t)px * 4) = normal_bits;
    *(uint16_t*)(a.rsm_depth.ptr + (size_t)layer * a.rsm_depth.slice_pitch + (size_t)py * a.rsm_depth.row_pitch + (size_t)px * 2) = (uint16_t)depth_code;
}

constexpr uint32_t kTileThreads = 256;   // 1024 (16 waves per tile, to shorten the densest tiles) measured 1.1x - 2.5x slower
constexpr uint32_t kSplit = kRasterSplit;  // bin lists longer than this are cut into parts of this many entries, one workgroup each
constexpr uint32_t kBigSlots = 64;       // workgroup-cooperative records per round of list entries; the rest fall back to their wave
struct BigRecord {
    EdgeSetup e;
    uint32_t rec_index;
    int32_t x0, x1, y0, y1;
};

template <bool GBUFFER, bool TEX>
__global__ __launch_bounds__(kTileThreads, (GBUFFER && !TEX) ? 3 : 1) void k_raster_tiles(const RasterArgs a) {
    __shared__ uint32_t s_depth[GBUFFER ? 1 : kTile * kTile];
    __shared__ unsigned long long s_key[GBUFFER ? kTile * kTile : 1];
    // Workgroups 0 .. ntiles-1 own a tile (and part 0 of its list); the rest take the further parts of the lists that k_split cut
    // into pieces of kSplit entries: the densest tiles of a scene would otherwise set the duration of the whole kernel.
    const uint32_t ntiles = a.tiles_x * a.tiles_y * a.num_views;
    uint32_t tile = blockIdx.x, part = 0;
    if (blockIdx.x >= ntiles) {
        const uint32_t k = blockIdx.x - ntiles;
        if (k >= min(a.counters[C_EXTRA], a.extra_capacity)) return;
        tile = a.extra_parts[k].x;
        part = a.extra_parts[k].y;
        if (tile >= ntiles) return;  // never for a pass whose buffers were large enough (a too-small pass is repeated by the host)
    }
    const uint32_t tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, view = tile / (a.tiles_x * a.tiles_y);
    const int32_t tile_x = (int32_t)tx * kTile, tile_y = (int32_t)ty * kTile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < kTile * kTile; i += kTileThreads) {
        if (GBUFFER) s_key[i] = 0ull; else s_depth[i] = 0xffffu;
    }
    __syncthreads();
    // The tile's list in rounds of 256 entries, one per thread.  A record whose bounding box covers at most kSmallArea pixels of the
    // tile is walked by its own lane, up to kMediumArea by its wave; the others go to an LDS list and are rasterised by the whole
    // workgroup, one after the other: lanes form an 8x8 pixel block, the four waves take alternate block rows of the bounding box.
    __shared__ BigRecord s_big[kBigSlots];
    __shared__ uint32_t s_nbig;
    // (a bin list that does not fit the buffer is not read: the host repeats the pass with a larger one)
    const uint32_t whole = (uint64_t)a.tile_offset[tile] + a.tile_count[tile] <= a.pairs_capacity ? a.tile_count[tile] : 0u;
    const uint32_t slot_of_tile = a.heavy_slot[tile] < a.merge_capacity ? a.heavy_slot[tile] : ~0u;  // ~0: the list is not split
    const uint32_t parts = slot_of_tile != ~0u ? (whole + kSplit - 1) / kSplit : 1u;
    if (part >= parts) return;
    const uint32_t first = slot_of_tile != ~0u ? part * kSplit : 0u;
    const uint32_t begin = a.tile_offset[tile] + first, count = slot_of_tile != ~0u ? min(kSplit, whole - min(whole, first)) : whole;
    for (uint32_t base = 0; base < count; base += kTileThreads) {
        if (tid == 0) s_nbig = 0;
        __syncthreads();
        // consecutive list entries go to different waves: a typical list is shorter than one round, and `base + tid` would hand
        // all of it to wave 0 (whose medium records are processed one after the other) while the other waves idle
        const uint32_t li = base + lane * (kTileThreads / 64u) + wave;
        uint32_t rec_index = 0, area = 0;
        int32_t x0 = 1, x1 = 0, y0 = 1, y1 = 0;
        EdgeSetup mine{};
        bool medium_rec = false;
        if (li < count) {
            rec_index = a.pairs[begin + li];
            const RasterRecord rec = a.records[rec_index];
            x0 = max((int32_t)rec.x0, tile_x); x1 = min((int32_t)rec.x1, tile_x + kTile - 1);
            y0 = max((int32_t)rec.y0, tile_y); y1 = min((int32_t)rec.y1, tile_y + kTile - 1);
            area = (uint32_t)((x1 - x0 + 1) * (y1 - y0 + 1));
            mine = edge_setup(rec);  // every lane sets up its own record: 64 set-ups for the price of one
            medium_rec = area > kSmallArea && area <= kMediumArea;
            if (area <= kSmallArea) {
                for (int32_t py = y0; py <= y1; py++)
                    for (int32_t px = x0; px <= x1; px++) test_pixel<GBUFFER, TEX>(a, mine, rec_index, px, py, tile_x, tile_y, s_depth, s_key);
            } else if (area > kMediumArea) {
                const uint32_t slot = atomicAdd(&s_nbig, 1u);
                if (slot < kBigSlots) {
                    s_big[slot].e = mine;
                    s_big[slot].rec_index = rec_index;
                    s_big[slot].x0 = x0; s_big[slot].x1 = x1; s_big[slot].y0 = y0; s_big[slot].y1 = y1;
                } else {
                    medium_rec = true;  // list full: the wave does it
                }
            }
        }
        // medium records: one at a time by the wave that read them, lanes as an 8x8 block sweeping the clipped bounding box; the
        // owner lane's set-up moves to scalar registers with v_readlane (no memory round trip per record)
        uint64_t medium = __ballot(medium_rec);
        while (medium) {
            const int src = __builtin_ctzll(medium);
            medium &= medium - 1;
            const uint32_t ri = readlane(rec_index, src);
            const int32_t bx0 = (int32_t)readlane((uint32_t)x0, src), bx1 = (int32_t)readlane((uint32_t)x1, src);
            const int32_t by0 = (int32_t)readlane((uint32_t)y0, src), by1 = (int32_t)readlane((uint32_t)y1, src);
            const EdgeSetup e = broadcast(mine, src);
            sweep<GBUFFER, TEX>(a, e, ri, bx0, bx1, bx1, by1, by0, 8, lane, tile_x, tile_y, s_depth, s_key);
        }
        __syncthreads();
        const uint32_t nbig = min(s_nbig, kBigSlots);
        for (uint32_t k = 0; k < nbig; k++) {
            const EdgeSetup e = s_big[k].e;  // same address in every lane: an LDS broadcast
            const uint32_t ri = s_big[k].rec_index;
            const int32_t bx0 = s_big[k].x0, bx1 = s_big[k].x1, by0 = s_big[k].y0, by1 = s_big[k].y1;
            if constexpr (kTileThreads == 256) {  // 4 waves: alternate block rows
                sweep<GBUFFER, TEX>(a, e, ri, bx0, bx1, bx1, by1, by0 + 8 * (int32_t)wave, 32, lane, tile_x, tile_y, s_depth, s_key);
            } else {  // 16 waves: block row (wave % 8) of the at most 8, left or right half of the block columns (wave / 8)
                const int32_t half = (((bx1 - bx0) >> 3) + 2) >> 1;
                const int32_t sx0 = bx0 + 8 * half * (int32_t)(wave >> 3), sx1 = min(bx1, sx0 + 8 * half - 1);
                if (sx0 <= bx1) sweep<GBUFFER, TEX>(a, e, ri, sx0, sx1, bx1, by1, by0 + 8 * (int32_t)(wave & 7u), 64, lane, tile_x, tile_y, s_depth, s_key);
            }
        }
        // every wave has read s_nbig / s_big of this round before thread 0 resets the counter for the next one (lists left unsplit
        // take several rounds)
        __syncthreads();
    }
    if (parts > 1) {
        // min / max are associative: every part folds its tile into the tile's buffer in global memory; the part that arrives last
        // (ticket) reads the merged tile back and writes the images
        __shared__ uint32_t s_last;
        const size_t base_index = (size_t)slot_of_tile * (kTile * kTile);
        for (uint32_t i = tid; i < kTile * kTile; i += kTileThreads) {
            if (GBUFFER) { if (s_key[i] != 0ull) atomicMax(&a.merge_keys[base_index + i], s_key[i]); }
            else { if (s_depth[i] != 0xffffu) atomicMin(&a.merge_depth[base_index + i], s_depth[i]); }
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(&a.tickets[slot_of_tile], 1u) == parts - 1u;
        __syncthreads();
        if (!s_last) return;
        __threadfence();
        for (uint32_t i = tid; i < kTile * kTile; i += kTileThreads) {
            if (GBUFFER) s_key[i] = __hip_atomic_load(&a.merge_keys[base_index + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else s_depth[i] = __hip_atomic_load(&a.merge_depth[base_index + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (!GBUFFER) {
        // 64 texels of D16 per row = 32 dwords; 256 threads write 8 rows per step
        uint8_t* base = (uint8_t*)a.shadowmap.ptr + (size_t)view * a.shadowmap.slice_pitch;
        const bool pair_ok = (a.shadowmap.row_pitch % 4u) == 0 && ((uintptr_t)a.shadowmap.ptr % 4u) == 0 && (a.shadowmap.slice_pitch % 4u) == 0;
        for (uint32_t i = tid; i < kTile * kTile / 2; i += kTileThreads) {
            const uint32_t row = i / (kTile / 2), col = (i % (kTile / 2)) * 2;
            const uint32_t px = (uint32_t)tile_x + col, py = (uint32_t)tile_y + row;
            if (py >= a.height || px >= a.width) continue;
            const uint32_t d0 = s_depth[row * kTile + col], d1 = s_depth[row * kTile + col + 1];
            uint8_t* dst = base + (size_t)py * a.shadowmap.row_pitch + (size_t)px * 2;
            if (pair_ok && px + 1 < a.width) {
                *(uint32_t*)dst = d0 | (d1 << 16);
            } else {
                *(uint16_t*)dst = (uint16_t)d0;
                if (px + 1 < a.width) *(uint16_t*)(dst + 2) = (uint16_t)d1;
            }
        }
    } else {
        // A thread's pixels are 4 rows apart in one column: consecutive ones usually belong to the same triangle, whose record,
        // varyings and material (three dependent gathers) are then kept from the previous pixel.
        uint32_t cached = ~0u;
        EdgeSetup e_c{};
        RasterAttr at_c{};
        sah_material m_c{};
        sah_material_textures mt_c{};
        for (uint32_t i = tid; i < kTile * kTile; i += kTileThreads) {
            const int32_t px = tile_x + (int32_t)(i % kTile), py = tile_y + (int32_t)(i / kTile);
            if ((uint32_t)px >= a.width || (uint32_t)py >= a.height) continue;
            const unsigned long long key = s_key[i];
            if (key == 0ull && a.rsm) {  // clear values, light_propagation_volume.cpp:586-606
                *(uint32_t*)(a.rsm_flux.ptr + (size_t)view * a.rsm_flux.slice_pitch + (size_t)py * a.rsm_flux.row_pitch + (size_t)px * 4) = 0u;
                *(uint32_t*)(a.rsm_normals.ptr + (size_t)view * a.rsm_normals.slice_pitch + (size_t)py * a.rsm_normals.row_pitch + (size_t)px * 4) = 0x00ff8080u;
                *(uint16_t*)(a.rsm_depth.ptr + (size_t)view * a.rsm_depth.slice_pitch + (size_t)py * a.rsm_depth.row_pitch + (size_t)px * 2) = 0xffffu;
            } else if (key == 0ull) {  // clear values, gbuffer_phase.cpp:66-87
                *(uint32_t*)(a.out_color.ptr + (size_t)py * a.out_color.pitch + (size_t)px * 4) = 0u;
                *(uint2*)(a.out_normals.ptr + (size_t)py * a.out_normals.pitch + (size_t)px * 8) = make_uint2(0x38003800u, 0x00003c00u);
                *(uint32_t*)(a.out_data.ptr + (size_t)py * a.out_data.pitch + (size_t)px * 4) = 0u;
                *(uint32_t*)(a.out_emission.ptr + (size_t)py * a.out_emission.pitch + (size_t)px * 4) = 0u;
                *(float*)(a.out_depth.ptr + (size_t)py * a.out_depth.pitch + (size_t)px * 4) = 0.0f;
            } else {
                // an unclipped triangle sits in the slot of its work item (one view: slot = running triangle number = seq / 8);
                // the fans of clipped ones were appended and are found through the table
                const uint32_t seq = (a.rsm ? ~(uint32_t)key : (uint32_t)key) & 0x7fffffffu, total = a.counters[C_TRIS];
                uint64_t r = (uint64_t)view * total + (seq >> 3);
                if ((seq & 7u) != 0u || r >= a.record_capacity || is_empty(a.records[r])) {
                    const uint64_t slot = (uint64_t)view * total * 8u + seq;
                    r = slot < a.seq_capacity ? a.seq_to_record[slot] : 0u;
                    if (r >= a.record_capacity) r = 0;  // only when a scratch buffer was too small: the pass is repeated
                }
                if ((uint32_t)r != cached) {
                    cached = (uint32_t)r;
                    e_c = edge_setup(a.records[r]);
                    at_c = a.attrs[r];
                    m_c = a.materials[min(at_c.material, a.num_materials - 1u)];  // k_setup validated it; the clamp only matters for stale slots of a repeated pass
                    if (TEX) mt_c = textures_of(a, at_c.material);
                }
                if (a.rsm) shade_rsm_and_store<TEX>(a, e_c, at_c, m_c, mt_c, view, px, py, 0xffffu - (uint32_t)(key >> 32));
                else shade_and_store<TEX>(a, e_c, at_c, m_c, mt_c, px, py, __uint_as_float((uint32_t)(key >> 32)));
            }
        }
    }
}

// Cuts long bin lists into parts: a tile with more than kSplit entries gets a merge buffer (initialised here to the identity of its
// depth test), a ticket, and one extra workgroup per further part.  Tiles beyond the scratch capacity stay unsplit (slower, not wrong).
template <bool GBUFFER>
__global__ __launch_bounds__(256) void k_split(const RasterArgs a) {
    const uint32_t ntiles = a.tiles_x * a.tiles_y * a.num_views;
    const uint32_t tile = blockIdx.x;  // one workgroup per tile: the merge buffer of a heavy tile is initialised by all 256 threads
    __shared__ uint32_t s_slot;
    if (tile >= ntiles) return;
    if (threadIdx.x == 0) {
        uint32_t slot = ~0u;
        const uint32_t count = a.tile_count[tile];
        if (count > kSplit && (uint64_t)a.tile_offset[tile] + count <= a.pairs_capacity) {
            const uint32_t parts = (count + kSplit - 1) / kSplit;
            const uint32_t h = atomicAdd(&a.counters[C_HEAVY], 1u);
            if (h < a.merge_capacity) {
                const uint32_t e = atomicAdd(&a.counters[C_EXTRA], parts - 1u);
                if ((uint64_t)e + parts - 1u <= a.extra_capacity) {
                    slot = h;
                    a.tickets[h] = 0u;
                    for (uint32_t p = 1; p < parts; p++) a.extra_parts[e + p - 1u] = make_uint2(tile, p);
                }
            }
        }
        a.heavy_slot[tile] = slot;
        s_slot = slot;
    }
    __syncthreads();
    if (s_slot == ~0u) return;
    const size_t base_index = (size_t)s_slot * (kTile * kTile);
    for (uint32_t i = threadIdx.x; i < kTile * kTile; i += 256) {
        if (GBUFFER) a.merge_keys[base_index + i] = 0ull; else a.merge_depth[base_index + i] = 0xffffu;
    }
}

// seq -> record index for the appended records (fans of clipped triangles; G-buffer resolve)
__global__ __launch_bounds__(256) void k_seq_table(const RasterArgs a) {
    const uint32_t nrec = record_count(a), first = a.counters[C_TRIS] * a.num_views;
    for (uint32_t r = first + blockIdx.x * 256 + threadIdx.x; r < nrec; r += gridDim.x * 256)
    {
        const uint64_t slot = (uint64_t)a.records[r].view * a.counters[C_TRIS] * 8u + a.records[r].seq;
        if (slot < a.seq_capacity) a.seq_to_record[slot] = r;
    }
}

}  // namespace

// Stage 1: scan the draws, set up the records, count the bins, scan the bins.  The caller then reads `counters` back.
hipError_t launch_raster_setup(const RasterArgs& a, bool gbuffer, hipStream_t st) {
    hipError_t e = hipMemsetAsync(a.counters, 0, 16 * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    const uint32_t ntiles = a.tiles_x * a.tiles_y * a.num_views;
    e = hipMemsetAsync(a.tile_count, 0, (size_t)ntiles * 2 * sizeof(uint32_t), st);  // tile_count and tile_cursor are adjacent
    if (e != hipSuccess) return e;
    if (a.num_primitives == 0) return hipSuccess;
    if (a.textures) hipLaunchKernelGGL(k_check_textures, dim3((a.num_textures + a.num_materials + 255u) / 256u), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, st, a.primitives, (const uint32_t*)nullptr, a.num_primitives, a.tri_base, &a.counters[C_TRIS]);
    if (gbuffer) {
        hipLaunchKernelGGL(k_setup<true>, dim3(1024), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_setup_clipped<true>, dim3(256), dim3(64), 0, st, a);
    } else {
        hipLaunchKernelGGL(k_setup<false>, dim3(1024), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_setup_clipped<false>, dim3(256), dim3(64), 0, st, a);
    }
    hipLaunchKernelGGL(k_bin<false>, dim3(1024), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, st, (const sah_primitive*)nullptr, (const uint32_t*)a.tile_count, ntiles, a.tile_offset, &a.counters[C_PAIRS]);
    return hipGetLastError();
}

// Stage 2: fill the bins, rasterise and write the images.
hipError_t launch_raster_tiles(const RasterArgs& a, bool gbuffer, hipStream_t st) {
    const uint32_t ntiles = a.tiles_x * a.tiles_y * a.num_views;
    if (a.num_primitives) {
        hipLaunchKernelGGL(k_bin<true>, dim3(1024), dim3(256), 0, st, a);
        if (gbuffer) hipLaunchKernelGGL(k_seq_table, dim3(64), dim3(256), 0, st, a);
    }
    // every tile gets its heavy_slot (~0 when its list stays whole), also for an empty scene
    if (gbuffer) hipLaunchKernelGGL(k_split<true>, dim3(ntiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_split<false>, dim3(ntiles), dim3(256), 0, st, a);
    // (the texture-sampling fragment stages are their own instantiations: 30-45 more VGPRs, which would cost the plain ones a wave per SIMD)
    const dim3 grid(ntiles + a.extra_capacity);
    if (gbuffer && a.textures) hipLaunchKernelGGL((k_raster_tiles<true, true>), grid, dim3(kTileThreads), 0, st, a);
    else if (gbuffer) hipLaunchKernelGGL((k_raster_tiles<true, false>), grid, dim3(kTileThreads), 0, st, a);
    else if (a.textures && a.shadow_attrs) hipLaunchKernelGGL((k_raster_tiles<false, true>), grid, dim3(kTileThreads), 0, st, a);
    else hipLaunchKernelGGL((k_raster_tiles<false, false>), grid, dim3(kTileThreads), 0, st, a);
    return hipGetLastError();
}

}  // namespace sah
