// The two passes between the RSM and the LPV:
//   "Extract VPLs"   RenderCore/shaders/gi/lpv/rsm_generate_vpls.comp:44-139 (dispatch: RenderCore/render/gi/light_propagation_volume.cpp:636-686)
//   "VPL Injection"  RenderCore/shaders/gi/lpv/vpl_injection.{vert,frag} drawn as a point list with additive blending (:699-760)
// Both are tiny (at most (res/2)^2 = 4096 lights per cascade at the default RSM resolution of 128) and launch bound; what matters
// is that their results are functions of the input.  The reference appends lights with atomicAdd and lets the blend unit add them
// in that order; include/sah_hip.h fixes the order to the ascending invocation index, so: the extraction parks every invocation's
// light in scratch and one workgroup compacts them with a prefix sum, and the injection adds the lights of a cell one after the other in list order, rounding to half after
// every addition, one thread per occupied cell (the thread of the cell's first light).  Arithmetic: GLSL fp32, every operator rounded.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "numerics.hpp"
#include "params.hpp"
#include "../../include/sah_hip.h"

namespace sah {
namespace {

struct Vpl {
    float position[3], color[3], normal[3];
};

SAH_DEV const uint8_t* texel(const VolumeArg& v, uint32_t layer, int x, int y, int bpp) {
    return v.ptr + (size_t)layer * v.slice_pitch + (size_t)y * v.row_pitch + (size_t)x * bpp;
}
SAH_DEV void mat_vec4(const float* m, float x, float y, float z, float w, float out[4]) {
    for (int r = 0; r < 4; r++) out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

struct ExtractArgs {
    VolumeArg flux, normals, depth;
    float inverse_rsm_vp[16], world_to_cascade[16];
    uint32_t cascade, res;
    float cascade_f, side;  // side = grid_cell_size * 32
    const float* srgb_lut;  // 256 sRGB8 -> linear, then 256 UNORM8 -> float (ctx->luts)
    sah_packed_vpl* list;
    uint32_t* count;
    sah_packed_vpl* candidates;  // scratch: one per invocation
    uint32_t* keep;              // scratch: 1 when the invocation stores its light
};

SAH_DEV Vpl load_rsm_vpl(const ExtractArgs& a, int x, int y) {
    const float depth = (float)*(const uint16_t*)texel(a.depth, a.cascade, x, y, 2) / 65535.0f;
    const float res = (float)a.res;
    const float tx = ((float)x + 0.5f) / res, ty = ((float)y + 0.5f) / res;
    float ws[4];
    mat_vec4(a.inverse_rsm_vp, tx * 2.0f - 1.0f, ty * 2.0f - 1.0f, depth, 1.0f, ws);
    Vpl l;
    for (int k = 0; k < 3; k++) l.position[k] = ws[k] / ws[3];
    const uint8_t* f = texel(a.flux, a.cascade, x, y, 4);
    const uint8_t* n = texel(a.normals, a.cascade, x, y, 4);
    for (int k = 0; k < 3; k++) {
        l.color[k] = a.srgb_lut[f[k]];
        l.normal[k] = a.srgb_lut[256 + n[k]] * 2.0f - 1.0f;
    }
    return l;
}
SAH_DEV void position_to_grid_cell(const ExtractArgs& a, const float p[3], float cell[3]) {
    float cp[4];
    mat_vec4(a.world_to_cascade, p[0], p[1], p[2], 1.0f, cp);
    cell[0] = __builtin_rintf((cp[0] + a.cascade_f) * a.side);
    cell[1] = __builtin_rintf(cp[1] * a.side);
    cell[2] = __builtin_rintf(cp[2] * a.side);
}
SAH_DEV float clamp_snorm(float v) { return __builtin_fminf(__builtin_fmaxf(v, -1.0f), 1.0f); }

// one invocation of rsm_generate_vpls.comp: true when it stores a light
SAH_DEV bool extract_one(const ExtractArgs& a, uint32_t gx, uint32_t gy, sah_packed_vpl& out) {
    const int x0 = (int)gx * 2, y0 = (int)gy * 2;
    float brightest = 0.0f, chosen[3] = {0.0f, 0.0f, 0.0f};
    for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
            const Vpl v = load_rsm_vpl(a, x0 + x, y0 + y);
            const float luma = (v.color[0] * 0.2126f + v.color[1] * 0.7152f) + v.color[2] * 0.0722f;
            if (luma > brightest) {
                brightest = luma;
                position_to_grid_cell(a, v.position, chosen);
            }
        }
    float pos[3] = {0.f, 0.f, 0.f}, col[3] = {0.f, 0.f, 0.f}, nrm[3] = {0.f, 0.f, 0.f}, n = 0.0f;
    for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
            const Vpl v = load_rsm_vpl(a, x0 + x, y0 + y);
            float cell[3];
            position_to_grid_cell(a, v.position, cell);
            const float d0 = cell[0] - chosen[0], d1 = cell[1] - chosen[1], d2 = cell[2] - chosen[2];
            if ((d0 * d0 + d1 * d1) + d2 * d2 < 3.0f) {
                for (int k = 0; k < 3; k++) { pos[k] = pos[k] + v.position[k]; col[k] = col[k] + v.color[k]; nrm[k] = nrm[k] + v.normal[k]; }
                n = n + 1.0f;
            }
        }
    if (n > 0.0f) {
        for (int k = 0; k < 3; k++) { pos[k] = pos[k] / n; col[k] = col[k] / n; nrm[k] = nrm[k] / n; }
        const float inv = 1.0f / __builtin_sqrtf((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
        for (int k = 0; k < 3; k++) nrm[k] = nrm[k] * inv;
    }
    const float len_c = __builtin_sqrtf((col[0] * col[0] + col[1] * col[1]) + col[2] * col[2]);
    const float len_n = __builtin_sqrtf((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
    if (!(len_c > 0.0f && len_n > 0.0f)) return false;
    out.data[0] = (uint32_t)f2h(pos[0]) | ((uint32_t)f2h(pos[1]) << 16);
    out.data[1] = (uint32_t)f2h(pos[2]) | ((uint32_t)f2h(col[0]) << 16);
    out.data[2] = (uint32_t)f2h(col[1]) | ((uint32_t)f2h(col[2]) << 16);
    uint32_t w = 0;
    for (int k = 0; k < 3; k++) w |= ((uint32_t)((int)__builtin_rintf(clamp_snorm(nrm[k]) * 127.0f) & 0xff)) << (8 * k);
    out.data[3] = w;
    return true;
}

// Two launches: every invocation evaluates its 2x2 footprint in parallel and parks its light (and whether it stores one) in scratch;
// one workgroup then compacts the stored lights in invocation order (ballot + prefix over chunks of 1024).
__global__ __launch_bounds__(256) void k_extract_candidates(const ExtractArgs a) {
    const uint32_t half_res = a.res / 2, total = half_res * half_res;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    sah_packed_vpl vpl{};
    const bool keep = extract_one(a, i % half_res, i / half_res, vpl);
    a.candidates[i] = vpl;
    a.keep[i] = keep ? 1u : 0u;
}
__global__ __launch_bounds__(1024) void k_extract_compact(const ExtractArgs a) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t half_res = a.res / 2, total = half_res * half_res;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < total; base += 1024) {
        const uint32_t i = base + tid;
        const bool keep = i < total && a.keep[i] != 0u;
        const uint64_t mask = __ballot(keep);
        if (lane == 0) s_wave[wave] = (uint32_t)__builtin_popcountll(mask);
        __syncthreads();
        uint32_t slot = s_carry + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; w++) slot += s_wave[w];
        if (keep) a.list[slot] = a.candidates[i];
        __syncthreads();
        if (tid == 0) {
            uint32_t sum = 0;
            for (int w = 0; w < 16; w++) sum += s_wave[w];
            s_carry += sum;
        }
        __syncthreads();
    }
    if (tid == 0) *a.count = s_carry;
}

struct InjectArgs {
    const sah_packed_vpl* list;
    const uint32_t* count;
    uint32_t capacity;
    float world_to_cascade[16];
    float cascade_f, num_cascades_f;
    VolumeArg rgb[3];
    uint32_t* cells;  // scratch: cell index per light, ~0 when the light is dropped
    float* terms;     // scratch: 12 blend sources per light in sorted order (k_inject_sorted), 16-byte aligned
};

struct Injected {
    float sh[4];
    float corrected[3];
};
SAH_DEV float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
SAH_DEV float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
SAH_DEV float fractf(float x) { return x - __builtin_floorf(x); }
SAH_DEV float snorm8(uint32_t b) { return __builtin_fmaxf((float)(int8_t)(uint8_t)b / 127.0f, -1.0f); }

// vertex + fragment stage of one light; returns its cell (x + W (y + H z)) or ~0
SAH_DEV uint32_t inject_one(const InjectArgs& a, const sah_packed_vpl& p, Injected& out) {
    const float position[3] = {h2f((uint16_t)p.data[0]), h2f((uint16_t)(p.data[0] >> 16)), h2f((uint16_t)p.data[1])};
    const float color[3] = {h2f((uint16_t)(p.data[1] >> 16)), h2f((uint16_t)p.data[2]), h2f((uint16_t)(p.data[2] >> 16))};
    float normal[3] = {snorm8(p.data[3]), snorm8(p.data[3] >> 8), snorm8(p.data[3] >> 16)};
    const float inv = 1.0f / __builtin_sqrtf((normal[0] * normal[0] + normal[1] * normal[1]) + normal[2] * normal[2]);
    for (int k = 0; k < 3; k++) normal[k] = normal[k] * inv;
    float cp[4];
    mat_vec4(a.world_to_cascade, position[0], position[1], position[2], 1.0f, cp);
    const float px = (cp[0] + a.cascade_f) / a.num_cascades_f;
    const float ndc_x = px * 2.0f - 1.0f, ndc_y = cp[1] * 2.0f - 1.0f, layer_f = cp[2] * 32.0f;
    const float len_n = __builtin_sqrtf((normal[0] * normal[0] + normal[1] * normal[1]) + normal[2] * normal[2]);
    const float len_c = __builtin_sqrtf((color[0] * color[0] + color[1] * color[1]) + color[2] * color[2]);
    if (len_n < 1.0f || len_c == 0.0f) return ~0u;
    const float W = (float)a.rgb[0].width, H = (float)a.rgb[0].height, D = (float)a.rgb[0].depth;
    const float xf = ndc_x * (W * 0.5f) + W * 0.5f, yf = ndc_y * (H * 0.5f) + H * 0.5f;
    if (!(xf >= 0.0f && xf < W && yf >= 0.0f && yf < H)) return ~0u;
    if (!(layer_f > -1.0f && layer_f < D)) return ~0u;
    const uint32_t cx = (uint32_t)__builtin_floorf(xf), cy = (uint32_t)__builtin_floorf(yf), cz = (uint32_t)(int)layer_f;
    float scaled[3];
    for (int k = 0; k < 3; k++) scaled[k] = color[k] * 1024.0f / 16384.0f;
    const float Kx = 0.0f, Ky = -1.0f / 3.0f, Kz = 2.0f / 3.0f, Kw = -1.0f;
    const float s1 = stepf(scaled[2], scaled[1]);
    const float p4[4] = {mixf(scaled[2], scaled[1], s1), mixf(scaled[1], scaled[2], s1), mixf(Kw, Kx, s1), mixf(Kz, Ky, s1)};
    const float s2 = stepf(p4[0], scaled[0]);
    const float q4[4] = {mixf(p4[0], scaled[0], s2), mixf(p4[1], p4[1], s2), mixf(p4[3], p4[2], s2), mixf(scaled[0], p4[0], s2)};
    const float d = q4[0] - __builtin_fminf(q4[3], q4[1]);
    const float e = 1.0e-10f;
    float hsv[3] = {__builtin_fabsf(q4[2] + (q4[3] - q4[1]) / (6.0f * d + e)), d / (q4[0] + e), q4[0]};
    hsv[1] = hsv[1] * 2.0f;
    const float k4[4] = {1.0f, 2.0f / 3.0f, 1.0f / 3.0f, 3.0f};
    for (int k = 0; k < 3; k++) {
        const float pk = __builtin_fabsf(fractf(hsv[0] + k4[k]) * 6.0f - k4[3]);
        out.corrected[k] = hsv[2] * mixf(k4[0], __builtin_fminf(__builtin_fmaxf(pk - k4[0], 0.0f), 1.0f), hsv[1]);
    }
    const float c0 = 0.886226925f, c1 = 1.02332671f;
    out.sh[0] = c0; out.sh[1] = -c1 * normal[1]; out.sh[2] = c1 * normal[2]; out.sh[3] = -c1 * normal[0];
    return cx + a.rgb[0].width * (cy + a.rgb[0].height * cz);
}

// Two launches over the list (the capacity bounds the grid; the count is read on the device).  Pass 1: the cell of every light.
// Pass 2: the thread of the FIRST light of a cell — no earlier list entry has the same cell — walks the rest of the list and adds
// every light of that cell in list order.  Quadratic in the list length, which is a few thousand entries of an L2-resident array;
// every cell has exactly one writer, so there is nothing to synchronise.
__global__ __launch_bounds__(256) void k_inject_cells(const InjectArgs a) {
    const uint32_t count = min(*a.count, a.capacity);
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    Injected tmp;
    a.cells[i] = inject_one(a, a.list[i], tmp);
}
__global__ __launch_bounds__(256) void k_inject_accumulate(const InjectArgs a) {
    const uint32_t count = min(*a.count, a.capacity);
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t cell = a.cells[i];
    if (cell == ~0u) return;
    for (uint32_t j = 0; j < i; j++)
        if (a.cells[j] == cell) return;
    const uint32_t W = a.rgb[0].width, H = a.rgb[0].height;
    const uint32_t cx = cell % W, cy = (cell / W) % H, cz = cell / (W * H);
    float acc[3][4];
    uint16_t* dst[3];
    for (int ch = 0; ch < 3; ch++) {
        dst[ch] = (uint16_t*)(a.rgb[ch].ptr + (size_t)cz * a.rgb[ch].slice_pitch + (size_t)cy * a.rgb[ch].row_pitch + (size_t)cx * 8);
        for (int k = 0; k < 4; k++) acc[ch][k] = h2f(dst[ch][k]);
    }
    for (uint32_t j = i; j < count; j++) {
        if (a.cells[j] != cell) continue;
        Injected v;
        inject_one(a, a.list[j], v);
        for (int ch = 0; ch < 3; ch++)
            for (int k = 0; k < 4; k++) acc[ch][k] = rh(acc[ch][k] + v.sh[k] * v.corrected[ch] / 3.1415927f);  // blend ONE / ONE, one rounding to half
    }
    for (int ch = 0; ch < 3; ch++)
        for (int k = 0; k < 4; k++) dst[ch][k] = f2h(acc[ch][k]);
}

// Lists of up to 4096 lights (the default RSM resolution gives exactly that capacity): one workgroup sorts (cell, list index) keys
// in LDS with a bitonic network — equal cells become contiguous runs in list order — and the thread at the head of each run adds
// the run to its cell.  Same arithmetic and order as the two-launch form above, without its quadratic scans.
constexpr uint32_t kSortCapacity = 4096;
// Bitonic network over 4096 keys held four per thread (key r of thread t is element r * 1024 + t), so that only the exchanges between
// waves go through LDS: partners 1024 or 2048 apart are two registers of one thread, partners less than 64 apart are two lanes of one
// wave (a shuffle), and the distances in between (64 .. 512: 18 of the 78 stages) take one LDS round trip each, ping-ponging two
// buffers.  As plain LDS compare-exchanges with a barrier per stage the sort was 120 of the kernel's 165 us.
SAH_DEV unsigned long long exch(unsigned long long x, unsigned long long y, bool keep_min) { return ((x < y) == keep_min) ? x : y; }
__global__ __launch_bounds__(1024) void k_inject_sorted(const InjectArgs a) {
    __shared__ unsigned long long s_buf[2][kSortCapacity];
    const uint32_t count = min(min(*a.count, a.capacity), kSortCapacity);
    const uint32_t t = threadIdx.x;
    unsigned long long key[4];
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) {
        const uint32_t i = r * 1024u + t;
        uint32_t cell = ~0u;
        if (i < count) {
            Injected tmp;
            cell = inject_one(a, a.list[i], tmp);
        }
        key[r] = ((unsigned long long)cell << 32) | i;
    }
    uint32_t flip = 0;
    for (uint32_t k = 2; k <= kSortCapacity; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= 1024u) {  // rows of one thread: (0,1)(2,3) for j = 1024, (0,2)(1,3) for j = 2048
                const uint32_t d = j >> 10;
#pragma unroll
                for (uint32_t r = 0; r < 4; r++) {
                    if (r & d) continue;
                    const bool ascending = (((r * 1024u + t) & k) == 0u);
                    const unsigned long long x = key[r], y = key[r | d];
                    key[r] = exch(x, y, ascending);
                    key[r | d] = exch(y, x, !ascending);  // the other one (keys are distinct: the index is part of them)
                }
            } else if (j < 64u) {  // lanes of one wave
#pragma unroll
                for (uint32_t r = 0; r < 4; r++) {
                    const uint32_t i = r * 1024u + t;
                    const unsigned long long y = __shfl_xor(key[r], (int)j, 64);
                    const bool ascending = (i & k) == 0u, lower = (i & j) == 0u;
                    key[r] = exch(key[r], y, lower == ascending);
                }
            } else {  // other waves: one LDS round trip
                unsigned long long* buf = s_buf[flip];
                flip ^= 1u;
#pragma unroll
                for (uint32_t r = 0; r < 4; r++) buf[r * 1024u + t] = key[r];
                __syncthreads();
#pragma unroll
                for (uint32_t r = 0; r < 4; r++) {
                    const uint32_t i = r * 1024u + t;
                    const unsigned long long y = buf[i ^ j];
                    const bool ascending = (i & k) == 0u, lower = (i & j) == 0u;
                    key[r] = exch(key[r], y, lower == ascending);
                }
            }
        }
    }
    unsigned long long* s_key = s_buf[flip];
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) s_key[r * 1024u + t] = key[r];
    __syncthreads();
    // The twelve blend sources of every light (sh[k] * corrected[ch] / pi), evaluated in parallel and stored in SORTED order: the serial
    // part left to the thread at the head of a run is then one 48-byte record and twelve add-and-round per light, read in sequence.
    // (With the fragment stage re-evaluated inside the run loop the accumulation was 97 of the kernel's 130 us.)
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) {
        const uint32_t i = r * 1024u + t;
        if (i < count && (uint32_t)(s_key[i] >> 32) != ~0u) {
            Injected v;
            inject_one(a, a.list[(uint32_t)s_key[i]], v);
            float* rec = a.terms + (size_t)i * 12u;
            for (int ch = 0; ch < 3; ch++)
                for (int k = 0; k < 4; k++) rec[ch * 4 + k] = v.sh[k] * v.corrected[ch] / 3.1415927f;
        }
    }
    __syncthreads();
    const uint32_t W = a.rgb[0].width, H = a.rgb[0].height;
    for (uint32_t i = threadIdx.x; i < count; i += 1024) {
        const uint32_t cell = (uint32_t)(s_key[i] >> 32);
        if (cell == ~0u || (i > 0 && (uint32_t)(s_key[i - 1] >> 32) == cell)) continue;  // dropped light, or not the head of its run
        const uint32_t cx = cell % W, cy = (cell / W) % H, cz = cell / (W * H);
        float acc[12];
        uint16_t* dst[3];
        for (int ch = 0; ch < 3; ch++) {
            dst[ch] = (uint16_t*)(a.rgb[ch].ptr + (size_t)cz * a.rgb[ch].slice_pitch + (size_t)cy * a.rgb[ch].row_pitch + (size_t)cx * 8);
            for (int k = 0; k < 4; k++) acc[ch * 4 + k] = h2f(dst[ch][k]);
        }
        for (uint32_t j = i; j < count && (uint32_t)(s_key[j] >> 32) == cell; j++) {
            const float4* rec = reinterpret_cast<const float4*>(a.terms + (size_t)j * 12u);
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            const float src[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
            for (int c = 0; c < 12; c++) acc[c] = rh(acc[c] + src[c]);  // blend ONE / ONE, one rounding to half
        }
        for (int ch = 0; ch < 3; ch++)
            for (int k = 0; k < 4; k++) dst[ch][k] = f2h(acc[ch * 4 + k]);
    }
}

}  // namespace

hipError_t launch_extract_vpls(const VolumeArg& flux, const VolumeArg& normals, const VolumeArg& depth, const sah_lpv_cascade_matrices& c, uint32_t cascade,
                               float grid_cell_size, const float* luts, sah_packed_vpl* list, uint32_t* count, void* scratch, hipStream_t st) {
    ExtractArgs a{};
    a.flux = flux; a.normals = normals; a.depth = depth;
    for (int i = 0; i < 16; i++) { a.inverse_rsm_vp[i] = c.inverse_rsm_vp[i]; a.world_to_cascade[i] = c.world_to_cascade[i]; }
    a.cascade = cascade;
    a.res = depth.width;
    a.cascade_f = (float)cascade;
    a.side = grid_cell_size * 32.0f;
    a.srgb_lut = luts;
    a.list = list;
    a.count = count;
    const uint32_t total = (a.res / 2) * (a.res / 2);
    a.candidates = (sah_packed_vpl*)scratch;  // total * 16 bytes, then total * 4 bytes of flags
    a.keep = (uint32_t*)(a.candidates + total);
    hipLaunchKernelGGL(k_extract_candidates, dim3((total + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_extract_compact, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_inject_vpls(const sah_packed_vpl* list, const uint32_t* count, uint32_t capacity, const sah_lpv_cascade_matrices& c, uint32_t cascade,
                              uint32_t num_cascades, const VolumeArg rgb[3], uint32_t* cells_scratch, hipStream_t st) {
    InjectArgs a{};
    a.list = list; a.count = count; a.capacity = capacity;
    for (int i = 0; i < 16; i++) a.world_to_cascade[i] = c.world_to_cascade[i];
    a.cascade_f = (float)cascade;
    a.num_cascades_f = (float)num_cascades;
    for (int i = 0; i < 3; i++) a.rgb[i] = rgb[i];
    a.cells = cells_scratch;
    a.terms = reinterpret_cast<float*>(cells_scratch + (((size_t)capacity + 1 + 3) & ~(size_t)3));  // 16-byte aligned, kSortCapacity * 12 floats
    if (capacity == 0) return hipSuccess;
    if (capacity <= kSortCapacity) {
        hipLaunchKernelGGL(k_inject_sorted, dim3(1), dim3(1024), 0, st, a);
        return hipGetLastError();
    }
    const dim3 grid((capacity + 255) / 256);
    hipLaunchKernelGGL(k_inject_cells, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_inject_accumulate, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace sah
