// Irradiance-cache probe maintenance for gfx950:
//   copy     RenderCore/shaders/gi/cache/copy_cascades.comp.slang:22-99        (host render/gi/irradiance_cache.cpp:455-486)
//   update   RenderCore/shaders/gi/cache/probe_depth_update.comp.slang:11-49, probe_light_cache_update.comp.slang:13-53,
//            probe_rtgi_update.comp.slang:13-53, probe_finalize.comp.slang:13-74, probe_update.slangi:4-37,
//            common/octahedral.slangi:18-54                                   (host irradiance_cache.cpp:585-724)
// The work is tiny (<= 1024 probes per frame, 32^3 probe cells) and launch bound; what matters is that the result is a function
// of the input.  The reference shaders are not (include/sah_hip.h): stores of one dispatch collide.  The order
// fixed by the ABI — ascending linear invocation index, program order inside an invocation — is implemented without serialising
// the arithmetic: the copy is written as a gather per destination texel (at most two candidate writers, the later one wins), and
// the updates compute every texel value in parallel, park it in LDS and let one thread replay the <= 4 stores per texel in order.
#include <hip/hip_runtime.h>

#include "../../include/sah_hip.h"
#include "octahedral.hpp"
#include "numerics.hpp"
#include "params.hpp"

namespace sah {

// ---- B10G11R11 <-> fp16 ---------------------------------------------------------------------------------------------------
// decode: uf11 = fp16 >> 4, uf10 = fp16 >> 5 (same exponent width and bias).  encode (float -> uf11/uf10): round toward zero,
// negatives -> 0, NaN -> canonical NaN, values above the largest finite -> largest finite (the choice documented in DESIGN.md §3).
SAH_DEV uint32_t f32_to_uf(float f, uint32_t mant_bits) {  // mant_bits = 6 (uf11) or 5 (uf10)
    const uint32_t x = __float_as_uint(f);
    const uint32_t exp_all = 0x1fu << mant_bits;
    if ((x & 0x7fffffffu) > 0x7f800000u) return exp_all | (1u << (mant_bits - 1));  // NaN
    if (x & 0x80000000u) return 0u;                                                // negative (incl. -inf, -0)
    if (x >= 0x7f800000u) return exp_all;                                          // +inf
    const uint32_t max_finite = exp_all - 1u;                                      // 0x7bf / 0x3df
    const uint32_t max_f32 = ((30u + 112u) << 23) | (((1u << mant_bits) - 1u) << (23u - mant_bits));
    if (x > max_f32) return max_finite;
    if (x < 0x38800000u) {  // below 2^-14: denormal in the small format, unit 2^-(14 + mant_bits)
        const uint32_t e = x >> 23;
        const uint32_t sh = (mant_bits == 6u ? 130u : 131u) - e;  // value = m * 2^(e - 150), unit 2^-20 (uf11) / 2^-19 (uf10)
        if (sh > 24u) return 0u;
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        return m >> sh;
    }
    return (((x >> 23) - 112u) << mant_bits) | ((x & 0x7fffffu) >> (23u - mant_bits));
}
SAH_DEV uint32_t encode_r11g11b10(Hn r, Hn g, Hn b) { return f32_to_uf(tof(r), 6u) | (f32_to_uf(tof(g), 6u) << 11) | (f32_to_uf(tof(b), 5u) << 22); }
SAH_DEV void decode_r11g11b10(uint32_t w, Hn (&o)[3]) {
    o[0] = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)((w & 0x7ffu) << 4)));
    o[1] = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(((w >> 11) & 0x7ffu) << 4)));
    o[2] = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(((w >> 22) & 0x3ffu) << 5)));
}
// a texel that goes through `half3 t = src[..]; dst[..] = t;`: every bit pattern survives except NaNs, which become the canonical one
SAH_DEV uint32_t roundtrip_r11g11b10(uint32_t w) {
    Hn c[3];
    decode_r11g11b10(w, c);
    return encode_r11g11b10(c[0], c[1], c[2]);
}
// half -> R8_UNORM store: clamp, * 255 + 0.5 in fp32, truncate (NaN -> 0)
SAH_DEV uint8_t half_to_unorm8(Hn v) {
    const float c = tof(v);
    if (!(c > 0.0f)) return 0;
    if (c >= 1.0f) return 255;
    return (uint8_t)(c * 255.0f + 0.5f);
}

SAH_DEV bool in_vol(const VolumeArg& v, int x, int y, int z) {
    return x >= 0 && y >= 0 && z >= 0 && (uint32_t)x < v.width && (uint32_t)y < v.height && (uint32_t)z < v.depth;
}
SAH_DEV uint8_t* vol_ptr(const VolumeArg& v, int x, int y, int z, uint32_t bpp) {
    return const_cast<uint8_t*>(v.ptr) + (size_t)z * v.slice_pitch + (size_t)y * v.row_pitch + (size_t)x * bpp;
}
SAH_DEV uint32_t load_word_or_zero(const VolumeArg& v, int x, int y, int z) {
    return in_vol(v, x, y, z) ? *reinterpret_cast<const uint32_t*>(vol_ptr(v, x, y, z, 4)) : 0u;
}

// ---- copy_cascades --------------------------------------------------------------------------------------------------------
struct CopyArgs {
    ProbeAtlasArgs src, dst;
    int move[4][3];  // (int3)cascade_movement[c]
};
// source cell of probe cell (x, y, z); false: the cell scrolls in and is initialised (copy_cascades.comp.slang:89-98)
SAH_DEV bool copy_source(const CopyArgs& a, int x, int y, int z, int (&s)[3]) {
    const int cascade = y / 8;
    s[0] = x - a.move[cascade][0];
    s[1] = y - a.move[cascade][1];
    s[2] = z - a.move[cascade][2];
    return s[0] >= 0 && s[1] >= 8 * cascade && s[2] >= 0 && s[0] < 32 && s[1] < 8 * (cascade + 1) && s[2] < 32;
}

// B10G11R11 atlases with BW x BH texel blocks (rtgi 7x8, light cache 13x13, average 1x1): one thread per destination texel of the
// 32 x 32 x 32 probe grid; every texel has exactly one writer (its own cell)
template <int BW, int BH> __global__ void __launch_bounds__(256) k_probe_copy_r11(CopyArgs a, VolumeArg src, VolumeArg dst) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    constexpr uint32_t W = 32u * BW, H = 32u * BH;
    if (idx >= W * H * 32u) return;
    const int X = (int)(idx % W), Y = (int)((idx / W) % H), L = (int)(idx / (W * H));
    const int cx = X / BW, cy = Y / BH;
    int s[3];
    uint32_t word = 0u;  // init_new_probe: half3(0)
    if (copy_source(a, cx, cy, L, s)) word = roundtrip_r11g11b10(load_word_or_zero(src, s[0] * BW + X % BW, s[1] * BH + Y % BH, s[2]));
    if (in_vol(dst, X, Y, L)) *reinterpret_cast<uint32_t*>(vol_ptr(dst, X, Y, L, 4)) = word;
}
__global__ void __launch_bounds__(256) k_probe_copy_validity(CopyArgs a) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= 32768u) return;
    const int X = (int)(idx & 31u), Y = (int)((idx >> 5) & 31u), L = (int)(idx >> 10);
    int s[3];
    uint8_t out = 255;  // validity_dest[index] = 0xff saturates to 1.0
    if (copy_source(a, X, Y, L, s)) {
        const uint8_t b = in_vol(a.src.validity, s[0], s[1], s[2]) ? *vol_ptr(a.src.validity, s[0], s[1], s[2], 1) : (uint8_t)0;
        out = half_to_unorm8(Hn((float)b / 255.0f));  // Texture2DArray<half> load, RWTexture2DArray<half> store
    }
    if (in_vol(a.dst.validity, X, Y, L)) *vol_ptr(a.dst.validity, X, Y, L, 1) = out;
}
// Depth atlas (12 x 12 blocks): a texel can be written by the copy of its own cell (if that cell copies) and by the misplaced
// clear of init_new_probe, which zeroes 12 x 12 texels at LIGHT-CACHE offsets (13 * cell).  Later linear invocation index wins;
// a texel nobody writes keeps its previous contents.
__global__ void __launch_bounds__(256) k_probe_copy_depth(CopyArgs a) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    constexpr uint32_t W = 32u * 12u, H = 32u * 12u;
    if (idx >= W * H * 32u) return;
    const int X = (int)(idx % W), Y = (int)((idx / W) % H), L = (int)(idx / (W * H));
    if (!in_vol(a.dst.depth, X, Y, L)) return;
    int s[3];
    // candidate A: copy_from_cell of cell (X / 12, Y / 12, L)
    const int ax = X / 12, ay = Y / 12;
    const bool a_writes = copy_source(a, ax, ay, L, s);
    // candidate B: init_new_probe of cell (X / 13, Y / 13, L), covering offsets 0..11 of its 13-texel pitch
    const int bx = X / 13, by = Y / 13;
    int sb[3];
    const bool b_writes = bx < 32 && by < 32 && X % 13 < 12 && Y % 13 < 12 && !copy_source(a, bx, by, L, sb);
    if (!a_writes && !b_writes) return;
    const int a_order = ay * 32 + ax, b_order = by * 32 + bx;  // same layer: compare (y, x)
    uint32_t word = 0u;
    if (a_writes && (!b_writes || a_order > b_order)) word = load_word_or_zero(a.src.depth, s[0] * 12 + X % 12, s[1] * 12 + Y % 12, s[2]);
    *reinterpret_cast<uint32_t*>(vol_ptr(a.dst.depth, X, Y, L, 4)) = word;
}

// ---- probe updates --------------------------------------------------------------------------------------------------------
struct UpdateArgs {
    ProbeAtlasArgs atl;
    VolumeArg trace;          // RGBA16F 20 x 20 x P
    const uint32_t* probes;   // P x uint3
    uint32_t num_probes;
    const uint32_t* slots;    // 32^3: list position + 1 of the listed probes (ordered_stores)
};
SAH_DEV void load_trace(const VolumeArg& t, int x, int y, int p, Hn (&o)[4]) {
    uint2 q = make_uint2(0u, 0u);  // out-of-range loads return 0
    if (in_vol(t, x, y, p)) q = *reinterpret_cast<const uint2*>(vol_ptr(t, x, y, p, 8));
    o[0] = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.x & 0xffffu)));
    o[1] = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.x >> 16)));
    o[2] = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.y & 0xffffu)));
    o[3] = Hn::raw(__builtin_bit_cast(_Float16, (uint16_t)(q.y >> 16)));
}
SAH_DEV int isign(int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }

// probe_update.slangi:4-37 for texel (tx, ty): up to four destination cells, in program order.  Returns the count.
SAH_DEV int border_targets(int rx, int ry, int tx, int ty, int (&ox)[4], int (&oy)[4]) {
    const bool edge_x = tx == 0 || tx == rx - 1, edge_y = ty == 0 || ty == ry - 1;
    int mx = tx - rx / 2, my = ty - ry / 2;
    mx += mx >= 0 ? 1 : 0;
    my += my >= 0 ? 1 : 0;
    int n = 0;
    ox[n] = tx;
    oy[n++] = ty;  // no +1: the interior lands on cells [0, res) of the (res + 2)-wide block
    if (edge_x && edge_y) {
        int dx = -mx, dy = -my;
        dx += mx >= 0 ? -1 : 0;
        dy += my >= 0 ? -1 : 0;
        ox[n] = dx + rx / 2;
        oy[n++] = dy + ry / 2;
    }
    if (edge_x) {
        int ex = mx + isign(mx), ey = -my;
        ex += ex >= 0 ? -1 : 0;
        ey += ey >= 0 ? -1 : 0;
        ox[n] = ex + rx / 2;
        oy[n++] = ey + ry / 2;
    }
    if (edge_y) {
        int ex = -mx, ey = my + isign(my);
        ex += ex >= 0 ? -1 : 0;
        ey += ey >= 0 ? -1 : 0;
        ox[n] = ex + rx / 2;
        oy[n++] = ey + ry / 2;
    }
    return n;
}
// The stores of a dispatch take effect in ascending linear invocation index — workgroup (list position of the probe) first, then
// ty * RX + tx, program order inside an invocation — so the LAST store to a cell is the one that stays.
//   inside a workgroup: every invocation announces its (up to four) stores with an LDS atomicMax of the order key invocation * 4 + k
//     on the cell and, after a barrier, performs only the ones whose key came out on top (one thread used to replay all ~300 stores in
//     order: 20 us per dispatch);
//   between workgroups: the cells of a probe's (R + 2)-wide block reach from -2 to R for the odd sizes (5, 11), so the block of a
//     neighbouring probe writes some of the same cells.  `slots` maps a probe cell to its position in the list + 1 (0: not listed;
//     filled by k_probe_slots before the updates, cleared by it after them); a store is dropped when a neighbour that is listed LATER
//     writes the same cell — it does if its own pattern, which is this workgroup's pattern shifted by one block, covers the cell.
// Same image as replaying every store of the dispatch in order.
constexpr int kProbeGrid = 32;  // probe cells per axis (validity atlas extent)
template <int RX, int RY>
SAH_DEV void ordered_stores(const VolumeArg& dst, const uint32_t* id, uint32_t list_pos, const uint32_t* slots, bool active, uint32_t value,
                            uint32_t* s_owner) {
    constexpr int kW = RX + 3, kH = RY + 3, kCells = kW * kH;  // relative cells -2 .. R
    for (int c = threadIdx.x; c < kCells; c += blockDim.x) s_owner[c] = 0u;
    __syncthreads();
    int ox[4], oy[4], n = 0;
    const int t = (int)threadIdx.x;
    if (active) {
        n = border_targets(RX, RY, t % RX, t / RX, ox, oy);
        for (int k = 0; k < n; k++) atomicMax(&s_owner[(oy[k] + 2) * kW + (ox[k] + 2)], (uint32_t)(t * 4 + k) + 1u);
    }
    __syncthreads();
    const int px = (int)id[0], py = (int)id[1], bz = (int)id[2];
    const int bx = px * (RX + 2), by = py * (RY + 2);
    if (px >= 0 && py >= 0 && bz >= 0 && px < kProbeGrid && py < kProbeGrid && bz < kProbeGrid &&
        slots[(bz * kProbeGrid + py) * kProbeGrid + px] != list_pos + 1u)
        return;  // listed again later: that workgroup's stores are the ones that stay (uniform over the workgroup, after its last barrier)
    for (int k = 0; k < n; k++) {
        if (s_owner[(oy[k] + 2) * kW + (ox[k] + 2)] != (uint32_t)(t * 4 + k) + 1u) continue;  // a later invocation of this probe stores there
        const int x = ox[k] + bx, y = oy[k] + by;
        bool beaten = false;
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int qx = px + dx, qy = py + dy;
                if ((dx == 0 && dy == 0) || qx < 0 || qy < 0 || qx >= kProbeGrid || qy >= kProbeGrid || bz < 0 || bz >= kProbeGrid) continue;
                const int rx = x - qx * (RX + 2), ry = y - qy * (RY + 2);  // the cell as the neighbour's block sees it
                if (rx < -2 || rx > RX || ry < -2 || ry > RY || s_owner[(ry + 2) * kW + (rx + 2)] == 0u) continue;  // the neighbour never writes it
                const uint32_t slot = slots[(bz * kProbeGrid + qy) * kProbeGrid + qx];
                beaten = beaten || (slot != 0u && slot - 1u > list_pos);
            }
        if (!beaten && in_vol(dst, x, y, bz)) *reinterpret_cast<uint32_t*>(vol_ptr(dst, x, y, bz, 4)) = value;
    }
}
// list position + 1 of every probe of the list into `slots` (set), or 0 again (clear)
__global__ void __launch_bounds__(256) k_probe_slots(const uint32_t* probes, uint32_t num_probes, uint32_t* slots, uint32_t set) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= num_probes) return;
    const uint32_t x = probes[3u * i], y = probes[3u * i + 1u], z = probes[3u * i + 2u];
    if (x < (uint32_t)kProbeGrid && y < (uint32_t)kProbeGrid && z < (uint32_t)kProbeGrid) {
        // a probe listed twice (the ABI asks for distinct probes; cheap to survive): the LATER listing owns the cell — under the store order
        // above it overwrites everything the earlier one stored — and the earlier workgroup stores nothing (ordered_stores)
        if (set) atomicMax(&slots[(z * kProbeGrid + y) * kProbeGrid + x], i + 1u);
        else slots[(z * kProbeGrid + y) * kProbeGrid + x] = 0u;
    }
}

// probe_depth_update.comp.slang:11-49 — one workgroup per probe, 10 x 10 texels
__global__ void __launch_bounds__(128) k_probe_depth_update(UpdateArgs a) {
    __shared__ uint32_t s_owner[13 * 13];
    const uint32_t p = blockIdx.x;
    const uint32_t* id = a.probes + 3u * p;
    uint32_t value = 0u;
    if (threadIdx.x < 100) {
        const int tx = threadIdx.x % 10, ty = threadIdx.x / 10;
        Hn depth = Hn::lit(0.f), n = Hn::lit(0.f);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            Hn t[4];
            load_trace(a.trace, tx * 2 + (i % 2), ty * 2 + (i / 2), (int)p, t);
            if (tof(t[3]) > 0.f) {
                depth = depth + t[3];  // "* weight" is commented out in the shader
                n = n + Hn::lit(1.f);
            }
        }
        depth = tof(n) > 0.f ? depth / n : Hn::lit(0.f);
        const Hn d2 = depth * depth;
        value = (uint32_t)__builtin_bit_cast(uint16_t, depth.v) | ((uint32_t)__builtin_bit_cast(uint16_t, d2.v) << 16);
    }
    ordered_stores<10, 10>(a.atl.depth, id, p, a.slots, threadIdx.x < 100, value, s_owner);
}

// probe_light_cache_update.comp.slang:13-53 — 11 x 11 texels, 2 x 2 trace texels each (filter = ceil(20 / 11))
__global__ void __launch_bounds__(128) k_probe_light_cache_update(UpdateArgs a) {
    __shared__ uint32_t s_owner[14 * 14];
    const uint32_t p = blockIdx.x;
    const uint32_t* id = a.probes + 3u * p;
    uint32_t value = 0u;
    if (threadIdx.x < 121) {
        const uint32_t tx = threadIdx.x % 11, ty = threadIdx.x / 11;
        const H3 direction = to_h(octahedral_direction(normalized_octahedral_coordinates(tx, ty, 11, 11)));
        const uint32_t filter = 2u;
        const uint32_t bx = (uint32_t)__builtin_floorf((float)tx * (float)filter), by = (uint32_t)__builtin_floorf((float)ty * (float)filter);
        H3 light = {Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f)};
        Hn n = Hn::lit(0.f);
        for (uint32_t i = 0; i < filter * filter; i++) {
            const uint32_t rx = bx + i % filter, ry = by + i / filter;
            Hn t[4];
            load_trace(a.trace, (int)rx, (int)ry, (int)p, t);
            if (tof(t[3]) > 0.f) {
                const F3 ray_dir = octahedral_direction(normalized_octahedral_coordinates(rx, ry, 20, 20));
                const Hn weight = Hn(dot(to_f(direction), ray_dir).v);  // dot(half3, float3) evaluates in float
                light = light + H3{t[0], t[1], t[2]} * weight;
                n = n + Hn::lit(1.f);
            }
        }
        if (tof(n) > 0.f) light = light / n;
        else light = {Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f)};
        value = encode_r11g11b10(light.x, light.y, light.z);
    }
    ordered_stores<11, 11>(a.atl.light_cache, id, p, a.slots, threadIdx.x < 121, value, s_owner);
}

// probe_rtgi_update.comp.slang:13-53 — 5 x 6 texels, 4 x 4 trace texels each (filter = 20 / 5)
__global__ void __launch_bounds__(64) k_probe_rtgi_update(UpdateArgs a) {
    __shared__ uint32_t s_owner[8 * 9];
    const uint32_t p = blockIdx.x;
    const uint32_t* id = a.probes + 3u * p;
    uint32_t value = 0u;
    if (threadIdx.x < 30) {
        const uint32_t tx = threadIdx.x % 5, ty = threadIdx.x / 5;
        H3 light = {Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f)};
        Hn n = Hn::lit(0.f);
        for (uint32_t i = 0; i < 16; i++) {
            Hn t[4];
            load_trace(a.trace, (int)(tx * 4u + i % 4u), (int)(ty * 4u + i / 4u), (int)p, t);
            if (tof(t[3]) > 0.f) {
                light = light + H3{t[0], t[1], t[2]};  // the cosine weight is computed but not used
                n = n + Hn::lit(1.f);
            }
        }
        if (tof(n) > 0.f) light = light / n;
        else light = {Hn::lit(0.f), Hn::lit(0.f), Hn::lit(0.f)};
        value = encode_r11g11b10(light.x, light.y, light.z);
    }
    ordered_stores<5, 6>(a.atl.rtgi, id, p, a.slots, threadIdx.x < 30, value, s_owner);
}

// probe_finalize.comp.slang:13-74 — validity from depth texels 0 and 64 (every lane tests the same two), average of 30 rtgi texels
__global__ void __launch_bounds__(64) k_probe_finalize(UpdateArgs a) {
    __shared__ uint32_t s_rtgi[30];
    const uint32_t p = blockIdx.x;
    const uint32_t* id = a.probes + 3u * p;
    const int px = (int)id[0], py = (int)id[1], pz = (int)id[2];
    if (threadIdx.x < 30) {
        const int x = threadIdx.x % 5, y = threadIdx.x / 6;
        s_rtgi[threadIdx.x] = load_word_or_zero(a.atl.rtgi, px * 7 + x + 1, py * 8 + y + 1, pz);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    uint32_t num_valid = 0;
    for (uint32_t idx = 0; idx < 100; idx += 64) {
        const int x = (int)(idx % 10), y = (int)(idx / 10);
        const uint32_t w = load_word_or_zero(a.atl.depth, px * 12 + x + 1, py * 12 + y + 1, pz);
        if ((float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)) > 0.f) num_valid += 64;
    }
    if (in_vol(a.atl.validity, px, py, pz)) *vol_ptr(a.atl.validity, px, py, pz, 1) = half_to_unorm8(Hn((float)num_valid) / Hn::lit(100.f));
    H3 sum;
    for (int lane = 0; lane < 30; lane++) {  // WaveActiveSum over the 30 active lanes: fp16, lane order (ABI definition)
        Hn c[3];
        decode_r11g11b10(s_rtgi[lane], c);
        const H3 v = {c[0], c[1], c[2]};
        sum = lane == 0 ? v : sum + v;
    }
    const H3 avg = sum / Hn::lit(30.f);
    if (in_vol(a.atl.average, px, py, pz)) *reinterpret_cast<uint32_t*>(vol_ptr(a.atl.average, px, py, pz, 4)) = encode_r11g11b10(avg.x, avg.y, avg.z);
}

// ---- launchers ------------------------------------------------------------------------------------------------------------
hipError_t launch_probe_copy(const ProbeAtlasArgs& src, const ProbeAtlasArgs& dst, const float movement[4][3], hipStream_t st) {
    CopyArgs a;
    a.src = src;
    a.dst = dst;
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 3; i++) {
            const float m = movement[c][i];  // (int3)float3: truncation; out-of-range / NaN movements scroll everything out
            a.move[c][i] = (m >= -64.f && m <= 64.f) ? (int)m : 64;
        }
    auto blocks = [](uint32_t n) { return dim3((n + 255u) / 256u); };
    hipLaunchKernelGGL((k_probe_copy_r11<7, 8>), blocks(224u * 256u * 32u), dim3(256), 0, st, a, src.rtgi, dst.rtgi);
    hipLaunchKernelGGL((k_probe_copy_r11<13, 13>), blocks(416u * 416u * 32u), dim3(256), 0, st, a, src.light_cache, dst.light_cache);
    hipLaunchKernelGGL((k_probe_copy_r11<1, 1>), blocks(32768u), dim3(256), 0, st, a, src.average, dst.average);
    hipLaunchKernelGGL(k_probe_copy_validity, blocks(32768u), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_probe_copy_depth, blocks(384u * 384u * 32u), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_probe_update(const ProbeAtlasArgs& atl, const VolumeArg& trace, const uint32_t* probes, uint32_t num_probes, uint32_t* slots,
                               hipStream_t st) {
    if (num_probes == 0) return hipSuccess;
    UpdateArgs a;
    a.atl = atl;
    a.trace = trace;
    a.probes = probes;
    a.num_probes = num_probes;
    a.slots = slots;
    hipLaunchKernelGGL(k_probe_slots, dim3((num_probes + 255) / 256), dim3(256), 0, st, probes, num_probes, slots, 1u);
    hipLaunchKernelGGL(k_probe_depth_update, dim3(num_probes), dim3(128), 0, st, a);
    hipLaunchKernelGGL(k_probe_light_cache_update, dim3(num_probes), dim3(128), 0, st, a);
    hipLaunchKernelGGL(k_probe_rtgi_update, dim3(num_probes), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_probe_finalize, dim3(num_probes), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_probe_slots, dim3((num_probes + 255) / 256), dim3(256), 0, st, probes, num_probes, slots, 0u);  // all zero again
    return hipGetLastError();
}

}  // namespace sah
