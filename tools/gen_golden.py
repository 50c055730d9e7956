"""Independent numpy restatement of the reference shaders on the hot path, used ONLY to generate the golden images under
tests/golden/ (SURVEY.md §7 step 1, §8-c fixture ii).  It is a second implementation — vectorised numpy, written from
the shader text (file:line cited per function), not from oracle/ — so that a misreading would have to be made twice to
go unnoticed.  The reference itself cannot run here and ships no golden images (parity unpinned).

    python tools/gen_golden.py          # rewrites tests/golden/*.npz

Arithmetic model: np.float32 per-operator rounding (numpy never fuses), fp16 steps via astype(float16), fp64 only where
the contract says so (pow5 product chain, libm transcendentals), and fma(a,b,c) emulated as float32(float64(a)*b + c)
(the product is exact in fp64; the one extra rounding is below 2^-29 relative and irrelevant for these image sizes).
"""
import hashlib
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from androidrenderer_amd import _abi, images, synth  # noqa: E402

f32 = np.float32
GOLDEN = os.path.join(ROOT, "tests", "golden")


def F(x):
    return np.asarray(x, dtype=f32)


def h(x):  # round to fp16, keep as fp32
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=f32).astype(np.float16).astype(f32)


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def srgb_lut():
    c = np.arange(256, dtype=np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(f32)


def dot3(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def normalize3(v, rnd=F):
    d = rnd(rnd(rnd(v[0] * v[0]) + rnd(v[1] * v[1])) + rnd(v[2] * v[2]))
    inv = rnd(f32(1.0) / rnd(np.sqrt(d)))
    return [rnd(v[0] * inv), rnd(v[1] * inv), rnd(v[2] * inv)]


def clamp01(x):
    return np.minimum(np.maximum(x, f32(0)), f32(1))  # fmax/fmin semantics are irrelevant here: inputs are finite


def pow5(x, rnd=F):
    d = np.asarray(x, np.float64)
    return rnd((d * d * d * d * d).astype(f32))


def mat_vec(m, v):  # column-major flat[16], left-to-right sum (GLSL M * v)
    return [F(F(F(m[0 + i] * v[0] + m[4 + i] * v[1]) + m[8 + i] * v[2]) + m[12 + i] * v[3]) for i in range(4)]


# ---- brdf.glsl:29-121 (rnd = F) / brdf.slangi:22-114 (rnd = h) ------------------------------------------------------
def brdf(base, n, rough, metal, l, v, rnd=F):
    r = rnd
    one, pi = r(f32(1.0)), r(f32(3.1415927))
    f0d = r(f32(0.04))
    f0 = [r(r(f0d * r(one - metal)) + r(base[i] * metal)) for i in range(3)]                       # mix(x,y,a) = x(1-a) + ya
    diff = [r(r(base[i] * r(one - f0d)) * r(one - metal)) for i in range(3)]
    hv = normalize3([r(v[i] + l[i]) for i in range(3)], r)
    dn = lambda a, b: r(r(r(a[0] * b[0]) + r(a[1] * b[1])) + r(a[2] * b[2]))
    NoV = r(dn(n, v) + r(f32(1e-5)))
    NoL = dn(n, l)
    NoH = clamp01(dn(n, hv))
    VoH = clamp01(dn(v, hv))
    dark = NoL <= 0
    NoV = np.abs(NoV)
    NoL = clamp01(NoL)
    LoH = clamp01(dn(l, hv))
    # Fd_Burley :46-52
    f90 = r(r(f32(0.5)) + r(r(r(r(f32(2.0)) * rough) * LoH) * LoH))
    schlick1 = lambda u: r(one + r(r(f90 - one) * pow5(clamp01(r(one - u)), r)))
    fdv = r(r(schlick1(NoL) * schlick1(NoV)) * r(one / pi))
    fd = [r(diff[i] * fdv) for i in range(3)]
    # D_GGX :29-32, V_SmithGGXCorrelated :36-42, F_Schlick :34
    k = r(rough / r(r(one - r(NoH * NoH)) + r(rough * rough)))
    D = r(r(k * k) * r(one / pi))
    a2 = r(rough * rough)
    GGXL = r(NoV * r(np.sqrt(r(r(r(r(r(-NoL) * a2) + NoL) * NoL) + a2))))
    GGXV = r(NoL * r(np.sqrt(r(r(r(r(r(-NoV) * a2) + NoV) * NoV) + a2))))
    Vis = r(r(f32(0.5)) / r(GGXV + GGXL))
    p = pow5(clamp01(r(one - VoH)), r)
    Fv = [r(f0[i] + r(r(one - f0[i]) * p)) for i in range(3)]
    DV = r(D * Vis)
    out = [r(fd[i] + r(DV * Fv[i])) for i in range(3)]
    return [np.where(dark, f32(0), o) for o in out]


# ---- samplers (Vulkan weighted sum, fma chain from +0) -------------------------------------------------------------------
def axis(coord, size):
    p = F(F(coord * f32(size)) - f32(0.5))
    f0 = np.floor(p)
    fr = F(p - f0)
    return f0.astype(np.int64), F(f32(1.0) - fr), fr


def bilinear_clamp(img, u, v):
    """img (H, W, C) fp32; CLAMP_TO_EDGE."""
    H, W = img.shape[:2]
    x0, wx0, fx = axis(u, W)
    y0, wy0, fy = axis(v, H)
    xa, xb = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    ya, yb = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    acc = np.zeros(u.shape + (img.shape[2],), dtype=f32)
    for (yy, xx, w) in ((ya, xa, F(wx0 * wy0)), (ya, xb, F(fx * wy0)), (yb, xa, F(wx0 * fy)), (yb, xb, F(fx * fy))):
        acc = fma(w[..., None], img[yy, xx], acc)
    return acc


def trilinear_border(vol, u, v, w):
    """vol (D, H, W, 4) fp32; CLAMP_TO_BORDER transparent black."""
    D, H, W = vol.shape[:3]
    x0, wx0, fx = axis(u, W)
    y0, wy0, fy = axis(v, H)
    z0, wz0, fz = axis(w, D)
    acc = np.zeros(u.shape + (4,), dtype=f32)
    wxy = [F(wx0 * wy0), F(fx * wy0), F(wx0 * fy), F(fx * fy)]
    for k in range(8):
        xx, yy, zz = x0 + (k & 1), y0 + ((k >> 1) & 1), z0 + (k >> 2)
        wt = F(wxy[k & 3] * (fz if (k >> 2) else wz0))
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H) & (zz >= 0) & (zz < D)
        t = np.where(ok[..., None], vol[np.clip(zz, 0, D - 1), np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], f32(0))
        acc = fma(wt[..., None], t, acc)
    return acc


def bilinear_repeat(img, u, v, layer=None):
    """img (H, W, C) or (L, H, W, C) fp32; REPEAT addressing; weights and fma chain as bilinear_clamp."""
    H, W = img.shape[-3], img.shape[-2]
    x0, wx0, fx = axis(u, W)
    y0, wy0, fy = axis(v, H)
    acc = np.zeros(u.shape + (img.shape[-1],), dtype=f32)
    for (dy, dx, w) in ((0, 0, F(wx0 * wy0)), (0, 1, F(fx * wy0)), (1, 0, F(wx0 * fy)), (1, 1, F(fx * fy))):
        yy, xx = np.mod(y0 + dy, H), np.mod(x0 + dx, W)
        t = img[yy, xx] if layer is None else img[layer, yy, xx]
        acc = fma(w[..., None], t, acc)
    return acc


def unpack_b10g11r11(words):
    """B10G11R11_UFLOAT_PACK32: r = bits 0-10 (5 exponent, 6 mantissa), g = bits 11-21, b = bits 22-31 (5 exponent, 5 mantissa);
    unsigned small floats with the fp16 exponent bias, so each is an fp16 bit pattern with the low mantissa bits zero."""
    w = words.astype(np.uint32)
    r = ((w & 0x7FF) << 4).astype(np.uint16).view(np.float16)
    g = (((w >> 11) & 0x7FF) << 4).astype(np.uint16).view(np.float16)
    b = (((w >> 22) & 0x3FF) << 5).astype(np.uint16).view(np.float16)
    return np.stack([r, g, b], axis=-1).astype(f32)


def length3(v, rnd=F):
    return rnd(np.sqrt(rnd(rnd(rnd(v[0] * v[0]) + rnd(v[1] * v[1])) + rnd(v[2] * v[2]))))


def cross3(a, b):
    return [F(F(a[1] * b[2]) - F(b[1] * a[2])), F(F(a[2] * b[0]) - F(b[2] * a[0])), F(F(a[0] * b[1]) - F(b[0] * a[1]))]


def octahedral_coordinates(d):
    """common/octahedral.slangi:56-63."""
    l1 = F(F(np.abs(d[0]) + np.abs(d[1])) + np.abs(d[2]))
    inv = F(f32(1) / l1)
    u, v = F(d[0] * inv), F(d[1] * inv)
    su, sv = np.where(u >= 0, f32(1), f32(-1)), np.where(v >= 0, f32(1), f32(-1))
    fu, fv = F(F(f32(1) - np.abs(v)) * su), F(F(f32(1) - np.abs(u)) * sv)
    neg = d[2] < 0
    return np.where(neg, fu, u), np.where(neg, fv, v)


def probe_uv(pidx, oct_uv, n):
    """common/octahedral.slangi:65-74 -> (u, v); the layer is pidx[2]."""
    out = []
    for i in range(2):
        total = F(f32(n[i]) + f32(2))
        size = F(total * f32(32))
        c = F(F(pidx[i].astype(f32) * total) + F(total * f32(0.5)))
        c = F(c + F(oct_uv[i] * F(f32(n[i]) * f32(0.5))))
        out.append(F(c / size))
    return out


def sample_cascade(loc, direction, cs, cmin, spacing, irr_atlas, depth_atlas, validity, psize=(5, 6)):
    """probe_sampling.slangi:6-106.  loc, direction: 3 fp32 arrays; cs: cascade index per element (int); cmin (3 arrays), spacing: that
    cascade's origin and probe spacing per element; atlases: irradiance (32, 256, 224, 3) fp32, depth moments (32, 384, 384, 2) fp32,
    validity (32, 32, 32) u8; psize = irradiance texels per probe (irradiance_cache.cpp:298-299).  Returns 3 fp32 arrays."""
    shape = np.shape(loc[0])
    ps = [F(F(loc[k] - cmin[k]) / spacing) for k in range(3)]
    mp = [np.floor(p) for p in ps]
    alpha = [clamp01(F(ps[k] - mp[k])) for k in range(3)]
    irr = [np.zeros(shape, dtype=f32) for _ in range(3)]
    weight = np.zeros(shape, dtype=f32)
    ioct = octahedral_coordinates(direction)
    for i in range(8):
        off = [f32(i & 1), f32((i >> 1) & 1), f32((i >> 2) & 1)]
        pl = [F(mp[k] + off[k]) for k in range(3)]
        to_probe = [F(pl[k] - ps[k]) for k in range(3)]
        dist = F(length3(to_probe) * spacing)
        pf = [pl[0], F(pl[1] + F(np.asarray(cs).astype(f32) * f32(8))), pl[2]]
        with np.errstate(invalid="ignore"):
            pidx = [np.where(p > 0, p, f32(0)).astype(np.int64) for p in pf]   # float -> uint: NaN / negative -> 0
        inr = (pidx[0] < 32) & (pidx[1] < 32) & (pidx[2] < 32)
        val = np.where(inr, validity[np.minimum(pidx[2], 31), np.minimum(pidx[1], 31), np.minimum(pidx[0], 31)], 0)
        live = val != 0
        tri = [np.maximum(f32(0.001), F(F(F(f32(1) - alpha[k]) * F(f32(1) - off[k])) + F(alpha[k] * off[k]))) for k in range(3)]
        tw = F(F(tri[0] * tri[1]) * tri[2])
        layer = np.minimum(pidx[2], 31)
        with np.errstate(invalid="ignore", divide="ignore"):
            doct = octahedral_coordinates([F(-t) for t in to_probe])
            duv = probe_uv(pidx, doct, (10, 10))
            dt = h(bilinear_repeat(depth_atlas, np.nan_to_num(duv[0]), np.nan_to_num(duv[1]), layer))
            dx, dy = dt[..., 0], dt[..., 1]
            variance = np.abs(h(h(dx * dx) - dy))
            v = F(dist - dx)
            cheb = F(variance / F(variance + F(v * v)))
            cheb = np.maximum(F(F(cheb * cheb) * cheb), f32(0))
            cheb = np.where(dist > dx, cheb, f32(1))
            pw = np.maximum(f32(0.05), cheb)
            pw = np.maximum(f32(0.000001), pw)
            k_crush = F(f32(1) / F(f32(0.2) * f32(0.2)))
            pw = np.where(pw < f32(0.2), F(pw * F(F(pw * pw) * k_crush)), pw)
            pw = F(pw * tw)
            iuv = probe_uv(pidx, ioct, psize)
            it = h(bilinear_repeat(irr_atlas, iuv[0], iuv[1], layer))
            for c in range(3):
                irr[c] = np.where(live, F(irr[c] + F(it[..., c] * pw)), irr[c])
            weight = np.where(live, F(weight + pw), weight)
    pi_h = h(f32(3.1415927))
    with np.errstate(invalid="ignore", divide="ignore"):
        return [np.where(weight == 0, f32(0), F(F(F(irr[c] / weight) * f32(2)) * pi_h)) for c in range(3)]


# ---- the Lighting pass, per sub-pass --------------------------------------------------------------------------------------
class Frame:
    def __init__(self, W, Hh, seed, sun_mode, gi, sky=False):
        self.W, self.H = W, Hh
        from tests import util  # only its input builder is used here, never the oracle
        self.f = util.golden_lighting_frame(W, Hh, seed, sun_mode, gi, sky=sky)
        a = self.f.arrays
        self.view, self.sun = self.f.view.gpu_data, self.f.sun.constants
        lut = srgb_lut()
        self.base = [lut[a["color"][..., i]] for i in range(3)]
        self.emis = [lut[a["emission"][..., i]] for i in range(3)]
        self.nrm = [a["normals"][..., i].astype(f32) for i in range(3)]
        self.rough = F(a["data"][..., 1].astype(f32) / f32(255.0))
        self.metal = F(a["data"][..., 2].astype(f32) / f32(255.0))
        self.depth = a["depth"]
        ys, xs = np.meshgrid(np.arange(Hh, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
        self.xs, self.ys = xs, ys

    def position(self, glsl):
        """directional_light.frag:45-53 (gl_FragCoord + 0.5 quirk) or directional_light.rt.slang:39-48."""
        v = self.view
        res = (f32(v.render_resolution[0]), f32(v.render_resolution[1]))
        if glsl:
            tx, ty = F(F(F(self.xs + f32(0.5)) + f32(0.5)) / res[0]), F(F(F(self.ys + f32(0.5)) + f32(0.5)) / res[1])
        else:
            tx, ty = F(F(self.xs + f32(0.5)) / res[0]), F(F(self.ys + f32(0.5)) / res[1])
        ndc = [F(F(tx * f32(2)) - f32(1)), F(F(ty * f32(2)) - f32(1)), self.depth, np.ones_like(self.depth)]
        ip = np.array(v.inverse_projection[:], dtype=f32)
        iv = np.array(v.inverse_view[:], dtype=f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            vs = mat_vec(ip, ndc)
            vsd = [F(vs[i] / vs[3]) for i in range(4)]
            ws = mat_vec(iv, [vsd[0], vsd[1], vsd[2], np.ones_like(self.depth) if glsl else vsd[3]])
        return vsd, ws

    def sun_csm(self):
        """directional_light.frag:96-149; returns the fragment colour (rgb) before blending."""
        vs, ws = self.position(True)
        view = np.array(self.view.view[:], dtype=f32)
        cam = [F(-view[12 + i]) for i in range(3)]
        N = normalize3(self.nrm)
        with np.errstate(invalid="ignore"):
            V = normalize3([F(ws[i] - cam[i]) for i in range(3)])
        sd = np.array(self.sun.direction_and_tan_size[:3], dtype=f32)
        L = normalize3([F(-sd[i]) for i in range(3)])
        ndotl = clamp01(dot3_r(N, L))
        # sample_csm :80-94
        cascade = np.zeros(self.depth.shape, dtype=np.int64)
        for i in range(4):
            cascade = np.where(vs[2] < f32(self.sun.data[i][0]), i + 1, cascade)
        with np.errstate(divide="ignore", invalid="ignore"):
            bias = F(F(f32(0.0005) * np.sqrt(F(f32(1) - F(ndotl * ndotl)))) / ndotl)
        shadow = np.ones(self.depth.shape, dtype=f32)
        sm = self.f.arrays["shadowmap"].astype(f32) / f32(65535.0)
        for c in range(4):
            M = np.array(self.sun.cascade_matrices[c][:], dtype=f32)
            B = np.zeros(16, dtype=f32)  # biasMat * M (matrix product first: GLSL is left-associative)
            for col in range(4):
                m = M[col * 4:col * 4 + 4]
                B[col * 4 + 0] = F(F(F(f32(0.5) * m[0] + f32(0) * m[1]) + f32(0) * m[2]) + f32(0.5) * m[3])
                B[col * 4 + 1] = F(F(F(f32(0) * m[0] + f32(0.5) * m[1]) + f32(0) * m[2]) + f32(0.5) * m[3])
                B[col * 4 + 2] = F(F(F(f32(0) * m[0] + f32(0) * m[1]) + f32(1) * m[2]) + f32(0) * m[3])
                B[col * 4 + 3] = F(F(F(f32(0) * m[0] + f32(0) * m[1]) + f32(0) * m[2]) + f32(1) * m[3])
            sp = mat_vec(B, [ws[0], ws[1], ws[2], np.ones_like(self.depth)])
            with np.errstate(invalid="ignore", divide="ignore"):
                sp = [F(sp[i] / sp[3]) for i in range(4)]
            inside = ~((sp[0] < 0) | (sp[1] < 0) | (sp[2] < 0) | (sp[0] > 1) | (sp[1] > 1) | (sp[2] > 1))
            ref = np.clip(F(sp[2] - bias), f32(0), f32(1))
            # PCF :338-352: compare LESS per tap, then filter
            Hs, Ws = sm.shape[1:]
            x0, wx0, fx = axis(np.where(inside, sp[0], f32(0.5)), Ws)
            y0, wy0, fy = axis(np.where(inside, sp[1], f32(0.5)), Hs)
            acc = np.zeros_like(ref)
            for (dy, dx, w) in ((0, 0, F(wx0 * wy0)), (0, 1, F(fx * wy0)), (1, 0, F(wx0 * fy)), (1, 1, F(fx * fy))):
                t = sm[c][np.clip(y0 + dy, 0, Hs - 1), np.clip(x0 + dx, 0, Ws - 1)]
                acc = fma(w, (ref < t).astype(f32), acc)
            s_c = np.where(inside, acc, f32(1))
            shadow = np.where(cascade == c, s_c, shadow)
        shadow = np.where(cascade > 3, f32(0), shadow)
        shadow = np.where(ndotl > 0, shadow, f32(1))
        b = brdf(self.base, N, self.rough, self.metal, L, V)
        col = np.array(self.sun.color[:3], dtype=f32)
        with np.errstate(invalid="ignore"):
            direct = [F(F(F(ndotl * b[i]) * col[i]) * shadow) for i in range(3)]
        bad = np.isnan(direct[0]) | np.isnan(direct[1]) | np.isnan(direct[2])
        return [np.where(bad, f32(0), F(d * f32(0.00031415927))) for d in direct]

    def lpv_overlay(self):
        """gi/lpv/overlay.frag:70-164 for finite volumes and roughness > 0 (specular term == 0)."""
        _, ws = self.position(True)
        N = normalize3(self.nrm)
        gi = self.f
        mats = [np.array(gi.lpv.matrices[c].world_to_cascade[:], dtype=f32) for c in range(4)]
        selected = np.zeros(self.depth.shape, dtype=np.int64)
        for i in (3, 2, 1, 0):
            cp = mat_vec(mats[i], ws)
            inside = (cp[0] > 0) & (cp[1] > 0) & (cp[2] > 0) & (cp[0] < 1) & (cp[1] < 1) & (cp[2] < 1)
            selected = np.where(inside, i, selected)
        nc = [np.full(self.depth.shape, f32(0.282094792)), F(f32(-0.488602512) * F(-N[1])), F(f32(0.488602512) * F(-N[2])),
              F(f32(-0.488602512) * F(F(-N[0]) * f32(-1)))]
        pos = [F(ws[0] + N[0]), F(ws[1] + N[1]), F(ws[2] + N[2]), F(ws[3] + f32(0))]
        cp = [np.zeros_like(self.depth) for _ in range(3)]
        for i in range(4):
            c = mat_vec(mats[i], pos)
            for k in range(3):
                cp[k] = np.where(selected == i, c[k], cp[k])
        cp[0] = F(F(cp[0] + selected.astype(f32)) / f32(4.0))
        vols = [gi.arrays[k].astype(f32) for k in ("lpv_r", "lpv_g", "lpv_b")]
        ind = []
        for vol in vols:
            t = trilinear_border(vol, cp[0], cp[1], cp[2])
            ind.append(F(F(F(t[..., 0] * nc[0] + t[..., 1] * nc[1]) + t[..., 2] * nc[2]) + t[..., 3] * nc[3]))
        fdv = brdf_fd_nn(self.base, N, self.rough, self.metal)
        ao = gi.arrays["ao"]
        total = [F(F(F(ind[i] * fdv[i]) * ao) + f32(0)) for i in range(3)]  # "+ specular_light * (Fr * 0)" == + 0 here
        bad = np.isnan(total[0]) | np.isnan(total[1]) | np.isnan(total[2])
        ex = f32(np.float32(math.pi) * np.float32(10.0))
        return [np.where(bad, f32(0), F(t * ex)) for t in total]

    def sun_rt(self):
        """directional_light.rt.slang:58-139 with shadow / num_samples taken from the mask plane; returns the fp32 addend."""
        _, ws = self.position(False)
        view = np.array(self.view.view[:], dtype=f32)
        cam = [F(-view[12 + i]) for i in range(3)]
        Nh = normalize3([h(n) for n in self.nrm], h)
        sd = np.array(self.sun.direction_and_tan_size[:3], dtype=f32)
        L = normalize3([F(-sd[i]) for i in range(3)])
        ndotl = h(clamp01(F(F(L[0] * Nh[0] + L[1] * Nh[1]) + L[2] * Nh[2])))
        with np.errstate(invalid="ignore"):
            V = [h(x) for x in normalize3([F(ws[i] - cam[i]) for i in range(3)])]
        b = brdf([h(c) for c in self.base], Nh, h(self.rough), h(self.metal), [h(x) for x in L], V, h)
        col = np.array(self.sun.color[:3], dtype=f32)
        rad = [F(h(ndotl * b[i]) * col[i]) for i in range(3)]
        mask = self.f.arrays["shadow_mask"]
        rad = [np.where(ndotl > 0, F(r * mask), r) for r in rad]
        return [F(r * f32(0.00031415927)) for r in rad]

    def emissive(self):
        return [F(e * f32(3.1415927)) for e in self.emis]

    def rtgi_overlay(self):
        """gi/rtgi/overlay.frag.slang:61-117 with num_extra_rays == 0: brdf(surface, ray dir, V) * ray irradiance * clamp(dir . N),
        all in half precision; NaN -> 0; returns rgb (alpha of the fragment is 1)."""
        _, ws = self.position(False)
        view = np.array(self.view.view[:], dtype=f32)
        cam = [F(-view[12 + i]) for i in range(3)]
        Nh = normalize3([h(n) for n in self.nrm], h)
        with np.errstate(invalid="ignore"):
            V = [h(x) for x in normalize3([F(ws[i] - cam[i]) for i in range(3)])]
        rb = self.f.arrays["ray_buffer"].astype(f32)
        ri = self.f.arrays["ray_irr"].astype(f32)
        d = [rb[..., i] for i in range(3)]
        irr = [ri[..., i] for i in range(3)]
        b = brdf([h(c) for c in self.base], Nh, h(self.rough), h(self.metal), d, V, h)
        dn = h(h(h(d[0] * Nh[0]) + h(d[1] * Nh[1])) + h(d[2] * Nh[2]))
        ndotl = h(clamp01(dn))
        rad = [h(h(b[i] * irr[i]) * ndotl) for i in range(3)]
        bad = np.isnan(rad[0]) | np.isnan(rad[1]) | np.isnan(rad[2])
        return [np.where(bad, f32(0), h(r / f32(1))) for r in rad]

    def point_lights(self, lights):
        """Extension a9: fp32, lights in index order, NaN terms dropped, sum * 0.00031415927."""
        _, ws = self.position(True)
        view = np.array(self.view.view[:], dtype=f32)
        cam = [F(-view[12 + i]) for i in range(3)]
        N = normalize3(self.nrm)
        with np.errstate(invalid="ignore"):
            V = normalize3([F(ws[i] - cam[i]) for i in range(3)])
        total = [np.zeros(self.depth.shape, dtype=f32) for _ in range(3)]
        for pl in np.asarray(lights, dtype=f32):
            with np.errstate(all="ignore"):
                lv = [F(pl[i] - ws[i]) for i in range(3)]
                d2 = dot3_r(lv, lv)
                inv = F(f32(1) / F(np.sqrt(d2)))
                L = [F(lv[i] * inv) for i in range(3)]
                dist = F(np.sqrt(d2))
                ndotl = clamp01(dot3_r(N, L))
                xr = F(dist / pl[3])
                x2 = F(xr * xr)
                x4 = F(x2 * x2)
                w = clamp01(F(f32(1) - x4))
                att = F(F(w * w) / np.maximum(d2, f32(1e-4)))
                b = brdf(self.base, N, self.rough, self.metal, L, V)
                k = F(pl[7] * att)
                c = [F(F(F(ndotl * b[i]) * pl[4 + i]) * k) for i in range(3)]
            bad = np.isnan(c[0]) | np.isnan(c[1]) | np.isnan(c[2])
            total = [F(total[i] + np.where(bad, f32(0), c[i])) for i in range(3)]
        return [F(t * f32(0.00031415927)) for t in total]

    def cache_overlay(self):
        """gi/cache/overlay.frag.slang:46-118 + probe_sampling.slangi:6-106 (debug_mode 0). Returns (rgb, drawn): the fragment is
        blended for every surface pixel, with colour 0 outside all cascades."""
        _, ws = self.position(False)
        loc = ws[:3]
        shape = self.depth.shape
        view = np.array(self.view.view[:], dtype=f32)
        cam = [F(-view[12 + i]) for i in range(3)]
        Nh = normalize3([h(n) for n in self.nrm], h)
        with np.errstate(invalid="ignore"):
            V = [h(x) for x in normalize3([F(loc[i] - cam[i]) for i in range(3)])]
        cascades = [([f32(m) for m in cmin], f32(sp)) for (cmin, sp) in self.f.probe_cascades()]
        ci = np.full(shape, 5, dtype=np.int64)
        for i in (3, 2, 1, 0):  # first match wins
            cmin, sp = cascades[i]
            cmax = [F(cmin[k] + F(f32(e) * sp)) for k, e in enumerate((32, 8, 32))]
            inside = np.ones(shape, dtype=bool)
            for k in range(3):
                inside &= (loc[k] > cmin[k]) & (loc[k] < cmax[k])
            ci = np.where(inside, i, ci)
        has = ci <= 3
        cs = np.where(has, ci, 0)
        spacing = np.array([c[1] for c in cascades], dtype=f32)[cs]
        cmin = [np.array([c[0][k] for c in cascades], dtype=f32)[cs] for k in range(3)]

        irr = sample_cascade(loc, Nh, cs, cmin, spacing, unpack_b10g11r11(self.f.arrays["probe_irr"]), self.f.arrays["probe_depth"].astype(f32),
                             self.f.arrays["probe_val"])
        irr_h = [h(x) for x in irr]
        b = brdf([h(c) for c in self.base], Nh, h(self.rough), h(self.metal), Nh, V, h)
        ex = np.float64(0.314159).astype(np.float16).astype(f32)
        col = [h(h(b[c] * irr_h[c]) * ex) for c in range(3)]
        bad = np.isnan(col[0]) | np.isnan(col[1]) | np.isnan(col[2])
        return [np.where(bad | ~has, f32(0), c) for c in col]

    def sky_fill(self):
        """sky/sky_unified.slang:185-206 (main_fs) with :54-166; transcendentals in fp64 rounded to fp32. Returns half rgb."""
        v = self.view
        res = (f32(v.render_resolution[0]), f32(v.render_resolution[1]))
        sx, sy = F(F(F(self.xs + f32(0.5)) + f32(0.5)) / res[0]), F(F(F(self.ys + f32(0.5)) + f32(0.5)) / res[1])   # fragcoord.xy + 0.5
        one = np.ones_like(sx)
        ip = np.array(v.inverse_projection[:], dtype=f32)
        iv = np.array(v.inverse_view[:], dtype=f32)
        vs = mat_vec(ip, [sx, sy, one, one])                     # clip xy is the [0,1] screen location (not remapped)
        vs = [F(vs[i] / vs[3]) for i in range(4)]
        wv = mat_vec(iv, [vs[0], vs[1], vs[2], np.zeros_like(sx)])
        n = normalize3(wv[:3])
        ray = [F(-n[0]), F(F(-n[1]) * f32(-1)), F(-n[2])]
        sd = np.array(self.sun.direction_and_tan_size[:3], dtype=f32)
        sn = normalize3([sd[0], sd[1], sd[2]])
        sun = [F(-x) for x in sn]
        rgb = sky_color(ray, sun, self.f.arrays["sky_v"].astype(f32), self.f.arrays["sky_t"].astype(f32))
        return [h(c) for c in rgb]


def sky_color(ray, sun, lut_v, lut_t):
    """get_sky_color (sky_unified.slang:137-166 with :54-135): ray, sun = 3 fp32 arrays / scalars each; returns 3 fp32 arrays"""
    sx = np.asarray(ray[0], f32)
    if True:

        PI = f32(3.14159265358)
        ground, atmosphere = f32(6.360), f32(6.460)
        view_pos = [f32(0), F(ground + f32(0.0002)), f32(0)]
        acos = lambda x: np.arccos(np.asarray(x, np.float64)).astype(f32)
        # getValFromSkyLUT :80-109
        height = length3(view_pos)
        up = [F(p / height) for p in view_pos]
        horizon = acos(clamp_s(F(np.sqrt(F(F(height * height) - F(ground * ground))) / height), f32(-1), f32(1)))
        ru = dot3_r(ray, up)
        altitude = F(horizon - acos(ru))
        right = cross3(sun, up)
        forward = cross3(up, right)
        with np.errstate(invalid="ignore", divide="ignore"):
            proj = normalize3([F(ray[i] - F(up[i] * ru)) for i in range(3)])
            sin_t, cos_t = dot3_r(proj, right), dot3_r(proj, forward)
            az = F(np.arctan(np.asarray(F(cos_t / sin_t), np.float64)).astype(f32) + PI)
        az = np.where(np.abs(altitude) > F(F(f32(0.5) * PI) - f32(0.0001)), f32(0), az)
        sgn = np.sign(altitude).astype(f32)
        vv = F(f32(0.5) + F(F(f32(0.5) * sgn) * F(np.sqrt(F(F(np.abs(altitude) * f32(2.0)) / PI)))))
        uu = F(az / F(f32(2.0) * PI))
        lum = bilinear_repeat(lut_v, uu, vv)
        # sunWithBloom :120-135
        solid = F(F(f32(0.53) * PI) / f32(180.0))
        min_cos = f32(np.cos(np.float64(solid)))
        cos_theta = dot3_r(ray, sun)
        offset = F(min_cos - cos_theta)
        gauss = F(np.exp(np.asarray(F(F(-offset) * f32(50000.0)), np.float64)).astype(f32) * f32(0.5))
        inv = F(F(f32(1.0) / F(f32(0.02) + F(offset * f32(300.0)))) * f32(0.01))
        s = np.where(cos_theta >= min_cos, f32(1), F(gauss + inv))
        # smoothstep(0.002h, 1.0h, .) :143
        e0, e1 = np.float64(0.002).astype(np.float16).astype(f32), f32(1.0)
        t = clamp01(F(F(s - e0) / F(e1 - e0)))
        s = F(F(t * t) * F(f32(3.0) - F(f32(2.0) * t)))
        # :144-155
        applied = length3([s, s, s]) > 0
        b = dot3_r(view_pos, ray)
        c = F(dot3_r(view_pos, view_pos) - F(ground * ground))
        discr = F(F(b * b) - c)
        with np.errstate(invalid="ignore"):
            root = np.sqrt(discr)
            hit_t = np.where(discr > F(b * b), F(F(-b) + root), F(F(-b) - root))
        hit_t = np.where(discr < 0, f32(-1), hit_t)
        hit_t = np.where((c > 0) & (b > 0), f32(-1), hit_t)
        # getValFromTLUT :111-118 at viewPos
        sun_cos = dot3_r(sun, up)
        tu = clamp_s(F(f32(0.5) + F(f32(0.5) * sun_cos)), f32(0), f32(1))
        tv = np.maximum(f32(0), np.minimum(f32(1), F(F(height - ground) / F(atmosphere - ground))))
        trans = bilinear_repeat(lut_t, np.full_like(sx, tu), np.full_like(sx, tv))
        out = []
        for ch in range(3):
            sl = np.where(applied, np.where(hit_t >= 0, f32(0), F(s * trans[..., ch])), s)
            out.append(F(F(F(lum[..., ch] + sl) * f32(20.0)) * f32(1.0)))
        return out


def dot3_r(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def clamp_s(x, lo, hi):
    return np.minimum(np.maximum(x, lo), hi)


def brdf_fd_nn(base, n, rough, metal):
    """Fd(surface, N, N) (overlay.frag:149) through the full formula."""
    one = f32(1.0)
    diff = [F(F(base[i] * F(one - f32(0.04))) * F(one - metal)) for i in range(3)]
    hv = normalize3([F(n[i] + n[i]) for i in range(3)])
    NoV = np.abs(F(dot3_r(n, n) + f32(1e-5)))
    NoL = clamp01(dot3_r(n, n))
    LoH = clamp01(dot3_r(n, hv))
    f90 = F(f32(0.5) + F(F(F(f32(2.0) * rough) * LoH) * LoH))
    s = lambda u: F(one + F(F(f90 - one) * pow5(clamp01(F(one - u)))))
    fdv = F(F(s(NoL) * s(NoV)) * F(one / f32(3.1415927)))
    return [F(d * fdv) for d in diff]


def compose(fr, sun_mode, gi, lights=None):
    """lighting_phase.cpp:99-134 with the RGBA16F blend roundings."""
    surf = fr.depth != 0
    lit = [np.zeros(fr.depth.shape, dtype=f32) for _ in range(4)]
    if sun_mode == _abi.SHADOW_MODE_CSM:
        s = fr.sun_csm()
        for i in range(3):
            lit[i] = np.where(surf, h(F(F(s[i] * s[i]) + F(lit[i] * lit[i]))), lit[i])  # SRC_COLOR / DST_COLOR blend
        # alpha factors ZERO/ZERO: stays 0
    if lights is not None:  # extension a9: additive, after the CSM sun, before the GI overlay
        pl = fr.point_lights(lights)
        for i in range(3):
            lit[i] = np.where(surf, h(F(lit[i] + pl[i])), lit[i])
        lit[3] = np.where(surf, h(F(lit[3] + f32(1))), lit[3])
    if gi in (_abi.GI_LPV, _abi.GI_RTGI, _abi.GI_CACHE):
        g = fr.lpv_overlay() if gi == _abi.GI_LPV else fr.rtgi_overlay() if gi == _abi.GI_RTGI else fr.cache_overlay()
        for i in range(3):
            lit[i] = np.where(surf, h(F(lit[i] + g[i])), lit[i])
        lit[3] = np.where(surf, h(F(lit[3] + f32(1))), lit[3])
    e = fr.emissive()
    for i in range(3):
        lit[i] = h(F(lit[i] + e[i]))
    lit[3] = h(F(lit[3] + f32(1)))
    if fr.f.has_sky:  # procedural_sky.cpp:151-172: depth == 0 pixels are overwritten (no blending)
        sky = fr.sky_fill()
        for i in range(3):
            lit[i] = np.where(surf, lit[i], sky[i])
        lit[3] = np.where(surf, lit[3], f32(1))
    if sun_mode == _abi.SHADOW_MODE_RT:
        a = fr.sun_rt()
        for i in range(3):
            lit[i] = np.where(surf, h(F(lit[i] + a[i])), lit[i])
    with np.errstate(over="ignore"):
        return np.stack(lit, axis=-1).astype(np.float16).view(np.uint16)


# ---- post chain ---------------------------------------------------------------------------------------------------------------
def bloom_downsample(src, dw, dh):
    """bloom_downsample.comp:16-52; src (H, W, 4) fp16 bits -> (dh, dw, 4) fp16 bits."""
    img = src.view(np.float16).astype(f32)[..., :3]
    Hs, Ws = img.shape[:2]
    ys, xs = np.meshgrid(np.arange(dh, dtype=f32), np.arange(dw, dtype=f32), indexing="ij")
    u, v = F(F(xs + f32(0.5)) / f32(dw)), F(F(ys + f32(0.5)) / f32(dh))
    ix, iy = F(f32(1) / f32(Ws)), F(f32(1) / f32(Hs))
    o = [F(ix * f32(-1)), F(iy * f32(-1)), F(ix * f32(1)), F(iy * f32(1))]

    def box(uu, vv):
        s = bilinear_clamp(img, F(uu + o[0]), F(vv + o[1]))
        s = F(s + bilinear_clamp(img, F(uu + o[2]), F(vv + o[1])))
        s = F(s + bilinear_clamp(img, F(uu + o[0]), F(vv + o[3])))
        s = F(s + bilinear_clamp(img, F(uu + o[2]), F(vv + o[3])))
        return F(s * f32(0.25))

    s = F(box(u, v) * f32(0.5))
    for (a, b) in ((0, 1), (2, 1), (0, 3), (2, 3)):
        s = F(s + F(box(F(u + o[a]), F(v + o[b])) * f32(0.125)))
    out = np.zeros((dh, dw, 4), dtype=np.float16)
    with np.errstate(over="ignore"):
        out[..., :3] = s.astype(np.float16)
    return out.view(np.uint16)


def tonemap(scene_bits, mips_bits, ow, oh):
    """scene_upsample.frag:20-72 + sRGB swap-chain write."""
    scene_img = scene_bits.view(np.float16).astype(f32)
    ys, xs = np.meshgrid(np.arange(oh, dtype=f32), np.arange(ow, dtype=f32), indexing="ij")
    u = F(F(xs + f32(0.5)) / f32(ow))
    v = F(f32(1) - F(F(ys + f32(0.5)) / f32(oh)))
    bloom = np.zeros((oh, ow, 3), dtype=f32)
    for mb in mips_bits[:6]:
        img = mb.view(np.float16).astype(f32)[..., :3]
        Hm, Wm = img.shape[:2]
        ix, iy = F(f32(1) / f32(Wm)), F(f32(1) / f32(Hm))
        o = [F(ix * f32(-1)), F(iy * f32(-1)), F(ix * f32(1)), F(iy * f32(1))]
        z = f32(0)
        taps = [((z, z), 4.0), ((o[0], z), 2.0), ((o[1], z), 2.0), ((z, o[2]), 2.0), ((z, o[3]), 2.0), ((o[0], o[1]), 1.0), ((o[2], o[1]), 1.0),
                ((o[0], o[3]), 1.0), ((o[2], o[3]), 1.0)]
        s = None
        for k, ((du, dv), wgt) in enumerate(taps):
            uu = u if (k == 0) else F(u + du)
            vv = v if (k == 0) else F(v + dv)
            t = F(bilinear_clamp(img, uu, vv) * f32(wgt))
            s = t if s is None else F(s + t)
        bloom = F(bloom + F(s / f32(16.0)))
    sc = bilinear_clamp(scene_img, u, v)[..., :3]
    c = F(sc + F(bloom * f32(0.014159)))
    luma = F(F(F(c[..., 0] * f32(0.2126)) + F(c[..., 1] * f32(0.7152))) + F(c[..., 2] * f32(0.0722)))
    with np.errstate(invalid="ignore", divide="ignore"):
        factor = F(luma / F(luma + f32(1)))
        mapped = F(c * factor[..., None])
        g = np.power(mapped.astype(np.float64), np.float64(f32(1.0) / f32(2.2))).astype(f32)
        d = g.astype(np.float64)
        s = np.where(d <= 0.0031308, 12.92 * d, 1.055 * np.power(d, 1.0 / 2.4) - 0.055).astype(f32)
    s = np.where(g >= 1, f32(1), np.where(g > 0, s, f32(0)))
    code = np.where(s >= 1, 255, np.where(s > 0, (F(F(s * f32(255)) + f32(0.5))).astype(np.int64), 0)).astype(np.uint8)
    out = np.full((oh, ow, 4), 255, dtype=np.uint8)
    out[..., :3] = code
    return out


def copy_scene(lit_bits):
    """util/copy_with_sampler.frag.slang:9-12 as drawn by scene_renderer.cpp:502-527: antialiased = bilinear (REPEAT) sample of
    lit_scene at SV_Position.xy * (1/W, 1/H); all four channels."""
    img = lit_bits.view(np.float16).astype(f32)
    Hh, W = img.shape[:2]
    ys, xs = np.meshgrid(np.arange(Hh, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    u, v = F(F(xs + f32(0.5)) * F(f32(1) / f32(W))), F(F(ys + f32(0.5)) * F(f32(1) / f32(Hh)))
    x0, wx0, fx = axis(u, W)
    y0, wy0, fy = axis(v, Hh)
    acc = np.zeros(img.shape, dtype=f32)
    for (dy, dx, w) in ((0, 0, F(wx0 * wy0)), (0, 1, F(fx * wy0)), (1, 0, F(wx0 * fy)), (1, 1, F(fx * fy))):
        acc = fma(w[..., None], img[(y0 + dy) % Hh, (x0 + dx) % W], acc)
    with np.errstate(over="ignore"):
        return acc.astype(np.float16).view(np.uint16)


def lpv_propagate(vols, steps, num_cascades=1):
    """gi/lpv/lpv_propagate.comp.slang:76-156 in half precision (numpy float16 arithmetic rounds after every operator), use_gv = false.
    vols: three (32, 32, 32 * num_cascades, 4) float16 arrays [z][y][x][c]; returns the three arrays after `steps` ping-pong steps."""
    hf = np.float16
    orient = np.array([[1, 0, 0, 0, 1, 0, 0, 0, 1], [-1, 0, 0, 0, 1, 0, 0, 0, -1], [0, 0, 1, 0, 1, 0, -1, 0, 0], [0, 0, -1, 0, 1, 0, 1, 0, 0],
                       [1, 0, 0, 0, 0, 1, 0, -1, 0], [1, 0, 0, 0, 0, -1, 0, 1, 0]], dtype=np.float32).reshape(6, 3, 3)
    dirs = np.array([[0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]], dtype=np.int64)
    sides = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]], dtype=np.float32)

    def hmul33(M, v):  # half3 = mul(half3x3, half3): products and sums in half, left to right
        M = M.astype(hf)
        return [hf(hf(hf(M[r, 0] * v[0]) + hf(M[r, 1] * v[1])) + hf(M[r, 2] * v[2])) for r in range(3)]

    def sh(d):  # float literal * half -> float, rounded by the half4 constructor
        return [hf(0.282094792), hf(f32(-0.488602512) * f32(d[1])), hf(f32(0.488602512) * f32(d[2])), hf(f32(-0.488602512) * f32(d[0]))]

    def lobe(d):
        return [hf(0.886226925), hf(f32(-1.02332671) * f32(d[1])), hf(f32(1.02332671) * f32(d[2])), hf(f32(-1.02332671) * f32(d[0]))]

    small, big = hf(0.4472135), hf(0.894427)
    direct_sa = hf(f32(hf(0.4006696846)) / f32(3.1415927))
    side_sa = hf(f32(hf(0.4234413544)) / f32(3.1415927))
    W = 32 * num_cascades
    cur = [v.copy() for v in vols]
    for _ in range(steps):
        nxt = []
        for vol in cur:
            acc = np.zeros((32, 32, W, 4), dtype=hf)
            for n in range(6):
                dx, dy, dz = (int(t) for t in dirs[n])
                # neighbour cell = cell - dir; cells whose neighbour index leaves [-1, 31] (per cascade, asymmetric) are skipped;
                # index -1 in x reads the previous cascade's last column (or zero at the volume edge), other out-of-range reads are zero
                coef = np.zeros((32, 32, W, 4), dtype=hf)
                skip = np.zeros((32, 32, W), dtype=bool)
                zs, ys, xs = np.meshgrid(np.arange(32), np.arange(32), np.arange(W), indexing="ij")
                cx = xs % 32
                nx, ny, nz = cx - dx, ys - dy, zs - dz
                skip = (nx < -1) | (ny < -1) | (nz < -1) | (nx > 31) | (ny > 31) | (nz > 31)
                gx = nx + (xs - cx)
                ok = (~skip) & (gx >= 0) & (gx < W) & (ny >= 0) & (ny < 32) & (nz >= 0) & (nz < 32)
                coef[ok] = vol[nz[ok], ny[ok], gx[ok]]
                with np.errstate(all="ignore"):
                    for s in range(4):
                        e = hmul33(orient[n], [hf(hf(sides[s, 0]) * small), hf(hf(sides[s, 1]) * small), big])
                        r = hmul33(orient[n], [hf(sides[s, 0]), hf(sides[s, 1]), hf(0)])
                        es, rl = sh(e), lobe(r)
                        dot = hf(hf(hf(coef[..., 0] * es[0]) + hf(coef[..., 1] * es[1])) + hf(coef[..., 2] * es[2]))
                        dot = hf(dot + hf(coef[..., 3] * es[3]))
                        m = np.maximum(hf(0), dot)
                        k = hf(side_sa * m)
                        add = np.stack([hf(hf(k * rl[c]) * hf(1)) for c in range(4)], axis=-1)
                        acc = np.where(skip[..., None], acc, hf(acc + add))
                    c = [hf(t) for t in dirs[n]]
                    cs, cl = sh(c), lobe(c)
                    dot = hf(hf(hf(coef[..., 0] * cs[0]) + hf(coef[..., 1] * cs[1])) + hf(coef[..., 2] * cs[2]))
                    dot = hf(dot + hf(coef[..., 3] * cs[3]))
                    m = np.maximum(hf(0), dot)
                    k = hf(direct_sa * m)
                    add = np.stack([hf(hf(k * cl[ch]) * hf(1)) for ch in range(4)], axis=-1)
                    acc = np.where(skip[..., None], acc, hf(acc + add))
            nxt.append(acc)
        cur = nxt
    return cur


# ---- irradiance-cache probe update (a11) ------------------------------------------------------------------------------------
def pack_b10g11r11(rgb_half):
    """half3 -> B10G11R11 as a storage-image write does it under the contract (round toward zero, negatives -> 0):
    the unsigned small floats are the leading bits of the fp16 pattern."""
    b = np.asarray(rgb_half, dtype=np.float16).view(np.uint16).astype(np.uint32)
    b = np.where(b & 0x8000, 0, b)
    return int(((b[0] >> 4) & 0x7FF) | (((b[1] >> 4) & 0x7FF) << 11) | (((b[2] >> 5) & 0x3FF) << 22))


def texel_octahedral_direction(tx, ty, nx, ny):
    """common/octahedral.slangi:25-39 then :44-50, fp32."""
    cx = F(F(F(F(f32(tx % nx) + f32(0.5)) / f32(nx)) * f32(2)) - f32(1))
    cy = F(F(F(F(f32(ty % ny) + f32(0.5)) / f32(ny)) * f32(2)) - f32(1))
    d = [cx, cy, F(F(f32(1) - np.abs(cx)) - np.abs(cy))]
    if d[2] < 0:
        sx, sy = (f32(1) if d[0] >= 0 else f32(-1)), (f32(1) if d[1] >= 0 else f32(-1))
        d[0], d[1] = F(F(f32(1) - np.abs(cy)) * sx), F(F(f32(1) - np.abs(cx)) * sy)
    return normalize3(d)


def border_targets(rx, ry, pid, tx, ty):
    """probe_update.slangi:4-37: the texels one invocation stores its value to, in program order (x, y, layer)."""
    bx, by, bz = int(pid[0]) * (rx + 2), int(pid[1]) * (ry + 2), int(pid[2])
    sgn = lambda v: (v > 0) - (v < 0)
    ge0 = lambda v: 1 if v >= 0 else 0
    edge_x, edge_y = tx in (0, rx - 1), ty in (0, ry - 1)
    mx, my = tx - rx // 2, ty - ry // 2
    mx, my = mx + ge0(mx), my + ge0(my)
    out = [(tx + bx, ty + by, bz)]
    if edge_x and edge_y:
        dx, dy = -mx - ge0(mx) + rx // 2, -my - ge0(my) + ry // 2
        out.append((dx + bx, dy + by, bz))
    if edge_x:
        ex, ey = mx + sgn(mx), -my
        ex, ey = ex - ge0(ex) + rx // 2, ey - ge0(ey) + ry // 2
        out.append((ex + bx, ey + by, bz))
    if edge_y:
        ex, ey = -mx, my + sgn(my)
        ex, ey = ex - ge0(ex) + rx // 2, ey - ge0(ey) + ry // 2
        out.append((ex + bx, ey + by, bz))
    return out


def probe_update(atl, trace, ids):
    """probe_depth_update / probe_light_cache_update / probe_rtgi_update / probe_finalize (.comp.slang), one workgroup per listed
    probe, with the order include/sah_hip.h defines where the shaders race: invocations in ascending linear index, stores in program
    order, the wave sum in lane order in fp16.  atl: dict of the five atlases (modified in place); trace (P, 20, 20, 4) fp16."""
    H16 = np.float16
    tr = trace.astype(H16)

    def fetch(p, x, y):  # out-of-range image loads return 0
        return tr[p, y, x] if (0 <= x < 20 and 0 <= y < 20) else np.zeros(4, H16)

    def store(img, targets, value):
        L, Hh, W = img.shape[:3]
        for (x, y, z) in targets:
            if 0 <= x < W and 0 <= y < Hh and 0 <= z < L:
                img[z, y, x] = value

    with np.errstate(over="ignore", invalid="ignore"):
        for p, pid in enumerate(ids):          # depth moments, 10 x 10 threads
            for ty in range(10):
                for tx in range(10):
                    depth, n = H16(0), H16(0)
                    for i in range(4):
                        d = fetch(p, tx * 2 + i % 2, ty * 2 + i // 2)[3]
                        if d > 0:
                            depth, n = H16(depth + d), H16(n + H16(1))
                    depth = H16(depth / n) if n > 0 else H16(0)
                    store(atl["depth"], border_targets(10, 10, pid, tx, ty), np.array([depth, H16(depth * depth)], H16))
        for p, pid in enumerate(ids):          # light cache, 11 x 11 threads, filter = ceil(20 / 11) = 2
            for ty in range(11):
                for tx in range(11):
                    direction = [np.float32(H16(c)) for c in texel_octahedral_direction(tx, ty, 11, 11)]
                    light, n = np.zeros(3, H16), H16(0)
                    for i in range(4):
                        rx, ry = tx * 2 + i % 2, ty * 2 + i // 2
                        t = fetch(p, rx, ry)
                        if t[3] > 0:
                            w = H16(dot3_r(direction, texel_octahedral_direction(rx, ry, 20, 20)))   # dot(half3, float3) is a float dot
                            light = np.array([H16(H16(t[c] * w) + light[c]) for c in range(3)], H16)
                            n = H16(n + H16(1))
                    light = np.array([H16(c / n) for c in light], H16) if n > 0 else np.zeros(3, H16)
                    store(atl["light_cache"], border_targets(11, 11, pid, tx, ty), pack_b10g11r11(light))
        for p, pid in enumerate(ids):          # irradiance, 5 x 6 threads, filter = 20 / 5 = 4 (in x AND y: rows 0..23)
            for ty in range(6):
                for tx in range(5):
                    light, n = np.zeros(3, H16), H16(0)
                    for i in range(16):
                        t = fetch(p, tx * 4 + i % 4, ty * 4 + i // 4)
                        if t[3] > 0:
                            light = np.array([H16(light[c] + t[c]) for c in range(3)], H16)
                            n = H16(n + H16(1))
                    light = np.array([H16(c / n) for c in light], H16) if n > 0 else np.zeros(3, H16)
                    store(atl["rtgi"], border_targets(5, 6, pid, tx, ty), pack_b10g11r11(light))
        for p, pid in enumerate(ids):          # finalize: 64 lanes, every lane tests texels idx = 0 and idx = 64 (the loop ignores the lane)
            x0, y0, z0 = int(pid[0]), int(pid[1]), int(pid[2])
            valid = 0
            for idx in (0, 64):
                d = atl["depth"][z0, y0 * 12 + idx // 10 + 1, x0 * 12 + idx % 10 + 1, 0]
                if d > 0:
                    valid += 64
            v = np.float32(H16(H16(valid) / H16(100)))
            atl["validity"][z0, y0, x0] = 255 if v >= 1 else (0 if not v > 0 else int(F(F(v * f32(255)) + f32(0.5))))
            total = None
            for lane in range(30):             # x = lane % 5, y = lane / 6 (sic)
                t = unpack_b10g11r11(atl["rtgi"][z0, y0 * 8 + lane // 6 + 1, x0 * 7 + lane % 5 + 1]).astype(H16)
                total = t if total is None else np.array([H16(total[c] + t[c]) for c in range(3)], H16)
            atl["average"][z0, y0, x0] = pack_b10g11r11(np.array([H16(c / H16(30)) for c in total], H16))
    return atl


# ---- scene rasteriser: depth pre-pass + G-buffer pass with material textures (f1) ---------------------------------------------------
# Written from include/sah_hip.h / DESIGN.md §5d (the rules Vulkan leaves open) and gltf_basic_pbr.slang:110-253 (SAH_MAIN_VIEW), for
# geometry that needs no clipping: vertex stage, 24.8 window coordinates, integer edge functions with the top-left rule, clockwise
# front faces, screen-linear depth from fp64 plane coefficients, perspective-correct varyings, fine quad derivatives, SampleBias.
from fractions import Fraction


def _fma64(a, b, c):
    """fp64 fused multiply-add, exactly rounded."""
    return float(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def _srgb_encode8(v):
    """half value (as fp32) -> sRGB8 code of an R8G8B8A8_SRGB store: OETF in fp64 rounded to fp32, then floor(s * 255 + 0.5) in fp32"""
    v = np.asarray(v, f32)
    d = v.astype(np.float64)
    with np.errstate(invalid="ignore"):
        s = np.where(d <= 0.0031308, 12.92 * d, 1.055 * np.power(np.maximum(d, 0.0), 1.0 / 2.4) - 0.055).astype(f32)
    s = np.where(v >= 1, f32(1), np.where(v > 0, s, f32(0)))  # NaN and non-positive values encode to 0
    return _unorm8(s)


def _unorm8(c):
    c = np.asarray(c, f32)
    with np.errstate(invalid="ignore"):
        q = np.floor(F(F(c * f32(255)) + f32(0.5)))
    return np.where(c >= 1, 255, np.where(c > 0, q, 0)).astype(np.uint8)


def _wrap(i, n, mode):
    if mode == _abi.ADDRESS_CLAMP_TO_EDGE:
        return np.clip(i, 0, n - 1)
    if mode == _abi.ADDRESS_MIRRORED_REPEAT:
        m = np.mod(i, 2 * n)
        return np.where(m < n, m, 2 * n - 1 - m)
    return np.mod(i, n)


def _texel_values(level, srgb):
    t = level.astype(np.float64) / 255.0
    out = (level.astype(f32) / f32(255.0)).astype(f32)
    if srgb:
        out[..., :3] = np.where(t[..., :3] <= 0.04045, t[..., :3] / 12.92, ((t[..., :3] + 0.055) / 1.055) ** 2.4).astype(f32)
    return out


def _sample_level(level, srgb, smp, filt, u, v):
    tex = _texel_values(level, srgb)
    hgt, wid = level.shape[:2]
    if filt == _abi.FILTER_NEAREST:
        x = _wrap(np.floor(F(u * f32(wid))).astype(np.int64), wid, smp.address_u)
        y = _wrap(np.floor(F(v * f32(hgt))).astype(np.int64), hgt, smp.address_v)
        return tex[y, x]
    x0, wx0, fx = axis(u, wid)
    y0, wy0, fy = axis(v, hgt)
    acc = np.zeros(u.shape + (4,), f32)
    for (dy, dx, w) in ((0, 0, F(wx0 * wy0)), (0, 1, F(fx * wy0)), (1, 0, F(wx0 * fy)), (1, 1, F(fx * fy))):
        t = tex[_wrap(y0 + dy, hgt, smp.address_v), _wrap(x0 + dx, wid, smp.address_u)]
        acc = fma(w[..., None], t, acc)
    return acc


def sample_bias(texture, uv, ddx, ddy, shader_bias, explicit_lod=None):
    """textures[i].SampleBias(uv, bias) under the rules of include/sah_hip.h (sah_texture); uv, ddx, ddy: pairs of fp32 arrays.
    explicit_lod: SampleLevel(uv, lod) instead — the level of detail is given, not derived (ddx / ddy unused)."""
    mips, fmt, smp = texture
    srgb = fmt == _abi.FORMAT_R8G8B8A8_SRGB
    if explicit_lod is not None:
        lam = np.full(uv[0].shape, f32(explicit_lod), f32)
    else:
        w0, h0 = f32(mips[0].shape[1]), f32(mips[0].shape[0])
        mxx, mxy, myx, myy = F(ddx[0] * w0), F(ddx[1] * h0), F(ddy[0] * w0), F(ddy[1] * h0)
        rho2 = np.fmax(F(F(mxx * mxx) + F(mxy * mxy)), F(F(myx * myx) + F(myy * myy)))
        with np.errstate(divide="ignore", invalid="ignore"):
            lam = np.where(rho2 > 0, F(f32(0.5) * np.log2(rho2.astype(np.float64)).astype(f32)), f32(-np.inf))
        A = f32(getattr(smp, "max_anisotropy", 0.0))
        if A > 1:  # anisotropic footprint (sah_hip.h "anisotropy"): per pixel N taps along the major axis, each a complete sample
            rx, ry = F(F(mxx * mxx) + F(mxy * mxy)), F(F(myx * myx) + F(myy * myy))
            rmin2 = np.fmin(rx, ry)
            with np.errstate(divide="ignore", invalid="ignore"):
                eta = np.where(rho2 > 0, np.where(rmin2 > 0, np.fmin(np.sqrt(F(rho2 / rmin2)).astype(f32), A), A), f32(1)).astype(f32)
                lam_a = np.where(rho2 > 0, F(lam - np.log2(eta.astype(np.float64)).astype(f32)), lam).astype(f32)
            N = np.ceil(eta).astype(np.int64)
            major_x = rx > ry
            d = (np.where(major_x, ddx[0], ddy[0]).astype(f32), np.where(major_x, ddx[1], ddy[1]).astype(f32))
            out = _sample_at_lambda(texture, uv, lam_a, shader_bias)  # N == 1
            for n in range(2, int(N.max()) + 1):
                acc = np.zeros(uv[0].shape + (4,), f32)
                for i in range(1, n + 1):
                    a = F(F(f32(i) / f32(n + 1)) - f32(0.5))
                    p = (F(uv[0] + F(a * d[0])), F(uv[1] + F(a * d[1])))
                    acc = F(acc + _sample_at_lambda(texture, p, lam_a, shader_bias))
                out = np.where((N == n)[..., None], F(acc / f32(n)), out)
            return out
    return _sample_at_lambda(texture, uv, lam, shader_bias)


def _sample_at_lambda(texture, uv, lam, shader_bias):
    """the part of a sample after the level of detail: bias, clamps, filter choice, level blend"""
    mips, fmt, smp = texture
    srgb = fmt == _abi.FORMAT_R8G8B8A8_SRGB
    lam = F(lam + F(f32(smp.mip_lod_bias) + f32(shader_bias)))
    lam = np.fmin(np.fmax(lam, f32(smp.min_lod)), f32(smp.max_lod))
    q = len(mips) - 1
    out = np.zeros(uv[0].shape + (4,), f32)
    mag = lam <= 0

    def level_sample(level):
        a = _sample_level(mips[level], srgb, smp, smp.mag_filter, uv[0], uv[1])
        b = _sample_level(mips[level], srgb, smp, smp.min_filter, uv[0], uv[1])
        return np.where(mag[..., None], a, b)
    if smp.mipmap_mode == _abi.FILTER_NEAREST:
        level = np.where(lam <= 0.5, 0, np.where(lam < q, np.minimum(np.ceil(F(lam + f32(0.5))) - 1, q), q)).astype(np.int64)
        for k in range(q + 1):
            out = np.where((level == k)[..., None], level_sample(k), out)
        return out
    d = np.fmin(np.fmax(lam, f32(0)), f32(q))
    hi = np.floor(d).astype(np.int64)
    lo = np.minimum(hi + 1, q)
    delta = F(d - np.floor(d))
    ta, tb = np.zeros_like(out), np.zeros_like(out)
    for k in range(q + 1):
        sk = level_sample(k)
        ta = np.where((hi == k)[..., None], sk, ta)
        tb = np.where((lo == k)[..., None], sk, tb)
    return F(F(F(f32(1) - delta)[..., None] * ta) + F(delta[..., None] * tb))


def raster_fragments(m, to_clip, W, Hh, bias, shade=True):
    """Every triangle of mesh.Mesh `m` (all of them inside the clip volume) in draw order, through `to_clip(world) -> clip`:
    yields (kept fragments mask, depth, G-buffer texels or None)."""
    vd_all = np.concatenate(m.vertex_data)
    pos_all = np.concatenate(m.positions)
    idx_all = np.concatenate(m.indices)
    ys, xs = np.meshgrid(np.arange(Hh, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")
    # draw order: SOLID primitives, then CUTOUT ones, list order inside a class
    prims = [p for t in (_abi.PRIMITIVE_TYPE_SOLID, _abi.PRIMITIVE_TYPE_CUTOUT) for p in m.primitives if int(p["type"]) == t]
    frags = []  # per triangle: coverage mask, depth, the shaded G-buffer texel arrays, alpha discard
    for p in prims:
        model = np.array(p["model"], f32)
        mat = m.materials[int(p["material"])]
        slots = m.material_textures[int(p["material"])]
        first, count, voff = int(p["first_index"]), int(p["index_count"]), int(p["vertex_offset"])
        for t in range(count // 3):
            ids = [voff + int(idx_all[first + 3 * t + k]) for k in range(3)]
            # vertex stage :126-141
            clip, vo = [], []
            for i in ids:
                world = mat_vec(model, [f32(pos_all[i][0]), f32(pos_all[i][1]), f32(pos_all[i][2]), f32(1)])
                clip.append(to_clip(world))
                vd = vd_all[i]
                rot = lambda v: [F(F(F(model[0 + r] * v[0]) + F(model[4 + r] * v[1])) + F(model[8 + r] * v[2])) for r in range(3)]
                n3 = [h(c) for c in normalize3(rot(vd["normal"]))]
                t3 = [h(c) for c in normalize3(rot(vd["tangent"][:3]))]
                col = [h(F(f32((int(vd["color"]) >> (8 * c)) & 0xFF) / f32(255))) for c in range(4)]
                vo.append({"color": col, "normal": n3, "tangent": t3 + [h(f32(vd["tangent"][3]))], "uv": [f32(vd["texcoord"][0]), f32(vd["texcoord"][1])]})
            # window coordinates: 8 sub-pixel bits, round to nearest even
            hw, hh = f32(W * 0.5), f32(Hh * 0.5)
            X, Y, Z, IW = [], [], [], []
            for c in clip:
                X.append(int(np.rint(F(F(F(c[0] / c[3]) * hw + hw) * f32(256)))))
                Y.append(int(np.rint(F(F(F(c[1] / c[3]) * hh + hh) * f32(256)))))
                Z.append(F(c[2] / c[3]))
                IW.append(F(f32(1) / c[3]))
            order = [0, 1, 2]
            area = (X[1] - X[0]) * (Y[2] - Y[0]) - (X[2] - X[0]) * (Y[1] - Y[0])
            if area == 0 or (area < 0 and int(p["type"]) == _abi.PRIMITIVE_TYPE_SOLID):
                continue  # degenerate, or a back face of a culled primitive (clockwise in window space = front)
            if area < 0:
                order, area = [0, 2, 1], -area
            Xo, Yo, Zo, Wo = [X[k] for k in order], [Y[k] for k in order], [Z[k] for k in order], [IW[k] for k in order]
            inv_area = F(f32(1) / f32(area))

            def edges(px, py):
                cx, cy = px * 256 + 128, py * 256 + 128
                out = []
                for i in range(3):
                    a, b = (i + 1) % 3, (i + 2) % 3
                    dx, dy = Xo[b] - Xo[a], Yo[b] - Yo[a]
                    out.append((dx * (cy - Yo[a]) - dy * (cx - Xo[a]), dx, dy))
                return out

            def lambdas(px, py):
                e = edges(px, py)
                with np.errstate(all="ignore"):
                    b = [F(e[i][0].astype(f32) * inv_area) for i in range(3)]
                    q = [F(b[i] * Wo[i]) for i in range(3)]
                    s = F(F(q[0] + q[1]) + q[2])
                    l = [F(q[i] / s) for i in range(3)]
                lam = [None, None, None]
                for w_i, in_i in enumerate(order):  # back to the input triangle's vertex order
                    lam[in_i] = l[w_i]
                return lam
            e = edges(xs, ys)
            cover = np.ones((Hh, W), bool)
            for (val, dx, dy) in e:
                cover &= (val > 0) | ((val == 0) & ((dy < 0) or (dy == 0 and dx > 0)))
            if not cover.any():
                continue
            # depth plane in fp64, every operator rounded; z(px, py) = fma(py, zy, fma(px, zx, zc))
            ea, eb, ec = [], [], []
            for i in range(3):
                a, b = (i + 1) % 3, (i + 2) % 3
                dx, dy = float(Xo[b] - Xo[a]), float(Yo[b] - Yo[a])
                ea.append(-256.0 * dy)
                eb.append(256.0 * dx)
                ec.append(dx * float(128 - Yo[a]) - dy * float(128 - Xo[a]))
            inv = 1.0 / float(area)
            z64 = [float(z) for z in Zo]
            zc = ((ec[0] * z64[0] + ec[1] * z64[1]) + ec[2] * z64[2]) * inv
            zx = ((ea[0] * z64[0] + ea[1] * z64[1]) + ea[2] * z64[2]) * inv
            zy = ((eb[0] * z64[0] + eb[1] * z64[1]) + eb[2] * z64[2]) * inv
            depth = np.zeros((Hh, W), f32)
            for (py, px) in np.argwhere(cover):
                depth[py, px] = f32(_fma64(py, zy, _fma64(px, zx, zc)))
            depth = np.clip(depth, f32(0), f32(1))
            # fragment stage :169-228
            lam = lambdas(xs, ys)
            lam_x, lam_y = lambdas(xs ^ 1, ys), lambdas(xs, ys ^ 1)
            interp_h = lambda key, c: h(F(F(F(lam[0] * vo[0][key][c]) + F(lam[1] * vo[1][key][c])) + F(lam[2] * vo[2][key][c])))
            interp_uv = lambda lm, c: F(F(F(lm[0] * vo[0]["uv"][c]) + F(lm[1] * vo[1]["uv"][c])) + F(lm[2] * vo[2]["uv"][c]))
            with np.errstate(all="ignore"):
                uv = [interp_uv(lam, c) for c in range(2)]
                uvx, uvy = [interp_uv(lam_x, c) for c in range(2)], [interp_uv(lam_y, c) for c in range(2)]
                ddx = [np.where(xs & 1, F(uv[c] - uvx[c]), F(uvx[c] - uv[c])) for c in range(2)]
                ddy = [np.where(ys & 1, F(uv[c] - uvy[c]), F(uvy[c] - uv[c])) for c in range(2)]

                def slot(index, constant):
                    if int(index) == _abi.TEXTURE_NONE:
                        return [np.broadcast_to(h(f32(constant[c])), (Hh, W)) for c in range(4)]
                    t4 = sample_bias(m.textures[int(index)], uv, ddx, ddy, bias)
                    return [h(t4[..., c]) for c in range(4)]
                base = slot(slots[0], mat["base_color_texel"])
                nmap = slot(slots[1], mat["normal_texel"])
                data = slot(slots[2], mat["data_texel"])
                emis = slot(slots[3], mat["emission_texel"])
                col = [interp_h("color", c) for c in range(4)]
                N = [interp_h("normal", c) for c in range(3)]
                T = [interp_h("tangent", c) for c in range(4)]
                tinted = [h(h(base[c] * col[c]) * h(f32(mat["base_color_tint"][c]))) for c in range(4)]
                discard = (int(p["type"]) == _abi.PRIMITIVE_TYPE_CUTOUT) & (tinted[3] <= f32(mat["opacity_threshold"]))
                cr = [h(h(N[1] * T[2]) - h(N[2] * T[1])), h(h(N[2] * T[0]) - h(N[0] * T[2])), h(h(N[0] * T[1]) - h(N[1] * T[0]))]
                B = [h(cr[c] * T[3]) for c in range(3)]
                ns = [h(h(nmap[c] * h(f32(2))) - h(f32(1))) for c in range(3)]
                nout = [h(h(h(ns[0] * T[c]) + h(ns[1] * B[c])) + h(ns[2] * N[c])) for c in range(3)]
                factor = [h(f32(0)), h(f32(mat["roughness_factor"])), h(f32(mat["metalness_factor"])), h(f32(0))]
                dat = [h(data[c] * factor[c]) for c in range(4)]
                em = [h(emis[c] * h(f32(mat["emission_factor"][c]))) for c in range(4)]
            texel = {"color": np.stack([_srgb_encode8(tinted[c]) if c < 3 else _unorm8(tinted[c]) for c in range(4)], -1),
                     "normals": np.stack([x.astype(np.float16).view(np.uint16) for x in nout] + [np.zeros((Hh, W), np.uint16)], -1),
                     "data": np.stack([_unorm8(d) for d in dat], -1),
                     "emission": np.stack([_srgb_encode8(em[c]) if c < 3 else _unorm8(em[c]) for c in range(4)], -1)}
            texel["_tinted"], texel["_normal"] = tinted, N  # half values, for the RSM variant of the fragment stage
            frags.append((cover & ~discard, depth, texel))
    return frags


def raster_gbuffer(m, view, W, Hh):
    """sah_gbuffer_render of mesh.Mesh `m` (every triangle inside the frustum)."""
    Vm = np.array(view.gpu_data.view[:], f32)
    Pm = np.array(view.gpu_data.projection[:], f32)
    frags = [(mask & (depth > 0), depth, texel)
             for (mask, depth, texel) in raster_fragments(m, lambda world: mat_vec(Pm, mat_vec(Vm, world)), W, Hh, view.gpu_data.material_texture_mip_bias)]
    # pass 1: depth (GREATER against the cleared 0); pass 2: EQUAL, every fragment at the settled depth overwrites: the last one stays
    out = {"color": np.zeros((Hh, W, 4), np.uint8), "normals": np.zeros((Hh, W, 4), np.uint16), "data": np.zeros((Hh, W, 4), np.uint8),
           "emission": np.zeros((Hh, W, 4), np.uint8), "depth": np.zeros((Hh, W), f32)}
    out["normals"][...] = np.array([0x3800, 0x3800, 0x3C00, 0], np.uint16)  # clear values gbuffer_phase.cpp:66-87
    for (mask, depth, _) in frags:
        out["depth"] = np.where(mask & (depth > out["depth"]), depth, out["depth"])
    for (mask, depth, texel) in frags:
        hit = mask & (depth == out["depth"])
        for k in out:
            if k != "depth":
                out[k] = np.where(hit[..., None], texel[k], out[k])
    return out



def fd_half(base, n, l, v):
    """Fd(surface, l, v) of brdf.slangi:59-83 in half precision for metalness = roughness = 0 (the RSM variant leaves gbuffer.data 0)."""
    r = h
    one = r(f32(1.0))
    diff = [r(r(base[i] * r(one - r(f32(0.04)))) * r(one - r(f32(0)))) for i in range(3)]
    hv = normalize3([r(v[i] + l[i]) for i in range(3)], r)
    dn = lambda a, b: r(r(r(a[0] * b[0]) + r(a[1] * b[1])) + r(a[2] * b[2]))
    NoV = np.abs(r(dn(n, v) + r(f32(1e-5))))
    NoL = dn(n, l)
    dark = NoL <= 0
    NoL = clamp01(NoL)
    LoH = clamp01(dn(l, hv))
    f90 = r(r(f32(0.5)) + r(r(r(r(f32(2.0)) * r(f32(0))) * LoH) * LoH))
    schlick1 = lambda u: r(one + r(r(f90 - one) * pow5(clamp01(r(one - u)), r)))
    fdv = r(r(schlick1(NoL) * schlick1(NoV)) * r(one / r(f32(3.1415927))))
    return [np.where(dark, f32(0), r(diff[i] * fdv)) for i in range(3)]


def raster_rsm(m, sun, cascades, num_cascades, res):
    """sah_rsm_render: flux (sRGB8), normal * 0.5 + 0.5 (UNORM8) and depth (D16, LESS: the first of equal codes in draw order stays)."""
    flux = np.zeros((num_cascades, res, res, 4), np.uint8)
    normals = np.zeros((num_cascades, res, res, 4), np.uint8)
    normals[...] = np.array([128, 128, 255, 0], np.uint8)  # clear (0.5, 0.5, 1, 0), light_propagation_volume.cpp:596-600
    depth = np.full((num_cascades, res, res), 0xFFFF, np.uint16)
    sd = np.array(sun.direction_and_tan_size[:3], f32)
    l = [h(F(-h(sd[i]))) for i in range(3)]
    for c in range(num_cascades):
        M = np.array(cascades[c].rsm_vp[:], f32)
        for (mask, z, texel) in raster_fragments(m, lambda world: mat_vec(M, world), res, res, 0.0):
            code = np.rint(F(z * f32(65535))).astype(np.int64)
            win = mask & (code < depth[c])
            N = texel["_normal"]
            with np.errstate(all="ignore"):
                fl = fd_half(texel["_tinted"][:3], N, l, N)
                nb = [_unorm8(h(h(N[i] * h(f32(0.5))) + h(f32(0.5)))) for i in range(3)]
            ft = np.stack([_srgb_encode8(fl[i]) for i in range(3)] + [np.full((res, res), 255, np.uint8)], -1)
            nt = np.stack(nb + [np.full((res, res), 255, np.uint8)], -1)
            flux[c] = np.where(win[..., None], ft, flux[c])
            normals[c] = np.where(win[..., None], nt, normals[c])
            depth[c] = np.where(win, code, depth[c]).astype(np.uint16)
    return {"flux": flux, "normals": normals, "depth": depth}


def raster_shadow(m, sun, num_cascades, res):
    """sah_shadow_render: D16 cascades, compare LESS against the cleared 1.0, masked geometry alpha-tested (texture mip bias 0)."""
    out = np.full((num_cascades, res, res), 0xFFFF, np.uint16)
    for c in range(num_cascades):
        M = np.array(sun.cascade_matrices[c][:], f32)
        for (mask, depth, _) in raster_fragments(m, lambda world: mat_vec(M, world), res, res, 0.0):
            code = np.rint(F(depth * f32(65535))).astype(np.int64)
            out[c] = np.where(mask & (code < out[c]), code, out[c]).astype(np.uint16)
    return out

# ---- LPV injection chain behind the RSM (f4): rsm_generate_vpls.comp:44-139, vpl_injection.vert:27-66, vpl_injection.frag:13-52 ----------
def length3_f(v):
    return F(np.sqrt(F(F(F(v[0] * v[0]) + F(v[1] * v[1])) + F(v[2] * v[2]))))


def extract_vpls(rsm, mats, cascade, cell_size):
    """One invocation per 2x2 RSM texels of layer `cascade`; the list in ascending invocation index (the order include/sah_hip.h fixes).
    Returns the packed lights (n, 4) uint32."""
    res = rsm["depth"].shape[1]
    half = res // 2
    inv_vp = np.array(mats.inverse_rsm_vp[:], f32)
    w2c = np.array(mats.world_to_cascade[:], f32)
    gy, gx = np.meshgrid(np.arange(half), np.arange(half), indexing="ij")
    lut = srgb_lut()

    def load(dx, dy):
        x, y = gx * 2 + dx, gy * 2 + dy
        depth = F(rsm["depth"][cascade][y, x].astype(f32) / f32(65535.0))
        tx, ty = F(F(x.astype(f32) + f32(0.5)) / f32(res)), F(F(y.astype(f32) + f32(0.5)) / f32(res))
        ws = mat_vec(inv_vp, [F(F(tx * f32(2)) - f32(1)), F(F(ty * f32(2)) - f32(1)), depth, np.ones_like(depth)])
        with np.errstate(all="ignore"):
            pos = [F(ws[i] / ws[3]) for i in range(3)]
        col = [lut[rsm["flux"][cascade][y, x, i]] for i in range(3)]
        nrm = [F(F(F(rsm["normals"][cascade][y, x, i].astype(f32) / f32(255.0)) * f32(2)) - f32(1)) for i in range(3)]
        return pos, col, nrm

    def cell_of(pos):
        cp = mat_vec(w2c, [pos[0], pos[1], pos[2], np.ones_like(pos[0])])
        side = F(f32(cell_size) * f32(32.0))
        with np.errstate(all="ignore"):
            return [np.rint(F(F(cp[0] + f32(cascade)) * side)), np.rint(F(cp[1] * side)), np.rint(F(cp[2] * side))]
    texels = [load(dx, dy) for dy in range(2) for dx in range(2)]   # loop order y outer, x inner
    cells = [cell_of(t[0]) for t in texels]
    brightest = np.zeros((half, half), f32)
    chosen = [np.zeros((half, half), f32) for _ in range(3)]
    for (pos, col, nrm), cell in zip(texels, cells):
        luma = F(F(F(col[0] * f32(0.2126)) + F(col[1] * f32(0.7152))) + F(col[2] * f32(0.0722)))
        better = luma > brightest
        brightest = np.where(better, luma, brightest)
        chosen = [np.where(better, cell[i], chosen[i]) for i in range(3)]
    acc = {k: [np.zeros((half, half), f32) for _ in range(3)] for k in ("pos", "col", "nrm")}
    n = np.zeros((half, half), f32)
    for (pos, col, nrm), cell in zip(texels, cells):
        with np.errstate(all="ignore"):
            d = [F(cell[i] - chosen[i]) for i in range(3)]
            near = F(F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2])) < 3
        for key, val in (("pos", pos), ("col", col), ("nrm", nrm)):
            acc[key] = [np.where(near, F(acc[key][i] + val[i]), acc[key][i]) for i in range(3)]
        n = np.where(near, F(n + f32(1)), n)
    with np.errstate(all="ignore"):
        has = n > 0
        pos = [np.where(has, F(acc["pos"][i] / n), acc["pos"][i]) for i in range(3)]
        col = [np.where(has, F(acc["col"][i] / n), acc["col"][i]) for i in range(3)]
        nav = [F(acc["nrm"][i] / n) for i in range(3)]
        nn = normalize3(nav)
        nrm = [np.where(has, nn[i], acc["nrm"][i]) for i in range(3)]
        keep = (length3_f(col) > 0) & (length3_f(nrm) > 0)
    half16 = lambda a: a.astype(np.float16).view(np.uint16).astype(np.uint32)
    snorm = lambda a: (np.rint(F(np.fmin(np.fmax(a, f32(-1)), f32(1)) * f32(127))).astype(np.int64) & 0xFF).astype(np.uint32)
    with np.errstate(all="ignore"):
        data = np.stack([half16(pos[0]) | (half16(pos[1]) << 16), half16(pos[2]) | (half16(col[0]) << 16), half16(col[1]) | (half16(col[2]) << 16),
                         snorm(nrm[0]) | (snorm(nrm[1]) << 8) | (snorm(nrm[2]) << 16)], axis=-1)
    return data[keep]  # boolean indexing walks the groups in row-major order = ascending invocation index


def inject_vpls(vpls, mats, cascade, num_cascades, vols):
    """Points of size 1 blended ONE / ONE into the three RGBA16F volumes (fp16 arrays (32, 32, 32 * cascades, 4), modified in place), in
    list order with one rounding to half per addition."""
    D, Hh, W = vols[0].shape[:3]
    w2c = np.array(mats.world_to_cascade[:], f32)
    mixf = lambda x, y, a: F(F(x * F(f32(1) - a)) + F(y * a))
    for p in vpls:
        hf = lambda bits: f32(np.array([bits & 0xFFFF], np.uint16).view(np.float16)[0])
        pos = [hf(int(p[0])), hf(int(p[0]) >> 16), hf(int(p[1]))]
        col = [hf(int(p[1]) >> 16), hf(int(p[2])), hf(int(p[2]) >> 16)]
        sn = lambda b: max(f32(np.int8(np.uint8(b & 0xFF))) / f32(127.0), f32(-1.0))
        with np.errstate(all="ignore"):
            nrm = normalize3([F(sn(int(p[3]))), F(sn(int(p[3]) >> 8)), F(sn(int(p[3]) >> 16))])
            cp = mat_vec(w2c, [pos[0], pos[1], pos[2], f32(1)])
            px = F(F(cp[0] + f32(cascade)) / f32(num_cascades))
            ndc = [F(F(px * f32(2)) - f32(1)), F(F(cp[1] * f32(2)) - f32(1))]
            layer = F(cp[2] * f32(32))
            if length3_f(nrm) < 1 or length3_f(col) == 0:
                continue
            xf, yf = F(F(ndc[0] * f32(W * 0.5)) + f32(W * 0.5)), F(F(ndc[1] * f32(Hh * 0.5)) + f32(Hh * 0.5))
            if not (0 <= xf < W and 0 <= yf < Hh and -1 < layer < D):
                continue
            cx, cy, cz = int(np.floor(xf)), int(np.floor(yf)), int(layer)
            sc = [F(F(c * f32(1024)) / f32(16384)) for c in col]
            # rgb2hsv :13-22
            Kx, Ky, Kz, Kw = f32(0), F(f32(-1) / f32(3)), F(f32(2) / f32(3)), f32(-1)
            step = lambda edge, x: f32(0) if x < edge else f32(1)
            s1 = step(sc[2], sc[1])
            P = [mixf(sc[2], sc[1], s1), mixf(sc[1], sc[2], s1), mixf(Kw, Kx, s1), mixf(Kz, Ky, s1)]
            s2 = step(P[0], sc[0])
            Q = [mixf(P[0], sc[0], s2), mixf(P[1], P[1], s2), mixf(P[3], P[2], s2), mixf(sc[0], P[0], s2)]
            d = F(Q[0] - np.fmin(Q[3], Q[1]))
            e = f32(1.0e-10)
            hsv = [np.abs(F(Q[2] + F(F(Q[3] - Q[1]) / F(F(f32(6) * d) + e)))), F(d / F(Q[0] + e)), Q[0]]
            hsv[1] = F(hsv[1] * f32(2))
            # hsv2rgb :24-29
            K = [f32(1), F(f32(2) / f32(3)), F(f32(1) / f32(3)), f32(3)]
            corrected = []
            for k in range(3):
                t = F(hsv[0] + K[k])
                pk = np.abs(F(F(F(t - np.floor(t)) * f32(6)) - K[3]))
                corrected.append(F(hsv[2] * mixf(K[0], np.fmin(np.fmax(F(pk - K[0]), f32(0)), f32(1)), hsv[1])))
            c0, c1 = f32(0.886226925), f32(1.02332671)
            sh = [c0, F(F(-c1) * nrm[1]), F(c1 * nrm[2]), F(F(-c1) * nrm[0])]
            for ch in range(3):
                for k in range(4):
                    src = F(F(sh[k] * corrected[ch]) / f32(3.1415927))
                    vols[ch][cz, cy, cx, k] = np.float16(F(f32(vols[ch][cz, cy, cx, k]) + src))
    return vols


# ---- sky LUT generators (f3): sky/common.glsl:8-110, transmittance_lut.comp, multiscattering_lut.comp, sky_view_lut.comp ----------------
# GLSL fp32, every operator rounded, constant expressions too; exp / sin / cos / acos / pow through fp64 rounded to fp32; rgba16f stores.
SKY_PI = f32(3.14159265358)
SKY_GROUND, SKY_ATMOSPHERE = f32(6.360), f32(6.460)


def _t64(fn, x):
    return fn(np.asarray(x, np.float64)).astype(f32)


def _len3(v):
    return F(np.sqrt(F(F(F(v[0] * v[0]) + F(v[1] * v[1])) + F(v[2] * v[2]))))


def _dot3(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def sky_ray_sphere(ro, rd, rad):
    """common.glsl:76-90"""
    b = _dot3(ro, rd)
    c = F(_dot3(ro, ro) - F(rad * rad))
    discr = F(F(b * b) - c)
    with np.errstate(invalid="ignore"):
        root = F(np.sqrt(discr))
        t = np.where(discr > F(b * b), F(F(-b) + root), F(F(-b) - root))
    t = np.where(discr < 0, f32(-1), t)
    return np.where((c > 0) & (b > 0), f32(-1), t)


def sky_scattering_values(pos):
    """common.glsl:51-70 -> rayleigh scattering (3), mie scattering, extinction (3)"""
    alt = F(np.maximum(f32(0), F(_len3(pos) - SKY_GROUND)) * f32(1000.0))
    ray_d = _t64(np.exp, F(F(-alt) / f32(8.0)))
    mie_d = _t64(np.exp, F(F(-alt) / f32(1.2)))
    ray = [F(f32(k) * ray_d) for k in (6.6, 12.3, 29.4)]
    ray_abs = F(f32(0.0) * ray_d)
    mie = F(f32(3.996) * mie_d)
    mie_abs = F(f32(4.4) * mie_d)
    oz_f = np.maximum(f32(0.0), F(f32(1.0) - F(np.abs(F(alt - f32(25.0))) / f32(15.0))))
    ozone = [F(f32(k) * oz_f) for k in (2.26, 1.54, 0.0)]
    ext = [F(F(F(F(ray[i] + ray_abs) + mie) + mie_abs) + ozone[i]) for i in range(3)]
    return ray, mie, ext


def sky_mie_phase(cos_t):
    g = f32(0.8)
    scale = F(f32(3.0) / F(f32(8.0) * SKY_PI))
    num = F(F(f32(1.0) - F(g * g)) * F(f32(1.0) + F(cos_t * cos_t)))
    base = F(F(f32(1.0) + F(g * g)) - F(F(f32(2.0) * g) * cos_t))
    denom = F(F(f32(2.0) + F(g * g)) * np.power(base.astype(np.float64), np.float64(f32(1.5))).astype(f32))
    return F(F(scale * num) / denom)


def sky_rayleigh_phase(cos_t):
    k = F(f32(3.0) / F(f32(16.0) * SKY_PI))
    return F(k * F(f32(1.0) + F(cos_t * cos_t)))


def sky_lut_value(lut, pos, sun):
    """getValFromTLUT / getValFromMultiScattLUT (common.glsl:94-110): linear, REPEAT"""
    height = _len3(pos)
    up = [F(p / height) for p in pos]
    cz = _dot3(sun, up)
    u = np.minimum(np.maximum(F(f32(0.5) + F(f32(0.5) * cz)), f32(0)), f32(1))
    v = np.maximum(f32(0.0), np.minimum(f32(1.0), F(F(height - SKY_GROUND) / F(SKY_ATMOSPHERE - SKY_GROUND))))
    t = bilinear_repeat(lut, u, v)
    return [t[..., i] for i in range(3)]


def sky_sun_dir_of_texel(xs, ys, w, hgt):
    u, v = F(xs.astype(f32) / f32(w)), F(ys.astype(f32) / f32(hgt))
    cos_t = F(F(f32(2.0) * u) - f32(1.0))
    theta = _t64(np.arccos, np.minimum(np.maximum(cos_t, f32(-1)), f32(1)))
    height = F(F(SKY_GROUND * F(f32(1) - v)) + F(SKY_ATMOSPHERE * v))  # mix(ground, atmosphere, v)
    pos = [np.zeros_like(height), height, np.zeros_like(height)]
    sd = [np.zeros_like(height), cos_t, F(-_t64(np.sin, theta))]
    ln = _len3(sd)
    inv = F(f32(1) / ln)   # normalize = v * (1 / length)
    return pos, [F(c * inv) for c in sd]


def sky_transmittance_lut():
    ys, xs = np.meshgrid(np.arange(64), np.arange(256), indexing="ij")
    pos, sun = sky_sun_dir_of_texel(xs, ys, 256, 64)
    blocked = sky_ray_sphere(pos, sun, SKY_GROUND) > 0
    atmo = sky_ray_sphere(pos, sun, SKY_ATMOSPHERE)
    t = np.zeros_like(atmo)
    tr = [np.ones_like(atmo) for _ in range(3)]
    for i in range(40):
        new_t = F(F(F(f32(i) + f32(0.3)) / f32(40.0)) * atmo)
        dt = F(new_t - t)
        t = new_t
        npos = [F(pos[k] + F(t * sun[k])) for k in range(3)]
        _, _, ext = sky_scattering_values(npos)
        tr = [F(tr[k] * _t64(np.exp, F(F(-dt) * ext[k]))) for k in range(3)]
    tr = [np.where(blocked, f32(0), c) for c in tr]
    return np.stack(tr + [np.ones_like(atmo)], -1).astype(np.float16)


def sky_multiscattering_lut(tlut16):
    tlut = tlut16.astype(f32)
    ys, xs = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    pos, sun = sky_sun_dir_of_texel(xs, ys, 32, 32)
    lum_total = [np.zeros(xs.shape, f32) for _ in range(3)]
    fms = [np.zeros(xs.shape, f32) for _ in range(3)]
    inv_samples = F(f32(1.0) / f32(64))
    albedo = f32(0.3)
    for i in range(8):
        for j in range(8):
            theta = F(F(SKY_PI * F(f32(i) + f32(0.5))) / f32(8))
            phi = _t64(np.arccos, np.minimum(np.maximum(F(f32(1.0) - F(F(f32(2.0) * F(f32(j) + f32(0.5))) / f32(8))), f32(-1)), f32(1)))
            cp, sp, ct, st = _t64(np.cos, phi), _t64(np.sin, phi), _t64(np.cos, theta), _t64(np.sin, theta)
            ray = [np.full(xs.shape, F(sp * st)), np.full(xs.shape, cp), np.full(xs.shape, F(sp * ct))]
            atmo = sky_ray_sphere(pos, ray, SKY_ATMOSPHERE)
            ground = sky_ray_sphere(pos, ray, SKY_GROUND)
            t_max = np.where(ground > 0, ground, atmo)
            cos_t = _dot3(ray, sun)
            mie_p, ray_p = sky_mie_phase(cos_t), sky_rayleigh_phase(F(-cos_t))
            lum = [np.zeros(xs.shape, f32) for _ in range(3)]
            lum_factor = [np.zeros(xs.shape, f32) for _ in range(3)]
            tr = [np.ones(xs.shape, f32) for _ in range(3)]
            t = np.zeros(xs.shape, f32)
            for step in range(20):
                new_t = F(F(F(f32(step) + f32(0.3)) / f32(20.0)) * t_max)
                dt = F(new_t - t)
                t = new_t
                npos = [F(pos[k] + F(t * ray[k])) for k in range(3)]
                rs, ms, ext = sky_scattering_values(npos)
                st_k = [_t64(np.exp, F(F(-dt) * ext[k])) for k in range(3)]
                sun_tr = sky_lut_value(tlut, npos, sun)
                for k in range(3):
                    no_phase = F(rs[k] + ms)
                    sc_f = F(F(no_phase - F(no_phase * st_k[k])) / ext[k])
                    lum_factor[k] = F(lum_factor[k] + F(tr[k] * sc_f))
                    in_sc = F(F(F(rs[k] * ray_p) + F(ms * mie_p)) * sun_tr[k])
                    integral = F(F(in_sc - F(in_sc * st_k[k])) / ext[k])
                    lum[k] = F(lum[k] + F(integral * tr[k]))
                    tr[k] = F(tr[k] * st_k[k])
            hit = [F(pos[k] + F(ground * ray[k])) for k in range(3)]
            ln = _len3(hit)
            with np.errstate(all="ignore"):
                inv = F(f32(1) / ln)
                hit_n = [F(F(hit[k] * inv) * SKY_GROUND) for k in range(3)]
                ground_tr = sky_lut_value(tlut, hit_n, sun)
            lit = (ground > 0) & (_dot3(pos, sun) > 0)
            for k in range(3):
                lum[k] = np.where(lit, F(lum[k] + F(F(tr[k] * albedo) * ground_tr[k])), lum[k])
                fms[k] = F(fms[k] + F(lum_factor[k] * inv_samples))
                lum_total[k] = F(lum_total[k] + F(lum[k] * inv_samples))
    psi = [F(lum_total[k] / F(f32(1.0) - fms[k])) for k in range(3)]
    return np.stack(psi + [np.ones(xs.shape, f32)], -1).astype(np.float16)


def sky_view_lut(tlut16, mslut16, light_dir):
    tlut, mslut = tlut16.astype(f32), mslut16.astype(f32)
    ys, xs = np.meshgrid(np.arange(200), np.arange(200), indexing="ij")
    u, v = F(xs.astype(f32) / f32(200)), F(ys.astype(f32) / f32(200))
    azimuth = F(F(F(u - f32(0.5)) * f32(2.0)) * SKY_PI)
    c_lo, c_hi = F(f32(1.0) - F(f32(2.0) * v)), F(F(v * f32(2.0)) - f32(1.0))
    adj_v = np.where(v < 0.5, F(F(-c_lo) * c_lo), F(c_hi * c_hi))
    view_pos = [f32(0), F(SKY_GROUND + f32(0.0002)), f32(0)]
    height = _len3(view_pos)
    up = [F(p / height) for p in view_pos]
    ratio = F(F(np.sqrt(F(F(height * height) - F(SKY_GROUND * SKY_GROUND)))) / height)
    horizon = F(_t64(np.arccos, np.minimum(np.maximum(ratio, f32(-1)), f32(1))) - F(f32(0.5) * SKY_PI))
    altitude = F(F(F(adj_v * f32(0.5)) * SKY_PI) - horizon)
    cos_alt = _t64(np.cos, altitude)
    ray = [F(cos_alt * _t64(np.sin, azimuth)), _t64(np.sin, altitude), F(F(-cos_alt) * _t64(np.cos, azimuth))]
    ld = [f32(c) for c in light_dir]
    sun_alt = F(F(f32(0.5) * SKY_PI) - _t64(np.arccos, _dot3([F(-c) for c in ld], up)))
    sun = [np.zeros(xs.shape, f32), np.full(xs.shape, _t64(np.sin, sun_alt)), np.full(xs.shape, F(-_t64(np.cos, sun_alt)))]
    vp = [np.full(xs.shape, c) for c in view_pos]
    atmo = sky_ray_sphere(vp, ray, SKY_ATMOSPHERE)
    ground = sky_ray_sphere(vp, ray, SKY_GROUND)
    t_max = np.where(ground < 0, atmo, ground)
    # raymarchScattering :17-56
    cos_t = _dot3(ray, sun)
    mie_p, ray_p = sky_mie_phase(cos_t), sky_rayleigh_phase(F(-cos_t))
    lum = [np.zeros(xs.shape, f32) for _ in range(3)]
    tr = [np.ones(xs.shape, f32) for _ in range(3)]
    t = np.zeros(xs.shape, f32)
    for i in range(32):
        new_t = F(F(F(f32(i) + f32(0.3)) / f32(32.0)) * t_max)
        dt = F(new_t - t)
        t = new_t
        npos = [F(vp[k] + F(t * ray[k])) for k in range(3)]
        rs, ms, ext = sky_scattering_values(npos)
        st_k = [_t64(np.exp, F(F(-dt) * ext[k])) for k in range(3)]
        sun_tr = sky_lut_value(tlut, npos, sun)
        psi = sky_lut_value(mslut, npos, sun)
        for k in range(3):
            ray_in = F(rs[k] * F(F(ray_p * sun_tr[k]) + psi[k]))
            mie_in = F(ms * F(F(mie_p * sun_tr[k]) + psi[k]))
            in_sc = F(ray_in + mie_in)
            integral = F(F(in_sc - F(in_sc * st_k[k])) / ext[k])
            lum[k] = F(lum[k] + F(integral * tr[k]))
            tr[k] = F(tr[k] * st_k[k])
    return np.stack(lum + [np.ones(xs.shape, f32)], -1).astype(np.float16)


def probe_copy(src, dst, movement):
    """copy_cascades.comp.slang:22-99 — one invocation per probe cell of the 32^3 grid, in ascending linear invocation index (x fastest):
    cells whose source (cell - (int3)movement[cascade], cascade = y / 8) lies in the same cascade copy their blocks, the others are
    initialised — with the depth clear at LIGHT-CACHE offsets (:39-45), which lands on other cells' depth blocks: the later write stays."""
    mv = [[int(np.trunc(c)) for c in row] for row in movement]
    for z in range(32):
        for y in range(32):
            cas = y // 8
            for x in range(32):
                sx, sy, sz = x - mv[cas][0], y - mv[cas][1], z - mv[cas][2]
                if 0 <= sx < 32 and 8 * cas <= sy < 8 * (cas + 1) and 0 <= sz < 32:
                    dst["rtgi"][z, y * 8:y * 8 + 8, x * 7:x * 7 + 7] = src["rtgi"][sz, sy * 8:sy * 8 + 8, sx * 7:sx * 7 + 7]
                    dst["light_cache"][z, y * 13:y * 13 + 13, x * 13:x * 13 + 13] = src["light_cache"][sz, sy * 13:sy * 13 + 13, sx * 13:sx * 13 + 13]
                    dst["depth"][z, y * 12:y * 12 + 12, x * 12:x * 12 + 12] = src["depth"][sz, sy * 12:sy * 12 + 12, sx * 12:sx * 12 + 12]
                    dst["average"][z, y, x] = src["average"][sz, sy, sx]
                    v = np.float16(f32(src["validity"][sz, sy, sx]) / f32(255))            # Texture2DArray<half> load of an R8_UNORM texel
                    dst["validity"][z, y, x] = _unorm8(f32(v))                             # and the store back
                else:
                    dst["rtgi"][z, y * 8:y * 8 + 8, x * 7:x * 7 + 7] = 0
                    dst["light_cache"][z, y * 13:y * 13 + 13, x * 13:x * 13 + 13] = 0
                    blk = dst["depth"][z, y * 13:y * 13 + 12, x * 13:x * 13 + 12]          # clipped at the atlas edge: stores outside are dropped
                    blk[...] = 0
                    dst["average"][z, y, x] = 0
                    dst["validity"][z, y, x] = 255
    return dst


# ---- ray tracing (f4, first slice): the hit rules of include/sah_hip.h ("ray tracing"), RTAO (ao/rtao.comp.slang:54-102), the shadow rays
# of the RT-mode sun (lighting/directional_light.rt.slang:91-125) and the occlusion any-hit stage (materials/gltf_basic_pbr.slang:291-318).
# Every ray against every triangle, vectorised over rays; written from the header and the shader text, not from oracle/rt.cpp.
def rt_world_triangles(arrays):
    """world-space triangles of every primitive: vertex = model * (p, 1), rows ((m0 x + m1 y) + m2 z) + m3; non-finite ones left out"""
    tris = []
    for p, prim in enumerate(arrays["primitives"]):
        mdl = prim["model"].astype(f32)
        for t in range(int(prim["index_count"]) // 3):
            idx = arrays["indices"][int(prim["first_index"]) + 3 * t:int(prim["first_index"]) + 3 * t + 3].astype(np.int64) + int(prim["vertex_offset"])
            v = np.zeros((3, 3), f32)
            for k in range(3):
                x, y, z = arrays["positions"][idx[k]].astype(f32)
                for c in range(3):
                    v[k, c] = F(F(F(F(mdl[c] * x) + F(mdl[4 + c] * y)) + F(mdl[8 + c] * z)) + mdl[12 + c])
            if np.isfinite(v).all():
                tris.append({"v": v, "primitive": p, "triangle": t, "vertices": idx, "cutout": int(prim["type"]) == _abi.PRIMITIVE_TYPE_CUTOUT})
    S = max([f32(0)] + [np.abs(t["v"]).max() for t in tris])
    return tris, F(f32(S) * f32(2.0 ** -16))


def _pick(v, k):  # v: (N, 3), k: (N,) -> (N,)
    return np.take_along_axis(v, k[:, None], axis=1)[:, 0]


class _RtRays:
    """per-ray set-up of the watertight test (make_ray): reciprocal direction, axis permutation, shear"""

    def __init__(self, o, d, tmin, tmax):
        n = o.shape[0]
        self.n, self.o, self.d = n, o, d
        self.tmin, self.tmax = np.broadcast_to(F(tmin), (n,)).astype(f32), np.broadcast_to(F(tmax), (n,)).astype(f32)
        self.finite = np.isfinite(o).all(axis=1) & np.isfinite(d).all(axis=1)
        with np.errstate(all="ignore"):
            self.inv = F(f32(1) / d)
            ad = np.abs(d)
            kz = np.zeros(n, np.int64)
            am = ad[:, 0].copy()
            kz = np.where(ad[:, 1] > am, 1, kz)
            am = np.where(ad[:, 1] > am, ad[:, 1], am)
            kz = np.where(ad[:, 2] > am, 2, kz)
            kx = (kz + 1) % 3
            ky = (kx + 1) % 3
            neg = _pick(d, kz) < 0
            self.kx, self.ky, self.kz = np.where(neg, ky, kx), np.where(neg, kx, ky), kz
            dz = _pick(d, kz)
            self.Sx, self.Sy, self.Sz = F(_pick(d, self.kx) / dz), F(_pick(d, self.ky) / dz), F(f32(1) / dz)

    def candidates(self, t, pad):
        """(candidate mask, t, b1, b2, front) of triangle record `t`: slab test of its padded box, then Woop / Benthin / Wald"""
        o, kx, ky, kz, Sx, Sy, Sz = self.o, self.kx, self.ky, self.kz, self.Sx, self.Sy, self.Sz
        with np.errstate(all="ignore"):
            v = t["v"]
            lo, hi = F(v.min(axis=0) - pad), F(v.max(axis=0) + pad)
            tn, tf = self.tmin.copy(), self.tmax.copy()
            for c in range(3):
                t0, t1 = F(F(lo[c] - o[:, c]) * self.inv[:, c]), F(F(hi[c] - o[:, c]) * self.inv[:, c])
                tn = np.fmax(tn, np.fmin(t0, t1))
                tf = np.fmin(tf, np.fmax(t0, t1))
            box = tn <= tf
            A, B, C = F(v[0][None, :] - o), F(v[1][None, :] - o), F(v[2][None, :] - o)
            Akz, Bkz, Ckz = _pick(A, kz), _pick(B, kz), _pick(C, kz)
            Ax, Ay = F(_pick(A, kx) - F(Sx * Akz)), F(_pick(A, ky) - F(Sy * Akz))
            Bx, By = F(_pick(B, kx) - F(Sx * Bkz)), F(_pick(B, ky) - F(Sy * Bkz))
            Cx, Cy = F(_pick(C, kx) - F(Sx * Ckz)), F(_pick(C, ky) - F(Sy * Ckz))
            U, V, Wd = F(F(Cx * By) - F(Cy * Bx)), F(F(Ax * Cy) - F(Ay * Cx)), F(F(Bx * Ay) - F(By * Ax))
            zero = (U == 0) | (V == 0) | (Wd == 0)
            d64 = lambda a, b, c2, e: (a.astype(np.float64) * b.astype(np.float64) - c2.astype(np.float64) * e.astype(np.float64)).astype(f32)
            U, V, Wd = np.where(zero, d64(Cx, By, Cy, Bx), U), np.where(zero, d64(Ax, Cy, Ay, Cx), V), np.where(zero, d64(Bx, Ay, By, Ax), Wd)
            mixed = ((U < 0) | (V < 0) | (Wd < 0)) & ((U > 0) | (V > 0) | (Wd > 0))
            det = F(F(U + V) + Wd)
            T = F(F(F(U * F(Sz * Akz)) + F(V * F(Sz * Bkz))) + F(Wd * F(Sz * Ckz)))
            tt = F(T / det)
            cand = self.finite & box & ~mixed & (det != 0) & (tt > self.tmin) & (tt < self.tmax)
            return cand, tt, F(V / det), F(Wd / det), det > 0


def rt_any_hit(arrays, tris, pad, o, d, tmin, tmax, cull_non_opaque, cull_front=False):
    """o, d: (N, 3) fp32; returns (N,) bool: is there an accepted candidate"""
    rays = _RtRays(o, d, tmin, tmax)
    hit = np.zeros(rays.n, bool)
    for t in tris:
        if cull_non_opaque and t["cutout"]:
            continue
        cand, tt, b1, b2, front = rays.candidates(t, pad)
        if cull_front:
            cand = cand & ~front
        if t["cutout"] and cand.any():
            cand = cand & rt_cutout_accepts(arrays, t, b1, b2)
        hit |= cand
    return hit


def rt_closest_hit(arrays, tris, pad, o, d, tmin, tmax, bounce=False):
    """RAY_FLAG_NONE (bounce: RAY_FLAG_CULL_NON_OPAQUE | RAY_FLAG_CULL_BACK_FACING_TRIANGLES, gltf_basic_pbr.slang:498-506): per ray the accepted
    candidate of smallest t, ties to the smallest (primitive, triangle).  Returns (index into tris or -1, t, b1, b2, front)."""
    rays = _RtRays(o, d, tmin, tmax)
    n = rays.n
    best = np.full(n, -1, np.int64)
    bt, bb1, bb2 = np.zeros(n, f32), np.zeros(n, f32), np.zeros(n, f32)
    bfront = np.zeros(n, bool)
    bprim, btri = np.full(n, 1 << 40, np.int64), np.full(n, 1 << 40, np.int64)
    for i, t in enumerate(tris):
        if bounce and t["cutout"]:
            continue
        cand, tt, b1, b2, front = rays.candidates(t, pad)
        if bounce:
            cand = cand & front
        better = cand & ((best < 0) | (tt < bt) | ((tt == bt) & ((t["primitive"] < bprim) | ((t["primitive"] == bprim) & (t["triangle"] < btri)))))
        if t["cutout"] and better.any():
            better = better & rt_cutout_accepts(arrays, t, b1, b2)
        best = np.where(better, i, best)
        bt, bb1, bb2 = np.where(better, tt, bt), np.where(better, b1, bb1), np.where(better, b2, bb2)
        bfront = np.where(better, front, bfront)
        bprim, btri = np.where(better, t["primitive"], bprim), np.where(better, t["triangle"], btri)
    return best, bt, bb1, bb2, bfront


def rt_cutout_accepts(arrays, t, b1, b2):
    """any-hit stage of the occlusion hit group (gltf_basic_pbr.slang:291-318), vectorised over the candidate rays"""
    prim = arrays["primitives"][t["primitive"]]
    mat = arrays["materials"][int(prim["material"])]
    vd = arrays["vertex_data"][t["vertices"]]
    b0 = F(F(f32(1) - b1) - b2)
    uv = [F(F(F(b0 * vd["texcoord"][0][k]) + F(b1 * vd["texcoord"][1][k])) + F(b2 * vd["texcoord"][2][k])) for k in range(2)]
    ua = [h(h(f32(int(c) >> 24)) / h(255.0)) for c in vd["color"]]                       # unpackUnorm4x8ToHalf(v.color).w
    ca = F(F(F(b0 * ua[0]) + F(b1 * ua[1])) + F(b2 * ua[2]))                             # float * half4, summed in fp32
    with np.errstate(invalid="ignore"):
        byte = np.where(h(ca) * f32(255) > 0, np.floor(h(h(ca) * h(255.0))), 0).astype(np.int64) & 0xff  # packUnorm4x8: (uint)(value.w * 255.h)
    colour_a = h(h(byte.astype(f32)) / h(255.0))
    texel_a = np.full(b1.shape, f32(mat["base_color_texel"][3]), f32)
    mt = arrays.get("material_textures")
    if mt is not None and len(arrays.get("textures", [])):
        ti = int(mt[int(prim["material"])][0])
        if ti != _abi.TEXTURE_NONE:
            texel_a = sample_bias(arrays["textures"][ti], uv, None, None, 0.0, explicit_lod=0.0)[..., 3]
    alpha = F(F(texel_a * f32(mat["base_color_tint"][3])) * colour_a)
    return ~(alpha <= f32(mat["opacity_threshold"]))


def rt_world_position(view, W, Hh, depth):
    """get_worldspace_position (rtao.comp.slang:27-36): (pixel + 0.5) / render_resolution, every component divided by w"""
    ys, xs = np.meshgrid(np.arange(Hh, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    with np.errstate(all="ignore"):
        tx, ty = F(F(xs + f32(0.5)) / f32(view.render_resolution[0])), F(F(ys + f32(0.5)) / f32(view.render_resolution[1]))
        ndc = [F(F(tx * f32(2)) - f32(1)), F(F(ty * f32(2)) - f32(1)), F(depth), np.ones_like(tx)]
        vs = mat_vec(np.array(view.inverse_projection[:], f32), ndc)
        vs = [F(vs[0] / vs[3]), F(vs[1] / vs[3]), F(vs[2] / vs[3]), F(vs[3] / vs[3])]
        ws = mat_vec(np.array(view.inverse_view[:], f32), vs)
    return np.stack(ws[:3], axis=-1)


def rt_noise(noise, xs, ys):
    t = noise[ys, xs].astype(f32)
    c = [F(F(F(t[..., k] / f32(255)) * f32(2)) - f32(1)) for k in range(3)]
    return np.stack(normalize3(c), axis=-1)


def rtao(arrays, view, depth, normals16, noise, spp, radius):
    Hh, W = depth.shape
    tris, pad = rt_world_triangles(arrays)
    pos = rt_world_position(view, W, Hh, depth).reshape(-1, 3)
    nh = normals16.view(np.float16).astype(f32)[..., :3]
    with np.errstate(all="ignore"):
        n = np.stack(normalize3([nh[..., 0], nh[..., 1], nh[..., 2]], rnd=h), axis=-1).reshape(-1, 3)
        ys, xs = np.meshgrid(np.arange(Hh), np.arange(W), indexing="ij")
        d = rt_noise(noise, xs % noise.shape[1], ys % noise.shape[0]).reshape(-1, 3)
        flip = F(F(F(d[:, 0] * n[:, 0]) + F(d[:, 1] * n[:, 1])) + F(d[:, 2] * n[:, 2])) < 0
        d = np.where(flip[:, None], F(d * f32(-1)), d)
    hit = rt_any_hit(arrays, tris, pad, pos, d, 0.01, radius, cull_non_opaque=True)
    ao = np.full(W * Hh, f32(spp), f32)
    for _ in range(spp):  # the same ray every time (one noise texel per pixel)
        ao = np.where(hit, F(ao - f32(1)), ao)
    with np.errstate(all="ignore"):
        return F(ao / f32(spp)).reshape(Hh, W)


def sun_shadow_mask(arrays, view, sun, depth, normals16, noise):
    Hh, W = depth.shape
    tris, pad = rt_world_triangles(arrays)
    L = normalize3([F(-f32(sun.direction_and_tan_size[k])) for k in range(3)])
    nh = normals16.view(np.float16).astype(f32)[..., :3]
    with np.errstate(all="ignore"):
        n = normalize3([nh[..., 0], nh[..., 1], nh[..., 2]], rnd=h)
        ndotl = h(np.fmin(np.fmax(F(F(F(L[0] * n[0]) + F(L[1] * n[1])) + F(L[2] * n[2])), f32(0)), f32(1)))
    traced = ((depth != 0) & (ndotl > 0)).reshape(-1)
    pos = rt_world_position(view, W, Hh, depth).reshape(-1, 3)
    ys, xs = np.meshgrid(np.arange(Hh, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    shadow = np.zeros(W * Hh, f32)
    phi = f32(1.618033988749895)
    i = 0
    while f32(i) < f32(sun.num_shadow_samples):
        q = F(f32(i) / phi)
        fx, fy = F(F(f32(2) + q) - np.floor(F(f32(2) + q))), F(F(f32(3) + q) - np.floor(F(f32(3) + q)))
        offx, offy = np.rint(F(fx * f32(128))), np.rint(F(fy * f32(128)))  # round half to even
        nx, ny = F(xs + offx).astype(np.int64) % 128, F(ys + offy).astype(np.int64) % 128
        nz = rt_noise(noise, nx, ny).reshape(-1, 3)
        with np.errstate(all="ignore"):
            dirv = np.stack(normalize3([F(L[k] + F(nz[:, k] * f32(sun.direction_and_tan_size[3]))) for k in range(3)]), axis=-1)
        hit = rt_any_hit(arrays, tris, pad, pos, dirv, 0.01, 100000.0, cull_non_opaque=False)
        shadow = F(shadow + np.where(hit, f32(0), f32(1)))
        i += 1
    with np.errstate(all="ignore"):
        mask = F(shadow / f32(sun.num_shadow_samples))
    return np.where(traced, mask, f32(1)).astype(f32).reshape(Hh, W)


def fd_general(base, n, rough, metal, l, v, rnd):
    """Fd(surface, l, v) of brdf.slangi:58-82 / brdf.glsl:65-89 (the diffuse half of brdf() above)"""
    r = rnd
    one = r(f32(1.0))
    diff = [r(r(base[i] * r(one - r(f32(0.04)))) * r(one - metal)) for i in range(3)]
    hv = normalize3([r(v[i] + l[i]) for i in range(3)], r)
    dn = lambda a, b: r(r(r(a[0] * b[0]) + r(a[1] * b[1])) + r(a[2] * b[2]))
    NoV = np.abs(r(dn(n, v) + r(f32(1e-5))))
    NoL = dn(n, l)
    dark = NoL <= 0
    c01 = lambda x: np.fmin(np.fmax(x, f32(0)), f32(1))
    NoL, LoH = c01(NoL), c01(dn(l, hv))
    f90 = r(r(f32(0.5)) + r(r(r(r(f32(2.0)) * rough) * LoH) * LoH))
    schlick1 = lambda u: r(one + r(r(f90 - one) * pow5(c01(r(one - u)), r)))
    fdv = r(r(schlick1(NoL) * schlick1(NoV)) * r(one / r(f32(3.1415927))))
    return [np.where(dark, f32(0), r(diff[i] * fdv)) for i in range(3)]


def rt_trace_gi(arrays, tris, pad, o, d, tmin, tmax, sun, sky_v, sky_t, noise, dx, dy, remaining=0, bounce=False):
    """TraceRay(RAY_TYPE_GI) with payload.remaining_bounces = `remaining` (the reference's generators: 0): the closest-hit stage of
    gltf_basic_pbr.slang:372-520 or the miss stage of sky_unified.slang:227-230.  `bounce`: the ray is a hit stage's bounce ray (its flags).
    o, d: (N, 3); dx, dy: DispatchRaysIndex().xy per ray.  Returns (irradiance (N, 3) fp32, ray_distance (N,))."""
    n = o.shape[0]
    best, t, b1, b2, front = rt_closest_hit(arrays, tris, pad, o, d, tmin, tmax, bounce)
    irr, dist = np.zeros((n, 3), f32), np.zeros(n, f32)
    finite = np.isfinite(o).all(axis=1) & np.isfinite(d).all(axis=1)
    miss = (best < 0) & finite
    sd = [f32(sun.direction_and_tan_size[k]) for k in range(4)]
    if miss.any():  # get_sky_color(WorldRayDirection(), sun_light.direction_and_tan_size.xyz, ...): the direction as stored
        with np.errstate(all="ignore"):
            sc = sky_color([d[miss, 0], d[miss, 1], d[miss, 2]], sd[:3], sky_v.astype(f32), sky_t.astype(f32))
        for c in range(3):
            irr[miss, c] = sc[c]
    light = normalize3([h(F(-sd[k])) for k in range(3)], h)
    mt = arrays.get("material_textures")
    for ti in np.unique(best[best >= 0]):
        tr = tris[int(ti)]
        sel = np.nonzero(best == ti)[0]
        prim = arrays["primitives"][tr["primitive"]]
        mat = arrays["materials"][int(prim["material"])]
        vd = arrays["vertex_data"][tr["vertices"]]
        pos = arrays["positions"][tr["vertices"]].astype(f32)
        B1, B2 = b1[sel], b2[sel]
        B0 = F(F(f32(1) - B1) - B2)
        mix3 = lambda a0, a1, a2: F(F(F(B0 * f32(a0)) + F(B1 * f32(a1))) + F(B2 * f32(a2)))
        with np.errstate(all="ignore"):
            normal = [mix3(vd["normal"][0][k], vd["normal"][1][k], vd["normal"][2][k]) for k in range(3)]
            uv = [mix3(vd["texcoord"][0][k], vd["texcoord"][1][k], vd["texcoord"][2][k]) for k in range(2)]
            colour = []
            for k in range(4):  # unpackUnorm4x8ToHalf, float * half4 sums in fp32, packUnorm4x8 (truncating), unpacked again
                u = [h(h(f32((int(c) >> (8 * k)) & 0xff)) / h(255.0)) for c in vd["color"]]
                cc = mix3(u[0], u[1], u[2])
                prod = h(h(cc) * h(255.0))
                byte = np.where(prod > 0, np.floor(prod), 0).astype(np.int64) & 0xff
                colour.append(h(h(byte.astype(f32)) / h(255.0)))
            mp = [mix3(pos[0][k], pos[1][k], pos[2][k]) for k in range(3)]
            mdl = prim["model"].astype(f32)
            loc = np.stack([F(F(F(F(mdl[k] * mp[0]) + F(mdl[4 + k] * mp[1])) + F(mdl[8 + k] * mp[2])) + F(mdl[12 + k] * f32(1))) for k in range(3)], axis=-1)
            texel = {}
            for slot, (key, col) in enumerate((("base_color_texel", 0), (None, None), ("data_texel", 2), ("emission_texel", 3))):
                if key is None:
                    continue
                tx = np.broadcast_to(np.array(mat[key], f32), sel.shape + (4,))
                if mt is not None and len(arrays.get("textures", [])):
                    tex_i = int(mt[int(prim["material"])][slot])
                    if tex_i != _abi.TEXTURE_NONE:
                        tx = sample_bias(arrays["textures"][tex_i], uv, None, None, 0.0, explicit_lod=0.0)  # SampleLevel(v.texcoord, 0)
                texel[key] = tx
            base = [h(F(F(texel["base_color_texel"][..., c] * f32(mat["base_color_tint"][c])) * colour[c])) for c in range(3)]
            nh = [h(x) for x in normal]  # no normal map in the ray-traced path, and not normalised
            rough = h(h(texel["data_texel"][..., 1]) * h(f32(mat["roughness_factor"])))
            metal = h(h(texel["data_texel"][..., 2]) * h(f32(mat["metalness_factor"])))
            emission = [h(h(texel["emission_texel"][..., c]) * h(f32(mat["emission_factor"][c]))) for c in range(3)]
            brdf_result = fd_general(base, nh, rough, metal, light, nh, h)
            ndotl = np.fmin(np.fmax(h(h(h(light[0] * nh[0]) + h(light[1] * nh[1])) + h(light[2] * nh[2])), f32(0)), f32(1))
            shadow = np.zeros(sel.shape, f32)
            lit = ndotl > 0
            if lit.any():
                nz = rt_noise(noise, dx[sel][lit] % 128, dy[sel][lit] % 128)
                sdir = np.stack(normalize3([F(light[k] + F(nz[:, k] * sd[3])) for k in range(3)]), axis=-1)
                # ACCEPT_FIRST_HIT_AND_END_SEARCH | CULL_NON_OPAQUE | CULL_FRONT_FACING_TRIANGLES
                occluded = rt_any_hit(arrays, tris, pad, loc[lit], sdir, 0.05, 100000.0, cull_non_opaque=True, cull_front=True)
                shadow[lit] = np.where(occluded, f32(0), f32(1))
            val = [F(F(F(F(brdf_result[c] * f32(sun.color[c])) * ndotl) * shadow) + emission[c]) for c in range(3)]
            if remaining > 0:  # :481-517 (a back-face hit is zeroed below whatever the bounce brings)
                nz = rt_noise(noise, dx[sel] % 128, dy[sel] % 128)
                nf = [nh[k].astype(f32) for k in range(3)]
                flip = F(F(F(nf[0] * nz[:, 0]) + F(nf[1] * nz[:, 1])) + F(nf[2] * nz[:, 2])) < 0
                bd = np.where(flip[:, None], F(nz * f32(-1)), nz)
                nirr, _ = rt_trace_gi(arrays, tris, pad, loc, bd, 0.05, 100000.0, sun, sky_v, sky_t, noise, dx[sel], dy[sel], remaining - 1, True)
                bl = [h(bd[:, k]) for k in range(3)]
                bb = brdf(base, nh, rough, metal, bl, nh, h)  # brdf(surface, bounce_ray.Direction, surface.normal) = Fd + Fr
                bndotl = h(np.fmin(np.fmax(F(F(F(bd[:, 0] * nf[0]) + F(bd[:, 1] * nf[1])) + F(bd[:, 2] * nf[2])), f32(0)), f32(1)))
                rad = np.stack([F(h(bndotl * bb[c]).astype(f32) * nirr[:, c]) for c in range(3)], axis=-1)
                fin = np.isfinite(rad).all(axis=1)
                val = [np.where(fin, F(val[c] + rad[:, c]), val[c]) for c in range(3)]
            for c in range(3):
                irr[sel, c] = np.where(front[sel], val[c], f32(0))  # HIT_KIND_TRIANGLE_BACK_FACE: black ...
            dist[sel] = np.where(front[sel], t[sel], F(t[sel] * f32(-1)))  # ... and a negative distance
    return irr, dist


def rtgi_trace(arrays, view, sun, sky_v, sky_t, depth, normals16, noise, bounces=0):
    """rtgi.rt.slang:56-110 -> (ray_buffer, ray_irradiance) as (H, W, 4) float16; texels the generator skips stay 0"""
    Hh, W = depth.shape
    tris, pad = rt_world_triangles(arrays)
    ys, xs = np.meshgrid(np.arange(Hh), np.arange(W), indexing="ij")
    go = ((xs.astype(f32) < f32(view.render_resolution[0])) & (ys.astype(f32) < f32(view.render_resolution[1])) & (depth != 0)).reshape(-1)
    pos = rt_world_position(view, W, Hh, depth).reshape(-1, 3)
    nrm = normals16.view(np.float16).astype(f32)[..., :3].reshape(-1, 3)  # as stored: not normalised here
    with np.errstate(all="ignore"):
        d = rt_noise(noise, xs % 128, ys % 128).reshape(-1, 3)
        flip = F(F(F(nrm[:, 0] * d[:, 0]) + F(nrm[:, 1] * d[:, 1])) + F(nrm[:, 2] * d[:, 2])) < 0
        d = np.where(flip[:, None], F(d * f32(-1)), d)
    idx = np.nonzero(go)[0]
    irr, dist = rt_trace_gi(arrays, tris, pad, pos[idx], d[idx], 0.01, 100000.0, sun, sky_v, sky_t, noise, xs.reshape(-1)[idx], ys.reshape(-1)[idx], bounces)
    irr = np.where(np.isnan(irr).any(axis=1)[:, None], f32(0), irr)
    rb, ri = np.zeros((Hh * W, 4), np.float16), np.zeros((Hh * W, 4), np.float16)
    with np.errstate(over="ignore"):
        rb[idx, :3], rb[idx, 3] = d[idx].astype(np.float16), dist.astype(np.float16)
        ri[idx, :3] = F(irr * f32(0.0031415927)).astype(np.float16)
    return rb.reshape(Hh, W, 4), ri.reshape(Hh, W, 4)


def probe_trace(arrays, cascades, probes, sun, sky_v, sky_t, noise, irr_words, depth_atlas, validity):
    """probe_tracing.rt.slang:39-106: cascades = [(min xyz, spacing)] x 4, probes (N, 3) -> (N, 20, 20, 4) float16"""
    tris, pad = rt_world_triangles(arrays)
    irr_atlas = unpack_b10g11r11(irr_words)
    out = np.zeros((len(probes), 20, 20, 4), np.float16)
    tys, txs = np.meshgrid(np.arange(20), np.arange(20), indexing="ij")
    dirs = np.array([[texel_octahedral_direction(tx, ty, 20, 20) for tx in range(20)] for ty in range(20)], f32).reshape(-1, 3)
    e = h(f32(0.0031415927))
    for p, (px, py, pz) in enumerate(np.asarray(probes, np.int64)):
        cascade = py // 8
        if cascade >= 4:
            continue
        cmin, spacing = [f32(v) for v in cascades[cascade][0]], f32(cascades[cascade][1])
        local = [f32(px), f32(py % 8), f32(pz)]
        origin = np.array([F(cmin[k] + F(local[k] * spacing)) for k in range(3)], f32)
        ray_distance = f32(8192.0) if cascade >= 3 else F(f32(cascades[cascade + 1][1]) * f32(4.0))
        o = np.broadcast_to(origin, (400, 3)).astype(f32)
        irr, dist = rt_trace_gi(arrays, tris, pad, o, dirs, 0.05, ray_distance, sun, sky_v, sky_t, noise, txs.reshape(-1), tys.reshape(-1))
        miss = dist == 0
        if miss.any():
            if cascade + 1 < 4:
                nmin, nsp = [f32(v) for v in cascades[cascade + 1][0]], f32(cascades[cascade + 1][1])
                end = [F(origin[k] + F(dirs[miss, k] * ray_distance)) for k in range(3)]
                sc = sample_cascade(end, [dirs[miss, 0], dirs[miss, 1], dirs[miss, 2]], np.full(int(miss.sum()), cascade + 1), [np.full(int(miss.sum()), v, f32) for v in nmin],
                                    np.full(int(miss.sum()), nsp, f32), irr_atlas, depth_atlas.astype(f32), validity)
                for c in range(3):
                    irr[miss, c] = sc[c]
            else:
                irr[miss] = F(irr[miss] * f32(10.0))
            dist = np.where(miss, ray_distance, dist)
        irr = np.where((dist < 0)[:, None], f32(0), irr)
        with np.errstate(over="ignore", invalid="ignore"):
            out[p, ..., :3] = h(h(irr) * e).astype(np.float16).reshape(20, 20, 3)
            out[p, ..., 3] = dist.astype(np.float16).reshape(20, 20)
    return out


def inputs_digest(arrays):
    m = hashlib.sha256()
    for k in sorted(arrays):
        m.update(k.encode())
        m.update(np.ascontiguousarray(arrays[k]).tobytes())
    return m.hexdigest()


def golden_rt():
    """RTAO and the sun shadow mask of tests/util.py: golden_raster_scene() (a textured CUTOUT wall behind a SOLID triangle) plus a floor and
    an occluder above it, on the golden G-buffer's depth and normal planes"""
    from tests import util
    W, Hh = 64, 36
    m, view, sun, noise = util.golden_rt_scene()
    gb = raster_gbuffer(m, view, W, Hh)
    arrays = m.arrays()
    ao = rtao(arrays, view.gpu_data, gb["depth"], gb["normals"], noise, 2, 3.0)
    mask = sun_shadow_mask(arrays, view.gpu_data, sun.constants, gb["depth"], gb["normals"], noise)
    np.savez_compressed(os.path.join(GOLDEN, f"rt_{W}x{Hh}.npz"), ao=ao, mask=mask, depth=gb["depth"], normals=gb["normals"])
    print("rt ok: occluded", int((ao == 0).sum()), "of", ao.size, "; shadow mask values", np.unique(mask))
    golden_rt_gi(m, view, sun, noise, gb)


def golden_rt_gi(m=None, view=None, sun=None, noise=None, gb=None):
    """the GI rays on the same scene: one GI ray per pixel (rtgi.rt.slang) and twelve probes of the irradiance cache (probe_tracing.rt.slang),
    with the sky LUTs, atlases and cascades of tests/util.py: golden_rt_gi_inputs()"""
    from tests import util
    W, Hh = 64, 36
    if m is None:
        m, view, sun, noise = util.golden_rt_scene()
        g = np.load(os.path.join(GOLDEN, f"rt_{W}x{Hh}.npz"))
        gb = {"depth": g["depth"], "normals": g["normals"]}
    arrays = m.arrays()
    gi = util.golden_rt_gi_inputs()
    rb, ri = rtgi_trace(arrays, view.gpu_data, sun.constants, gi["sky_v"], gi["sky_t"], gb["depth"], gb["normals"], noise)
    trace = probe_trace(arrays, gi["cascades"], gi["probes"], sun.constants, gi["sky_v"], gi["sky_t"], noise, gi["irr"], gi["pdepth"], gi["val"])
    np.savez_compressed(os.path.join(GOLDEN, f"rt_gi_{W}x{Hh}.npz"), ray_buffer=rb.view(np.uint16), ray_irradiance=ri.view(np.uint16), trace=trace.view(np.uint16))
    # the same GI rays with the hit stage's bounce branch on (remaining_bounces = 1, 2: sah_rt_set_bounces)
    b = {}
    for nb in (1, 2):
        rb_b, ri_b = rtgi_trace(arrays, view.gpu_data, sun.constants, gi["sky_v"], gi["sky_t"], gb["depth"], gb["normals"], noise, bounces=nb)
        assert np.array_equal(rb_b.view(np.uint16), rb.view(np.uint16))  # directions and first-hit distances do not depend on the bounces
        b[f"ray_irradiance_{nb}"] = ri_b.view(np.uint16)
        print(f"rt gi, {nb} bounce(s): texels brighter than with {nb - 1}:", int((ri_b.astype(f32) > (ri if nb == 1 else prev).astype(f32)).any(-1).sum()))
        prev = ri_b
    np.savez_compressed(os.path.join(GOLDEN, f"rt_gi_bounces_{W}x{Hh}.npz"), **b)
    d = rb[..., 3].astype(f32)
    td = trace[..., 3].astype(f32)
    print("rt gi ok: rtgi hits", int((d > 0).sum()), "back", int((d < 0).sum()), "misses", int((d == 0).sum()), "; probe rays front", int((td > 0).sum()),
          "back", int((td < 0).sum()))


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    if "--only-aniso" in sys.argv:  # the textured G-buffer scene with anisotropic samplers (sah_hip.h "anisotropy")
        from tests import util
        rm, rview = util.golden_raster_scene(anisotropic=True)
        gb = raster_gbuffer(rm, rview, 64, 36)
        np.savez_compressed(os.path.join(GOLDEN, "raster_gbuffer_aniso_64x36.npz"), **gb)
        iso = np.load(os.path.join(GOLDEN, "raster_gbuffer_64x36.npz"))
        print("raster_gbuffer_aniso ok: covered", int((gb["depth"] > 0).sum()), "texels that differ from the isotropic golden:",
              {k: int((gb[k] != iso[k]).any(axis=-1).sum()) if gb[k].ndim == 3 else int((gb[k] != iso[k]).sum()) for k in ("color", "normals", "data", "emission")})
        return
    if "--only-rt-gi" in sys.argv:
        golden_rt_gi()
        return
    if "--only-rt" in sys.argv:
        return golden_rt()
    golden_rt()
    W, Hh = 64, 36
    for name, sun_mode, gi, seed in (("lighting_csm_lpv", _abi.SHADOW_MODE_CSM, _abi.GI_LPV, 101), ("lighting_rt", _abi.SHADOW_MODE_RT, _abi.GI_NONE, 102),
                                     ("lighting_csm", _abi.SHADOW_MODE_CSM, _abi.GI_NONE, 103)):
        fr = Frame(W, Hh, seed, sun_mode, gi)
        lit = compose(fr, sun_mode, gi)
        np.savez_compressed(os.path.join(GOLDEN, f"{name}_{W}x{Hh}.npz"), lit=lit, seed=seed, sun_mode=sun_mode, gi=gi,
                            inputs_sha256=inputs_digest(fr.f.arrays))
        print(name, "ok", lit.shape)
    # RT sun + RTGI reconstruction (a1b + a5), CSM sun + 12 point lights (a1 + a9)
    fr = Frame(W, Hh, 105, _abi.SHADOW_MODE_RT, _abi.GI_RTGI)
    lit = compose(fr, _abi.SHADOW_MODE_RT, _abi.GI_RTGI)
    np.savez_compressed(os.path.join(GOLDEN, f"lighting_rt_rtgi_{W}x{Hh}.npz"), lit=lit, seed=105, sun_mode=_abi.SHADOW_MODE_RT, gi=_abi.GI_RTGI,
                        inputs_sha256=inputs_digest(fr.f.arrays))
    print("lighting_rt_rtgi ok")
    fr = Frame(W, Hh, 106, _abi.SHADOW_MODE_CSM, _abi.GI_NONE)
    lights = synth.point_lights(fr.f.view, 12, 6.0, seed=107)
    lit = compose(fr, _abi.SHADOW_MODE_CSM, _abi.GI_NONE, lights=lights)
    np.savez_compressed(os.path.join(GOLDEN, f"lighting_csm_lights_{W}x{Hh}.npz"), lit=lit, seed=106, sun_mode=_abi.SHADOW_MODE_CSM, gi=_abi.GI_NONE,
                        lights=lights, inputs_sha256=inputs_digest(fr.f.arrays))
    print("lighting_csm_lights ok")
    # irradiance-cache gather (a4) and sky fill (a6): the reference's default configuration (RT sun + probe GI + sky), and the
    # fast path's sky (CSM + LPV + sky)
    for name, sun_mode, gi, seed in (("lighting_rt_cache_sky", _abi.SHADOW_MODE_RT, _abi.GI_CACHE, 109),
                                     ("lighting_csm_lpv_sky", _abi.SHADOW_MODE_CSM, _abi.GI_LPV, 110)):
        fr = Frame(W, Hh, seed, sun_mode, gi, sky=True)
        lit = compose(fr, sun_mode, gi)
        np.savez_compressed(os.path.join(GOLDEN, f"{name}_{W}x{Hh}.npz"), lit=lit, seed=seed, sun_mode=sun_mode, gi=gi, sky=1,
                            inputs_sha256=inputs_digest(fr.f.arrays))
        print(name, "ok", "sky pixels:", int((fr.depth == 0).sum()))
    # copy scene (a13) of the CSM + LPV image
    src = np.load(os.path.join(GOLDEN, f"lighting_csm_lpv_{W}x{Hh}.npz"))["lit"]
    np.savez_compressed(os.path.join(GOLDEN, f"copy_scene_{W}x{Hh}.npz"), out=copy_scene(src))
    print("copy_scene ok")
    # LPV propagate (a10): two cascades, sparse injected light incl. cells on the cascade seam and the volume border, 3 steps
    g = synth.rng(108)
    vols = [np.zeros((32, 32, 64, 4), dtype=np.float16) for _ in range(3)]
    cells = [(0, 0, 0), (31, 31, 63), (5, 7, 31), (5, 7, 32), (16, 0, 40), (0, 16, 33)] + [tuple(int(t) for t in g.integers(0, (32, 32, 64))) for _ in range(40)]
    for v in vols:
        for (z, y, x) in cells:
            v[z, y, x] = g.uniform(-1.0, 2.0, 4).astype(np.float16)
    out = lpv_propagate(vols, steps=3, num_cascades=2)
    np.savez_compressed(os.path.join(GOLDEN, "lpv_propagate_2c_3steps.npz"), **{f"in{i}": vols[i].view(np.uint16) for i in range(3)},
                        **{f"out{i}": out[i].view(np.uint16) for i in range(3)})
    print("lpv_propagate ok")

    # probe maintenance (a11): 48 probes incl. the grid corners, an all-miss and an all-hit probe; stored as the changed texels
    atl, trace, ids = synth.probe_maintenance_inputs(seed=111, num_probes=48)
    before = {k: v.copy() for k, v in atl.items()}
    probe_update(atl, trace, ids)
    sparse = {}
    for k in atl:
        a, b = atl[k], before[k]
        changed = (a != b) if a.ndim == 3 else (a.view(np.uint16) != b.view(np.uint16)).any(axis=-1)
        idx = np.argwhere(changed).astype(np.int16)
        sparse[f"{k}_idx"] = idx
        sparse[f"{k}_val"] = a[tuple(idx.T.astype(np.int64))] if a.dtype != np.float16 else a.view(np.uint16)[tuple(idx.T.astype(np.int64))]
        print("probe_update", k, len(idx), "texels changed")
    np.savez_compressed(os.path.join(GOLDEN, "probe_update_48.npz"), seed=111, num_probes=48, **sparse)

    # scene rasteriser with material textures (f1): depth pre-pass + G-buffer pass of tests/util.py: golden_raster_scene()
    from tests import util
    rm, rview = util.golden_raster_scene()
    gb = raster_gbuffer(rm, rview, W, Hh)
    np.savez_compressed(os.path.join(GOLDEN, f"raster_gbuffer_{W}x{Hh}.npz"), **gb)
    print("raster_gbuffer ok: covered", int((gb["depth"] > 0).sum()), "of", W * Hh)
    # sun shadow cascades of the same scene (f2): two cascades fitted to the camera, 48^2 texels each
    rsun = util.golden_raster_sun(rview)
    sm = raster_shadow(rm, rsun.constants, 2, 48)
    np.savez_compressed(os.path.join(GOLDEN, "raster_shadow_2x48.npz"), shadowmap=sm)
    print("raster_shadow ok: covered", int((sm != 0xFFFF).sum()), "of", sm.size)
    # the LPV's reflective shadow map of the same scene (f4): four cascades of 32^2 texels
    rlpv = util.golden_raster_lpv(rview, rsun)
    rsm = raster_rsm(rm, rsun.constants, rlpv.matrices, 4, 32)
    np.savez_compressed(os.path.join(GOLDEN, "raster_rsm_4x32.npz"), **rsm)
    print("raster_rsm ok: covered", int((rsm["depth"] != 0xFFFF).sum()), "of", rsm["depth"].size)
    # VPL extraction and injection of that RSM (f4): the four lists, and the non-zero cells of the three volumes
    vols = [np.zeros((32, 32, 128, 4), np.float16) for _ in range(3)]
    lists = {}
    for c in range(4):
        lists[f"vpls_{c}"] = extract_vpls(rsm, rlpv.matrices[c], c, 0.25)
        inject_vpls(lists[f"vpls_{c}"], rlpv.matrices[c], c, 4, vols)
    cells = np.argwhere(np.any([v.view(np.uint16).any(axis=-1) for v in vols], axis=0)).astype(np.int16)
    at = tuple(cells.T.astype(np.int64))
    np.savez_compressed(os.path.join(GOLDEN, "lpv_inject_4x32.npz"), cells=cells, **lists, **{f"vol_{i}": vols[i].view(np.uint16)[at] for i in range(3)})
    print("lpv_inject ok:", [len(lists[f"vpls_{c}"]) for c in range(4)], "lights,", len(cells), "cells lit")

    # probe scroll (a11, copy_cascades): mixed movements, the last cascade at rest; the fixture is the digest of every atlas
    src_atl, _, _ = synth.probe_maintenance_inputs(seed=112, num_probes=4)
    dst_atl = {k: np.full_like(v, 0x55 if v.dtype == np.uint8 else 0x3555 if v.dtype == np.uint32 else 7.0) for k, v in src_atl.items()}
    movement = [[1.0, 0.0, -2.0], [0.0, 1.7, 0.0], [-3.2, -1.0, 2.9], [0.0, 0.0, 0.0]]
    probe_copy(src_atl, dst_atl, movement)
    np.savez_compressed(os.path.join(GOLDEN, "probe_copy.npz"), seed=112, movement=np.array(movement, np.float32),
                        **{f"sha256_{k}": hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in dst_atl.items()})
    print("probe_copy ok")
    # sky LUT generators (f3) for the sun direction of tests/test_sky_luts.py; every fourth row of the two larger LUTs is kept
    light = (0.3, -0.8, 0.52)
    with np.errstate(all="ignore"):
        tl = sky_transmittance_lut()
        msl = sky_multiscattering_lut(tl)
        svl = sky_view_lut(tl, msl, light)
    np.savez_compressed(os.path.join(GOLDEN, "sky_luts.npz"), light=np.array(light, np.float32), transmittance_rows=tl.view(np.uint16)[::4],
                        multiscattering=msl.view(np.uint16), sky_view_rows=svl.view(np.uint16)[::4])
    print("sky_luts ok")

    scene_img = synth.hdr_scene(W, Hh, seed=104).view(np.uint16)
    mips, src = [], scene_img
    for (mw, mh) in images.bloom_mip_sizes(W, Hh, 6):
        src = bloom_downsample(src, mw, mh)
        mips.append(src)
    out = tonemap(scene_img, mips, W, Hh)
    np.savez_compressed(os.path.join(GOLDEN, f"post_{W}x{Hh}.npz"), seed=104, final=out, **{f"mip{i}": m for i, m in enumerate(mips)})
    print("post ok")


if __name__ == "__main__":
    main()
