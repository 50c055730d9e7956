"""GPU parity for the post chain (copy scene, bloom pyramid, tonemap) and LPV propagate/clear vs the oracle."""
import ctypes as C

import numpy as np
import pytest

from androidrenderer_amd import _abi, images, synth
from tests import util

pytestmark = pytest.mark.gpu


def _mips_np(w, h, n=6):
    return [np.zeros((mh, mw, 4), dtype=np.uint16) for (mw, mh) in images.bloom_mip_sizes(w, h, n)]


def _run_post_oracle(scene, w, h):
    o = util.oracle()
    mips = _mips_np(w, h)
    chain = images.mipchain(mips)
    sp = images.plane(scene, _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert o.orc_bloom(C.byref(sp), C.byref(chain)) == 0
    out = np.zeros((h, w, 4), dtype=np.uint8)
    op = images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)
    assert o.orc_tonemap(C.byref(sp), C.byref(chain), C.byref(op), 0, 0) == 0
    return mips, out


def _run_post_hip(ctx, scene, w, h, rows=None):
    import torch
    sc = util.to_torch(scene.view(np.uint16))
    mips = [torch.zeros(m.shape, dtype=torch.int16, device="cuda") for m in _mips_np(w, h)]
    chain = images.mipchain(mips)
    sp = images.plane(sc, _abi.FORMAT_R16G16B16A16_SFLOAT)
    ctx.bloom(sp, chain)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    op = images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)
    if rows is None:
        ctx.tonemap(sp, chain, op)
    else:
        for r0, r1 in rows:
            ctx.tonemap(sp, chain, op, r0, r1)
    torch.cuda.synchronize()
    return [util.from_torch(m, np.uint16) for m in mips], out.cpu().numpy()


@pytest.mark.parametrize("size", [(256, 144), (250, 130), (1920, 1080), (333, 187), (64, 36), (40, 24), (9, 5), (2048, 96), (97, 512)])
def test_bloom_and_tonemap(hip_ctx, size):
    w, h = size
    scene = synth.hdr_scene(w, h, seed=21)
    if size == (333, 187):  # signed, huge, inf and NaN texels: the interpolated path must fall back to the per-pixel filter where it has to
        f = scene.view(np.float16).reshape(h, w, 4)
        rng = np.random.default_rng(3)
        ys, xs = rng.integers(0, h, 400), rng.integers(0, w, 400)
        f[ys[:150], xs[:150], :3] *= np.float16(-1.0)
        f[ys[150:300], xs[150:300], :3] = np.float16(60000.0)
        f[ys[300:350], xs[300:350], 0] = np.float16(np.inf)
        f[ys[350:], xs[350:], 1] = np.float16(np.nan)
    ref_mips, ref_out = _run_post_oracle(scene.view(np.uint16), w, h)
    got_mips, got_out = _run_post_hip(hip_ctx, scene, w, h)
    for i, (a, b) in enumerate(zip(got_mips, ref_mips)):
        d = util.f16_ulp_diff(a, b)
        print(util.report_ulp(f"bloom mip {i} {a.shape}", d))
        assert d.max() == 0
    dc = np.abs(got_out.astype(np.int32) - ref_out.astype(np.int32))
    print(f"tonemap: max code diff {dc.max()}, {int((dc > 0).sum())}/{dc.size} differ")
    assert dc.max() == 0  # identical UNORM8 codes (the contract allows +-1)


def test_tonemap_row_shards(hip_ctx):
    w, h = 200, 117
    scene = synth.hdr_scene(w, h, seed=22)
    _, full = _run_post_hip(hip_ctx, scene, w, h)
    _, parts = _run_post_hip(hip_ctx, scene, w, h, rows=[(0, 27), (27, 54), (54, 85), (85, 117)])
    assert np.array_equal(full, parts)


def test_copy_scene(hip_ctx):
    import torch
    o = util.oracle()
    w, h = 160, 90
    scene = synth.hdr_scene(w, h, seed=23).view(np.uint16)
    for (ow, oh) in ((160, 90), (320, 180), (200, 100)):
        ref = np.zeros((oh, ow, 4), dtype=np.uint16)
        sp, rp = images.plane(scene, _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(ref, _abi.FORMAT_R16G16B16A16_SFLOAT)
        assert o.orc_copy_scene(C.byref(sp), C.byref(rp)) == 0
        sc = util.to_torch(scene)
        out = torch.zeros((oh, ow, 4), dtype=torch.int16, device="cuda")
        hip_ctx.copy_scene(images.plane(sc, _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(out, _abi.FORMAT_R16G16B16A16_SFLOAT))
        torch.cuda.synchronize()
        got = util.from_torch(out, np.uint16)
        d = util.f16_ulp_diff(got, ref)
        print(util.report_ulp(f"copy {ow}x{oh}", d))
        assert d.max() <= 1
        if (ow, oh) == (w, h):
            # same resolution: the sample lands on texel centres up to fp32 noise in u*W - 0.5, so the copy is the identity
            # except where a ~1e-7 weight on a much brighter neighbour moves the last bits (SURVEY a13 calls it exact; it is not)
            assert (util.f16_ulp_diff(got, scene) <= 1).mean() > 0.99


@pytest.mark.parametrize("steps", [4, 5, 1, 9, 32])
def test_lpv_propagate_and_clear(hip_ctx, steps):
    import torch
    o = util.oracle()
    vols = synth.lpv_volumes(4, seed=31)
    a_np = [v.view(np.uint16).copy() for v in vols]
    b_np = [np.full_like(v, 0x3C00) for v in a_np]  # garbage in B: every cell is overwritten
    a_v = (_abi.Volume * 3)(*[images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in a_np])
    b_v = (_abi.Volume * 3)(*[images.volume(v, _abi.FORMAT_R16G16B16A16_SFLOAT) for v in b_np])
    assert o.orc_lpv_propagate(a_v, b_v, 4, steps) == 0
    a_t = [util.to_torch(v.view(np.uint16).copy()) for v in vols]
    b_t = [torch.full_like(t, 0x3C00) for t in a_t]
    hip_ctx.lpv_propagate([images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in a_t],
                          [images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in b_t], 4, steps)
    torch.cuda.synchronize()
    for name, ts, ns in (("A", a_t, a_np), ("B", b_t, b_np)):
        for c in range(3):
            got = util.from_torch(ts[c], np.uint16)
            assert np.array_equal(got, ns[c]), f"LPV {name}[{c}] differs: {util.report_ulp('lpv', util.f16_ulp_diff(got, ns[c]))}"
    # clear
    hip_ctx.lpv_clear(*[images.volume(t, _abi.FORMAT_R16G16B16A16_SFLOAT) for t in a_t], images.volume(b_t[0], _abi.FORMAT_R16G16B16A16_SFLOAT), 4)
    torch.cuda.synchronize()
    assert all(int(t.abs().max()) == 0 for t in a_t) and int(b_t[0].abs().max()) == 0


@pytest.mark.parametrize("case", ["extremes", "nonfinite", "three_cascades_padded"])
def test_lpv_propagate_hot_form_equals_general_form_and_oracle(hip_ctx, case):
    """The propagation's hot form (csrc/lpv.hip: zero table entries dropped, shared products — exact for finite coefficients) against the general
    form of the same library (sah_debug_set(force_general)) and the oracle, on volumes made to break it: magnitudes up to the largest half (sums
    that overflow), denormals, negative zeros, exact cancellations; waves that hold an inf / NaN (they must take the general form); three cascades
    in volumes larger than the propagated cells, with pitches that are not the tight ones."""
    import torch
    o = util.oracle()
    rng = np.random.default_rng({"extremes": 71, "nonfinite": 72, "three_cascades_padded": 73}[case])
    nc = 3 if case == "three_cascades_padded" else 4
    w, h, d = (32 * nc + 5, 34, 33) if case == "three_cascades_padded" else (32 * nc, 32, 32)
    vols = []
    for c in range(3):
        kind = rng.integers(0, 8, (d, h, w, 4))
        v = rng.uniform(-2.0, 2.0, (d, h, w, 4)).astype(np.float16)
        v = np.where(kind == 0, np.float16(0.0), v)
        v = np.where(kind == 1, np.float16(-0.0), v)
        v = np.where(kind == 2, (rng.uniform(-1, 1, v.shape) * 6.0e-6).astype(np.float16), v)       # denormals
        v = np.where(kind == 3, (rng.choice([-1.0, 1.0], v.shape) * rng.uniform(3.0e4, 65504.0, v.shape)).astype(np.float16), v)
        v = np.where(kind == 4, np.float16(0.5), v)                                                   # equal magnitudes: exact cancellations
        v = np.where(kind == 5, np.float16(-0.5), v)
        if case == "nonfinite":
            zz, yy, xx = rng.integers(0, d, 40), rng.integers(0, h, 40), rng.integers(0, w, 40)
            v[zz[:15], yy[:15], xx[:15], 0] = np.float16(np.inf)
            v[zz[15:30], yy[15:30], xx[15:30], 2] = np.float16(-np.inf)
            v[zz[30:], yy[30:], xx[30:], 3] = np.float16(np.nan)
        vols.append(np.ascontiguousarray(v))
    steps = 3

    def desc(arrs):
        return [images.volume(a, _abi.FORMAT_R16G16B16A16_SFLOAT) for a in arrs]
    a_np = [v.view(np.uint16).copy() for v in vols]
    b_np = [np.full_like(v, 0x3C00) for v in a_np]
    assert o.orc_lpv_propagate((_abi.Volume * 3)(*desc(a_np)), (_abi.Volume * 3)(*desc(b_np)), nc, steps) == 0
    results = []
    for force_general in (False, True):
        hip_ctx.debug_set(force_general=force_general)
        try:
            a_t = [util.to_torch(v.view(np.uint16).copy()) for v in vols]
            b_t = [torch.full_like(t, 0x3C00) for t in a_t]
            hip_ctx.lpv_propagate(desc(a_t), desc(b_t), nc, steps)
            torch.cuda.synchronize()
            results.append([util.from_torch(t, np.uint16).reshape(a_np[0].shape) for t in a_t + b_t])
        finally:
            hip_ctx.debug_set(force_general=False)
    want = a_np + b_np

    def same(x, y):  # bit for bit, except that a NaN is a NaN whatever its payload (numerics contract: DESIGN.md section 3)
        xn, yn = (x & 0x7FFF) > 0x7C00, (y & 0x7FFF) > 0x7C00
        return bool(np.all((x == y) | (xn & yn)))
    for i in range(6):
        assert same(results[0][i], want[i]), f"hot form differs from the oracle in volume {i}: {int(np.sum(results[0][i] != want[i]))} halves"
        assert same(results[1][i], want[i]), f"general form differs from the oracle in volume {i}"
    if case != "nonfinite":
        assert np.isfinite(vols[0].astype(np.float32)).all()


@pytest.mark.parametrize("size", [(256, 144), (250, 130), (333, 187), (64, 36), (9, 5), (2048, 96), (97, 512), (1920, 1080)])
def test_copy_scene_and_bloom_mip0_in_one_pass(hip_ctx, size):
    """sah_copy_scene_bloom_mip0_rows == sah_copy_scene_rows followed by sah_bloom_mip0_rows == the oracle, bit for bit: whole frames (odd
    extents: the last tile column / row owns the odd texels; extents the fused form does not take fall back to the two passes), non-finite
    and negative texels included."""
    import torch
    w, h = size
    o = util.oracle()
    scene = synth.hdr_scene(w, h, seed=31)
    f = scene.reshape(h, w, 4)
    rng = np.random.default_rng(5)
    ys, xs = rng.integers(0, h, 60), rng.integers(0, w, 60)
    f[ys[:20], xs[:20], :3] *= np.float16(-1.0)
    f[ys[20:40], xs[20:40], 0] = np.float16(np.inf)
    f[ys[40:], xs[40:], 1] = np.float16(np.nan)
    lit = scene.view(np.uint16)
    aa_ref = np.zeros((h, w, 4), np.uint16)
    mips_ref = _mips_np(w, h, 1)
    assert o.orc_copy_scene(C.byref(images.plane(lit, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.plane(aa_ref, _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
    assert o.orc_bloom(C.byref(images.plane(aa_ref, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.mipchain(mips_ref))) == 0
    lit_t = util.to_torch(lit)
    aa_t = torch.full((h, w, 4), 0x3C00, dtype=torch.int16, device="cuda")
    mip_t = [torch.full(mips_ref[0].shape, 0x3C00, dtype=torch.int16, device="cuda")]
    hip_ctx.copy_scene_bloom_mip0(images.plane(lit_t, _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(aa_t, _abi.FORMAT_R16G16B16A16_SFLOAT), images.mipchain(mip_t))
    torch.cuda.synchronize()
    assert np.array_equal(util.from_torch(aa_t, np.uint16), aa_ref)
    assert np.array_equal(util.from_torch(mip_t[0], np.uint16), mips_ref[0])


def test_copy_scene_and_bloom_mip0_in_one_pass_row_ranges(hip_ctx):
    """Row-sharded form: antialiased rows and mip 0 rows given separately — rows beyond the mip rows' sources are copied by plain launches,
    rows not asked for keep their contents, and any partition gives the unsharded result."""
    import torch
    w, h = 300, 170
    o = util.oracle()
    lit = synth.hdr_scene(w, h, seed=33).view(np.uint16)
    aa_ref = np.zeros((h, w, 4), np.uint16)
    mips_ref = _mips_np(w, h, 1)
    assert o.orc_copy_scene(C.byref(images.plane(lit, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.plane(aa_ref, _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
    assert o.orc_bloom(C.byref(images.plane(aa_ref, _abi.FORMAT_R16G16B16A16_SFLOAT)), C.byref(images.mipchain(mips_ref))) == 0
    lit_t = util.to_torch(lit)
    lp = images.plane(lit_t, _abi.FORMAT_R16G16B16A16_SFLOAT)
    sentinel = 0x7E01
    for parts in ([((0, 170), (0, 85))], [((0, 24), (0, 11)), ((20, 25), (11, 12)), ((21, 121), (12, 60)), ((119, 170), (60, 85))],
                  [((60, 110), (40, 45))], [((75, 100), (40, 45))], [((0, 170), (80, 85))], [((5, 6), (0, 85))]):
        aa_t = torch.full((h, w, 4), sentinel, dtype=torch.int16, device="cuda")
        mip_t = [torch.full(mips_ref[0].shape, sentinel, dtype=torch.int16, device="cuda")]
        ap, mc = images.plane(aa_t, _abi.FORMAT_R16G16B16A16_SFLOAT), images.mipchain(mip_t)
        aa_rows, mip_rows = np.zeros(h, bool), np.zeros(mips_ref[0].shape[0], bool)
        for (a0, a1), (m0, m1) in parts:
            hip_ctx.copy_scene_bloom_mip0(lp, ap, mc, (a0, a1), (m0, m1))
            aa_rows[a0:a1] = True
            mip_rows[m0:m1] = True
        torch.cuda.synchronize()
        aa, mip = util.from_torch(aa_t, np.uint16), util.from_torch(mip_t[0], np.uint16)
        assert np.array_equal(aa[aa_rows], aa_ref[aa_rows]), parts
        assert (aa[~aa_rows] == sentinel).all(), parts
        assert np.array_equal(mip[mip_rows], mips_ref[0][mip_rows]), parts
        assert (mip[~mip_rows] == sentinel).all(), parts


def test_bloom_in_two_steps_equals_bloom(hip_ctx):
    """sah_bloom_mip0_rows over any partition of mip 0's rows + sah_bloom_from_mip0 == sah_bloom (the row-sharded pyramid)."""
    import torch
    w, h = 300, 170
    scene = synth.hdr_scene(w, h, seed=24)
    want, _ = _run_post_hip(hip_ctx, scene, w, h)
    sc = util.to_torch(scene.view(np.uint16))
    mips = [torch.full(m.shape, 0x3C00, dtype=torch.int16, device="cuda") for m in _mips_np(w, h)]
    chain = images.mipchain(mips)
    sp = images.plane(sc, _abi.FORMAT_R16G16B16A16_SFLOAT)
    for r0, r1 in ((0, 11), (11, 12), (12, 60), (60, 85)):
        hip_ctx.bloom_mip0_rows(sp, chain, r0, r1)
    hip_ctx.bloom_from_mip0(sp, chain)
    torch.cuda.synchronize()
    for a, b in zip(mips, want):
        assert np.array_equal(util.from_torch(a, np.uint16), b)


@pytest.mark.parametrize("mip", [0, 1, 2, 5])
def test_bloom_split_at_any_mip_equals_bloom(hip_ctx, mip):
    """sah_bloom_mip_rows over a partition of every mip up to `mip` + sah_bloom_from_mip(mip) == sah_bloom — the pyramid cut at the mip a
    sharded frame exchanges (mip 1 since round 4); odd extents, where a mip is not exactly half of its source"""
    import torch
    w, h = 301, 171
    scene = synth.hdr_scene(w, h, seed=25)
    want, _ = _run_post_hip(hip_ctx, scene, w, h)
    sc = util.to_torch(scene.view(np.uint16))
    mips = [torch.full(m.shape, 0x3C00, dtype=torch.int16, device="cuda") for m in _mips_np(w, h)]
    chain = images.mipchain(mips)
    sp = images.plane(sc, _abi.FORMAT_R16G16B16A16_SFLOAT)
    for m in range(mip + 1):
        rows = mips[m].shape[0]
        cuts = sorted({0, rows // 3, rows // 3 + 1, (2 * rows) // 3, rows})
        for r0, r1 in zip(cuts, cuts[1:]):
            hip_ctx.bloom_mip_rows(sp, chain, m, r0, r1)
    hip_ctx.bloom_from_mip(sp, chain, mip)
    torch.cuda.synchronize()
    for a, b in zip(mips, want):
        assert np.array_equal(util.from_torch(a, np.uint16), b)
    with pytest.raises(Exception):
        hip_ctx.bloom_mip_rows(sp, chain, 6, 0, 1)  # no such mip


@pytest.mark.parametrize("case", ["no_mips", "one_mip", "three_mips", "output_2x", "output_smaller", "wide_21_9"])
def test_tonemap_unusual_chains(hip_ctx, case):
    """Fewer than six bloom mips (the kernel's staging loads are unconditional: absent mips alias the scene), an output that is not
    the scene's resolution (a smaller one leaves the staged rectangles: whole-workgroup fallback to the per-pixel filter) and an
    aspect ratio beyond the staged reach."""
    import torch
    o = util.oracle()
    w, h = (336, 144) if case == "wide_21_9" else (192, 108)
    ow, oh = {"output_2x": (384, 216), "output_smaller": (120, 70)}.get(case, (w, h))
    n = {"no_mips": 0, "one_mip": 1, "three_mips": 3}.get(case, 6)
    scene = synth.hdr_scene(w, h, seed=29).view(np.uint16)
    full = _mips_np(w, h)
    sp = images.plane(scene, _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert o.orc_bloom(C.byref(sp), C.byref(images.mipchain(full))) == 0
    mips = full[:n]
    ref = np.zeros((oh, ow, 4), dtype=np.uint8)
    assert o.orc_tonemap(C.byref(sp), C.byref(images.mipchain(mips)), C.byref(images.plane(ref, _abi.FORMAT_R8G8B8A8_SRGB)), 0, 0) == 0
    tm = [util.to_torch(m) for m in mips]
    out = torch.zeros((oh, ow, 4), dtype=torch.uint8, device="cuda")
    sc = util.to_torch(scene)
    hip_ctx.tonemap(images.plane(sc, _abi.FORMAT_R16G16B16A16_SFLOAT), images.mipchain(tm), images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    if n:
        assert not np.array_equal(ref, np.zeros_like(ref))
    assert np.array_equal(got, ref), f"{case}: {int((got != ref).sum())} of {got.size} codes differ"


# ---- tolerance mode of the composite (SAH_TONEMAP_TOLERANCE_1CODE): within one code of the strict kernel, which the oracle pins ------
def _both_modes(ctx, scene_u16, mips_np, ow, oh, rows=None):
    import torch
    sc = util.to_torch(scene_u16)
    tm = [util.to_torch(m) for m in mips_np]
    sp, mc = images.plane(sc, _abi.FORMAT_R16G16B16A16_SFLOAT), images.mipchain(tm)
    outs = []
    for flags in (0, _abi.TONEMAP_TOLERANCE_1CODE):
        out = torch.zeros((oh, ow, 4), dtype=torch.uint8, device="cuda")
        op = images.plane(out, _abi.FORMAT_R8G8B8A8_SRGB)
        for r0, r1 in (rows or [(0, 0)]):
            ctx.tonemap(sp, mc, op, r0, r1, flags=flags)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    return outs


def _code_histogram(name, strict, tol):
    d = np.abs(strict.astype(np.int32) - tol.astype(np.int32))
    hist = np.bincount(np.minimum(d.reshape(-1), 3), minlength=4)
    print(f"{name}: |code difference| histogram [0, 1, 2, >=3] = {hist.tolist()} ({100.0 * hist[1] / d.size:.4f} % at 1)")
    return d


@pytest.mark.parametrize("size", [(256, 144), (250, 130), (1920, 1080), (333, 187), (64, 36), (40, 24), (9, 5), (2048, 96), (97, 512)])
def test_tonemap_tolerance_mode_within_one_code(hip_ctx, size):
    w, h = size
    scene = synth.hdr_scene(w, h, seed=21)
    if size == (333, 187):  # signed, huge, inf and NaN texels
        f = scene.view(np.float16).reshape(h, w, 4)
        rng = np.random.default_rng(3)
        ys, xs = rng.integers(0, h, 400), rng.integers(0, w, 400)
        f[ys[:150], xs[:150], :3] *= np.float16(-1.0)
        f[ys[150:300], xs[150:300], :3] = np.float16(60000.0)
        f[ys[300:350], xs[300:350], 0] = np.float16(np.inf)
        f[ys[350:], xs[350:], 1] = np.float16(np.nan)
    ref_mips, ref_out = _run_post_oracle(scene.view(np.uint16), w, h)
    strict, tol = _both_modes(hip_ctx, scene.view(np.uint16), ref_mips, w, h)
    assert np.array_equal(strict, ref_out)  # strict is the oracle's image
    d = _code_histogram(f"tonemap tolerance {w}x{h}", strict, tol)
    if size == (333, 187):
        # the bound is stated for finite texels.  An inf / NaN texel poisons the pixels whose taps reach it in both modes, but which products
        # meet a zero weight (0 * inf = NaN) depends on the order of evaluation, so the rim of a poisoned region may differ
        assert (d <= 1).mean() > 0.97
    else:
        assert d.max() <= 1
        assert (d == 1).mean() < 0.01


def test_tonemap_tolerance_mode_rows_and_unusual_chains(hip_ctx):
    o = util.oracle()
    for case, (w, h), (ow, oh), n in (("rows", (200, 117), (200, 117), 6), ("one_mip", (192, 108), (192, 108), 1), ("three_mips", (192, 108), (192, 108), 3),
                                      ("output_2x", (192, 108), (384, 216), 6), ("output_smaller", (192, 108), (120, 70), 6), ("wide_21_9", (336, 144), (336, 144), 6),
                                      ("no_mips", (192, 108), (192, 108), 0)):
        scene = synth.hdr_scene(w, h, seed=29).view(np.uint16)
        full = _mips_np(w, h)
        sp = images.plane(scene, _abi.FORMAT_R16G16B16A16_SFLOAT)
        assert o.orc_bloom(C.byref(sp), C.byref(images.mipchain(full))) == 0
        rows = [(0, 27), (27, 54), (54, 85), (85, 117)] if case == "rows" else None
        strict, tol = _both_modes(hip_ctx, scene, full[:n], ow, oh, rows=rows)
        d = _code_histogram(f"tonemap tolerance {case}", strict, tol)
        assert d.max() <= 1, case
