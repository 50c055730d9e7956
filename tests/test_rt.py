"""Ray tracing (first slice): acceleration structure, RTAO, sun shadow mask.
CPU: known answers for the hit rules of include/sah_hip.h on the oracle (which tests every triangle against every ray).
GPU: HIP (box hierarchy) against the oracle, bit for bit, on the atrium, triangle soups with alpha-tested CUTOUT geometry, degenerate /
non-finite / empty input, and sorts of more than one LDS chunk."""
import ctypes as C
import math

import numpy as np
import pytest

from androidrenderer_amd import _abi, images, mesh, scene, synth
from tests import util


class RtCase:
    """A scene, a view, the G-buffer planes the generators read (rasterised by the ORACLE, so the two sides get identical inputs), a
    blue-noise stand-in and a sun."""

    def __init__(self, m, width, height, view=None, seed=1, gbuffer=None, noise_size=128):
        self.mesh, self.width, self.height = m, width, height
        self.view = view or scene.SceneView.default(width, height)
        self.arrays = m.arrays()
        self.host_keep = []
        self.host_geo = mesh.geometry(mesh.with_counts(self.arrays), self.host_keep)
        g = synth.rng(seed)
        self.noise = g.integers(0, 256, (noise_size, noise_size, 4), dtype=np.uint8)
        if gbuffer is None:
            gbuffer = {"color": np.zeros((height, width, 4), np.uint8), "normals": np.zeros((height, width, 4), np.uint16),
                       "data": np.zeros((height, width, 4), np.uint8), "emission": np.zeros((height, width, 4), np.uint8),
                       "depth": np.zeros((height, width), np.float32)}
            gb = images.gbuffer(gbuffer)
            assert util.oracle().orc_gbuffer_render(C.byref(self.host_geo), C.byref(self.view.gpu_data), C.byref(gb), None) == 0
        self.gbuffer = gbuffer
        self.sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_RT, num_shadow_samples=4.0)

    def planes(self, depth, normals, noise, out):
        return (images.plane(depth, _abi.FORMAT_D32_SFLOAT), images.plane(normals, _abi.FORMAT_R16G16B16A16_SFLOAT),
                images.plane(noise, _abi.FORMAT_R8G8B8A8_UNORM), images.plane(out, _abi.FORMAT_R32_SFLOAT))

    def oracle_rtao(self, spp=1, radius=1.0):
        out = np.zeros((self.height, self.width), np.float32)
        d, n, z, o = self.planes(self.gbuffer["depth"], self.gbuffer["normals"], self.noise, out)
        assert util.oracle().orc_rtao(C.byref(self.host_geo), C.byref(self.view.gpu_data), C.byref(d), C.byref(n), C.byref(z), spp, radius, C.byref(o)) == 0
        return out

    def oracle_mask(self):
        out = np.zeros((self.height, self.width), np.float32)
        d, n, z, o = self.planes(self.gbuffer["depth"], self.gbuffer["normals"], self.noise, out)
        assert util.oracle().orc_sun_shadow_mask(C.byref(self.host_geo), C.byref(self.view.gpu_data), C.byref(self.sun.constants), C.byref(d), C.byref(n),
                                                 C.byref(z), C.byref(o)) == 0
        return out

    # ---- HIP side (device copies made on first use) ----
    def device(self):
        if not hasattr(self, "_dev"):
            import torch
            dev_arrays = mesh.to_device(self.arrays)
            keep = []
            self._dev = {"geo": mesh.geometry(dev_arrays, keep), "keep": keep,
                         "depth": torch.from_numpy(self.gbuffer["depth"]).cuda(), "normals": torch.from_numpy(self.gbuffer["normals"].view(np.int16)).cuda(),
                         "noise": torch.from_numpy(self.noise).cuda()}
        return self._dev

    def hip_build(self, ctx):
        return ctx.rt_build(self.device()["geo"])

    def hip_rtao(self, ctx, spp=1, radius=1.0):
        import torch
        dv = self.device()
        out = torch.full((self.height, self.width), -7.0, dtype=torch.float32, device="cuda")
        d, n, z, o = self.planes(dv["depth"], dv["normals"], dv["noise"], out)
        ctx.rtao(self.view.gpu_data, d, n, z, spp, radius, o)
        torch.cuda.synchronize()
        return out.cpu().numpy()

    def hip_mask(self, ctx):
        import torch
        dv = self.device()
        out = torch.full((self.height, self.width), -7.0, dtype=torch.float32, device="cuda")
        d, n, z, o = self.planes(dv["depth"], dv["normals"], dv["noise"], out)
        ctx.sun_shadow_mask(self.view.gpu_data, self.sun.constants, d, n, z, o)
        torch.cuda.synchronize()
        return out.cpu().numpy()


def _same_bits(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _floor_and_plate(plate_type=_abi.PRIMITIVE_TYPE_SOLID, plate_alpha=255, threshold=0.5, plate_y=1.0):
    """a floor quad at y = 0 and a plate above its +z half: rays that go straight up from the floor meet the plate iff z > 0"""
    m = mesh.Mesh()
    mat = m.add_material(mesh.material())
    cut = m.add_material(mesh.material(opacity_threshold=threshold))
    floor = [(-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6)]
    m.add_primitive(floor, [(0, 1, 0)] * 4, (0, 2, 1, 0, 3, 2), mat)
    plate = [(-6, plate_y, 0), (6, plate_y, 0), (6, plate_y, 6), (-6, plate_y, 6)]
    col = np.full(4, (plate_alpha << 24) | 0xffffff, np.uint32)
    m.add_primitive(plate, [(0, -1, 0)] * 4, (0, 1, 2, 0, 2, 3), cut if plate_type == _abi.PRIMITIVE_TYPE_CUTOUT else mat, ptype=plate_type, colors=col)
    return m


def _top_down_view(w, h):
    v = scene.SceneView()
    v.rotate(math.radians(-89.0), math.radians(90.0))  # looking (almost) straight down from above the origin
    v.set_position([0.0, 0.5, 0.0])                    # below the plate: the camera sees the floor only
    v.set_render_resolution(w, h)
    v.set_perspective_projection(75.0, float(w) / float(h), 0.05)
    v.update_transforms()
    return v


def _floor_positions(case):
    """world-space position of every pixel, as the generators compute it"""
    o = util.oracle()
    pos = np.zeros((case.height, case.width, 3), np.float32)
    out = (C.c_float * 3)()
    for y in range(case.height):
        for x in range(case.width):
            o.orc_worldspace_location_slang(C.byref(case.view.gpu_data), x, y, float(case.gbuffer["depth"][y, x]), out)
            pos[y, x] = out[:]
    return pos


def test_rtao_known_answer_plate_over_floor():
    case = RtCase(_floor_and_plate(), 32, 18, view=_top_down_view(32, 18))
    case.noise[...] = (128, 255, 128, 0)  # direction normalize((1/255, 1, 1/255)): up
    assert (case.gbuffer["depth"] > 0).all()
    pos = _floor_positions(case)
    assert np.abs(pos[..., 1]).max() < 1e-3  # the camera sees the floor
    ao = case.oracle_rtao(spp=1, radius=5.0)
    clear = np.abs(pos[..., 2]) > 0.05  # pixels whose ray passes the plate's z = 0 edge within the slope of the direction are not asserted
    assert np.array_equal(ao[clear], np.where(pos[..., 2][clear] > 0, 0.0, 1.0).astype(np.float32))
    # tmax is exclusive and finite: a plate farther than the radius does not occlude
    assert (case.oracle_rtao(spp=1, radius=0.9) == 1.0).all()
    # spp > 1 is the same ray spp times (quirk: one noise texel for all): the same image; spp = 0 divides 0 by 0
    assert _same_bits(case.oracle_rtao(spp=3, radius=5.0), ao)
    assert np.isnan(case.oracle_rtao(spp=0, radius=5.0)).all()


def test_rtao_ignores_cutout_geometry_and_hits_back_faces():
    solid = RtCase(_floor_and_plate(), 24, 14, view=_top_down_view(24, 14))
    cutout = RtCase(_floor_and_plate(plate_type=_abi.PRIMITIVE_TYPE_CUTOUT), 24, 14, view=_top_down_view(24, 14))
    for c in (solid, cutout):
        c.noise[...] = (128, 255, 128, 0)
    assert (solid.oracle_rtao(radius=5.0) == 0.0).any()         # the plate's normal faces the ray's origin or not: no face culling
    assert (cutout.oracle_rtao(radius=5.0) == 1.0).all()        # RAY_FLAG_CULL_NON_OPAQUE


def test_shadow_mask_alpha_test_and_untraced_pixels():
    up_sun = [0.0, -1.0, 0.0]  # light travels down: L = up
    for alpha, threshold, occludes in ((255, 0.5, True), (64, 0.5, False), (128, 0.5, True), (127, 0.5, False)):
        case = RtCase(_floor_and_plate(plate_type=_abi.PRIMITIVE_TYPE_CUTOUT, plate_alpha=alpha, threshold=threshold), 24, 14, view=_top_down_view(24, 14))
        case.sun.set_direction(up_sun)
        case.sun.constants.direction_and_tan_size[3] = 0.0  # no cone: every sample is L itself
        mask = case.oracle_mask()
        pos = _floor_positions(case)
        under = pos[..., 2] > 0.05
        # alpha = (texel 1 * tint 1) * unpack(pack(colour alpha)): 128 / 255 = 0.50196 > 0.5 is accepted, 127 / 255 is not
        assert (mask[under] == (0.0 if occludes else 1.0)).all(), (alpha, mask[under])
        assert (mask[pos[..., 2] < -0.05] == 1.0).all()
    # surfaces facing away from the light and sky pixels are not traced: 1.0
    case = RtCase(_floor_and_plate(), 24, 14, view=_top_down_view(24, 14))
    case.sun.set_direction([0.0, 1.0, 0.0])  # light from below: ndotl = 0 on the floor
    assert (case.oracle_mask() == 1.0).all()
    case.gbuffer["depth"][...] = 0.0
    case.sun.set_direction(up_sun)
    assert (case.oracle_mask() == 1.0).all()
    assert (case.oracle_rtao(radius=5.0) == 1.0).all()  # rtao has no depth test: the origin of a sky pixel is not finite, nothing is hit


def test_shadow_mask_counts_samples():
    case = RtCase(_floor_and_plate(), 24, 14, view=_top_down_view(24, 14))
    case.sun.set_direction([0.0, -1.0, 0.0])
    for n in (1.0, 3.0, 8.0):
        case.sun.constants.num_shadow_samples = n
        mask = case.oracle_mask()
        assert set(np.unique(mask)).issubset({np.float32(k) / np.float32(n) for k in range(int(n) + 1)})
    case.sun.constants.num_shadow_samples = 0.0
    assert np.isnan(case.oracle_mask()).all()  # shadow / 0 = 0 / 0, as in the shader


def test_watertight_shared_edges_and_degenerate_triangles():
    # a fan of thin triangles around a centre, all sharing edges; rays from below straight up through points ON the shared edges
    m = mesh.Mesh()
    mat = m.add_material(mesh.material())
    n = 24
    ring = [(3.0 * math.cos(2 * math.pi * k / n), 2.0, 3.0 * math.sin(2 * math.pi * k / n)) for k in range(n)]
    pos = [(0.0, 2.0, 0.0)] + ring
    idx = []
    for k in range(n):
        idx += [0, 1 + k, 1 + (k + 1) % n]
    idx += [1, 1, 2, 0, 5, 5]  # degenerate triangles never hit
    m.add_primitive(pos, [(0, -1, 0)] * len(pos), idx, mat)
    m.add_primitive([(-9, 0, -9), (9, 0, -9), (9, 0, 9), (-9, 0, 9)], [(0, 1, 0)] * 4, (0, 2, 1, 0, 3, 2), mat)
    case = RtCase(m, 48, 27, view=_top_down_view(48, 27))
    case.view.set_position([0.0, 1.0, 0.0])
    case.view.update_transforms()
    case = RtCase(m, 48, 27, view=case.view)
    case.noise[...] = (128, 255, 128, 0)
    ao = case.oracle_rtao(radius=10.0)
    pos = _floor_positions(case)
    r = np.hypot(pos[..., 0], pos[..., 2])
    floor = np.abs(pos[..., 1]) < 1e-3
    assert (ao[floor & (r < 2.8)] == 0.0).all()  # no ray slips between two triangles of the fan
    assert (ao[floor & (r > 3.1)] == 1.0).all()
    stats = (C.c_uint32 * 4)()
    pad = C.c_float()
    assert util.oracle().orc_rt_stats(C.byref(case.host_geo), stats, C.byref(pad)) == 0
    assert stats[0] == n + 2 + 2 and stats[1] == 0 and pad.value == np.float32(9.0) * np.float32(2.0 ** -16)


def test_non_finite_vertices_are_left_out():
    m = mesh.Mesh()
    mat = m.add_material(mesh.material())
    m.add_primitive([(0, 0, 0), (1, 0, 0), (0, 1, 0), (float("nan"), 0, 0), (float("inf"), 1, 1)], [(0, 0, 1)] * 5, (0, 1, 2, 0, 1, 3, 0, 2, 4), mat)
    g = mesh.geometry(mesh.with_counts(m.arrays()), [])
    stats = (C.c_uint32 * 4)()
    assert util.oracle().orc_rt_stats(C.byref(g), stats, None) == 0
    assert list(stats)[:2] == [1, 2]


# ---- GPU: HIP == oracle ---------------------------------------------------------------------------------------------------------

def _check_both(ctx, case, spp=1, radius=1.5, expect_tris=None):
    stats = case.hip_build(ctx)
    o_stats = (C.c_uint32 * 4)()
    assert util.oracle().orc_rt_stats(C.byref(case.host_geo), o_stats, None) == 0
    assert stats[0] == o_stats[0] and stats[1] == o_stats[1], (stats, list(o_stats))
    if expect_tris is not None:
        assert stats[0] == expect_tris
    ao_h, ao_o = case.hip_rtao(ctx, spp, radius), case.oracle_rtao(spp, radius)
    assert _same_bits(ao_h, ao_o), f"rtao: {int((ao_h.view(np.uint32) != ao_o.view(np.uint32)).sum())} texels differ"
    mk_h, mk_o = case.hip_mask(ctx), case.oracle_mask()
    assert _same_bits(mk_h, mk_o), f"shadow mask: {int((mk_h.view(np.uint32) != mk_o.view(np.uint32)).sum())} texels differ"
    return ao_o, mk_o


@pytest.mark.gpu
def test_hip_known_answer_scenes(hip_ctx):
    for plate_type in (_abi.PRIMITIVE_TYPE_SOLID, _abi.PRIMITIVE_TYPE_CUTOUT):
        case = RtCase(_floor_and_plate(plate_type=plate_type, plate_alpha=200), 32, 18, view=_top_down_view(32, 18))
        case.sun.set_direction([0.1, -1.0, 0.2])
        _check_both(hip_ctx, case, radius=5.0, expect_tris=4)


@pytest.mark.gpu
@pytest.mark.parametrize("subdiv,size", [(1, (96, 54)), (4, (64, 36))])
def test_hip_atrium(hip_ctx, subdiv, size):
    case = RtCase(mesh.atrium(subdiv), *size)
    ao, mask = _check_both(hip_ctx, case, radius=1.0, expect_tris=31 * 12 * subdiv * subdiv)  # subdiv 4: 5952 triangles = three sort chunks
    assert 0.02 < (ao == 0.0).mean() < 0.9       # something is occluded, something is not
    assert 0.02 < (mask < 1.0).mean() < 0.98


@pytest.mark.gpu
@pytest.mark.parametrize("seed,textured", [(1, False), (2, True), (3, True), (4, False)])
def test_hip_triangle_soups_with_cutouts(hip_ctx, seed, textured):
    m = mesh.random_soup(seed, triangles=400, textured=textured)
    case = RtCase(m, 64, 36, seed=seed)
    case.sun.set_direction([0.3, -1.0, 0.2])
    case.sun.constants.num_shadow_samples = 3.0
    ao, mask = _check_both(hip_ctx, case, spp=2, radius=4.0)
    assert (mask < 1.0).any() and (mask == 1.0).any()


@pytest.mark.gpu
def test_hip_random_planes_against_the_oracle(hip_ctx):
    """rays from arbitrary origins: random depth and normals instead of a rasterised G-buffer (grazing rays, origins inside geometry,
    sky pixels, non-finite normals)"""
    g = synth.rng(77)
    m = mesh.random_soup(9, triangles=600, textured=True)
    W, H = 80, 45
    gb = {"depth": np.where(g.uniform(size=(H, W)) < 0.1, 0.0, g.uniform(0.003, 0.2, (H, W))).astype(np.float32),
          "normals": g.normal(size=(H, W, 4)).astype(np.float16).view(np.uint16)}
    gb["normals"][0, :4] = 0x7c00  # inf
    gb["normals"][1, :4] = 0       # zero vector: normalize gives NaN
    case = RtCase(m, W, H, seed=5, gbuffer=gb)
    case.sun.constants.num_shadow_samples = 2.0
    _check_both(hip_ctx, case, spp=1, radius=6.0)


@pytest.mark.gpu
def test_hip_empty_and_degenerate_scenes(hip_ctx):
    empty = mesh.Mesh()
    empty.add_material(mesh.material())
    case = RtCase(empty, 16, 9, gbuffer={"depth": np.full((9, 16), 0.05, np.float32), "normals": np.full((9, 16, 4), 0x3c00, np.uint16)})
    ao, mask = _check_both(hip_ctx, case, expect_tris=0)
    assert (ao == 1.0).all() and (mask == 1.0).all()
    bad = mesh.Mesh()
    mat = bad.add_material(mesh.material())
    bad.add_primitive([(0, 0, 0), (1, 0, 0), (0, 1, 0), (float("nan"), 0, 0), (float("inf"), 1, 1)], [(0, 0, 1)] * 5, (0, 1, 2, 0, 1, 3, 0, 2, 4, 1, 1, 1), mat)
    case = RtCase(bad, 16, 9, gbuffer={"depth": np.full((9, 16), 0.05, np.float32), "normals": np.full((9, 16, 4), 0x3c00, np.uint16)})
    _check_both(hip_ctx, case, expect_tris=2)


@pytest.mark.gpu
def test_hip_rebuild_replaces_the_structure(hip_ctx):
    a = RtCase(_floor_and_plate(), 24, 14, view=_top_down_view(24, 14))
    b = RtCase(_floor_and_plate(plate_y=3.0), 24, 14, view=_top_down_view(24, 14))
    for c in (a, b):
        c.noise[...] = (128, 255, 128, 0)
    a.hip_build(hip_ctx)
    first = a.hip_rtao(hip_ctx, radius=2.0)
    b.hip_build(hip_ctx)
    second = b.hip_rtao(hip_ctx, radius=2.0)
    assert _same_bits(first, a.oracle_rtao(radius=2.0)) and _same_bits(second, b.oracle_rtao(radius=2.0))
    assert (first == 0.0).any() and (second == 1.0).all()


@pytest.mark.gpu
def test_hip_rejects_bad_arguments(hip_ctx):
    from androidrenderer_amd import lib
    import torch
    fresh = lib.Context(device=0)
    case = RtCase(_floor_and_plate(), 16, 9, view=_top_down_view(16, 9))
    with pytest.raises(lib.SahError):  # no structure yet
        case.hip_rtao(fresh)
    case.hip_build(fresh)
    case.sun.constants.num_shadow_samples = 1.0e9
    with pytest.raises(lib.SahError):
        case.hip_mask(fresh)
    with pytest.raises(lib.SahError):
        case.hip_rtao(fresh, spp=100000)
    torch.cuda.synchronize()
    fresh.close()
