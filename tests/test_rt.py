"""Ray tracing: acceleration structure, RTAO, sun shadow mask, and the GI rays (probe tracing, RTGI) with their hit shading.
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

    # ---- GI rays: sky LUTs, the irradiance cache the probe misses sample, cascades centred on `centre` ----
    def gi_arrays(self):
        if not hasattr(self, "_gi"):
            luts, at = synth.sky_luts(self.gi_seed + 200), synth.probe_atlases(self.gi_seed + 600)
            self._gi = {"sky_t": luts["transmittance"], "sky_v": luts["sky_view"], "irr": at["irradiance"], "pdepth": at["depth"], "val": at["validity"]}
        return self._gi

    gi_seed = 3
    cascade_centre = (0.0, 1.0, 0.0)
    cascade_spacing = 0.5

    def _sky(self, a):
        return _abi.SkyLuts(images.plane(a["sky_t"], _abi.FORMAT_R16G16B16A16_SFLOAT), images.plane(a["sky_v"], _abi.FORMAT_R16G16B16A16_SFLOAT))

    def probe_desc(self, a, probes_ptr, num_probes, noise, results):
        """_abi.ProbeTraceDesc over host or device arrays `a` (gi_arrays() keys); returns (desc, keepalive)"""
        d = _abi.ProbeTraceDesc()
        for c, (cmin, spacing) in enumerate(util.rt_gi_cascades(self.cascade_centre, self.cascade_spacing)):
            d.cascades[c].probe_spacing = spacing
            for i in range(3):
                d.cascades[c].min[i] = cmin[i]
        sky, nz = self._sky(a), images.plane(noise, _abi.FORMAT_R8G8B8A8_UNORM)
        d.probes_to_update, d.num_probes = probes_ptr, num_probes
        d.sun, d.sky, d.noise = C.pointer(self.sun.constants), C.pointer(sky), C.pointer(nz)
        d.probe_irradiance = images.volume(a["irr"], _abi.FORMAT_B10G11R11_UFLOAT_PACK32)
        d.probe_depth = images.volume(a["pdepth"], _abi.FORMAT_R16G16_SFLOAT)
        d.probe_validity = images.volume(a["val"], _abi.FORMAT_R8_UNORM)
        d.probe_size[0], d.probe_size[1] = 5, 6
        d.trace_results = images.volume(results, _abi.FORMAT_R16G16B16A16_SFLOAT)
        return d, (sky, nz)

    def oracle_probe_trace(self, probes):
        """probes: (N, 3) uint32 -> (N, 20, 20, 4) float16"""
        probes = np.ascontiguousarray(probes, np.uint32)
        out = np.full((max(len(probes), 1), 20, 20, 4), 0x7bff, np.uint16)
        d, keep = self.probe_desc(self.gi_arrays(), probes.ctypes.data, len(probes), self.noise, out)
        assert util.oracle().orc_probe_trace(C.byref(self.host_geo), C.byref(d)) == 0
        return out[:len(probes)].view(np.float16)

    def oracle_rtgi(self):
        """-> (ray_buffer, ray_irradiance), (H, W, 4) float16 each; texels the generator skips keep the 0x7bff fill"""
        rb, ri = (np.full((self.height, self.width, 4), 0x7bff, np.uint16) for _ in range(2))
        a = self.gi_arrays()
        d, n, z, _ = self.planes(self.gbuffer["depth"], self.gbuffer["normals"], self.noise, np.zeros((1, 1), np.float32))
        sky = self._sky(a)
        assert util.oracle().orc_rtgi_trace(C.byref(self.host_geo), C.byref(self.view.gpu_data), C.byref(self.sun.constants), C.byref(sky), C.byref(d), C.byref(n),
                                            C.byref(z), C.byref(images.plane(rb, _abi.FORMAT_R16G16B16A16_SFLOAT)),
                                            C.byref(images.plane(ri, _abi.FORMAT_R16G16B16A16_SFLOAT))) == 0
        return rb.view(np.float16), ri.view(np.float16)

    # ---- HIP side (device copies made on first use) ----
    def device(self):
        if not hasattr(self, "_dev"):
            import torch
            from androidrenderer_amd.frame import to_torch
            dev_arrays = mesh.to_device(self.arrays)
            keep = []
            self._dev = {"geo": mesh.geometry(dev_arrays, keep), "keep": keep,
                         "depth": torch.from_numpy(self.gbuffer["depth"]).cuda(), "normals": torch.from_numpy(self.gbuffer["normals"].view(np.int16)).cuda(),
                         "noise": torch.from_numpy(self.noise).cuda()}
            self._dev["gi"] = {k: to_torch(v) for k, v in self.gi_arrays().items()}
        return self._dev

    def hip_probe_trace(self, ctx, probes):
        import torch
        dv = self.device()
        probes = np.ascontiguousarray(probes, np.uint32)
        pd = torch.from_numpy(probes.view(np.int32).reshape(-1)).cuda() if len(probes) else torch.zeros(3, dtype=torch.int32, device="cuda")
        out = torch.full((max(len(probes), 1), 20, 20, 4), 0x7bff, dtype=torch.int16, device="cuda")
        d, keep = self.probe_desc(dv["gi"], pd.data_ptr(), len(probes), dv["noise"], out)
        ctx.probe_trace(d)
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.float16)[:len(probes)]

    def hip_rtgi(self, ctx):
        import torch
        dv = self.device()
        rb, ri = (torch.full((self.height, self.width, 4), 0x7bff, dtype=torch.int16, device="cuda") for _ in range(2))
        d, n, z, _ = self.planes(dv["depth"], dv["normals"], dv["noise"], torch.zeros((1, 1), dtype=torch.float32, device="cuda"))
        ctx.rtgi_trace(self.view.gpu_data, self.sun.constants, self._sky(dv["gi"]), d, n, z, images.plane(rb, _abi.FORMAT_R16G16B16A16_SFLOAT),
                       images.plane(ri, _abi.FORMAT_R16G16B16A16_SFLOAT))
        torch.cuda.synchronize()
        return rb.cpu().numpy().view(np.float16), ri.cpu().numpy().view(np.float16)

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


def test_rtao_infinite_and_nan_distance_mean_the_largest_finite_float():
    """include/sah_hip.h "tmax": minNum(tmax, FLT_MAX).  (The HIP walk's boxes for absent children rely on a finite tmax.)"""
    case = RtCase(_floor_and_plate(), 24, 14, view=_top_down_view(24, 14))
    case.noise[...] = (128, 255, 128, 0)
    far = case.oracle_rtao(radius=3.0e38)
    assert (far == 0.0).any() and (far == 1.0).any()
    assert _same_bits(case.oracle_rtao(radius=float("inf")), far)
    assert _same_bits(case.oracle_rtao(radius=float("nan")), far)


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


# ---- GI rays: known answers on the oracle ----------------------------------------------------------------------------------------

def _texel_directions():
    o = util.oracle()
    dirs = np.zeros((20, 20, 3), np.float32)
    c2, d3 = (C.c_float * 2)(), (C.c_float * 3)()
    for ty in range(20):
        for tx in range(20):
            o.orc_octahedral_direction_of_texel(tx, ty, 20, 20, c2, d3)
            dirs[ty, tx] = d3[:]
    return dirs


E_FACTOR = np.float32(0.0031415927)


def test_probe_rays_distance_facing_and_miss():
    case = RtCase(_floor_and_plate(), 24, 14, view=_top_down_view(24, 14))
    case.sun.set_direction([0.1, -1.0, 0.2])
    # cascade 0: min = centre - extent / 2 + 0.013 = (-7.987, -0.987, -7.987), spacing 0.5
    between = (16, 3, 22)   # (0.013, 0.513, 3.013): above the floor, below the plate
    outside = (16, 3, 10)   # (0.013, 0.513, -2.987): above the floor, no plate overhead
    below = (16, 1, 22)     # (0.013, -0.487, 3.013): under the floor
    far = (16, 3 + 24, 16)  # last cascade (spacing 4): (0.052, -2.948, 0.052)
    r = case.oracle_probe_trace(np.array([between, outside, below, far], np.uint32)).astype(np.float32)
    dirs = _texel_directions()
    up, down = dirs[..., 1] > 0.3, dirs[..., 1] < -0.3
    # the floor's (v1 - v0) x (v2 - v0) is +y: FRONT facing from above — distance +t, lit; from below BACK facing: -t and black
    for k in (0, 1):
        assert np.allclose(r[k][down][:, 3], 0.513 / -dirs[down][:, 1], rtol=3e-3)
        assert (r[k][down][:, :3] > 0).all() and len(np.unique(r[k][down][:, 0])) == 1  # one normal, one light vector: one value
    assert np.allclose(r[2][up][:, 3], -(0.487 / dirs[up][:, 1]), rtol=3e-3) and (r[2][up][:, :3] == 0).all()
    # the plate (normal -y, front facing from below): hit, but the light comes from above: black
    assert np.allclose(r[0][up][:, 3], 0.487 / dirs[up][:, 1], rtol=3e-3) and (r[0][up][:, :3] == 0).all()
    # misses report the ray's full length — 4 x the next cascade's spacing, 8192 in the last — and carry the next cascade's irradiance or 10 x sky
    assert (r[1][up][:, 3] == 4.0).all() and (r[2][down][:, 3] == 4.0).all() and np.isfinite(r[1][up][:, :3]).all()
    assert (r[3][down][:, 3] == 8192.0).all() and (r[3][down][:, :3] > 0).all()
    # a probe id outside the four cascades writes zeros (the shader would read past a 4-entry array)
    assert (case.oracle_probe_trace(np.array([(1, 32, 1)], np.uint32)) == 0).all()


def _floor_with_occluder(kind):
    m = mesh.Mesh()
    mat = m.add_material(mesh.material())
    m.add_primitive([(-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6)], [(0, 1, 0)] * 4, (0, 2, 1, 0, 3, 2), mat)
    quad = [(-6, 2, -6), (6, 2, -6), (6, 2, 6), (-6, 2, 6)]
    if kind == "plate_facing_down":
        m.add_primitive(quad, [(0, -1, 0)] * 4, (0, 1, 2, 0, 2, 3), mat)
    elif kind == "plate_facing_up":
        m.add_primitive(quad, [(0, 1, 0)] * 4, (0, 2, 1, 0, 3, 2), mat)
    elif kind == "box":
        m.add_box((-6, 2, -6), (6, 3, 6), mat)
    return m


def test_gi_shadow_ray_culls_front_faces():
    """the hit stage's shadow ray carries CULL_FRONT_FACING_TRIANGLES (gltf_basic_pbr.slang:452-461): a single-sided sheet facing the
    shaded point does not shadow it; its back, or a closed box, does"""
    lit = {}
    for kind in ("none", "plate_facing_down", "plate_facing_up", "box"):
        case = RtCase(_floor_with_occluder(kind), 16, 9, view=_top_down_view(16, 9))
        case.sun.set_direction([0.0, -1.0, 0.001])
        case.sun.constants.direction_and_tan_size[3] = 0.0
        r = case.oracle_probe_trace(np.array([(16, 3, 16)], np.uint32)).astype(np.float32)[0]  # (0.013, 0.513, 0.013)
        down = _texel_directions()[..., 1] < -0.5
        lit[kind] = np.unique(r[down][:, 0])
        assert len(lit[kind]) == 1
    assert lit["none"][0] > 0 and lit["plate_facing_down"][0] == lit["none"][0]
    assert lit["plate_facing_up"][0] == 0 and lit["box"][0] == 0


def test_rtgi_rays_emission_miss_and_untraced_pixels():
    m = mesh.Mesh()
    mat = m.add_material(mesh.material())
    glow = m.add_material(mesh.material(emission=(2.0, 3.0, 4.0, 0.0)))
    m.add_primitive([(-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6)], [(0, 1, 0)] * 4, (0, 2, 1, 0, 3, 2), mat)
    m.add_primitive([(-6, 1, 0), (6, 1, 0), (6, 1, 6), (-6, 1, 6)], [(0, -1, 0)] * 4, (0, 1, 2, 0, 2, 3), glow)
    case = RtCase(m, 24, 14, view=_top_down_view(24, 14))
    case.noise[...] = (128, 255, 128, 0)
    case.sun.set_direction([0.1, -1.0, 0.2])
    for i in range(3):
        case.sun.constants.color[i] = 0.0  # sun off: the irradiance of a hit is its emission
    case.gbuffer["depth"][0, :] = 0.0      # a row of sky pixels
    rb, ri = case.oracle_rtgi()
    assert (rb[0].view(np.uint16) == 0x7bff).all() and (ri[0].view(np.uint16) == 0x7bff).all()  # not written
    pos = _floor_positions(case)[1:]
    rb, ri = rb[1:].astype(np.float32), ri[1:].astype(np.float32)
    d = np.array([1.0 / 255.0, 1.0, 1.0 / 255.0], np.float32)
    d /= np.linalg.norm(d)
    assert np.allclose(rb[..., :3], d, atol=1e-3)
    under, clear = pos[..., 2] > 0.05, pos[..., 2] < -0.05
    assert np.allclose(rb[under][:, 3], 1.0 / d[1], rtol=2e-3)
    want = (np.array([2.0, 3.0, 4.0], np.float32) * E_FACTOR).astype(np.float16).astype(np.float32)
    assert np.array_equal(ri[under][:, :3], np.broadcast_to(want, ri[under][:, :3].shape)) and (ri[..., 3] == 0).all()
    assert (rb[clear][:, 3] == 0).all() and (ri[clear][:, :3] > 0).all()  # miss: distance stays 0, sky colour
    assert len(np.unique(ri[clear][:, 0])) == 1                            # one direction: one sky colour


class _bounces:
    """remaining_bounces of the generators' rays on both sides for the length of a `with` block (the reference's generators: 0)"""

    def __init__(self, n, ctx=None):
        self.n, self.ctx = n, ctx

    def __enter__(self):
        assert util.oracle().orc_rt_set_bounces(self.n) == 0
        if self.ctx is not None:
            self.ctx.rt_set_bounces(self.n)

    def __exit__(self, *exc):
        util.oracle().orc_rt_set_bounces(0)
        if self.ctx is not None:
            self.ctx.rt_set_bounces(0)


def _floor_under(kind):
    """a floor and, above its +z half, a glowing sheet: facing down (front face to the floor), facing up, or facing down but alpha-tested"""
    m = mesh.Mesh()
    mat = m.add_material(mesh.material())
    glow = m.add_material(mesh.material(emission=(2.0, 3.0, 4.0, 0.0), opacity_threshold=0.5))
    m.add_primitive([(-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6)], [(0, 1, 0)] * 4, (0, 2, 1, 0, 3, 2), mat)
    quad = [(-6, 1, 0), (6, 1, 0), (6, 1, 6), (-6, 1, 6)]
    if kind == "down":
        m.add_primitive(quad, [(0, -1, 0)] * 4, (0, 1, 2, 0, 2, 3), glow)
    elif kind == "up":
        m.add_primitive(quad, [(0, 1, 0)] * 4, (0, 2, 1, 0, 3, 2), glow)
    elif kind == "cutout":
        m.add_primitive(quad, [(0, -1, 0)] * 4, (0, 1, 2, 0, 2, 3), glow, ptype=_abi.PRIMITIVE_TYPE_CUTOUT, colors=np.full(4, 0xffffffff, np.uint32))
    return m


def test_gi_bounce_branch_known_answers():
    """gltf_basic_pbr.slang:481-517 with remaining_bounces > 0: one more GI ray from the hit along the noise direction (into the normal's
    hemisphere), CULL_NON_OPAQUE | CULL_BACK_FACING_TRIANGLES, whose irradiance comes back through ndotl * brdf"""
    def rtgi(kind, bounces, sun_on=False):
        case = RtCase(_floor_under(kind), 24, 14, view=_top_down_view(24, 14))
        case.noise[...] = (128, 0, 128, 0)  # straight DOWN: the floor's bounce ray flips it up, the pixel's own ray too
        case.sun.set_direction([0.1, -1.0, 0.2])
        for i in range(3):
            case.sun.constants.color[i] = 1.0 if sun_on else 0.0  # (dim: the sheet's emission must stay visible beside it in fp16)
        with _bounces(bounces):
            r = case.oracle_probe_trace(np.array([(16, 3, 22), (16, 3, 10), (16, 1, 22)], np.uint32)).astype(np.float32)
        return r
    down = _texel_directions()[..., 1] < -0.5
    up = _texel_directions()[..., 1] > 0.5
    none0, none1 = rtgi("none", 0), rtgi("none", 1)
    assert (none0[0][down][:, :3] == 0).all()                      # sun off, no bounce: a lit-by-nothing floor
    sky = none1[0][down][:, :3]
    assert (sky > 0).all() and len(np.unique(sky[:, 0])) == 1      # the bounce ray leaves the scene: brdf x sky colour, one direction
    assert np.array_equal(none0[..., 3], none1[..., 3])           # distances are the first hit's
    # under the sheet's front face: its emission comes back, 2 : 3 : 4 through a white dielectric's brdf
    lit = rtgi("down", 1)
    under, beside = lit[0][down][:, :3], lit[1][down][:, :3]
    assert len(np.unique(under[:, 0])) == 1 and np.allclose(under[0, 1] / under[0, 0], 1.5, rtol=2e-2) and np.allclose(under[0, 2] / under[0, 0], 2.0, rtol=2e-2)
    assert np.array_equal(beside, none1[1][down][:, :3])          # no sheet overhead there: sky
    # ... its back face is culled (the ray goes through to the sky), and so is an alpha-tested sheet, opaque texels or not
    for kind in ("up", "cutout"):
        assert np.array_equal(rtgi(kind, 1)[0][down][:, :3], sky)
    # a back-face first hit stays black and negative whatever a bounce would bring (probe under the floor, looking up)
    assert (lit[2][up][:, :3] == 0).all() and (lit[2][up][:, 3] < 0).all()
    # the rays that start between floor and sheet and go UP hit the sheet first: emission, plus — with a bounce — what the floor below reflects
    e0, e1, e2 = (rtgi("down", b, sun_on=True)[0][up][:, :3] for b in (0, 1, 2))
    want = (np.array([2.0, 3.0, 4.0], np.float32) * E_FACTOR).astype(np.float16).astype(np.float32)
    assert np.array_equal(e0, np.broadcast_to(want, e0.shape))  # (the sun is behind the sheet)
    assert (e1 > e0).all()  # + the sunlit floor below (the floor's shadow ray culls the sheet's front face)
    assert (e2 > e1).all()  # + what that floor sees in turn: the sheet's emission


def test_closest_hit_prefers_smaller_t_then_smaller_ids():
    # two coincident emissive sheets (different primitives) above the floor and a third one farther up: the nearest wins; of the two
    # coincident ones, the smaller primitive index
    m = mesh.Mesh()
    mat = m.add_material(mesh.material())
    m.add_primitive([(-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6)], [(0, 1, 0)] * 4, (0, 2, 1, 0, 3, 2), mat)
    sheet = [(-6, 1, -6), (6, 1, -6), (6, 1, 6), (-6, 1, 6)]
    for level, emission in ((2.0, (9.0, 9.0, 9.0, 0.0)), (1.0, (1.0, 0.0, 0.0, 0.0)), (1.0, (0.0, 1.0, 0.0, 0.0))):
        m.add_primitive([(x, level, z) for (x, _, z) in sheet], [(0, -1, 0)] * 4, (0, 1, 2, 0, 2, 3), m.add_material(mesh.material(emission=emission)))
    case = RtCase(m, 16, 9, view=_top_down_view(16, 9))
    case.noise[...] = (128, 255, 128, 0)
    for i in range(3):
        case.sun.constants.color[i] = 0.0
    rb, ri = case.oracle_rtgi()
    ri = ri.astype(np.float32)
    want = (np.array([1.0, 0.0, 0.0], np.float32) * E_FACTOR).astype(np.float16).astype(np.float32)
    assert np.array_equal(ri[..., :3], np.broadcast_to(want, ri[..., :3].shape))


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
@pytest.mark.parametrize("subdiv,size", [(1, (96, 54)), (4, (64, 36)), (12, (48, 27))])
def test_hip_atrium(hip_ctx, subdiv, size):
    case = RtCase(mesh.atrium(subdiv), *size)
    # subdiv 4: 5952 triangles = three sort chunks, six refinement windows (the last one partial); subdiv 12: 53 568 = 52 windows + 320
    ao, mask = _check_both(hip_ctx, case, radius=1.0, expect_tris=31 * 12 * subdiv * subdiv)
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
@pytest.mark.parametrize("samples,tan_size", [(1.0, 0.0095), (8.0, 0.0095), (8.0, 0.0), (7.5, 0.5), (33.0, 0.02), (34.0, 0.02), (70.0, 0.1)])
def test_hip_shadow_mask_sample_counts_and_cones(hip_ctx, samples, tan_size):
    """The mask kernel's paths: sample 0 alone; beams (2 .. 33 samples: the others of a pixel in one 32-bit mask) with a narrow cone, no
    cone at all (every ray the same: inv_lo == inv_hi) and a cone so wide that directions change sign (pixels walked ray by ray); 34
    samples and more (no beams, (pixel, sample) pairs in several chunks); samples beyond the 64 tabulated noise offsets; a fractional
    bound (the shader's loop runs ceil(bound) times and divides by the bound).  Cutout geometry: a beam's rays run the any-hit stage
    one by one."""
    m = mesh.random_soup(21, triangles=500, cutout_fraction=0.4, textured=True)
    case = RtCase(m, 48, 27, seed=3)
    case.sun.set_direction([0.25, -1.0, 0.15])
    case.sun.constants.num_shadow_samples = samples
    case.sun.constants.direction_and_tan_size[3] = tan_size
    case.hip_build(hip_ctx)
    got, want = case.hip_mask(hip_ctx), case.oracle_mask()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (want < 1.0).any() and (want > 0.0).any()


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
@pytest.mark.parametrize("triangles", [5, 405, 1029])  # 7 n triangles with n = 1, 101, 257
def test_hip_infinite_ray_distance(hip_ctx, triangles):
    """tmax = +inf (and NaN) is the largest finite float.  Triangle counts that are not multiples of four leave absent children in the last
    group of every level: with an infinite tmax a ray whose direction has no negative component used to pass their [+inf, +inf] boxes and
    walk into nodes that do not exist.  Random normals and noise give rays of every sign pattern, all-positive directions included."""
    g = synth.rng(1000 + triangles)
    m = mesh.random_soup(triangles, triangles=triangles, textured=False)
    W, H = 64, 36
    gb = {"depth": g.uniform(0.003, 0.2, (H, W)).astype(np.float32), "normals": g.normal(size=(H, W, 4)).astype(np.float16).view(np.uint16)}
    gb["normals"][: H // 2] = np.array([0.57, 0.57, 0.57, 0.0], np.float16).view(np.uint16)  # hemisphere of (+, +, +)
    case = RtCase(m, W, H, seed=triangles, gbuffer=gb)
    case.noise[: 64] = (255, 255, 255, 0)  # direction normalize((1, 1, 1)): no negative component
    stats = case.hip_build(hip_ctx)
    assert stats[0] % 4 != 0
    for radius in (float("inf"), float("nan"), 3.0e38):
        got, want = case.hip_rtao(hip_ctx, radius=radius), case.oracle_rtao(radius=radius)
        assert _same_bits(got, want), (radius, int((got != want).sum()))
    assert triangles < 100 or ((want == 0.0).any() and (want == 1.0).any())


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


# ---- GPU: the GI rays, HIP == oracle ----------------------------------------------------------------------------------------------

def _probe_ids(seed, count):
    g = synth.rng(seed)
    ids = g.integers(0, 32, (count, 3)).astype(np.uint32)
    ids[0] = (16, 3, 16)    # next to the cascade centre
    ids[1] = (5, 40, 7)     # outside the four cascades: zeros
    return ids


def _check_gi(ctx, case, probes):
    case.hip_build(ctx)
    t_h, t_o = case.hip_probe_trace(ctx, probes), case.oracle_probe_trace(probes)
    assert np.array_equal(t_h.view(np.uint16), t_o.view(np.uint16)), f"probe trace: {int((t_h.view(np.uint16) != t_o.view(np.uint16)).any(-1).sum())} texels differ"
    (rb_h, ri_h), (rb_o, ri_o) = case.hip_rtgi(ctx), case.oracle_rtgi()
    assert np.array_equal(rb_h.view(np.uint16), rb_o.view(np.uint16)), f"ray buffer: {int((rb_h.view(np.uint16) != rb_o.view(np.uint16)).any(-1).sum())} texels differ"
    assert np.array_equal(ri_h.view(np.uint16), ri_o.view(np.uint16)), f"ray irradiance: {int((ri_h.view(np.uint16) != ri_o.view(np.uint16)).any(-1).sum())} texels differ"
    return t_o, rb_o, ri_o


@pytest.mark.gpu
def test_hip_gi_rays_known_answer_scenes(hip_ctx):
    for kind in ("none", "plate_facing_down", "plate_facing_up", "box"):
        case = RtCase(_floor_with_occluder(kind), 32, 18, view=_top_down_view(32, 18))
        _check_gi(hip_ctx, case, _probe_ids(3, 12))


@pytest.mark.gpu
@pytest.mark.parametrize("subdiv,size", [(1, (96, 54)), (4, (48, 27))])
def test_hip_gi_rays_atrium(hip_ctx, subdiv, size):
    case = RtCase(mesh.atrium(subdiv), *size)
    trace, rb, ri = _check_gi(hip_ctx, case, _probe_ids(5, 40))
    d = trace.astype(np.float32)[..., 3]
    assert (d > 0).any() and (d < 0).any()                   # probes inside boxes see back faces
    assert (rb.astype(np.float32)[..., 3] > 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,textured", [(1, False), (2, True), (3, True)])
def test_hip_gi_rays_triangle_soups_with_cutouts(hip_ctx, seed, textured):
    case = RtCase(mesh.random_soup(seed, triangles=400, textured=textured), 64, 36, seed=seed)
    case.sun.set_direction([0.3, -1.0, 0.2])
    _check_gi(hip_ctx, case, _probe_ids(seed, 30))


@pytest.mark.gpu
def test_hip_gi_rays_random_planes_and_empty_scene(hip_ctx):
    g = synth.rng(78)
    W, H = 80, 45
    gb = {"depth": np.where(g.uniform(size=(H, W)) < 0.1, 0.0, g.uniform(0.003, 0.2, (H, W))).astype(np.float32),
          "normals": g.normal(size=(H, W, 4)).astype(np.float16).view(np.uint16)}
    gb["normals"][0, :4] = 0x7c00
    gb["normals"][1, :4] = 0
    _check_gi(hip_ctx, RtCase(mesh.random_soup(9, triangles=600, textured=True), W, H, seed=5, gbuffer=gb), _probe_ids(9, 24))
    empty = mesh.Mesh()
    empty.add_material(mesh.material())
    case = RtCase(empty, 16, 9, gbuffer={"depth": np.full((9, 16), 0.05, np.float32), "normals": np.full((9, 16, 4), 0x3c00, np.uint16)})
    trace, rb, ri = _check_gi(hip_ctx, case, _probe_ids(2, 8))
    assert (rb.astype(np.float32)[..., 3] == 0).all()
    assert _check_gi(hip_ctx, case, np.zeros((0, 3), np.uint32))[0].shape[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("bounces", [1, 2])
def test_hip_gi_rays_with_bounces(hip_ctx, bounces):
    """sah_rt_set_bounces: the hit stage's bounce branch (gltf_basic_pbr.slang:481-517), bit for bit"""
    with _bounces(bounces, hip_ctx):
        for kind in ("none", "down", "up", "cutout"):
            case = RtCase(_floor_under(kind), 32, 18, view=_top_down_view(32, 18))
            for i in range(3):
                case.sun.constants.color[i] = 1.0
            _check_gi(hip_ctx, case, _probe_ids(3, 12))
        case = RtCase(mesh.atrium(1), 96, 54)
        _, _, ri_b = _check_gi(hip_ctx, case, _probe_ids(5, 40))
        for seed, textured in ((1, False), (2, True)):
            soup = RtCase(mesh.random_soup(seed, triangles=400, textured=textured), 64, 36, seed=seed)
            soup.sun.set_direction([0.3, -1.0, 0.2])
            _check_gi(hip_ctx, soup, _probe_ids(seed, 30))
    # ... and the setting is the context's: back at 0 the atrium's rays carry less light than they just did
    _, _, ri_0 = _check_gi(hip_ctx, case, _probe_ids(5, 8))
    a, b = ri_0.astype(np.float32)[..., :3], ri_b.astype(np.float32)[..., :3]
    assert (b >= a).all() and (b > a).any()


@pytest.mark.gpu
def test_hip_rejects_more_bounces_than_it_instantiates(hip_ctx):
    from androidrenderer_amd.lib import SahError
    with pytest.raises(SahError):
        hip_ctx.rt_set_bounces(3)
    hip_ctx.rt_set_bounces(0)


@pytest.mark.gpu
def test_hip_probe_trace_feeds_probe_update(hip_ctx):
    """the frame's order (irradiance_cache.cpp dispatch_probe_updates): trace the probes, then fold the 20 x 20 results into the atlases"""
    import torch
    from androidrenderer_amd.frame import to_torch
    case = RtCase(mesh.atrium(1), 32, 18)
    atl, _, ids = synth.probe_maintenance_inputs(seed=31, num_probes=24)
    trace = case.oracle_probe_trace(ids)
    o_atl = {k: v.copy() for k, v in atl.items()}
    tv = images.volume(trace.view(np.uint16), _abi.FORMAT_R16G16B16A16_SFLOAT)
    assert util.oracle().orc_probe_update(C.byref(util.probe_atlases_desc(o_atl)), C.byref(tv), ids.ctypes.data, len(ids)) == 0
    case.hip_build(hip_ctx)
    dv = case.device()
    pd = torch.from_numpy(ids.view(np.int32).reshape(-1)).cuda()
    out = torch.zeros((len(ids), 20, 20, 4), dtype=torch.int16, device="cuda")
    d, keep = case.probe_desc(dv["gi"], pd.data_ptr(), len(ids), dv["noise"], out)
    h_atl = {k: to_torch(v) for k, v in atl.items()}
    hip_ctx.probe_trace(d)
    hip_ctx.probe_update(util.probe_atlases_desc(h_atl), images.volume(out, _abi.FORMAT_R16G16B16A16_SFLOAT), pd.data_ptr(), len(ids))
    torch.cuda.synchronize()
    for k in o_atl:
        assert np.array_equal(h_atl[k].cpu().numpy().view(o_atl[k].dtype), o_atl[k]), k


@pytest.mark.gpu
def test_hip_gi_generators_reject_bad_arguments(hip_ctx):
    from androidrenderer_amd import lib
    import torch
    case = RtCase(_floor_and_plate(), 16, 9, view=_top_down_view(16, 9))
    case.hip_build(hip_ctx)
    dv = case.device()
    small_noise = torch.zeros((64, 64, 4), dtype=torch.uint8, device="cuda")
    out = torch.zeros((1, 20, 20, 4), dtype=torch.int16, device="cuda")
    pd = torch.zeros(3, dtype=torch.int32, device="cuda")
    d, keep = case.probe_desc(dv["gi"], pd.data_ptr(), 1, small_noise, out)
    with pytest.raises(lib.SahError):
        hip_ctx.probe_trace(d)
    d, keep = case.probe_desc(dv["gi"], pd.data_ptr(), 2, dv["noise"], out)  # results hold one probe
    with pytest.raises(lib.SahError):
        hip_ctx.probe_trace(d)
    d, keep = case.probe_desc(dv["gi"], 0, 1, dv["noise"], out)
    with pytest.raises(lib.SahError):
        hip_ctx.probe_trace(d)


@pytest.mark.gpu
def test_hip_row_windows_trace_the_same_pixels(hip_ctx):
    """sah_rt_set_rows: the per-pixel generators write the rows of the window only, and a pixel's result does not depend on the window —
    three windows (cut inside 16-row tiles) assemble the full-frame planes; (0, 0) restores the whole frame"""
    import torch
    case = RtCase(mesh.atrium(1), 80, 45)
    case.sun.constants.num_shadow_samples = 3.0
    case.hip_build(hip_ctx)
    try:
        full = (case.hip_rtao(hip_ctx, 1, 4.0), case.hip_mask(hip_ctx)) + case.hip_rtgi(hip_ctx)
        parts = [np.full_like(f, 0) for f in full]
        for r0, r1 in ((0, 17), (17, 40), (40, 45)):
            hip_ctx.rt_set_rows(r0, r1)
            got = (case.hip_rtao(hip_ctx, 1, 4.0), case.hip_mask(hip_ctx)) + case.hip_rtgi(hip_ctx)
            for k, g in enumerate(got):
                fill = -7.0 if k < 2 else np.uint16(0x7bff).view(np.float16)  # the test planes' fill: rows outside the window keep it
                outside = np.ones(45, bool)
                outside[r0:r1] = False
                assert (g[outside].view(np.uint32 if k < 2 else np.uint16) == np.array(fill, g.dtype).view(np.uint32 if k < 2 else np.uint16)).all(), (k, r0, r1)
                parts[k][r0:r1] = g[r0:r1]
        for k in range(4):
            # (RTGI planes: texels the generator skips keep the fill in `full` as well)
            a, b = parts[k].view(np.uint32 if k < 2 else np.uint16), full[k].view(np.uint32 if k < 2 else np.uint16)
            assert np.array_equal(a, b), f"plane {k}: {int((a != b).sum())} values differ"
        hip_ctx.rt_set_rows(30, 4000)  # clipped to the plane
        assert np.array_equal(case.hip_rtao(hip_ctx, 1, 4.0)[30:], full[0][30:])
        with pytest.raises(Exception):
            hip_ctx.rt_set_rows(5, 4)
    finally:
        hip_ctx.rt_set_rows(0, 0)
        torch.cuda.synchronize()
    assert np.array_equal(case.hip_rtao(hip_ctx, 1, 4.0), full[0])
