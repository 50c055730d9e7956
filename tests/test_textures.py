"""Material textures of the scene rasteriser (gltf_basic_pbr.slang:177-226, `textures[i].SampleBias(texcoord, mip_bias)`).
CPU: known answers for the sampling rules include/sah_hip.h fixes (level selection from quad derivatives, bias and clamps, filters,
address modes, sRGB decode) on the oracle.  GPU: HIP against the oracle, bit for bit, on textured triangle soups (every sampler
combination), the textured atrium, and alpha-tested geometry whose alpha comes from a texture, in all three passes."""
import ctypes as C
import math

import numpy as np
import pytest

from androidrenderer_amd import _abi, images, mesh, scene, synth
from tests import util
from tests.test_raster import _hip_gbuffer, _hip_shadow, _oracle_gbuffer, _oracle_shadow, _ortho_sun, _quad


def _srgb_decode(byte):
    c = byte / 255.0
    return np.float32(c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4)


def _wall(m, mat, scale=(1.0, 1.0), offset=(0.0, 0.0)):
    """A wall 4 m ahead of the default camera, perpendicular to the view axis (texcoords are then affine in window space), with
    texcoords = (z, y) * scale + offset."""
    pos = np.array([(-3, -3, -4), (-3, -3, 4), (-3, 5, 4), (-3, 5, -4)], np.float32)
    uv = np.stack([pos[:, 2] * scale[0] + offset[0], pos[:, 1] * scale[1] + offset[1]], axis=-1)
    return m.add_primitive(pos, [(-1, 0, 0)] * 4, (0, 1, 2, 0, 2, 3), mat, ptype=_abi.PRIMITIVE_TYPE_CUTOUT, texcoords=uv)


def _level_texture(size=64, srgb=False):
    """level i is the constant byte 20 * (i + 1) in every channel: the colour of a pixel names the level it was sampled from"""
    return [np.full((h, w, 4), 20 * (i + 1), np.uint8) for i, (w, h) in enumerate(mesh.mip_sizes(size, size))], srgb


def _texels_per_pixel(view, w):
    """window-space footprint of the wall's z coordinate: pixels per metre at 4 m = w / (2 * 4 * tan(fov_x / 2)) -> metres per pixel"""
    proj = np.array(view.gpu_data.projection[:], np.float32).reshape(4, 4).T  # column-major storage
    return 2.0 * 4.0 / (proj[0, 0] * w)


@pytest.mark.parametrize("bias,min_lod,max_lod,mipmap", [(0.0, 0.0, 1000.0, 0), (1.0, 0.0, 1000.0, 0), (-1.0, 0.0, 1000.0, 0), (0.0, 3.0, 1000.0, 0),
                                                          (0.0, 0.0, 1.0, 0), (0.0, 0.0, 1000.0, 1), (0.5, 0.0, 1000.0, 1)])
def test_level_selection_follows_the_quad_derivatives(bias, min_lod, max_lod, mipmap):
    w, h = 64, 36
    view = scene.SceneView.default(w, h)
    mpp = _texels_per_pixel(view, w)                    # metres per pixel on the wall
    want_lambda = 2.2                                   # texels of level 0 per pixel = 2^2.2
    s = (2.0 ** want_lambda) / (64.0 * mpp)             # texcoord units per metre
    m = mesh.Mesh()
    mips, srgb = _level_texture(64)
    t = m.add_texture(mips, srgb, mesh.sampler(mag=0, min=0, mipmap=mipmap, bias=bias, min_lod=min_lod, max_lod=max_lod))
    mat = m.add_material(mesh.material(), data=t)
    _wall(m, mat, scale=(s, s))
    out, _ = _oracle_gbuffer(m.arrays(), view, w, h)
    hit = out["depth"] > 0
    assert hit.sum() > w * h // 2
    lam = min(max(want_lambda + bias, min_lod), max_lod)
    got = out["data"][hit][:, 1].astype(np.float64)     # data.g = texel.g * roughness_factor (0.5), as UNORM8
    if mipmap == 0:
        level = 0 if lam <= 0.5 else math.ceil(lam + 0.5) - 1
        want = round(20 * (level + 1) / 255.0 * 0.5 * 255.0)
        assert np.all(np.abs(got - want) <= 1), (np.unique(got), want)
    else:
        hi = math.floor(lam)
        d = lam - hi
        want = ((1 - d) * 20 * (hi + 1) + d * 20 * (hi + 2)) * 0.5
        assert np.all(np.abs(got - want) <= 1.5), (np.unique(got), want)
    # the view's material_texture_mip_bias is the shader-side bias of the G-buffer pass (gltf_basic_pbr.slang:175-178)
    view.gpu_data.material_texture_mip_bias = 1.0
    out2, _ = _oracle_gbuffer(m.arrays(), view, w, h)
    lam2 = min(max(want_lambda + bias + 1.0, min_lod), max_lod)
    if mipmap == 0 and (0 if lam2 <= 0.5 else math.ceil(lam2 + 0.5) - 1) != (0 if lam <= 0.5 else math.ceil(lam + 0.5) - 1):
        assert not np.array_equal(out2["data"], out["data"])


@pytest.mark.parametrize("aniso,lam_major,lam_minor,want_level", [(0.0, 3.2, 1.2, 3), (8.0, 3.2, 1.2, 1), (2.0, 3.2, 1.2, 2), (8.0, 5.2, 0.2, 2),
                                                                  (16.0, 5.2, 1.7, 2), (8.0, 2.2, 2.2, 2)])
def test_anisotropic_footprint_takes_its_level_from_the_minor_axis(aniso, lam_major, lam_minor, want_level):
    """sah_hip.h "anisotropy": eta = min(rho_max / rho_min, A), lambda = log2(rho_max) - log2(eta) — up to A times more detail than the
    isotropic footprint; on a texture whose level i is the constant 20 (i + 1) the N taps agree, so the colour names the level"""
    w, h = 64, 36
    view = scene.SceneView.default(w, h)
    mpp = _texels_per_pixel(view, w)
    su, sv = (2.0 ** lam_major) / (64.0 * mpp), (2.0 ** lam_minor) / (64.0 * mpp)
    m = mesh.Mesh()
    mips, srgb = _level_texture(64)
    t = m.add_texture(mips, srgb, mesh.sampler(mag=0, min=0, mipmap=0, max_anisotropy=aniso))
    mat = m.add_material(mesh.material(), data=t)
    _wall(m, mat, scale=(su, sv))
    out, _ = _oracle_gbuffer(m.arrays(), view, w, h)
    hit = out["depth"] > 0
    assert hit.sum() > w * h // 2
    got = out["data"][hit][:, 1]
    eta = min(2.0 ** (lam_major - lam_minor), aniso) if aniso > 1 else 1.0
    lam = lam_major - math.log2(eta)
    level = 0 if lam <= 0.5 else math.ceil(lam + 0.5) - 1
    assert level == want_level
    want = round(20 * (level + 1) / 255.0 * 0.5 * 255.0)
    assert (np.abs(got.astype(np.int64) - want) <= 0).mean() > 0.97, (np.unique(got), want)  # (a few pixels at the wall's silhouette extrapolate)


def test_anisotropic_taps_average_along_the_major_axis():
    """one level, NEAREST, steps in u every 8 texels (0 | 200 | 0 ...); a footprint of 3.9 texels per pixel along u and 1 along v
    makes N = 4 taps spread over 2.3 texels around the sample point: pixels next to the step average k of 4 taps on the bright side
    (0, 50, 100, 150, 200), where the isotropic sample knows 0 and 200 only"""
    w, h = 64, 36
    view = scene.SceneView.default(w, h)
    mpp = _texels_per_pixel(view, w)
    tex = np.zeros((64, 64, 4), np.uint8)
    tex[:, (np.arange(64) // 8) % 2 == 1] = 200
    seen = {}
    for aniso in (0.0, 8.0):
        m = mesh.Mesh()
        t = m.add_texture([tex], False, mesh.sampler(mag=0, min=0, mipmap=0, max_anisotropy=aniso))
        mat = m.add_material(mesh.material(rough=1.0), data=t)
        _wall(m, mat, scale=(3.9 / (64.0 * mpp), 1.0 / (64.0 * mpp)), offset=(0.5, 0.0))
        out, _ = _oracle_gbuffer(m.arrays(), view, w, h)
        seen[aniso] = set(int(v) for v in np.unique(out["data"][out["depth"] > 0][:, 1]))
    assert seen[0.0] == {0, 200}
    assert seen[8.0] <= {0, 50, 100, 150, 200} and len(seen[8.0] & {50, 100, 150}) >= 2, seen[8.0]


@pytest.mark.parametrize("mode", [_abi.ADDRESS_REPEAT, _abi.ADDRESS_MIRRORED_REPEAT, _abi.ADDRESS_CLAMP_TO_EDGE])
def test_address_modes_with_the_nearest_filter(mode):
    """a 4 x 1 texture with texels 10, 20, 30, 40 over texcoords [-1, 2]: the sequence of texel values along a row names the mode"""
    w, h = 192, 108
    view = scene.SceneView.default(w, h)
    m = mesh.Mesh()
    tex = np.zeros((1, 4, 4), np.uint8)
    tex[0, :, :] = np.array([10, 20, 30, 40])[:, None]
    t = m.add_texture([tex], False, mesh.sampler(mag=0, min=0, mipmap=0, address_u=mode, address_v=mode))
    mat = m.add_material(mesh.material(rough=1.0), data=t)
    _wall(m, mat, scale=(3.0 / 8.0, 0.0), offset=(0.5, 0.5))  # the wall spans z in [-4, 4]: u from -1 to 2
    out, _ = _oracle_gbuffer(m.arrays(), view, w, h)
    hit = out["depth"][h // 2] > 0
    assert hit.sum() > w // 2
    row = out["data"][h // 2, hit, 1].astype(int)
    runs = [int(v) for i, v in enumerate(row) if i == 0 or v != row[i - 1]]
    if runs[0] > runs[-1]:
        runs = runs[::-1]                                      # the wall's z axis may run right to left on screen
    want = {_abi.ADDRESS_REPEAT: [10, 20, 30, 40] * 3, _abi.ADDRESS_MIRRORED_REPEAT: [40, 30, 20, 10, 10, 20, 30, 40, 40, 30, 20, 10],
            _abi.ADDRESS_CLAMP_TO_EDGE: [10, 20, 30, 40]}[mode]
    if mode == _abi.ADDRESS_MIRRORED_REPEAT:
        merged = [want[0]]
        for v in want[1:]:
            if v != merged[-1]:
                merged.append(v)
        want = merged
        if runs[0] != want[0]:
            runs = runs[::-1]
    assert runs == want, (runs, want)


def test_one_by_one_textures_equal_constant_texels():
    """A 1x1 texture is the constant texel sah_material holds for the slot: UNORM as byte / 255, sRGB through the decode."""
    view = scene.SceneView.default(120, 68)
    bytes_ = {"base": (200, 90, 30, 255), "normal": (140, 100, 250, 255), "data": (0, 180, 77, 0), "emission": (255, 128, 3, 0)}

    def build(with_textures):
        src = mesh.random_soup(31, triangles=200)
        m = mesh.Mesh()
        m.positions, m.vertex_data, m.indices, m.primitives = src.positions, src.vertex_data, src.indices, src.primitives
        slots = {}
        if with_textures:
            for name, srgb in (("base", True), ("normal", False), ("data", False), ("emission", True)):
                slots[name] = m.add_texture([np.array(bytes_[name], np.uint8).reshape(1, 1, 4)], srgb, mesh.random_sampler(synth.rng(5)))
        for k in range(4):
            mat = src.materials[k].copy()
            for name, field, srgb in (("base", "base_color_texel", True), ("normal", "normal_texel", False), ("data", "data_texel", False),
                                      ("emission", "emission_texel", True)):
                b = bytes_[name]
                mat[field] = [(_srgb_decode(b[c]) if srgb and c < 3 else np.float32(b[c]) / np.float32(255.0)) for c in range(4)]
                if with_textures:
                    mat[field] = (7.0, 7.0, 7.0, 7.0)  # must not be read
            if with_textures:
                m.add_material(mat, base_color=slots["base"], normal=slots["normal"], data=slots["data"], emission=slots["emission"])
            else:
                m.add_material(mat)
        return m
    a, _ = _oracle_gbuffer(build(False).arrays(), view, 120, 68)
    b, _ = _oracle_gbuffer(build(True).arrays(), view, 120, 68)
    assert (a["depth"] > 0).sum() > 500
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_linear_filter_interpolates_between_texel_centres():
    """2 x 1 texture (0, 255) with CLAMP and the linear filter at magnification: a ramp between the two texel centres"""
    w, h = 256, 144
    view = scene.SceneView.default(w, h)
    m = mesh.Mesh()
    tex = np.zeros((1, 2, 4), np.uint8)
    tex[0, 1] = 255
    t = m.add_texture([tex], False, mesh.sampler(address_u=_abi.ADDRESS_CLAMP_TO_EDGE, address_v=_abi.ADDRESS_CLAMP_TO_EDGE))
    mat = m.add_material(mesh.material(rough=1.0), data=t)
    _wall(m, mat, scale=(1.0 / 8.0, 0.0), offset=(0.5, 0.5))  # u from 0 to 1 over the wall
    out, _ = _oracle_gbuffer(m.arrays(), view, w, h)
    hit = out["depth"][h // 2] > 0
    row = out["data"][h // 2, hit, 1].astype(int)
    n = len(row)
    assert n > w // 2
    if row[0] > row[-1]:
        row = row[::-1]
    assert row[0] == 0 and row[-1] == 255 and np.all(np.diff(row) >= 0)
    mid = row[n // 4 + 2: 3 * n // 4 - 2]
    assert np.all(np.abs(np.diff(mid) - 255.0 * 2 / n) <= 1.01)   # slope: the two texel centres are n / 2 pixels apart


def test_bad_texture_tables_are_refused():
    m = mesh.random_soup(3, triangles=40, textured=True)
    arrays = m.arrays()
    arrays["material_textures"] = arrays["material_textures"].copy()
    arrays["material_textures"][1, 2] = 99  # beyond the table
    view = scene.SceneView.default(32, 18)
    out = {"color": np.zeros((18, 32, 4), np.uint8), "normals": np.zeros((18, 32, 4), np.uint16), "data": np.zeros((18, 32, 4), np.uint8),
           "emission": np.zeros((18, 32, 4), np.uint8), "depth": np.zeros((18, 32), np.float32)}
    keep = []
    g = mesh.geometry(mesh.with_counts(arrays), keep)
    gb = images.gbuffer(out)
    assert util.oracle().orc_gbuffer_render(C.byref(g), C.byref(view.gpu_data), C.byref(gb), None) == _abi.SAH_ERR_INVALID_ARGUMENT


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------------

def _same(got, want, what):
    for k in want:
        diff = int((got[k] != want[k]).sum())
        assert diff == 0, f"{what}: plane {k}: {diff} values differ"


@pytest.mark.gpu
@pytest.mark.parametrize("seed,size,bias", [(41, (200, 120), 0.0), (42, (333, 187), -0.5), (43, (64, 36), 1.5), (44, (640, 360), 0.0)])
def test_hip_textured_gbuffer_matches_oracle(hip_ctx, seed, size, bias):
    w, h = size
    view = scene.SceneView.default(w, h)
    view.gpu_data.material_texture_mip_bias = bias
    arrays = mesh.random_soup(seed, triangles=600, textured=True).arrays()
    got, got_stats = _hip_gbuffer(hip_ctx, arrays, view, w, h)
    want, want_stats = _oracle_gbuffer(arrays, view, w, h)
    assert (want["depth"] > 0).mean() > 0.3
    _same(got, want, f"soup {seed}")
    assert np.array_equal(got_stats[:4], want_stats[:4])


@pytest.mark.gpu
def test_hip_textured_atrium_matches_oracle(hip_ctx):
    w, h = 480, 270
    view = scene.SceneView.default(w, h)
    g = synth.rng(77)
    m = mesh.atrium(subdiv=2)
    tex = [m.add_texture(*mesh.random_texture(g, 128, 128, None, srgb, mesh.sampler())) for srgb in (True, False, False, True)]
    m.material_textures = [(tex[0], tex[1], tex[2], tex[3] if i == 5 else _abi.TEXTURE_NONE) for i in range(len(m.materials))]
    arrays = m.arrays()
    got, _ = _hip_gbuffer(hip_ctx, arrays, view, w, h)
    want, _ = _oracle_gbuffer(arrays, view, w, h)
    _same(got, want, "atrium")
    flat, _ = _oracle_gbuffer(mesh.atrium(subdiv=2).arrays(), view, w, h)
    assert not np.array_equal(flat["color"], want["color"])  # the textures did something


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [51, 52])
def test_hip_textured_alpha_test_in_the_shadow_and_rsm_passes(hip_ctx, seed):
    """masked geometry whose alpha comes from the base-colour texture (mip bias 0 in these passes): shadow cascades and RSM"""
    from tests.test_lpv_inject import _hip_rsm, _oracle_rsm, _setup
    arrays = mesh.random_soup(seed, triangles=400, cutout_fraction=1.0, textured=True).arrays()
    sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_CSM)
    view = scene.SceneView.default(320, 180)
    sun.update_shadow_cascades(view, resolution=256)
    got, _ = _hip_shadow(hip_ctx, arrays, sun.constants, 4, (256, 256))
    want, _ = _oracle_shadow(arrays, sun.constants, 4, (256, 256))
    assert np.array_equal(got, want), int((got != want).sum())
    assert (want != 0xffff).mean() > 0.02
    # the same scene with every fragment kept (alpha threshold below every texel) casts a different shadow: the texture alpha matters
    _, sun2, lpv = _setup()
    want_rsm = _oracle_rsm(arrays, sun2, lpv)
    got_rsm = _hip_rsm(hip_ctx, arrays, sun2, lpv)
    for k in ("depth", "flux", "normals"):
        g = got_rsm[k].cpu().numpy()
        g = g.view(np.uint16) if k == "depth" else g
        assert np.array_equal(g, want_rsm[k]), f"rsm {k}: {int((g != want_rsm[k]).sum())} values differ"


@pytest.mark.gpu
def test_hip_refuses_bad_texture_tables_without_touching_them(hip_ctx):
    import torch
    arrays = mesh.random_soup(3, triangles=40, textured=True).arrays()
    view = scene.SceneView.default(32, 18)
    table_bytes = C.sizeof(_abi.Texture) * len(arrays["textures"])
    for what in ("binding", "levels", "format", "pointer"):
        dev = mesh.to_device(arrays)
        if what == "binding":
            bad = arrays["material_textures"].copy()
            bad[1, 2] = 99
            dev["material_textures"] = torch.from_numpy(np.frombuffer(bad.tobytes(), dtype=np.uint8).copy()).cuda()
        g = mesh.geometry(dev, [])
        if what != "binding":  # patch texture 0 of the device-side table
            table = next(t for t in g._alive if hasattr(t, "data_ptr") and t.dim() == 1 and t.dtype == torch.uint8 and t.numel() == table_bytes)
            host = bytearray(table.cpu().numpy().tobytes())
            t0 = _abi.Texture.from_buffer(host)
            if what == "levels":
                t0.num_mips = 15
            elif what == "format":
                t0.mips[0].format = _abi.FORMAT_R16G16B16A16_SFLOAT
            else:
                t0.mips[1].ptr = 0
            del t0
            table.copy_(torch.frombuffer(host, dtype=torch.uint8))
        out = {"color": torch.zeros((18, 32, 4), dtype=torch.uint8, device="cuda"), "normals": torch.zeros((18, 32, 4), dtype=torch.int16, device="cuda"),
               "data": torch.zeros((18, 32, 4), dtype=torch.uint8, device="cuda"), "emission": torch.zeros((18, 32, 4), dtype=torch.uint8, device="cuda"),
               "depth": torch.zeros((18, 32), dtype=torch.float32, device="cuda")}
        with pytest.raises(Exception) as err:
            hip_ctx.gbuffer_render(g, view.gpu_data, images.gbuffer(out), None)
        assert "texture" in str(err.value), (what, str(err.value))
    torch.cuda.synchronize()
