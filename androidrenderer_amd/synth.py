"""Synthetic inputs for the hot path. Everything is generated with numpy's counter-based Philox
generator from a fixed seed, on the host, in the exact storage formats the reference allocates
(RenderCore/render/scene_renderer.cpp:580-649).

Two G-buffer flavours:
  random_gbuffer   independent random texels (BASELINE config "synthetic random G-buffer")
  atrium_gbuffer   a procedural Sponza-like atrium (floor, walls, two rows of box columns, open roof) ray-cast
                   analytically; the Sponza asset itself is absent from the reference tree (.MISSING_LARGE_BLOBS), so
                   the "Sponza" configs run on this stand-in.  Reports must say which flavour was used.
"""
import math

import numpy as np

SEED_BASE = 0x5A48


def rng(seed):
    return np.random.Generator(np.random.Philox(SEED_BASE + seed))


def _srgb_encode_u8(lin):
    lin = np.clip(lin, 0.0, 1.0)
    s = np.where(lin <= 0.0031308, lin * 12.92, 1.055 * np.power(lin, 1.0 / 2.4) - 0.055)
    return np.clip(np.rint(s * 255.0), 0, 255).astype(np.uint8)


def random_gbuffer(width, height, seed=1, sky_fraction=0.10, z_near=0.05):
    """10 % sky, depth = z_near/d with d log-uniform in [0.5,100] m, normals uniform on the sphere times
    U[0.5,1.5] (fp16), uniform colour/data/emission bytes with data.r = data.a = 0, emission non-zero on 1 % of pixels."""
    g = rng(seed)
    n = width * height
    d = np.exp(g.uniform(math.log(0.5), math.log(100.0), n)).astype(np.float32)
    depth = (np.float32(z_near) / d).astype(np.float32)
    depth[g.random(n) < sky_fraction] = 0.0
    v = g.standard_normal((n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v *= g.uniform(0.5, 1.5, (n, 1)).astype(np.float32)
    normals = np.zeros((n, 4), dtype=np.float16)
    normals[:, :3] = v.astype(np.float16)
    color = g.integers(0, 256, (n, 4), dtype=np.uint8)
    data = g.integers(0, 256, (n, 4), dtype=np.uint8)
    data[:, 0] = 0
    data[:, 3] = 0
    emission = g.integers(0, 256, (n, 4), dtype=np.uint8)
    emission[g.random(n) >= 0.01] = 0
    return {
        "color": color.reshape(height, width, 4),
        "normals": normals.reshape(height, width, 4),
        "data": data.reshape(height, width, 4),
        "emission": emission.reshape(height, width, 4),
        "depth": depth.reshape(height, width),
    }


_ATRIUM_BOXES = None


def _atrium_boxes():
    """(min, max, material) of every box in the atrium: floor y=0, long walls z=+-4.5, end walls x=+-14, open roof above
    y=11 (sky), gallery slabs at y=5, two rows of 0.7 m columns at z=+-2.6 every 3.5 m, small emissive lamp boxes."""
    global _ATRIUM_BOXES
    if _ATRIUM_BOXES is None:
        boxes = [((-14.5, -1.0, -5.0), (14.5, 0.0, 5.0), 0), ((-14.5, 0.0, 4.5), (14.5, 11.0, 5.5), 1),
                 ((-14.5, 0.0, -5.5), (14.5, 11.0, -4.5), 1), ((14.0, 0.0, -5.0), (15.0, 11.0, 5.0), 2),
                 ((-15.0, 0.0, -5.0), (-14.0, 11.0, 5.0), 2), ((-14.0, 5.0, 2.95), (14.0, 5.3, 4.5), 3),
                 ((-14.0, 5.0, -4.5), (14.0, 5.3, -2.95), 3)]
        for i in range(8):
            cx = -12.25 + 3.5 * i
            for cz in (-2.6, 2.6):
                boxes.append(((cx - 0.35, 0.0, cz - 0.35), (cx + 0.35, 5.0, cz + 0.35), 4))
            boxes.append(((cx - 0.15, 3.2, -4.5), (cx + 0.15, 3.5, -4.3), 5))
        _ATRIUM_BOXES = boxes
    return _ATRIUM_BOXES


def atrium_gbuffer(width, height, view, seed=2, device="cpu"):
    """Procedural Sponza-like atrium seen from `view` (a scene.SceneView), ray-cast analytically with torch on `device`
    (the arithmetic is a few hundred ops per pixel, so 4K takes well under a second on the GPU). Materials vary per
    surface with low-amplitude per-pixel noise (numpy Philox, so the noise is device independent); the lamp boxes are
    emissive. Returns numpy arrays in the reference's storage formats."""
    import torch
    g = rng(seed)
    H, W = height, width
    dev = torch.device(device)
    f32 = torch.float32
    ys = (torch.arange(H, dtype=f32, device=dev) + 0.5) / H * 2.0 - 1.0
    xs = (torch.arange(W, dtype=f32, device=dev) + 0.5) / W * 2.0 - 1.0
    proj = np.array(view.gpu_data.projection[:], dtype=np.float32).reshape(4, 4)  # [col,row]
    inv_view = np.array(view.gpu_data.inverse_view[:], dtype=np.float32).reshape(4, 4)
    dvx = (xs / float(proj[0, 0])).view(1, W).expand(H, W)
    dvy = (ys / float(proj[1, 1])).view(H, 1).expand(H, W)
    rot = torch.tensor(inv_view[:3, :3], device=dev)  # rows = rotation columns
    d = dvx.unsqueeze(-1) * rot[0] + dvy.unsqueeze(-1) * rot[1] - rot[2]  # view-space ray (x, y, -1): z_view = -t
    o = torch.tensor(inv_view[3, :3], device=dev)
    inv = 1.0 / d
    t_best = torch.full((H, W), float("inf"), dtype=f32, device=dev)
    axis_best = torch.zeros((H, W), dtype=torch.int64, device=dev)
    m_best = torch.full((H, W), -1, dtype=torch.int64, device=dev)
    for bmin, bmax, mat in _atrium_boxes():
        t0 = (torch.tensor(bmin, device=dev) - o) * inv
        t1 = (torch.tensor(bmax, device=dev) - o) * inv
        tn = torch.minimum(t0, t1)
        tf = torch.maximum(t0, t1)
        tnear, axis = tn.max(dim=-1)
        tfar = tf.min(dim=-1).values
        closer = (tnear <= tfar) & (tnear > 1e-3) & (tnear < t_best)
        t_best = torch.where(closer, tnear, t_best)
        axis_best = torch.where(closer, axis, axis_best)
        m_best = torch.where(closer, torch.full_like(m_best, mat), m_best)
    hit = torch.isfinite(t_best)
    t_hit = torch.where(hit, t_best, torch.ones_like(t_best))
    depth = torch.where(hit, float(view.near_value) / t_hit, torch.zeros_like(t_best))
    nrm = torch.zeros((H, W, 3), dtype=f32, device=dev)
    nrm.scatter_(-1, axis_best.unsqueeze(-1), -torch.sign(torch.gather(d, -1, axis_best.unsqueeze(-1))))
    pos = o + d * torch.where(hit, t_best, torch.zeros_like(t_best)).unsqueeze(-1)
    base_lin = torch.tensor([[0.45, 0.40, 0.33], [0.60, 0.52, 0.42], [0.50, 0.30, 0.22], [0.55, 0.55, 0.50], [0.62, 0.58, 0.50],
                             [0.9, 0.8, 0.6]], dtype=f32, device=dev)
    rough = torch.tensor([0.65, 0.8, 0.7, 0.5, 0.4, 0.3], dtype=f32, device=dev)
    metal = torch.tensor([0.0, 0.0, 0.0, 0.1, 0.0, 0.9], dtype=f32, device=dev)
    mi = m_best.clamp(0, 5)
    checker = torch.floor(pos * 2.0).sum(dim=-1).remainder(2.0)
    noise = torch.from_numpy(g.uniform(-1.0, 1.0, (H, W, 7)).astype(np.float32)).to(dev)
    col = base_lin[mi] * (0.8 + 0.2 * checker).unsqueeze(-1) + 0.02 * noise[..., 0:3]
    col = col.clamp(0.0, 1.0)
    srgb = torch.where(col <= 0.0031308, col * 12.92, 1.055 * col.clamp_min(1e-12).pow(1.0 / 2.4) - 0.055)
    color = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    color[..., :3] = torch.round(srgb * 255.0).clamp(0, 255).to(torch.uint8)
    color[..., 3] = 255
    data = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    data[..., 1] = torch.round((rough[mi] + 0.03 * noise[..., 3]) * 255.0).clamp(1, 255).to(torch.uint8)
    data[..., 2] = torch.round(metal[mi] * 255.0).clamp(0, 255).to(torch.uint8)
    normals = torch.zeros((H, W, 4), dtype=torch.float16, device=dev)
    normals[..., :3] = (nrm + 0.04 * noise[..., 4:7]).to(torch.float16)
    emission = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    lamp = m_best == 5
    emission[lamp] = torch.tensor([255, 214, 170, 0], dtype=torch.uint8, device=dev)
    miss = ~hit
    for a in (color, data, normals, emission):
        a[miss] = 0
    return {"color": color.cpu().numpy(), "normals": normals.cpu().numpy(), "data": data.cpu().numpy(),
            "emission": emission.cpu().numpy(), "depth": depth.cpu().numpy()}


def ao_plane(width, height, seed=3):
    return rng(seed).random((height, width), dtype=np.float32)


def shadow_mask(width, height, seed=4, samples=8):
    """RT-mode sun visibility: k/samples, as `shadow / num_shadow_samples` produces
    (RenderCore/shaders/lighting/directional_light.rt.slang:93-125)."""
    k = rng(seed).integers(0, samples + 1, (height, width))
    return (k.astype(np.float32) / np.float32(samples)).astype(np.float32)


def lpv_volumes(num_cascades=4, seed=5, zero_fraction=0.7):
    """Three RGBA16F volumes (32*n) x 32 x 32, fp16 U[0,2] with ~70 % zero cells."""
    g = rng(seed)
    vols = []
    for _ in range(3):
        v = g.uniform(0.0, 2.0, (32, 32, 32 * num_cascades, 4)).astype(np.float16)
        v[g.random((32, 32, 32 * num_cascades)) < zero_fraction] = 0
        # signed directional bands so the SH dot can cancel, as real propagated light does
        v[..., 1:] *= np.where(g.random((32, 32, 32 * num_cascades, 3)) < 0.5, -0.5, 0.5).astype(np.float16)
        vols.append(v)
    return vols  # indexed [z][y][x][c]


def shadowmap(resolution=1024, cascades=4, seed=6):
    """D16_UNORM array, values in [0.3, 0.7] (reduced from 4096^2 to keep fixtures bounded)."""
    g = rng(seed)
    return g.integers(int(0.3 * 65535), int(0.7 * 65535), (cascades, resolution, resolution), dtype=np.uint16)


def atrium_shadowmap(sun_constants, resolution=4096, cascades=4, device="cpu"):
    """Cascaded shadow map of the procedural atrium (SURVEY §8-f2 stand-in for the CSM depth pass, directional_light.cpp:286-327):
    every D16 texel holds the shadow-space depth of the nearest atrium box along the sun direction, found by intersecting the
    texel's ray — the segment between the cascade's z = 0 and z = 1 planes through cascade_inverse_matrices — with the boxes
    (torch on `device`).  No occluder: 1.0.  The same boxes produce the G-buffer (atrium_gbuffer), so lit / shadowed regions are
    coherent with the geometry, which random depth noise (shadowmap()) is not."""
    import torch
    dev = torch.device(device)
    f32 = torch.float32
    out = np.zeros((cascades, resolution, resolution), dtype=np.uint16)
    c = (torch.arange(resolution, dtype=f32, device=dev) + 0.5) / resolution * 2.0 - 1.0
    boxes = _atrium_boxes()
    rows = max(1, min(resolution, (1 << 22) // resolution))  # ~4M rays per chunk
    for ci in range(cascades):
        inv = torch.tensor(np.array(sun_constants.cascade_inverse_matrices[ci][:], dtype=np.float32).reshape(4, 4), device=dev)  # [col][row]

        def unproject(x, y, z):
            p = x.unsqueeze(-1) * inv[0] + y.unsqueeze(-1) * inv[1] + z * inv[2] + inv[3]
            return p[..., :3] / p[..., 3:4]

        for r0 in range(0, resolution, rows):
            yy, xx = torch.meshgrid(c[r0:r0 + rows], c, indexing="ij")
            p0, p1 = unproject(xx, yy, 0.0), unproject(xx, yy, 1.0)
            d = p1 - p0
            d = torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
            inv_d = 1.0 / d
            t_best = torch.ones(xx.shape, dtype=f32, device=dev)
            for bmin, bmax, _ in boxes:
                t0 = (torch.tensor(bmin, device=dev) - p0) * inv_d
                t1 = (torch.tensor(bmax, device=dev) - p0) * inv_d
                tnear = torch.minimum(t0, t1).max(dim=-1).values
                tfar = torch.maximum(t0, t1).min(dim=-1).values
                hit = (tnear <= tfar) & (tfar > 0.0)
                t_best = torch.where(hit, torch.minimum(t_best, tnear.clamp_min(0.0)), t_best)
            out[ci, r0:r0 + rows] = torch.round(t_best.clamp(0.0, 1.0) * 65535.0).to(torch.int32).cpu().numpy().astype(np.uint16)
    return out


def sky_luts(seed=7):
    """Stand-in LUT contents (the LUT generators are a 'next' row): smooth positive RGBA16F fields."""
    g = rng(seed)

    def smooth(h, w, scale):
        y, x = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
        out = np.zeros((h, w, 4), dtype=np.float32)
        for c in range(3):
            a, b, p = g.uniform(0.5, 3.0), g.uniform(0.5, 3.0), g.uniform(0, 6.28)
            out[..., c] = scale * (0.55 + 0.45 * np.sin(a * 6.28 * x + p) * np.cos(b * 3.14 * y))
        out[..., 3] = 1.0
        return out.astype(np.float16)

    return {"transmittance": smooth(64, 256, 1.0), "sky_view": smooth(200, 200, 0.3)}


def point_lights(view, count, radius, seed=8):
    """SURVEY §8-d lights: positions uniform in the view-frustum slab d in [1,60] m, colour U[0,1]^3, intensity
    log-uniform [1e3,1e5]. Returns a (count, 8) float32 array laid out as sah_point_light."""
    g = rng(seed)
    proj = np.array(view.gpu_data.projection[:], dtype=np.float32).reshape(4, 4)
    inv_view = np.array(view.gpu_data.inverse_view[:], dtype=np.float32).reshape(4, 4)
    d = g.uniform(1.0, 60.0, count).astype(np.float32)
    nx = g.uniform(-1.0, 1.0, count).astype(np.float32)
    ny = g.uniform(-1.0, 1.0, count).astype(np.float32)
    pv = np.stack([nx / proj[0, 0] * d, ny / proj[1, 1] * d, -d, np.ones_like(d)], axis=-1)
    pw = np.einsum("ki,nk->ni", inv_view, pv)[:, :3]
    out = np.zeros((count, 8), dtype=np.float32)
    out[:, 0:3] = pw
    out[:, 3] = radius
    out[:, 4:7] = g.random((count, 3), dtype=np.float32)
    out[:, 7] = np.exp(g.uniform(math.log(1e3), math.log(1e5), count)).astype(np.float32)
    return out


def pack_r11g11b10(rgb):
    """fp32 (...,3) >= 0 -> B10G11R11_UFLOAT_PACK32 words, truncating (round toward zero)."""
    h = np.clip(rgb, 0.0, 65024.0).astype(np.float16)  # exact path through fp16 bit patterns
    hb = h.view(np.uint16).astype(np.uint32)
    r = (hb[..., 0] >> 4) & 0x7FF
    gch = (hb[..., 1] >> 4) & 0x7FF
    b = (hb[..., 2] >> 5) & 0x3FF
    return (r | (gch << 11) | (b << 22)).astype(np.uint32)


def probe_atlases(seed=9):
    """Irradiance cache atlases (RenderCore/render/gi/irradiance_cache.cpp:94-183): irradiance U[0,4] packed R11G11B10
    224x256x32, depth moments RG16F 384x384x32 with y >= x^2, validity R8 32x32x32 with 90 % ones."""
    g = rng(seed)
    irr = pack_r11g11b10(g.uniform(0.0, 4.0, (32, 256, 224, 3)).astype(np.float32))
    dx = g.uniform(0.2, 6.0, (32, 384, 384)).astype(np.float32)
    dy = dx * dx + g.uniform(0.0, 1.0, (32, 384, 384)).astype(np.float32)
    depth = np.stack([dx, dy], axis=-1).astype(np.float16)
    validity = np.where(g.random((32, 32, 32)) < 0.9, 255, 0).astype(np.uint8)
    return {"irradiance": irr, "depth": depth, "validity": validity}


def rtgi_planes(width, height, seed=10):
    g = rng(seed)
    v = g.standard_normal((height, width, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    ray = np.zeros((height, width, 4), dtype=np.float16)
    ray[..., :3] = v.astype(np.float16)
    ray[..., 3] = g.uniform(0.1, 30.0, (height, width)).astype(np.float16)
    irr = np.zeros((height, width, 4), dtype=np.float16)
    irr[..., :3] = g.uniform(0.0, 8.0, (height, width, 3)).astype(np.float16)
    noise = g.integers(0, 256, (128, 128, 4), dtype=np.uint8)
    return {"ray_buffer": ray, "ray_irradiance": irr, "noise": noise}


def hdr_scene(width, height, seed=11):
    """An RGBA16F lit-scene stand-in for the post-chain tests: mostly < 4 with sparse bright highlights."""
    g = rng(seed)
    c = g.gamma(1.2, 0.5, (height, width, 3)).astype(np.float32)
    hot = g.random((height, width)) < 0.002
    c[hot] *= 200.0
    out = np.zeros((height, width, 4), dtype=np.float16)
    out[..., :3] = np.minimum(c, 60000.0).astype(np.float16)
    out[..., 3] = 1.0
    return out


def probe_maintenance_inputs(seed=12, num_probes=48):
    """Inputs of the irradiance-cache maintenance passes (a11; RenderCore/render/gi/irradiance_cache.cpp:94-183, 585-724): the five
    probe atlases (random finite R11G11B10 / RG16F / R8 contents), trace results 20 x 20 x P RGBA16F (a = hit distance, ~25 % misses
    with a <= 0) and P distinct probe ids that include the grid corners."""
    g = rng(seed)

    def r11(shape):
        return pack_r11g11b10(g.uniform(0.0, 8.0, shape + (3,)).astype(np.float32))

    atl = {
        "rtgi": r11((32, 256, 224)),
        "light_cache": r11((32, 416, 416)),
        "depth": g.uniform(0.0, 6.0, (32, 384, 384, 2)).astype(np.float16),
        "average": r11((32, 32, 32)),
        "validity": g.integers(0, 256, (32, 32, 32), dtype=np.uint8),
    }
    trace = np.zeros((num_probes, 20, 20, 4), dtype=np.float16)
    trace[..., :3] = g.uniform(0.0, 5.0, (num_probes, 20, 20, 3)).astype(np.float16)
    dist = g.uniform(0.05, 30.0, (num_probes, 20, 20)).astype(np.float16)
    miss = g.random((num_probes, 20, 20)) < 0.25
    dist[miss] = np.where(g.random(int(miss.sum())) < 0.5, np.float16(-1.0), np.float16(0.0))
    trace[..., 3] = dist
    if num_probes > 2:
        trace[1, ..., 3] = np.float16(-1.0)  # a probe whose rays all miss
        trace[2, ..., 3] = np.float16(2.5)   # a probe whose rays all hit
    cells = g.permutation(32 * 32 * 32)[:num_probes]
    ids = np.stack([cells % 32, (cells // 32) % 32, cells // 1024], axis=-1).astype(np.uint32)
    corners = np.array([[0, 0, 0], [31, 31, 31], [31, 0, 31], [0, 31, 0]], dtype=np.uint32)
    k = min(len(corners), num_probes)
    # keep ids distinct: drop any random id that equals a corner, then prepend the corners
    keep = [tuple(i) for i in ids if tuple(i) not in {tuple(c) for c in corners[:k]}]
    ids = np.array([tuple(c) for c in corners[:k]] + keep, dtype=np.uint32)[:num_probes]
    return atl, trace, np.ascontiguousarray(ids)
