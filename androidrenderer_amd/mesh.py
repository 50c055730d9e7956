"""Triangle meshes for the scene rasteriser: host arrays in the layouts of include/sah_hip.h
(sah_vertex_data, sah_material, sah_primitive) and the sah_scene_geometry descriptor over numpy (host) or torch (device)
storage.  The atrium is the same set of boxes synth.atrium_gbuffer / synth.atrium_shadowmap ray-cast, so the rasterised
G-buffer and shadow cascades can stand in for them."""
import ctypes as C

import numpy as np

from . import _abi, synth

VERTEX_DATA = np.dtype([("normal", np.float32, 3), ("tangent", np.float32, 4), ("texcoord", np.float32, 2), ("color", np.uint32)])
MATERIAL = np.dtype([("base_color_tint", np.float32, 4), ("emission_factor", np.float32, 4), ("metalness_factor", np.float32),
                     ("roughness_factor", np.float32), ("opacity_threshold", np.float32), ("padding1", np.float32),
                     ("base_color_texel", np.float32, 4), ("normal_texel", np.float32, 4), ("data_texel", np.float32, 4),
                     ("emission_texel", np.float32, 4)])
PRIMITIVE = np.dtype([("model", np.float32, 16), ("first_index", np.uint32), ("index_count", np.uint32), ("vertex_offset", np.int32),
                      ("type", np.uint32), ("material", np.uint32), ("padding", np.uint32, 3)])
assert VERTEX_DATA.itemsize == C.sizeof(_abi.VertexData) == 40
assert MATERIAL.itemsize == C.sizeof(_abi.Material) == 112
assert PRIMITIVE.itemsize == C.sizeof(_abi.Primitive) == 96

IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


def sampler(mag=_abi.FILTER_LINEAR, min=_abi.FILTER_LINEAR, mipmap=_abi.FILTER_LINEAR, address_u=_abi.ADDRESS_REPEAT, address_v=_abi.ADDRESS_REPEAT,
            bias=0.0, min_lod=0.0, max_lod=1000.0, max_anisotropy=0.0):
    """sah_sampler; the defaults are what gltf_model.cpp:520-586 makes of a glTF sampler with linear filters (VK_LOD_CLAMP_NONE = 1000),
    except for its anisotropy: pass max_anisotropy=8.0 for that (0 = off keeps the committed golden vectors of the isotropic path valid)."""
    return _abi.Sampler(mag, min, mipmap, address_u, address_v, bias, min_lod, max_lod, max_anisotropy, 0)


def mip_sizes(width, height, count=None):
    out = [(width, height)]
    while (out[-1] != (1, 1)) and (count is None or len(out) < count):
        out.append((max(out[-1][0] // 2, 1), max(out[-1][1] // 2, 1)))
    return out


def material(base=(1, 1, 1, 1), rough=0.5, metal=0.0, emission=(0, 0, 0, 0), opacity_threshold=0.0, normal_texel=(0.5, 0.5, 1.0, 1.0)):
    m = np.zeros((), dtype=MATERIAL)
    m["base_color_tint"] = base
    m["emission_factor"] = emission
    m["metalness_factor"], m["roughness_factor"], m["opacity_threshold"] = metal, rough, opacity_threshold
    m["base_color_texel"] = (1, 1, 1, 1)
    m["normal_texel"] = normal_texel
    m["data_texel"] = (0, 1, 1, 0)
    m["emission_texel"] = (1, 1, 1, 1)
    return m


class Mesh:
    """Growable host mesh.  `primitives` are draws over shared vertex / index streams."""

    def __init__(self):
        self.positions, self.vertex_data, self.indices, self.primitives, self.materials = [], [], [], [], []
        self.textures, self.material_textures = [], []

    def add_material(self, m, base_color=_abi.TEXTURE_NONE, normal=_abi.TEXTURE_NONE, data=_abi.TEXTURE_NONE, emission=_abi.TEXTURE_NONE):
        """`m` from material(); the four keyword arguments are indices returned by add_texture (default: the constant texel of `m`)."""
        self.materials.append(m)
        self.material_textures.append((base_color, normal, data, emission))
        return len(self.materials) - 1

    def add_texture(self, mips, srgb=False, smp=None):
        """mips: list of (h, w, 4) uint8 arrays, level 0 first."""
        assert 1 <= len(mips) <= _abi.MAX_TEXTURE_MIPS
        self.textures.append(([np.ascontiguousarray(m, dtype=np.uint8) for m in mips], _abi.FORMAT_R8G8B8A8_SRGB if srgb else _abi.FORMAT_R8G8B8A8_UNORM,
                              smp if smp is not None else sampler()))
        return len(self.textures) - 1

    def add_primitive(self, positions, normals, indices, material_index, model=IDENTITY, ptype=_abi.PRIMITIVE_TYPE_SOLID, colors=None,
                      tangents=None, texcoords=None):
        positions = np.asarray(positions, dtype=np.float32).reshape(-1, 3)
        n = positions.shape[0]
        vd = np.zeros(n, dtype=VERTEX_DATA)
        vd["normal"] = np.asarray(normals, dtype=np.float32).reshape(-1, 3)
        if tangents is None:  # any unit vector orthogonal to the normal, handedness +1
            nrm = vd["normal"]
            helper = np.where(np.abs(nrm[:, 1:2]) < 0.9, np.array([[0.0, 1.0, 0.0]], np.float32), np.array([[1.0, 0.0, 0.0]], np.float32))
            t = np.cross(helper, nrm)
            t /= np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-20)
            tangents = np.concatenate([t, np.ones((n, 1), np.float32)], axis=1)
        vd["tangent"] = np.asarray(tangents, dtype=np.float32).reshape(-1, 4)
        vd["color"] = 0xffffffff if colors is None else np.asarray(colors, dtype=np.uint32)
        if texcoords is not None:
            vd["texcoord"] = np.asarray(texcoords, dtype=np.float32).reshape(-1, 2)
        first_vertex = sum(p.shape[0] for p in self.positions)
        first_index = sum(i.shape[0] for i in self.indices)
        self.positions.append(positions)
        self.vertex_data.append(vd)
        idx = np.asarray(indices, dtype=np.uint32).reshape(-1)
        self.indices.append(idx)
        p = np.zeros((), dtype=PRIMITIVE)
        p["model"] = np.asarray(model, dtype=np.float32).reshape(16)
        p["first_index"], p["index_count"], p["vertex_offset"] = first_index, idx.shape[0], first_vertex
        p["type"], p["material"] = ptype, material_index
        self.primitives.append(p)
        return len(self.primitives) - 1

    def add_instance(self, primitive_index, model, material_index=None):
        """Another draw of an existing primitive's index range with its own model matrix (instancing)."""
        p = self.primitives[primitive_index].copy()
        p["model"] = np.asarray(model, dtype=np.float32).reshape(16)
        if material_index is not None:
            p["material"] = material_index
        self.primitives.append(p)

    def add_box(self, bmin, bmax, material_index, subdiv=1, uv_scale=0.25, **kw):
        """Axis-aligned box, outward normals, every face a grid of subdiv x subdiv quads; texcoords = the two in-face world
        coordinates * uv_scale (tiling textures)."""
        bmin, bmax = np.asarray(bmin, np.float32), np.asarray(bmax, np.float32)
        pos, nrm, idx, uvs = [], [], [], []
        n1 = subdiv + 1
        for axis in range(3):
            for sign in (-1.0, 1.0):
                u, v = (axis + 1) % 3, (axis + 2) % 3
                n = np.zeros(3, np.float32)
                n[axis] = sign
                base = len(pos)
                tu = np.linspace(bmin[u], bmax[u], n1, dtype=np.float32)
                tv = np.linspace(bmin[v], bmax[v], n1, dtype=np.float32)
                for j in range(n1):
                    for i in range(n1):
                        c = np.zeros(3, np.float32)
                        c[axis] = bmax[axis] if sign > 0 else bmin[axis]
                        c[u], c[v] = tu[i], tv[j]
                        pos.append(c)
                        nrm.append(n)
                        uvs.append((np.float32(tu[i]) * np.float32(uv_scale), np.float32(tv[j]) * np.float32(uv_scale)))
                # counter-clockwise seen from outside in the right-handed y-up world (glTF); the projection has no y flip, so these
                # arrive clockwise in window coordinates (front faces, render_scene.cpp:196-197)
                quad = (0, 1, 2, 0, 2, 3) if sign > 0 else (0, 2, 1, 0, 3, 2)
                for j in range(subdiv):
                    for i in range(subdiv):
                        corners = (base + j * n1 + i, base + j * n1 + i + 1, base + (j + 1) * n1 + i + 1, base + (j + 1) * n1 + i)
                        idx += [corners[q] for q in quad]
        return self.add_primitive(pos, nrm, idx, material_index, texcoords=uvs, **kw)

    def arrays(self):
        def cat(parts, dtype, shape_tail=()):
            return np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros((0,) + shape_tail, dtype=dtype)
        return {"positions": cat(self.positions, np.float32, (3,)), "vertex_data": cat(self.vertex_data, VERTEX_DATA),
                "indices": cat(self.indices, np.uint32),
                "primitives": np.array(self.primitives, dtype=PRIMITIVE) if self.primitives else np.zeros(0, PRIMITIVE),
                "materials": np.array(self.materials, dtype=MATERIAL) if self.materials else np.zeros(0, MATERIAL),
                "textures": list(self.textures),
                "material_textures": np.array(self.material_textures, dtype=np.uint32).reshape(-1, 4)}


def geometry(arrays, keep=None):
    """sah_scene_geometry over `arrays` (numpy arrays -> host addresses for the oracle; torch uint8 tensors -> device addresses).
    `keep` collects the objects that must outlive the descriptor."""
    g = _abi.SceneGeometry()
    g._alive = []  # everything the descriptor points to lives as long as the descriptor itself

    def addr(a):
        g._alive.append(a)
        if keep is not None:
            keep.append(a)
        if hasattr(a, "data_ptr"):
            return a.data_ptr() if a.numel() else None
        return a.ctypes.data if a.size else None
    g.vertex_positions, g.vertex_data, g.indices = addr(arrays["positions"]), addr(arrays["vertex_data"]), addr(arrays["indices"])
    g.primitives, g.materials = addr(arrays["primitives"]), addr(arrays["materials"])
    g.num_vertices, g.num_indices = arrays["counts"]["vertices"], arrays["counts"]["indices"]
    g.num_primitives, g.num_materials = arrays["counts"]["primitives"], arrays["counts"]["materials"]
    textures = arrays.get("textures") or []
    if textures:
        table = (_abi.Texture * len(textures))()
        on_device = False
        for t, (mips, fmt, smp) in zip(table, textures):
            t.num_mips, t.sampler = len(mips), smp
            for i, m in enumerate(mips):
                on_device = on_device or hasattr(m, "data_ptr")
                h, w = int(m.shape[0]), int(m.shape[1])
                t.mips[i] = _abi.Plane(addr(m), w, h, w * 4, fmt)
        if on_device:  # the kernels read the table itself: it has to live in device memory too
            import torch
            dev = textures[0][0][0].device
            table_t = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
            g.textures = addr(table_t)
        else:
            g._alive.append(table)
            g.textures = C.addressof(table)
        g.num_textures = len(textures)
        g.material_textures = addr(arrays["material_textures"])
    return g


def with_counts(arrays):
    out = dict(arrays)
    out["counts"] = {"vertices": int(arrays["positions"].shape[0]), "indices": int(arrays["indices"].shape[0]),
                     "primitives": int(arrays["primitives"].shape[0]), "materials": int(arrays["materials"].shape[0])}
    return out


def to_device(arrays, device="cuda"):
    """Byte copies of the host arrays on `device` (torch uint8 tensors), with the element counts carried along."""
    import torch
    host = with_counts(arrays)
    out = {"counts": host["counts"]}
    for k in ("positions", "vertex_data", "indices", "primitives", "materials", "material_textures"):
        if k not in arrays:
            continue
        raw = np.frombuffer(np.ascontiguousarray(arrays[k]).tobytes(), dtype=np.uint8)
        out[k] = torch.from_numpy(raw.copy()).to(device)
    out["textures"] = [([torch.from_numpy(m.copy()).to(device) for m in mips], fmt, smp) for (mips, fmt, smp) in arrays.get("textures", [])]
    return out


# materials of synth.atrium_gbuffer: floor, long walls, end walls, gallery slabs, columns, lamps
_ATRIUM_BASE = [(0.45, 0.40, 0.33), (0.60, 0.52, 0.42), (0.50, 0.30, 0.22), (0.55, 0.55, 0.50), (0.62, 0.58, 0.50), (0.9, 0.8, 0.6)]
_ATRIUM_ROUGH = [0.65, 0.8, 0.7, 0.5, 0.4, 0.3]
_ATRIUM_METAL = [0.0, 0.0, 0.0, 0.1, 0.0, 0.9]


def atrium(subdiv=1):
    """The procedural atrium as a mesh: one primitive per box (12 * subdiv^2 triangles each), six materials, emissive lamps."""
    m = Mesh()
    for i in range(6):
        emission = (1.0, 0.67, 0.4, 0.0) if i == 5 else (0, 0, 0, 0)
        m.add_material(material(base=_ATRIUM_BASE[i] + (1.0,), rough=_ATRIUM_ROUGH[i], metal=_ATRIUM_METAL[i], emission=emission))
    for bmin, bmax, mat in synth._atrium_boxes():
        m.add_box(bmin, bmax, mat, subdiv=subdiv)
    return m


def random_texture(g, width, height, levels=None, srgb=False, smp=None, alpha=(0, 256)):
    """(mips, srgb, sampler) with INDEPENDENT random content per level, so that a wrong level selection cannot hide."""
    mips = []
    for (w, h) in mip_sizes(width, height, levels):
        t = g.integers(0, 256, (h, w, 4), dtype=np.uint8)
        t[..., 3] = g.integers(alpha[0], alpha[1], (h, w), dtype=np.uint8)
        mips.append(t)
    return mips, srgb, smp


def random_sampler(g):
    return sampler(mag=int(g.integers(0, 2)), min=int(g.integers(0, 2)), mipmap=int(g.integers(0, 2)), address_u=int(g.integers(0, 3)),
                   address_v=int(g.integers(0, 3)), bias=float(g.choice([0.0, 0.0, -0.75, 0.5, 1.25])), min_lod=float(g.choice([0.0, 0.0, 1.0])),
                   max_lod=float(g.choice([1000.0, 1000.0, 2.5, 0.25])), max_anisotropy=float(g.choice([0.0, 8.0, 8.0, 2.5, 16.0])))


def random_soup(seed, triangles=400, extent=6.0, size=(0.05, 3.0), cutout_fraction=0.3, instances=3, textured=False):
    """Triangle soup for parity tests: sizes from sub-pixel to screen-filling, random vertex colours (alpha drives the cutout
    test), random normals / tangents, a few instanced draws with rotated model matrices, degenerate and duplicate triangles.
    textured: random texcoords in [-2, 3]^2 and every material slot bound to one of eight random textures (sizes 1x1 .. 97x33, full
    or truncated mip chains, every filter / mipmap mode / address mode / LOD clamp combination the sampler struct can express)."""
    g = synth.rng(seed)
    m = Mesh()
    tex = []
    if textured:
        for (w, h, levels, srgb) in ((64, 64, None, True), (97, 33, None, False), (1, 1, None, True), (16, 128, 3, False), (256, 256, None, True),
                                     (8, 8, 1, False), (5, 3, None, True), (32, 16, None, False)):
            tex.append(m.add_texture(*random_texture(g, w, h, levels, srgb, random_sampler(g))))

    def slots():
        if not textured:
            return {}
        pick = lambda: int(g.choice(tex)) if g.uniform() < 0.85 else _abi.TEXTURE_NONE
        return dict(base_color=pick(), normal=pick(), data=pick(), emission=pick())
    mats = [m.add_material(material(base=tuple(g.uniform(0.2, 1.0, 3)) + (1.0,), rough=float(g.uniform(0.05, 1)), metal=float(g.uniform(0, 1)),
                                    emission=tuple(g.uniform(0, 2, 3)) + (0.0,), opacity_threshold=0.5,
                                    normal_texel=tuple(g.uniform(0.3, 0.7, 2)) + (1.0, 1.0)), **slots()) for _ in range(4)]
    prims = []
    for chunk in range(4):
        n = triangles // 4
        centre = g.uniform(-extent, extent, (n, 1, 3)).astype(np.float32)
        scale = np.exp(g.uniform(np.log(size[0]), np.log(size[1]), (n, 1, 1))).astype(np.float32)
        pos = (centre + scale * g.uniform(-1, 1, (n, 3, 3)).astype(np.float32)).reshape(-1, 3)
        if n > 8:
            pos[3:6] = pos[0:3]          # exact duplicate: a depth tie (G-buffer: the later draw stays; RSM: the earlier one)
            pos[8] = pos[7]              # degenerate
        nrm = g.normal(size=(3 * n, 3)).astype(np.float32)
        tan = np.concatenate([g.normal(size=(3 * n, 3)), g.choice([-1.0, 1.0], (3 * n, 1))], axis=1).astype(np.float32)
        col = g.integers(0, 1 << 32, 3 * n, dtype=np.uint64).astype(np.uint32)
        ptype = _abi.PRIMITIVE_TYPE_CUTOUT if g.uniform() < cutout_fraction or chunk == 3 else _abi.PRIMITIVE_TYPE_SOLID
        idx = g.permutation(3 * n).astype(np.uint32) if chunk == 1 else np.arange(3 * n, dtype=np.uint32)
        uv = g.uniform(-2.0, 3.0, (3 * n, 2)).astype(np.float32) if textured else None
        prims.append(m.add_primitive(pos, nrm, idx, mats[chunk], ptype=ptype, colors=col, tangents=tan, texcoords=uv))
    for k in range(instances):
        a = float(g.uniform(0, 2 * np.pi))
        model = np.eye(4, dtype=np.float32)
        model[0, 0], model[0, 2], model[2, 0], model[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
        model[:3, 3] = g.uniform(-2, 2, 3)
        m.add_instance(prims[k % len(prims)], model.T.reshape(16), mats[(k + 1) % 4])  # column-major
    return m
