"""Argument fuzz of the C ABI on a machine WITHOUT a HIP device (run by tests/test_abi_fuzz.py in a child process, so that a fault is a
failed test and not a dead test runner).  A detached context (sah_debug_create_detached: no device behind it) receives random extents,
pitches, formats, row ranges, made-up device addresses and null sub-pointers through every entry point include/sah_hip.h declares; each
call must come back with a status code of the sah_status enum — the argument checks of csrc/api*.cpp are complete exactly when nothing
here can reach a host-side dereference of a device address, a division by a zero extent or an out-of-range table index.  Device addresses
are never dereferenced on the host by contract; host structures (uniform blocks, descriptors, lists of descriptors) are either NULL or
valid memory filled with random bits, because a caller's invalid HOST pointer is outside what a C ABI can check.

    python tests/abi_fuzz_child.py SEED ITERATIONS      (prints one line per entry point: calls and the status codes seen)
"""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from androidrenderer_amd import _abi, lib  # noqa: E402

STATUS = set(range(-6, 1))
FORMATS = [9, 37, 43, 76, 83, 97, 100, 122, 124, 126, 0, 1, 50, 999, 0xFFFFFFFF]
EXTENTS = [0, 1, 2, 3, 7, 16, 31, 32, 33, 63, 64, 100, 128, 224, 256, 384, 416, 1080, 1920, 3840, 4096, 65535, 65536, 2 ** 31 - 1, 2 ** 32 - 1]
ADDRESSES = [0, 0x1000, 0x1001, 0x1002, 0x1004, 0x1008, 0x7f00_0000_0000, 0xffff_ffff_ffff_fff0, 8]


class Fuzz:
    def __init__(self, seed):
        self.g = np.random.default_rng(seed)
        self.keep = []
        self.wild = 0.1

    def pick(self, seq):
        return seq[int(self.g.integers(0, len(seq)))]

    # Most fields are what a correct caller would pass and a few are not (`wild`): a call whose every argument is random fails its first
    # check and never reaches the later ones
    def is_wild(self):
        return self.g.random() < self.wild

    def u32(self, sane=(0, 0, 1, 16, 64, 270, 1080)):
        if not self.is_wild():
            return int(self.pick(sane))
        return int(self.pick(EXTENTS)) if self.g.random() < 0.7 else int(self.g.integers(0, 2 ** 32))

    def addr(self):
        return int(self.pick(ADDRESSES)) if self.is_wild() else 0x7f12_3456_0000 + 0x100 * int(self.g.integers(0, 4096))

    def pitch(self, width, fmt):
        bpp = _abi.FORMAT_BPP.get(fmt, 4)
        exact = (width * bpp) & 0xFFFFFFFF
        if not self.is_wild():
            return int(self.pick([exact, (exact + 255) & 0xFFFFFF00]))
        return int(self.pick([0, (exact - 1) & 0xFFFFFFFF, (exact + 1) & 0xFFFFFFFF, (exact + 2) & 0xFFFFFFFF, int(self.g.integers(0, 2 ** 32))]))

    def extent(self, like):
        if like is not None and not self.is_wild():
            return like
        n = len(like) if like is not None else 2
        return tuple(int(self.pick(EXTENTS)) if self.is_wild() else int(self.pick([16, 64, 100, 128, 1080, 1920])) for _ in range(n))

    def plane(self, fmt=None, like=None):
        fmt = fmt if (fmt is not None and not self.is_wild()) else self.pick(FORMATS)
        w, h = self.extent(like if like is not None else None)
        return _abi.Plane(self.addr(), w, h, self.pitch(w, fmt), fmt)

    def volume(self, fmt=None, like=None):
        fmt = fmt if (fmt is not None and not self.is_wild()) else self.pick(FORMATS)
        w, h, d = self.extent(like if like is not None else (64, 64, 4))
        rp = self.pitch(w, fmt)
        sp = (rp * h) & 0xFFFFFFFF if not self.is_wild() else int(self.pick([0, (rp * h - 1) & 0xFFFFFFFF, (rp * h + 4) & 0xFFFFFFFF, int(self.g.integers(0, 2 ** 32))]))
        return _abi.Volume(self.addr(), w, h, d, rp, sp, fmt)

    def random_bits(self, struct):
        """a host structure of the given ctypes type filled with random bytes (uniform blocks: any floats, NaN and inf included)"""
        s = struct()
        raw = self.g.integers(0, 256, C.sizeof(struct), dtype=np.uint8).tobytes()
        C.memmove(C.byref(s), raw, C.sizeof(struct))
        self.keep.append(s)
        return s

    def ptr(self, obj, null_p=0.06):
        """byref(obj), or NULL with a small probability"""
        if self.g.random() < null_p:
            return None
        self.keep.append(obj)
        return C.pointer(obj)

    def mipchain(self, like=None):
        mc = _abi.MipChain()
        mc.num_mips = int(self.pick([6, 6, 6, 2, 8])) if not self.is_wild() else int(self.pick([0, 1, 9, 200, 2 ** 31, 2 ** 32 - 1]))
        w, h = like or self.extent(None)
        for i in range(_abi.MAX_BLOOM_MIPS):
            w, h = max(1, w // 2), max(1, h // 2)
            mc.mips[i] = self.plane(_abi.FORMAT_R16G16B16A16_SFLOAT, like=(w, h))
        return mc

    def gi(self):
        gi = self.random_bits(_abi.GI)
        gi.kind = int(self.pick([0, 1, 2, 3])) if not self.is_wild() else int(self.pick([4, 0xFFFFFFFF]))
        gi.lpv_red, gi.lpv_green, gi.lpv_blue = (self.volume(_abi.FORMAT_R16G16B16A16_SFLOAT, like=(128, 32, 32)) for _ in range(3))
        casc = (_abi.LpvCascadeMatrices * 4)()
        self.keep.append(casc)
        gi.lpv_cascades = None if self.g.random() < 0.2 else C.cast(casc, C.POINTER(_abi.LpvCascadeMatrices))
        gi.lpv_num_cascades = 4 if not self.is_wild() else int(self.pick([0, 1, 5, 2 ** 31]))
        gi.probe_irradiance = self.volume(_abi.FORMAT_B10G11R11_UFLOAT_PACK32, like=(224, 256, 32))
        gi.probe_depth = self.volume(_abi.FORMAT_R16G16_SFLOAT, like=(384, 384, 32))
        gi.probe_validity = self.volume(_abi.FORMAT_R8_UNORM, like=(32, 32, 32))
        gi.probe_size[0], gi.probe_size[1] = (5, 6) if not self.is_wild() else (int(self.pick([0, 31, 2 ** 31])), int(self.pick([0, 31, 2 ** 31])))
        size = self.extent(None)
        gi.ray_buffer, gi.ray_irradiance = (self.plane(_abi.FORMAT_R16G16B16A16_SFLOAT, like=size) for _ in range(2))
        gi.noise = self.plane(_abi.FORMAT_R8G8B8A8_UNORM, like=(128, 128))
        gi.num_extra_rays = int(self.pick([0, 0, 3, 2 ** 31]))
        gi.lpv_generation = int(self.pick([0, 1, _abi.GENERATION_TRACKED]))
        gi.probe_generation = int(self.pick([0, 1, _abi.GENERATION_TRACKED]))
        return gi

    def lighting_desc(self):
        size = self.extent(None)
        d = _abi.LightingDesc()
        gb = _abi.GBuffer(self.plane(_abi.FORMAT_R8G8B8A8_SRGB, size), self.plane(_abi.FORMAT_R16G16B16A16_SFLOAT, size), self.plane(_abi.FORMAT_R8G8B8A8_UNORM, size),
                          self.plane(_abi.FORMAT_R8G8B8A8_SRGB, size), self.plane(_abi.FORMAT_D32_SFLOAT, size))
        d.gbuffer = self.ptr(gb)
        d.ao = self.ptr(self.plane(_abi.FORMAT_R32_SFLOAT, size), 0.4)
        d.lit = self.ptr(self.plane(_abi.FORMAT_R16G16B16A16_SFLOAT, size))
        d.view = self.ptr(self.random_bits(_abi.ViewData))
        sun = self.random_bits(_abi.SunLightConstants)
        sun.shadow_mode = int(self.pick([0, 1, 2])) if not self.is_wild() else int(self.pick([3, 0xFFFFFFFF]))
        d.sun = self.ptr(sun)
        d.shadowmap = self.ptr(self.volume(_abi.FORMAT_D16_UNORM, like=(4096, 4096, 4)), 0.3)
        d.shadow_mask = self.ptr(self.plane(_abi.FORMAT_R32_SFLOAT, size), 0.4)
        d.lights = self.ptr(_abi.LightList(self.addr(), int(self.pick([0, 1, 64, 1024, 2 ** 31]))), 0.6)
        d.gi = self.ptr(self.gi(), 0.3)
        d.sky = self.ptr(_abi.SkyLuts(self.plane(_abi.FORMAT_R16G16B16A16_SFLOAT, (256, 64)), self.plane(_abi.FORMAT_R16G16B16A16_SFLOAT, (200, 200))), 0.4)
        d.flags = int(self.pick([0, 1, 2, 3, 0xFFFFFFFF]))
        d.row_begin, d.row_end = (0, 0) if not self.is_wild() else (self.u32(), self.u32())
        return d

    def scene(self):
        s = _abi.SceneGeometry(self.addr(), self.addr(), self.addr(), self.addr(), self.addr(), self.u32(), self.u32(), self.u32(), self.u32(), self.addr(),
                               self.addr(), self.u32(), 0)
        return s

    def atlases(self):
        return _abi.ProbeAtlases(self.volume(_abi.FORMAT_B10G11R11_UFLOAT_PACK32, (224, 256, 32)), self.volume(_abi.FORMAT_B10G11R11_UFLOAT_PACK32, (416, 416, 32)),
                                 self.volume(_abi.FORMAT_R16G16_SFLOAT, (384, 384, 32)), self.volume(_abi.FORMAT_B10G11R11_UFLOAT_PACK32, (32, 32, 32)),
                                 self.volume(_abi.FORMAT_R8_UNORM, (32, 32, 32)))


def main():
    seed, iterations = int(sys.argv[1]), int(sys.argv[2])
    L = lib.load()
    L.sah_debug_create_detached.argtypes = [C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    rc = L.sah_debug_create_detached(C.byref(h))
    if rc == _abi.SAH_ERR_UNSUPPORTED:
        print("SKIP: a HIP device is present (the fuzz's made-up addresses must not reach a GPU)")
        return 0
    assert rc == 0 and h.value, rc
    f = Fuzz(seed)
    seen = collections.defaultdict(collections.Counter)
    P, V = C.POINTER(_abi.Plane), C.POINTER(_abi.Volume)

    def ctx():
        return None if f.g.random() < 0.03 else h

    def opt(obj, null_p=0.06):
        return f.ptr(obj, null_p)

    def floats(n):
        a = (C.c_float * n)(*[float(x) for x in f.g.standard_normal(n)])
        f.keep.append(a)
        return None if f.g.random() < 0.06 else a

    size = lambda: f.extent(None)
    rgba16 = _abi.FORMAT_R16G16B16A16_SFLOAT

    def vols3():
        a = (_abi.Volume * 3)(*[f.volume(rgba16, (128, 32, 32)) for _ in range(3)])
        f.keep.append(a)
        return None if f.g.random() < 0.1 else a

    def cascades():
        a = (_abi.LpvCascadeMatrices * 4)()
        f.keep.append(a)
        return None if f.g.random() < 0.15 else a

    def handles(n=16):
        b = C.create_string_buffer(bytes(f.g.integers(0, 256, _abi.IPC_HANDLE_BYTES * n, dtype=np.uint8)), _abi.IPC_HANDLE_BYTES * n)
        f.keep.append(b)
        return None if f.g.random() < 0.15 else b

    def chain_create():
        plan = f.random_bits(_abi.ChainPlan)
        frames = (_abi.ChainFrame * 2)()
        f.keep.append(frames)
        for k in range(2):
            s = size()
            for j in range(2):
                d = f.lighting_desc()
                f.keep.append(d)
                if f.g.random() < 0.8:
                    frames[k].lighting[j] = C.pointer(d)
            frames[k].lit, frames[k].antialiased, frames[k].out = f.plane(rgba16, s), f.plane(rgba16, s), f.plane(_abi.FORMAT_R8G8B8A8_SRGB, s)
            frames[k].bloom = f.mipchain(s)
        out = C.c_void_p()
        rc = L.sah_chain_create(ctx(), opt(plan), frames if f.g.random() < 0.9 else None, f.u32(), f.u32(), None, None, None, C.byref(out) if f.g.random() < 0.95 else None)
        if rc == 0 and out.value:  # (cannot happen without a device; kept honest anyway)
            L.sah_chain_destroy(out)
        return rc

    def probe_trace():
        d = f.random_bits(_abi.ProbeTraceDesc)
        d.probes_to_update, d.num_probes = f.addr(), int(f.pick([0, 1, 1024, 2 ** 31]))
        d.sun, d.sky, d.noise = opt(f.random_bits(_abi.SunLightConstants)), opt(_abi.SkyLuts(f.plane(rgba16, (256, 64)), f.plane(rgba16, (200, 200)))), opt(f.plane(_abi.FORMAT_R8G8B8A8_UNORM, (128, 128)))
        d.probe_irradiance, d.probe_depth, d.probe_validity = f.volume(122, (224, 256, 32)), f.volume(83, (384, 384, 32)), f.volume(9, (32, 32, 32))
        d.trace_results = f.volume(rgba16, (20, 20, 1024))
        return L.sah_probe_trace(ctx(), opt(d))

    L.sah_comm_unique_id.argtypes = [C.c_void_p]
    calls = {
        "sah_status_string": lambda: 0 if L.sah_status_string(int(f.g.integers(-50, 50))) is not None else -1,
        "sah_last_error": lambda: 0 if L.sah_last_error(ctx()) is not None else -1,
        "sah_set_stream": lambda: L.sah_set_stream(ctx(), None),
        "sah_sync": lambda: L.sah_sync(ctx()),
        "sah_lighting": lambda: L.sah_lighting(ctx(), opt(f.lighting_desc(), 0.05)),
        "sah_copy_scene": lambda: L.sah_copy_scene(ctx(), opt(f.plane(rgba16)), opt(f.plane(rgba16))),
        "sah_copy_scene_rows": lambda: L.sah_copy_scene_rows(ctx(), opt(f.plane(rgba16)), opt(f.plane(rgba16)), f.u32(), f.u32()),
        "sah_copy_scene_bloom_mip0_rows": lambda: (lambda s: L.sah_copy_scene_bloom_mip0_rows(ctx(), opt(f.plane(rgba16, s)), opt(f.plane(rgba16, s)), opt(f.mipchain(s)), f.u32(), f.u32(), f.u32(), f.u32()))(size()),
        "sah_bloom": lambda: (lambda s: L.sah_bloom(ctx(), opt(f.plane(rgba16, s)), opt(f.mipchain(s))))(size()),
        "sah_bloom_mip0_rows": lambda: (lambda s: L.sah_bloom_mip0_rows(ctx(), opt(f.plane(rgba16, s)), opt(f.mipchain(s)), f.u32(), f.u32()))(size()),
        "sah_bloom_from_mip0": lambda: (lambda s: L.sah_bloom_from_mip0(ctx(), opt(f.plane(rgba16, s)), opt(f.mipchain(s))))(size()),
        "sah_bloom_mip_rows": lambda: (lambda s: L.sah_bloom_mip_rows(ctx(), opt(f.plane(rgba16, s)), opt(f.mipchain(s)), f.u32(), f.u32(), f.u32()))(size()),
        "sah_bloom_from_mip": lambda: (lambda s: L.sah_bloom_from_mip(ctx(), opt(f.plane(rgba16, s)), opt(f.mipchain(s)), f.u32()))(size()),
        "sah_tonemap": lambda: (lambda s: L.sah_tonemap(ctx(), opt(f.plane(rgba16, s)), opt(f.mipchain(s)), opt(f.plane(_abi.FORMAT_R8G8B8A8_SRGB, s)), f.u32(), f.u32()))(size()),
        "sah_tonemap_ex": lambda: (lambda s: L.sah_tonemap_ex(ctx(), opt(f.plane(rgba16, s)), opt(f.mipchain(s)), opt(f.plane(_abi.FORMAT_R8G8B8A8_SRGB, s)), f.u32(), f.u32(), f.u32()))(size()),
        "sah_lpv_clear": lambda: L.sah_lpv_clear(ctx(), *[opt(f.volume(rgba16, (128, 32, 32)), 0.3) for _ in range(4)], f.u32()),
        "sah_lpv_propagate": lambda: L.sah_lpv_propagate(ctx(), vols3(), vols3(), int(f.pick([0, 1, 4, 4, 5, 2 ** 31])), int(f.pick([0, 1, 2, 3]))),
        "sah_sky_update_luts": lambda: L.sah_sky_update_luts(ctx(), opt(f.plane(rgba16, (256, 64))), opt(f.plane(rgba16, (32, 32))), opt(f.plane(rgba16, (200, 200))), floats(3)),
        "sah_ao_clear": lambda: L.sah_ao_clear(ctx(), opt(f.plane(_abi.FORMAT_R32_SFLOAT))),
        "sah_probe_copy": lambda: L.sah_probe_copy(ctx(), opt(f.atlases()), opt(f.atlases()), C.cast(floats(12), C.POINTER(C.c_float * 3))),
        "sah_probe_update": lambda: L.sah_probe_update(ctx(), opt(f.atlases()), opt(f.volume(rgba16, (20, 20, 1024))), f.addr(), int(f.pick([0, 1, 1024, 2 ** 31]))),
        "sah_probe_notify_updated": lambda: L.sah_probe_notify_updated(ctx(), opt(f.volume(122, (224, 256, 32))), f.addr(), int(f.pick([0, 1, 1024, 2 ** 31]))),
        "sah_shadow_render": lambda: L.sah_shadow_render(ctx(), opt(f.scene()), opt(f.random_bits(_abi.SunLightConstants)), int(f.pick([0, 1, 4, 5, 2 ** 31])), opt(f.volume(124, (4096, 4096, 4))), f.addr()),
        "sah_gbuffer_render": lambda: (lambda s: L.sah_gbuffer_render(ctx(), opt(f.scene()), opt(f.random_bits(_abi.ViewData)), opt(_abi.GBuffer(f.plane(43, s), f.plane(rgba16, s), f.plane(37, s), f.plane(43, s), f.plane(126, s))), f.addr()))(size()),
        "sah_rsm_render": lambda: L.sah_rsm_render(ctx(), opt(f.scene()), opt(f.random_bits(_abi.SunLightConstants)), cascades(), int(f.pick([0, 1, 4, 5, 2 ** 31])), opt(_abi.RsmTargets(f.volume(43, (128, 128, 4)), f.volume(37, (128, 128, 4)), f.volume(124, (128, 128, 4)))), f.addr()),
        "sah_lpv_extract_vpls": lambda: L.sah_lpv_extract_vpls(ctx(), opt(_abi.RsmTargets(f.volume(43, (128, 128, 4)), f.volume(37, (128, 128, 4)), f.volume(124, (128, 128, 4)))), cascades(), f.u32(), float(f.g.standard_normal()), f.addr(), f.addr()),
        "sah_lpv_inject_vpls": lambda: L.sah_lpv_inject_vpls(ctx(), f.addr(), f.addr(), f.u32(), cascades(), f.u32(), int(f.pick([0, 1, 4, 5, 2 ** 31])), vols3()),
        "sah_rt_build": lambda: L.sah_rt_build(ctx(), opt(f.scene()), None if f.g.random() < 0.3 else (C.c_uint32 * 4)()),
        "sah_rtao": lambda: (lambda s: L.sah_rtao(ctx(), opt(f.random_bits(_abi.ViewData)), opt(f.plane(126, s)), opt(f.plane(rgba16, s)), opt(f.plane(37, (128, 128))), f.u32(), float(f.g.standard_normal()), opt(f.plane(100, s))))(size()),
        "sah_sun_shadow_mask": lambda: (lambda s: L.sah_sun_shadow_mask(ctx(), opt(f.random_bits(_abi.ViewData)), opt(f.random_bits(_abi.SunLightConstants)), opt(f.plane(126, s)), opt(f.plane(rgba16, s)), opt(f.plane(37, (128, 128))), opt(f.plane(100, s))))(size()),
        "sah_probe_trace": probe_trace,
        "sah_rtgi_trace": lambda: (lambda s: L.sah_rtgi_trace(ctx(), opt(f.random_bits(_abi.ViewData)), opt(f.random_bits(_abi.SunLightConstants)), opt(_abi.SkyLuts(f.plane(rgba16, (256, 64)), f.plane(rgba16, (200, 200)))), opt(f.plane(126, s)), opt(f.plane(rgba16, s)), opt(f.plane(37, (128, 128))), opt(f.plane(rgba16, s)), opt(f.plane(rgba16, s))))(size()),
        "sah_rt_set_rows": lambda: L.sah_rt_set_rows(ctx(), f.u32(), f.u32()),
        "sah_rt_set_bounces": lambda: L.sah_rt_set_bounces(ctx(), f.u32()),
        "sah_allgather_rows": lambda: L.sah_allgather_rows(ctx(), opt(f.plane(rgba16)), f.u32(), f.u32()),
        "sah_allgather_rows_reversed": lambda: L.sah_allgather_rows_reversed(ctx(), opt(f.plane(43)), f.u32(), f.u32()),
        "sah_allgather_bytes": lambda: L.sah_allgather_bytes(ctx(), f.addr(), int(f.g.integers(0, 2 ** 40))),
        "sah_comm_set_stream": lambda: L.sah_comm_set_stream(ctx(), None),
        "sah_comm_wait": lambda: L.sah_comm_wait(ctx()),
        "sah_comm_unique_id": lambda: L.sah_comm_unique_id(None),
        "sah_ipc_open": lambda: L.sah_ipc_open(ctx(), handles(1)),
        "sah_ipc_connect": lambda: L.sah_ipc_connect(ctx(), handles()),
        "sah_ipc_export": lambda: L.sah_ipc_export(ctx(), f.addr(), int(f.g.integers(0, 2 ** 40)), handles(1)),
        "sah_ipc_register": lambda: L.sah_ipc_register(ctx(), f.addr(), int(f.g.integers(0, 2 ** 40)), handles()),
        "sah_ipc_unregister": lambda: L.sah_ipc_unregister(ctx(), f.addr()),
        "sah_ipc_reset": lambda: L.sah_ipc_reset(ctx()),
        "sah_bloom_source_rows": lambda: L.sah_bloom_source_rows(f.u32((0, 1, 37, 75, 1080)), f.u32((0, 1, 18, 37, 540)), f.u32(), f.u32(), None if f.g.random() < 0.1 else (C.c_uint32 * 2)()),
        "sah_chain_create": chain_create,
        "sah_chain_submit": lambda: L.sah_chain_submit(None, None, None),
        "sah_chain_flush": lambda: L.sah_chain_flush(None),
        "sah_chain_counts": lambda: L.sah_chain_counts(None, None, None),
        "sah_create": lambda: L.sah_create(None if f.g.random() < 0.3 else C.byref(C.c_void_p()), int(f.g.integers(-2, 20)), int(f.g.integers(-2, 20)), int(f.g.integers(-2, 20)), None),
    }
    uncovered = sorted(set(lib.EXPORTS) - set(calls) - {"sah_abi_version", "sah_destroy", "sah_chain_destroy"})
    assert not uncovered, f"entry points without a fuzz case: {uncovered}"
    names = sorted(calls)
    for i in range(iterations):
        for name in names:
            rc = calls[name]()
            if rc not in STATUS:
                print(f"FAIL: {name} returned {rc}, not a sah_status (iteration {i}, seed {seed})")
                return 1
            seen[name][rc] += 1
        f.keep.clear()
    L.sah_chain_destroy(None)
    L.sah_destroy(None)
    L.sah_destroy(h)
    for name in names:
        print(f"{name:34s} " + "  ".join(f"{_abi_name(rc)}: {n}" for rc, n in sorted(seen[name].items(), reverse=True)))
    print(f"OK: {iterations} iterations x {len(names)} entry points, seed {seed}")
    return 0


def _abi_name(rc):
    return {0: "ok", -1: "invalid_argument", -2: "unsupported_format", -3: "hip", -4: "no_device", -5: "comm", -6: "unsupported"}[rc]


if __name__ == "__main__":
    sys.exit(main())
