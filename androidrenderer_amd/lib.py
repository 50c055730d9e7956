"""ctypes binding of libsah_hip.so (the C ABI in include/sah_hip.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this module raises, and every
entry point raises SahError on a non-zero status."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAH_HIP_LIBRARY") or os.path.join(_HERE, "libsah_hip.so")  # override: A/B builds of the same library


class SahError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"sah status {status}: {message}")
        self.status = status


_lib = None

# every symbol include/sah_hip.h declares
EXPORTS = ["sah_abi_version", "sah_status_string", "sah_last_error", "sah_create", "sah_destroy", "sah_comm_unique_id", "sah_set_stream",
           "sah_sync", "sah_lighting", "sah_copy_scene", "sah_copy_scene_rows", "sah_copy_scene_bloom_mip0_rows", "sah_bloom", "sah_bloom_mip0_rows", "sah_bloom_from_mip0", "sah_bloom_mip_rows", "sah_bloom_from_mip", "sah_bloom_source_rows", "sah_tonemap", "sah_tonemap_ex", "sah_lpv_clear", "sah_lpv_propagate", "sah_probe_notify_updated",
           "sah_sky_update_luts", "sah_ao_clear", "sah_probe_copy", "sah_probe_update", "sah_shadow_render", "sah_gbuffer_render", "sah_rsm_render", "sah_lpv_extract_vpls",
           "sah_lpv_inject_vpls", "sah_rt_build", "sah_rtao", "sah_sun_shadow_mask", "sah_probe_trace", "sah_rtgi_trace", "sah_rt_set_rows", "sah_rt_set_bounces", "sah_allgather_rows", "sah_allgather_rows_reversed", "sah_allgather_bytes", "sah_comm_set_stream", "sah_comm_wait",
           "sah_ipc_open", "sah_ipc_connect", "sah_ipc_export", "sah_ipc_register", "sah_ipc_unregister", "sah_ipc_reset",
           "sah_chain_create", "sah_chain_submit", "sah_chain_flush", "sah_chain_counts", "sah_chain_destroy"]


def load():
    """Loads the library (building is a separate, explicit step: python -m androidrenderer_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found — build it with `python -m androidrenderer_amd.build` (needs hipcc); "
                          "there is no CPU fallback for the product path")
    lib = C.CDLL(LIB_PATH)
    lib.sah_abi_version.restype = C.c_int
    lib.sah_status_string.restype = C.c_char_p
    lib.sah_status_string.argtypes = [C.c_int]
    lib.sah_last_error.restype = C.c_char_p
    lib.sah_last_error.argtypes = [C.c_void_p]
    lib.sah_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.sah_destroy.argtypes = [C.c_void_p]
    lib.sah_destroy.restype = None
    lib.sah_comm_unique_id.argtypes = [C.c_void_p]
    lib.sah_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.sah_sync.argtypes = [C.c_void_p]
    lib.sah_debug_deferred_pixels.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.sah_debug_copy_rebuilds.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    lib.sah_lighting.argtypes = [C.c_void_p, C.POINTER(_abi.LightingDesc)]
    lib.sah_copy_scene.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.Plane)]
    lib.sah_copy_scene_rows.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.c_uint32, C.c_uint32]
    lib.sah_bloom.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain)]
    lib.sah_copy_scene_bloom_mip0_rows.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain)] + [C.c_uint32] * 4
    lib.sah_bloom_mip0_rows.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain), C.c_uint32, C.c_uint32]
    lib.sah_bloom_from_mip0.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain)]
    lib.sah_bloom_mip_rows.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain), C.c_uint32, C.c_uint32, C.c_uint32]
    lib.sah_bloom_from_mip.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain), C.c_uint32]
    lib.sah_bloom_source_rows.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)]
    lib.sah_tonemap.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain), C.POINTER(_abi.Plane), C.c_uint32, C.c_uint32]
    lib.sah_tonemap_ex.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.MipChain), C.POINTER(_abi.Plane), C.c_uint32, C.c_uint32, C.c_uint32]
    lib.sah_lpv_clear.argtypes = [C.c_void_p] + [C.POINTER(_abi.Volume)] * 4 + [C.c_uint32]
    lib.sah_lpv_propagate.argtypes = [C.c_void_p, C.POINTER(_abi.Volume), C.POINTER(_abi.Volume), C.c_uint32, C.c_uint32]
    lib.sah_allgather_rows.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.c_uint32, C.c_uint32]
    lib.sah_allgather_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.sah_allgather_rows_reversed.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.c_uint32, C.c_uint32]
    lib.sah_comm_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.sah_comm_wait.argtypes = [C.c_void_p]
    lib.sah_ao_clear.argtypes = [C.c_void_p, C.POINTER(_abi.Plane)]
    lib.sah_sky_update_luts.argtypes = [C.c_void_p, C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.POINTER(C.c_float)]
    lib.sah_probe_copy.argtypes =[C.c_void_p, C.POINTER(_abi.ProbeAtlases), C.POINTER(_abi.ProbeAtlases), C.POINTER(C.c_float * 3)]
    lib.sah_probe_update.argtypes = [C.c_void_p, C.POINTER(_abi.ProbeAtlases), C.POINTER(_abi.Volume), C.c_void_p, C.c_uint32]
    lib.sah_probe_notify_updated.argtypes = [C.c_void_p, C.POINTER(_abi.Volume), C.c_void_p, C.c_uint32]
    lib.sah_shadow_render.argtypes = [C.c_void_p, C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.SunLightConstants), C.c_uint32,
                                      C.POINTER(_abi.Volume), C.c_void_p]
    lib.sah_gbuffer_render.argtypes = [C.c_void_p, C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.ViewData), C.POINTER(_abi.GBuffer), C.c_void_p]
    lib.sah_rsm_render.argtypes = [C.c_void_p, C.POINTER(_abi.SceneGeometry), C.POINTER(_abi.SunLightConstants), C.POINTER(_abi.LpvCascadeMatrices),
                                   C.c_uint32, C.POINTER(_abi.RsmTargets), C.c_void_p]
    lib.sah_lpv_extract_vpls.argtypes = [C.c_void_p, C.POINTER(_abi.RsmTargets), C.POINTER(_abi.LpvCascadeMatrices), C.c_uint32, C.c_float, C.c_void_p,
                                         C.c_void_p]
    lib.sah_lpv_inject_vpls.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(_abi.LpvCascadeMatrices), C.c_uint32, C.c_uint32,
                                        C.POINTER(_abi.Volume)]
    lib.sah_rt_build.argtypes = [C.c_void_p, C.POINTER(_abi.SceneGeometry), C.POINTER(C.c_uint32)]
    lib.sah_rtao.argtypes = [C.c_void_p, C.POINTER(_abi.ViewData), C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.POINTER(_abi.Plane), C.c_uint32, C.c_float,
                             C.POINTER(_abi.Plane)]
    lib.sah_sun_shadow_mask.argtypes = [C.c_void_p, C.POINTER(_abi.ViewData), C.POINTER(_abi.SunLightConstants), C.POINTER(_abi.Plane), C.POINTER(_abi.Plane),
                                        C.POINTER(_abi.Plane), C.POINTER(_abi.Plane)]
    lib.sah_rt_set_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.sah_rt_set_bounces.argtypes = [C.c_void_p, C.c_uint32]
    lib.sah_probe_trace.argtypes = [C.c_void_p, C.POINTER(_abi.ProbeTraceDesc)]
    lib.sah_rtgi_trace.argtypes = [C.c_void_p, C.POINTER(_abi.ViewData), C.POINTER(_abi.SunLightConstants), C.POINTER(_abi.SkyLuts)] + [C.POINTER(_abi.Plane)] * 5
    lib.sah_ipc_open.argtypes = [C.c_void_p, C.c_void_p]
    lib.sah_ipc_connect.argtypes = [C.c_void_p, C.c_void_p]
    lib.sah_ipc_export.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.sah_ipc_register.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.sah_ipc_unregister.argtypes = [C.c_void_p, C.c_void_p]
    lib.sah_ipc_reset.argtypes = [C.c_void_p]
    lib.sah_chain_create.argtypes = [C.c_void_p, C.POINTER(_abi.ChainPlan), C.POINTER(_abi.ChainFrame), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(C.c_void_p)]
    lib.sah_chain_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sah_chain_flush.argtypes = [C.c_void_p]
    lib.sah_chain_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.sah_chain_destroy.argtypes = [C.c_void_p]
    lib.sah_debug_chain_graphs.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.sah_chain_destroy.restype = None
    lib.sah_debug_set.argtypes = [C.c_void_p, C.c_int, C.c_int]
    _lib = lib
    return lib


class Context:
    """One sah_ctx: a device, a stream, optionally an RCCL communicator."""

    def __init__(self, device=0, rank=0, world=1, comm_id=None, stream=None):
        self.lib = load()
        h = C.c_void_p()
        idbuf = None
        if comm_id is not None:
            idbuf = C.create_string_buffer(bytes(comm_id), 128)
        rc = self.lib.sah_create(C.byref(h), device, rank, world, idbuf)
        if rc != _abi.SAH_OK:
            raise SahError(rc, self.lib.sah_status_string(rc).decode())
        self.handle = h
        if stream is not None:
            self.set_stream(stream)

    def _check(self, rc):
        if rc != _abi.SAH_OK:
            msg = self.lib.sah_last_error(self.handle).decode() or self.lib.sah_status_string(rc).decode()
            raise SahError(rc, msg)

    def set_stream(self, hip_stream_handle):
        self._check(self.lib.sah_set_stream(self.handle, C.c_void_p(hip_stream_handle)))

    def debug_set(self, force_general=False, force_ppt=0):
        """Testing hook: run the general kernel instead of the fast one / force pixels-per-thread."""
        self._check(self.lib.sah_debug_set(self.handle, int(force_general), int(force_ppt)))

    def deferred_pixels(self):
        """Analysis hook: pixels the last fast-path lighting call handed to the fix-up kernel (synchronises)."""
        n = C.c_uint64()
        self._check(self.lib.sah_debug_deferred_pixels(self.handle, C.byref(n)))
        return int(n.value)

    def copy_rebuilds(self):
        """Test hook: (full rebuilds of the LPV gather copy, of the fp32 irradiance copy) by lighting() since the context was made."""
        out = (C.c_uint32 * 2)()
        self._check(self.lib.sah_debug_copy_rebuilds(self.handle, out))
        return int(out[0]), int(out[1])

    def sync(self):
        self._check(self.lib.sah_sync(self.handle))

    def lighting(self, desc):
        self._check(self.lib.sah_lighting(self.handle, C.byref(desc)))

    def copy_scene(self, lit, out, row_begin=0, row_end=0):
        self._check(self.lib.sah_copy_scene_rows(self.handle, C.byref(lit), C.byref(out), row_begin, row_end))

    def copy_scene_bloom_mip0(self, lit, out, chain, aa_rows=(0, 0), mip_rows=(0, 0)):
        """Copy scene + bloom mip 0 in one pass over lit (rows of `antialiased`, rows of mip 0; (0, 0) = all)."""
        self._check(self.lib.sah_copy_scene_bloom_mip0_rows(self.handle, C.byref(lit), C.byref(out), C.byref(chain), aa_rows[0], aa_rows[1], mip_rows[0], mip_rows[1]))

    def bloom(self, scene, chain):
        self._check(self.lib.sah_bloom(self.handle, C.byref(scene), C.byref(chain)))

    def bloom_mip0_rows(self, scene, chain, row_begin, row_end):
        self._check(self.lib.sah_bloom_mip0_rows(self.handle, C.byref(scene), C.byref(chain), row_begin, row_end))

    def bloom_mip_rows(self, scene, chain, mip, row_begin, row_end):
        """rows of mip `mip` from its source (the scene for mip 0, mip - 1 otherwise)"""
        self._check(self.lib.sah_bloom_mip_rows(self.handle, C.byref(scene), C.byref(chain), mip, row_begin, row_end))

    def bloom_from_mip(self, scene, chain, mip):
        """mips mip + 1 .. from mip `mip`"""
        self._check(self.lib.sah_bloom_from_mip(self.handle, C.byref(scene), C.byref(chain), mip))

    def bloom_from_mip0(self, scene, chain):
        self._check(self.lib.sah_bloom_from_mip0(self.handle, C.byref(scene), C.byref(chain)))

    def tonemap(self, scene, chain, out, row_begin=0, row_end=0, flags=0):
        """flags: 0 = strict (bit-identical to the oracle), _abi.TONEMAP_TOLERANCE_1CODE = within one code of it, about twice as fast"""
        self._check(self.lib.sah_tonemap_ex(self.handle, C.byref(scene), C.byref(chain), C.byref(out), row_begin, row_end, flags))

    def lpv_clear(self, red, green, blue, geometry, num_cascades):
        null = C.POINTER(_abi.Volume)()
        args = [C.byref(v) if v is not None else null for v in (red, green, blue, geometry)]
        self._check(self.lib.sah_lpv_clear(self.handle, *args, num_cascades))

    def lpv_propagate(self, a_rgb, b_rgb, num_cascades, steps):
        a = (_abi.Volume * 3)(*a_rgb)
        b = (_abi.Volume * 3)(*b_rgb)
        self._check(self.lib.sah_lpv_propagate(self.handle, a, b, num_cascades, steps))

    def sky_update_luts(self, transmittance, multiscattering, sky_view, light_vector):
        lv = (C.c_float * 3)(*[float(v) for v in light_vector])
        self._check(self.lib.sah_sky_update_luts(self.handle, C.byref(transmittance), C.byref(multiscattering), C.byref(sky_view), lv))

    def ao_clear(self, ao):
        self._check(self.lib.sah_ao_clear(self.handle, C.byref(ao)))

    def probe_copy(self, src, dst, cascade_movement):
        """src, dst: _abi.ProbeAtlases; cascade_movement: 4 x 3 floats (probe cells per cascade)."""
        mv = ((C.c_float * 3) * 4)(*[(C.c_float * 3)(*[float(v) for v in row]) for row in cascade_movement])
        self._check(self.lib.sah_probe_copy(self.handle, C.byref(src), C.byref(dst), mv))

    def probe_update(self, atlases, trace_results, probes_to_update_ptr, num_probes):
        """probes_to_update_ptr: device address of num_probes packed uint32 triples."""
        self._check(self.lib.sah_probe_update(self.handle, C.byref(atlases), C.byref(trace_results), C.c_void_p(probes_to_update_ptr), num_probes))

    def probe_notify_updated(self, probe_irradiance, probes_ptr, num_probes):
        """The caller rewrote these probes' blocks of the irradiance atlas itself: a tracked fp32 copy of it is patched (see sah_hip.h)."""
        self._check(self.lib.sah_probe_notify_updated(self.handle, C.byref(probe_irradiance), C.c_void_p(probes_ptr), num_probes))

    def shadow_render(self, scene, sun, num_cascades, shadowmap, stats_ptr=None):
        """scene: _abi.SceneGeometry of device addresses; stats_ptr: device address of 8 uint32 or None."""
        self._check(self.lib.sah_shadow_render(self.handle, C.byref(scene), C.byref(sun), num_cascades, C.byref(shadowmap), C.c_void_p(stats_ptr)))

    def gbuffer_render(self, scene, view, gbuffer, stats_ptr=None):
        self._check(self.lib.sah_gbuffer_render(self.handle, C.byref(scene), C.byref(view), C.byref(gbuffer), C.c_void_p(stats_ptr)))

    def rsm_render(self, scene, sun, cascades, num_cascades, rsm, stats_ptr=None):
        """cascades: (LpvCascadeMatrices * n) host array; rsm: _abi.RsmTargets of device volumes."""
        self._check(self.lib.sah_rsm_render(self.handle, C.byref(scene), C.byref(sun), cascades, num_cascades, C.byref(rsm), C.c_void_p(stats_ptr)))

    def lpv_extract_vpls(self, rsm, cascades, cascade_index, grid_cell_size, vpl_list_ptr, vpl_count_ptr):
        self._check(self.lib.sah_lpv_extract_vpls(self.handle, C.byref(rsm), cascades, cascade_index, grid_cell_size, C.c_void_p(vpl_list_ptr),
                                                  C.c_void_p(vpl_count_ptr)))

    def lpv_inject_vpls(self, vpl_list_ptr, vpl_count_ptr, capacity, cascades, cascade_index, num_cascades, rgb):
        vols = (_abi.Volume * 3)(*rgb)
        self._check(self.lib.sah_lpv_inject_vpls(self.handle, C.c_void_p(vpl_list_ptr), C.c_void_p(vpl_count_ptr), capacity, cascades, cascade_index,
                                                 num_cascades, vols))

    def rt_build(self, scene):
        """(Re)builds the context's acceleration structure over `scene` (_abi.SceneGeometry of device addresses, which must stay alive while
        rays are traced); returns [triangles kept, left out, levels, 0]."""
        stats = (C.c_uint32 * _abi.RT_STATS_WORDS)()
        self._check(self.lib.sah_rt_build(self.handle, C.byref(scene), stats))
        self._rt_scene = scene  # keeps the descriptor's arrays alive
        return list(stats)

    def rtao(self, view, depth, normals, noise, samples_per_pixel, max_ray_distance, ao_out):
        self._check(self.lib.sah_rtao(self.handle, C.byref(view), C.byref(depth), C.byref(normals), C.byref(noise), samples_per_pixel, max_ray_distance,
                                      C.byref(ao_out)))

    def sun_shadow_mask(self, view, sun, depth, normals, noise, mask_out):
        self._check(self.lib.sah_sun_shadow_mask(self.handle, C.byref(view), C.byref(sun), C.byref(depth), C.byref(normals), C.byref(noise), C.byref(mask_out)))

    def rt_set_rows(self, row_begin=0, row_end=0):
        """output rows [row_begin, row_end) for rtao / sun_shadow_mask / rtgi_trace from now on; (0, 0) = all"""
        self._check(self.lib.sah_rt_set_rows(self.handle, row_begin, row_end))

    def rt_set_bounces(self, num_bounces=0):
        """remaining_bounces of the rays probe_trace / rtgi_trace generate from now on (0..2; 0 = what the reference's generators set)"""
        self._check(self.lib.sah_rt_set_bounces(self.handle, num_bounces))

    def probe_trace(self, desc):
        """desc: _abi.ProbeTraceDesc (device addresses); 400 GI rays per probe into desc.trace_results."""
        self._check(self.lib.sah_probe_trace(self.handle, C.byref(desc)))

    def rtgi_trace(self, view, sun, sky, depth, normals, noise, ray_buffer, ray_irradiance):
        self._check(self.lib.sah_rtgi_trace(self.handle, C.byref(view), C.byref(sun), C.byref(sky), C.byref(depth), C.byref(normals), C.byref(noise),
                                            C.byref(ray_buffer), C.byref(ray_irradiance)))

    def allgather_rows(self, image, rows_per_rank, allocated_rows=None):
        """image: _abi.Plane over a buffer of `allocated_rows` (default image.height) rows; in place, on the context's stream."""
        self._check(self.lib.sah_allgather_rows(self.handle, C.byref(image), rows_per_rank, image.height if allocated_rows is None else allocated_rows))

    def allgather_rows_reversed(self, image, rows_per_rank, allocated_rows=None):
        self._check(self.lib.sah_allgather_rows_reversed(self.handle, C.byref(image), rows_per_rank,
                                                         image.height if allocated_rows is None else allocated_rows))

    def comm_set_stream(self, hip_stream_handle):
        self._check(self.lib.sah_comm_set_stream(self.handle, C.c_void_p(hip_stream_handle)))

    def comm_wait(self):
        self._check(self.lib.sah_comm_wait(self.handle))

    # ---- direct exchange (include/sah_hip.h): handles are plain bytes that the caller carries between the ranks
    def ipc_open(self):
        buf = C.create_string_buffer(_abi.IPC_HANDLE_BYTES)
        self._check(self.lib.sah_ipc_open(self.handle, buf))
        return buf.raw

    def ipc_connect(self, handles):
        """handles: every rank's ipc_open() result, in rank order"""
        self._check(self.lib.sah_ipc_connect(self.handle, C.create_string_buffer(b"".join(handles), len(handles) * _abi.IPC_HANDLE_BYTES)))

    def ipc_export(self, device_ptr, nbytes):
        buf = C.create_string_buffer(_abi.IPC_HANDLE_BYTES)
        self._check(self.lib.sah_ipc_export(self.handle, C.c_void_p(device_ptr), nbytes, buf))
        return buf.raw

    def ipc_register(self, device_ptr, nbytes, handles):
        self._check(self.lib.sah_ipc_register(self.handle, C.c_void_p(device_ptr), nbytes,
                                              C.create_string_buffer(b"".join(handles), len(handles) * _abi.IPC_HANDLE_BYTES)))

    def ipc_unregister(self, device_ptr):
        """Before a registered buffer is freed (every rank, same order): waits for the context's streams and frees the registration."""
        self._check(self.lib.sah_ipc_unregister(self.handle, C.c_void_p(device_ptr)))

    def ipc_reset(self):
        """After SahError COMM from the direct exchange, collectively: sync (ignore its error), barrier, ipc_reset, barrier (see sah_hip.h)."""
        self._check(self.lib.sah_ipc_reset(self.handle))

    def allgather_bytes(self, device_ptr, bytes_per_rank):
        self._check(self.lib.sah_allgather_bytes(self.handle, C.c_void_p(device_ptr), bytes_per_rank))

    # ---- the sharded frame as a loop of the library (sah_chain_*): handles are plain integers, androidrenderer_amd/chain.py wraps them
    def chain_create(self, plan, frames, tonemap_flags, chain_flags, work_stream, reduce_stream, post_stream):
        h = C.c_void_p()
        self._check(self.lib.sah_chain_create(self.handle, C.byref(plan), frames, tonemap_flags, chain_flags, C.c_void_p(work_stream),
                                              C.c_void_p(reduce_stream) if reduce_stream else None, C.c_void_p(post_stream) if post_stream else None, C.byref(h)))
        return h

    def chain_submit(self, chain, begin_event=None, end_event=None):
        self._check(self.lib.sah_chain_submit(chain, C.c_void_p(begin_event) if begin_event else None, C.c_void_p(end_event) if end_event else None))

    def chain_flush(self, chain):
        self._check(self.lib.sah_chain_flush(chain))

    def chain_graphs(self, chain):
        """(graph replays, captures, 1 if a capture failed and the chain fell back to direct calls)"""
        out = (C.c_uint64 * 3)()
        self._check(self.lib.sah_debug_chain_graphs(chain, out))
        if out[2]:
            print("[sah] " + self.lib.sah_last_error(self.handle).decode())
        return int(out[0]), int(out[1]), int(out[2])

    def chain_destroy(self, chain):
        self.lib.sah_chain_destroy(chain)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sah_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    lib = load()
    buf = C.create_string_buffer(128)
    rc = lib.sah_comm_unique_id(buf)
    if rc != _abi.SAH_OK:
        raise SahError(rc, lib.sah_status_string(rc).decode())
    return buf.raw
