"""Describe numpy arrays (host) or torch tensors (device) as sah_plane / sah_volume for the C ABI."""
import numpy as np

from . import _abi


def _ptr_and_strides(a):
    """Returns (address, shape, byte strides, keepalive) for a numpy array or a torch tensor."""
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return a.ctypes.data, a.shape, a.strides, a
    # torch tensor
    if not a.is_contiguous():
        raise ValueError("tensor must be contiguous")
    es = a.element_size()
    return a.data_ptr(), tuple(a.shape), tuple(s * es for s in a.stride()), a


def plane(a, fmt):
    """2D image from an array shaped (H, W[, C]); the texel size must equal the format's bytes per pixel."""
    ptr, shape, strides, _ = _ptr_and_strides(a)
    h, w = shape[0], shape[1]
    bpp = _abi.FORMAT_BPP[fmt]
    if strides[1] != bpp:
        raise ValueError(f"texel stride {strides[1]} != {bpp} bytes for format {fmt}")
    return _abi.Plane(ptr, w, h, strides[0], fmt)


def volume(a, fmt):
    """3D image / 2D array from an array shaped (D, H, W[, C])."""
    ptr, shape, strides, _ = _ptr_and_strides(a)
    d, h, w = shape[0], shape[1], shape[2]
    bpp = _abi.FORMAT_BPP[fmt]
    if strides[2] != bpp:
        raise ValueError(f"texel stride {strides[2]} != {bpp} bytes for format {fmt}")
    return _abi.Volume(ptr, w, h, d, strides[1], strides[0], fmt)


def gbuffer(g):
    """dict with color/normals/data/emission/depth arrays → sah_gbuffer."""
    return _abi.GBuffer(
        plane(g["color"], _abi.FORMAT_R8G8B8A8_SRGB),
        plane(g["normals"], _abi.FORMAT_R16G16B16A16_SFLOAT),
        plane(g["data"], _abi.FORMAT_R8G8B8A8_UNORM),
        plane(g["emission"], _abi.FORMAT_R8G8B8A8_SRGB),
        plane(g["depth"], _abi.FORMAT_D32_SFLOAT),
    )


def bloom_mip_sizes(width, height, num_mips=6):
    """Bloom pyramid extents: mip0 = output/2, then Vulkan mip rule max(1, n/2) (bloomer.cpp:268-285)."""
    w, h = width // 2, height // 2
    out = []
    for _ in range(num_mips):
        out.append((max(1, w), max(1, h)))
        w, h = max(1, w // 2), max(1, h // 2)
    return out


def mipchain(mips):
    """list of (H, W, 4) fp16 arrays → sah_mipchain."""
    mc = _abi.MipChain()
    mc.num_mips = len(mips)
    for i, m in enumerate(mips):
        mc.mips[i] = plane(m, _abi.FORMAT_R16G16B16A16_SFLOAT)
    return mc
