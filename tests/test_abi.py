"""CPU-only: the C-ABI library loads, exports every symbol include/sah_hip.h declares, its structs have the reference's
sizes, and it fails loudly (status codes, never a CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

from androidrenderer_amd import _abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "sah_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sah_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    l = lib.load()
    declared = _declared_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(l, name), f"libsah_hip.so does not export {name}"
    assert sorted(lib.EXPORTS) == declared


def test_abi_version_and_status_strings():
    l = lib.load()
    assert l.sah_abi_version() == 6  # 6: SAH_GENERATION_TRACKED (the context keeps its gather copies current), sah_chain_*
    assert l.sah_status_string(0) == b"ok"
    assert l.sah_status_string(_abi.SAH_ERR_NO_DEVICE) == b"no HIP device"


def test_struct_sizes_match_reference_layouts():
    # RenderCore/shared/view_data.hpp:6-40, sun_light_constants.hpp:10-44, lpv.hpp:6-11, gi_probe.hpp:5-8
    assert C.sizeof(_abi.ViewData) == 432
    assert C.sizeof(_abi.SunLightConstants) == 640
    assert C.sizeof(_abi.LpvCascadeMatrices) == 256
    assert C.sizeof(_abi.ProbeCascade) == 16
    assert C.sizeof(_abi.PointLight) == 32
    assert C.sizeof(_abi.Plane) == 24
    assert C.sizeof(_abi.Volume) == 32


def test_invalid_arguments_are_status_codes():
    l = lib.load()
    h = C.c_void_p()
    assert l.sah_create(None, 0, 0, 1, None) == _abi.SAH_ERR_INVALID_ARGUMENT
    assert l.sah_create(C.byref(h), 0, 3, 2, None) == _abi.SAH_ERR_INVALID_ARGUMENT
    assert l.sah_lighting(None, None) == _abi.SAH_ERR_INVALID_ARGUMENT
    assert l.sah_sync(None) == _abi.SAH_ERR_INVALID_ARGUMENT


def test_no_gpu_means_no_context_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.SahError) as e:
        lib.Context()
    assert e.value.status == _abi.SAH_ERR_NO_DEVICE


def test_product_path_does_not_reference_the_oracle():
    """The library and the package must not import / link / execute anything under oracle/."""
    pkg = os.path.join(ROOT, "androidrenderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle/" not in text.replace("the oracle/", ""), f"{f} mentions the oracle"
