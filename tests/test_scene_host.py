"""CPU-only: host-side producers of the uniform blocks (androidrenderer_amd/scene.py mirrors SceneView / DirectionalLight /
LPV cascade transforms) and the synthetic-input generators."""
import math

import numpy as np

from androidrenderer_amd import _abi, images, scene, synth


def _m(a):
    return np.array(a[:], dtype=np.float64).reshape(4, 4).T  # column-major float[16] -> maths matrix


def test_scene_view_matrices_are_consistent():
    v = scene.SceneView.default(1280, 720)
    view, inv_view = _m(v.gpu_data.view), _m(v.gpu_data.inverse_view)
    proj, inv_proj = _m(v.gpu_data.projection), _m(v.gpu_data.inverse_projection)
    assert np.allclose(view @ inv_view, np.eye(4), atol=1e-5)
    assert np.allclose(proj @ inv_proj, np.eye(4), atol=1e-5)
    # infinite reversed-Z (scene_view.cpp:13-27): depth = z_near / -z_view, 0 at infinity
    p = proj @ np.array([0.3, -0.2, -5.0, 1.0])
    assert math.isclose(p[2] / p[3], 0.05 / 5.0, rel_tol=1e-6)
    assert (v.gpu_data.render_resolution[0], v.gpu_data.render_resolution[1]) == (1280.0, 720.0)
    # start-up camera (scene_renderer.cpp:53-54): at (-7,1,0) looking down +x
    assert np.allclose(v.forward, [1, 0, 0], atol=1e-6)
    assert np.allclose(inv_view[:3, 3], [-7, 1, 0], atol=1e-6)


def test_reference_uniform_blocks_have_the_structure_the_fast_kernel_detects():
    """inverse_projection separable, inverse_view affine, LPV cascades scale+translate, CSM matrices affine."""
    v = scene.SceneView.default(3840, 2160)
    P = np.array(v.gpu_data.inverse_projection[:])
    assert all(P[i] == 0 for i in (1, 2, 3, 4, 6, 7, 8, 9))
    V = np.array(v.gpu_data.inverse_view[:])
    assert V[3] == 0 and V[7] == 0 and V[11] == 0 and V[15] == 1
    sun = scene.DirectionalLight(shadow_mode=_abi.SHADOW_MODE_CSM)
    sun.update_shadow_cascades(v)
    for c in range(4):
        m = np.array(sun.constants.cascade_matrices[c][:])
        assert m[3] == 0 and m[7] == 0 and m[11] == 0 and m[15] == 1
    assert [sun.constants.data[i][0] for i in range(4)] == sorted([sun.constants.data[i][0] for i in range(4)], reverse=True)
    lpv = scene.LpvCascades()
    lpv.update_cascade_transforms(v, sun)
    for c in range(4):
        m = np.array(lpv.matrices[c].world_to_cascade[:])
        assert all(m[i] == 0 for i in (1, 2, 3, 4, 6, 7, 8, 9, 11)) and m[15] == 1
        assert math.isclose(m[0], 0.5 / (32 * 0.25 * 2 ** c), rel_tol=1e-6)  # bias 0.5 * 1 / cascade size


def test_sun_defaults():
    s = scene.DirectionalLight()
    d = np.array(s.constants.direction_and_tan_size[:3])
    assert np.allclose(d, np.array([0.1, -1, -1]) / np.linalg.norm([0.1, -1, -1]), atol=1e-7)
    assert math.isclose(s.constants.direction_and_tan_size[3], math.tan(math.radians(0.545)), rel_tol=1e-6)
    assert list(s.constants.color) == [80000.0, 80000.0, 80000.0, 0.0]
    assert s.constants.shadow_mode == _abi.SHADOW_MODE_RT and s.constants.num_shadow_samples == 8.0


def test_synthetic_gbuffers_are_deterministic_and_in_reference_formats():
    a, b = synth.random_gbuffer(64, 32, seed=3), synth.random_gbuffer(64, 32, seed=3)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert a["color"].dtype == np.uint8 and a["normals"].dtype == np.float16 and a["depth"].dtype == np.float32
    assert np.all(a["data"][..., 0] == 0) and np.all(a["data"][..., 3] == 0)
    sky = (a["depth"] == 0).mean()
    assert 0.03 < sky < 0.2
    v = scene.SceneView.default(96, 54)
    g = synth.atrium_gbuffer(96, 54, v)
    assert g["depth"].shape == (54, 96) and (g["depth"] == 0).mean() < 0.3
    gb = images.gbuffer(g)
    assert gb.normals.row_pitch_bytes == 96 * 8 and gb.depth.format == _abi.FORMAT_D32_SFLOAT


def test_bloom_mip_sizes():
    assert images.bloom_mip_sizes(3840, 2160) == [(1920, 1080), (960, 540), (480, 270), (240, 135), (120, 67), (60, 33)]
    assert images.bloom_mip_sizes(8, 4) == [(4, 2), (2, 1), (1, 1), (1, 1), (1, 1), (1, 1)]
