"""Host-side producers of the uniform blocks the hot path consumes, restated in numpy float32.

These mirror the reference's scene/material plumbing that sits *in front of* the path (SURVEY.md §8 "types"):
  SceneView                 RenderCore/render/scene_view.cpp:13-27,138-187
  DirectionalLight          RenderCore/render/directional_light.cpp:84-260, render_scene.cpp:25-27
  LightPropagationVolume    RenderCore/render/gi/light_propagation_volume.cpp:455-546 (cascade transforms)
They only produce inputs (matrices, constants); all per-pixel work happens behind the C ABI.
Matrices are glm-style column-major: a (4,4) numpy array `m` is indexed m[col, row] and flattens to the
float[16] the ABI expects.
"""
import math

import numpy as np

from . import _abi

f32 = np.float32


def _v3(x):
    return np.asarray(x, dtype=f32).reshape(3)


def normalize(v):
    v = _v3(v)
    return (v * (f32(1.0) / np.sqrt(np.dot(v, v), dtype=f32))).astype(f32)


def mat_mul(a, b):
    """glm `a * b` for column-major (col,row)-indexed arrays."""
    # result[col j][row i] = sum_k a[k][i] * b[j][k]
    return np.einsum("ki,jk->ji", a.astype(f32), b.astype(f32)).astype(f32)


def mat_vec(m, v):
    return np.einsum("ki,k->i", m.astype(f32), np.asarray(v, dtype=f32)).astype(f32)


def mat_inverse(m):
    """glm::inverse stand-in. Evaluated in float64 and rounded to fp32 (these are inputs to the path)."""
    # (col,row) indexing means m as a numpy matrix is the transpose of the maths matrix; inverse commutes.
    return np.linalg.inv(m.astype(np.float64)).astype(f32)


def look_at(eye, center, up):
    """glm::lookAt (right-handed)."""
    eye, center, up = _v3(eye), _v3(center), _v3(up)
    f = normalize(center - eye)
    s = normalize(np.cross(f, up).astype(f32))
    u = np.cross(s, f).astype(f32)
    m = np.zeros((4, 4), dtype=f32)
    m[0, 0], m[1, 0], m[2, 0] = s
    m[0, 1], m[1, 1], m[2, 1] = u
    m[0, 2], m[1, 2], m[2, 2] = -f
    m[3, 0] = -np.dot(s, eye)
    m[3, 1] = -np.dot(u, eye)
    m[3, 2] = np.dot(f, eye)
    m[3, 3] = 1.0
    return m


def ortho(left, right, bottom, top, z_near, z_far):
    """glm::ortho with GLM_FORCE_DEPTH_ZERO_TO_ONE (Vulkan clip space), right-handed."""
    m = np.zeros((4, 4), dtype=f32)
    m[0, 0] = f32(2.0) / f32(right - left)
    m[1, 1] = f32(2.0) / f32(top - bottom)
    m[2, 2] = -f32(1.0) / f32(z_far - z_near)
    m[3, 0] = -f32(right + left) / f32(right - left)
    m[3, 1] = -f32(top + bottom) / f32(top - bottom)
    m[3, 2] = -f32(z_near) / f32(z_far - z_near)
    m[3, 3] = 1.0
    return m


def perspective_fov(fov, width, height, z_near, z_far):
    """glm::perspectiveFov (RH, zero-to-one)."""
    h = f32(math.cos(0.5 * fov) / math.sin(0.5 * fov))
    w = f32(h * f32(height) / f32(width))
    m = np.zeros((4, 4), dtype=f32)
    m[0, 0] = w
    m[1, 1] = h
    m[2, 2] = f32(z_far) / f32(z_near - z_far)
    m[2, 3] = -1.0
    m[3, 2] = -f32(z_far * z_near) / f32(z_far - z_near)
    return m


def inf_depth_reverse_z_perspective(fov_rads, aspect, z_near):
    """scene_view.cpp:13-27."""
    t = f32(1.0) / f32(math.tan(f32(fov_rads) * f32(0.5)))
    m = np.zeros((4, 4), dtype=f32)
    m[0, 0] = t / f32(aspect)
    m[1, 1] = t
    m[2, 3] = -1.0
    m[3, 2] = z_near
    return m


def _fill(dst, arr):
    flat = np.asarray(arr, dtype=f32).reshape(-1)
    for i, v in enumerate(flat):
        dst[i] = float(v)


class SceneView:
    """scene_view.hpp:17-125 — camera → ViewDataGPU."""

    def __init__(self):
        self.fov = 75.0
        self.aspect = 16.0 / 9.0
        self.near_value = 0.05
        self.position = np.zeros(3, dtype=f32)
        self.pitch = 0.0
        self.yaw = 0.0
        self.forward = np.zeros(3, dtype=f32)
        self.jitter = np.zeros(2, dtype=f32)
        self.gpu_data = _abi.ViewData()
        self.view = np.zeros((4, 4), dtype=f32)
        self.projection = np.zeros((4, 4), dtype=f32)

    def set_render_resolution(self, w, h):
        self.gpu_data.render_resolution[0] = float(w)
        self.gpu_data.render_resolution[1] = float(h)

    def set_position(self, p):
        self.position = _v3(p)

    def rotate(self, delta_pitch, delta_yaw):
        self.pitch += delta_pitch
        self.yaw += delta_yaw

    def set_perspective_projection(self, fov, aspect, near_value):
        self.fov, self.aspect, self.near_value = fov, aspect, near_value

    def update_transforms(self):
        # refresh_view_matrices, scene_view.cpp:138-148
        p, y = f32(self.pitch), f32(self.yaw)
        self.forward = np.array([math.cos(p) * math.sin(y), math.sin(p), math.cos(p) * math.cos(y)], dtype=f32)
        right = np.array([math.sin(y - math.pi / 2.0), 0.0, math.cos(y - math.pi / 2.0)], dtype=f32)
        up = np.cross(right, self.forward).astype(f32)
        _fill(self.gpu_data.last_frame_view, np.array(self.gpu_data.view[:], dtype=f32))
        self.view = look_at(self.position, self.position + self.forward, up)
        inv_view = mat_inverse(self.view)
        _fill(self.gpu_data.view, self.view)
        _fill(self.gpu_data.inverse_view, inv_view)
        # refresh_projection_matrices, scene_view.cpp:150-187
        _fill(self.gpu_data.last_frame_projection, np.array(self.gpu_data.projection[:], dtype=f32))
        self.projection = inf_depth_reverse_z_perspective(math.radians(self.fov), self.aspect, self.near_value)
        proj = self.projection.copy()
        proj[2, 0] += self.jitter[0] * f32(2.0) / f32(self.gpu_data.render_resolution[0])
        proj[2, 1] += self.jitter[1] * f32(2.0) / f32(self.gpu_data.render_resolution[1])
        _fill(self.gpu_data.projection, proj)
        _fill(self.gpu_data.inverse_projection, mat_inverse(proj))
        pt = proj.T  # glm::transpose
        fx = pt[3] + pt[0]
        fy = pt[3] + pt[1]
        fx = fx / np.linalg.norm(fx[:3])
        fy = fy / np.linalg.norm(fy[:3])
        self.gpu_data.frustum[0], self.gpu_data.frustum[1] = float(fx[0]), float(fx[2])
        self.gpu_data.frustum[2], self.gpu_data.frustum[3] = float(fy[1]), float(fy[2])
        self.gpu_data.z_near = self.near_value
        return self.gpu_data

    @staticmethod
    def default(width, height):
        """The reference's start-up camera: scene_renderer.cpp:53-54,105-116."""
        v = SceneView()
        v.rotate(0.0, math.radians(90.0))
        v.set_position([-7.0, 1.0, 0.0])
        v.set_render_resolution(width, height)
        v.set_perspective_projection(75.0, float(width) / float(height), 0.05)
        v.update_transforms()
        return v


class DirectionalLight:
    """directional_light.cpp:84-260 — SunLightConstants producer."""

    def __init__(self, shadow_mode=_abi.SHADOW_MODE_RT, num_shadow_samples=8.0):
        self.constants = _abi.SunLightConstants()
        self.angular_size = 0.545
        self.set_direction([0.1, -1.0, -1.0])  # render_scene.cpp:25
        self.set_color([80000.0, 80000.0, 80000.0, 0.0])  # render_scene.cpp:27
        self.constants.shadow_mode = shadow_mode
        self.constants.num_shadow_samples = num_shadow_samples

    def set_direction(self, d):
        n = normalize(d)
        for i in range(3):
            self.constants.direction_and_tan_size[i] = float(n[i])
        self.constants.direction_and_tan_size[3] = float(f32(math.tan(math.radians(self.angular_size))))

    def set_color(self, c):
        for i in range(4):
            self.constants.color[i] = float(c[i])

    def update_shadow_cascades(self, view, num_cascades=4, max_shadow_distance=128.0, split_lambda=0.95, resolution=4096):
        """directional_light.cpp:84-230 (CSM cascade fitting)."""
        z_near = f32(view.near_value)
        clip_range = z_near + f32(max_shadow_distance)
        ratio = clip_range / z_near
        splits = []
        for i in range(num_cascades):
            p = f32(i + 1) / f32(num_cascades)
            log = z_near * f32(math.pow(ratio, p))
            uniform = z_near + f32(max_shadow_distance) * p
            d = f32(split_lambda) * (log - uniform) + uniform
            splits.append(f32((d - z_near) / clip_range))
        last = z_near
        light_dir = normalize(self.constants.direction_and_tan_size[0:3])
        for i in range(num_cascades):
            split = splits[i]
            corners = np.array([[-1, 1, -1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, 1], [1, 1, 1], [1, -1, 1], [-1, -1, 1]],
                               dtype=f32)
            proj = perspective_fov(view.fov, view.aspect, 1.0, last * f32(max_shadow_distance), split * f32(max_shadow_distance))
            inv_cam = mat_inverse(mat_mul(proj, view.view))
            world = []
            for c in corners:
                t = mat_vec(inv_cam, [c[0], c[1], c[2], 1.0])
                world.append(t[:3] / t[3])
            world = np.array(world, dtype=f32)
            center = (world.sum(axis=0) / f32(8)).astype(f32)
            radius = f32(max(np.linalg.norm(w - center) for w in world))
            radius = f32(radius * 2)
            radius = f32(math.ceil(radius * 16.0) / 16.0)
            lv = look_at(center - light_dir * radius, center, [0.0, 1.0, 0.0])
            lp = ortho(-radius, radius, -radius, radius, 0.0, radius + radius)
            m = mat_mul(lp, lv)
            for k in range(4):
                self.constants.data[i][k] = 0.0
            self.constants.data[i][0] = float(f32(split * clip_range * f32(-1)))
            _fill(self.constants.cascade_matrices[i], m)
            _fill(self.constants.cascade_inverse_matrices[i], mat_inverse(m))
            last = splits[i]
        self.constants.csm_resolution[0] = resolution
        self.constants.csm_resolution[1] = resolution
        return self.constants


class LpvCascades:
    """light_propagation_volume.cpp:455-546 — world_to_cascade / rsm matrices for the LPV cascades."""

    def __init__(self, num_cells=32, base_cell_size=0.25, num_cascades=4, behind_camera_percent=0.1):
        self.num_cells, self.base_cell_size, self.num_cascades = num_cells, base_cell_size, num_cascades
        self.behind = behind_camera_percent
        self.matrices = (_abi.LpvCascadeMatrices * num_cascades)()

    def update_cascade_transforms(self, view, light):
        offset_scale = f32(0.5) - f32(self.behind)
        bias = np.array([[0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 0.5, 0], [0.5, 0.5, 0.5, 1.0]], dtype=f32)  # columns
        ldir = _v3(light.constants.direction_and_tan_size[0:3])
        for ci in range(self.num_cascades):
            cell = f32(self.base_cell_size) * f32(2.0 ** ci)
            size = f32(self.num_cells) * cell
            offset = view.position + view.forward * (size * offset_scale)
            q = (offset / (cell * f32(2))).astype(f32)
            rounded = (np.sign(q) * np.floor(np.abs(q) + f32(0.5))).astype(f32)  # glm::round: halves away from zero (np.round goes to even)
            snapped = (rounded * cell * f32(2)).astype(f32)
            scale = f32(1.0) / size
            w2c = np.eye(4, dtype=f32)
            w2c[0, 0] = w2c[1, 1] = w2c[2, 2] = scale  # glm::scale(I, s)
            tr = np.eye(4, dtype=f32)
            tr[3, :3] = -snapped
            w2c = mat_mul(w2c, tr)  # glm::translate(m, v) = m * T(v)
            w2c = mat_mul(bias, w2c)
            half = size / f32(2)
            pull = size * f32(2)
            rsm_view = look_at(snapped - ldir * pull, snapped, [0.0, 1.0, 0.0])
            rsm_proj = ortho(-half, half, -half, half, 0.0, pull * f32(2))
            rsm_vp = mat_mul(rsm_proj, rsm_view)
            m = self.matrices[ci]
            _fill(m.rsm_vp, rsm_vp)
            _fill(m.inverse_rsm_vp, mat_inverse(rsm_vp))
            _fill(m.world_to_cascade, w2c)
            _fill(m.cascade_to_world, mat_inverse(w2c))
        return self.matrices
