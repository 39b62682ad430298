"""Camera / transform math of the boundary, restated from the reference's call sites in float32.

The reference builds its matrices with glam 0.29.2 (Cargo.lock:1640-1643):
  - ``CameraOrbitControl::view``  = ``Mat4::look_at_rh(pos, target, Vec3::Y)``        src/app.rs:1237-1239
  - ``CameraOrbitControl::projection`` = ``Mat4::perspective_rh(fovy, aspect, near, far)``  src/app.rs:1241-1243
  - first-person ``gs::Camera {pos, z, vertical_fov, pitch, yaw}``                    src/app.rs:1247, 1293-1300
  - model rotation ``Quat::from_euler(EulerRot::ZYX, rot.z, rot.y, rot.x)`` in degrees  src/app.rs:1123-1130
All matrices are column-major 4x4 stored as flat float32[16] (``Mat4::to_cols_array``), which is what
``gsx_update_camera`` takes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

f32 = np.float32


def _v3(v) -> np.ndarray:
    return np.asarray(v, dtype=f32).reshape(3)


def _normalize(v: np.ndarray) -> np.ndarray:
    return (v / f32(np.sqrt(np.dot(v, v), dtype=f32))).astype(f32)


def look_to_rh(eye, direction, up) -> np.ndarray:
    """glam ``Mat4::look_to_rh``; returns column-major float32[16]."""
    eye, direction, up = _v3(eye), _v3(direction), _v3(up)
    f = _normalize(direction)
    s = _normalize(np.cross(f, up).astype(f32))
    u = np.cross(s, f).astype(f32)
    m = np.zeros((4, 4), dtype=f32)  # m[col][row]
    m[0] = (s[0], u[0], -f[0], 0.0)
    m[1] = (s[1], u[1], -f[1], 0.0)
    m[2] = (s[2], u[2], -f[2], 0.0)
    m[3] = (-np.dot(eye, s), -np.dot(eye, u), np.dot(eye, f), 1.0)
    return m.reshape(16)


def look_at_rh(eye, center, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """glam ``Mat4::look_at_rh(eye, center, up)`` (src/app.rs:1238)."""
    return look_to_rh(eye, _v3(center) - _v3(eye), up)


def perspective_rh(fov_y: float, aspect: float, z_near: float, z_far: float) -> np.ndarray:
    """glam ``Mat4::perspective_rh``: right-handed, NDC depth in [0, 1] (src/app.rs:1242)."""
    fov_y, aspect, z_near, z_far = f32(fov_y), f32(aspect), f32(z_near), f32(z_far)
    sin_fov, cos_fov = f32(math.sin(0.5 * float(fov_y))), f32(math.cos(0.5 * float(fov_y)))
    h = f32(cos_fov / sin_fov)
    w = f32(h / aspect)
    r = f32(z_far / (z_near - z_far))
    m = np.zeros((4, 4), dtype=f32)
    m[0, 0] = w
    m[1, 1] = h
    m[2, 2] = r
    m[2, 3] = -1.0
    m[3, 2] = f32(r * z_near)
    return m.reshape(16)


def quat_from_euler_zyx(z: float, y: float, x: float) -> np.ndarray:
    """glam ``Quat::from_euler(EulerRot::ZYX, z, y, x)`` = Rz(z) * Ry(y) * Rx(x); returns x,y,z,w."""

    def axis(ax, ang):
        s, c = math.sin(0.5 * ang), math.cos(0.5 * ang)
        q = [0.0, 0.0, 0.0, c]
        q[ax] = s
        return q

    def mul(a, b):
        ax, ay, az, aw = a
        bx, by, bz, bw = b
        return [
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz,
        ]

    return np.asarray(mul(mul(axis(2, z), axis(1, y)), axis(0, x)), dtype=f32)


def quat_rotate(q, v) -> np.ndarray:
    q = np.asarray(q, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    u, w = q[:3], q[3]
    return (2.0 * np.dot(u, v) * u + (w * w - np.dot(u, u)) * v + 2.0 * w * np.cross(u, v)).astype(f32)


@dataclass
class CameraOrbitControl:
    """The app's orbit camera (src/app.rs:1208-1244); implements ``gs::CameraTrait``."""

    target: np.ndarray = field(default_factory=lambda: np.zeros(3, f32))
    pos: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -1.0], f32))  # Vec3::NEG_Z, app.rs:1192
    z: tuple = (0.1, 1e4)  # app.rs:1193
    vertical_fov: float = math.radians(60.0)  # app.rs:1194

    def view(self) -> np.ndarray:
        return look_at_rh(self.pos, self.target, (0.0, 1.0, 0.0))

    def projection(self, aspect_ratio: float) -> np.ndarray:
        return perspective_rh(self.vertical_fov, aspect_ratio, self.z[0], self.z[1])


@dataclass
class Camera:
    """``gs::Camera`` first-person control {pos, z, vertical_fov, pitch, yaw} (src/app.rs:1247, 1293-1300).

    forward = (sin(yaw) cos(pitch), sin(pitch), cos(yaw) cos(pitch)), the inverse of the app's
    ``yaw = atan2(dir.x, dir.z)``, ``pitch = asin(dir.y)`` (app.rs:1298-1299)."""

    pos: np.ndarray = field(default_factory=lambda: np.zeros(3, f32))
    z: tuple = (0.1, 1e4)
    vertical_fov: float = math.radians(60.0)
    pitch: float = 0.0
    yaw: float = 0.0

    def get_forward(self) -> np.ndarray:
        cp = math.cos(self.pitch)
        return np.array([math.sin(self.yaw) * cp, math.sin(self.pitch), math.cos(self.yaw) * cp], dtype=f32)

    def get_right(self) -> np.ndarray:
        return _normalize(np.cross(self.get_forward(), np.array([0, 1, 0], f32)).astype(f32))

    def yaw_by(self, delta: float) -> None:
        self.yaw = (self.yaw + delta) % (2.0 * math.pi)

    def pitch_by(self, delta: float) -> None:
        lim = math.pi / 2 - 1e-6
        self.pitch = min(max(self.pitch + delta, -lim), lim)

    def view(self) -> np.ndarray:
        return look_to_rh(self.pos, self.get_forward(), (0.0, 1.0, 0.0))

    def projection(self, aspect_ratio: float) -> np.ndarray:
        return perspective_rh(self.vertical_fov, aspect_ratio, self.z[0], self.z[1])


@dataclass
class ModelTransform:
    """``GaussianSplattingModelTransform`` {pos, rot (Euler degrees), scale} (src/app.rs:1100-1131)."""

    pos: np.ndarray = field(default_factory=lambda: np.zeros(3, f32))
    rot: np.ndarray = field(default_factory=lambda: np.zeros(3, f32))
    scale: np.ndarray = field(default_factory=lambda: np.ones(3, f32))

    def quat(self) -> np.ndarray:
        r = np.radians(np.asarray(self.rot, dtype=np.float64))
        return quat_from_euler_zyx(r[2], r[1], r[0])

    def world_center(self, center=(0.0, 0.0, 0.0)) -> np.ndarray:
        """``quat * (center * scale) + pos`` (src/app.rs:1044-1046)."""
        return quat_rotate(self.quat(), _v3(center) * _v3(self.scale)) + _v3(self.pos)


def orbit_pose(index: int, poses: int = 240, radius: float = 6.0, height: float = 1.5) -> CameraOrbitControl:
    """Benchmark camera path: orbit radius 6 about the origin at height 1.5, 240 poses."""
    ang = 2.0 * math.pi * (index % poses) / poses
    pos = np.array([radius * math.sin(ang), height, -radius * math.cos(ang)], dtype=f32)
    return CameraOrbitControl(target=np.zeros(3, f32), pos=pos)


class PrecomputedCamera:
    """A camera whose view / projection matrices were evaluated once (``CameraTrait`` as the viewer sees it: two
    matrices).  The frame loop of bench.py prepares the orbit's 240 cameras before it starts the clock."""

    def __init__(self, cam, aspect_ratio: float):
        self.pos = np.asarray(cam.pos, f32)
        self._view = np.ascontiguousarray(cam.view(), f32)
        self._proj = np.ascontiguousarray(cam.projection(aspect_ratio), f32)
        self._aspect = float(aspect_ratio)

    def view(self) -> np.ndarray:
        return self._view

    def projection(self, aspect_ratio: float) -> np.ndarray:
        if abs(float(aspect_ratio) - self._aspect) > 1e-12:
            raise ValueError("PrecomputedCamera was built for another aspect ratio")
        return self._proj


def model_render_order(camera_pos, centers: dict) -> list:
    """Keys far -> near by squared distance of ``world_center`` to the camera (src/tab/scene.rs:533-558)."""
    cam = _v3(camera_pos).astype(np.float32)
    dist = {k: float(np.sum((_v3(c) - cam) ** 2, dtype=f32)) for k, c in centers.items()}
    # `sorted_by(|a, b| dist[b].partial_cmp(dist[a]))` is a stable descending sort
    return [k for k, _ in sorted(dist.items(), key=lambda kv: -kv[1])]
