"""Deterministic synthetic LiDAR / depth-camera scans of an analytic scene.

Scene (seed 0x611D): ground plane z=0, a 60 x 40 x 8 m room (4 walls + ceiling), 12 axis-aligned boxes and
8 vertical cylinders at seeded positions.  Sensor rays are intersected analytically; Gaussian range
noise sigma = 0.01 m seeded per frame (seed + frame_id).  All coordinates are rounded to float32 so that
the FP64 CPU oracle and the HIP kernels consume bit-identical inputs.

This is host-side workload generation for tests and bench.py (numpy only); it is not on the hot path.
"""
import math
import os

import numpy as np

SCENE_SEED = 0x611D


class Scene:
    def __init__(self, half_x, half_y, height, boxes, cylinders):
        self.half_x, self.half_y, self.height = float(half_x), float(half_y), float(height)
        self.boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 6)  # xmin ymin zmin xmax ymax zmax
        self.cylinders = np.asarray(cylinders, dtype=np.float64).reshape(-1, 4)  # cx cy r h

    @staticmethod
    def default(seed=SCENE_SEED):
        rng = np.random.default_rng(seed)
        boxes = []
        for _ in range(12):
            cx, cy = rng.uniform(-26, 26), rng.uniform(-17, 17)
            sx, sy, sz = rng.uniform(1.0, 4.0), rng.uniform(1.0, 4.0), rng.uniform(0.8, 4.0)
            boxes.append([cx - sx / 2, cy - sy / 2, 0.0, cx + sx / 2, cy + sy / 2, sz])
        cyls = []
        for _ in range(8):
            cyls.append([rng.uniform(-26, 26), rng.uniform(-17, 17), rng.uniform(0.3, 0.9), rng.uniform(2.0, 7.0)])
        return Scene(30.0, 20.0, 8.0, boxes, cyls)

    @staticmethod
    def small_room(seed=SCENE_SEED + 1):
        """8 x 6 x 3 m indoor sub-scene for the dense depth-camera stream (everything within ~6 m)."""
        rng = np.random.default_rng(seed)
        boxes = []
        for _ in range(6):
            cx, cy = rng.uniform(-3.2, 3.2), rng.uniform(-2.3, 2.3)
            sx, sy, sz = rng.uniform(0.4, 1.2), rng.uniform(0.4, 1.2), rng.uniform(0.4, 1.8)
            boxes.append([cx - sx / 2, cy - sy / 2, 0.0, cx + sx / 2, cy + sy / 2, sz])
        cyls = [[rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(0.1, 0.25), rng.uniform(1.0, 2.5)] for _ in range(3)]
        return Scene(4.0, 3.0, 3.0, boxes, cyls)

    def raycast(self, origin, dirs):
        """Distance along each unit ray to the first surface (inf if none).  origin: 3, dirs: N x 3 (world)."""
        o = np.asarray(origin, dtype=np.float64).reshape(3)
        d = np.asarray(dirs, dtype=np.float64).reshape(-1, 3)
        n = d.shape[0]
        best = np.full(n, np.inf)
        eps = 1e-9

        def axis_plane(axis, value, lo_a, hi_a, a_idx, lo_b, hi_b, b_idx):
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (value - o[axis]) / d[:, axis]
            ha = o[a_idx] + t * d[:, a_idx]
            hb = o[b_idx] + t * d[:, b_idx]
            ok = (t > eps) & (ha >= lo_a) & (ha <= hi_a) & (hb >= lo_b) & (hb <= hi_b)
            np.minimum(best, np.where(ok, t, np.inf), out=best)

        hx, hy, hz = self.half_x, self.half_y, self.height
        axis_plane(2, 0.0, -hx, hx, 0, -hy, hy, 1)  # floor
        axis_plane(2, hz, -hx, hx, 0, -hy, hy, 1)  # ceiling
        axis_plane(0, -hx, -hy, hy, 1, 0.0, hz, 2)
        axis_plane(0, hx, -hy, hy, 1, 0.0, hz, 2)
        axis_plane(1, -hy, -hx, hx, 0, 0.0, hz, 2)
        axis_plane(1, hy, -hx, hx, 0, 0.0, hz, 2)

        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
        for b in self.boxes:
            t0 = (b[:3] - o) * inv
            t1 = (b[3:] - o) * inv
            tn = np.nanmax(np.minimum(t0, t1), axis=1)
            tf = np.nanmin(np.maximum(t0, t1), axis=1)
            ok = (tf >= tn) & (tn > eps)
            np.minimum(best, np.where(ok, tn, np.inf), out=best)

        for cx, cy, r, h in self.cylinders:
            ox, oy = o[0] - cx, o[1] - cy
            a = d[:, 0] ** 2 + d[:, 1] ** 2
            bq = 2.0 * (ox * d[:, 0] + oy * d[:, 1])
            c = ox * ox + oy * oy - r * r
            disc = bq * bq - 4.0 * a * c
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (-bq - np.sqrt(np.maximum(disc, 0.0))) / (2.0 * a)
            z = o[2] + t * d[:, 2]
            ok = (disc > 0.0) & (a > 1e-12) & (t > eps) & (z >= 0.0) & (z <= h)
            np.minimum(best, np.where(ok, t, np.inf), out=best)
        return best


def lidar_directions(rings, azimuths, vfov_deg=22.5):
    """Unit ray directions (sensor frame), ring-major: index = ring * azimuths + az."""
    el = np.deg2rad(np.linspace(-vfov_deg, vfov_deg, rings))
    az = np.arange(azimuths) * (2.0 * math.pi / azimuths)
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (rings, azimuths))], axis=-1)
    return d.reshape(-1, 3)


def pinhole_directions(width, height, hfov_deg, vfov_deg):
    """Unit ray directions for a depth camera looking along +x (sensor frame), row-major pixels."""
    tx, ty = math.tan(math.radians(hfov_deg) / 2), math.tan(math.radians(vfov_deg) / 2)
    u = (np.arange(width) + 0.5) / width * 2 - 1
    v = (np.arange(height) + 0.5) / height * 2 - 1
    uu, vv = np.meshgrid(u, v)
    d = np.stack([np.ones_like(uu), -uu * tx, -vv * ty], axis=-1).reshape(-1, 3)
    return d / np.linalg.norm(d, axis=1, keepdims=True)


def pose(x, y, z, yaw=0.0, pitch=0.0, roll=0.0):
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [x, y, z]
    return T


def scan(scene, T_world_sensor, dirs, frame_id=0, sigma=0.01, seed=SCENE_SEED, max_range=np.inf, min_range=0.0):
    """One scan in the SENSOR frame: N x 3 float32 (rays without a return inside [min_range, max_range] are dropped)."""
    T = np.asarray(T_world_sensor, dtype=np.float64)
    cache = os.environ.get("GLIM_AMD_SCAN_CACHE")  # workload-generation cache for back-to-back A/B runs of bench.py (never on the hot path)
    if cache:
        import hashlib

        h = hashlib.sha1()
        for a in (T, np.ascontiguousarray(dirs), scene.boxes, scene.cylinders, np.array([scene.half_x, scene.half_y, scene.height, frame_id, sigma, seed, max_range, min_range], dtype=np.float64)):
            h.update(np.ascontiguousarray(a).tobytes())
        path = os.path.join(cache, h.hexdigest() + ".npy")
        if os.path.exists(path):
            return np.load(path)
        pts = _scan(scene, T, dirs, frame_id, sigma, seed, max_range, min_range)
        os.makedirs(cache, exist_ok=True)
        np.save(path + ".tmp.npy", pts)
        os.replace(path + ".tmp.npy", path)
        return pts
    return _scan(scene, T, dirs, frame_id, sigma, seed, max_range, min_range)


def _scan(scene, T, dirs, frame_id, sigma, seed, max_range, min_range):
    dw = dirs @ T[:3, :3].T
    t = scene.raycast(T[:3, 3], dw)
    rng = np.random.default_rng(seed + int(frame_id))
    t = t + rng.normal(0.0, sigma, size=t.shape)
    ok = np.isfinite(t) & (t >= min_range) & (t <= max_range)
    pts = dirs[ok] * t[ok, None]
    return pts.astype(np.float32)


def arc_trajectory(n, step=0.5, yaw_step_deg=2.0, start=(-10.0, -6.0, 1.8), yaw0_deg=10.0):
    """n sensor poses on a gentle arc (step metres forward, yaw_step degrees per frame)."""
    poses = []
    x, y, z = start
    yaw = math.radians(yaw0_deg)
    for _ in range(n):
        poses.append(pose(x, y, z, yaw))
        x += step * math.cos(yaw)
        y += step * math.sin(yaw)
        yaw += math.radians(yaw_step_deg)
    return poses


def grid_trajectory(nx, ny, spacing=2.0, z=1.8):
    """nx*ny sensor poses on a boustrophedon grid walk centred in the room (config 4: 16 x 16 at 2 m)."""
    poses = []
    x0, y0 = -(nx - 1) * spacing / 2, -(ny - 1) * spacing / 2
    for j in range(ny):
        cols = range(nx) if j % 2 == 0 else range(nx - 1, -1, -1)
        for i in cols:
            yaw = 0.0 if j % 2 == 0 else math.pi
            poses.append(pose(x0 + i * spacing, y0 + j * spacing, z, yaw + 0.03 * ((i * 7 + j * 3) % 5)))
    return poses


def relative_pose(T_world_a, T_world_b):
    """T_a_b (maps frame-b points into frame a)."""
    return np.linalg.inv(T_world_a) @ T_world_b
