"""Seeded synthetic RGB-D frame pairs shaped like the reference's inputs.

No dataset ships with the reference and there is no network, so bench and tests
render their own scenes: 6-12 textured planes in front of a back wall, ray-cast to
a uint16 depth image (depth factor 5000, ~5 % holes) and an 8-bit BGR image, from
two camera poses a small SE(3) apart.  Point selection stands in for the
reference's DSO pixel selector (thirdparty/cvo/src/pcd_generator.cpp:145-155,
out of scope here): the top-gradient valid-depth pixel of every g x g cell, in
scan order like pcd_generator.cpp:466-497.  Back-projection and the 5-channel
feature (B, G, R, dx, dy; raw 0-255 values, Q7) follow
pcd_generator.cpp:471-476 and :593-609.

Everything is numpy on the host; the same arrays feed the oracle and the HIP path.
"""
from __future__ import annotations

import dataclasses
import numpy as np

# config/TUM1.yaml:8-20 and config/ETH3D_training_1.yaml:10-13 of the reference
TUM1 = dict(w=640, h=480, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, depth_factor=5000.0, cell=10)
ETH3D = dict(w=736, h=456, fx=726.28741455078, fy=726.28741455078, cx=354.6496887207, cy=186.46566772461,
             depth_factor=5000.0, cell=6)


@dataclasses.dataclass
class Cloud:
    xyz: np.ndarray    # (n, 3) float32, AoS like cloud_t (data_type.h:30)
    feat: np.ndarray   # (5, n) float32, channel-major = Eigen col-major N x 5 (data_type.h:75)

    @property
    def n(self) -> int:
        return int(self.xyz.shape[0])


@dataclasses.dataclass
class Pair:
    fixed: Cloud
    moving: Cloud
    true_transform: np.ndarray   # (3, 4) float64, maps moving-frame points into the fixed frame


def _rodrigues(axis: np.ndarray, angle: float) -> np.ndarray:
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def random_motion(rng: np.random.Generator, max_deg: float, max_trans: float):
    axis = rng.normal(size=3)
    ang = np.deg2rad(rng.uniform(0.25 * max_deg, max_deg))
    t = rng.normal(size=3)
    t = t / np.linalg.norm(t) * rng.uniform(0.25 * max_trans, max_trans)
    return _rodrigues(axis, ang), t


def _make_scene(rng: np.random.Generator):
    planes = []
    # back wall: unbounded, faces the camera, 3.5-4 m away
    planes.append(dict(p0=np.array([0.0, 0.0, rng.uniform(3.5, 4.0)]), n=np.array([0.0, 0.0, -1.0]),
                       u=np.array([1.0, 0.0, 0.0]), v=np.array([0.0, 1.0, 0.0]), ext=(1e9, 1e9)))
    for _ in range(int(rng.integers(6, 13))):
        z = rng.uniform(0.8, 3.2)
        p0 = np.array([rng.uniform(-0.6, 0.6) * z, rng.uniform(-0.45, 0.45) * z, z])
        n = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), -1.0])
        n /= np.linalg.norm(n)
        u = np.cross(n, np.array([0.0, 1.0, 0.0])); u /= np.linalg.norm(u)
        v = np.cross(n, u)
        planes.append(dict(p0=p0, n=n, u=u, v=v, ext=(rng.uniform(0.15, 0.6), rng.uniform(0.15, 0.6))))
    for pl in planes:   # texture: 8 random 2-D sinusoids per colour channel
        pl["amp"] = rng.uniform(8.0, 28.0, size=(3, 8))
        pl["freq"] = rng.uniform(-9.0, 9.0, size=(3, 8, 2))
        pl["phase"] = rng.uniform(0, 2 * np.pi, size=(3, 8))
        pl["base"] = rng.uniform(70.0, 185.0, size=3)
    return planes


def _render(planes, R_wc, t_wc, cam, rng: np.random.Generator):
    w, h = cam["w"], cam["h"]
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d_cam = np.stack([(xs - cam["cx"]) / cam["fx"], (ys - cam["cy"]) / cam["fy"], np.ones_like(xs)], axis=-1)
    d_w = d_cam @ R_wc.T                                  # (h, w, 3)
    best_s = np.full((h, w), np.inf)
    best_id = np.full((h, w), -1, dtype=np.int32)
    best_a = np.zeros((h, w)); best_b = np.zeros((h, w))
    for pid, pl in enumerate(planes):
        denom = d_w @ pl["n"]
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (pl["n"] @ (pl["p0"] - t_wc)) / denom
        hit = t_wc + s[..., None] * d_w - pl["p0"]
        a = hit @ pl["u"]; b = hit @ pl["v"]
        ok = (s > 0.2) & np.isfinite(s) & (np.abs(a) <= pl["ext"][0]) & (np.abs(b) <= pl["ext"][1]) & (s < best_s)
        best_s = np.where(ok, s, best_s); best_id = np.where(ok, pid, best_id)
        best_a = np.where(ok, a, best_a); best_b = np.where(ok, b, best_b)
    bgr = np.zeros((h, w, 3))
    for pid, pl in enumerate(planes):
        m = best_id == pid
        if not m.any():
            continue
        a = best_a[m]; b = best_b[m]
        for c in range(3):
            arg = 2 * np.pi * (pl["freq"][c, :, 0][None, :] * a[:, None] + pl["freq"][c, :, 1][None, :] * b[:, None]) + pl["phase"][c][None, :]
            bgr[m, c] = pl["base"][c] + (pl["amp"][c][None, :] * np.sin(arg)).sum(axis=1)
    bgr += rng.normal(0.0, 3.0, size=bgr.shape)
    bgr8 = np.clip(np.rint(bgr), 0, 255).astype(np.uint8)
    valid = np.isfinite(best_s) & (best_id >= 0)
    depth = np.where(valid, np.clip(np.rint(best_s * cam["depth_factor"]), 0, 65535), 0).astype(np.uint16)
    holes = rng.random(size=depth.shape) < 0.05
    depth[holes] = 0
    return bgr8, depth


def select_cloud(bgr8: np.ndarray, depth: np.ndarray, cam) -> Cloud:
    """Top-gradient valid-depth pixel per cell -> positions + (B,G,R,dx,dy) features."""
    h, w = depth.shape
    g = cam["cell"]
    b = bgr8[..., 0].astype(np.float32); gch = bgr8[..., 1].astype(np.float32); r = bgr8[..., 2].astype(np.float32)
    gray = (np.float32(0.299) * r + np.float32(0.587) * gch + np.float32(0.114) * b).astype(np.float32)
    dx = np.zeros_like(gray); dy = np.zeros_like(gray)
    dx[:, 1:-1] = np.float32(0.5) * (gray[:, 2:] - gray[:, :-2])
    dy[1:-1, :] = np.float32(0.5) * (gray[2:, :] - gray[:-2, :])
    score = dx * dx + dy * dy
    score = np.where(depth != 0, score, np.float32(-1.0))
    hc, wc = h // g, w // g
    sc = score[: hc * g, : wc * g].reshape(hc, g, wc, g).transpose(0, 2, 1, 3).reshape(hc, wc, g * g)
    arg = sc.argmax(axis=2)
    ok = np.take_along_axis(sc, arg[..., None], axis=2)[..., 0] >= 0
    cy_idx, cx_idx = np.nonzero(ok)
    py = cy_idx * g + arg[cy_idx, cx_idx] // g
    px = cx_idx * g + arg[cy_idx, cx_idx] % g
    order = np.lexsort((px, py))                         # scan order: y then x (pcd_generator.cpp:466-467)
    py = py[order]; px = px[order]
    dep = depth[py, px].astype(np.float32)
    z = (dep / np.float32(cam["depth_factor"])).astype(np.float32)                                       # pcd_generator.cpp:473
    x = ((px.astype(np.float32) - np.float32(cam["cx"])) * z / np.float32(cam["fx"])).astype(np.float32)  # :475
    y = ((py.astype(np.float32) - np.float32(cam["cy"])) * z / np.float32(cam["fy"])).astype(np.float32)  # :476
    xyz = np.ascontiguousarray(np.stack([x, y, z], axis=1), dtype=np.float32)
    feat = np.ascontiguousarray(np.stack([b[py, px], gch[py, px], r[py, px], dx[py, px], dy[py, px]], axis=0), dtype=np.float32)
    return Cloud(xyz=xyz, feat=feat)


def make_pair(index: int, cam=TUM1, max_deg: float = 2.0, max_trans: float = 0.03, base_seed: int = 20240) -> Pair:
    """Pair `index` (seed = 20240 + index): fixed = frame A, moving = frame B."""
    rng = np.random.default_rng(base_seed + index)
    planes = _make_scene(rng)
    R_a, t_a = np.eye(3), np.zeros(3)
    R_ab, t_ab = random_motion(rng, max_deg, max_trans)   # pose of camera B in frame A
    img_a, dep_a = _render(planes, R_a, t_a, cam, rng)
    img_b, dep_b = _render(planes, R_ab, t_ab, cam, rng)
    fixed = select_cloud(img_a, dep_a, cam)
    moving = select_cloud(img_b, dep_b, cam)
    true_tf = np.concatenate([R_ab, t_ab[:, None]], axis=1)   # p_A = R_ab p_B + t_ab
    return Pair(fixed=fixed, moving=moving, true_transform=true_tf)


def make_frames(index: int, cam=TUM1, max_deg: float = 2.0, max_trans: float = 0.03, base_seed: int = 20240):
    """The two RGB-D frames of pair `index` as images (what cvo::set_pcd is given, cvo.cpp:345): (bgr8 h x w x 3,
    depth16 h x w) for frame A and for frame B, plus the true transform.  Same scene / motion / noise as make_pair."""
    rng = np.random.default_rng(base_seed + index)
    planes = _make_scene(rng)
    R_a, t_a = np.eye(3), np.zeros(3)
    R_ab, t_ab = random_motion(rng, max_deg, max_trans)
    img_a, dep_a = _render(planes, R_a, t_a, cam, rng)
    img_b, dep_b = _render(planes, R_ab, t_ab, cam, rng)
    return (img_a, dep_a), (img_b, dep_b), np.concatenate([R_ab, t_ab[:, None]], axis=1)


def make_sequence(index: int, n_frames: int = 4, cam=TUM1, max_deg: float = 1.5, max_trans: float = 0.025, base_seed: int = 30300):
    """An RGB-D sequence of one static scene: frames[k] = (bgr8, depth16) seen from camera k, poses[k] = (4, 4) pose of camera k
    in the frame of camera 0 (random inter-frame motion up to max_deg / max_trans, like make_pair)."""
    rng = np.random.default_rng(base_seed + index)
    planes = _make_scene(rng)
    R, t = np.eye(3), np.zeros(3)
    frames, poses = [], []
    for k in range(n_frames):
        if k > 0:
            dR, dt = random_motion(rng, max_deg, max_trans)           # pose of camera k in the frame of camera k-1
            t = R @ dt + t; R = R @ dR
        frames.append(_render(planes, R, t, cam, rng))
        P = np.eye(4); P[:3, :3] = R; P[:3, 3] = t
        poses.append(P)
    return frames, poses


def camera_tuple(cam):
    """(scaling_factor, fx, fy, cx, cy) = cvo::camera_info (data_type.h:33-39)."""
    return (float(cam["depth_factor"]), float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"]))


def make_small_pair(seed: int, n: int = 300, max_deg: float = 2.0, max_trans: float = 0.03) -> Pair:
    """Small unstructured pair for second-scale oracle tests: points on a few
    random surfaces with smooth colour fields, moving = rigidly displaced + noise."""
    rng = np.random.default_rng(seed)
    uv = rng.uniform(-1, 1, size=(n, 2))
    z = 1.5 + 0.3 * np.sin(2.0 * uv[:, 0]) + 0.2 * uv[:, 1] + 0.5 * (rng.integers(0, 2, size=n))
    pts = np.stack([uv[:, 0] * 0.8, uv[:, 1] * 0.6, z], axis=1)

    def colour(p):
        f = np.stack([128 + 60 * np.sin(3 * p[:, 0] + 1.0) + 30 * np.cos(5 * p[:, 1]),
                      128 + 50 * np.sin(4 * p[:, 1] + 0.3) + 30 * np.cos(2 * p[:, 2]),
                      128 + 40 * np.sin(2 * p[:, 0] - 2 * p[:, 1]),
                      20 * np.cos(3 * p[:, 0]), 20 * np.sin(3 * p[:, 1])], axis=0)
        return f

    R, t = random_motion(rng, max_deg, max_trans)
    # moving-frame coordinates of (a jittered resampling of) the same surface points
    jitter = rng.normal(0, 0.002, size=pts.shape)
    pts_b_world = pts + jitter
    moving_xyz = (pts_b_world - t) @ R                    # p_B = R^T (p_A - t)
    fixed = Cloud(xyz=np.ascontiguousarray(pts, dtype=np.float32),
                  feat=np.ascontiguousarray(colour(pts) + rng.normal(0, 2, size=(5, n)), dtype=np.float32))
    moving = Cloud(xyz=np.ascontiguousarray(moving_xyz, dtype=np.float32),
                   feat=np.ascontiguousarray(colour(pts_b_world) + rng.normal(0, 2, size=(5, n)), dtype=np.float32))
    return Pair(fixed=fixed, moving=moving, true_transform=np.concatenate([R, t[:, None]], axis=1))
