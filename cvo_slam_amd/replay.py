"""Sequence replay harness: the on-disk side of the reference's `run_SLAM` / `cvo_main` drivers.

* association file  -- `run_SLAM.cpp:101-131`: one line per frame, four whitespace-separated tokens
                       `rgb_timestamp rgb_path depth_timestamp depth_path` (TUM `associate.py` output)
* images            -- `run_SLAM.cpp:134-143`: `cv::imread(rgb)` (8-bit, BGR in memory) and
                       `cv::imread(depth, ANYDEPTH)` (16-bit); decoded here by a minimal PNG reader
                       (non-interlaced, 8/16-bit, gray / RGB / RGBA), since the image has no OpenCV
* calibration       -- the `Camera.fx/fy/cx/cy` and `DepthMapFactor` keys of the reference's yaml (`cvo.cpp:18-33`)
* trajectory        -- `run_SLAM.cpp:79-84`: `timestamp tx ty tz qx qy qz qw` per frame
* replay            -- frame-to-frame CVO odometry through the C ABI (set_pcd / match_odometry from the images,
                       `update_fixed_pcd` after every frame, poses chained like `accum_transform`, `cvo.cpp:816`);
                       the reference's keyframe / loop-closure logic is control plane and stays out of scope.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np


# ----------------------------------------------------------------------------- PNG (the subset RGB-D datasets use)
def read_png(path: str) -> np.ndarray:
    """(h, w) uint8/uint16 for gray, (h, w, 3) uint8 for colour in R,G,B order.  Non-interlaced 8/16-bit PNGs."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if hdr is None:
        raise ValueError(f"{path}: no IHDR chunk")
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or ctype not in (0, 2, 4, 6):
        raise ValueError(f"{path}: unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    bpp = ch * depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:                                   # Sub: running sum per byte lane
            cur = line.copy().reshape(-1, bpp)
            cur = (np.cumsum(cur, axis=0) & 255).reshape(-1)
        else:                                           # Average / Paeth: sequential in x
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if ft == 3:
                    pred = (a + b) >> 1
                elif ft == 4:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError(f"{path}: bad filter type {ft}")
                cur[x] = (line[x] + pred) & 255
        out[y] = cur
        prev = cur
    if depth == 16:
        img = out.reshape(h, w, ch, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]
    else:
        img = out.reshape(h, w, ch)
    if ch == 2:
        img = img[..., :1]
    if ch == 4:
        img = img[..., :3]
    return np.ascontiguousarray(img[..., 0] if img.shape[-1] == 1 else img)


def write_png(path: str, img: np.ndarray) -> None:
    """8-bit (h, w) / (h, w, 3) RGB or 16-bit (h, w) gray; filter 0, for tests and synthetic sequences."""
    img = np.asarray(img)
    if img.dtype == np.uint16 and img.ndim == 2:
        depth, ctype, raw = 16, 0, img.astype(">u2").tobytes()
        stride = img.shape[1] * 2
    elif img.dtype == np.uint8 and img.ndim in (2, 3):
        depth, ctype = 8, (0 if img.ndim == 2 else 2)
        raw = np.ascontiguousarray(img).tobytes()
        stride = img.shape[1] * (1 if img.ndim == 2 else 3)
    else:
        raise ValueError("write_png: uint8 gray/RGB or uint16 gray only")
    h, w = img.shape[:2]
    lines = b"".join(b"\x00" + raw[y * stride:(y + 1) * stride] for y in range(h))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(lines, 3)) + chunk(b"IEND", b""))


# ----------------------------------------------------------------------------- run_SLAM.cpp:101-131
def read_associations(path: str):
    """[(rgb_timestamp, rgb_path, depth_path), ...]; empty lines skipped, the depth timestamp is read and dropped."""
    out = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if len(tok) < 4:
                raise ValueError(f"{path}: expected 'rgb_time rgb_path depth_time depth_path', got {line!r}")
            out.append((tok[0], tok[1], tok[3]))
    return out


def read_calibration(path: str):
    """(scaling_factor, fx, fy, cx, cy) from the reference's yaml keys (cvo.cpp:18-33)."""
    want = {"Camera.fx": None, "Camera.fy": None, "Camera.cx": None, "Camera.cy": None, "DepthMapFactor": None}
    with open(path) as f:
        for line in f:
            if ":" in line and not line.lstrip().startswith(("#", "%")):
                k, v = line.split(":", 1)
                if k.strip() in want:
                    want[k.strip()] = float(v.split("#")[0].strip())
    missing = [k for k, v in want.items() if v is None]
    if missing:
        raise ValueError(f"{path}: missing {missing}")
    return (want["DepthMapFactor"], want["Camera.fx"], want["Camera.fy"], want["Camera.cx"], want["Camera.cy"])


def load_frame(folder: str, rgb_path: str, depth_path: str):
    """(bgr8, depth16) as cv::imread / cv::imread(ANYDEPTH) would return them (run_SLAM.cpp:134-143)."""
    rgb = read_png(os.path.join(folder, rgb_path))
    dep = read_png(os.path.join(folder, depth_path))
    if rgb.ndim == 2:
        rgb = np.repeat(rgb[..., None], 3, axis=2)
    if rgb.dtype != np.uint8 or dep.ndim != 2:
        raise ValueError("expected an 8-bit colour image and a single-channel depth image")
    return np.ascontiguousarray(rgb[..., ::-1]), dep.astype(np.uint16)


# ----------------------------------------------------------------------------- run_SLAM.cpp:79-84
def rotation_to_quaternion(R: np.ndarray):
    """(x, y, z, w) of a rotation matrix, w >= 0 (Eigen::Quaterniond(R) up to the sign convention)."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R))); j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s; q[j] = (R[j, i] + R[i, j]) / s; q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    q /= np.linalg.norm(q)
    return q if q[3] >= 0 else -q


def write_trajectory(path: str, stamps, poses) -> None:
    """`timestamp tx ty tz qx qy qz qw` per frame; poses: (3, 4) or (4, 4) camera-to-world."""
    with open(path, "w") as f:
        for ts, P in zip(stamps, poses):
            P = np.asarray(P, np.float64)
            q = rotation_to_quaternion(P[:3, :3])
            f.write(f"{ts} {P[0, 3]:.9g} {P[1, 3]:.9g} {P[2, 3]:.9g} {q[0]:.9g} {q[1]:.9g} {q[2]:.9g} {q[3]:.9g}\n")


# ----------------------------------------------------------------------------- frame-to-frame odometry replay
def replay_odometry(frames, camera, params=None, device: int = 0, num_want: int = 3000):
    """frames: iterable of (bgr8, depth16).  Returns (poses, info): poses[k] = (4, 4) pose of camera k in the frame of camera 0
    (chained like accum_transform, cvo.cpp:816, but from the final transform of every alignment), info[k] = dict(iterations,
    nnz, points)."""
    import cvo_slam_amd as ca
    g = ca.Cvo(params, device=device)
    g.set_num_want(num_want)
    pose = np.eye(4)
    poses, info = [], []
    for k, (bgr, dep) in enumerate(frames):
        if k == 0:
            g.set_pcd_images(bgr, dep, camera)                      # cvo.cpp:352-360: the first frame only fills the fixed cloud
            poses.append(pose.copy()); info.append(dict(iterations=0, nnz=0, points=g.get_cloud(0)[0].shape[0]))
            continue
        T = g.match_odometry_images(bgr, dep, camera)               # moving (frame k) -> fixed (frame k-1), cvo.cpp:461-473
        step = np.eye(4); step[:3, :] = T
        pose = pose @ step
        poses.append(pose.copy())
        info.append(dict(iterations=g.get_iteration_number() + 1, nnz=g.get_A_nonzero(), points=g.get_fixed_and_moving_number()[1]))
        g.update_fixed_pcd()                                        # cvo.cpp:578-582: this frame is the next one's reference
    g.close()
    return poses, info


def replay_sequence(folder: str, assoc: str, calib: str, out_path: str, max_frames: int = 0, device: int = 0):
    """The `cvo_main` loop (thirdparty/cvo/src/cvo_main.cpp:28-66) on a TUM-format sequence; writes the trajectory file."""
    entries = read_associations(assoc)
    if max_frames > 0:
        entries = entries[:max_frames]
    cam = read_calibration(calib)
    frames = (load_frame(folder, r, d) for (_, r, d) in entries)
    poses, info = replay_odometry(frames, cam, device=device)
    write_trajectory(out_path, [e[0] for e in entries], poses)
    return poses, info
