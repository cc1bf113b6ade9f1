"""On-disk formats of the sequence replay harness: PNG subset, association file (run_SLAM.cpp:101-131),
calibration yaml keys (cvo.cpp:18-33), trajectory lines (run_SLAM.cpp:79-84)."""
import os
import struct
import zlib

import numpy as np
import pytest

from cvo_slam_amd import replay


def test_png_round_trip_rgb8_and_gray16(tmp_path):
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    dep = rng.integers(0, 65536, size=(37, 53), dtype=np.uint16)
    replay.write_png(str(tmp_path / "c.png"), rgb); replay.write_png(str(tmp_path / "d.png"), dep)
    np.testing.assert_array_equal(replay.read_png(str(tmp_path / "c.png")), rgb)
    np.testing.assert_array_equal(replay.read_png(str(tmp_path / "d.png")), dep)


@pytest.mark.parametrize("ftype", [1, 2, 3, 4])
def test_png_scanline_filters(tmp_path, ftype):
    """Encoders use the Sub / Up / Average / Paeth filters; encode one by hand per type and decode it."""
    rng = np.random.default_rng(ftype)
    img = rng.integers(0, 256, size=(9, 11, 3), dtype=np.uint8)
    h, w, bpp = 9, 11, 3
    flat = img.reshape(h, w * bpp).astype(np.int32)
    lines = b""
    for y in range(h):
        out = np.zeros(w * bpp, np.int32)
        for x in range(w * bpp):
            a = flat[y, x - bpp] if x >= bpp else 0
            b = flat[y - 1, x] if y > 0 else 0
            c = flat[y - 1, x - bpp] if (y > 0 and x >= bpp) else 0
            if ftype == 1: pred = a
            elif ftype == 2: pred = b
            elif ftype == 3: pred = (a + b) >> 1
            else:
                p = a + b - c; pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[x] = (flat[y, x] - pred) & 255
        lines += bytes([ftype]) + out.astype(np.uint8).tobytes()
    chunk = lambda k, b: struct.pack(">I", len(b)) + k + b + struct.pack(">I", zlib.crc32(k + b) & 0xFFFFFFFF)
    path = tmp_path / "f.png"
    path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(lines)) + chunk(b"IEND", b""))
    np.testing.assert_array_equal(replay.read_png(str(path)), img)


def test_association_calibration_and_trajectory_files(tmp_path):
    (tmp_path / "assoc.txt").write_text("1305031453.359684 rgb/1305031453.359684.png 1305031453.374112 depth/1305031453.374112.png\n\n"
                                        "1305031453.391690 rgb/a.png 1305031453.404816 depth/b.png\n")
    a = replay.read_associations(str(tmp_path / "assoc.txt"))
    assert a == [("1305031453.359684", "rgb/1305031453.359684.png", "depth/1305031453.374112.png"), ("1305031453.391690", "rgb/a.png", "depth/b.png")]
    (tmp_path / "cam.yaml").write_text("%YAML:1.0\n# camera\nCamera.fx: 517.306408\nCamera.fy: 516.469215\nCamera.cx: 318.643040\nCamera.cy: 255.313989\n"
                                       "Camera.k1: 0.26\nDepthMapFactor: 5000.0\n")
    assert replay.read_calibration(str(tmp_path / "cam.yaml")) == (5000.0, 517.306408, 516.469215, 318.643040, 255.313989)
    with pytest.raises(ValueError):
        (tmp_path / "bad.yaml").write_text("Camera.fx: 1\n"); replay.read_calibration(str(tmp_path / "bad.yaml"))
    # quaternion of a known rotation: 90 degrees about z -> (0, 0, sin 45, cos 45)
    P = np.eye(4); P[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]; P[:3, 3] = [1, 2, 3]
    replay.write_trajectory(str(tmp_path / "traj.txt"), ["12.5", "13.5"], [np.eye(4), P])
    rows = [l.split() for l in (tmp_path / "traj.txt").read_text().splitlines()]
    assert rows[0] == ["12.5", "0", "0", "0", "0", "0", "0", "1"]
    assert rows[1][0] == "13.5" and [float(v) for v in rows[1][1:4]] == [1, 2, 3]
    np.testing.assert_allclose([float(v) for v in rows[1][4:]], [0, 0, np.sqrt(0.5), np.sqrt(0.5)], atol=1e-9)
    # 180-degree rotation (trace < 0 branch)
    q = replay.rotation_to_quaternion(np.diag([1.0, -1.0, -1.0]))
    np.testing.assert_allclose(np.abs(q), [1, 0, 0, 0], atol=1e-12)
