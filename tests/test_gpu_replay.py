"""Sequence replay end to end: a synthetic RGB-D sequence is written to disk in the TUM layout (PNGs +
association file + calibration yaml), replayed frame to frame through the C ABI (point clouds generated on the GPU from the
images), and the trajectory file is compared with the known camera motion."""
import os

import numpy as np
import pytest

from helpers import rot_trans_err

pytestmark = pytest.mark.gpu


def test_replay_synthetic_sequence_from_disk(hiplib, tmp_path):
    from cvo_slam_amd import replay, synth
    frames, poses_true = synth.make_sequence(1, n_frames=5)
    os.makedirs(tmp_path / "rgb"); os.makedirs(tmp_path / "depth")
    lines = []
    for k, (bgr, dep) in enumerate(frames):
        ts = f"{1000.0 + k / 30.0:.6f}"
        replay.write_png(str(tmp_path / "rgb" / f"{ts}.png"), bgr[..., ::-1])          # files hold R,G,B; cv::imread hands back B,G,R
        replay.write_png(str(tmp_path / "depth" / f"{ts}.png"), dep)
        lines.append(f"{ts} rgb/{ts}.png {ts} depth/{ts}.png")
    (tmp_path / "assoc.txt").write_text("\n".join(lines) + "\n")
    cam = synth.TUM1
    (tmp_path / "cam.yaml").write_text(f"%YAML:1.0\nCamera.fx: {cam['fx']}\nCamera.fy: {cam['fy']}\nCamera.cx: {cam['cx']}\nCamera.cy: {cam['cy']}\n"
                                       f"DepthMapFactor: {cam['depth_factor']}\n")
    # the loader returns exactly the rendered images
    b0, d0 = replay.load_frame(str(tmp_path), lines[0].split()[1], lines[0].split()[3])
    np.testing.assert_array_equal(b0, frames[0][0]); np.testing.assert_array_equal(d0, frames[0][1])
    out = str(tmp_path / "traj.txt")
    poses, info = replay.replay_sequence(str(tmp_path), str(tmp_path / "assoc.txt"), str(tmp_path / "cam.yaml"), out)
    assert len(poses) == 5 and all(i["points"] > 2000 for i in info)
    rows = [l.split() for l in open(out).read().splitlines()]
    assert [r[0] for r in rows] == [l.split()[0] for l in lines] and all(len(r) == 8 for r in rows)
    for k in range(1, 5):
        re, te = rot_trans_err(poses[k][:3, :], poses_true[k][:3, :])
        assert re < 6e-3 and te < 2.5e-2, (k, re, te)                                   # a few millimetres / tenths of a degree per step, chained
        np.testing.assert_allclose([float(v) for v in rows[k][1:4]], poses[k][:3, 3], rtol=1e-6, atol=1e-8)
        q = replay.rotation_to_quaternion(poses[k][:3, :3])
        np.testing.assert_allclose([float(v) for v in rows[k][4:]], q, atol=1e-7)
    # the same frames straight from memory give the same poses (the disk formats are lossless)
    poses_mem, _ = replay.replay_odometry(frames, synth.camera_tuple(cam))
    for a, b in zip(poses, poses_mem):
        np.testing.assert_array_equal(a, b)
