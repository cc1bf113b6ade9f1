"""The dependency-free C++ mirror of cvo::cvo (cvo_slam_amd/csrc/cvo_hip.hpp): compiles against
the C ABI with plain g++, and (on the GPU) replays the LocalTracker call sequence
(local_tracker.cpp:228-251, 356-431, 506) with results equal to the oracle's."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from helpers import make_tf, rot_trans_err


def build_mini_tracker(tmp_path, hiplib):
    exe = str(tmp_path / "mini_tracker")
    libdir = os.path.dirname(hiplib.lib_path())
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "mini_tracker.cpp"),
                           "-o", exe, f"-L{libdir}", "-lcvo_hip", f"-Wl,-rpath,{libdir}"])
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path, hiplib):
    exe = build_mini_tracker(tmp_path, hiplib)
    r = subprocess.run([exe, str(tmp_path / "none")], capture_output=True, text=True)
    assert r.returncode == 2 and "need at least 3 frames" in r.stderr


@pytest.mark.gpu
def test_cpp_mini_tracker_matches_oracle(tmp_path, hiplib, oracle):
    from cvo_slam_amd import synth
    rng = np.random.default_rng(8)
    base = synth.make_small_pair(41, n=500)
    frames = [(base.fixed.xyz, base.fixed.feat)]
    for k in range(3):
        tf = make_tf(rng.normal(size=3), 0.008 * (k + 1), 0.008 * (k + 1) * rng.normal(size=3)).astype(np.float64)
        frames.append((((base.moving.xyz.astype(np.float64) - tf[:, 3]) @ tf[:, :3]).astype(np.float32), base.moving.feat))
    for k, (x, f) in enumerate(frames):
        with open(tmp_path / f"frame_{k}.bin", "wb") as fp:
            fp.write(struct.pack("i", x.shape[0])); fp.write(np.ascontiguousarray(x, np.float32).tobytes()); fp.write(np.ascontiguousarray(f, np.float32).tobytes())
    exe = build_mini_tracker(tmp_path, hiplib)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "cvo not initialized !"                             # cvo.cpp:463-466 behaviour through the mirror
    got = [(l.split()[0], np.array(l.split()[1:13], float).reshape(3, 4), float(l.split()[13]), int(l.split()[14]), int(l.split()[15]), float(l.split()[16]))
           for l in lines[1:-1]]

    # the same sequence on the oracle
    odo, kf = oracle.OracleCvo(), oracle.OracleCvo()
    want = []
    odo.set_pcd(*frames[0]); kf.set_pcd(*frames[0])
    rc, T = odo.match(*frames[1]); assert rc == 0
    rc, s = odo.compute_innerproduct(T.astype(np.float32)); want.append(("init_odo", T, s))
    odo.update_fixed_pcd(); kf.reset_transform(T.astype(np.float32))
    for f in frames[2:]:
        rc, T = odo.match(*f); rc, s = odo.compute_innerproduct(T.astype(np.float32)); want.append(("odo", T, s))
        odo.update_fixed_pcd()
        kf.reset_initial(T.astype(np.float32))
        rc, Tk = kf.match(*f); rc, s = kf.compute_innerproduct(Tk.astype(np.float32)); want.append(("kf", Tk, s))
        kf.update_previous_pcd()
    assert len(got) == len(want)
    for (lab_g, Tg, vg, ng, ig, cg), (lab_w, Tw, sw) in zip(got, want):
        assert lab_g == lab_w
        re, te = rot_trans_err(Tg, Tw)
        assert re <= 1e-4 and te <= 1e-4
        assert ng == sw["inn_post"][1] and ig == sw["inliers"]
        assert vg == pytest.approx(sw["inn_post"][0], rel=1e-5) and cg == pytest.approx(sw["cos_angle"], rel=1e-5)
    st = kf.get_state()
    assert [int(v) for v in lines[-1].split()[1:]] == [st["iter"], st["A_nonzero"], st["num_fixed"], st["num_moving"]]
