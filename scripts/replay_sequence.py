#!/usr/bin/env python3
"""Frame-to-frame CVO odometry over a TUM-format RGB-D sequence (association file + PNGs), the counterpart of the reference's
`cvo_main` / `run_SLAM` drivers.  Writes `timestamp tx ty tz qx qy qz qw` per frame.

  python scripts/replay_sequence.py --folder /data/rgbd_dataset_freiburg1_desk --assoc assoc.txt --calib TUM1.yaml
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvo_slam_amd import replay

ap = argparse.ArgumentParser()
ap.add_argument("--folder", required=True, help="sequence root (the paths of the association file are relative to it)")
ap.add_argument("--assoc", required=True, help="association file: rgb_time rgb_path depth_time depth_path")
ap.add_argument("--calib", required=True, help="yaml with Camera.fx/fy/cx/cy and DepthMapFactor")
ap.add_argument("--out", default=None, help="trajectory file (default <folder>/cvo_poses_qt.txt)")
ap.add_argument("--max-frames", type=int, default=0)
args = ap.parse_args()
folder = args.folder if args.folder.endswith("/") else args.folder + "/"
assoc = args.assoc if os.path.isabs(args.assoc) else os.path.join(folder, args.assoc)
out = args.out or os.path.join(folder, "cvo_poses_qt.txt")
t0 = time.perf_counter()
poses, info = replay.replay_sequence(folder, assoc, args.calib, out, args.max_frames)
dt = time.perf_counter() - t0
its = [i["iterations"] for i in info[1:]]
print(f"{len(poses)} frames in {dt:.2f} s ({len(poses) / dt:.1f} frames/s incl. PNG decoding), mean iterations {sum(its) / max(1, len(its)):.1f}; trajectory -> {out}")
