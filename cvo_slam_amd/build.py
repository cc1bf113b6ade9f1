"""Compile libcvo_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["cvo_kernels.hip", "cvo_score_kernels.hip", "cvo_pcd_kernels.hip", "cvo_selftest.hip", "cvo_capi.hip", "cvo_hip.hpp"]
HEADERS = ["cvo_device.h", "cvo_math.hpp", os.path.join("..", "..", "include", "cvo_hip.h")]


def lib_path() -> str:
    return os.path.join(HERE, "libcvo_hip.so")


def is_stale() -> bool:
    out = lib_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(CSRC, "Makefile")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    if force or is_stale():
        cmd = ["make", "-C", CSRC, "-j4"] + ([] if verbose else ["-s"])
        subprocess.check_call(cmd)
    return lib_path()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
