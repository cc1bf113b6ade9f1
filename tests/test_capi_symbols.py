"""The C-ABI library builds, loads, exports every symbol include/cvo_hip.h declares,
and refuses to compute without a gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cvo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cvo_[a-zA-Z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hiplib):
    L = hiplib.load_library()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/cvo_hip.h but not exported"
    assert sorted(hiplib.api.ABI_SYMBOLS) == names                        # the python mirror binds exactly the ABI
    out = subprocess.check_output(["nm", "-D", "--defined-only", hiplib.lib_path()], text=True)
    exported = set(re.findall(r" T (cvo_[a-zA-Z0-9_]+)", out))
    assert set(names) <= exported


def test_default_params_are_the_reference_constants(hiplib):
    p = hiplib.default_params()                                           # cvo.cpp:35-51
    assert (p.ell, p.sigma, p.sp_thres, p.c, p.d) == pytest.approx((0.15, 0.1, 8e-3, 7.0, 7.0))
    assert (p.c_ell, p.c_sigma, p.max_iter) == (200.0, 1.0, 2000)
    assert (p.min_step, p.eps, p.eps_2) == pytest.approx((0.2, 5e-5, 1e-5))


def test_code_object_is_gfx950_only(hiplib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={hiplib.lib_path()}"],
                         capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets


def test_no_gpu_means_loud_failure_not_fallback(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert hiplib.device_count() == 0
    with pytest.raises(hiplib.CvoError) as e:
        hiplib.Cvo()
    assert e.value.code == 5                                              # CVO_ERR_NO_DEVICE
    with pytest.raises(hiplib.CvoError):
        hiplib.CvoBatch(4)


def test_product_never_touches_the_oracle():
    """Nothing under cvo_slam_amd/ or include/ may import, include or link oracle/."""
    bad = []
    for base in ("cvo_slam_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"pyoracle|cvo_oracle|libcvo_oracle|import oracle|from oracle", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
