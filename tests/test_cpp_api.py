"""The C++ mirror of the reference's host interface (include/rayverb/rayverb.h over the C-ABI):
tests/cpp/test_rayverb_api.cpp restates the reference's three gtest suites with fixtures that inherit
from the production classes.  On a machine without a GPU it must compile, link and fail with cl::Error
(no CPU fallback); on the GPU box it must pass."""
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "parallel-reverb-raytracer_amd")
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "test_rayverb_api")


def _build():
    subprocess.check_call(["make", "-C", PKG, "-j4"], stdout=subprocess.DEVNULL)
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    assets = os.path.join(ROOT, "tests", "golden", "assets")
    subprocess.check_call([
        "g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "shims"),
        '-DTEST_OBJ="%s"' % os.path.join(assets, "large_square.obj"), '-DTEST_MAT="%s"' % os.path.join(assets, "mat.json"),
        os.path.join(ROOT, "tests", "cpp", "test_rayverb_api.cpp"), "-o", BIN,
        "-L" + PKG, "-lrayverb", "-lrvb_hip", "-Wl,-rpath," + PKG])


def test_cpp_api_compiles_and_refuses_to_run_without_gpu():
    import torch
    _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; see the gpu-marked test")
    r = subprocess.run([BIN], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 2 and "cl::Error" in r.stdout and "no CPU path" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("resident", ["0", "1"])
def test_reference_gtest_suites_through_cpp_api(resident):
    """Default mode (every stage uploads the vector it is handed, as the reference does: an edit of any single element
    between stages is seen) and the opt-in device-resident handover (RVB_API_RESIDENT=1)."""
    _build()
    env = dict(os.environ, RVB_API_RESIDENT=resident)
    r = subprocess.run([BIN], capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "all reference gtest cases passed" in r.stdout
