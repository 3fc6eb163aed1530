"""The impulse-response pipeline behind the C-ABI (rvb_pipeline_*, csrc/pipeline.hip) driven by a C++11 caller:
tests/cpp/test_pipeline.cpp sends twenty jobs with their own microphone / source through four contexts and holds every result
against the step-by-step calls on a fifth context, bit for bit (exact mode), then the HRTF model with a facing per job and the
error paths.  Without a GPU the program must compile, link against librvb_hip.so alone (plain g++, no HIP headers) and stop at
rvb_create: there is no CPU path."""
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "parallel-reverb-raytracer_amd")
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "test_pipeline")


def _build():
    subprocess.check_call(["make", "-C", PKG, "-j4"], stdout=subprocess.DEVNULL)
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_pipeline.cpp"), "-o", BIN, "-L" + PKG, "-lrvb_hip", "-Wl,-rpath," + PKG])


def test_pipeline_caller_compiles_and_refuses_to_run_without_gpu():
    import torch
    _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; see the gpu-marked test")
    r = subprocess.run([BIN], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 2 and "no CPU path" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_caller_gets_bit_identical_impulse_responses_from_the_pipeline():
    _build()
    r = subprocess.run([BIN], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "all pipeline checks passed" in r.stdout
