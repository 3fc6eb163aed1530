import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import rvb_import  # noqa: E402

rvb_import.load()

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (this repo's C restatement), built on demand.  Test infrastructure only."""
    so = os.path.join(ROOT, "oracle", "_build", "librvb_oracle.so")
    src = os.path.join(ROOT, "oracle", "rvb_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/librvb_oracle.so"])
    import pyoracle
    return pyoracle.Oracle("port")


@pytest.fixture(scope="session")
def gpu_oracle():
    """The same oracle source compiled for the GPU, one thread per ray, brute force (oracle/gpu_oracle.hip): checks whole
    full-size runs.  Test infrastructure only; needs a GPU."""
    so = os.path.join(ROOT, "oracle", "_build", "librvb_oracle_gpu.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/librvb_oracle_gpu.so"])
    # load order: torch's HIP runtime, then the product library, then this one (two HIP runtimes live in the process, see capi.py)
    from parallel_reverb_raytracer_amd import capi
    capi.load_library()
    import pyoracle
    return pyoracle.Oracle("gpu")


@pytest.fixture(scope="session")
def reference_oracle():
    """The reference's own kernels compiled for the host (oracle/_ref); skipped where not built."""
    import pyoracle
    if not pyoracle.have_ref():
        pytest.skip("oracle/_ref/librvb_ref.so not built (needs /root/reference, build container only)")
    return pyoracle.Oracle("reference")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_scene(g):
    from parallel_reverb_raytracer_amd.dtypes import aligned_copy
    return aligned_copy(g["triangles"]), aligned_copy(g["vertices"]), aligned_copy(g["surfaces"])


def golden_impulses(g, prefix):
    from parallel_reverb_raytracer_amd.dtypes import IMPULSE, aligned_zeros
    imp = aligned_zeros(g[prefix + "_in_time"].shape[0], IMPULSE)
    imp["volume"] = g[prefix + "_in_volume"]
    imp["position"][:, :3] = g[prefix + "_in_position"]
    imp["time"] = g[prefix + "_in_time"]
    return imp
