"""TEST INFRASTRUCTURE ONLY: ctypes access to the CPU oracle (oracle/_build/librvb_oracle.so) and,
when present, to the host-compiled reference kernels (oracle/_ref/librvb_ref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import rvb_import  # noqa: E402

_pkg = rvb_import.load()
from parallel_reverb_raytracer_amd.dtypes import (ATTENUATED, IMPULSE, NUM_IMAGE_SOURCE, SURFACE, TRIANGLE,  # noqa: E402
                                                  aligned_copy, aligned_zeros)

ORACLE_SO = os.path.join(_HERE, "_build", "librvb_oracle.so")
ORACLE_GPU_SO = os.path.join(_HERE, "_build", "librvb_oracle_gpu.so")
REF_SO = os.path.join(_HERE, "_ref", "librvb_ref.so")

_c_f = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_u64 = ctypes.c_uint64


def _ptr(a):
    return a.ctypes.data_as(_vp)


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def _f8(v):
    return (ctypes.c_float * 8)(*[float(x) for x in v])


def cpu_threads(cap=16):
    """Threads worth starting: CPU affinity, clipped by a cgroup-v2 quota and by `cap` (a GPU box gives
    one GPU's share of the host, 16 cores, while reporting all of them)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("RVB_CPU_THREADS", str(cap)))))


def have_ref():
    return os.path.exists(REF_SO)


class Oracle:
    """kind='port': this repo's C restatement; kind='reference': the reference's kernel text
    compiled for the host (build container only, see oracle/ref/build_ref.sh)."""

    def __init__(self, kind="port"):
        self.kind = kind
        if kind == "port":
            self.lib = ctypes.CDLL(ORACLE_SO)
            self.lib.rvbo_find_predelay.restype = ctypes.c_float
            self.lib.rvbo_flatten_bins.restype = _u64
            self.lib.rvbo_collect_images.restype = _u64
            self.lib.rvbo_hrtf_index.restype = ctypes.c_int64
        elif kind == "gpu":
            # the kernel part of rvb_oracle.c compiled for the GPU, one thread per ray (oracle/gpu_oracle.hip): raytrace only
            self.lib = ctypes.CDLL(ORACLE_GPU_SO)
        elif kind == "reference":
            self.lib = ctypes.CDLL(REF_SO)
        elif kind == "reference_alt":            # oracle/sensitivity.py: the other conforming built-ins (build_ref.sh alt)
            self.lib = ctypes.CDLL(os.path.join(_HERE, "_ref", "librvb_ref_alt.so"))
        else:
            raise ValueError(kind)

    # -- kernel raytrace ---------------------------------------------------------------
    def raytrace(self, scene, mic, source, directions, nreflections, air, nthreads=0):
        triangles, vertices, surfaces = scene
        triangles, vertices, surfaces = aligned_copy(triangles), aligned_copy(vertices), aligned_copy(surfaces)
        directions = aligned_copy(np.asarray(directions, dtype=np.float32).reshape(-1, 4))
        nrays = directions.shape[0]
        impulses = aligned_zeros(nrays * nreflections, IMPULSE)
        image = aligned_zeros(nrays * NUM_IMAGE_SOURCE, IMPULSE)
        index = aligned_zeros(nrays * NUM_IMAGE_SOURCE, np.uint64)
        args = [_ptr(directions), _u64(nrays), _ptr(triangles), _u64(triangles.shape[0]), _ptr(vertices),
                _ptr(surfaces), _f3(mic), _f3(source), _u64(nreflections), _f8(air),
                _ptr(impulses), _ptr(image), _ptr(index)]
        if self.kind == "port":
            self.lib.rvbo_raytrace(*args, ctypes.c_int(nthreads if nthreads > 0 else cpu_threads()))
        elif self.kind == "gpu":
            rc = self.lib.rvbo_gpu_raytrace(_ptr(directions), _u64(nrays), _ptr(triangles), _u64(triangles.shape[0]), _ptr(vertices),
                                            _u64(vertices.shape[0]), _ptr(surfaces), _u64(surfaces.shape[0]), _f3(mic), _f3(source),
                                            _u64(nreflections), _f8(air), _ptr(impulses), _ptr(image), _ptr(index), ctypes.c_int(0))
            if rc:
                raise RuntimeError("rvbo_gpu_raytrace failed")
        else:
            self.lib.rvb_ref_raytrace(*args)
        return impulses, image, index

    # -- kernels attenuate / hrtf ------------------------------------------------------
    def attenuate_speaker(self, mic, impulses, direction, coefficient):
        impulses = aligned_copy(impulses)
        out = aligned_zeros(impulses.shape[0], ATTENUATED)
        fn = self.lib.rvbo_attenuate_speaker if self.kind in ("port", "gpu") else self.lib.rvb_ref_attenuate
        fn(_f3(mic), _ptr(impulses), _u64(impulses.shape[0]), _f3(direction), ctypes.c_float(coefficient), _ptr(out))
        return out

    def attenuate_hrtf(self, mic, impulses, table_channel, pointing, up, channel):
        """table_channel: [360][180][8] float32 for this ear."""
        impulses = aligned_copy(impulses)
        table = aligned_zeros(360 * 180 * 8 + 8, np.float32)
        table[:360 * 180 * 8] = np.asarray(table_channel, dtype=np.float32).reshape(-1)
        out = aligned_zeros(impulses.shape[0], ATTENUATED)
        fn = self.lib.rvbo_attenuate_hrtf if self.kind == "port" else self.lib.rvb_ref_hrtf
        fn(_f3(mic), _ptr(impulses), _u64(impulses.shape[0]), _ptr(table), _f3(pointing), _f3(up),
           _u64(channel), _ptr(out))
        return out

    # -- port-only helpers ---------------------------------------------------------------
    def hrtf_index(self, pointing, up, direction):
        return int(self.lib.rvbo_hrtf_index(_f3(pointing), _f3(up), _f3(direction)))

    def closest_hit(self, scene, origin, direction):
        triangles, vertices, _ = scene
        prim, dist = _u64(0), ctypes.c_float(0)
        hit = self.lib.rvbo_closest_hit(_f3(origin), _f3(direction), _ptr(triangles), _u64(triangles.shape[0]),
                                        _ptr(vertices), ctypes.byref(prim), ctypes.byref(dist))
        return bool(hit), int(prim.value), float(dist.value)

    def collect_images(self, image, index, remove_direct):
        nrays = index.shape[0] // NUM_IMAGE_SOURCE
        out = aligned_zeros(nrays * NUM_IMAGE_SOURCE, IMPULSE)
        n = self.lib.rvbo_collect_images(_ptr(image), _ptr(index), _u64(nrays), ctypes.c_int(int(remove_direct)),
                                         _ptr(out), _u64(out.shape[0]))
        return out[:n].copy()

    def find_predelay(self, channels):
        n = channels[0].shape[0]
        arr = (_vp * len(channels))(*[c.ctypes.data for c in channels])
        return float(self.lib.rvbo_find_predelay(arr, _u64(len(channels)), _u64(n)))

    def fix_predelay(self, impulses, seconds):
        self.lib.rvbo_fix_predelay(_ptr(impulses), _u64(impulses.shape[0]), ctypes.c_float(seconds))

    def flatten(self, impulses, samplerate):
        """reference flattenImpulses for one channel -> [8][nbins] float32."""
        impulses = aligned_copy(impulses)
        nbins = int(self.lib.rvbo_flatten_bins(_ptr(impulses), _u64(impulses.shape[0]), ctypes.c_float(samplerate)))
        out = np.zeros((8, nbins), dtype=np.float32)
        self.lib.rvbo_flatten(_ptr(impulses), _u64(impulses.shape[0]), ctypes.c_float(samplerate), _ptr(out), _u64(nbins))
        return out
