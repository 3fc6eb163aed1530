#!/usr/bin/env python3
"""Developer tool: a longer randomized parity campaign than the test suite runs — HIP path against the brute-force CPU oracle,
bit for bit, on random triangle soups (duplicates, coplanar overlaps, degenerate and non-finite triangles), tessellated rooms of
random resolution, random source / microphone positions, ray counts and reflection counts.    python tools/fuzz_parity.py [cases] [gpu]
With `gpu` the checker is the oracle source compiled for the GPU (oracle/gpu_oracle.hip, one thread per ray, brute force), which
affords scenes of tens of thousands of triangles and tens of thousands of rays per case."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rvb_import  # noqa: E402

rvb_import.load()
from parallel_reverb_raytracer_amd import capi, dtypes, scenes  # noqa: E402
import pyoracle  # noqa: E402  (checker)
from test_gpu_parity import triangle_soup  # noqa: E402


def same(a, b):
    return (np.array_equal(a["volume"], b["volume"]) and np.array_equal(a["time"], b["time"])
            and np.array_equal(a["position"][:, :3], b["position"][:, :3]))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    big = len(sys.argv) > 2 and sys.argv[2] == "gpu"
    rng = np.random.default_rng(2026 + (1 if big else 0))
    ctx = capi.Context(0)                                   # (the product library loads before the GPU build of the oracle)
    oracle, port = pyoracle.Oracle("gpu" if big else "port"), pyoracle.Oracle("port")
    impulses = images = bad = 0
    t0 = time.perf_counter()
    for case in range(cases):
        kind = case % 3
        if kind == 0:
            scene, extent = triangle_soup(1000 + case), 4.0
        elif kind == 1:
            scene, extent = scenes.rotated_square_room(n=int(rng.integers(1, 60 if big else 12))), 10.0
        else:
            scene, extent = scenes.cathedral(int(rng.integers(600, 40000 if big else 5000)))[0], 9.0
        mic = rng.uniform(-extent, extent, 3) * (0.3, 0.1, 0.3) + (0, 2.0 if kind else 0.0, 0)
        src = rng.uniform(-extent, extent, 3) * (0.3, 0.1, 0.3) + (0, 2.5 if kind else 0.0, 0)
        nrays, nrefl = int(rng.integers(1, 30000 if big else 3000)), int(rng.integers(1, 120 if big else 40))
        if os.environ.get("FUZZ_REFLECTION_MULTIPLE"):      # e.g. 32: whole runs of grouping keys (the path kernels' LDS key runs, escaped rays included)
            m = int(os.environ["FUZZ_REFLECTION_MULTIPLE"])
            nrefl = max(m, (nrefl + m - 1) // m * m)
        dirs = scenes.sphere_directions(nrays, seed=case + 1)
        ctx.set_scene(scene)
        ctx.raytrace(mic, src, dirs, nrefl, dtypes.AIR_COEFFICIENTS)
        want, image, index = oracle.raytrace(scene, mic, src, dirs, nrefl, dtypes.AIR_COEFFICIENTS)
        got_images, want_images = ctx.get_raw_images(False), port.collect_images(image, index, False)
        ok = same(ctx.get_raw_diffuse(), want) and got_images.shape == want_images.shape and same(got_images, want_images)
        impulses += want.shape[0]
        images += want_images.shape[0]
        bad += 0 if ok else 1
        print("case %2d %-9s %5d triangles %5d rays x %2d  %s" % (case, ("soup", "room", "cathedral")[kind], scene[0].shape[0], nrays, nrefl,
                                                                  "ok" if ok else "MISMATCH"), flush=True)
    print("%d cases, %d diffuse impulses and %d image-source impulses compared bit for bit, %d mismatching cases, %.0f s"
          % (cases, impulses, images, bad, time.perf_counter() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
