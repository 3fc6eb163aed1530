#!/usr/bin/env python3
"""Developer tool: duration of the binning stage at workload C2 in both modes (HIP events around the stage, one IR on the GPU).
    python tools/mode_bench.py [rays] [reflections] [triangles]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
from parallel_reverb_raytracer_amd import capi, dtypes, scenes  # noqa: E402


def main():
    import torch
    nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    nrefl = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    ntri = int(sys.argv[3]) if len(sys.argv) > 3 else 75000
    scene, info = scenes.cathedral(ntri)
    ctx = capi.Context(0)
    ctx.set_scene(scene)
    ctx.raytrace(info["mic"], info["source"], scenes.sphere_directions(nrays, seed=1), nrefl, dtypes.AIR_COEFFICIENTS)
    images = ctx.get_raw_images(False)
    ctx.ir_configure_speakers(info["mic"], [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], capi.IR_ALL, images)
    lo, hi = ctx.ir_time_range()
    nbins = ctx.ir_bins(hi, lo, 44100.0)
    for mode, name in ((capi.IR_FAST, "fast"), (capi.IR_EXACT, "exact")):
        times = []
        for _ in range(6):
            hist = torch.zeros((2, 8, nbins), device="cuda", dtype=torch.float32)
            torch.cuda.synchronize()
            ctx.ir_accumulate_tensor(lo, 44100.0, nbins, mode, hist)
            ctx.synchronize()
            times.append(sum(v for _, v in ctx.last_timings()))
        print("%s: %s ms (nbins %d)" % (name, " ".join("%.3f" % t for t in times), nbins), flush=True)


if __name__ == "__main__":
    main()
