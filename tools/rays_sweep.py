#!/usr/bin/env python3
"""Developer tool: duration of the path kernel alone on the GPU as a function of the number of rays in the launch, at the C2
scene — shows where a launch stops fitting in one resident round of waves (the waves beyond it start when the first ones finish
and then run a whole 128-bounce chain almost alone).  RVB_PATH_LANES=2|4 picks the kernel (read once per process).
    RVB_PATH_LANES=2 python tools/rays_sweep.py 163840 196608 200000 ...
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
from parallel_reverb_raytracer_amd import capi, dtypes, scenes  # noqa: E402


def main():
    counts = [int(x) for x in sys.argv[1:]] or [100000, 200000]
    nrefl = int(os.environ.get("SWEEP_REFLECTIONS", "128"))
    # SWEEP_SCENE=atrium: the C4 stand-in (SWEEP_TRIANGLES=262000 SWEEP_REFLECTIONS=256 for the configuration itself)
    make = {"cathedral": scenes.cathedral, "atrium": scenes.atrium, "hall": scenes.concert_hall}[os.environ.get("SWEEP_SCENE", "cathedral")]
    (scene, info) = make(int(os.environ.get("SWEEP_TRIANGLES", "75000")))
    group = int(os.environ.get("SWEEP_GROUP", "1"))      # > 1: that many contexts, n rays each, ONE path-kernel launch (rvb_trace_group)
    ctxs = [capi.Context(0) for _ in range(group)]
    for c in ctxs:
        c.set_scene(scene)
    ctx = ctxs[0]
    for n in counts:
        for c in ctxs:
            c.set_directions(scenes.sphere_directions(n, seed=1))
        times = {}
        for _ in range(4):
            if group > 1:
                capi.Context.trace_group(ctxs, [info["mic"]] * group, [info["source"]] * group, nrefl, dtypes.AIR_COEFFICIENTS)
            else:
                ctx.trace(info["mic"], info["source"], nrefl, dtypes.AIR_COEFFICIENTS)
            for c in ctxs:
                c.synchronize()
            for k, v in ctx.last_timings():
                times.setdefault(k, []).append(v)
        line = " ".join("%s %.3f" % (k, float(np.median(v[1:]))) for k, v in times.items())
        path = [float(np.median(v[1:])) for k, v in times.items() if k.startswith("path")][0]
        print("lanes %s rays %d x %d path/100k %.3f | %s" % (os.environ.get("RVB_PATH_LANES", "auto"), n, group, path * 1e5 / (n * group), line), flush=True)


if __name__ == "__main__":
    main()
