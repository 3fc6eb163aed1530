#!/usr/bin/env python3
"""Developer tool: where a wave of the two-lane path kernel spends its cycles (vote / node step until its loads are back / rest of
the node step / the same for leaf steps / shading steps), from s_memtime stamps of a DIAGNOSTIC build of the library
(-DRVB_STAMPS=1: tools/build_variant.sh stamps -DRVB_STAMPS=1, RVB_LIB=.../_variants/lib_stamps.so; never shipped — a stamp drains the
LDS queue and costs about 40 cycles).     RVB_STAMPS=1 RVB_PATH_LANES=2 RVB_LIB=... python tools/pair_stamps.py 100000 200000 800000"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
from parallel_reverb_raytracer_amd import capi, dtypes, scenes  # noqa: E402


def pipeline_clock(irs):
    """The clock the chip holds while the bench pipeline runs (-DRVB_STAMPS=2 build, RVB_STAMPS NOT set so that the stamps add up over the traces):
    shader cycles / 100-MHz ticks of the path kernel's loop, summed over every wave of every trace."""
    scene, info = scenes.cathedral(75000)
    ctxs = [capi.Context(0) for _ in range(4)]
    dirs = scenes.sphere_directions(100000, seed=1)
    for k, c in enumerate(ctxs):
        if k: c.share_scene(ctxs[0])
        else: c.set_scene(scene)
        c.set_directions(dirs)
    pipe = capi.Pipeline(ctxs)
    import time
    try:
        pipe.configure_speakers([(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], 128, dtypes.AIR_COEFFICIENTS, 44100.0, True, capi.IR_EXACT)
        sent = taken = 0
        t0 = None
        while taken < irs:
            while sent < irs and pipe.pending() < 8:
                pipe.submit(info["mic"], info["source"])
                sent += 1
            pipe.next(copy=False)
            taken += 1
            if taken == 16:
                t0 = time.perf_counter()
        seconds = time.perf_counter() - t0
    finally:
        pipe.close()
    cycles = ticks = waves = 0
    for c in ctxs:
        st = c.debug_stamps()
        cycles += st[9]; ticks += st[11]; waves += st[10]
    print("pipeline: %d IRs, %.3f ms per IR (stamped build); path waves %d, loop %.0f shader cycles and %.0f ticks of 100 MHz per wave: clock %.3f GHz"
          % (irs, seconds * 1e3 / (irs - 16), waves, cycles / waves, ticks / waves, cycles / ticks * 0.1))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "pipeline":
        return pipeline_clock(int(sys.argv[2]))
    counts = [int(x) for x in sys.argv[1:]] or [100000]
    scene, info = scenes.cathedral(75000)
    ctx = capi.Context(0)
    ctx.set_scene(scene)
    names = ["vote", "node: loads", "node: rest", "leaf: loads", "leaf: rest", "shading"]
    for n in counts:
        ctx.set_directions(scenes.sphere_directions(n, seed=1))
        for _ in range(2):
            ctx.trace(info["mic"], info["source"], 128, dtypes.AIR_COEFFICIENTS)
            ctx.synchronize()
        st = ctx.debug_stamps()
        path_ms = [v for k, v in ctx.last_timings() if k.startswith("path")][0]
        waves, loop = st[10], st[9]
        if not waves:
            print("rays %d: no stamps (is RVB_LIB a -DRVB_STAMPS=1 build, RVB_STAMPS=1 set, RVB_PATH_LANES=2?)" % n)
            continue
        steps = dict(zip(["node", "leaf", "shading"], st[6:9]))
        print("rays %d: path kernel %.3f ms (stamped build), %d waves, %.0f shader cycles per wave in the loop, clock %.3f GHz (cycles / 100-MHz ticks)" % (n, path_ms, waves, loop / waves, loop / max(1, st[11]) * 0.1))
        if not sum(st[:6]):
            continue
        total = sum(st[:6])
        for k, name in enumerate(names):
            per = ""
            if name.startswith("node"): per = "  %.0f cycles per node step" % (st[k] / max(1, steps["node"]))
            if name.startswith("leaf"): per = "  %.0f cycles per leaf step" % (st[k] / max(1, steps["leaf"]))
            if name == "shading": per = "  %.0f cycles per shading step" % (st[k] / max(1, steps["shading"]))
            if name == "vote": per = "  %.0f cycles per loop iteration" % (st[k] / max(1, sum(steps.values())))
            print("   %-12s %5.1f %%%s" % (name, 100.0 * st[k] / total, per))
        print("   steps per wave: node %.0f leaf %.0f shading %.0f" % tuple(steps[k] / waves for k in ("node", "leaf", "shading")))


if __name__ == "__main__":
    main()
