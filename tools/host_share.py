#!/usr/bin/env python3
"""Developer tool: how much of the pipelined IR rate is host time?  Wall time and CPU time of this process for K pipelined IRs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
from parallel_reverb_raytracer_amd import capi, distributed, dtypes, scenes  # noqa: E402


def main():
    import torch
    mode = capi.IR_EXACT if (len(sys.argv) > 1 and sys.argv[1] == "exact") else capi.IR_FAST
    scene, info = scenes.cathedral(75000)
    dirs = torch.from_numpy(np.ascontiguousarray(scenes.sphere_directions(100000, seed=1))).cuda()
    ctxs = []
    for _ in range(4):
        c = capi.Context(0)
        c.set_scene(scene)
        c.set_directions_device(dirs.data_ptr(), 100000)
        ctxs.append(c)
    pipe = distributed.IrPipeline(ctxs)
    args = (info["mic"], info["source"], 128, dtypes.AIR_COEFFICIENTS)
    kw = dict(speakers_dir=[(-1, 0, -1), (1, 0, -1)], speakers_coeff=[0.5, 0.5], sample_rate=44100.0, trim_predelay=True, mode=mode,
              device=torch.device("cuda", 0))
    pipe.run(8, args, kw)
    torch.cuda.synchronize()
    w0, c0 = time.perf_counter(), time.process_time()
    pipe.run(40, args, kw)
    for c in ctxs:
        c.synchronize()
    torch.cuda.synchronize()
    w1, c1 = time.perf_counter(), time.process_time()
    print("40 IRs: wall %.2f ms per IR, CPU time of the host process %.2f ms per IR" % ((w1 - w0) * 25, (c1 - c0) * 25))


if __name__ == "__main__":
    main()
