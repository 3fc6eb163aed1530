#!/usr/bin/env python3
"""Developer probe: throughput of the whole IR generation when T contexts on one GPU work on independent IRs
concurrently (one host thread per context), against one context doing them back to back."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
import torch  # noqa: E402
from parallel_reverb_raytracer_amd import capi, distributed, dtypes, scenes  # noqa: E402


def main():
    nrays, nrefl = 100000, 128
    scene, info = scenes.cathedral(75000)
    mic, src = info["mic"], info["source"]
    dirs = torch.from_numpy(np.ascontiguousarray(scenes.sphere_directions(nrays, seed=1))).cuda()
    torch.cuda.synchronize()
    for nthreads in (1, 2, 3):
        ctxs = []
        for _ in range(nthreads):
            c = capi.Context(0)
            c.set_scene(scene)
            c.set_directions_device(dirs.data_ptr(), nrays)
            ctxs.append(c)

        def work(c, count):
            for _ in range(count):
                distributed.generate_ir(c, mic, src, nrefl, dtypes.AIR_COEFFICIENTS, [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], 44100.0,
                                        trim_predelay=True, mode=capi.IR_FAST, device=torch.device("cuda", 0))

        for c in ctxs:
            work(c, 2)                                  # warm-up (allocations)
        torch.cuda.synchronize()
        total = 12
        per = total // nthreads
        t0 = time.perf_counter()
        threads = [threading.Thread(target=work, args=(c, per)) for c in ctxs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%d context(s): %d IRs in %.2f ms = %.3f ms per IR, %.3g ray-bounces/s" % (nthreads, per * nthreads, dt * 1e3, dt * 1e3 / (per * nthreads), per * nthreads * nrays * nrefl / dt))
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
