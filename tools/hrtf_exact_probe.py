#!/usr/bin/env python3
"""Developer tool: the binning stage of BASELINE config C5 (hall stand-in, HRTF, 100k rays x 128) a few times in a row, for a
rocprofv3 --kernel-trace --stats pass or for the library's own event timings.
    python tools/hrtf_exact_probe.py [pair] [fast]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
import torch  # noqa: E402
from parallel_reverb_raytracer_amd import capi, dtypes, scenes  # noqa: E402


def main():
    pair = int(sys.argv[1]) if len(sys.argv) > 1 else 31
    mode = capi.IR_FAST if (len(sys.argv) > 2 and sys.argv[2] == "fast") else capi.IR_EXACT
    scene, _ = scenes.concert_hall(30000)
    src, mic = scenes.source_mic_pairs(64, seed=0)
    table = scenes.hrtf_synthetic_table()
    ctx = capi.Context(0)
    ctx.set_scene(scene)
    ctx.set_directions(scenes.sphere_directions(100000, seed=1))
    facing = src[pair] - mic[pair]
    facing = facing / np.linalg.norm(facing)
    ctx.trace(mic[pair], src[pair], 128, dtypes.AIR_COEFFICIENTS)
    images = capi.merge_images(ctx.get_image_candidates(), ctx.get_direct(), False)
    ctx.ir_configure_hrtf(mic[pair], table, facing, (0, 1, 0), capi.IR_ALL, images)
    lo, hi = ctx.ir_time_range()
    nbins = ctx.ir_bins(hi, lo, 44100.0)
    times = []
    for _ in range(6):
        hist = torch.zeros((2, 8, nbins), device="cuda", dtype=torch.float32)
        ctx.ir_accumulate_tensor(lo, 44100.0, nbins, mode, hist)
        ctx.synchronize()
        times.append(dict(ctx.last_timings()))
    audible = int((hist != 0).any(dim=0).any(dim=0).sum())
    print("pair %d nbins %d bins with sound %d | %s" % (pair, nbins, audible,
          " ".join("%s %.3f" % (k, float(np.median([t[k] for t in times[1:]]))) for k in times[0])))


if __name__ == "__main__":
    main()
