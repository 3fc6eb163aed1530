#!/usr/bin/env python3
"""Developer tool: the premise of a per-wavefront histogram tile (north_star technique N4) measured on workload C2 — how many of
the 64 impulses a wave of histogram_fast_kernel stages share a time bin with another impulse of the same 64?  A shuffle / DPP
reduce inside the wave before the atomic flush saves exactly those atomics.
    python tools/duplicate_bins.py > gpurun_out/<tag>_duplicate_bins.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
from parallel_reverb_raytracer_amd import capi, dtypes, scenes  # noqa: E402


def main():
    nrays, nrefl, sr = 100000, 128, np.float32(44100.0)
    scene, info = scenes.cathedral(75000)
    ctx = capi.Context(0)
    ctx.set_scene(scene)
    ctx.raytrace(info["mic"], info["source"], scenes.sphere_directions(nrays, seed=1), nrefl, dtypes.AIR_COEFFICIENTS)
    d = ctx.get_raw_diffuse()
    nonzero = (d["volume"] != 0).any(axis=1)
    t = d["time"]
    predelay = t[nonzero & (t != 0)].min()
    bins = np.round(np.where(t > predelay, t - predelay, np.float32(0)) * sr).astype(np.int64)
    bins[~nonzero] = -1 - np.arange((~nonzero).sum())               # zero-volume impulses issue no atomics: make them unique
    groups = bins[: (bins.shape[0] // 64) * 64].reshape(-1, 64)
    s = np.sort(groups, axis=1)
    dup = (s[:, 1:] == s[:, :-1]) & (s[:, 1:] >= 0)                  # each True = one atomic a wave-level reduce would save
    saved = int(dup.sum())
    issued = int(nonzero[: groups.size].sum())
    print(json.dumps({"workload": "C2: 100k rays x 128, cathedral stand-in, 44.1 kHz, trim_predelay",
                      "impulses": int(bins.shape[0]), "impulses_with_volume": issued, "bins": int(bins.max() + 1),
                      "waves_of_64": int(groups.shape[0]), "impulses_sharing_a_bin_within_their_wave": saved,
                      "fraction_of_atomics_a_wave_reduce_would_save": saved / max(1, issued),
                      "waves_with_any_shared_bin": float(dup.any(axis=1).mean())}))


if __name__ == "__main__":
    main()
