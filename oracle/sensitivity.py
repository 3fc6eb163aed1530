#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (build container: needs /root/reference).

How much of "the reference OpenCL path" is defined by the parts OpenCL leaves to the implementation?  The parity definition of
this repo is the reference kernel text with CORRECTLY ROUNDED built-ins and no contraction (oracle/ref/ref_builtins.cl).  This
script builds the same kernel text a second time with another conforming choice — fused multiply-adds wherever `a * b + c`
appears (FP_CONTRACT is ON by default in OpenCL C) and in dot / cross / length, normalize as a multiply by the reciprocal square
root, pow and atan2 in binary32 — and reports, case by case, what changes:

  * discrete flips per impulse: a different hit point (another triangle won or the path forked), visible <-> blocked,
    a different time bin at 44.1 kHz;  rays whose path forked at some bounce (everything after that bounce differs);
  * the per-band-bin error of the binned impulse response (speaker model, fixPredelay, flattenImpulses) against the
    correctly rounded definition: relative to the band's largest value, and the share of band-bins outside 1e-5 relative.

    python oracle/sensitivity.py > profiles/rNN_builtin_sensitivity.json
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import pyoracle  # noqa: E402
from parallel_reverb_raytracer_amd import scenes  # noqa: E402
from parallel_reverb_raytracer_amd.dtypes import AIR_COEFFICIENTS, aligned_copy  # noqa: E402

SR = 44100.0
SPEAKERS = [((-1, 0, -1), 0.5), ((1, 0, -1), 0.5)]


def ir(oracle_port, oracle, mic, impulses):
    chans = [oracle.attenuate_speaker(mic, impulses, d, c) for d, c in SPEAKERS]
    pd = oracle_port.find_predelay(chans)
    flat = []
    for c in chans:
        oracle_port.fix_predelay(c, pd)
        flat.append(oracle_port.flatten(c, SR))
    n = max(f.shape[1] for f in flat)
    out = np.zeros((2, 8, n), np.float32)
    for ch, f in enumerate(flat):
        out[ch, :, :f.shape[1]] = f
    return out


def compare(name, scene, mic, source, dirs, nrefl, port, ref, alt):
    a, a_img, a_idx = ref.raytrace(scene, mic, source, dirs, nrefl, AIR_COEFFICIENTS)
    b, b_img, b_idx = alt.raytrace(scene, mic, source, dirs, nrefl, AIR_COEFFICIENTS)
    nrays = dirs.shape[0]
    pa, pb = a["position"][:, :3].reshape(nrays, nrefl, 3), b["position"][:, :3].reshape(nrays, nrefl, 3)
    moved = np.abs(pa - pb).max(axis=2) > 1e-3                      # another triangle, or a forked path
    forked = moved.any(axis=1)
    first_fork = np.where(forked, moved.argmax(axis=1), nrefl)
    before = np.arange(nrefl)[None, :] < first_fork[:, None]         # impulses of a ray up to its fork: comparable one to one
    ta, tb = a["time"].reshape(nrays, nrefl), b["time"].reshape(nrays, nrefl)
    vis_flip = ((ta == 0) != (tb == 0)) & before
    both = (ta != 0) & (tb != 0) & before
    bin_flip = (np.round(ta * np.float32(SR)) != np.round(tb * np.float32(SR))) & both
    va, vb = a["volume"].reshape(nrays, nrefl, 8), b["volume"].reshape(nrays, nrefl, 8)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(va - vb) / np.abs(va)
    rel = np.where(both[:, :, None] & (va != 0), rel, 0.0)
    ia, ib = ir(port, ref, mic, a), ir(port, alt, mic, b)
    n = max(ia.shape[2], ib.shape[2])
    ia, ib = (np.pad(x, ((0, 0), (0, 0), (0, n - x.shape[2]))).astype(np.float64) for x in (ia, ib))
    err = np.abs(ia - ib)
    band_max = np.abs(ia).max(axis=2, keepdims=True)
    nz = ia != 0
    outside = np.zeros(err.shape, bool)
    outside[nz] = err[nz] / np.abs(ia[nz]) > 1e-5
    outside |= ~nz & (err > 0)
    return {"case": name, "rays": int(nrays), "reflections": int(nrefl), "triangles": int(scene[0].shape[0]), "impulses": int(nrays * nrefl),
            "rays_whose_path_forked": int(forked.sum()), "median_bounce_of_fork": float(np.median(first_fork[forked])) if forked.any() else None,
            "impulses_before_any_fork": int(before.sum()),
            "visibility_flips_before_fork": int(vis_flip.sum()), "time_bin_flips_before_fork": int(bin_flip.sum()),
            "max_relative_volume_error_before_fork": float(rel.max()),
            "image_source_slots_differing": int((a_idx != b_idx).sum()),
            "ir_max_abs_err_over_band_max": float((err / np.maximum(band_max, 1e-300)).max()),
            "ir_fraction_band_bins_outside_1e-5_relative": float(outside.mean()), "ir_band_bins": int(err.size)}


def main():
    if not os.path.exists("/root/reference/rayverb/kernel.cpp"):
        sys.exit("needs the reference checkout (build container only)")
    subprocess.check_call([os.path.join(HERE, "ref", "build_ref.sh")], stdout=sys.stderr)
    subprocess.check_call([os.path.join(HERE, "ref", "build_ref.sh"), "alt"], stdout=sys.stderr)
    subprocess.check_call(["make", "-C", HERE, "_build/librvb_oracle.so"], stdout=sys.stderr)
    port, ref, alt = pyoracle.Oracle("port"), pyoracle.Oracle("reference"), pyoracle.Oracle("reference_alt")
    cases = []
    for name in ("trace_large_square", "trace_echo_tunnel", "trace_vault", "trace_random_pillars"):
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
        scene = (aligned_copy(g["triangles"]), aligned_copy(g["vertices"]), aligned_copy(g["surfaces"]))
        cases.append(compare("golden " + name, scene, g["mic"], g["source"], g["directions"], int(g["nreflections"]), port, ref, alt))
        print(cases[-1]["case"], "done", file=sys.stderr)
    scene, info = scenes.cathedral(75000)
    rays = int(os.environ.get("RVB_SENSITIVITY_RAYS", "384"))
    cases.append(compare("C2 sample (cathedral stand-in)", scene, info["mic"], info["source"], scenes.sphere_directions(rays, seed=1), 128, port, ref, alt))
    json.dump({"definition": "reference kernel text, correctly rounded built-ins, no contraction (oracle/ref/ref_builtins.cl)",
               "alternative": "same text with fused multiply-adds (FP_CONTRACT ON), rsqrt-multiply normalize, binary32 powf / atan2f",
               "cases": cases}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
