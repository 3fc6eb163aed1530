#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: regenerates tests/golden/*.npz in the build container.

Every *output* array in these fixtures comes from the reference's own OpenCL C kernel text
compiled for the host (oracle/_ref/librvb_ref.so, recipe oracle/ref/build_ref.sh) — i.e. from
the reference itself run here, as SURVEY.md §8(c) prescribes.  Inputs are either the reference's
demo assets parsed into arrays (data files the reference's own tests use:
tests/CMakeLists.txt:22-25 -> demo/assets/test_models/large_square.obj + materials/mat.json)
or seeded synthetic data.  Needs /root/reference; the committed .npz files are what travels.

    python oracle/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import pyoracle  # noqa: E402
from parallel_reverb_raytracer_amd import dtypes, scenes  # noqa: E402

REF = os.environ.get("RVB_REFERENCE_ROOT", "/root/reference")
ASSETS = os.path.join(REF, "demo", "assets")
OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def trace_case(ref, name, scene, mic, source, directions, nrefl):
    impulses, image, index = ref.raytrace(scene, mic, source, directions, nrefl, dtypes.AIR_COEFFICIENTS)
    save(name,
         triangles=scene[0], vertices=scene[1], surfaces=scene[2],
         mic=np.asarray(mic, np.float32), source=np.asarray(source, np.float32),
         directions=np.asarray(directions, np.float32), nreflections=np.int64(nrefl),
         air=dtypes.AIR_COEFFICIENTS,
         impulse_volume=impulses["volume"], impulse_position=impulses["position"][:, :3], impulse_time=impulses["time"],
         image_volume=image["volume"], image_position=image["position"][:, :3], image_time=image["time"],
         image_index=index)


def axis_impulses(n, seed):
    """The input of reference tests/attenuation_tests.h:21-31 / tests/hrtf_tests.cpp:9-20:
    unit-volume impulses at +-10 on each axis, then the origin; times seeded here."""
    rng = np.random.default_rng(seed)
    imp = dtypes.aligned_zeros(n, dtypes.IMPULSE)
    imp["volume"] = 1.0
    imp["position"][:6, :3] = [[-10, 0, 0], [10, 0, 0], [0, -10, 0], [0, 10, 0], [0, 0, -10], [0, 0, 10]]
    imp["time"] = rng.uniform(0, 100, n).astype(np.float32)
    return imp


def random_impulses(n, seed):
    rng = np.random.default_rng(seed)
    imp = dtypes.aligned_zeros(n, dtypes.IMPULSE)
    imp["volume"] = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
    imp["position"][:, :3] = rng.uniform(-30, 30, (n, 3)).astype(np.float32)
    imp["time"] = rng.uniform(0.001, 3.0, n).astype(np.float32)
    imp["volume"][::7] = 0.0                      # zero-volume impulses (quirk Q2: compared as {0,0})
    imp["position"][3, :3] = [0.5, -10.0, 0.25]   # nearly straight down
    return imp


def main():
    if not pyoracle.have_ref():
        sys.exit("oracle/_ref/librvb_ref.so missing: run `make -C oracle` in the build container first")
    os.makedirs(OUT, exist_ok=True)
    ref = pyoracle.Oracle("reference")

    # 1. the reference's RaytracerTest scene / positions (tests/raytrace_tests.cpp:6-17, raytrace_tests.h:23-25)
    sq = scenes.load_obj(os.path.join(ASSETS, "test_models", "large_square.obj"), os.path.join(ASSETS, "materials", "mat.json"))
    dirs = np.zeros((24, 4), np.float32)
    dirs[:6, :3] = [[0, 0, -1], [0, 0, 1], [0, -1, 0], [0, 1, 0], [-1, 0, 0], [1, 0, 0]]
    dirs[6:8, 2] = -1.0
    dirs[8:] = scenes.sphere_directions(16, seed=11)
    trace_case(ref, "trace_large_square", sq, (0, 2, 0), (0, 2, 2), dirs, 128)

    # 2. C1-like shoebox (demo echo_tunnel.obj, positions of demo/assets/configs/near_c.json)
    tunnel = scenes.load_obj(os.path.join(ASSETS, "test_models", "echo_tunnel.obj"), os.path.join(ASSETS, "materials", "mat.json"))
    trace_case(ref, "trace_echo_tunnel", tunnel, (0, 1, 2), (0, 1, 0), scenes.sphere_directions(96, seed=1), 16)

    # 3. 614-triangle scene with occluders (demo random_pillars.obj)
    pillars = scenes.load_obj(os.path.join(ASSETS, "test_models", "random_pillars.obj"), os.path.join(ASSETS, "materials", "mat.json"))
    trace_case(ref, "trace_random_pillars", pillars, (3.0, 1.5, -4.0), (-6.0, 1.6, 5.0), scenes.sphere_directions(48, seed=2), 24)

    # 4. 3754-triangle multi-material scene (demo vault.obj + vault.json materials, positions of configs/vault.json)
    vault = scenes.load_obj(os.path.join(ASSETS, "test_models", "vault.obj"), os.path.join(ASSETS, "materials", "vault.json"))
    trace_case(ref, "trace_vault", vault, (0, 1.75, 6), (0, 1.75, 0), scenes.sphere_directions(40, seed=3), 20)

    # 5. kernel attenuate: gtest input + seeded impulses, the three gtest speakers + an oblique one
    cases = {"axis": axis_impulses(64, 5), "random": random_impulses(512, 6)}
    speakers = [((0, 0, 1), 0.0), ((0, 0, 1), 0.5), ((0, 0, 1), 1.0), ((-1, 0, -1), 0.5)]
    out = {}
    for cname, imp in cases.items():
        mic = (0, 0, 0) if cname == "axis" else (1.0, 1.5, -2.0)
        out[cname + "_in_volume"], out[cname + "_in_position"], out[cname + "_in_time"] = imp["volume"], imp["position"][:, :3], imp["time"]
        out[cname + "_mic"] = np.asarray(mic, np.float32)
        for si, (d, c) in enumerate(speakers):
            att = ref.attenuate_speaker(mic, imp, d, c)
            nz = np.any(imp["volume"] != 0, axis=1)
            att["volume"][~nz] = 0
            att["time"][~nz] = 0
            out["%s_s%d_volume" % (cname, si)], out["%s_s%d_time" % (cname, si)] = att["volume"], att["time"]
    out["speaker_direction"] = np.asarray([s[0] for s in speakers], np.float32)
    out["speaker_coefficient"] = np.asarray([s[1] for s in speakers], np.float32)
    save("attenuate_speaker", **out)

    # 6. kernel hrtf with the regenerated test table: the four gtest head orientations + an oblique one
    table = scenes.hrtf_test_table()
    smooth = scenes.hrtf_synthetic_table()
    configs = [((0, 0, 1), (0, 1, 0)), ((1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0)), ((-1, 0, 0), (0, 1, 0)),
               ((0.6, 0.0, -0.8), (0, 1, 0))]
    out = {}
    for cname, imp in cases.items():
        mic = (0, 0, 0) if cname == "axis" else (1.0, 1.5, -2.0)
        out[cname + "_in_volume"], out[cname + "_in_position"], out[cname + "_in_time"] = imp["volume"], imp["position"][:, :3], imp["time"]
        out[cname + "_mic"] = np.asarray(mic, np.float32)
        for ci, (facing, up) in enumerate(configs):
            for ch in (0, 1):
                for tname, tab in (("test", table), ("smooth", smooth)):
                    att = ref.attenuate_hrtf(mic, imp, tab[ch], facing, up, ch)
                    nz = np.any(imp["volume"] != 0, axis=1)
                    att["volume"][~nz] = 0
                    att["time"][~nz] = 0
                    out["%s_c%d_ch%d_%s_volume" % (cname, ci, ch, tname)] = att["volume"]
                    out["%s_c%d_ch%d_%s_time" % (cname, ci, ch, tname)] = att["time"]
    out["facing"] = np.asarray([c[0] for c in configs], np.float32)
    out["up"] = np.asarray([c[1] for c in configs], np.float32)
    save("attenuate_hrtf", **out)


if __name__ == "__main__":
    main()
