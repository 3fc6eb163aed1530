#!/usr/bin/env python3
"""Developer tool: raw dumps of a stand-in scene and its seeded ray directions for tools/travsim.cpp / tools/travforms.cpp.

    tools/dump_scene.py <cathedral|atrium|hall> <triangles> <nrays> <outdir>     -> tris.bin verts.bin dirs.bin + source / mic on stdout
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rvb_import  # noqa: E402

rvb_import.load()
from parallel_reverb_raytracer_amd import scenes  # noqa: E402

kind, ntri, nrays, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
scene, info = {"cathedral": scenes.cathedral, "atrium": scenes.atrium, "hall": scenes.concert_hall}[kind](ntri)
os.makedirs(out, exist_ok=True)
scene[0].tofile(os.path.join(out, "tris.bin"))
np.ascontiguousarray(scene[1], dtype=np.float32).tofile(os.path.join(out, "verts.bin"))
np.ascontiguousarray(scenes.sphere_directions(nrays, seed=1), dtype=np.float32).tofile(os.path.join(out, "dirs.bin"))
print("triangles %d source %s mic %s" % (scene[0].shape[0], " ".join("%r" % float(x) for x in info["source"]), " ".join("%r" % float(x) for x in info["mic"])))
