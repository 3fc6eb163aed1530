#!/usr/bin/env python3
"""Developer tool / bench.py helper: dumps the C2 workload as raw arrays and runs parallel-reverb-raytracer_amd/_build/api_flow on it."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
from parallel_reverb_raytracer_amd import scenes  # noqa: E402


def run(nrays=100000, nrefl=128, ntri=75000, repeats=3, env=None):
    exe = os.path.join(ROOT, "parallel-reverb-raytracer_amd", "_build", "api_flow")
    if not os.path.exists(exe):
        raise RuntimeError("%s not built (make -C parallel-reverb-raytracer_amd)" % exe)
    scene, info = scenes.cathedral(ntri)
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        paths = []
        for name, arr in (("tris", scene[0]), ("verts", scene[1]), ("surfaces", scene[2]), ("dirs", scenes.sphere_directions(nrays, seed=1))):
            paths.append(os.path.join(d, name + ".bin"))
            np.ascontiguousarray(arr).tofile(paths[-1])
        args = [exe] + paths + [str(nrefl)] + ["%r" % float(x) for x in info["source"]] + ["%r" % float(x) for x in info["mic"]] + [str(repeats)]
        out = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, **(env or {})), timeout=600)
    if out.returncode != 0:
        raise RuntimeError("api_flow failed: " + out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    print(json.dumps(run(*[int(x) for x in sys.argv[1:4]])))
