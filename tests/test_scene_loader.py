"""the Wavefront OBJ + material JSON loader that replaces Assimp for `Raytracer(obj, mat)`.
Assimp's triangle order is third-party and unpinned (SURVEY §8(c) gap 2); what is checked here is what any
correct triangulation must satisfy: n-2 triangles per polygon, total area equal to the polygons' area (the concave
polygons of the reference's bedroom.obj included: a fan would get this wrong), material -> surface mapping with
surface 0 as default and the file's materials in bytewise name order (reference rayverb.cpp:336-354)."""
import json
import os
import subprocess

import numpy as np

from conftest import ROOT

PKG = os.path.join(ROOT, "parallel-reverb-raytracer_amd")
ASSETS = os.path.join(ROOT, "tests", "golden", "assets")


def _tool():
    out = os.path.join(ROOT, "tests", "cpp", "_build", "scene_loader_tool")
    subprocess.check_call(["make", "-C", PKG, "-j4"], stdout=subprocess.DEVNULL)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "shims"),
                           os.path.join(ROOT, "tests", "cpp", "scene_loader_tool.cpp"), "-o", out, "-L" + PKG, "-lrayverb", "-lrvb_hip",
                           "-Wl,-rpath," + PKG])
    return out


def _polygons(path):
    verts, polys, mats, cur = [], [], [], None
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            verts.append([float(x) for x in p[1:4]])
        elif p[0] == "usemtl":
            cur = p[1]
        elif p[0] == "f":
            polys.append([int(q.split("/")[0]) - 1 for q in p[1:]])
            mats.append(cur)
    return np.asarray(verts), polys, mats


def _polygon_area(v, poly):
    n = np.zeros(3)
    for i in range(len(poly)):                      # Newell: exact for planar polygons, concave or not
        n += np.cross(v[poly[i]], v[poly[(i + 1) % len(poly)]])
    return 0.5 * np.linalg.norm(n)


def _run(tool, obj, mat):
    out = subprocess.run([tool, os.path.join(ASSETS, obj), os.path.join(ASSETS, mat)], capture_output=True, text=True, check=True).stdout.splitlines()
    head = out[0].split()
    per = {int(line.split()[1]): (line.split()[2], int(line.split()[3])) for line in out[1:]}
    return int(head[1]), int(head[3]), int(head[5]), float(head[7]), per


def test_bedroom_concave_polygons_are_ear_clipped():
    tool = _tool()
    v, polys, mats = _polygons(os.path.join(ASSETS, "bedroom.obj"))
    ntri, nvert, nsurf, area, per = _run(tool, "bedroom.obj", "mat.json")
    assert ntri == sum(len(p) - 2 for p in polys) == 88
    assert nvert == len(v) and nsurf == 42                          # default + the 41 entries of mat.json
    np.testing.assert_allclose(area, sum(_polygon_area(v, p) for p in polys), rtol=1e-6)
    names = sorted(json.load(open(os.path.join(ASSETS, "mat.json"))))
    for idx, (name, count) in per.items():
        assert idx > 0 and names[idx - 1] == name
        assert count == sum(len(p) - 2 for p, m in zip(polys, mats) if m == name)


def test_unknown_material_falls_back_to_default_surface(tmp_path):
    tool = _tool()
    (tmp_path / "m.json").write_text('{"other": {"specular": [1,1,1,1,1,1,1,1], "diffuse": [1,1,1,1,1,1,1,1]}}')
    out = subprocess.run([tool, os.path.join(ASSETS, "large_square.obj"), str(tmp_path / "m.json")], capture_output=True, text=True, check=True).stdout
    assert "triangles 12" in out and "surface 0 (default) 12" in out
    bad = subprocess.run([tool, os.path.join(ASSETS, "large_square.obj"), os.path.join(ASSETS, "large_square.obj")], capture_output=True, text=True)
    assert bad.returncode == 1 and "error" in bad.stdout
