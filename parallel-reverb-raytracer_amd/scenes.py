"""Seeded synthetic inputs for the hot path: scenes, ray directions, HRTF tables.

The scenes BASELINE.json names (Sibenik, Sponza, a concert hall) are not part of the
reference and not available offline; these generators build
closed stand-in meshes of matched triangle count.  Everything is deterministic.

Scene = (triangles[TRIANGLE], vertices[n][4] float32, surfaces[SURFACE]); surface 0 is the
reference's built-in default (reference rayverb/rayverb.cpp:336-341).
"""
import json
import math

import numpy as np

from .dtypes import SURFACE, TRIANGLE, aligned_zeros, float3_array

# reference rayverb/rayverb.cpp:336-339
DEFAULT_SURFACE = (
    [0.92, 0.92, 0.93, 0.93, 0.94, 0.95, 0.95, 0.95],
    [0.50, 0.90, 0.95, 0.95, 0.95, 0.95, 0.95, 0.95],
)

# Own stand-in materials (plausible octave-band reflectances; not taken from the reference).
STANDIN_MATERIALS = {
    "stone_floor": ([0.98, 0.98, 0.97, 0.97, 0.96, 0.95, 0.94, 0.93], [0.94, 0.90, 0.86, 0.82, 0.78, 0.72, 0.66, 0.60]),
    "lime_wall":   ([0.97, 0.96, 0.95, 0.95, 0.94, 0.93, 0.92, 0.91], [0.95, 0.92, 0.88, 0.84, 0.80, 0.74, 0.68, 0.62]),
    "plaster":     ([0.90, 0.92, 0.94, 0.95, 0.95, 0.94, 0.93, 0.92], [0.93, 0.90, 0.87, 0.83, 0.79, 0.73, 0.67, 0.61]),
    "marble":      ([0.99, 0.99, 0.98, 0.98, 0.98, 0.97, 0.97, 0.96], [0.96, 0.93, 0.89, 0.85, 0.80, 0.75, 0.69, 0.63]),
    "wood":        ([0.82, 0.86, 0.90, 0.91, 0.92, 0.92, 0.91, 0.90], [0.92, 0.90, 0.86, 0.82, 0.78, 0.72, 0.66, 0.60]),
    "glass":       ([0.75, 0.84, 0.90, 0.93, 0.95, 0.95, 0.95, 0.95], [0.95, 0.91, 0.87, 0.83, 0.79, 0.73, 0.67, 0.61]),
}


def make_surfaces(materials):
    """[default] + materials in sorted-name order, as reference rayverb/rayverb.cpp:341-354 does."""
    names = sorted(materials)                      # std::map<string, ...> order == bytewise order
    out = aligned_zeros(len(names) + 1, SURFACE)
    out[0]["specular"] = DEFAULT_SURFACE[0]
    out[0]["diffuse"] = DEFAULT_SURFACE[1]
    for i, n in enumerate(names):
        out[i + 1]["specular"] = materials[n][0]
        out[i + 1]["diffuse"] = materials[n][1]
    return out, {n: i + 1 for i, n in enumerate(names)}


class MeshBuilder:
    """Accumulates vertices / triangles; welds nothing (shared edges use identical coordinates)."""

    def __init__(self):
        self.verts = []
        self.tris = []
        self.nverts = 0

    def add(self, verts, tris, surface):
        verts = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
        tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
        self.verts.append(verts)
        t = np.empty((tris.shape[0], 4), dtype=np.int64)
        t[:, 0] = surface
        t[:, 1:] = tris + self.nverts
        self.tris.append(t)
        self.nverts += verts.shape[0]

    def grid(self, origin, du, dv, nu, nv, surface):
        """Planar nu x nv quad grid spanned by du, dv from origin (two triangles per quad)."""
        origin = np.asarray(origin, dtype=np.float64)
        du = np.asarray(du, dtype=np.float64)
        dv = np.asarray(dv, dtype=np.float64)
        us = np.linspace(0.0, 1.0, nu + 1)
        vs = np.linspace(0.0, 1.0, nv + 1)
        pts = origin[None, None, :] + us[:, None, None] * du[None, None, :] + vs[None, :, None] * dv[None, None, :]
        self.param_grid(pts, surface)

    def param_grid(self, pts, surface):
        """pts[nu+1][nv+1][3] -> quads split into two triangles."""
        nu, nv = pts.shape[0] - 1, pts.shape[1] - 1
        idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        tris = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
        self.add(pts.reshape(-1, 3), tris, surface)

    def box(self, lo, hi, surface, n=(1, 1, 1)):
        lo = np.asarray(lo, dtype=np.float64)
        hi = np.asarray(hi, dtype=np.float64)
        d = hi - lo
        ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
        self.grid(lo, ex, ez, n[0], n[2], surface)
        self.grid(lo + ey, ex, ez, n[0], n[2], surface)
        self.grid(lo, ex, ey, n[0], n[1], surface)
        self.grid(lo + ez, ex, ey, n[0], n[1], surface)
        self.grid(lo, ez, ey, n[2], n[1], surface)
        self.grid(lo + ex, ez, ey, n[2], n[1], surface)

    def prism(self, cx, cz, radius, y0, y1, sides, segs, surface, cap=True):
        """Vertical n-gon column."""
        ang = np.linspace(0.0, 2.0 * math.pi, sides + 1)
        ang[-1] = 0.0
        ys = np.linspace(y0, y1, segs + 1)
        pts = np.empty((sides + 1, segs + 1, 3))
        pts[:, :, 0] = cx + radius * np.cos(ang)[:, None]
        pts[:, :, 2] = cz + radius * np.sin(ang)[:, None]
        pts[:, :, 1] = ys[None, :]
        self.param_grid(pts, surface)
        if cap:
            ring = pts[:-1, -1, :]
            centre = np.array([[cx, y1, cz]])
            verts = np.concatenate([centre, ring])
            tris = [[0, 1 + i, 1 + (i + 1) % sides] for i in range(sides)]
            self.add(verts, tris, surface)

    def finish(self, surfaces):
        verts = np.concatenate(self.verts)
        tris = np.concatenate(self.tris)
        t = aligned_zeros(tris.shape[0], TRIANGLE)
        t["surface"], t["v0"], t["v1"], t["v2"] = tris[:, 0], tris[:, 1], tris[:, 2], tris[:, 3]
        return t, float3_array(verts), surfaces


def shoebox(w=4.0, h=7.0, d=100.0):
    """12-triangle box centred on the origin (shape of demo echo_tunnel.obj)."""
    surfaces, ids = make_surfaces({"wall": STANDIN_MATERIALS["lime_wall"]})
    m = MeshBuilder()
    m.box([-w / 2, -h / 2, -d / 2], [w / 2, h / 2, d / 2], ids["wall"])
    return m.finish(surfaces)


def rotated_square_room(n=1, half_diagonal=27.0, height=27.0):
    """The reference test room's shape (45-degree rotated square) with every face
    tessellated n x n: 12*n*n triangles."""
    surfaces, ids = make_surfaces({"FrontColor": ([0.99, 0.99, 0.99, 0.98, 0.98, 0.96, 0.96, 0.96],
                                                  [0.95, 0.9, 0.85, 0.8, 0.75, 0.7, 0.65, 0.6])})
    s = ids["FrontColor"]
    r = half_diagonal
    c = [np.array([-r, 0.0, 0.0]), np.array([0.0, 0.0, -r]), np.array([r, 0.0, 0.0]), np.array([0.0, 0.0, r])]
    up = np.array([0.0, height, 0.0])
    m = MeshBuilder()
    m.grid(c[0], c[1] - c[0], c[3] - c[0], n, n, s)
    m.grid(c[0] + up, c[1] - c[0], c[3] - c[0], n, n, s)
    for i in range(4):
        a, b = c[i], c[(i + 1) % 4]
        m.grid(a, b - a, up, n, n, s)
    return m.finish(surfaces)


def cathedral(target_triangles=75000):
    """Closed cathedral-like stand-in for config C2/C3: nave with barrel vault, two column rows,
    side chapels' altars and pew blocks.  ~60 x 27 x 24 m.  Returns the scene and a dict with
    suggested source / mic positions."""
    surfaces, ids = make_surfaces(STANDIN_MATERIALS)
    L, W, HW = 60.0, 24.0, 15.0          # length (x), width (z), wall height (vault springs from here)

    def build(s):
        m = MeshBuilder()
        nx, nz, ny = max(2, int(40 * s)), max(2, int(16 * s)), max(2, int(10 * s))
        m.grid([-L / 2, 0, -W / 2], [L, 0, 0], [0, 0, W], nx, nz, ids["stone_floor"])
        m.grid([-L / 2, 0, -W / 2], [L, 0, 0], [0, HW, 0], nx, ny, ids["lime_wall"])
        m.grid([-L / 2, 0, W / 2], [L, 0, 0], [0, HW, 0], nx, ny, ids["lime_wall"])
        m.grid([-L / 2, 0, -W / 2], [0, 0, W], [0, HW, 0], nz, ny, ids["lime_wall"])
        m.grid([L / 2, 0, -W / 2], [0, 0, W], [0, HW, 0], nz, ny, ids["glass"])
        # barrel vault: half cylinder of radius W/2 along x, springing at y = HW
        na = max(4, int(24 * s))
        ang = np.linspace(0.0, math.pi, na + 1)
        xs = np.linspace(-L / 2, L / 2, nx + 1)
        cz = (W / 2) * np.cos(ang)
        cy = HW + (W / 2) * np.sin(ang)
        cz[0], cz[-1], cy[0], cy[-1] = W / 2, -W / 2, HW, HW
        pts = np.empty((nx + 1, na + 1, 3))
        pts[:, :, 0] = xs[:, None]
        pts[:, :, 1] = cy[None, :]
        pts[:, :, 2] = cz[None, :]
        m.param_grid(pts, ids["plaster"])
        for x in (-L / 2, L / 2):             # tympana closing the vault ends
            ring = np.stack([np.full(na + 1, x), cy, cz], -1)
            verts = np.concatenate([[[x, HW, 0.0]], ring])
            m.add(verts, [[0, 1 + i, 2 + i] for i in range(na)], ids["lime_wall"])
        # two rows of columns
        ncol = 8
        sides, segs = max(6, int(12 * s)), max(2, int(10 * s))
        for i in range(ncol):
            x = -L / 2 + (i + 0.5) * L / ncol
            for z in (-W / 4, W / 4):
                m.prism(x, z, 0.6, 0.0, 11.0, sides, segs, ids["marble"])
        # pew blocks and an altar
        npew = max(2, int(14 * s))
        for i in range(npew):
            x = -L / 2 + 6.0 + i * (L - 20.0) / npew
            for z0, z1 in ((-5.0, -1.0), (1.0, 5.0)):
                m.box([x, 0.0, z0], [x + 0.5, 0.9, z1], ids["wood"], n=(1, 1, max(1, int(3 * s))))
        m.box([L / 2 - 6.0, 0.0, -2.0], [L / 2 - 4.0, 1.2, 2.0], ids["marble"], n=(2, 2, 4))
        return m

    lo, hi = 0.2, 8.0
    for _ in range(30):                       # bisection on the tessellation scale
        mid = 0.5 * (lo + hi)
        if sum(t.shape[0] for t in build(mid).tris) < target_triangles:
            lo = mid
        else:
            hi = mid
    scene = build(hi).finish(surfaces)
    info = {"source": (-18.0, 1.7, 0.7), "mic": (14.0, 1.6, -0.9), "bounds": ((-L / 2, 0, -W / 2), (L / 2, HW + W / 2, W / 2))}
    return scene, info


def atrium(target_triangles=262000):
    """High-poly stand-in for config C4 (Sponza-like): two-storey atrium with galleries,
    many columns and small props (divergence stress)."""
    surfaces, ids = make_surfaces(STANDIN_MATERIALS)
    L, W, H = 36.0, 16.0, 14.0

    def build(s):
        m = MeshBuilder()
        nx, nz, ny = max(2, int(36 * s)), max(2, int(16 * s)), max(2, int(14 * s))
        m.box([-L / 2, 0, -W / 2], [L / 2, H, W / 2], ids["lime_wall"], n=(nx, ny, nz))
        # galleries along both long walls at mid height
        for z0, z1 in ((-W / 2, -W / 2 + 3.0), (W / 2 - 3.0, W / 2)):
            m.box([-L / 2 + 0.01, 6.0, z0 + (0.01 if z0 < 0 else 0.0)], [L / 2 - 0.01, 6.4, z1 - (0.01 if z1 > 0 else 0.0)],
                  ids["stone_floor"], n=(nx, 1, max(1, int(3 * s))))
        ncol = 12
        sides, segs = max(6, int(10 * s)), max(2, int(6 * s))
        for i in range(ncol):
            x = -L / 2 + (i + 0.5) * L / ncol
            for z in (-W / 2 + 3.0, W / 2 - 3.0):
                m.prism(x, z, 0.35, 0.0, 6.0, sides, segs, ids["marble"], cap=False)
                m.prism(x, z, 0.3, 6.4, 12.0, sides, segs, ids["marble"])
        rng = np.random.default_rng(7)
        nprops = max(4, int(60 * s * s))
        for _ in range(nprops):                # small props on the floor
            x, z = rng.uniform(-L / 2 + 1, L / 2 - 2), rng.uniform(-4.0, 3.0)
            w, h = rng.uniform(0.2, 0.8), rng.uniform(0.3, 1.5)
            m.box([x, 0.0, z], [x + w, h, z + w], ids["wood"], n=(1, 1, 1))
        return m

    lo, hi = 0.2, 12.0
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        if sum(t.shape[0] for t in build(mid).tris) < target_triangles:
            lo = mid
        else:
            hi = mid
    scene = build(hi).finish(surfaces)
    info = {"source": (-10.0, 1.6, 0.3), "mic": (9.0, 1.5, -0.4)}
    return scene, info


def concert_hall(target_triangles=30000):
    """Stand-in for config C5: fan-less shoebox hall with raked audience blocks and a stage."""
    surfaces, ids = make_surfaces(STANDIN_MATERIALS)
    L, W, H = 40.0, 22.0, 16.0

    def build(s):
        m = MeshBuilder()
        nx, nz, ny = max(2, int(30 * s)), max(2, int(16 * s)), max(2, int(12 * s))
        m.box([-L / 2, 0, -W / 2], [L / 2, H, W / 2], ids["plaster"], n=(nx, ny, nz))
        m.box([-L / 2 + 0.01, 0.0, -W / 2 + 2.0], [-L / 2 + 8.0, 1.1, W / 2 - 2.0], ids["wood"], n=(max(1, int(6 * s)), 1, nz))
        nrows = max(2, int(16 * s))
        for i in range(nrows):
            x = -L / 2 + 10.0 + i * (L - 14.0) / nrows
            m.box([x, 0.0, -W / 2 + 1.5], [x + 0.8, 0.5 + 0.12 * i, W / 2 - 1.5], ids["wood"], n=(1, 1, max(1, int(8 * s))))
        return m

    lo, hi = 0.2, 12.0
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        if sum(t.shape[0] for t in build(mid).tris) < target_triangles:
            lo = mid
        else:
            hi = mid
    scene = build(hi).finish(surfaces)
    return scene, {"bounds": ((-L / 2, 0, -W / 2), (L / 2, H, W / 2))}


def source_mic_pairs(n, seed=0):
    """Seeded (source, mic) pairs inside the concert_hall volume (config C5)."""
    rng = np.random.default_rng(seed)
    src = np.stack([rng.uniform(-18.5, -13.0, n), rng.uniform(1.3, 2.2, n), rng.uniform(-7.0, 7.0, n)], -1)
    mic = np.stack([rng.uniform(-8.0, 17.0, n), rng.uniform(2.6, 4.0, n), rng.uniform(-8.5, 8.5, n)], -1)
    return src.astype(np.float32), mic.astype(np.float32)


# ---------------------------------------------------------------------------------------
# ray directions: spherePoint(z, theta) of reference rayverb/helpers.cpp:63-67 with
# z ~ U[-1,1), theta ~ U[-pi,pi) from a counter-based generator (the reference seeds
# std::default_random_engine from the wall clock, helpers.cpp:74-75: not reproducible).

def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def sphere_directions(n, seed=1, first=0):
    """Directions first .. first+n of the seeded stream, [n][4] float32.  `first` lets a rank
    generate its own contiguous shard of the global ray set without generating the rest."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + n, dtype=np.uint64)
        base = np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
        a = _splitmix64(base + np.uint64(2) * i)
        b = _splitmix64(base + np.uint64(2) * i + np.uint64(1))
    u = (a >> np.uint64(40)).astype(np.float64) / float(1 << 24)      # [0,1) with 24 bits
    v = (b >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    z = (2.0 * u - 1.0).astype(np.float32)
    theta = ((2.0 * v - 1.0) * math.pi).astype(np.float32)
    ztemp = np.sqrt(np.float32(1.0) - z * z).astype(np.float32)
    d = np.stack([ztemp * np.cos(theta), ztemp * np.sin(theta), z], -1).astype(np.float32)
    return float3_array(d)


# ---------------------------------------------------------------------------------------
# HRTF tables: [2][360][180][8] float32 (reference rayverb/rayverb.h:255-257)

def hrtf_test_table():
    """The reference's *test* table, regenerated from its generating rule
    (reference hrtf_analysis/generate_test_hrtf_data.py:5-8 and analyse_hrtf.py:28-109):
    grid points every 15 degrees hold their own (azimuth, elevation) in every band pair
    and the rest is filled bilinearly.  Used only for index-selection tests; entry
    [ch][a][e] for grid points is (a, e) repeated."""
    a = np.arange(360, dtype=np.float64)
    e = np.arange(180, dtype=np.float64)
    a_min = 15.0 * np.floor(a / 15.0)
    a_max = a_min + 15.0
    e_min = 15.0 * np.floor(e / 15.0)
    e_max = e_min + 15.0
    a_ratio = (a - a_min) / (a_max - a_min)
    e_ratio = (e - e_min) / (e_max - e_min)
    v_lo = a_min % 360.0                     # get_entry looks the azimuth up modulo 360 ...
    v_hi = a_max % 360.0                     # ... so the 360-degree grid column holds azimuth 0
    band0 = v_lo + (v_hi - v_lo) * a_ratio   # same for both elevation corners
    band1 = e_min + (e_max - e_min) * e_ratio
    table = np.zeros((2, 360, 180, 8), dtype=np.float32)
    table[:, :, :, 0] = band0[None, :, None]
    table[:, :, :, 1] = band1[None, None, :]
    return table


def hrtf_synthetic_table():
    """Smooth analytic 8-band head-shadow pattern (stand-in: the reference's IRCAM-derived
    HRTF_DATA blob is missing, SURVEY.md §8(c) gap 1 — HRTF *values* are parity-unpinned)."""
    a = np.deg2rad(np.arange(360, dtype=np.float64) - 180.0)[:, None, None]
    e = np.deg2rad(90.0 - np.arange(180, dtype=np.float64))[None, :, None]
    band = np.arange(8, dtype=np.float64)[None, None, :]
    shadow = 0.08 * (band + 1.0)
    table = np.empty((2, 360, 180, 8), dtype=np.float32)
    for ch, sign in ((0, -1.0), (1, 1.0)):
        lateral = sign * np.sin(a) * np.cos(e)
        table[ch] = (0.55 + 0.45 * np.exp(-shadow * (1.0 - lateral))).astype(np.float32)
    return table


# ---------------------------------------------------------------------------------------
# Wavefront OBJ + material JSON reader used to turn the reference's demo assets into test
# fixtures.  (The product's loader is the C++ one; this one only feeds tests and bench.)

def load_obj(obj_path, material_json_path):
    materials = {}
    with open(material_json_path) as f:
        for name, entry in json.load(f).items():
            materials[name] = (entry["specular"], entry["diffuse"])
    surfaces, ids = make_surfaces(materials)
    verts, tris = [], []
    current = 0
    with open(obj_path) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v":
                verts.append([float(x) for x in parts[1:4]])
            elif parts[0] == "usemtl":
                current = ids.get(parts[1], 0)
            elif parts[0] == "f":
                idx = [int(p.split("/")[0]) for p in parts[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):          # fan (convex polygons only)
                    tris.append([current, idx[0], idx[k], idx[k + 1]])
    tris = np.asarray(tris, dtype=np.int64)
    t = aligned_zeros(tris.shape[0], TRIANGLE)
    t["surface"], t["v0"], t["v1"], t["v2"] = tris[:, 0], tris[:, 1], tris[:, 2], tris[:, 3]
    return t, float3_array(np.asarray(verts)), surfaces
