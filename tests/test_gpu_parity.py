"""GPU parity tests: the HIP path, called through the C-ABI, against the golden vectors of the
reference's own kernels and against the CPU oracle on seeded inputs.

Bars (BASELINE.json north_star: 1e-5 relative per band-bin; tighter here where the arithmetic allows):
  * positions, times, triangle indices, image-source keys, bin indices: bit-exact;
  * volumes: bit-exact (pow is evaluated in binary64 and rounded once on both sides — a mismatch of
    1 ULP is possible in principle with probability ~1e-8 per value and would fail loudly here);
  * exact-mode histograms: bit-exact with the reference's serial summation;
  * fast-mode (float atomics) histograms: |diff| <= 1e-5 * |ref| + n_bin * 2^-23 * sum|terms| per band-bin.
"""
import numpy as np
import pytest

from conftest import golden_impulses, golden_scene, load_golden
from parallel_reverb_raytracer_amd import dtypes, scenes
from parallel_reverb_raytracer_amd.dtypes import AIR_COEFFICIENTS, IMPULSE, NUM_IMAGE_SOURCE

pytestmark = pytest.mark.gpu

TRACE_CASES = ["trace_large_square", "trace_echo_tunnel", "trace_random_pillars", "trace_vault"]


@pytest.fixture(scope="module", params=["four_lanes_per_ray", "two_lanes_per_ray", "one_lane_per_ray"])
def ctx(request):
    """Every test of this module runs with all three path kernels: the quad kernel (what a small launch gets), the pair kernel
    (forced here by announcing many concurrent traces, rvb_set_concurrent_traces) and the one-lane kernel of round 4
    (rvb_set_path_lanes)."""
    from parallel_reverb_raytracer_amd import capi
    c = capi.Context(0)          # raises when librvb_hip.so or the GPU is missing: no fallback
    if request.param == "two_lanes_per_ray":
        c.set_concurrent_traces(1 << 20)
    if request.param == "one_lane_per_ray":
        c.set_path_lanes(1)
    yield c
    c.close()


def assert_impulses_equal(got, want, what=""):
    assert got.shape == want.shape, what
    assert np.array_equal(got["position"][:, :3], want["position"][:, :3]), what + " positions"
    assert np.array_equal(got["time"], want["time"]), what + " times"
    assert np.array_equal(got["volume"], want["volume"]), what + " volumes"


@pytest.mark.parametrize("name", TRACE_CASES)
def test_trace_matches_reference_kernel_golden(ctx, oracle, name):
    g = load_golden(name)
    nrefl = int(g["nreflections"])
    ctx.set_scene(golden_scene(g))
    ctx.raytrace(g["mic"], g["source"], g["directions"], nrefl, g["air"])
    got = ctx.get_raw_diffuse()
    assert np.array_equal(got["position"][:, :3], g["impulse_position"])
    assert np.array_equal(got["time"], g["impulse_time"])
    assert np.array_equal(got["volume"], g["impulse_volume"])
    assert not got["pad"].any() and not got["position"][:, 3].any()
    # image sources: the reference's per-ray slots -> its de-dup map (restated in the oracle) vs ours
    nrays = g["directions"].shape[0]
    image = np.zeros(nrays * NUM_IMAGE_SOURCE, dtype=IMPULSE)
    image["volume"], image["time"] = g["image_volume"], g["image_time"]
    image["position"][:, :3] = g["image_position"]
    for remove_direct in (False, True):
        want = oracle.collect_images(image, g["image_index"], remove_direct)
        assert_impulses_equal(ctx.get_raw_images(remove_direct), want, "images")
    # every valid per-ray slot individually
    cand = ctx.get_image_candidates()
    idx = g["image_index"].reshape(nrays, NUM_IMAGE_SOURCE)
    rays, slots = np.nonzero(idx[:, 1:])
    assert cand.shape[0] == rays.shape[0]
    assert np.array_equal(cand["ray"], rays) and np.array_equal(cand["slot"], slots + 1)
    assert np.array_equal(cand["index"], idx[rays, slots + 1])
    assert_impulses_equal(cand["impulse"], image.reshape(nrays, NUM_IMAGE_SOURCE)[rays, slots + 1], "candidates")
    assert_impulses_equal(ctx.get_direct(), image[:1], "direct")


def test_reference_gtest_known_answers(ctx):
    """reference tests/raytrace_tests.h:35-47 through the HIP path."""
    g = load_golden("trace_large_square")
    ctx.set_scene(golden_scene(g))
    ctx.raytrace(g["mic"], g["source"], g["directions"], 128, g["air"])
    pos = ctx.get_raw_diffuse()["position"].reshape(-1, 128, 4)[:, :, :3]
    bounce0 = [(0, 2, -27), (0, 2, 27), (0, 0, 2), (0, 27, 2), (-25, 2, 2), (25, 2, 2)]
    bounce1 = [(0, 0, 0), (0, 0, 0), (0, 27, 2), (0, 0, 2), (-25, 2, -2), (25, 2, -2)]
    for r in range(6):
        np.testing.assert_array_almost_equal_nulp(pos[r, 0], np.float32(bounce0[r]), nulp=4)
        np.testing.assert_array_almost_equal_nulp(pos[r, 1] + np.float32(64), np.float32(bounce1[r]) + np.float32(64), nulp=4)


SEEDED = [
    ("room_768", lambda: (scenes.rotated_square_room(n=8), (0.5, 2.0, 0.25), (-3.0, 4.0, 2.0)), 512, 24),
    ("cathedral_3k", lambda: (scenes.cathedral(3000)[0], (14.0, 1.6, -0.9), (-18.0, 1.7, 0.7)), 384, 20),
    ("atrium_2k", lambda: (scenes.atrium(2000)[0], (9.0, 1.5, -0.4), (-10.0, 1.6, 0.3)), 256, 12),
    ("shoebox", lambda: (scenes.shoebox(), (0.0, 1.0, 2.0), (0.0, 1.0, 0.0)), 1000, 16),     # config C1 at full size
]


@pytest.mark.parametrize("name,make,nrays,nrefl", SEEDED, ids=[s[0] for s in SEEDED])
def test_trace_matches_oracle_on_seeded_scenes(ctx, oracle, name, make, nrays, nrefl):
    """BVH traversal must return exactly the brute-force winner (SURVEY §8(a) R2 tie rule)."""
    scene, mic, src = make()
    dirs = scenes.sphere_directions(nrays, seed=17)
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    want, image, index = oracle.raytrace(scene, mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    assert_impulses_equal(ctx.get_raw_diffuse(), want, name)
    for remove_direct in (False, True):
        assert_impulses_equal(ctx.get_raw_images(remove_direct), oracle.collect_images(image, index, remove_direct), name + " images")
    executed = int(np.count_nonzero(want["position"][:, :3].any(axis=1) | (want["time"] != 0) | want["volume"].any(axis=1)))
    assert ctx.executed_bounces() >= executed


def test_quad_shadow_kernel_gives_the_same_bytes(ctx):
    """The shadow kernel runs two lanes per record; the four-lane kernel it replaced stays in the library for measurements
    (RVB_SHADOW_LANES=4, read once per process).  A child process traces a seeded scene with it: same bytes."""
    import os
    import subprocess
    import sys
    import zlib
    scene, info = scenes.cathedral(3000)
    dirs = scenes.sphere_directions(700, seed=23)
    ctx.set_scene(scene)
    ctx.raytrace(info["mic"], info["source"], dirs, 40, AIR_COEFFICIENTS)
    mine = zlib.crc32(ctx.get_raw_diffuse().tobytes())
    code = ("import sys, zlib; sys.path.insert(0, %r); import rvb_import; rvb_import.load();"
            "from parallel_reverb_raytracer_amd import capi, scenes;"
            "from parallel_reverb_raytracer_amd.dtypes import AIR_COEFFICIENTS;"
            "scene, info = scenes.cathedral(3000); c = capi.Context(0); c.set_scene(scene);"
            "c.raytrace(info['mic'], info['source'], scenes.sphere_directions(700, seed=23), 40, AIR_COEFFICIENTS);"
            "print('CRC', zlib.crc32(c.get_raw_diffuse().tobytes()), dict(c.last_timings()).keys())") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RVB_SHADOW_LANES="4"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("CRC")][-1]
    assert "shadow_kernel" in line and "shadow_pair_kernel" not in line
    assert int(line.split()[1]) == mine
    # ... and the one-lane-per-record form of round 4 (RVB_SHADOW_LANES=1: measured, not the default)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RVB_SHADOW_LANES="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("CRC")][-1]
    assert "shadow_lane_kernel" in line and int(line.split()[1]) == mine


def test_own_radix_sort_gives_the_same_bytes(ctx, oracle):
    """RVB_SORT=own (csrc/radix_sort.hip instead of rocPRIM's radix sort, read once per process) in a child process: the grouped
    trace and the exact-mode histogram of a seeded scene are the bytes this process gets."""
    import os
    import subprocess
    import sys
    import zlib
    code = ("import sys, zlib, numpy as np; sys.path.insert(0, %r); import rvb_import; rvb_import.load();"
            "from parallel_reverb_raytracer_amd import capi, scenes;"
            "from parallel_reverb_raytracer_amd.dtypes import AIR_COEFFICIENTS;"
            "scene, info = scenes.cathedral(3000); c = capi.Context(0); c.set_scene(scene);"
            "c.raytrace(info['mic'], info['source'], scenes.sphere_directions(5000, seed=29), 30, AIR_COEFFICIENTS);"
            "c.ir_configure_speakers(info['mic'], [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], capi.IR_ALL, c.get_raw_images(False));"
            "hist = c.ir_download(True, 44100.0, capi.IR_EXACT);"
            "print('CRC', zlib.crc32(c.get_raw_diffuse().tobytes()), zlib.crc32(np.ascontiguousarray(hist).tobytes()), hist.shape)"
            ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    crcs = []
    for sort in ("own", "rocprim"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RVB_SORT=sort), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        crcs.append([l for l in out.stdout.splitlines() if l.startswith("CRC")][-1])
    assert crcs[0] == crcs[1]


def grazing_directions(n, seed, max_slope):
    """Unit vectors within `max_slope` of the horizontal plane, random azimuth: rays that skim the floor and ceiling of a room
    and meet their triangles nearly edge-on, where |det| sits just above the reference's 1e-4 rejection threshold and the float
    Möller–Trumbore result is least accurate (kernel.cpp:62-88)."""
    rng = np.random.default_rng(seed)
    az = rng.uniform(-np.pi, np.pi, n)
    slope = rng.uniform(-max_slope, max_slope, n)
    d = np.stack([np.cos(az), slope, np.sin(az)], -1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    from parallel_reverb_raytracer_amd.dtypes import float3_array
    return float3_array(d.astype(np.float32))


@pytest.mark.parametrize("height,slope", [(0.002, 0.003), (0.02, 0.0005), (0.2, 0.02), (2.0, 0.003)])
def test_grazing_rays_match_brute_force(ctx, oracle, height, slope):
    """The BVH may only prune what the brute-force scan would reject: stress the padded boxes and the cull slack with
    rays launched millimetres above a finely tessellated floor at grazing angles (bit-exact against the oracle)."""
    scene = scenes.rotated_square_room(n=16)                       # 3072 triangles, 38 m x 27 m
    mic, src = (-3.0, 4.0, 2.0), (0.5, height, 0.25)
    dirs = grazing_directions(4096, seed=int(height * 1000) + 1, max_slope=slope)
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, 16, AIR_COEFFICIENTS)
    want, image, index = oracle.raytrace(scene, mic, src, dirs, 16, AIR_COEFFICIENTS)
    assert_impulses_equal(ctx.get_raw_diffuse(), want, "grazing h=%g slope=%g" % (height, slope))
    assert_impulses_equal(ctx.get_raw_images(False), oracle.collect_images(image, index, False), "grazing images")


def test_many_surfaces_take_the_global_memory_path(ctx, oracle):
    """More surfaces than the quad kernels stage in LDS (rvb_lds_surfaces budget): the path / shadow instantiations that read
    the coefficient rows from HBM must give the same bytes."""
    from parallel_reverb_raytracer_amd.dtypes import SURFACE, aligned_zeros
    tris, verts, _ = scenes.rotated_square_room(n=6)                 # 432 triangles
    rng = np.random.default_rng(21)
    surfaces = aligned_zeros(300, SURFACE)                           # 300 x 64 B = 19 KB > the 5 KB per-workgroup budget
    surfaces["specular"] = rng.uniform(0.5, 0.99, (300, 8)).astype(np.float32)
    surfaces["diffuse"] = rng.uniform(0.3, 0.95, (300, 8)).astype(np.float32)
    tris = tris.copy()
    tris["surface"] = rng.integers(0, 300, tris.shape[0])
    scene = (tris, verts, surfaces)
    mic, src = (0.5, 2.0, 0.25), (-3.0, 4.0, 2.0)
    dirs = scenes.sphere_directions(777, seed=23)                    # ragged: not a multiple of the 16 rays per wave
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, 19, AIR_COEFFICIENTS)
    want, image, index = oracle.raytrace(scene, mic, src, dirs, 19, AIR_COEFFICIENTS)
    assert_impulses_equal(ctx.get_raw_diffuse(), want, "300 surfaces")
    assert_impulses_equal(ctx.get_raw_images(False), oracle.collect_images(image, index, False), "300 surfaces images")


def triangle_soup(seed):
    """An open scene made to stress the acceleration structure's bookkeeping: random triangles of every size, exact duplicates
    (equal distances: the lowest index must win, kernel.cpp:180-188), coplanar overlapping pairs, zero-area and millimetre
    triangles (never hittable: |det| < 1e-4, quirk Q7), and vertices that are NaN or infinite."""
    from parallel_reverb_raytracer_amd.dtypes import SURFACE, TRIANGLE, aligned_zeros, float3_array
    rng = np.random.default_rng(seed)
    verts, tris = [], []

    def add(p0, p1, p2):
        base = len(verts)
        verts.extend([p0, p1, p2])
        tris.append((int(rng.integers(0, 3)), base, base + 1, base + 2))

    for _ in range(500):                                              # ordinary triangles, 5 cm .. 4 m
        c = rng.uniform(-6, 6, 3)
        size = 10.0 ** rng.uniform(-1.3, 0.6)
        add(c + rng.normal(size=3) * size, c + rng.normal(size=3) * size, c + rng.normal(size=3) * size)
    for k in rng.integers(0, 500, 40):                                # exact duplicates of earlier triangles (new vertex copies)
        _, a, b, c3 = tris[k]
        add(np.array(verts[a]), np.array(verts[b]), np.array(verts[c3]))
    for _ in range(30):                                               # coplanar overlapping pairs
        c = rng.uniform(-5, 5, 3)
        u, v = rng.normal(size=3), rng.normal(size=3)
        add(c, c + u, c + v)
        add(c + 0.25 * u, c + 1.25 * u, c + 0.25 * u + v)
    for _ in range(20):                                               # zero area and millimetre size
        c = rng.uniform(-5, 5, 3)
        add(c, c, c + rng.normal(size=3))
        add(c, c + rng.normal(size=3) * 1e-3, c + rng.normal(size=3) * 1e-3)
    for bad in (np.nan, np.inf, -np.inf):                             # not finite
        c = rng.uniform(-5, 5, 3)
        add(c, c + 1.0, np.array([bad, 0.0, 1.0]))
    t = aligned_zeros(len(tris), TRIANGLE)
    arr = np.asarray(tris, dtype=np.uint64)
    t["surface"], t["v0"], t["v1"], t["v2"] = arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3]
    surfaces = aligned_zeros(3, SURFACE)
    surfaces["specular"] = rng.uniform(0.6, 0.99, (3, 8)).astype(np.float32)
    surfaces["diffuse"] = rng.uniform(0.4, 0.9, (3, 8)).astype(np.float32)
    return t, float3_array(np.asarray(verts, dtype=np.float64)), surfaces


@pytest.mark.parametrize("nrefl", [7, 64, 96])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_triangle_soup_matches_brute_force(ctx, oracle, seed, nrefl):
    """Random soups are OPEN scenes: rays escape after any number of bounces.  64 and 96 bounces are whole 32-bounce runs of grouping
    keys (the path kernels then write the keys of a ray in 64-byte runs through LDS, and an escaped ray has to finish its run and fill
    the runs it never reached); 7 takes the one-store-per-record form."""
    scene = triangle_soup(seed)
    mic, src = (0.3, -0.2, 0.1), (-0.5, 0.4, 0.2)
    dirs = scenes.sphere_directions(3000 if nrefl == 7 else 1500, seed=40 + seed)
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    want, image, index = oracle.raytrace(scene, mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    assert_impulses_equal(ctx.get_raw_diffuse(), want, "soup %d" % seed)
    assert_impulses_equal(ctx.get_raw_images(False), oracle.collect_images(image, index, False), "soup %d images" % seed)
    assert np.count_nonzero(want["time"]) > 100                       # the case is not vacuous: rays do hit and see the microphone
    escaped = want["position"].reshape(dirs.shape[0], nrefl, -1)[:, -1, :3]
    assert (~escaped.any(axis=1)).sum() > 100                         # ... and many rays leave before the last bounce
    # the grouped shadow pass must have seen every record exactly once: a second trace gives the same bytes
    first = ctx.get_raw_diffuse().tobytes()
    ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    assert ctx.get_raw_diffuse().tobytes() == first


@pytest.mark.parametrize("nrays,nrefl", [(1, 1), (1, 700), (17, 333), (32, 1000)])
def test_long_chains_and_tiny_launches(ctx, oracle, nrays, nrefl):
    """One ray, a partial quad group, and chains far longer than the bench's 128 bounces (the job loop keeps a ray's state for
    the whole chain; volumes shrink towards denormals, which are kept)."""
    scene = scenes.rotated_square_room(n=3)
    mic, src = (1.0, 3.0, -2.0), (-2.0, 5.0, 1.5)
    dirs = scenes.sphere_directions(nrays, seed=nrays + nrefl)
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    want, image, index = oracle.raytrace(scene, mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    assert_impulses_equal(ctx.get_raw_diffuse(), want, "%d rays x %d" % (nrays, nrefl))
    assert_impulses_equal(ctx.get_raw_images(True), oracle.collect_images(image, index, True), "images")
    # (a ray may leak through an edge of the closed room exactly as it does in the brute-force scan: the slots after that are empty)
    executed = int(np.count_nonzero(want["position"][:, :3].any(axis=1) | (want["time"] != 0) | want["volume"].any(axis=1)))
    assert executed <= ctx.executed_bounces() <= nrays * nrefl


def test_edge_cases_empty_and_ragged(ctx, oracle):
    scene = scenes.rotated_square_room(n=1)
    ctx.set_scene(scene)
    # zero rays
    ctx.raytrace((0, 2, 0), (0, 2, 2), np.zeros((0, 4), np.float32), 8, AIR_COEFFICIENTS)
    assert ctx.get_raw_diffuse().shape[0] == 0 and ctx.get_image_candidates().shape[0] == 0
    # a ragged count (not a multiple of the wave size) and a single reflection
    dirs = scenes.sphere_directions(67, seed=3)
    ctx.raytrace((0, 2, 0), (0, 2, 2), dirs, 1, AIR_COEFFICIENTS)
    want, _, _ = oracle.raytrace(scene, (0, 2, 0), (0, 2, 2), dirs, 1, AIR_COEFFICIENTS)
    assert_impulses_equal(ctx.get_raw_diffuse(), want)
    # source outside the model: every ray escapes or hits from outside, must still agree
    ctx.raytrace((0, 2, 0), (0, 200, 0), dirs, 4, AIR_COEFFICIENTS)
    want, image, index = oracle.raytrace(scene, (0, 2, 0), (0, 200, 0), dirs, 4, AIR_COEFFICIENTS)
    assert_impulses_equal(ctx.get_raw_diffuse(), want)
    assert_impulses_equal(ctx.get_raw_images(False), oracle.collect_images(image, index, False))


def test_directions_are_unit_vectors_by_contract(ctx, oracle):
    """rvb_set_directions refuses what is not (nearly) a unit vector; lengths inside the accepted [0.5, 2] still trace to the
    brute-force result (the own-plane skip stands down for them, the pruning margins are derived for that range)."""
    from parallel_reverb_raytracer_amd import capi
    scene, info = scenes.cathedral(3000)
    ctx.set_scene(scene)
    dirs = scenes.sphere_directions(640, seed=17).copy()
    for bad in (np.float32(0.0), np.float32(3.0), np.float32(np.nan)):
        broken = dirs.copy()
        broken[5, :3] *= bad
        with pytest.raises(capi.RvbError) as e:
            ctx.set_directions(broken)
        assert e.value.code == 1
    scaled = dirs.copy()
    scaled[::3, :3] *= np.float32(1.75)
    scaled[1::3, :3] *= np.float32(0.6)
    ctx.raytrace(info["mic"], info["source"], scaled, 24, AIR_COEFFICIENTS)
    want, image, index = oracle.raytrace(scene, info["mic"], info["source"], scaled, 24, AIR_COEFFICIENTS)
    assert_impulses_equal(ctx.get_raw_diffuse(), want)
    assert_impulses_equal(ctx.get_raw_images(False), oracle.collect_images(image, index, False))


def test_scene_validation_rejects_bad_indices(ctx):
    from parallel_reverb_raytracer_amd import capi
    tri, vert, surf = scenes.rotated_square_room(n=1)
    bad = tri.copy()
    bad["v2"][3] = vert.shape[0]
    with pytest.raises(capi.RvbError):
        ctx.set_scene((bad, vert, surf))
    ctx.set_scene((tri, vert, surf))


def test_attenuate_speaker_matches_golden(ctx):
    g = load_golden("attenuate_speaker")
    for case in ("axis", "random"):
        imp = golden_impulses(g, case)
        for si in range(g["speaker_coefficient"].shape[0]):
            out = ctx.attenuate_speaker(g[case + "_mic"], imp, g["speaker_direction"][si], float(g["speaker_coefficient"][si]))
            assert np.array_equal(out["volume"], g["%s_s%d_volume" % (case, si)]), (case, si)
            assert np.array_equal(out["time"], g["%s_s%d_time" % (case, si)]), (case, si)
            assert not out["pad"].any()


def test_attenuate_speaker_device_entry_equals_host_entry(ctx, oracle):
    """rvb_attenuate_speaker_device (HBM in, HBM out) runs the same kernel as rvb_attenuate_speaker: same bytes, ragged n."""
    import torch
    from parallel_reverb_raytracer_amd import dtypes
    rng = np.random.default_rng(11)
    n = 100003                                                       # not a multiple of anything the kernel tiles by
    imp = np.zeros(n, dtype=dtypes.IMPULSE)
    imp["volume"] = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
    imp["volume"][::7] = 0                                           # zero-volume impulses -> {0, 0} (quirk Q2)
    imp["position"][:, :3] = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    imp["position"][5, :3] = (1.0, 2.0, 3.0)                         # = mic: normalize(0) stays 0 (quirk Q6)
    imp["time"] = rng.uniform(0.01, 3, n).astype(np.float32)
    mic, direction, coeff = (1.0, 2.0, 3.0), (0.3, -0.2, 0.9), 0.5
    want = ctx.attenuate_speaker(mic, imp, direction, coeff)
    d_in = torch.from_numpy(imp.view(np.uint8).reshape(-1)).cuda()
    d_out = torch.full((n * 64,), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.attenuate_speaker_device(mic, d_in.data_ptr(), n, direction, coeff, d_out.data_ptr())
    ctx.synchronize()
    got = d_out.cpu().numpy().view(dtypes.ATTENUATED)
    assert np.array_equal(got["volume"], want["volume"]) and np.array_equal(got["time"], want["time"])
    ref = oracle.attenuate_speaker(mic, imp, direction, coeff)
    nz = imp["volume"].any(axis=1)
    assert np.array_equal(got["volume"][nz], ref["volume"][nz]) and np.array_equal(got["time"][nz], ref["time"][nz])
    assert not got["volume"][~nz].any() and not got["time"][~nz].any()
    # the HRTF kernel through its device entry, both ears
    table = scenes.hrtf_synthetic_table()
    facing, up = (0.6, 0.0, 0.8), (0.0, 1.0, 0.0)
    for ear in (0, 1):
        want_h = ctx.attenuate_hrtf(mic, imp, table[ear], facing, up, ear)
        d_out.fill_(0xCD)
        torch.cuda.synchronize()
        ctx.attenuate_hrtf_device(mic, d_in.data_ptr(), n, table[ear], facing, up, ear, d_out.data_ptr())
        ctx.synchronize()
        got_h = d_out.cpu().numpy().view(dtypes.ATTENUATED)
        assert np.array_equal(got_h["volume"], want_h["volume"]) and np.array_equal(got_h["time"], want_h["time"])
        ref_h = oracle.attenuate_hrtf(mic, imp, table[ear], facing, up, ear)
        assert np.array_equal(got_h["volume"][nz], ref_h["volume"][nz]) and np.array_equal(got_h["time"][nz], ref_h["time"][nz])


def test_attenuate_hrtf_matches_golden(ctx):
    g = load_golden("attenuate_hrtf")
    tables = {"test": scenes.hrtf_test_table(), "smooth": scenes.hrtf_synthetic_table()}
    for case in ("axis", "random"):
        imp = golden_impulses(g, case)
        for ci in range(g["facing"].shape[0]):
            for ch in (0, 1):
                for tname, tab in tables.items():
                    out = ctx.attenuate_hrtf(g[case + "_mic"], imp, tab[ch], g["facing"][ci], g["up"][ci], ch)
                    key = "%s_c%d_ch%d_%s" % (case, ci, ch, tname)
                    assert np.array_equal(out["volume"], g[key + "_volume"]), key
                    assert np.array_equal(out["time"], g[key + "_time"]), key


def test_attenuate_ragged_sizes(ctx, oracle):
    rng = np.random.default_rng(4)
    for n in (0, 1, 3, 15, 16, 17, 255, 1025):
        imp = dtypes.aligned_zeros(n, IMPULSE)
        imp["volume"] = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        imp["position"][:, :3] = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
        imp["time"] = rng.uniform(0.01, 2, n).astype(np.float32)
        imp["volume"][::5] = 0
        got = ctx.attenuate_speaker((1, 2, 3), imp, (0.3, -1, 0.2), 0.7)
        want = oracle.attenuate_speaker((1, 2, 3), imp, (0.3, -1, 0.2), 0.7)
        assert np.array_equal(got["volume"], want["volume"]) and np.array_equal(got["time"], want["time"]), n


def test_flatten_is_bit_exact_with_serial_order(ctx, oracle):
    """flattenImpulses (reference rayverb.cpp:48-77): float sums in impulse order."""
    rng = np.random.default_rng(8)
    for n in (0, 1, 1000, 40000):
        att = dtypes.aligned_zeros(n, dtypes.ATTENUATED)
        att["volume"] = (rng.uniform(-1, 1, (n, 8)) * rng.choice([1e-3, 1.0, 30.0], (n, 1))).astype(np.float32)
        att["time"] = rng.uniform(0, 0.05, n).astype(np.float32)       # ~2200 bins: heavy collisions
        got = ctx.flatten(att, 44100.0)
        want = oracle.flatten(att, 44100.0)
        assert got.shape == want.shape and np.array_equal(got, want), n


def test_flatten_fill_after_other_calls_on_the_context_is_not_served_from_overwritten_buffers(ctx, oracle):
    """rvb_flatten's size query leaves the uploaded array and its keys on the device for the fill that follows.  Anything
    else the context does in between — a materialised attenuate (its own staging buffers), an exact-mode IR (the sort
    buffers), another flatten — must not turn that fill into a histogram of overwritten data."""
    import ctypes
    from parallel_reverb_raytracer_amd import capi
    rng = np.random.default_rng(81)
    n = 30000
    att = dtypes.aligned_zeros(n, dtypes.ATTENUATED)
    att["volume"] = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
    att["time"] = rng.uniform(0, 0.05, n).astype(np.float32)
    want = oracle.flatten(att, 44100.0)
    lib, h = ctx.lib, ctx.handle

    def query():
        nb = ctypes.c_uint64(0)
        ctx._check(lib.rvb_flatten(h, att.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n), ctypes.c_float(44100.0), None, ctypes.c_uint64(0), ctypes.byref(nb)))
        return nb.value

    def fill(nb):
        out = np.zeros((8, nb), np.float32)
        got = ctypes.c_uint64(0)
        ctx._check(lib.rvb_flatten(h, att.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n), ctypes.c_float(44100.0),
                                   out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(nb), ctypes.byref(got)))
        return out

    imp = dtypes.aligned_zeros(n, dtypes.IMPULSE)
    imp["volume"] = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
    imp["position"][:, :3] = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    imp["time"] = rng.uniform(0, 1, n).astype(np.float32)
    # 1. query -> materialised attenuate on the same context -> fill
    nb = query()
    ctx.attenuate_speaker((0, 1, 0), imp, (0.3, -1, 0.2), 0.7)
    assert np.array_equal(fill(nb), want)
    # 2. query -> exact-mode IR on the same context (rewrites the sort buffers) -> fill
    scene, info = scenes.cathedral(3000)
    ctx.set_scene(scene)
    ctx.raytrace(info["mic"], info["source"], scenes.sphere_directions(2048, seed=3), 32, AIR_COEFFICIENTS)
    nb = query()
    ctx.ir_configure_speakers(info["mic"], [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], capi.IR_ALL, ctx.get_raw_images(False))
    ctx.ir_download(True, 44100.0, capi.IR_EXACT)
    assert np.array_equal(fill(nb), want)
    # 3. query -> flatten of another array -> fill
    other = att.copy()
    other["time"] *= np.float32(0.5)
    nb = query()
    assert np.array_equal(ctx.flatten(other, 44100.0), oracle.flatten(other, 44100.0))
    assert np.array_equal(fill(nb), want)


def _oracle_ir(oracle, mic, impulses, speakers, trim, sr, hrtf=None):
    if hrtf is None:
        chans = [oracle.attenuate_speaker(mic, impulses, d, c) for d, c in speakers]
    else:
        table, facing, up = hrtf
        chans = [oracle.attenuate_hrtf(mic, impulses, table[ch], facing, up, ch) for ch in (0, 1)]
    if trim:
        pd = oracle.find_predelay(chans)
        for c in chans:
            oracle.fix_predelay(c, pd)
    flat = [oracle.flatten(c, sr) for c in chans]
    nb = max(f.shape[1] for f in flat)
    return flat, nb, chans


@pytest.mark.parametrize("trim", [False, True])
@pytest.mark.parametrize("model", ["speakers", "hrtf"])
def test_fused_ir_exact_and_fast_modes(ctx, oracle, model, trim):
    from parallel_reverb_raytracer_amd import capi
    scene, info = scenes.cathedral(3000)
    mic, src = info["mic"], info["source"]
    dirs = scenes.sphere_directions(512, seed=23)
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, 24, AIR_COEFFICIENTS)
    images = ctx.get_raw_images(False)
    all_raw = np.concatenate([ctx.get_raw_diffuse(), images])
    speakers = [((-1, 0, -1), 0.5), ((1, 0, -1), 0.5)]
    hrtf = (scenes.hrtf_synthetic_table(), (1.0, 0.0, 0.2), (0.0, 1.0, 0.0)) if model == "hrtf" else None
    flat, nb, chans = _oracle_ir(oracle, mic, all_raw, speakers, trim, 44100.0, hrtf)
    if hrtf is None:
        ctx.ir_configure_speakers(mic, [s[0] for s in speakers], [s[1] for s in speakers], capi.IR_ALL, images)
    else:
        ctx.ir_configure_hrtf(mic, hrtf[0], hrtf[1], hrtf[2], capi.IR_ALL, images)
    exact = ctx.ir_download(trim, 44100.0, capi.IR_EXACT)
    # the reference bins every channel on its own maxtime; channels are compared over their own length
    assert exact.shape[2] == nb
    for ch in range(2):
        n = flat[ch].shape[1]
        assert np.array_equal(exact[ch][:, :n], flat[ch]), (model, trim, ch)
        assert not exact[ch][:, n:].any()
    fast = ctx.ir_download(trim, 44100.0, capi.IR_FAST)
    assert fast.shape == exact.shape
    # rounding bound for a re-ordered float sum: n_bin * eps * sum|terms| (+ the 1e-5 relative bar)
    for ch in range(2):
        bins = np.round(chans[ch]["time"] * np.float32(44100.0)).astype(np.int64)
        absum = np.zeros((8, exact.shape[2]), np.float64)
        count = np.zeros(exact.shape[2], np.float64)
        np.add.at(count, bins, 1.0)
        for b in range(8):
            np.add.at(absum[b], bins, np.abs(chans[ch]["volume"][:, b].astype(np.float64)))
        bound = 1e-5 * np.abs(exact[ch]) + count[None, :] * 2.0 ** -23 * absum + 1e-30
        assert (np.abs(fast[ch].astype(np.float64) - exact[ch]) <= bound).all(), (model, trim, ch)


def test_histogram_leaves_for_pinned_host_memory_on_the_export_stream(ctx):
    """rvb_host_alloc + rvb_copy_to_pinned_host_async + rvb_synchronize_exports: the copy starts behind the binning that fills the
    device histogram, a second trace enqueued right behind it does not disturb it, and what lands is what rvb_ir_download returns."""
    import ctypes
    import torch
    from parallel_reverb_raytracer_amd import capi
    scene, info = scenes.cathedral(3000)
    mic, src = info["mic"], info["source"]
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, scenes.sphere_directions(4096, seed=31), 32, AIR_COEFFICIENTS)
    images = ctx.get_raw_images(False)
    ctx.ir_configure_speakers(mic, [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], capi.IR_ALL, images)
    want = ctx.ir_download(True, 44100.0, capi.IR_EXACT)
    lo, hi = ctx.ir_time_range()
    nbins = ctx.ir_bins(hi, lo, 44100.0)
    assert want.shape == (2, 8, nbins)
    nbytes = want.nbytes
    host = ctypes.c_void_p()
    ctx._check(ctx.lib.rvb_host_alloc(ctx.handle, ctypes.c_uint64(nbytes), ctypes.byref(host)))
    try:
        landed = np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_float)), shape=(2, 8, nbins))
        landed[...] = np.nan
        hist = torch.zeros((2, 8, nbins), device="cuda", dtype=torch.float32)
        ctx.ir_accumulate_tensor(lo, 44100.0, nbins, capi.IR_EXACT, hist)
        ctx._check(ctx.lib.rvb_copy_to_pinned_host_async(ctx.handle, host, ctypes.c_void_p(hist.data_ptr()), ctypes.c_uint64(nbytes)))
        ctx.trace(mic, src, 32, AIR_COEFFICIENTS)                   # the next trace does not wait for the link, nor the copy for the trace
        ctx.synchronize_exports()
        assert np.array_equal(landed, want)
        ctx.synchronize()
        # an odd byte count and a pageable destination take the runtime's plain copy: same bytes
        plain = np.full(nbins * 16 - 3, np.nan, dtype=np.float32)
        ctx._check(ctx.lib.rvb_copy_to_pinned_host_async(ctx.handle, plain.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(hist.data_ptr()),
                                                         ctypes.c_uint64(plain.nbytes)))
        ctx.synchronize_exports()
        assert np.array_equal(plain, want.reshape(-1)[:plain.shape[0]])
    finally:
        ctx._check(ctx.lib.rvb_host_free(ctx.handle, host))
