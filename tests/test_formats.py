"""On-disk formats around the hot path: the reference's impulse.dump (reference
rayverb/helpers.cpp:19-59) written by the C++ host library and by the Python module, and the RVBHIST1
binary impulse-response dump.  The reference holds no fixture of an impulse.dump; its writer is rapidjson's
(absent here), so the text is checked through its consumer's contract — every line parses as JSON and carries
exactly the float values the reference would print — and digit-for-digit against Python's shortest repr
(parity unpinned for the rare cases where rapidjson's Grisu2 emits one digit more)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "parallel-reverb-raytracer_amd")
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "formats_roundtrip")


@pytest.fixture(scope="module")
def tool():
    subprocess.check_call(["make", "-C", PKG, "-j4"], stdout=subprocess.DEVNULL)
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "include", "shims"), os.path.join(ROOT, "tests", "cpp", "formats_roundtrip.cpp"),
                           "-o", BIN, "-L" + PKG, "-lrayverb", "-lrvb_hip", "-Wl,-rpath," + PKG])
    return BIN


def _impulses(nrays, nrefl, seed=4):
    from parallel_reverb_raytracer_amd import dtypes
    rng = np.random.default_rng(seed)
    imp = np.zeros(nrays * nrefl, dtype=dtypes.IMPULSE)
    imp["volume"] = (rng.uniform(-1, 1, (imp.shape[0], 8)) * 10.0 ** rng.integers(-9, 1, (imp.shape[0], 1))).astype(np.float32)
    imp["position"][:, :3] = rng.uniform(-30, 30, (imp.shape[0], 3)).astype(np.float32)
    imp["time"] = rng.uniform(0, 3, imp.shape[0]).astype(np.float32)
    if imp.shape[0] > 6:
        imp[3] = np.zeros(1, dtype=dtypes.IMPULSE)            # an escaped ray's zero slot
        imp["position"][5, :3] = (1.0, -0.0, 1e22)             # integers, negative zero, exponent form
        imp["volume"][6] = 1e-7
    return imp


def test_impulse_dump_cpp_matches_reference_rule(tool, tmp_path):
    from parallel_reverb_raytracer_amd import formats
    nrays, nrefl = 7, 5
    imp = _impulses(nrays, nrefl)
    raw, out = str(tmp_path / "imp.bin"), str(tmp_path / "impulse.dump")
    imp.tofile(raw)
    subprocess.check_call([tool, "dump", raw, str(nrays), str(nrefl), out])
    lines = open(out).read().splitlines()
    assert len(lines) == nrays                               # one line per ray (helpers.cpp:27-58)
    want_avg = np.zeros(imp.shape[0], dtype=np.float32)
    for k in range(8):                                       # helpers.cpp:46-49: float accumulation, then / 8
        want_avg = (want_avg + imp["volume"][:, k]).astype(np.float32)
    want_avg = (want_avg / np.float32(8)).astype(np.float32)
    for i, line in enumerate(lines):
        ray = json.loads(line)
        assert len(ray) == nrefl
        for j, rec in enumerate(ray):
            assert list(rec.keys()) == ["position", "volume"]
            k = i * nrefl + j
            assert [np.float32(x) for x in rec["position"]] == list(imp["position"][k, :3])
            assert rec["volume"] == float(want_avg[k])
    # digit for digit what the Python writer emits (shortest round-trip digits; rapidjson layout rules)
    py = str(tmp_path / "py.dump")
    formats.write_impulse_dump(py, imp, nrays, nrefl)
    pos_c, vol_c = formats.read_impulse_dump(out)
    pos_p, vol_p = formats.read_impulse_dump(py)
    assert np.array_equal(pos_c, pos_p) and np.array_equal(vol_c, vol_p)
    assert "9.999999778196308e21" in lines[1] and "[1.0,-0.0," in lines[1]      # record 5 = ray 1, reflection 0: float 1e22 as a double


def test_impulse_dump_number_layout(tool, tmp_path):
    """rapidjson Prettify layout: plain decimals for exponents in (-6, 21], d.ddde[-]x otherwise."""
    from parallel_reverb_raytracer_amd import dtypes
    imp = np.zeros(1, dtype=dtypes.IMPULSE)
    cases = [(0.5, "0.5"), (1.0, "1.0"), (123456.0, "123456.0"), (1e21, "1e21"), (1e-6, "0.000001"), (1e-7, "1e-7"),
             (1.5e-9, "1.5e-9"), (-2.25, "-2.25")]
    for value, text in cases:
        imp["position"][0, :3] = (np.float32(value), 0, 0)
        expect = text if float(np.float32(value)) == value else None
        raw, out = str(tmp_path / "one.bin"), str(tmp_path / "one.dump")
        imp.tofile(raw)
        subprocess.check_call([tool, "dump", raw, "1", "1", out])
        line = open(out).read().strip()
        got = line[len('[{"position":['):].split(",")[0]
        assert float(got) == float(np.float32(value))
        if expect is not None:
            assert got == expect, (value, got)


def test_ir_dump_roundtrip_cpp_and_python(tool, tmp_path):
    from parallel_reverb_raytracer_amd import dtypes, formats
    rng = np.random.default_rng(2)
    hist = rng.normal(size=(2, 8, 1234)).astype(np.float32)
    images = _impulses(3, 1)
    a, b = str(tmp_path / "a.rvbh"), str(tmp_path / "b.rvbh")
    formats.write_ir_dump(a, hist, 44100.0, 0.0942, images)
    subprocess.check_call([tool, "ir", a, b])                 # C++ reads and re-writes
    assert open(a, "rb").read() == open(b, "rb").read()
    back = formats.read_ir_dump(b)
    assert np.array_equal(back["histogram"], hist) and back["sample_rate"] == 44100.0
    assert np.float32(back["predelay"]) == np.float32(0.0942)
    assert back["images"].tobytes() == images.tobytes()
    # malformed inputs are refused
    open(str(tmp_path / "bad.rvbh"), "wb").write(b"RVBHIST0" + bytes(40))
    assert subprocess.run([tool, "ir", str(tmp_path / "bad.rvbh"), b], capture_output=True).returncode == 3
    open(str(tmp_path / "short.rvbh"), "wb").write(open(a, "rb").read()[:-10])
    assert subprocess.run([tool, "ir", str(tmp_path / "short.rvbh"), b], capture_output=True).returncode == 3
    with pytest.raises(ValueError):
        formats.read_ir_dump(str(tmp_path / "short.rvbh"))


@pytest.mark.gpu
def test_impulse_dump_of_a_gpu_trace_equals_the_dump_of_the_oracles_impulses(tool, oracle, tmp_path):
    """SURVEY §8(f)-4 on traced data: the impulses of a GPU trace (reference demo model bedroom.obj, the DIAGNOSTIC path of
    cmd/main.cpp:270-278: raytrace -> getRawDiffuse -> print_diagnostic, helpers.cpp:19-59) and the CPU oracle's impulses of the same
    rays give the same impulse.dump, byte for byte; escaped rays' zero slots and the float -> double digits included.  The reference
    holds no impulse.dump fixture, so the pin is the oracle's impulses through the same writer plus the reader's contract."""
    from parallel_reverb_raytracer_amd import capi, formats, scenes
    from parallel_reverb_raytracer_amd.dtypes import AIR_COEFFICIENTS
    assets = os.path.join(ROOT, "tests", "golden", "assets")
    scene = scenes.load_obj(os.path.join(assets, "bedroom.obj"), os.path.join(assets, "mat.json"))
    mic, src, nrays, nrefl = (0.5, 0.2, -0.5), (-0.6, 0.4, 1.0), 96, 24
    dirs = scenes.sphere_directions(nrays, seed=21)
    ctx = capi.Context(0)
    try:
        ctx.set_scene(scene)
        ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
        got = ctx.get_raw_diffuse()
    finally:
        ctx.close()
    want, _, _ = oracle.raytrace(scene, mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    files = {}
    for name, imp in (("gpu", got), ("oracle", want)):
        raw, out = str(tmp_path / (name + ".bin")), str(tmp_path / (name + ".dump"))
        np.ascontiguousarray(imp).tofile(raw)
        subprocess.check_call([tool, "dump", raw, str(nrays), str(nrefl), out])
        files[name] = open(out, "rb").read()
    assert files["gpu"] == files["oracle"] and len(files["gpu"].splitlines()) == nrays
    pos, vol = formats.read_impulse_dump(str(tmp_path / "gpu.dump"))
    assert np.array_equal(pos.reshape(-1, 3), got["position"][:, :3]) and np.abs(vol).max() > 0
