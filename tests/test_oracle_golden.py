"""The CPU oracle against (i) the known answers of the reference's own gtest suites and (ii) the
golden vectors produced by the reference's kernels compiled for the host (tests/golden/*.npz,
generator oracle/make_golden.py).  Bit-exact: the oracle *is* the definition of parity."""
import numpy as np
import pytest

from conftest import golden_impulses, golden_scene, load_golden
from parallel_reverb_raytracer_amd import scenes

TRACE_CASES = ["trace_large_square", "trace_echo_tunnel", "trace_random_pillars", "trace_vault"]


@pytest.mark.parametrize("name", TRACE_CASES)
def test_raytrace_matches_reference_kernel_output(oracle, name):
    g = load_golden(name)
    nrefl = int(g["nreflections"])
    imp, img, idx = oracle.raytrace(golden_scene(g), g["mic"], g["source"], g["directions"], nrefl, g["air"])
    assert np.array_equal(imp["position"][:, :3], g["impulse_position"])
    assert np.array_equal(imp["time"], g["impulse_time"])
    assert np.array_equal(imp["volume"], g["impulse_volume"])
    assert np.array_equal(idx, g["image_index"])
    assert np.array_equal(img["position"][:, :3], g["image_position"])
    assert np.array_equal(img["time"], g["image_time"])
    assert np.array_equal(img["volume"], g["image_volume"])


def test_raytrace_known_answers_of_reference_gtest(oracle):
    """reference tests/raytrace_tests.h:35-47 (ASSERT_FLOAT_EQ = 4 ULP)."""
    g = load_golden("trace_large_square")
    nrefl = int(g["nreflections"])
    imp, _, _ = oracle.raytrace(golden_scene(g), g["mic"], g["source"], g["directions"], nrefl, g["air"])
    pos = imp["position"].reshape(-1, nrefl, 4)[:, :, :3]
    bounce0 = [(0, 2, -27), (0, 2, 27), (0, 0, 2), (0, 27, 2), (-25, 2, 2), (25, 2, 2)]
    bounce1 = [(0, 0, 0), (0, 0, 0), (0, 27, 2), (0, 0, 2), (-25, 2, -2), (25, 2, -2)]
    for r in range(6):
        np.testing.assert_array_almost_equal_nulp(pos[r, 0], np.float32(bounce0[r]), nulp=4)
        np.testing.assert_array_almost_equal_nulp(pos[r, 1] + np.float32(64), np.float32(bounce1[r]) + np.float32(64), nulp=4)
    # the two rays that hit a wall corner escape: every later slot keeps the host zero-fill
    assert not imp.reshape(-1, nrefl)[0:2, 1:]["volume"].any()
    # digits measured with the host-compiled reference kernel
    t = imp["time"].reshape(-1, nrefl)
    v = imp["volume"].reshape(-1, nrefl, 8)
    assert t[0, 0] == np.float32(0.164705887) and v[0, 0, 0] == np.float32(-0.66132009)
    assert t[2, 1] == np.float32(0.159058452) and v[2, 1, 0] == np.float32(0.926073313)


def test_attenuate_speaker_golden_and_gtest_answers(oracle):
    """reference tests/attenuation_tests.h:67-101."""
    g = load_golden("attenuate_speaker")
    for case in ("axis", "random"):
        imp = golden_impulses(g, case)
        for si in range(g["speaker_coefficient"].shape[0]):
            out = oracle.attenuate_speaker(g[case + "_mic"], imp, g["speaker_direction"][si], float(g["speaker_coefficient"][si]))
            assert np.array_equal(out["volume"], g["%s_s%d_volume" % (case, si)])
            assert np.array_equal(out["time"], g["%s_s%d_time" % (case, si)])
    imp = golden_impulses(g, "axis")
    expect = {0: [1, 1, 1, 1, 1, 1], 1: [.5, .5, .5, .5, 0, 1], 2: [0, 0, 0, 0, -1, 1]}
    for si, want in expect.items():
        out = oracle.attenuate_speaker((0, 0, 0), imp, (0, 0, 1), float(g["speaker_coefficient"][si]))
        np.testing.assert_allclose(out["volume"][:6, 0], want, rtol=0, atol=1e-7)
        assert (out["volume"] == out["volume"][:, :1]).all()            # all 8 bands equal
        assert np.array_equal(out["time"], imp["time"])                  # AttenuationTest.Timing
        # impulses *at* the microphone rely on normalize(0) = 0  ->  gain 1 - shape
        np.testing.assert_allclose(out["volume"][6:, 0], 1 - float(g["speaker_coefficient"][si]), atol=1e-7)


def test_attenuate_hrtf_golden_and_gtest_answers(oracle):
    """reference tests/hrtf_tests.cpp:42-85 with the regenerated (azimuth, elevation)-encoding table."""
    g = load_golden("attenuate_hrtf")
    tables = {"test": scenes.hrtf_test_table(), "smooth": scenes.hrtf_synthetic_table()}
    for case in ("axis", "random"):
        imp = golden_impulses(g, case)
        for ci in range(g["facing"].shape[0]):
            for ch in (0, 1):
                for tname, tab in tables.items():
                    out = oracle.attenuate_hrtf(g[case + "_mic"], imp, tab[ch], g["facing"][ci], g["up"][ci], ch)
                    key = "%s_c%d_ch%d_%s" % (case, ci, ch, tname)
                    assert np.array_equal(out["volume"], g[key + "_volume"]), key
                    assert np.array_equal(out["time"], g[key + "_time"]), key
    # which impulse must select front [180][90], back [0][90], [90][90], [270][90] per head orientation
    imp = golden_impulses(g, "axis")
    t = tables["test"]
    rows = {0: {5: 180, 4: 0, 0: 90, 1: 270}, 1: {1: 180, 0: 0, 5: 90, 4: 270},
            2: {4: 180, 5: 0, 1: 90, 0: 270}, 3: {0: 180, 1: 0, 4: 90, 5: 270}}
    for ci, sel in rows.items():
        out = oracle.attenuate_hrtf((0, 0, 0), imp, t[0], g["facing"][ci], g["up"][ci], 0)
        for impulse, az in sel.items():
            assert np.array_equal(out["volume"][impulse], t[0, az, 90]), (ci, impulse)


def test_hrtf_straight_down_selects_next_azimuth_row(oracle):
    """Quirk Q5: e = 90 - (-90) = 180 indexes the next azimuth row."""
    assert oracle.hrtf_index((0, 0, 1), (0, 1, 0), (0, -1, 0)) == 180 * 180 + 180 == 181 * 180 + 0


def test_port_equals_host_compiled_reference_on_fresh_inputs(oracle, reference_oracle):
    """Build-container only: a seeded case that is not among the fixtures."""
    scene = scenes.rotated_square_room(n=3)
    dirs = scenes.sphere_directions(40, seed=99)
    from parallel_reverb_raytracer_amd.dtypes import AIR_COEFFICIENTS
    a = oracle.raytrace(scene, (1, 2, 0.5), (-3, 4, 2), dirs, 40, AIR_COEFFICIENTS)
    b = reference_oracle.raytrace(scene, (1, 2, 0.5), (-3, 4, 2), dirs, 40, AIR_COEFFICIENTS)
    for x, y in zip(a, b):
        if x.dtype.names:
            for f in ("volume", "time"):
                assert np.array_equal(x[f], y[f])
            assert np.array_equal(x["position"][:, :3], y["position"][:, :3])
        else:
            assert np.array_equal(x, y)
