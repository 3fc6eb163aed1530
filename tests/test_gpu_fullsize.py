"""Parity at BASELINE.json's full sizes.  The brute-force oracle cannot run 100k x 128 on 75k
triangles (~15 000 core-seconds), so the full-size runs are checked through
  * a seeded SAMPLE of their rays re-traced by the oracle on the same full scene (bit-exact: a ray's
    impulses do not depend on the other rays),
  * shard invariance: tracing the ray set in two halves gives the same bytes as tracing it at once
    (the multi-GPU decomposition is exact per impulse) and the same de-duplicated image sources,
  * run-to-run determinism of everything but the float-atomic histogram,
  * exact-mode histogram == oracle flattenImpulses on a sample small enough for the CPU.
Stand-in scenes (Sibenik / Sponza / a concert hall are not available offline)."""
import zlib

import numpy as np
import pytest

from parallel_reverb_raytracer_amd import scenes
from parallel_reverb_raytracer_amd.dtypes import AIR_COEFFICIENTS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from parallel_reverb_raytracer_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8))


def _same(a, b):
    return all(np.array_equal(a[f], b[f]) for f in ("volume", "time")) and np.array_equal(a["position"][:, :3], b["position"][:, :3])


def test_c2_cathedral_100k_x_128_sampled_against_oracle(ctx, oracle):
    scene, info = scenes.cathedral(75000)
    mic, src = info["mic"], info["source"]
    nrays, nrefl = 100000, 128
    dirs = scenes.sphere_directions(nrays, seed=1)
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    full = ctx.get_raw_diffuse().reshape(nrays, nrefl)
    assert ctx.executed_bounces() > 0.95 * nrays * nrefl            # closed scene: rays keep bouncing
    images_full = ctx.get_raw_images(False)
    candidates_full = ctx.get_image_candidates()

    # (1) seeded sample of rays, brute force over all 75k triangles
    sample = np.sort(np.random.default_rng(5).choice(nrays, 192, replace=False))     # 24.6 k bounces x 75 k triangles
    want, image, index = oracle.raytrace(scene, mic, src, dirs[sample], nrefl, AIR_COEFFICIENTS)
    assert _same(full[sample].reshape(-1), want)
    # image-source slots of the sampled rays
    idx = index.reshape(len(sample), 10)
    for k, ray in enumerate(sample):
        mine = candidates_full[candidates_full["ray"] == ray]
        slots = np.nonzero(idx[k, 1:])[0] + 1
        assert np.array_equal(mine["slot"], slots) and np.array_equal(mine["index"], idx[k, slots])
        assert _same(mine["impulse"], image.reshape(len(sample), 10)[k, slots])

    # (2) shard invariance: two halves == the whole, byte for byte (checksum of checksums)
    crc_full = [_crc(full[:nrays // 2]), _crc(full[nrays // 2:])]
    from parallel_reverb_raytracer_amd import capi
    shards, cands = [], []
    for first in (0, nrays // 2):
        ctx.set_directions(dirs[first:first + nrays // 2])
        ctx.trace(mic, src, nrefl, AIR_COEFFICIENTS, ray_offset=first)
        shards.append(_crc(ctx.get_raw_diffuse()))
        cands.append(ctx.get_image_candidates())
    assert shards == crc_full
    merged = capi.merge_images(np.concatenate(cands[::-1]), ctx.get_direct(), False)      # shard order must not matter
    assert _same(merged, images_full)

    # (3) determinism of the trace itself
    ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    assert _crc(ctx.get_raw_diffuse()) == _crc(full)


def fast_vs_exact_report(fast, exact):
    """How far the float-atomic histogram is from the serial-order one: the claimed tolerance is 1e-5 x the band's largest
    |value| (sign-alternating volumes cancel inside a bin, so a purely relative bar per band-bin is not
    meaningful where the sum is ~0); also reported: the fraction of band-bins outside a pure 1e-5 RELATIVE error."""
    fast64, exact64 = fast.astype(np.float64), exact.astype(np.float64)
    err = np.abs(fast64 - exact64)
    band_max = np.abs(exact64).max(axis=2, keepdims=True)                  # [channels][8][1]
    nonzero = exact64 != 0
    rel = np.zeros_like(err)
    rel[nonzero] = err[nonzero] / np.abs(exact64[nonzero])
    outside = (rel > 1e-5) | (~nonzero & (err > 0))
    return {"max_abs_err_over_band_max": float((err / np.maximum(band_max, 1e-300)).max()),
            "fraction_band_bins_outside_1e-5_relative": float(outside.mean()),
            "max_relative_err_where_exact_nonzero": float(rel.max()),
            "band_bins": int(err.size), "band_bins_differing": int((err > 0).sum())}


def test_c2_final_ir_exact_mode_equals_oracle_chain_and_fast_mode_is_within_tolerance(ctx, oracle):
    """BASELINE config C2 at full size, the FINAL impulse response: the traced impulses are downloaded and run through the
    oracle's chain attenuate -> findPredelay / fixPredelay -> flattenImpulses on the CPU (reference cmd/main.cpp:280-298,
    rayverb.h:49-97, rayverb.cpp:48-77); the exact-mode [2][8][nbins] histogram must be bit-equal to it, and the
    float-atomic (fast) histogram within 1e-5 of each band's maximum."""
    import json
    import os
    from parallel_reverb_raytracer_amd import capi
    scene, info = scenes.cathedral(75000)
    mic, src = info["mic"], info["source"]
    nrays, nrefl, sr = 100000, 128, 44100.0
    speakers = [((-1, 0, -1), 0.5), ((1, 0, -1), 0.5)]                   # bench.py's two cardioids
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, scenes.sphere_directions(nrays, seed=1), nrefl, AIR_COEFFICIENTS)
    images = ctx.get_raw_images(False)
    all_raw = np.concatenate([ctx.get_raw_diffuse(), images])
    chans = [oracle.attenuate_speaker(mic, all_raw, d, c) for d, c in speakers]
    del all_raw
    pd = oracle.find_predelay(chans)
    flat = []
    for c in chans:
        oracle.fix_predelay(c, pd)
        flat.append(oracle.flatten(c, sr))
    del chans
    ctx.ir_configure_speakers(mic, [s[0] for s in speakers], [s[1] for s in speakers], capi.IR_ALL, images)
    exact = ctx.ir_download(True, sr, capi.IR_EXACT)
    assert exact.shape == (2, 8, max(f.shape[1] for f in flat)) and exact.shape[2] > 500000
    for ch in range(2):
        n = flat[ch].shape[1]
        assert np.array_equal(exact[ch][:, :n], flat[ch]), ch
        assert not exact[ch][:, n:].any()
    fast = ctx.ir_download(True, sr, capi.IR_FAST)
    assert fast.shape == exact.shape
    report = fast_vs_exact_report(fast, exact)
    print("fast_vs_exact at C2:", json.dumps(report))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(report, open(os.path.join(out, "fast_vs_exact_c2.json"), "w"))
    assert report["max_abs_err_over_band_max"] <= 1e-5


def test_c4_atrium_262k_triangles_256_bounces_small_ray_set(ctx, oracle):
    scene, info = scenes.atrium(262000)
    assert scene[0].shape[0] > 250000
    nrays, nrefl = 4096, 256
    dirs = scenes.sphere_directions(nrays, seed=4)
    ctx.set_scene(scene)
    ctx.raytrace(info["mic"], info["source"], dirs, nrefl, AIR_COEFFICIENTS)
    got = ctx.get_raw_diffuse().reshape(nrays, nrefl)
    sample = np.array([0, 17, 1023, 4095])
    want, _, _ = oracle.raytrace(scene, info["mic"], info["source"], dirs[sample], nrefl, AIR_COEFFICIENTS)
    assert _same(got[sample].reshape(-1), want)


def _ulp_distance(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2147483648) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2147483648) - ib, ib)
    return np.abs(ia - ib)


def check_every_ray_against_gpu_brute_force(ctx, oracle, gpu_oracle, scene, mic, src, dirs, nrefl, cross_check_rays=48):
    """The product's trace of ALL rays against the oracle source run brute force on the GPU (one thread per ray, every triangle
    tested for every query).  The GPU build of the oracle is first held against its CPU build on a few rays; then every impulse of
    every ray is compared: position and time bit for bit, volumes within one ulp (the device's binary64 pow and glibc's may round
    a handful of values differently), image-source slots exactly.  Returns a small report."""
    nrays = dirs.shape[0]
    sample = np.sort(np.random.default_rng(3).choice(nrays, cross_check_rays, replace=False))
    cpu, cpu_image, cpu_index = oracle.raytrace(scene, mic, src, dirs[sample], nrefl, AIR_COEFFICIENTS)
    gpu, gpu_image, gpu_index = gpu_oracle.raytrace(scene, mic, src, dirs[sample], nrefl, AIR_COEFFICIENTS)
    assert np.array_equal(cpu["position"], gpu["position"]) and np.array_equal(cpu["time"], gpu["time"])
    assert np.array_equal(cpu_index, gpu_index) and np.array_equal(cpu_image["position"], gpu_image["position"])
    assert _ulp_distance(cpu["volume"], gpu["volume"]).max() <= 1

    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    got = ctx.get_raw_diffuse()
    cands = ctx.get_image_candidates()
    want, image, index = gpu_oracle.raytrace(scene, mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    assert np.array_equal(got["position"][:, :3], want["position"][:, :3]), "a hit point differs from brute force"
    assert np.array_equal(got["time"], want["time"]), "an arrival time differs from brute force"
    ulps = _ulp_distance(got["volume"], want["volume"])
    assert ulps.max() <= 1
    # image sources: every valid slot of every ray
    idx = index.reshape(nrays, 10)
    rays, slots = np.nonzero(idx[:, 1:])
    order = np.lexsort((slots, rays))
    assert cands.shape[0] == rays.shape[0]
    assert np.array_equal(cands["ray"], rays[order].astype(np.uint64)) and np.array_equal(cands["slot"], (slots[order] + 1).astype(np.uint32))
    assert np.array_equal(cands["index"], idx[rays[order], slots[order] + 1].astype(np.uint32))
    img = image.reshape(nrays, 10)[rays[order], slots[order] + 1]
    assert np.array_equal(cands["impulse"]["position"][:, :3], img["position"][:, :3]) and np.array_equal(cands["impulse"]["time"], img["time"])
    # the other path kernel (two lanes per ray instead of four, or the reverse): the same bytes
    first_lanes = 2 if nrays >= 196608 else 4
    ctx.set_concurrent_traces(1 << 20 if first_lanes == 4 else 1)
    try:
        ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
        again = ctx.get_raw_diffuse()
        assert again.tobytes() == got.tobytes(), "the two-lane and four-lane path kernels disagree"
        assert ctx.get_image_candidates().tobytes() == cands.tobytes()
    finally:
        ctx.set_concurrent_traces(1)
    # ... the one-lane-per-ray kernel of round 4 (measured, not chosen by any launch): the same bytes again
    ctx.set_path_lanes(1)
    try:
        ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
        assert "path_lane_kernel" in dict(ctx.last_timings())
        assert _crc(ctx.get_raw_diffuse()) == _crc(got), "the one-lane path kernel disagrees"
        assert ctx.get_image_candidates().tobytes() == cands.tobytes()
    finally:
        ctx.set_path_lanes(0)
    # ... and the kernel the bench pipeline times: the traces of TWO contexts in ONE launch (rvb_trace_group ->
    # path_pair_group_kernel), same scene, rays, source and microphone on both, as distributed.IrPipeline issues them
    group_checked = False
    if 2 * nrays >= 196608:
        from parallel_reverb_raytracer_amd import capi
        other = capi.Context(0)
        try:
            other.set_scene(scene)
            for c in (ctx, other):
                c.set_directions(dirs)
            capi.Context.trace_group([ctx, other], [mic, mic], [src, src], nrefl, AIR_COEFFICIENTS, [0, 0])
            want_crc, want_cands = _crc(got), cands.tobytes()
            for c in (ctx, other):
                assert "path_pair_kernel" in dict(c.last_timings()), "the group launch did not use the two-lane group kernel"
                assert _crc(c.get_raw_diffuse()) == want_crc, "a trace of the two-trace group launch differs from the same trace alone"
                assert c.get_image_candidates().tobytes() == want_cands
            group_checked = True
        finally:
            other.close()
    return {"rays": int(nrays), "impulses": int(got.shape[0]), "volume_values_off_by_one_ulp": int((ulps == 1).sum()),
            "image_source_slots": int(rays.shape[0]), "both_path_kernels": True, "two_trace_group_launch": group_checked}


def test_c2_every_ray_of_the_full_run_against_brute_force_on_the_gpu(ctx, oracle, gpu_oracle):
    """BASELINE config C2, ALL 100 000 rays x 128 bounces: 12.8 M closest-hit queries, 12.8 M shadow rays and every image-source
    validation against a brute-force scan of all 75 252 triangles (about 2 x 10^12 triangle tests, on the GPU)."""
    scene, info = scenes.cathedral(75000)
    report = check_every_ray_against_gpu_brute_force(ctx, oracle, gpu_oracle, scene, info["mic"], info["source"],
                                                     scenes.sphere_directions(100000, seed=1), 128)
    print("C2 exhaustive:", report)


def test_c4_sixteen_thousand_rays_x_256_bounces_every_ray_against_brute_force_on_the_gpu(ctx, oracle, gpu_oracle):
    """BASELINE config C4's scene and depth (263 k triangles, 256 bounces): every impulse of 16 384 rays against brute force on
    the GPU (2 x 10^12 triangle tests; one thread per ray, so the run time is that of ONE ray whatever the count) — the divergence-stress scene, whose long chains amplify any wrong decision."""
    scene, info = scenes.atrium(262000)
    report = check_every_ray_against_gpu_brute_force(ctx, oracle, gpu_oracle, scene, info["mic"], info["source"],
                                                     scenes.sphere_directions(16384, seed=4), 256, cross_check_rays=12)
    print("C4 exhaustive:", report)


def test_c3_per_gpu_share_125k_rays_at_a_ray_offset(ctx, oracle):
    """BASELINE config C3 = 1M rays x 128 over 8 GPUs: what ONE of the eight ranks runs — its contiguous 125 000-ray shard of
    the global seeded set (here rank 5's: ray_offset 625 000), traced as one resident round of waves (125 000 rays = 7 813
    workgroups of 16 on 8 192 wave slots... one round at 8 waves per SIMD).  Sampled rays against brute force, image-source
    candidates carrying GLOBAL ray numbers, run-to-run determinism, and the exact-mode histogram of the shard against the
    oracle chain on a prefix small enough for the CPU."""
    from parallel_reverb_raytracer_amd import capi, distributed
    scene, info = scenes.cathedral(75000)
    mic, src = info["mic"], info["source"]
    world, rank, nrefl = 8, 5, 128
    first, nrays = distributed.shard_range(1000000, rank, world)
    assert (first, nrays) == (625000, 125000)
    dirs = scenes.sphere_directions(nrays, seed=1, first=first)
    assert np.array_equal(dirs[:7], scenes.sphere_directions(1000000, seed=1)[first:first + 7])      # the shard IS a slice of the global set
    ctx.set_scene(scene)
    ctx.set_directions(dirs)
    ctx.trace(mic, src, nrefl, AIR_COEFFICIENTS, ray_offset=first)
    full = ctx.get_raw_diffuse().reshape(nrays, nrefl)
    cands = ctx.get_image_candidates()
    assert cands.shape[0] == 0 or (cands["ray"].min() >= first and cands["ray"].max() < first + nrays)
    sample = np.sort(np.random.default_rng(9).choice(nrays, 96, replace=False))
    want, image, index = oracle.raytrace(scene, mic, src, dirs[sample], nrefl, AIR_COEFFICIENTS)
    assert _same(full[sample].reshape(-1), want)
    idx = index.reshape(len(sample), 10)
    for k, ray in enumerate(sample):
        mine = cands[cands["ray"] == first + ray]
        slots = np.nonzero(idx[k, 1:])[0] + 1
        assert np.array_equal(mine["slot"], slots) and np.array_equal(mine["index"], idx[k, slots])
    crc = _crc(full)
    # the shard's histogram as rank 5 forms it (diffuse impulses only; rank 0 adds the merged images): exact mode == serial sum
    ctx.ir_configure_speakers(mic, [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], capi.IR_DIFFUSE, None)
    lo, hi = ctx.ir_time_range()
    exact = ctx.ir_download(True, 44100.0, capi.IR_EXACT)
    chans = [oracle.attenuate_speaker(mic, full.reshape(-1), d, 0.5) for d in ((-1, 0, -1), (1, 0, -1))]
    pd = oracle.find_predelay(chans)
    assert pd == lo
    for ch in range(2):
        oracle.fix_predelay(chans[ch], pd)
        flat = oracle.flatten(chans[ch], 44100.0)
        assert np.array_equal(exact[ch][:, :flat.shape[1]], flat) and not exact[ch][:, flat.shape[1]:].any()
    ctx.trace(mic, src, nrefl, AIR_COEFFICIENTS, ray_offset=first)
    assert _crc(ctx.get_raw_diffuse()) == crc


def test_c4_atrium_100k_rays_x_256_bounces(ctx, oracle):
    """BASELINE config C4 at its full size: 100 000 rays x 256 bounces on the 263k-triangle atrium stand-in (1.6 GB of impulses):
    sampled rays against brute force over all triangles, shard invariance, determinism."""
    scene, info = scenes.atrium(262000)
    mic, src = info["mic"], info["source"]
    nrays, nrefl = 100000, 256
    dirs = scenes.sphere_directions(nrays, seed=4)
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    full = ctx.get_raw_diffuse().reshape(nrays, nrefl)
    assert ctx.executed_bounces() > 0.9 * nrays * nrefl
    sample = np.sort(np.random.default_rng(6).choice(nrays, 40, replace=False))       # 10 k bounces x 263 k triangles
    want, _, _ = oracle.raytrace(scene, mic, src, dirs[sample], nrefl, AIR_COEFFICIENTS)
    assert _same(full[sample].reshape(-1), want)
    crc_halves = [_crc(full[:nrays // 2]), _crc(full[nrays // 2:])]
    images_full = ctx.get_raw_images(False)
    del full
    from parallel_reverb_raytracer_amd import capi
    got, cands = [], []
    for first in (0, nrays // 2):
        ctx.set_directions(dirs[first:first + nrays // 2])
        ctx.trace(mic, src, nrefl, AIR_COEFFICIENTS, ray_offset=first)
        got.append(_crc(ctx.get_raw_diffuse()))
        cands.append(ctx.get_image_candidates())
    assert got == crc_halves
    assert _same(capi.merge_images(np.concatenate(cands[::-1]), ctx.get_direct(), False), images_full)


def test_c5_per_gpu_share_8_pairs_x_100k_rays_hrtf_four_pairs_per_launch(ctx, oracle):
    """BASELINE config C5 = 64 (source, listener) pairs over 8 GPUs, HRTF: ONE rank's share — 8 pairs x 100 000 rays x 128 —
    through generate_pair_irs with four pairs per launch (400 000 rays per launch, launches alternating between two contexts),
    in the benchmarked exact mode: every pair's [2][8][nbins] histogram must be bit-identical to that pair traced and binned
    alone; sampled rays of two pairs against brute force; the float-atomic mode within the stated tolerance of the exact one."""
    import torch
    from parallel_reverb_raytracer_amd import capi, distributed
    scene, _ = scenes.concert_hall(30000)
    src, mic = scenes.source_mic_pairs(64, seed=0)
    table = scenes.hrtf_synthetic_table()
    nrays, nrefl = 100000, 128
    first, count = distributed.shard_range(64, 3, 8)                    # rank 3 of 8: pairs 24 .. 31
    assert count == 8
    pairs = list(range(first, first + count))
    dirs = scenes.sphere_directions(nrays, seed=1)
    other = capi.Context(0)
    try:
        for c in (ctx, other):
            c.set_scene(scene)
            c.set_directions(dirs)
        device = torch.device("cuda", 0)

        def model_for(i):
            p = pairs[i]
            facing = src[p] - mic[p]
            return distributed.HrtfModel(table, facing / np.linalg.norm(facing), (0, 1, 0))

        batched = distributed.generate_pair_irs([ctx, other], [(mic[p], src[p]) for p in pairs], nrefl, AIR_COEFFICIENTS, model_for,
                                                44100.0, device=device, mode=capi.IR_EXACT, pairs_per_launch=4)
        assert sorted(batched) == list(range(count))
        rng = np.random.default_rng(12)
        for k, p in enumerate(pairs):
            hist, info = distributed.generate_ir(ctx, mic[p], src[p], nrefl, AIR_COEFFICIENTS, model=model_for(k), sample_rate=44100.0,
                                                 trim_predelay=True, mode=capi.IR_EXACT, device=device)
            got, got_info = batched[k]
            assert got_info["nbins"] == info["nbins"] and got_info["images"] == info["images"]
            assert torch.equal(got, hist) and bool(hist.any())
            if k in (0, 5):
                sample = np.sort(rng.choice(nrays, 48, replace=False))
                want, _, _ = oracle.raytrace(scene, mic[p], src[p], dirs[sample], nrefl, AIR_COEFFICIENTS)
                assert _same(ctx.get_raw_diffuse().reshape(nrays, nrefl)[sample].reshape(-1), want)
            if k == 7:          # pair 31: ten million audible impulses (pair 24's microphone sits in a corner most shadow rays cannot reach)
                fast, _ = distributed.generate_ir(ctx, mic[p], src[p], nrefl, AIR_COEFFICIENTS, model=model_for(k), sample_rate=44100.0,
                                                  trim_predelay=True, mode=capi.IR_FAST, device=device)
                report = fast_vs_exact_report(fast.cpu().numpy(), hist.cpu().numpy())
                print("fast_vs_exact at C5 (HRTF):", report)
                assert report["max_abs_err_over_band_max"] <= 1e-5 and report["band_bins_differing"] > 1000
    finally:
        other.close()


def test_c5_hall_source_listener_pairs_hrtf(ctx, oracle):
    """A few of the 64 (source, listener) pairs: trace + HRTF-attenuated, predelay-trimmed IR in exact mode
    against the oracle chain attenuate -> fixPredelay -> flattenImpulses."""
    from parallel_reverb_raytracer_amd import capi
    scene, _ = scenes.concert_hall(30000)
    src, mic = scenes.source_mic_pairs(64, seed=0)
    table = scenes.hrtf_synthetic_table()
    ctx.set_scene(scene)
    for pair in (0, 31, 63):
        dirs = scenes.sphere_directions(192, seed=pair + 1)
        facing = src[pair] - mic[pair]
        facing = facing / np.linalg.norm(facing)
        ctx.raytrace(mic[pair], src[pair], dirs, 24, AIR_COEFFICIENTS)
        want, image, index = oracle.raytrace(scene, mic[pair], src[pair], dirs, 24, AIR_COEFFICIENTS)
        assert _same(ctx.get_raw_diffuse(), want)
        images = ctx.get_raw_images(False)
        assert _same(images, oracle.collect_images(image, index, False))
        ctx.ir_configure_hrtf(mic[pair], table, facing, (0, 1, 0), capi.IR_ALL, images)
        ir = ctx.ir_download(True, 44100.0, capi.IR_EXACT)
        all_raw = np.concatenate([want, images])
        chans = [oracle.attenuate_hrtf(mic[pair], all_raw, table[ch], facing, (0, 1, 0), ch) for ch in (0, 1)]
        pd = oracle.find_predelay(chans)
        for ch in (0, 1):
            oracle.fix_predelay(chans[ch], pd)
            flat = oracle.flatten(chans[ch], 44100.0)
            assert np.array_equal(ir[ch][:, :flat.shape[1]], flat) and not ir[ch][:, flat.shape[1]:].any()


def test_c5_pairs_through_the_two_context_pipeline_equal_one_at_a_time(ctx):
    """BASELINE config C5's shape — source / listener pairs of one hall, HRTF — run as jobs of IrPipeline (the next pair's trace
    enqueued on the second context before the current pair is finished): every pair's exact-mode IR must be bit-identical to
    the IR of that pair computed alone."""
    import torch
    from parallel_reverb_raytracer_amd import capi, distributed
    scene, _ = scenes.concert_hall(30000)
    src, mic = scenes.source_mic_pairs(64, seed=0)
    table = scenes.hrtf_synthetic_table()
    nrays, nrefl, pairs = 4096, 32, [0, 9, 17, 40, 63]
    dirs = scenes.sphere_directions(nrays, seed=3)
    other = capi.Context(0)
    try:
        for c in (ctx, other):
            c.set_scene(scene)
            c.set_directions(dirs)
        device = torch.device("cuda", 0)

        def job(p):
            facing = src[p] - mic[p]
            facing = facing / np.linalg.norm(facing)
            return ((mic[p], src[p], nrefl, AIR_COEFFICIENTS),
                    dict(model=distributed.HrtfModel(table, facing, (0, 1, 0)), sample_rate=44100.0, trim_predelay=True,
                         mode=capi.IR_EXACT, device=device))

        alone = []
        for p in pairs:
            args, kwargs = job(p)
            hist, info = distributed.generate_ir(ctx, *args, **kwargs)
            alone.append((hist.cpu().numpy(), info["nbins"], info["images"]))
        got = []
        distributed.IrPipeline([ctx, other]).run_jobs([job(p) for p in pairs],
                                                      lambda hist, info, tracer: got.append((hist.cpu().numpy(), info["nbins"], info["images"])))
        assert len(got) == len(pairs)
        for (h0, n0, i0), (h1, n1, i1) in zip(alone, got):
            assert n0 == n1 and i0 == i1 and np.array_equal(h0, h1)
        assert any(h.any() for h, _, _ in got)

        # the same pairs through generate_pair_irs, three pairs per launch (rvb_trace_pairs), launches alternating between contexts
        def model_for(i):
            return job(pairs[i])[1]["model"]
        batched = distributed.generate_pair_irs([ctx, other], [(mic[p], src[p]) for p in pairs], nrefl, AIR_COEFFICIENTS, model_for,
                                                44100.0, device=device, mode=capi.IR_EXACT, pairs_per_launch=3)
        assert sorted(batched) == list(range(len(pairs)))
        for k in range(len(pairs)):
            hist, info = batched[k]
            assert info["nbins"] == alone[k][1] and info["images"] == alone[k][2] and np.array_equal(hist.cpu().numpy(), alone[k][0])
    finally:
        other.close()


def test_pairs_in_one_launch_equal_pairs_one_by_one(ctx):
    """rvb_trace_pairs: several (source, microphone) pairs of one hall traced in ONE launch give, pair by pair, the bytes that
    tracing each pair alone gives — raw impulses, image-source candidates, direct path — and the same exact-mode IR, for the
    speaker model and for HRTF."""
    from parallel_reverb_raytracer_amd import capi
    scene, _ = scenes.concert_hall(30000)
    src, mic = scenes.source_mic_pairs(64, seed=0)
    table = scenes.hrtf_synthetic_table()
    nrays, nrefl, pairs = 5000, 24, [3, 11, 40, 41, 63]           # 5000: not a multiple of the 16 rays per wave
    dirs = scenes.sphere_directions(nrays, seed=7)
    ctx.set_scene(scene)
    ctx.set_directions(dirs)

    def irs(p):
        images = capi.merge_images(cands, ctx.get_direct(), False)
        out = []
        ctx.ir_configure_speakers(mic[p], [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], capi.IR_ALL, images)
        out.append(ctx.ir_download(True, 44100.0, capi.IR_EXACT))
        facing = src[p] - mic[p]
        ctx.ir_configure_hrtf(mic[p], table, facing / np.linalg.norm(facing), (0, 1, 0), capi.IR_ALL, images)
        out.append(ctx.ir_download(True, 44100.0, capi.IR_EXACT))
        return images, out

    alone = []
    for p in pairs:
        ctx.trace(mic[p], src[p], nrefl, AIR_COEFFICIENTS)
        cands = ctx.get_image_candidates()
        diffuse = ctx.get_raw_diffuse()
        images, ir = irs(p)
        alone.append((diffuse, cands, ctx.get_direct(), images, ir))

    ctx.trace_pairs(mic[pairs], src[pairs], nrefl, AIR_COEFFICIENTS)
    everything = ctx.get_raw_diffuse().reshape(len(pairs), nrays * nrefl)
    all_cands = ctx.get_image_candidates()
    assert all_cands.shape[0] == sum(a[1].shape[0] for a in alone)
    for k, p in enumerate(pairs):
        diffuse, cands_alone, direct, images_alone, ir_alone = alone[k]
        assert _same(everything[k], diffuse)
        ctx.select_pair(k)
        cands = ctx.get_pair_candidates(k, all_cands)
        assert np.array_equal(cands["ray"], cands_alone["ray"]) and np.array_equal(cands["slot"], cands_alone["slot"])
        assert np.array_equal(cands["index"], cands_alone["index"]) and _same(cands["impulse"], cands_alone["impulse"])
        assert _same(ctx.get_direct(), direct)
        images, ir = irs(p)
        assert _same(images, images_alone)
        for a, b in zip(ir, ir_alone):
            assert a.shape == b.shape and np.array_equal(a, b)
    assert any(a[4][0].any() for a in alone)


def test_trace_group_one_launch_for_three_contexts_equals_three_traces(ctx):
    """rvb_trace_group: the path kernels of several contexts in ONE launch (what distributed.IrPipeline does with the traces of a
    group).  Three contexts with their own ray sets (one of them ragged: not a multiple of the 32 rays per wave), sources,
    microphones and ray offsets; the raw diffuse impulses, the image-source candidates, the direct path and an exact-mode IR of each
    must be the bytes the same context produces with rvb_trace."""
    from parallel_reverb_raytracer_amd import capi
    scene, info = scenes.cathedral(20000)
    counts, nrefl = [90000, 70001, 60000], 6                       # 220 001 rays in all: enough for the two-lane kernel
    mics = [info["mic"], (10.0, 2.0, 1.0), (-5.0, 1.5, -2.0)]
    sources = [info["source"], (-12.0, 1.7, 0.5), (6.0, 2.5, 3.0)]
    offsets = [0, 90000, 1 << 33]
    others = [capi.Context(0), capi.Context(0)]
    contexts = [ctx] + others
    try:
        want = []
        for c, n, m, s, off in zip(contexts, counts, mics, sources, offsets):
            c.set_scene(scene)
            c.set_directions(scenes.sphere_directions(n, seed=n))
            c.trace(m, s, nrefl, AIR_COEFFICIENTS, ray_offset=off)
            images = capi.merge_images(c.get_image_candidates(), c.get_direct(), False)
            c.ir_configure_speakers(m, [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], capi.IR_ALL, images)
            want.append((c.get_raw_diffuse().tobytes(), c.get_image_candidates().tobytes(), c.get_direct().tobytes(),
                         c.ir_download(True, 44100.0, capi.IR_EXACT)))
        capi.Context.trace_group(contexts, mics, sources, nrefl, AIR_COEFFICIENTS, offsets)
        for c, m, (diffuse, cands, direct, ir) in zip(contexts, mics, want):
            assert "path_pair_kernel" in dict(c.last_timings())
            assert c.get_raw_diffuse().tobytes() == diffuse
            assert c.get_image_candidates().tobytes() == cands and c.get_direct().tobytes() == direct
            images = capi.merge_images(c.get_image_candidates(), c.get_direct(), False)
            c.ir_configure_speakers(m, [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], capi.IR_ALL, images)
            again = c.ir_download(True, 44100.0, capi.IR_EXACT)
            assert again.shape == ir.shape and np.array_equal(again, ir)
        # a small group falls back to one launch per context (four lanes per ray): same results again
        for c in contexts:
            c.set_directions(scenes.sphere_directions(777, seed=5))
        small = []
        for c, m, s in zip(contexts, mics, sources):
            c.trace(m, s, 12, AIR_COEFFICIENTS)
            small.append(c.get_raw_diffuse().tobytes())
        capi.Context.trace_group(contexts, mics, sources, 12, AIR_COEFFICIENTS)
        for c, b in zip(contexts, small):
            assert c.get_raw_diffuse().tobytes() == b
    finally:
        for c in others:
            c.close()


def test_c5_hrtf_rows_by_binary32_atan2_and_the_two_ear_list_equal_the_always_exact_evaluation(ctx):
    """The HRTF table row of an impulse needs only the INTEGER parts of two angles in degrees; the kernels take them from the binary32
    atan2f and fall back to the correctly rounded (binary64) evaluation when an angle lies within 2e-3 degrees of an integer
    (stream_kernels.hip, angle_deg).  At BASELINE config C5's full size — every impulse of 100 000 rays x 128 bounces, two pairs —
    the materialised `hrtf` kernel of both ears and the exact-mode [2][8][nbins] histogram (one combined two-ear list, one sort, one
    fold) must give the bytes of the always-exact evaluation with one sorted list per ear (RVB_HRTF_EXACT_ROWS=1,
    RVB_HRTF_SPLIT_EARS=1: the round-2 path, which the oracle-chain tests above pin)."""
    import os
    import torch
    from parallel_reverb_raytracer_amd import capi
    scene, _ = scenes.concert_hall(30000)
    src, mic = scenes.source_mic_pairs(64, seed=0)
    table = scenes.hrtf_synthetic_table()
    nrays, nrefl = 100000, 128
    ctx.set_scene(scene)
    ctx.set_directions(scenes.sphere_directions(nrays, seed=1))
    saved = {k: os.environ.get(k) for k in ("RVB_HRTF_EXACT_ROWS", "RVB_HRTF_SPLIT_EARS")}

    def switches(on):
        for k in saved:
            if on:
                os.environ[k] = "1"
            else:
                os.environ.pop(k, None)

    try:
        for pair in (31, 5):
            facing = src[pair] - mic[pair]
            facing = facing / np.linalg.norm(facing)
            ctx.trace(mic[pair], src[pair], nrefl, AIR_COEFFICIENTS)
            d_in, n = ctx.diffuse_device()
            images = capi.merge_images(ctx.get_image_candidates(), ctx.get_direct(), False)
            out = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
            got = {}
            for exact in (False, True):
                switches(exact)
                crcs = []
                for ch in (0, 1):
                    ctx.attenuate_hrtf_device(mic[pair], d_in, n, table[ch], facing, (0, 1, 0), ch, out.data_ptr())
                    ctx.synchronize()
                    crcs.append(_crc(out.cpu().numpy()))
                ctx.ir_configure_hrtf(mic[pair], table, facing, (0, 1, 0), capi.IR_ALL, images)
                got[exact] = (crcs, ctx.ir_download(True, 44100.0, capi.IR_EXACT), ctx.ir_download(True, 44100.0, capi.IR_FAST))
            assert got[False][0] == got[True][0], "materialised hrtf kernel: a table row differs from the always-exact evaluation"
            assert got[False][1].shape == got[True][1].shape and np.array_equal(got[False][1], got[True][1]) and got[True][1].any()
            band_max = np.abs(got[True][1]).max(axis=2, keepdims=True)
            assert (np.abs(got[False][2].astype(np.float64) - got[True][1]) <= 1e-5 * band_max).all()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
