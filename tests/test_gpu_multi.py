"""Several "devices" behind the C-ABI (rvb_multi_*, csrc/multi.hip) on the one GPU of the test box: the same GPU listed twice or
three times gives real ray shards, real merges and the exact-mode chain; RCCL itself needs distinct devices, so its binding is
exercised with a one-device communicator (RVB_MULTI_REHEARSE_RCCL).  More than one physical GPU is the driver's scaling run."""
import numpy as np
import pytest

from parallel_reverb_raytracer_amd import scenes
from parallel_reverb_raytracer_amd.dtypes import AIR_COEFFICIENTS

pytestmark = pytest.mark.gpu

SPEAKERS = ([(-1, 0, -1), (1, 0, -1)], [0.5, 0.5])


def _same(a, b):
    return all(np.array_equal(a[f], b[f]) for f in ("volume", "time")) and np.array_equal(a["position"][:, :3], b["position"][:, :3])


@pytest.fixture(scope="module")
def case():
    scene, info = scenes.cathedral(12000)
    return scene, info["mic"], info["source"], scenes.sphere_directions(6001, seed=11), 48     # 6001: shards of unequal size


@pytest.fixture(scope="module")
def single(case):
    from parallel_reverb_raytracer_amd import capi
    scene, mic, src, dirs, nrefl = case
    ctx = capi.Context(0)
    ctx.set_scene(scene)
    ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
    images = ctx.get_raw_images(False)
    out = {"diffuse": ctx.get_raw_diffuse(), "images": images}
    ctx.ir_configure_speakers(mic, SPEAKERS[0], SPEAKERS[1], capi.IR_ALL, images)
    out["speakers"] = ctx.ir_download(True, 44100.0, capi.IR_EXACT)
    table = scenes.hrtf_synthetic_table()
    ctx.ir_configure_hrtf(mic, table, (1.0, 0.0, 0.2), (0.0, 1.0, 0.0), capi.IR_ALL, images)
    out["hrtf"] = ctx.ir_download(True, 44100.0, capi.IR_EXACT)
    ctx.close()
    return out


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_shards_on_one_gpu_reproduce_a_single_context_bit_for_bit(case, single, devices):
    from parallel_reverb_raytracer_amd import capi
    scene, mic, src, dirs, nrefl = case
    m = capi.MultiContext(devices)
    try:
        m.set_scene(scene)
        m.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
        ranges = [m.shard(i) for i in range(len(devices))]
        assert ranges[0][0] == 0 and sum(c for _, c in ranges) == dirs.shape[0] and all(ranges[i + 1][0] == ranges[i][0] + ranges[i][1] for i in range(len(devices) - 1))
        assert _same(m.get_raw_diffuse(), single["diffuse"])
        assert _same(m.get_raw_images(False), single["images"])
        # exact mode: the devices continue ONE serial sum in ray order
        exact = m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], True, 44100.0, capi.IR_EXACT)
        assert exact.shape == single["speakers"].shape and np.array_equal(exact, single["speakers"])
        table = scenes.hrtf_synthetic_table()
        hrtf = m.ir_hrtf(mic, table, (1.0, 0.0, 0.2), (0.0, 1.0, 0.0), True, 44100.0, capi.IR_EXACT)
        assert hrtf.shape == single["hrtf"].shape and np.array_equal(hrtf, single["hrtf"])
        # fast mode: float atomics per shard, then the sum over the shards (peer copies here: RCCL refuses one GPU twice)
        fast = m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], True, 44100.0, capi.IR_FAST)
        assert not m.used_rccl()
        band_max = np.abs(exact).max(axis=2, keepdims=True)
        assert (np.abs(fast.astype(np.float64) - exact) <= 1e-5 * band_max).all() and fast.any()
        # diffuse only / images only add up to the same bins
        only_images = m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], False, 44100.0, capi.IR_EXACT, which=capi.IR_IMAGES)
        assert only_images.any() and only_images.shape[2] <= exact.shape[2]
    finally:
        m.close()


@pytest.mark.parametrize("blocks", [1, 3, 8])
def test_systolic_exact_chain_gives_the_same_bytes_for_any_block_count(case, single, blocks):
    """The exact mode's histogram travels from device to device in bin-range blocks (rvb_multi_set_chain_blocks): device g folds
    block k while device g + 1 folds block k - 1.  One, three and eight blocks, four "devices" (the test GPU listed four times):
    speaker and HRTF histograms bit-equal to a single context — the reference's serial order, rayverb.cpp:67-74, :708-714."""
    from parallel_reverb_raytracer_amd import capi
    scene, mic, src, dirs, nrefl = case
    m = capi.MultiContext([0, 0, 0, 0])
    try:
        assert m.peer_links() == 0                       # one physical GPU: nothing to enable
        m.set_chain_blocks(blocks)
        m.set_scene(scene)
        m.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
        exact = m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], True, 44100.0, capi.IR_EXACT)
        assert exact.shape == single["speakers"].shape and np.array_equal(exact, single["speakers"])
        hrtf = m.ir_hrtf(mic, scenes.hrtf_synthetic_table(), (1.0, 0.0, 0.2), (0.0, 1.0, 0.0), True, 44100.0, capi.IR_EXACT)
        assert hrtf.shape == single["hrtf"].shape and np.array_equal(hrtf, single["hrtf"])
        diffuse_only = m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], True, 44100.0, capi.IR_EXACT, which=capi.IR_DIFFUSE)
        assert diffuse_only.any()
    finally:
        m.close()


def test_a_device_without_rays_passes_the_blocks_on(case):
    """Three rays on four devices: one shard is empty and only hands the histogram's blocks to the next device."""
    from parallel_reverb_raytracer_amd import capi
    scene, mic, src, dirs, nrefl = case
    ctx = capi.Context(0)
    m = capi.MultiContext([0, 0, 0, 0])
    try:
        ctx.set_scene(scene)
        ctx.raytrace(mic, src, dirs[:3], nrefl, AIR_COEFFICIENTS)
        ctx.ir_configure_speakers(mic, SPEAKERS[0], SPEAKERS[1], capi.IR_ALL, ctx.get_raw_images(False))
        want = ctx.ir_download(True, 44100.0, capi.IR_EXACT)
        m.set_chain_blocks(3)
        m.set_scene(scene)
        m.raytrace(mic, src, dirs[:3], nrefl, AIR_COEFFICIENTS)
        assert sorted(m.shard(i)[1] for i in range(4)) == [0, 1, 1, 1]
        got = m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], True, 44100.0, capi.IR_EXACT)
        assert got.shape == want.shape and np.array_equal(got, want)
    finally:
        m.close()
        ctx.close()


def test_exact_mode_in_two_steps_equals_one_call(case, single):
    """rvb_ir_exact_prepare + rvb_ir_exact_fold over all bins in five uneven ranges = rvb_ir_accumulate(RVB_IR_EXACT); a call that reuses
    the sort buffers in between voids the prepared list and the fold refuses."""
    import torch
    from parallel_reverb_raytracer_amd import capi
    scene, mic, src, dirs, nrefl = case
    ctx = capi.Context(0)
    try:
        ctx.set_scene(scene)
        ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
        for name, configure in (("speakers", lambda: ctx.ir_configure_speakers(mic, SPEAKERS[0], SPEAKERS[1], capi.IR_ALL, single["images"])),
                                ("hrtf", lambda: ctx.ir_configure_hrtf(mic, scenes.hrtf_synthetic_table(), (1.0, 0.0, 0.2), (0.0, 1.0, 0.0), capi.IR_ALL, single["images"]))):
            configure()
            lo, hi = ctx.ir_time_range()
            nbins = ctx.ir_bins(hi, lo, 44100.0)
            assert nbins == single[name].shape[2]
            hist = torch.zeros((2, 8, nbins), device="cuda", dtype=torch.float32)
            ctx.ir_exact_prepare(lo, 44100.0, nbins)
            cuts = [0, 17, nbins // 3, nbins // 3 + 1, nbins - 5, nbins]
            for b0, b1 in zip(cuts[:-1], cuts[1:]):
                ctx.ir_exact_fold_tensor(nbins, b0, b1, hist)
            ctx.synchronize()
            assert np.array_equal(hist.cpu().numpy(), single[name]), name
            configure()                                   # voids the prepared list
            with pytest.raises(capi.RvbError):
                ctx.ir_exact_fold_tensor(nbins, 0, nbins, hist)
    finally:
        ctx.close()


def test_histogram_leaves_for_the_host_in_bin_ranges(case, single):
    """rvb_ir_accumulate_export: the exact-mode histogram is copied to pinned host memory bin range by bin range behind the folds
    (1, 3, 8 ranges and the fast mode's single copy): same bytes as the device histogram, and as ir_download."""
    import torch
    from parallel_reverb_raytracer_amd import capi
    scene, mic, src, dirs, nrefl = case
    ctx = capi.Context(0)
    try:
        ctx.set_scene(scene)
        ctx.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
        ctx.ir_configure_speakers(mic, SPEAKERS[0], SPEAKERS[1], capi.IR_ALL, single["images"])
        lo, hi = ctx.ir_time_range()
        nbins = ctx.ir_bins(hi, lo, 44100.0)
        for slices in (1, 3, 8):
            hist = torch.zeros((2, 8, nbins), device="cuda", dtype=torch.float32)
            host = torch.full((2, 8, nbins), -1.0, dtype=torch.float32).pin_memory()
            ctx.ir_accumulate_export_tensor(lo, 44100.0, nbins, capi.IR_EXACT, hist, host, slices)
            ctx.synchronize()
            ctx.synchronize_exports()
            assert np.array_equal(host.numpy(), single["speakers"]) and np.array_equal(hist.cpu().numpy(), single["speakers"]), slices
        hist = torch.zeros((2, 8, nbins), device="cuda", dtype=torch.float32)
        host = torch.full((2, 8, nbins), -1.0, dtype=torch.float32).pin_memory()
        ctx.ir_accumulate_export_tensor(lo, 44100.0, nbins, capi.IR_FAST, hist, host)
        ctx.synchronize()
        ctx.synchronize_exports()
        assert np.array_equal(host.numpy(), hist.cpu().numpy()) and host.numpy().any()
    finally:
        ctx.close()


def test_rccl_all_reduce_is_bound_and_runs(case, single):
    """librccl.so loaded at run time, one communicator over the device list, ncclAllReduce in place on the histogram: with one
    device the sum is the identity, so the result must equal the plain fast-mode histogram up to its own atomics order."""
    from parallel_reverb_raytracer_amd import capi
    scene, mic, src, dirs, nrefl = case
    m = capi.MultiContext([0], flags=capi.MULTI_REHEARSE_RCCL)
    try:
        m.set_scene(scene)
        m.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
        fast = m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], True, 44100.0, capi.IR_FAST)
        assert m.used_rccl(), "RCCL was not used: librccl.so missing or the communicator could not be created"
        exact = single["speakers"]
        band_max = np.abs(exact).max(axis=2, keepdims=True)
        assert fast.shape == exact.shape and (np.abs(fast.astype(np.float64) - exact) <= 1e-5 * band_max).all()
        assert np.array_equal(m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], True, 44100.0, capi.IR_EXACT), exact)
    finally:
        m.close()


def test_two_processes_chained_exact_mode_equals_one_context(tmp_path):
    """One process per rank (bench.py --gpus 2 starts its own two ranks; gloo, both on the test box's one GPU): with --exact-chain
    the ranks continue ONE serial sum in ray order, so the [2][8][nbins] histogram of 2 x 3000 rays must have the bytes of ONE
    context tracing the same 6000 rays; with the default all-reduce of per-rank sums it is within the stated tolerance."""
    import os
    import re
    import subprocess
    import sys
    import zlib
    import torch
    from conftest import ROOT
    from parallel_reverb_raytracer_amd import capi, distributed
    args = ["--gpus", "2", "--backend", "gloo", "--share-gpu", "--rays", "3000", "--reflections", "24", "--triangles", "6000",
            "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, RVB_BENCH_CRC="1")
    found = {}
    for label, extra in (("chain", ["--exact-chain"]), ("allreduce", [])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + extra, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        line = __import__("json").loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2
        m = re.search(r"histogram nbins (\d+) crc32 (\d+)", r.stderr)
        assert m, r.stderr[-2000:]
        found[label] = (int(m.group(1)), int(m.group(2)), line["timed_region_check"])
    scene, info = scenes.cathedral(6000)
    ctx = capi.Context(0)
    try:
        ctx.set_scene(scene)
        ctx.set_directions(scenes.sphere_directions(6000, seed=1))
        hist, meta = distributed.generate_ir(ctx, info["mic"], info["source"], 24, AIR_COEFFICIENTS, SPEAKERS[0], SPEAKERS[1], 44100.0,
                                             trim_predelay=True, mode=capi.IR_EXACT, device=torch.device("cuda", 0))
        want = (int(meta["nbins"]), zlib.crc32(hist.cpu().numpy().tobytes()))
    finally:
        ctx.close()
    assert found["chain"][:2] == want and found["chain"][2]["required"] == "bit-equal"
    assert found["allreduce"][0] == want[0] and found["allreduce"][2]["max_abs_err_over_band_max"] <= 1e-5
