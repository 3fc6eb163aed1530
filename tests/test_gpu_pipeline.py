"""The pipeline behind the C-ABI (rvb_pipeline_*, csrc/pipeline.hip) through its ctypes binding, against distributed.generate_ir on one
context: incomplete groups, more contexts than jobs, image sources only / diffuse only, no predelay trimming, both histogram modes, the
HRTF model with a facing per job, results taken late (the ring of pinned buffers), a pipeline re-configured between batches."""
import numpy as np
import pytest

from parallel_reverb_raytracer_amd import dtypes, scenes

pytestmark = pytest.mark.gpu

SPEAKERS = ([(-1, 0, -1), (1, 0, -1)], [0.5, 0.5])
NREFL = 24


@pytest.fixture(scope="module")
def rig():
    from parallel_reverb_raytracer_amd import capi
    scene, info = scenes.concert_hall(6000)
    dirs = scenes.sphere_directions(9000, seed=31)
    ctxs = [capi.Context(0) for _ in range(5)]
    for k, c in enumerate(ctxs):
        if k in (1, 2, 3):
            c.share_scene(ctxs[0])          # the pipeline's contexts read ONE copy of the scene (rvb_share_scene); the solo context has its own
        else:
            c.set_scene(scene)
        c.set_directions(dirs)
    src, mic = scenes.source_mic_pairs(9, seed=5)
    yield ctxs[:4], ctxs[4], [(tuple(float(x) for x in m), tuple(float(x) for x in s)) for s, m in zip(src, mic)]      # (microphone, source)
    for c in ctxs:
        c.close()


def _solo(ctx, mic, src, mode, which, trim, model=None, remove_direct=False):
    import torch
    from parallel_reverb_raytracer_amd import distributed
    hist, info = distributed.generate_ir(ctx, mic, src, NREFL, dtypes.AIR_COEFFICIENTS, SPEAKERS[0], SPEAKERS[1], 44100.0, trim_predelay=trim,
                                         mode=mode, which=which, remove_direct=remove_direct, device=torch.device("cuda", 0), model=model)
    ctx.synchronize()
    return hist.cpu().numpy(), info


@pytest.mark.parametrize("ncontexts,group,njobs", [(4, 0, 3), (4, 0, 9), (3, 3, 4), (1, 0, 2), (2, 1, 5)])
def test_jobs_through_the_pipeline_equal_one_context_bit_for_bit(rig, ncontexts, group, njobs):
    from parallel_reverb_raytracer_amd import capi
    ctxs, solo, pairs = rig
    pipe = capi.Pipeline(ctxs[:ncontexts], group)
    try:
        pipe.configure_speakers(SPEAKERS[0], SPEAKERS[1], NREFL, dtypes.AIR_COEFFICIENTS, 44100.0, True, capi.IR_EXACT)
        for mic, src in pairs[:njobs]:
            pipe.submit(mic, src)
        assert pipe.pending() == njobs
        for k, (mic, src) in enumerate(pairs[:njobs]):
            got, info = pipe.next()
            want, winfo = _solo(solo, mic, src, capi.IR_EXACT, capi.IR_ALL, True)
            assert info["job"] == k and info["nbins"] == winfo["nbins"] and info["images"] == winfo["images"]
            assert np.float32(info["predelay"]) == np.float32(winfo["predelay"])
            assert got.shape == want.shape and np.array_equal(got, want) and got.any()
        assert pipe.pending() == 0
        with pytest.raises(capi.RvbError):
            pipe.next()
    finally:
        pipe.close()


def test_reconfigured_between_batches_images_only_diffuse_only_fast_mode_and_hrtf(rig):
    from parallel_reverb_raytracer_amd import capi, distributed
    ctxs, solo, pairs = rig
    pipe = capi.Pipeline(ctxs)
    try:
        # image sources only, the direct path removed, no predelay trimming
        pipe.configure_speakers(SPEAKERS[0], SPEAKERS[1], NREFL, dtypes.AIR_COEFFICIENTS, 44100.0, False, capi.IR_EXACT, which=capi.IR_IMAGES, remove_direct=True)
        for mic, src in pairs[:3]:
            pipe.submit(mic, src)
        views = [pipe.next(copy=False) for _ in range(3)]            # taken late: three results stay valid in the ring of 8 buffers
        for (got, info), (mic, src) in zip(views, pairs[:3]):
            want, winfo = _solo(solo, mic, src, capi.IR_EXACT, capi.IR_IMAGES, False, remove_direct=True)
            assert info["predelay"] == 0.0 and info["images"] == winfo["images"]
            assert np.array_equal(np.asarray(got), want)
        # diffuse only, float atomics: within the fast mode's tolerance of the exact histogram
        pipe.configure_speakers(SPEAKERS[0], SPEAKERS[1], NREFL, dtypes.AIR_COEFFICIENTS, 44100.0, True, capi.IR_FAST, which=capi.IR_DIFFUSE)
        mic, src = pairs[4]
        pipe.submit(mic, src)
        got, info = pipe.next()
        want, _ = _solo(solo, mic, src, capi.IR_EXACT, capi.IR_DIFFUSE, True)
        band_max = np.abs(want).max(axis=2, keepdims=True)
        assert info["images"] == 0 and (np.abs(got.astype(np.float64) - want) <= 1e-5 * band_max).all() and got.any()
        # HRTF, every job facing its source
        table = scenes.hrtf_synthetic_table()
        pipe.configure_hrtf(table, (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), NREFL, dtypes.AIR_COEFFICIENTS, 44100.0, True, capi.IR_EXACT)
        facings = []
        for mic, src in pairs[5:9]:
            d = np.array(src, np.float64) - np.array(mic, np.float64)
            d[1] = 0.0
            facing = tuple(float(x) for x in d / np.linalg.norm(d))
            facings.append(facing)
            pipe.submit(mic, src, facing, (0.0, 1.0, 0.0))
        for (mic, src), facing in zip(pairs[5:9], facings):
            got, info = pipe.next()
            want, winfo = _solo(solo, mic, src, capi.IR_EXACT, capi.IR_ALL, True, model=distributed.HrtfModel(table, facing, (0.0, 1.0, 0.0)))
            assert info["nbins"] == winfo["nbins"] and np.array_equal(got, want)
    finally:
        pipe.close()


def test_hrtf_table_stays_on_the_device_for_table_none(rig):
    """rvb_ir_configure_hrtf(table = NULL): the previous call's table stays (what the pipeline does from a context's second job on); a
    context that never had one refuses; a one-ear attenuate call in between voids it."""
    from parallel_reverb_raytracer_amd import capi
    ctxs, solo, pairs = rig
    table = scenes.hrtf_synthetic_table()
    mic, src = pairs[0]
    ctx = capi.Context(0)
    try:
        ctx.set_scene(scenes.concert_hall(6000)[0])
        ctx.raytrace(mic, src, scenes.sphere_directions(2000, seed=3), 12, dtypes.AIR_COEFFICIENTS)
        with pytest.raises(capi.RvbError):
            ctx.ir_configure_hrtf(mic, None, (0.0, 0.0, 1.0), (0.0, 1.0, 0.0))
        ctx.ir_configure_hrtf(mic, table, (0.0, 0.0, 1.0), (0.0, 1.0, 0.0))
        want = ctx.ir_download(True, 44100.0, capi.IR_EXACT)
        ctx.ir_configure_hrtf(mic, None, (0.0, 0.0, 1.0), (0.0, 1.0, 0.0))
        assert np.array_equal(ctx.ir_download(True, 44100.0, capi.IR_EXACT), want) and want.any()
        ctx.ir_configure_hrtf(mic, None, (1.0, 0.0, 0.0), (0.0, 1.0, 0.0))              # another orientation, the same table
        turned = ctx.ir_download(True, 44100.0, capi.IR_EXACT)
        ctx.ir_configure_hrtf(mic, table, (1.0, 0.0, 0.0), (0.0, 1.0, 0.0))
        assert np.array_equal(ctx.ir_download(True, 44100.0, capi.IR_EXACT), turned) and not np.array_equal(turned, want)
        ctx.attenuate_hrtf(mic, ctx.get_raw_diffuse()[:8], table[0], (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), 0)      # one ear's table takes the device image
        with pytest.raises(capi.RvbError):
            ctx.ir_configure_hrtf(mic, None, (0.0, 0.0, 1.0), (0.0, 1.0, 0.0))
    finally:
        ctx.close()


def test_a_shared_scene_outlives_its_first_holder_and_a_new_scene_on_one_context_leaves_the_other_alone():
    """rvb_share_scene: the sharer traces the same bytes as the holder; the holder may then load another scene or be destroyed —
    the sharer keeps the buffers it was given; a context without a scene cannot be shared from."""
    from parallel_reverb_raytracer_amd import capi
    hall, _ = scenes.concert_hall(5000)
    room = scenes.shoebox()
    dirs = scenes.sphere_directions(4000, seed=3)
    src, mic = scenes.source_mic_pairs(1, seed=9)
    mic, src = tuple(float(x) for x in mic[0]), tuple(float(x) for x in src[0])
    holder, sharer, empty = capi.Context(0), capi.Context(0), capi.Context(0)
    try:
        with pytest.raises(capi.RvbError):
            sharer.share_scene(empty)
        holder.set_scene(hall)
        sharer.share_scene(holder)
        assert sharer.scene_info() == holder.scene_info()
        for c in (holder, sharer):
            c.set_directions(dirs)
        want, _ = _solo(holder, mic, src, capi.IR_EXACT, capi.IR_ALL, True)
        got, _ = _solo(sharer, mic, src, capi.IR_EXACT, capi.IR_ALL, True)
        assert np.array_equal(got, want) and want.any()
        holder.set_scene(room)                        # the holder moves on: a scene of its own, the shared buffers stay the sharer's
        assert holder.scene_info() != sharer.scene_info()
        again, _ = _solo(sharer, mic, src, capi.IR_EXACT, capi.IR_ALL, True)
        assert np.array_equal(again, want)
        holder.close()
        holder = None
        last, _ = _solo(sharer, mic, src, capi.IR_EXACT, capi.IR_ALL, True)
        assert np.array_equal(last, want)
    finally:
        for c in (holder, sharer, empty):
            if c is not None:
                c.close()
