"""BASELINE configs C3 and C5 at their FULL decomposition, rehearsed on the one GPU of the test box.

C3 = 1 000 000 rays x 128 bounces as eight contiguous ray shards + one reduce of the per-band histograms: the reference
processes its rays in independent groups (rayverb.cpp:586-591) and merges image sources lowest-ray-first (rayverb.cpp:654-676);
here the eight shards are the eight "devices" of an rvb_multi that lists GPU 0 eight times (capi.MultiContext([0] * 8)) — the
same code path, threads, merges and exact-mode chain that eight physical GPUs run, minus xGMI.

C5 = 64 (source, listener) pairs, HRTF, eight blocks of eight pairs (one block per GPU of the node, no collective at all):
here every block runs, one after the other, through the two-context pipeline a rank uses.

More than one physical GPU is the driver's scaling run; what is checked here is that the decomposition itself is exact."""
import zlib

import numpy as np
import pytest

from parallel_reverb_raytracer_amd import scenes
from parallel_reverb_raytracer_amd.dtypes import AIR_COEFFICIENTS

pytestmark = pytest.mark.gpu

SPEAKERS = ([(-1, 0, -1), (1, 0, -1)], [0.5, 0.5])


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8))


def _same(a, b):
    return all(np.array_equal(a[f], b[f]) for f in ("volume", "time")) and np.array_equal(a["position"][:, :3], b["position"][:, :3])


def test_c3_one_million_rays_as_eight_ray_shards_and_one_reduce(oracle):
    import torch
    from parallel_reverb_raytracer_amd import capi, distributed
    scene, info = scenes.cathedral(75000)
    mic, src = info["mic"], info["source"]
    world, total, nrefl, sr = 8, 1000000, 128, 44100.0
    dirs = scenes.sphere_directions(total, seed=1)

    # ---- the reference of the comparison: the same eight slices, each traced by ONE plain context -------------------------
    ctx = capi.Context(0)
    device = torch.device("cuda", 0)
    shard_crc, cands, ranges, first_of = [], [], [], []
    try:
        ctx.set_scene(scene)
        sample_rays = np.sort(np.random.default_rng(17).choice(total, 64, replace=False))
        sampled = {}
        for g in range(world):
            first, n = distributed.shard_range(total, g, world)
            assert n == 125000
            first_of.append(first)
            ctx.set_directions(dirs[first:first + n])
            ctx.trace(mic, src, nrefl, AIR_COEFFICIENTS, ray_offset=first)
            diffuse = ctx.get_raw_diffuse()
            shard_crc.append(_crc(diffuse))
            mine = sample_rays[(sample_rays >= first) & (sample_rays < first + n)]
            for r in mine:
                sampled[int(r)] = diffuse.reshape(n, nrefl)[r - first].copy()
            del diffuse
            cands.append(ctx.get_image_candidates())
            ctx.ir_configure_speakers(mic, SPEAKERS[0], SPEAKERS[1], capi.IR_DIFFUSE, None)
            ranges.append(ctx.ir_time_range())
        direct = ctx.get_direct()
        # sampled rays of the global set against brute force over all 75 k triangles (a ray does not depend on its shard)
        want, _, _ = oracle.raytrace(scene, mic, src, dirs[sample_rays], nrefl, AIR_COEFFICIENTS)
        want = want.reshape(len(sample_rays), nrefl)
        for k, r in enumerate(sample_rays):
            assert _same(sampled[int(r)], want[k]), r
        images = capi.merge_images(np.concatenate(cands[::-1]), direct, False)          # (order of the shards must not matter)
        ctx.ir_configure_speakers(mic, SPEAKERS[0], SPEAKERS[1], capi.IR_IMAGES, images)
        lo, hi = distributed.combine_time_ranges(ranges + [ctx.ir_time_range()])
        nbins = ctx.ir_bins(hi, lo, sr)
        # the 8-step chain of single contexts: every shard is re-traced and folds its impulses, in ray order, on top of what the
        # histogram holds; the merged image sources go last (reference order: rayverb.cpp:708-714)
        chain = torch.zeros((2, 8, nbins), device=device, dtype=torch.float32)
        for g in range(world):
            first, n = distributed.shard_range(total, g, world)
            ctx.set_directions(dirs[first:first + n])
            ctx.trace(mic, src, nrefl, AIR_COEFFICIENTS, ray_offset=first)
            ctx.ir_configure_speakers(mic, SPEAKERS[0], SPEAKERS[1], capi.IR_DIFFUSE, None)
            ctx.ir_accumulate_tensor(lo, sr, nbins, capi.IR_EXACT, chain)
            ctx.synchronize()
        ctx.ir_configure_speakers(mic, SPEAKERS[0], SPEAKERS[1], capi.IR_IMAGES, images)
        ctx.ir_accumulate_tensor(lo, sr, nbins, capi.IR_EXACT, chain)
        ctx.synchronize()
        chain = chain.cpu().numpy()
    finally:
        ctx.close()
    assert chain.any() and nbins > 500000

    # ---- the decomposition under test: eight shards behind the C-ABI ------------------------------------------------------
    m = capi.MultiContext([0] * world)
    try:
        m.set_scene(scene)
        m.raytrace(mic, src, dirs, nrefl, AIR_COEFFICIENTS)
        shards = [m.shard(g) for g in range(world)]
        assert shards == [distributed.shard_range(total, g, world) for g in range(world)]
        diffuse = m.get_raw_diffuse().reshape(total, nrefl)                  # 8.2 GB, every shard writes its slice
        for g, (first, n) in enumerate(shards):
            assert _crc(diffuse[first:first + n]) == shard_crc[g], "shard %d differs from the same slice traced by one context" % g
        for k, r in enumerate(sample_rays):
            assert _same(diffuse[r], want[k])
        del diffuse
        assert _same(m.get_raw_images(False), images)
        exact = m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], True, sr, capi.IR_EXACT)
        assert exact.shape == chain.shape and np.array_equal(exact, chain), "exact-mode IR of the eight shards != the chained single contexts"
        fast = m.ir_speakers(mic, SPEAKERS[0], SPEAKERS[1], True, sr, capi.IR_FAST)
        band_max = np.abs(exact.astype(np.float64)).max(axis=2, keepdims=True)
        worst = float((np.abs(fast.astype(np.float64) - exact) / band_max).max())
        print("C3 (1M rays, 8 shards): nbins %d, images %d, fast-vs-exact max |err| / band max %.3g" % (nbins, images.shape[0], worst))
        assert worst <= 1e-5 and fast.any()
    finally:
        m.close()


def test_c5_all_64_pairs_hrtf_as_eight_blocks_of_eight(oracle):
    """Every one of the 64 (source, listener) pairs of the hall stand-in at 100 000 rays x 128 bounces, HRTF, exact mode, the way
    eight ranks run them: rank r takes pairs shard_range(64, r, 8) through generate_pair_irs (two contexts, four pairs per
    launch).  Each pair's [2][8][nbins] histogram must be bit-identical to that pair traced and binned alone on a third context;
    sampled rays of a few pairs against brute force; blocks are disjoint and cover all pairs."""
    import torch
    from parallel_reverb_raytracer_amd import capi, distributed
    scene, _ = scenes.concert_hall(30000)
    src, mic = scenes.source_mic_pairs(64, seed=0)
    table = scenes.hrtf_synthetic_table()
    nrays, nrefl, world = 100000, 128, 8
    dirs = scenes.sphere_directions(nrays, seed=1)
    device = torch.device("cuda", 0)
    contexts = [capi.Context(0) for _ in range(3)]
    try:
        for c in contexts:
            c.set_scene(scene)
            c.set_directions(dirs)
        solo = contexts[2]

        def model_for(p):
            facing = src[p] - mic[p]
            return distributed.HrtfModel(table, facing / np.linalg.norm(facing), (0, 1, 0))

        pairs = [(mic[p], src[p]) for p in range(64)]
        seen, audible = [], 0
        rng = np.random.default_rng(21)
        for rank in range(world):
            block = distributed.generate_pair_irs(contexts[:2], pairs, nrefl, AIR_COEFFICIENTS, model_for, 44100.0, rank=rank, world=world,
                                                  device=device, mode=capi.IR_EXACT, pairs_per_launch=4)
            first, count = distributed.shard_range(64, rank, world)
            assert sorted(block) == list(range(first, first + count)) and count == 8
            for p in sorted(block):
                got, got_info = block[p]
                hist, info = distributed.generate_ir(solo, mic[p], src[p], nrefl, AIR_COEFFICIENTS, model=model_for(p), sample_rate=44100.0,
                                                     trim_predelay=True, mode=capi.IR_EXACT, device=device)
                assert got_info["nbins"] == info["nbins"] and got_info["images"] == info["images"] and got_info["predelay"] == info["predelay"]
                assert torch.equal(got, hist), "pair %d: block result differs from the pair alone" % p
                audible += int(bool(hist.any()))
                if p in (2, 29, 47, 63):
                    sample = np.sort(rng.choice(nrays, 24, replace=False))
                    want, _, _ = oracle.raytrace(scene, mic[p], src[p], dirs[sample], nrefl, AIR_COEFFICIENTS)
                    assert _same(solo.get_raw_diffuse().reshape(nrays, nrefl)[sample].reshape(-1), want)
                seen.append(p)
            del block
            torch.cuda.empty_cache()
        assert seen == list(range(64)) and audible >= 60
    finally:
        for c in contexts:
            c.close()
