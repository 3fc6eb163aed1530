"""world_size-2 `gloo` test of the N>1 path (distributed.generate_ir): ray-range shards, candidate
all-gather + lowest-ray-wins merge, time-range all-reduce, histogram all-reduce(sum).  The per-rank
compute is the CPU oracle (tests/oracle_tracer.py), so this runs without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, total_rays, nrefl, capacity=None, chain=False, hrtf=False, blocks=8):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rvb_import
    rvb_import.load()
    import torch.distributed as dist
    import pyoracle
    from oracle_tracer import OracleTracer
    from parallel_reverb_raytracer_amd import capi, distributed, dtypes, scenes
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if capacity is not None:
        distributed.EXCHANGE_CAPACITY = capacity     # force the second (overflow) round of the control all-gather
    scene, info = scenes.cathedral(1200)
    first, count = distributed.shard_range(total_rays, rank, world)
    dirs = scenes.sphere_directions(count, seed=9, first=first)
    tracer = OracleTracer(pyoracle.Oracle("port"), scene, dirs)
    import torch
    landed = []

    def host_out(shape):                              # where bench.py asks for the finished histogram on the host (pinned memory on a GPU box)
        landed.append(torch.full(shape, float("nan"), dtype=torch.float32))
        return landed[-1]

    model = distributed.HrtfModel(scenes.hrtf_synthetic_table(), (1.0, 0.0, 0.2), (0.0, 1.0, 0.0)) if hrtf else None
    hist, meta = distributed.generate_ir(tracer, info["mic"], info["source"], nrefl, dtypes.AIR_COEFFICIENTS,
                                         [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], 44100.0, trim_predelay=True,
                                         mode=capi.IR_EXACT, rank=rank, world=world, ray_offset=first, device="cpu", host_out=host_out,
                                         chain_exact=chain, chain_blocks=blocks, model=model)
    assert len(landed) == 1 and meta["host"] is landed[0] and torch.equal(landed[0], hist)     # the host copy is the REDUCED histogram
    np.savez(os.path.join(out_dir, "rank%d_of%d%s%s%s.npz" % (rank, world, "" if capacity is None else "_cap%d" % capacity, "_chain" if chain else "", "_hrtf" if hrtf else "")), hist=hist.numpy(), nbins=meta["nbins"],
             predelay=meta["predelay"], images=meta["images"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from parallel_reverb_raytracer_amd import distributed
    for total in (0, 1, 7, 64, 100000, 1000003):
        for world in (1, 2, 3, 8):
            got = [distributed.shard_range(total, r, world) for r in range(world)]
            assert sum(c for _, c in got) == total
            assert all(got[r][0] + got[r][1] == got[r + 1][0] for r in range(world - 1)) and got[0][0] == 0
            assert max(c for _, c in got) - min(c for _, c in got) <= 1


def test_two_rank_gloo_equals_single_process(tmp_path, oracle):
    import torch.multiprocessing as mp
    total_rays, nrefl = 96, 10
    _worker(0, 1, 0, str(tmp_path), total_rays, nrefl)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), total_rays, nrefl), nprocs=2, join=True)
    one = np.load(os.path.join(str(tmp_path), "rank0_of1.npz"))
    two = [np.load(os.path.join(str(tmp_path), "rank%d_of2.npz" % r)) for r in (0, 1)]
    # both ranks hold the same reduced histogram, binned with the global predelay / global length
    assert np.array_equal(two[0]["hist"], two[1]["hist"])
    assert int(two[0]["nbins"]) == int(one["nbins"]) and float(two[0]["predelay"]) == float(one["predelay"])
    assert int(two[0]["images"]) == int(one["images"]) and int(two[1]["images"]) == 0     # only rank 0 adds the merged images
    # the shard sums differ from the serial order only by float re-association
    a, b = one["hist"].astype(np.float64), two[0]["hist"].astype(np.float64)
    scale = np.abs(a).max(axis=2, keepdims=True) + 1e-30
    assert (np.abs(a - b) <= 1e-5 * scale).all()
    assert np.abs(a).sum() > 0


def test_two_rank_chained_exact_mode_is_bit_identical_to_a_single_process(tmp_path, oracle):
    """chain_exact: the ranks continue ONE serial float sum in ray order (rank 0's impulses, then rank 1's on top, the merged image
    sources last) instead of all-reducing their own sums: the result must be the single-process histogram bit for bit, on both ranks."""
    import torch.multiprocessing as mp
    total_rays, nrefl = 96, 10
    _worker(0, 1, 0, str(tmp_path), total_rays, nrefl)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), total_rays, nrefl, None, True), nprocs=2, join=True)
    one = np.load(os.path.join(str(tmp_path), "rank0_of1.npz"))
    two = [np.load(os.path.join(str(tmp_path), "rank%d_of2_chain.npz" % r)) for r in (0, 1)]
    assert np.array_equal(two[0]["hist"], one["hist"]) and np.array_equal(two[1]["hist"], one["hist"]) and one["hist"].any()
    assert int(two[0]["nbins"]) == int(one["nbins"]) and int(two[1]["images"]) == int(one["images"])


def test_two_rank_chained_exact_mode_with_the_hrtf_model_and_three_blocks(tmp_path, oracle):
    """The same chain with the HRTF listener model (each ear has its own bins) and the histogram travelling in three bin-range
    blocks: rank 1 folds block k while rank 0 is already folding block k + 1."""
    import torch.multiprocessing as mp
    total_rays, nrefl = 64, 8
    _worker(0, 1, 0, str(tmp_path), total_rays, nrefl, None, False, True)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), total_rays, nrefl, None, True, True, 3), nprocs=2, join=True)
    one = np.load(os.path.join(str(tmp_path), "rank0_of1_hrtf.npz"))
    two = [np.load(os.path.join(str(tmp_path), "rank%d_of2_chain_hrtf.npz" % r)) for r in (0, 1)]
    assert np.array_equal(two[0]["hist"], one["hist"]) and np.array_equal(two[1]["hist"], one["hist"]) and one["hist"].any()


def test_four_rank_chain_in_five_blocks_and_all_reduce(tmp_path, oracle):
    """More ranks than blocks in flight: four ranks, the histogram in five bin-range blocks — bit-equal to one process on every rank;
    the same four ranks with the all-reduce of their own serial sums: equal on every rank, within the stated tolerance of one process."""
    import torch.multiprocessing as mp
    total_rays, nrefl = 96, 10
    _worker(0, 1, 0, str(tmp_path), total_rays, nrefl)
    one = np.load(os.path.join(str(tmp_path), "rank0_of1.npz"))
    mp.spawn(_worker, args=(4, _free_port(), str(tmp_path), total_rays, nrefl, None, True, False, 5), nprocs=4, join=True)
    chain = [np.load(os.path.join(str(tmp_path), "rank%d_of4_chain.npz" % r)) for r in range(4)]
    assert all(np.array_equal(c["hist"], one["hist"]) for c in chain) and one["hist"].any()
    mp.spawn(_worker, args=(4, _free_port(), str(tmp_path), total_rays, nrefl), nprocs=4, join=True)
    summed = [np.load(os.path.join(str(tmp_path), "rank%d_of4.npz" % r)) for r in range(4)]
    assert all(np.array_equal(c["hist"], summed[0]["hist"]) for c in summed)
    band_max = np.abs(one["hist"]).max(axis=2, keepdims=True)
    assert (np.abs(summed[0]["hist"].astype(np.float64) - one["hist"]) <= 1e-5 * band_max).all()


def test_candidate_exchange_overflow_round(tmp_path, oracle):
    """A shard with more image-source candidates than the first all-gather carries triggers exactly one more
    round; the result is the one of the roomy exchange."""
    import torch.multiprocessing as mp
    total_rays, nrefl = 96, 10
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), total_rays, nrefl), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), total_rays, nrefl, 0), nprocs=2, join=True)
    roomy = np.load(os.path.join(str(tmp_path), "rank0_of2.npz"))
    tight = np.load(os.path.join(str(tmp_path), "rank0_of2_cap0.npz"))
    assert int(roomy["images"]) >= 2, "the case needs at least one candidate besides the direct path"
    assert np.array_equal(roomy["hist"], tight["hist"]) and int(roomy["images"]) == int(tight["images"])


def test_time_range_combination():
    from parallel_reverb_raytracer_amd import distributed
    assert distributed.combine_time_ranges([(0.0, 0.0), (0.5, 2.0), (0.25, 1.0)]) == (0.25, 2.0)
    assert distributed.combine_time_ranges([(0.0, 0.0)]) == (0.0, 0.0)
    assert distributed.combine_time_ranges([]) == (0.0, 0.0)


def _pipeline_worker(rank, world, port, out_dir, total_rays, nrefl):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rvb_import
    rvb_import.load()
    import torch.distributed as dist
    import pyoracle
    from oracle_tracer import OracleTracer
    from parallel_reverb_raytracer_amd import capi, distributed, dtypes, scenes
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, info = scenes.cathedral(1200)
    first, count = distributed.shard_range(total_rays, rank, world)
    dirs = scenes.sphere_directions(count, seed=9, first=first)
    tracers = [OracleTracer(pyoracle.Oracle("port"), scene, dirs) for _ in range(2)]
    got = []
    pipe = distributed.IrPipeline(tracers)
    kwargs = dict(speakers_dir=[(-1, 0, -1), (1, 0, -1)], speakers_coeff=[0.5, 0.5], sample_rate=44100.0, trim_predelay=True,
                  mode=capi.IR_EXACT, rank=rank, world=world, ray_offset=first, device="cpu")
    args = (info["mic"], info["source"], nrefl, dtypes.AIR_COEFFICIENTS)
    pipe.run(3, args, kwargs, lambda hist, meta, tracer: got.append((hist.numpy().copy(), meta["nbins"], tracers.index(tracer))))
    pipe.run(2, args, kwargs, lambda hist, meta, tracer: got.append((hist.numpy().copy(), meta["nbins"], tracers.index(tracer))))
    np.savez(os.path.join(out_dir, "pipe_rank%d_of%d.npz" % (rank, world)), hists=np.stack([g[0] for g in got]),
             slots=np.array([g[2] for g in got]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_ir_pipeline_alternates_contexts_and_changes_nothing(tmp_path, oracle):
    """IrPipeline (two contexts, the next IR's trace enqueued before the current IR is finished) returns, IR by IR, what
    generate_ir returns — with one rank and with two gloo ranks (collectives stay in program order)."""
    import torch.multiprocessing as mp
    total_rays, nrefl = 64, 8
    _worker(0, 1, 0, str(tmp_path), total_rays, nrefl)
    plain = np.load(os.path.join(str(tmp_path), "rank0_of1.npz"))["hist"]
    _pipeline_worker(0, 1, 0, str(tmp_path), total_rays, nrefl)
    one = np.load(os.path.join(str(tmp_path), "pipe_rank0_of1.npz"))
    assert list(one["slots"]) == [0, 1, 0, 1, 0]                       # alternates, and continues across run() calls
    assert all(np.array_equal(h, plain) for h in one["hists"])
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), total_rays, nrefl), nprocs=2, join=True)
    two = [np.load(os.path.join(str(tmp_path), "pipe_rank%d_of2.npz" % r)) for r in (0, 1)]
    assert np.array_equal(two[0]["hists"], two[1]["hists"])
    assert all(np.array_equal(h, two[0]["hists"][0]) for h in two[0]["hists"])
    a, b = plain.astype(np.float64), two[0]["hists"][0].astype(np.float64)
    assert (np.abs(a - b) <= 1e-5 * (np.abs(a).max(axis=2, keepdims=True) + 1e-30)).all()


def _pairs_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rvb_import
    rvb_import.load()
    import torch.distributed as dist
    import pyoracle
    from oracle_tracer import OracleTracer
    from parallel_reverb_raytracer_amd import capi, distributed, dtypes, scenes
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, _ = scenes.concert_hall(900)
    src, mic = scenes.source_mic_pairs(5, seed=2)
    table = scenes.hrtf_synthetic_table()
    dirs = scenes.sphere_directions(40, seed=4)
    tracers = [OracleTracer(pyoracle.Oracle("port"), scene, dirs) for _ in range(2)]

    def model(i):
        facing = src[i] - mic[i]
        return distributed.HrtfModel(table, facing / np.linalg.norm(facing), (0, 1, 0))

    got = distributed.generate_pair_irs(tracers, [(mic[i], src[i]) for i in range(5)], 6, dtypes.AIR_COEFFICIENTS, model, 44100.0,
                                        rank=rank, world=world, mode=capi.IR_EXACT)
    np.savez(os.path.join(out_dir, "pairs_rank%d_of%d.npz" % (rank, world)), index=np.array(sorted(got)),
             **{"hist%d" % i: got[i][0].numpy() for i in got})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_source_listener_pairs_shard_over_ranks_without_a_collective(tmp_path, oracle):
    """Config C5's decomposition: 5 pairs on 2 ranks (3 + 2), HRTF model, every pair's IR equal to the single-process one."""
    import torch.multiprocessing as mp
    _pairs_worker(0, 1, 0, str(tmp_path))
    mp.spawn(_pairs_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = np.load(os.path.join(str(tmp_path), "pairs_rank0_of1.npz"))
    two = [np.load(os.path.join(str(tmp_path), "pairs_rank%d_of2.npz" % r)) for r in (0, 1)]
    assert list(one["index"]) == [0, 1, 2, 3, 4]
    assert list(two[0]["index"]) == [0, 1, 2] and list(two[1]["index"]) == [3, 4]
    for r in (0, 1):
        for i in two[r]["index"]:
            assert np.array_equal(two[r]["hist%d" % i], one["hist%d" % i])
    assert any(one["hist%d" % i].any() for i in range(5))
