#!/usr/bin/env python3
"""Developer tool: per-kernel timings of the BASELINE.json configurations other than the bench line's (C3 per-GPU share,
C4, C5) on one GPU — the numbers DESIGN.md quotes for them.  Same code path as bench.py (fused IR); the histogram mode is the
bench's default (exact) unless `fast` is given.    python tools/config_bench.py [fast] > gpurun_out/config_bench.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
import torch  # noqa: E402
from parallel_reverb_raytracer_amd import capi, dtypes, scenes  # noqa: E402

MODE = capi.IR_FAST if (len(sys.argv) > 1 and sys.argv[1] == "fast") else capi.IR_EXACT


def run(name, scene, mic, src, nrays, nrefl, model, steps=4):
    ctx = capi.Context(0)
    t0 = time.perf_counter()
    ctx.set_scene(scene)
    build_ms = (time.perf_counter() - t0) * 1e3
    dirs = torch.from_numpy(np.ascontiguousarray(scenes.sphere_directions(nrays, seed=1))).cuda()
    torch.cuda.synchronize()
    ctx.set_directions_device(dirs.data_ptr(), nrays)
    table = scenes.hrtf_synthetic_table() if model == "hrtf" else None
    facing = np.asarray(src, np.float32) - np.asarray(mic, np.float32)
    facing = facing / np.linalg.norm(facing)
    kernels, wall = {}, []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        ctx.trace(mic, src, nrefl, dtypes.AIR_COEFFICIENTS)
        cands, direct = ctx.get_image_candidates(), ctx.get_direct()
        trace_t = dict(ctx.last_timings())
        images = capi.merge_images(cands, direct, False)
        if model == "hrtf":
            ctx.ir_configure_hrtf(mic, table, facing, (0, 1, 0), capi.IR_ALL, images)
        else:
            ctx.ir_configure_speakers(mic, [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], capi.IR_ALL, images)
        lo, hi = ctx.ir_time_range()
        range_t = dict(ctx.last_timings())
        nbins = ctx.ir_bins(hi, lo, 44100.0)
        hist = torch.zeros((2, 8, nbins), device="cuda", dtype=torch.float32)
        ctx.ir_accumulate_tensor(lo, 44100.0, nbins, MODE, hist)
        ctx.synchronize()
        acc_t = dict(ctx.last_timings())
        if it:                                   # first iteration allocates
            wall.append((time.perf_counter() - t0) * 1e3)
            for d in (trace_t, range_t, acc_t):
                for k, v in d.items():
                    kernels.setdefault(k, []).append(v)
    ms = float(np.mean(wall))
    out = {"config": name, "triangles": int(scene[0].shape[0]), "rays": nrays, "reflections": nrefl, "model": model,
           "ms_per_ir": ms, "ray_bounces_per_sec": nrays * nrefl / (ms * 1e-3), "executed_bounces": int(ctx.executed_bounces()),
           "kernel_ms": {k: float(np.mean(v)) for k, v in kernels.items()}, "nbins": int(nbins), "images": int(images.shape[0]),
           "scene_build_upload_ms": build_ms, "bvh": ctx.scene_info()}
    ctx.close()
    return out


def pairs_pipelined(scene, src, mic, nrays, nrefl, npairs):
    """C5's per-GPU share (8 of the 64 pairs): the pairs as jobs of the two-context pipeline against one at a time."""
    from parallel_reverb_raytracer_amd import distributed
    table = scenes.hrtf_synthetic_table()
    dirs = torch.from_numpy(np.ascontiguousarray(scenes.sphere_directions(nrays, seed=1))).cuda()
    torch.cuda.synchronize()
    ctxs = []
    for _ in range(2):
        c = capi.Context(0)
        if ctxs and not os.environ.get("CONFIG_BENCH_OWN_SCENES"):
            c.share_scene(ctxs[0])              # one copy of the scene for the contexts of the GPU
        else:
            c.set_scene(scene)
        c.set_directions_device(dirs.data_ptr(), nrays)
        ctxs.append(c)
    device = torch.device("cuda", 0)

    def job(p):
        facing = src[p] - mic[p]
        facing = facing / np.linalg.norm(facing)
        return ((mic[p], src[p], nrefl, dtypes.AIR_COEFFICIENTS),
                dict(model=distributed.HrtfModel(table, facing, (0, 1, 0)), sample_rate=44100.0, trim_predelay=True,
                     mode=MODE, device=device))

    jobs = [job(p) for p in range(npairs)]
    out = {}
    for label, tracers in (("one_at_a_time", ctxs[:1]), ("two_contexts", ctxs)):
        pipe = distributed.IrPipeline(tracers)
        pipe.run_jobs(jobs[:2])                              # warm-up
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.run_jobs(jobs)
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        out[label + "_ms_per_pair"] = (time.perf_counter() - t0) * 1e3 / npairs
    # the same pairs, 4 and 8 per launch (Context.trace_pairs), the launches alternating between the two contexts
    pair_list = [(mic[p], src[p]) for p in range(npairs)]

    def model_for(p):
        facing = src[p] - mic[p]
        return distributed.HrtfModel(table, facing / np.linalg.norm(facing), (0, 1, 0))

    for per_launch in (4, 8):
        distributed.generate_pair_irs(ctxs, pair_list, nrefl, dtypes.AIR_COEFFICIENTS, model_for, 44100.0, device=device, pairs_per_launch=per_launch, mode=MODE)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        distributed.generate_pair_irs(ctxs, pair_list, nrefl, dtypes.AIR_COEFFICIENTS, model_for, 44100.0, device=device, pairs_per_launch=per_launch, mode=MODE)
        torch.cuda.synchronize()
        out["%d_pairs_per_launch_ms_per_pair" % per_launch] = (time.perf_counter() - t0) * 1e3 / npairs
    for c in ctxs:
        c.close()
    out.update({"config": "C5 per-GPU share: %d pairs, concert-hall stand-in, %d rays x %d, HRTF" % (npairs, nrays, nrefl),
                "ray_bounces_per_sec": nrays * nrefl / (out["two_contexts_ms_per_pair"] * 1e-3), "ms_per_ir": out["two_contexts_ms_per_pair"],
                "kernel_ms": {}})
    return out


def native_pipeline(name, scene, jobs, nrays, nrefl, hrtf, contexts=4, repeats=3):
    """The configuration's impulse responses back to back through the pipeline behind the C-ABI (rvb_pipeline_*, csrc/pipeline.hip):
    jobs = [(mic, source)] (HRTF: every listener faces its source), each `repeats` times; ms per impulse response with the histogram
    in the pipeline's pinned ring."""
    dirs = torch.from_numpy(np.ascontiguousarray(scenes.sphere_directions(nrays, seed=1))).cuda()
    torch.cuda.synchronize()
    ctxs = []
    for _ in range(contexts):
        c = capi.Context(0)
        if ctxs and not os.environ.get("CONFIG_BENCH_OWN_SCENES"):
            c.share_scene(ctxs[0])              # one copy of the scene for the contexts of the GPU
        else:
            c.set_scene(scene)
        c.set_directions_device(dirs.data_ptr(), nrays)
        ctxs.append(c)
    pipe = capi.Pipeline(ctxs)
    if hrtf:
        pipe.configure_hrtf(scenes.hrtf_synthetic_table(), (0, 0, 1), (0, 1, 0), nrefl, dtypes.AIR_COEFFICIENTS, 44100.0, True, MODE)
    else:
        pipe.configure_speakers([(-1, 0, -1), (1, 0, -1)], [0.5, 0.5], nrefl, dtypes.AIR_COEFFICIENTS, 44100.0, True, MODE)

    def run(todo):
        sent = taken = 0
        nbins = 0
        while taken < len(todo):
            while sent < len(todo) and pipe.pending() < 2 * contexts:
                mic, src = todo[sent]
                if hrtf:
                    f = np.asarray(src, np.float64) - np.asarray(mic, np.float64)
                    pipe.submit(mic, src, tuple(f / np.linalg.norm(f)), (0.0, 1.0, 0.0))
                else:
                    pipe.submit(mic, src)
                sent += 1
            _, info = pipe.next(copy=False)
            nbins = info["nbins"]
            taken += 1
        return nbins
    run(jobs[:contexts])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nbins = run(jobs * repeats)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / (len(jobs) * repeats)
    pipe.close()
    for c in ctxs:
        c.close()
    return {"config": name + " — native pipeline, %d contexts" % contexts, "triangles": int(scene[0].shape[0]), "rays": nrays, "reflections": nrefl,
            "model": "hrtf" if hrtf else "speakers", "ms_per_ir": ms, "ray_bounces_per_sec": nrays * nrefl / (ms * 1e-3), "nbins_last": int(nbins), "kernel_ms": {}}


def main():
    results = []
    if os.environ.get("CONFIG_BENCH_ONLY") == "c5":      # the C5 pipeline leg alone, a few times (CONFIG_BENCH_OWN_SCENES=1: a scene copy per context)
        hall, _ = scenes.concert_hall(30000)
        src, mic = scenes.source_mic_pairs(64, seed=0)
        for _ in range(3):
            r = native_pipeline("C5 per-GPU share: 8 pairs, 100k rays x 128, HRTF", hall, [(tuple(map(float, mic[p])), tuple(map(float, src[p]))) for p in range(8)], 100000, 128, True, contexts=4)
            print("%s: %.3f ms per IR" % (r["config"], r["ms_per_ir"]))
        return
    # round 4: the configurations through the pipeline behind the C-ABI, each in a process state of its own as far as this tool goes
    # (measured first: a leg that runs after the one-context legs below reads 10-20 % slower on the same box)
    hall, _ = scenes.concert_hall(30000)
    src, mic = scenes.source_mic_pairs(64, seed=0)
    for nctx in (4, 2):
        results.append(native_pipeline("C5 per-GPU share: 8 pairs, 100k rays x 128, HRTF", hall, [(tuple(map(float, mic[p])), tuple(map(float, src[p]))) for p in range(8)],
                                       100000, 128, True, contexts=nctx))
    scene, info = scenes.cathedral(75000)
    results.append(native_pipeline("C3 per-GPU share: 125k rays x 128", scene, [(info["mic"], info["source"])] * 8, 125000, 128, False))
    scene, info = scenes.atrium(262000)
    results.append(native_pipeline("C4: atrium stand-in, 100k rays x 256", scene, [(info["mic"], info["source"])] * 6, 100000, 256, False, repeats=2))
    results.append(pairs_pipelined(hall, src, mic, 100000, 128, 8))
    scene, info = scenes.cathedral(75000)
    results.append(run("C3 per-GPU share: cathedral stand-in, 125k rays x 128", scene, info["mic"], info["source"], 125000, 128, "speakers"))
    scene, info = scenes.atrium(262000)
    results.append(run("C4: atrium stand-in, 100k rays x 256", scene, info["mic"], info["source"], 100000, 256, "speakers"))
    results.append(run("C5 one of 64 pairs: concert-hall stand-in, 100k rays x 128, HRTF", hall, mic[0], src[0], 100000, 128, "hrtf"))
    print(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
