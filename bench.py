#!/usr/bin/env python3
"""bench.py — ray-bounces/s and IR-generation wall-clock of the hot path on MI355X.

One step = one full impulse-response generation for one batch of rays, everything resident in HBM:
    trace (path + image-source + shadow kernels)  ->  image-source candidates to the host, de-dup
    ->  fused attenuate + predelay + time-binning into [channels][8][nbins]  (-> RCCL sum over ranks)
Workload at N=1 = BASELINE.json configs[1]: cathedral stand-in (~75k triangles; Sibenik itself is
not available offline), 100k rays x 128 bounces x 8 bands, two cardioid speakers, 44.1 kHz,
trim_predelay.  With N>1 ranks (one process per GPU, torch.distributed/RCCL) every rank traces its own
100k-ray shard of one seeded global ray set (weak scaling, BASELINE.json configs[2] at N=8 = 800k rays)
and the per-band histograms are summed with one all-reduce; there is no other data-path collective.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import rvb_import  # noqa: E402

rvb_import.load()
from parallel_reverb_raytracer_amd import capi, distributed, dtypes, scenes  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4   # 256 CUs x 4 SIMDs, one VALU wave-instruction per 4 cycles at 2.4 GHz = 614.4 G/s
PMC_FILE = "r01e_pmc_n1.json"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--rays", type=int, default=100000, help="rays per GPU")
    p.add_argument("--reflections", type=int, default=128)
    p.add_argument("--triangles", type=int, default=75000)
    p.add_argument("--sample-rate", type=float, default=44100.0)
    p.add_argument("--mode", choices=["fast", "exact"], default="fast")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    p.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    p.add_argument("--rehearse-collectives", action="store_true",
                   help="with one rank: create the process group anyway and run every collective of the multi-GPU path on it")
    p.add_argument("--contexts", type=int, default=4, help="contexts per GPU that take turns (4: the traces of IRs i+2, i+3 are enqueued "
                   "together, beside the grouping / binning / host stages of IRs i, i+1; 2: plain alternation; 1: strictly one IR at a time)")
    return p.parse_args()


def cpu_baseline(scene, mic, src, nrefl, target_seconds):
    """The reference's own kernel text, compiled for the host by oracle/ref/build_ref.sh (oracle/_ref/librvb_ref.so, built in the
    build container and shipped to the GPU box; `kind: "reference"`), OpenMP over rays on the box's CPU share, timed on a bounded
    prefix of the same seeded ray set.  Without that library: this repo's C restatement of it (`kind: "port"`).  Checker code
    used as a *reported baseline*, never as part of the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes
    import pyoracle
    cores = pyoracle.cpu_threads()
    kind = "reference" if pyoracle.have_ref() else "port"
    try:
        oracle = pyoracle.Oracle(kind)
        if kind == "reference":
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(ctypes.c_int(cores))      # its harness uses the OpenMP default otherwise
    except OSError:
        kind, oracle = "port", pyoracle.Oracle("port")
    rays, spent, done = 16 * cores, 0.0, 0              # the reference harness hands out chunks of 16 rays per thread
    while True:
        dirs = scenes.sphere_directions(rays, seed=1, first=done)
        t0 = time.perf_counter()
        oracle.raytrace(scene, mic, src, dirs, nrefl, dtypes.AIR_COEFFICIENTS, nthreads=cores)
        spent += time.perf_counter() - t0
        done += rays
        if spent >= target_seconds or done >= 8192:
            break
        rate = done / spent
        rays = int(max(16 * cores, min(8192 - done, rate * (target_seconds - spent) * 1.1)))
    what = ("the reference's kernel `raytrace` (rayverb/kernel.cpp) compiled for the host" if kind == "reference"
            else "this repo's C restatement of the reference kernel")
    return {"value": done * nrefl / spent, "unit": "ray-bounces/s", "cores": cores, "kind": kind,
            "sample": "%s, first %d rays x %d bounces of the same ray set and scene, brute force over all triangles as the reference does, %.1f s"
                      % (what, done, nrefl, spent)}


def main():
    args = parse()
    # Rank 0 owes the driver ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner to
    # stdout when a communicator is created, gloo its connection notes), so everything that is not the result goes to stderr:
    # file descriptor 1 is pointed at stderr for the whole run and the JSON line is written to the saved original.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if args.share_gpu:
        local_rank = 0
    grouped = world > 1 or args.rehearse_collectives
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    (scene, info) = scenes.cathedral(args.triangles)
    mic, src = info["mic"], info["source"]
    nrays, nrefl, sr = args.rays, args.reflections, args.sample_rate
    speakers_dir, speakers_coeff = [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5]
    mode = capi.IR_FAST if args.mode == "fast" else capi.IR_EXACT

    # this rank's contiguous shard of the global seeded ray set, resident in HBM before timing starts
    dirs = torch.from_numpy(np.ascontiguousarray(scenes.sphere_directions(nrays, seed=1, first=rank * nrays))).to(device)
    torch.cuda.synchronize()
    contexts, scene_ms = [], 0.0
    for _ in range(max(1, args.contexts)):
        c = capi.Context(local_rank)
        t0 = time.perf_counter()
        c.set_scene(scene)
        scene_ms = (time.perf_counter() - t0) * 1e3
        c.set_directions_device(dirs.data_ptr(), nrays)
        contexts.append(c)
    ctx = contexts[0]
    pipeline = distributed.IrPipeline(contexts)

    kernel_ms, solo_ms = {}, {}
    state = {}
    trace_args = (mic, src, nrefl, dtypes.AIR_COEFFICIENTS)

    def ir_kwargs(sink):
        def on_stage(_name, tracer):
            if sink is not None:
                for k, v in tracer.last_timings():
                    sink.setdefault(k, []).append(v)
        return dict(speakers_dir=speakers_dir, speakers_coeff=speakers_coeff, sample_rate=sr, trim_predelay=True, mode=mode,
                    rank=rank, world=world, ray_offset=rank * nrays, device=device, on_stage=on_stage, collectives=grouped)

    def keep(hist, info, _tracer):
        state.update(hist=hist, nbins=info["nbins"], images=info["images"], predelay=info["predelay"])

    def fence():
        for c in contexts:
            c.synchronize()
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: every context allocates its buffers, then one IR strictly alone for the per-kernel "solo" durations and the
    # single-IR latency (the timed region below overlaps two IRs when --contexts 2)
    for c in contexts:
        distributed.generate_ir(c, *trace_args, **ir_kwargs(None))
    fence()
    solo_irs = 4
    t0 = time.perf_counter()
    for _ in range(solo_irs):
        distributed.generate_ir(ctx, *trace_args, **ir_kwargs(solo_ms))
        ctx.synchronize()
    solo_latency_ms = (time.perf_counter() - t0) * 1e3 / solo_irs
    pipeline.run(args.warmup, trace_args, ir_kwargs(None), keep)
    fence()
    t0 = time.perf_counter()
    pipeline.run(args.steps, trace_args, ir_kwargs(kernel_ms), keep)
    fence()
    elapsed = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    ms_per_step = elapsed / args.steps * 1e3
    bounces_per_step = world * nrays * nrefl
    value = bounces_per_step / (elapsed / args.steps)
    executed = ctx.executed_bounces()

    attenuate_probe = None
    if rank == 0:
        # the materialised attenuate kernel (reference kernel `attenuate`, what SpeakerAttenuator::attenuate launches per
        # channel): 64 B read + 64 B written per impulse, on the traced impulses of this very workload, HBM to HBM.
        d_in, n_imp = ctx.diffuse_device()
        out_buf = torch.empty(n_imp * 64, dtype=torch.uint8, device=device)
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            ctx.attenuate_speaker_device(mic, d_in, n_imp, speakers_dir[0], speakers_coeff[0], out_buf.data_ptr())
            ctx.synchronize()
            times.append(dict(ctx.last_timings()).get("attenuate_kernel"))
        attenuate_probe = {"impulses": int(n_imp), "ms": float(np.mean(times[1:])), "bytes": 128.0 * n_imp}
        del out_buf
    if rank == 0:
        avg = {k: float(np.mean(v)) for k, v in kernel_ms.items()}
        # algorithmic bytes (materialised formulation; split per kernel in DESIGN.md)
        algorithmic = {
            "record_sort_kernels": 0.0,
            "path_kernel": 16.0 * nrays,
            "image_kernel": 720.0 * nrays,
            "shadow_kernel": 64.0 * nrays * nrefl,
            "time_range_kernel": 64.0 * nrays * nrefl,
            "histogram_fast_kernel": 64.0 * nrays * nrefl,
        }
        solo = {k: float(np.mean(v)) for k, v in solo_ms.items()}       # one IR alone on the GPU (untimed pass above)
        trace_ms = sum(solo.get(k, 0.0) for k in ("path_kernel", "image_kernel", "shadow_kernel"))
        dominant = max(solo, key=solo.get)                              # by the time the kernel itself needs
        # Kernel durations for the rooflines come from the solo pass (HIP events around each launch with one IR on the GPU): in the
        # timed region the kernels of two IRs interleave and the events around a launch then span its neighbour's work as well.
        # rocprofv3 --kernel-trace --stats of `bench.py --contexts 1` agrees with them (profiles/).
        ach = algorithmic.get(dominant, 0.0) / (solo[dominant] * 1e-3) / 1e9
        # HBM traffic per launch from the committed PMC passes of this same command (profiles/), if the workload matches
        traffic, valu_insts, valu_per_ir = {}, {}, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
            if (nrays, nrefl, args.triangles, world) == (100000, 128, 75000, 1):
                # per IR: the binning kernel runs twice per IR (diffuse impulses, then the few image sources) under one timing label
                traffic = {k: v.get("hbm_bytes_per_ir", v["hbm_bytes_per_launch"]) for k, v in pmc["kernels"].items() if "hbm_bytes_per_launch" in v}
                valu_insts = {k: v["SQ_INSTS_VALU"] for k, v in pmc["kernels"].items() if "SQ_INSTS_VALU" in v}
                valu_per_ir = pmc.get("valu_wave_instructions_per_ir")
        except (OSError, ValueError, KeyError):
            pass
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic.get(dominant),
                    "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)" % PMC_FILE if traffic else None,
                    "avg_launch_ms": solo[dominant], "elapsed_ms_in_timed_region": avg.get(dominant),
                    "note": "avg_launch_ms: HIP events around the launch with one IR on the GPU (in the timed region two IRs share it); "
                            "trace kernels are VALU-issue-bound by construction (scene + BVH are cache-resident): see roofline_valu; "
                            "the HBM-bound kernels of the path are reported in roofline_stream"}
        # What actually bounds the trace kernels: VALU issue.  achieved = VALU wave-instructions (PMC SQ_INSTS_VALU of this same
        # command, a property of the workload) / time; peak = 1024 SIMDs x one wave-instruction per 4 cycles.  Per kernel against
        # its duration when it has the GPU to itself (kernel_ms_solo: in the timed region two IRs share the GPU and a kernel's
        # elapsed time includes its neighbour's work); whole_step = every kernel of one IR against the timed region's ms_per_step.
        valu = {}
        for k in ("path_kernel", "shadow_kernel"):
            if k in valu_insts and solo.get(k):
                a = valu_insts[k] / (solo[k] * 1e-3) / 1e9
                valu[k] = {"bound": "valu_issue", "achieved": a, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                           "frac": a / VALU_PEAK_GINST, "avg_launch_ms": solo[k], "valu_instructions_per_launch": valu_insts[k]}
        if valu_per_ir:
            a = valu_per_ir * world / (elapsed / args.steps) / 1e9
            valu["whole_step"] = {"bound": "valu_issue", "achieved": a, "peak": VALU_PEAK_GINST * world, "unit": "G wave-instructions/s",
                                  "frac": a / (VALU_PEAK_GINST * world), "valu_instructions_per_ir": valu_per_ir}
        stream = {}
        for k in ("shadow_kernel", "time_range_kernel", "histogram_fast_kernel"):
            if k in solo and solo[k] > 0:                # against the kernel's duration with the GPU to itself
                a = algorithmic[k] / (solo[k] * 1e-3) / 1e9
                stream[k] = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                             "avg_launch_ms": solo[k], "traffic": traffic.get(k)}
        if attenuate_probe and attenuate_probe["ms"]:
            a = attenuate_probe["bytes"] / (attenuate_probe["ms"] * 1e-3) / 1e9
            stream["attenuate_kernel"] = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                                          "avg_launch_ms": attenuate_probe["ms"], "traffic": None,
                                          "note": "materialised attenuate of the workload's %d traced impulses, HBM to HBM, 64 B read + 64 B written each" % attenuate_probe["impulses"]}
        out = {
            "metric": "ray_bounces_per_sec", "value": value, "unit": "ray-bounces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cathedral stand-in (Sibenik unavailable offline), %d triangles, %d rays/GPU x %d bounces x 8 bands, "
                                   "2 cardioid speakers, %.0f Hz, trim_predelay, output_mode all, histogram mode %s"
                                   % (scene[0].shape[0], nrays, nrefl, sr, args.mode),
                       "triangles": int(scene[0].shape[0]), "rays_per_gpu": nrays, "reflections": nrefl,
                       "sharding": "ray-range shards, replicated scene, one all-reduce(sum) of [2][8][nbins] histograms",
                       "pipelining": "%d contexts per GPU take turns: traces are enqueued %d at a time, the next group before the current one is finished"
                                     % (len(contexts), max(1, len(contexts) // 2))},
            "ir_gen_wall_ms": solo_latency_ms, "contexts_per_gpu": len(contexts),
            "trace_only_ray_bounces_per_sec": (nrays * nrefl) / (trace_ms * 1e-3) if trace_ms else None,
            "executed_bounces_rank0": int(executed), "nominal_bounces_rank0": nrays * nrefl,
            "kernel_elapsed_ms_timed_region": avg, "kernel_ms": solo, "nbins": state["nbins"], "image_sources": state["images"], "predelay_s": state["predelay"],
            "scene_build_upload_ms": scene_ms, "bvh": ctx.scene_info(),
            "roofline": roofline, "roofline_stream": stream, "roofline_valu": valu,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, mic, src, nrefl, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
