#!/usr/bin/env python3
"""bench.py — ray-bounces/s and IR-generation wall-clock of the hot path on MI355X.

One step = one full impulse-response generation for one batch of rays, inputs resident in HBM, result ON THE HOST
("until [channels][8][nbins] is on the host"):
    trace (path + image-source + shadow kernels)  ->  image-source candidates to the host, de-dup
    ->  fused attenuate + predelay + time-binning into [channels][8][nbins]  (-> RCCL sum over ranks)
    ->  the finished histogram copied to pinned host memory on a side stream (an IR counts once it is there)
Workload at N=1 = BASELINE.json configs[1]: cathedral stand-in (~75k triangles; Sibenik itself is
not available offline), 100k rays x 128 bounces x 8 bands, two cardioid speakers, 44.1 kHz,
trim_predelay.  With N>1 ranks (one process per GPU, torch.distributed/RCCL) the workload is configs[2]'s
shape: every rank traces its contiguous 125k-ray shard of one seeded global set of N x 125k rays (N=8: the
config's 1M rays; weak scaling) and the per-band histograms are summed with one all-reduce; there is no other
data-path collective.

The binning runs in EXACT mode by default: every rank's histogram is the reference's serial float sum over its
impulses, bit for bit (tests/test_gpu_fullsize.py checks it against the oracle chain at this very size).  The
float-atomic mode is reported beside it (`fast_mode`) with its measured distance from the exact histogram.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts its N rank processes itself (child processes of a
parent that never touches the GPU) and relays rank 0's line; the driver's own `python -m torch.distributed.run ... bench.py
--gpus N` form is taken as it comes.

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
# the guide's vector issue rate: v_fma_f32 (wave64) 2 cycles on each of 256 x 4 SIMDs at 2.4 GHz
VALU_FMA_PEAK_GINST = 256 * 4 * 2.4 / 2
HELD_CLOCK_GHZ = 2.32      # measured in the pipeline of this bench (profiles/r04c_pair_stamps_n1.txt), against the 2.4 GHz the issue model assumes
PMC_FILE = "r04c_pmc_n1.json"      # committed PMC passes of this command (tools/profile.sh): instruction mix, lane utilisation, HBM bytes
BYTES_PER_BOUNCE = 69.75          # 64 B Impulse per bounce + (16 B direction + 10 x 72 B image slots) per ray at 128 bounces


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200, help="timed impulse responses (200 = about a second: long enough for a stable number and for the driver's samplers)")
    p.add_argument("--warmup", type=int, default=8)
    p.add_argument("--rays", type=int, default=0, help="rays per GPU (default: 100000 on one GPU = config C2, 125000 per GPU otherwise = config C3's share)")
    p.add_argument("--reflections", type=int, default=128)
    p.add_argument("--triangles", type=int, default=75000)
    p.add_argument("--sample-rate", type=float, default=44100.0)
    p.add_argument("--mode", choices=["fast", "exact"], default="exact")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="only the timed region and the per-kernel pass (profiling runs): no fast-mode leg, "
                   "no fast-vs-exact comparison, no API-flow leg, no attenuate probe")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    p.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    p.add_argument("--rehearse-collectives", action="store_true",
                   help="with one rank: create the process group anyway and run every collective of the multi-GPU path on it")
    p.add_argument("--own-scenes", action="store_true", help="every context builds and uploads a copy of the scene of its own (before round 4's rvb_share_scene)")
    p.add_argument("--exact-chain", action="store_true", help="several ranks, exact mode: the ranks continue ONE serial sum in ray order (bit-identical "
                   "to one GPU; their binning stages run one after the other) instead of all-reducing their own serial sums")
    p.add_argument("--no-host-copy", action="store_true", help="leave the finished histograms in HBM (the round-1/2 metric; diagnostic)")
    p.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts its own ranks (0: a free one)")
    p.add_argument("--python-pipeline", action="store_true", help="one GPU: drive the timed region with distributed.IrPipeline (Python over the C-ABI) "
                   "instead of the pipeline behind the C-ABI (rvb_pipeline_*, csrc/pipeline.hip: what a C++ caller gets — the default on one GPU; with "
                   "several ranks the Python pipeline runs, it carries the torch.distributed collectives); the other one is reported beside the line")
    p.add_argument("--native", action="store_true", help="(default on one GPU since round 4; kept so that older command lines still run)")
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
    what = ("the TEXT of the reference's kernel `raytrace` (rayverb/kernel.cpp) compiled for the host with the 13 OpenCL built-ins of "
            "oracle/ref/ref_builtins.cl (this repository's definitions)" if kind == "reference"
            else "this repo's C restatement of the reference kernel")
    return {"value": done * nrefl / spent, "unit": "ray-bounces/s", "cores": cores, "kind": kind,
            "sample": "%s, first %d rays x %d bounces of the same ray set and scene, brute force over all triangles as the reference does, %.1f s"
                      % (what, done, nrefl, spent)}


def fast_vs_exact(fast, exact):
    """Distance of the float-atomic histogram from the serial-order one (torch tensors [channels][8][nbins] on the GPU): the claimed
    tolerance is 1e-5 x the band's largest |value| (sign-alternating volumes cancel inside a bin: a relative bar per band-bin is
    not meaningful where the sum is ~0); the fraction of band-bins outside a pure 1e-5 relative error is reported with it."""
    import torch
    f, e = fast.double(), exact.double()
    err = (f - e).abs()
    band_max = e.abs().amax(dim=2, keepdim=True).clamp_min(1e-300)
    nonzero = e != 0
    rel = torch.where(nonzero, err / e.abs().clamp_min(1e-300), torch.zeros_like(err))
    outside = (rel > 1e-5) | (~nonzero & (err > 0))
    return {"max_abs_err_over_band_max": float((err / band_max).max()), "claimed_tolerance": 1e-5,
            "fraction_band_bins_outside_1e-5_relative": float(outside.double().mean()),
            "band_bins": int(err.numel()), "band_bins_differing": int((err > 0).sum())}


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as CHILD processes — this parent has not made a
    single GPU call (nothing here imports a HIP runtime before this point) and only relays rank 0's JSON line and the exit
    code.  One process per GPU, rendezvous on 127.0.0.1."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["RVB_BENCH_SELF_LAUNCHED"] = "1"
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for raw in proc.stdout:                               # rank 0's one JSON line (everything else the ranks print goes to stderr)
        text = raw.decode(errors="replace")
        if text.lstrip().startswith("{") and '"metric"' in text:
            line = text
        else:
            sys.stderr.write(text)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    if line is not None and rc == 0:
        got = json.loads(line).get("n_gpus")
        if got != args.gpus:
            print("bench.py: %d ranks took part, --gpus %d was asked for" % (got, args.gpus), file=sys.stderr)
            rc = 1
    if line is not None and rc == 0:
        sys.stdout.write(line)
        sys.stdout.flush()
    sys.exit(rc)


def main():
    args = parse()
    args.native = not args.python_pipeline
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                                # does not return
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
    if world != args.gpus or (grouped and dist.get_world_size() != world):
        # a line that says n_gpus N must come from N ranks: refuse anything else (the driver computes scaling from it)
        print("bench.py: --gpus %d but %d rank(s) in the process group (WORLD_SIZE %d)"
              % (args.gpus, dist.get_world_size() if grouped else 1, world), file=sys.stderr)
        sys.exit(3)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    (scene, info) = scenes.cathedral(args.triangles)
    mic, src = info["mic"], info["source"]
    rays_per_gpu = args.rays if args.rays else (100000 if world == 1 else 125000)
    nrefl, sr = args.reflections, args.sample_rate
    speakers_dir, speakers_coeff = [(-1, 0, -1), (1, 0, -1)], [0.5, 0.5]
    mode = capi.IR_FAST if args.mode == "fast" else capi.IR_EXACT

    # this rank's contiguous shard of the global seeded ray set, resident in HBM before timing starts
    first_ray, nrays = distributed.shard_range(rays_per_gpu * world, rank, world)
    dirs = torch.from_numpy(np.ascontiguousarray(scenes.sphere_directions(nrays, seed=1, first=first_ray))).to(device)
    torch.cuda.synchronize()
    contexts, scene_ms = [], 0.0
    for _ in range(max(1, args.contexts)):
        c = capi.Context(local_rank)
        if contexts and not args.own_scenes:
            c.share_scene(contexts[0])          # one hierarchy in HBM (and in the L2s) for all the contexts of this GPU: rvb_share_scene
        else:
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

    def ir_kwargs(sink, ir_mode=None):
        def on_stage(_name, tracer):
            if sink is not None:
                for k, v in tracer.last_timings():
                    sink.setdefault(k, []).append(v)
        return dict(speakers_dir=speakers_dir, speakers_coeff=speakers_coeff, sample_rate=sr, trim_predelay=True,
                    mode=mode if ir_mode is None else ir_mode, host_out=host_out if to_host else None, chain_exact=args.exact_chain,
                    rank=rank, world=world, ray_offset=first_ray, device=device, on_stage=on_stage, collectives=grouped)

    # an impulse response is generated once [channels][8][nbins] is ON THE HOST.  Every histogram is copied to
    # pinned host memory behind its binning (rvb_copy_to_pinned_host_async: 54 MB at C2, about a millisecond of PCIe, on the
    # context's export stream beside the next IRs' kernels); the timed region ends when the last copy has landed (fence() waits for
    # every stream of the device).
    to_host = not args.no_host_copy
    host_ring, in_flight, slot_of = [], {}, {}

    def host_out(shape):
        """A pinned buffer of the ring for the IR being generated.  The ring slot is tied to the IR through the buffer itself (slot_of:
        address -> slot; keep() files the device histogram and its tracer under that slot), not through shared 'last slot' state: the
        pipeline calls this for every IR of a group before it hands any of them to keep()."""
        if not host_ring or tuple(host_ring[0].shape) != tuple(shape):
            torch.cuda.synchronize()
            for c in contexts:
                c.synchronize_exports()
            host_ring[:] = [torch.empty(shape, dtype=torch.float32, pin_memory=True) for _ in range(2 * max(1, args.contexts))]
            in_flight.clear()
            slot_of.clear()
            slot_of.update({h.data_ptr(): i for i, h in enumerate(host_ring)})
            state["slot"] = 0
        slot = state["slot"] % len(host_ring)
        state["slot"] += 1
        if slot in in_flight:                               # the copy that last used this buffer (2 x contexts IRs ago): done long since
            in_flight.pop(slot)[1].synchronize_exports()
        return host_ring[slot]

    def keep(hist, info, tracer):
        state.update(hist=hist, nbins=info["nbins"], images=info["images"], predelay=info["predelay"], host=info.get("host"))
        if info.get("host") is not None:
            # the device histogram stays alive (and its ring slot taken) until its export has been waited for
            in_flight[slot_of[info["host"].data_ptr()]] = (hist, tracer)
        state["kept"] = state.get("kept", 0) + 1

    def fence():
        for c in contexts:
            c.synchronize()
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, kwargs):
        fence()
        t0 = time.perf_counter()
        pipeline.run(steps, trace_args, kwargs, keep)
        fence()
        elapsed = time.perf_counter() - t0
        if grouped:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
        return elapsed

    # untimed: every context allocates its buffers, then a few IRs strictly alone for the per-kernel "solo" durations and the
    # latency of one IR — to the histogram in HBM (ir_gen_wall_ms) and on to pinned host memory (ir_gen_to_host_ms)
    for c in contexts:
        keep(*distributed.generate_ir(c, *trace_args, **ir_kwargs(None)), c)      # (the histogram stays alive until its export has landed)
    fence()
    for c in contexts:
        c.set_concurrent_traces(1)                  # alone on the GPU: what a caller with one IR gets (the pipeline announced its group)
    solo_irs = 4
    t0 = time.perf_counter()
    for _ in range(solo_irs):
        keep(*distributed.generate_ir(ctx, *trace_args, **ir_kwargs(solo_ms)), ctx)
        ctx.synchronize()
        ctx.synchronize_exports()
    solo_latency_ms = (time.perf_counter() - t0) * 1e3 / solo_irs
    in_hbm = dict(ir_kwargs(None), host_out=None)
    for _ in range(solo_irs):
        distributed.generate_ir(ctx, *trace_args, **in_hbm)
    fence()
    t0 = time.perf_counter()
    for _ in range(solo_irs):
        distributed.generate_ir(ctx, *trace_args, **in_hbm)
        ctx.synchronize()
    hbm_latency_ms = (time.perf_counter() - t0) * 1e3 / solo_irs
    to_host_latency_ms, solo_latency_ms = solo_latency_ms, hbm_latency_ms      # (the first solo pass above ran with the host copy)
    for c in contexts:
        c.set_concurrent_traces(pipeline.group_size(len(contexts)))
    if pipeline.group_size(len(contexts)) > 1:
        # the path kernel the pipeline's contexts use (two lanes per ray when two traces are in flight), also alone on the GPU
        pair_solo = {}
        for _ in range(2):
            keep(*distributed.generate_ir(ctx, *trace_args, **ir_kwargs(pair_solo)), ctx)
            ctx.synchronize()
        for k, v in pair_solo.items():
            if k not in solo_ms:
                solo_ms[k] = v

    def timed_native(steps, warmup, native_mode):
        """The same region through rvb_pipeline_* (csrc/pipeline.hip): jobs submitted a few ahead of the results taken, every histogram
        in the pipeline's pinned ring when rvb_pipeline_next returns.  Returns (seconds, last histogram as a tensor, info)."""
        native = capi.Pipeline(contexts, group=int(os.environ.get("RVB_PIPELINE_GROUP", 0)))
        try:
            native.configure_speakers(speakers_dir, speakers_coeff, nrefl, dtypes.AIR_COEFFICIENTS, sr, True, native_mode)

            def run(count):
                submitted = taken = 0
                last = []
                while taken < count:
                    while submitted < count and native.pending() < 2 * len(contexts):
                        native.submit(mic, src)
                        submitted += 1
                    last = (last + [native.next(copy=False)])[-len(contexts):]      # (a result stays valid until `contexts` more have been taken)
                    taken += 1
                return last
            run(warmup)
            fence()
            t0 = time.perf_counter()
            last = run(steps)
            fence()
            seconds = time.perf_counter() - t0
            state["native_landed"] = [torch.from_numpy(v.copy()) for v, _ in last]      # the region's last histograms as they landed in the pinned ring
            view, info = last[-1]
            return seconds, torch.from_numpy(view.copy()), info
        finally:
            native.close()
            for c in contexts:
                c.set_concurrent_traces(pipeline.group_size(len(contexts)))

    native_leg = None
    if args.native and world == 1 and not grouped:
        elapsed, native_hist, native_info = timed_native(args.steps, args.warmup, mode)
        state.update(host=native_hist, hist=native_hist, nbins=native_info["nbins"], images=native_info["images"], predelay=native_info["predelay"])
        host_ring[:] = []                                   # (the ring check below is the Python pipeline's)
        # per-kernel elapsed times of the timed region are the Python pipeline's hook: one short pass for the report
        pipeline.run(4, trace_args, ir_kwargs(kernel_ms), keep)
        state.update(host=native_hist, hist=native_hist, nbins=native_info["nbins"], images=native_info["images"], predelay=native_info["predelay"])
        host_ring[:] = []
    else:
        pipeline.run(args.warmup, trace_args, ir_kwargs(None), keep)
        elapsed = timed(args.steps, ir_kwargs(kernel_ms))

    ms_per_step = elapsed / args.steps * 1e3
    bounces_per_step = world * rays_per_gpu * nrefl
    value = bounces_per_step / (elapsed / args.steps)
    executed = ctx.executed_bounces()

    if rank == 0:
        print("bench.py: timed region %.3f ms per step" % ms_per_step, file=sys.stderr)
    # What the timed region produced must be what one IR generated strictly alone produces: the last histogram of the pipelined,
    # fused (two traces per path-kernel launch) region — as it landed on the host — against a solo IR of the same arguments.
    # Exact mode on one rank: bit for bit (the serial-order sum has one value).  Several ranks add their serial sums with an
    # all-reduce, float atomics have no fixed order: there the bar is the stated tolerance, 1e-5 of each band's largest value.
    timed_hist = (state["host"] if to_host else state["hist"].cpu()).clone()
    # every buffer of the ring holds one of the region's last 2 x contexts IRs as it landed on the host (the timed IRs are identical jobs)
    landed = [h.clone() for h in host_ring] if to_host and args.steps >= len(host_ring) else []
    if args.native and world == 1 and not grouped:
        landed = state.pop("native_landed", [])
    solo_hist, _ = distributed.generate_ir(ctx, *trace_args, **dict(ir_kwargs(None), host_out=None))
    fence()
    solo_host = solo_hist.cpu()
    bit_equal = bool(torch.equal(timed_hist, solo_host))
    ring_equal = [bool(torch.equal(h, solo_host)) if h.shape == solo_host.shape else False for h in landed]
    band_max = solo_host.double().abs().amax(dim=2, keepdim=True).clamp_min(1e-300)
    worst = float(((timed_hist.double() - solo_host.double()).abs() / band_max).max()) if timed_hist.shape == solo_host.shape else float("inf")
    must_be_equal = args.mode == "exact" and (world == 1 or args.exact_chain)
    timed_check = {"last_timed_histogram_equals_solo_ir": bit_equal, "max_abs_err_over_band_max": worst,
                   "landed_histograms_checked": len(ring_equal), "landed_histograms_equal_solo_ir": int(sum(ring_equal)),
                   "required": "bit-equal" if must_be_equal else "<= 1e-5 of each band's maximum"}
    if (must_be_equal and not (bit_equal and all(ring_equal))) or worst > 1e-5 or not bool(solo_host.any()):
        print("bench.py: the timed region's histogram differs from a solo IR: %s" % json.dumps(timed_check), file=sys.stderr)
        sys.exit(4)
    if rank == 0 and os.environ.get("RVB_BENCH_CRC"):       # (checks of the multi-rank modes against one context tracing all the rays)
        import zlib
        print("bench.py: histogram nbins %d crc32 %d" % (solo_host.shape[2], zlib.crc32(solo_host.numpy().tobytes())), file=sys.stderr)
    del solo_hist, solo_host, timed_hist, landed

    # the other binning mode through the same timed pipeline, and how far the float-atomic histogram is from the exact one
    other_mode, comparison = None, None
    if not args.no_extras:
        other = capi.IR_EXACT if mode == capi.IR_FAST else capi.IR_FAST
        if args.native and world == 1 and not grouped:      # (the same driver as the headline)
            other_elapsed = timed_native(args.steps, 2, other)[0]
        else:
            pipeline.run(2, trace_args, ir_kwargs(None, other), keep)
            other_elapsed = timed(args.steps, ir_kwargs(None, other))
        other_mode = {"mode": "exact" if other == capi.IR_EXACT else "fast", "value": bounces_per_step / (other_elapsed / args.steps),
                      "ms_per_step": other_elapsed / args.steps * 1e3}
        h_fast, _ = distributed.generate_ir(ctx, *trace_args, **ir_kwargs(None, capi.IR_FAST))
        h_exact, _ = distributed.generate_ir(ctx, *trace_args, **ir_kwargs(None, capi.IR_EXACT))
        torch.cuda.synchronize()
        if rank == 0:
            comparison = fast_vs_exact(h_fast, h_exact)
        del h_fast, h_exact

    python_leg = None
    if not args.no_extras and not args.native and world == 1 and not grouped:
        # the same region driven through the C-ABI's own pipeline (what a C++ caller gets), its last histogram held against a solo IR
        n_elapsed, n_hist, n_info = timed_native(args.steps, 4, mode)
        check, _ = distributed.generate_ir(ctx, *trace_args, **dict(ir_kwargs(None), host_out=None))
        fence()
        native_leg = {"what": "rvb_pipeline_* (csrc/pipeline.hip): the schedule behind the C-ABI, driven through ctypes", "ms_per_step": n_elapsed / args.steps * 1e3,
                      "value": bounces_per_step / (n_elapsed / args.steps), "last_histogram_equals_solo_ir": bool(torch.equal(n_hist, check.cpu()))}
        del check, n_hist
    if not args.no_extras and args.native and world == 1 and not grouped:
        # ... and the Python driver (distributed.IrPipeline: what N > 1 ranks run, with the collectives) beside the native one
        pipeline.run(4, trace_args, ir_kwargs(None), keep)
        p_elapsed = timed(args.steps, ir_kwargs(None))
        python_leg = {"what": "distributed.IrPipeline (Python over the C-ABI), same schedule", "ms_per_step": p_elapsed / args.steps * 1e3,
                      "value": bounces_per_step / (p_elapsed / args.steps)}
    attenuate_probe, api_flow = None, None
    if rank == 0 and not args.no_extras:
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
        if world == 1:
            # the same workload through the reference's own API (cmd/main.cpp:241-298): Raytracer::raytrace -> getAllRaw ->
            # SpeakerAttenuator::attenuate -> fixPredelay -> flattenImpulses, every stage's result handed over as std::vector
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import api_flow_run
                api_flow = api_flow_run.run(nrays, nrefl, args.triangles, repeats=3, env={"RVB_DEVICE": str(local_rank)})
                api_flow["what"] = ("tools/api_flow.cpp: the reference's call sequence through the C++ mirror of its classes, best of 3; "
                                    "the PCIe floor of that API is 3 x 0.82 GB of result vectors")
            except Exception as e:      # the leg is a report, not the metric
                api_flow = {"error": str(e)[:300]}
    if rank == 0:
        avg = {k: float(np.mean(v)) for k, v in kernel_ms.items()}
        solo = {k: float(np.mean(v)) for k, v in solo_ms.items()}       # one IR alone on the GPU (untimed pass above)
        # (the library names a timing after the kernel that ran: path_kernel / path_pair_kernel, shadow_kernel / shadow_pair_kernel)
        trace_ms = (solo.get("path_kernel", solo.get("path_pair_kernel", 0.0))          # (one IR alone: the path kernel of the solo pass)
                    + sum(solo.get(k, 0.0) for k in ("image_kernel", "shadow_kernel", "shadow_pair_kernel")))
        in_timed_region = {k: v for k, v in solo.items() if k in kernel_ms}     # (path_pair_kernel, not path_kernel, when traces run in pairs)
        dominant = max(in_timed_region or solo, key=solo.get)           # by the time the kernel itself needs
        # Kernel durations for the rooflines come from the solo pass (HIP events around each launch with one IR on the GPU): in the
        # timed region the kernels of several IRs interleave and the events around a launch then span its neighbour's work as well.
        # rocprofv3 --kernel-trace --stats of `bench.py --contexts 1` agrees with them (profiles/).
        pmc = {}
        try:
            loaded = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
            if (nrays, nrefl, args.triangles, world) == (100000, 128, 75000, 1):
                pmc = loaded["kernels"]
        except (OSError, ValueError, KeyError):
            pass
        from_profile = "profiles/%s (rocprofv3 --pmc passes of this command, per launch)" % PMC_FILE

        def hbm_traffic(k):
            # calibrated: FETCH_SIZE x 2 for streaming kernels, x 1 for gathers (tools/fetch_calibration.hip, profiles/r04_fetch_calibration_n1.txt)
            v = pmc.get(k, {})
            return v.get("hbm_bytes_calibrated_per_ir", v.get("hbm_bytes_calibrated_per_launch", v.get("hbm_bytes_per_ir", v.get("hbm_bytes_per_launch"))))

        # the contract's roofline line, for the dominant kernel: the trace stage's algorithmic bytes (69.75 B per
        # ray-bounce, ALL of them charged to this one kernel) over its launch duration, against HBM.  A ray's bounces are a dependent
        # chain over a cache-resident scene, so this fraction is small by construction; what bounds the kernel is VALU issue
        # (roofline_valu), and the HBM-bound kernels of the path are in roofline_stream.
        ach = BYTES_PER_BOUNCE * nrays * nrefl / (solo[dominant] * 1e-3) / 1e9
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": hbm_traffic(dominant),
                    "traffic_source": from_profile + ", (FETCH_SIZE + WRITE_SIZE) x 1024: a gather kernel, whose FETCH_SIZE is calibrated at x 1 "
                                      "(profiles/r04_fetch_calibration_n1.txt; the guide's x 2 holds for wide streaming reads)" if hbm_traffic(dominant) else None,
                    "algorithmic_bytes_per_launch": BYTES_PER_BOUNCE * nrays * nrefl, "avg_launch_ms": solo[dominant],
                    "elapsed_ms_in_timed_region": avg.get(dominant),
                    "note": "avg_launch_ms: HIP events around the launch on the context's stream with one IR on the GPU; the whole trace "
                            "stage's bytes are charged to this kernel; it is VALU-issue-bound (roofline_valu)"}
        # VALU issue model: the kernel's dynamic instruction mix (per-class PMC counters) x the issue cost of each class measured on
        # this chip at 8 waves per SIMD (tools/inst_probe.hip -> profiles/r02_inst_probe.log) = the time the launch needs if every
        # SIMD issued VALU work back to back; frac = that time / the measured launch time.  frac_of_fma_peak prices every
        # instruction at the guide's v_fma_f32 rate instead (2 cycles per wave64 instruction).  useful_lane_fraction = active lanes
        # per issued VALU instruction / 64 (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)): masked-off lanes of the quad
        # kernels (a wave's 16 rays are not all in the step kind being executed) are issued but do nothing.
        valu = {}
        for k in ("path_kernel", "path_pair_kernel", "shadow_kernel", "shadow_pair_kernel"):
            v = pmc.get(k, {})
            if "SQ_INSTS_VALU" in v and solo.get(k):
                a = v["SQ_INSTS_VALU"] / (solo[k] * 1e-3) / 1e9
                # frac: against the guide's vector issue rate (every instruction priced as a v_fma_f32: 2 cycles per wave64 instruction per
                # SIMD); frac_of_issue_model: against the time this kernel's own instruction mix needs at the issue costs measured per class
                entry = {"bound": "valu_issue", "achieved": a, "unit": "G wave-instructions/s", "avg_launch_ms": solo[k],
                         "valu_instructions_per_launch": v["SQ_INSTS_VALU"], "numerators": from_profile,
                         "peak": VALU_FMA_PEAK_GINST, "frac": a / VALU_FMA_PEAK_GINST, "frac_of_fma_peak": a / VALU_FMA_PEAK_GINST}
                if v.get("valu_issue_model_ms"):
                    entry.update(frac_of_issue_model=v["valu_issue_model_ms"] / solo[k],
                                 issue_model_ms=v["valu_issue_model_ms"], instruction_mix=v.get("valu_mix"))
                if v.get("valu_lane_utilisation") is not None:
                    entry["useful_lane_fraction"] = v["valu_lane_utilisation"]
                valu[k] = entry
        # the whole step in the timed region: issue-model time of every kernel of one IR (per launch x launches per IR, the path kernel
        # being the one the timed region ran) over the measured time per IR = the share of the time the SIMDs issue VALU work
        if pmc and args.mode == "exact":
            irs = max(1, int(loaded.get("irs_in_command") or 0))
            ran_pairs = "path_pair_kernel" in avg
            per_ir, parts = 0.0, {}
            for k, v in pmc.items():
                if not v.get("valu_issue_model_ms") or k == "attenuate_kernel":
                    continue
                if k == ("path_kernel" if ran_pairs else "path_pair_kernel"):
                    continue
                launches_per_ir = 1.0 if k in ("path_kernel", "path_pair_kernel") else v.get("_launches", irs) / irs
                parts[k] = v["valu_issue_model_ms"] * launches_per_ir
                per_ir += parts[k]
            valu["whole_step"] = {"bound": "valu_issue", "issue_model_ms_per_ir": per_ir, "ms_per_ir_timed_region": ms_per_step,
                                  "frac": per_ir / ms_per_step, "issue_model_ms_by_kernel": parts, "numerators": from_profile,
                                  "note": "sum over the kernels of one IR of (VALU issue-model time per launch x launches per IR) / measured time per IR",
                                  # the issue costs are cycles at 2.4 GHz; the chip holds 2.32 GHz while this pipeline runs (shader cycles over 100-MHz
                                  # ticks around the path kernel's loop: tools/pair_stamps.py pipeline, profiles/r04c_pair_stamps_n1.txt)
                                  "held_clock_ghz": HELD_CLOCK_GHZ, "frac_at_held_clock": per_ir / ms_per_step * 2.4 / HELD_CLOCK_GHZ}
        stream = {}
        algorithmic = {"shadow_kernel": 128.0 * nrays * nrefl,          # 64-byte work record read, 64-byte Impulse written
                       "shadow_pair_kernel": 128.0 * nrays * nrefl,
                       "time_range_kernel": 64.0 * nrays * nrefl, "histogram_fast_kernel": 64.0 * nrays * nrefl,
                       "exact_mode": (64.0 + 8.0 + 64.0) * nrays * nrefl}   # keys pass, (key, index) pairs out, gather of the records
        for k in ("shadow_kernel", "shadow_pair_kernel", "time_range_kernel", "histogram_fast_kernel", "exact_mode"):
            if k in solo and solo[k] > 0:                # against the kernel's duration with the GPU to itself
                a = algorithmic[k] / (solo[k] * 1e-3) / 1e9
                stream[k] = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                             "avg_launch_ms": solo[k], "traffic": hbm_traffic(k)}
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
                                   "2 cardioid speakers, %.0f Hz, trim_predelay, output_mode all, histogram mode %s%s"
                                   % (scene[0].shape[0], rays_per_gpu, nrefl, sr, args.mode,
                                      "" if world == 1 else " (BASELINE configs[2]'s shape: %d rays over %d GPUs)" % (rays_per_gpu * world, world)),
                       "triangles": int(scene[0].shape[0]), "rays_per_gpu": rays_per_gpu, "reflections": nrefl,
                       "histogram_mode": args.mode + ((": the reference's serial float sum over all impulses, bit for bit" if world == 1 or args.exact_chain else
                                                       ": every rank's histogram is the serial float sum over ITS impulses; the ranks' histograms are then "
                                                       "added by one all-reduce, so the result is within 1e-5 of each band's maximum of the single-GPU "
                                                       "serial sum, not bit-equal to it (the bit-equal chain over devices is rvb_multi_*, csrc/multi.hip)")
                                                      if args.mode == "exact" else ": float atomics, order-dependent in the last bits"),
                       "histogram_seconds": state["nbins"] / sr,
                       "sharding": "ray-range shards, replicated scene, one all-reduce(sum) of [2][8][nbins] histograms",
                       "pipelining": "%d contexts per GPU take turns: traces are enqueued %d at a time, the next group before the current one is finished; "
                                     "the timed impulse responses are IDENTICAL jobs (same microphone, source and rays every step; jobs that differ — "
                                     "the 64 pairs of config C5 — run in tests/test_gpu_decomposition.py and tests/cpp/test_pipeline.cpp)"
                                     % (len(contexts), max(1, len(contexts) // 2))},
            "lanes_per_ray": {"path_kernel_in_timed_region": 2 if "path_pair_kernel" in avg else 4,
                              "path_kernel_one_ir_alone": 4 if "path_kernel" in solo else 2, "shadow_kernel": 2 if "shadow_pair_kernel" in solo else 4,
                              "rule": "rvb_path_lanes_for (csrc/trace_kernels.hip): two lanes per ray when rays per launch x traces in flight >= 196 608"},
            # one IR strictly alone, wall clock incl. the Python host path: until [channels][8][nbins] is on the host /
            # with the histogram left in HBM
            "ir_gen_wall_ms": to_host_latency_ms, "ir_gen_to_host_ms": to_host_latency_ms, "ir_gen_wall_ms_histogram_in_hbm": solo_latency_ms,
            "result_on_host_in_timed_region": to_host, "timed_region_check": timed_check, "contexts_per_gpu": len(contexts),
            "fast_mode" if args.mode == "exact" else "exact_mode": other_mode, "fast_vs_exact": comparison, "api_flow": api_flow,
            "driver": ("rvb_pipeline_* (csrc/pipeline.hip, the C-ABI's own pipeline, through ctypes)" if args.native and world == 1 and not grouped
                       else "distributed.IrPipeline (Python over the C-ABI; the N > 1 form with torch.distributed collectives)"),
            "native_pipeline": native_leg, "python_pipeline": python_leg,
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
