"""Multi-GPU impulse-response generation: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by rays: the scene is replicated, rank r traces the contiguous ray
range [r*n, (r+1)*n) of one seeded global ray set, and the only data-path collective is ONE
all-reduce(sum) of the [channels][8][nbins] histograms.  ONE small control collective makes the shards
agree on the binning: an all-gather of a few-KB block per rank holding the shard's diffuse time range
— the inputs of findPredelay (reference rayverb.h:49-74) and MAX_SAMPLE (rayverb.cpp:57) — and its
few valid image-source candidates.  Every rank merges the candidates itself with the reference's
"lowest ray index wins" rule (rayverb.cpp:654-676), so all ranks know the global time range without a
second exchange; rank 0 alone adds the merged image impulses to its histogram.

`tracer` is a capi.Context (GPU) or any object with the same methods (the CPU tests drive this module
with an oracle-backed stand-in).
"""
import numpy as np

from . import capi
from .dtypes import IMPULSE


def shard_range(total_rays, rank, world):
    """Contiguous ray range of `rank`; ranges differ by at most one ray."""
    base, extra = divmod(int(total_rays), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


EXCHANGE_CAPACITY = 64          # image-source candidates per rank carried by the first (normally only) exchange
_HEADER_BYTES = 16              # u64 candidate count, f32 min non-zero time, f32 max time


def _pack_block(candidates, lo, hi, capacity):
    block = np.zeros(_HEADER_BYTES + capacity * capi.IMAGE_CANDIDATE.itemsize, dtype=np.uint8)
    block[:8] = np.frombuffer(np.uint64(candidates.shape[0]).tobytes(), dtype=np.uint8)
    block[8:16] = np.frombuffer(np.array([lo, hi], dtype=np.float32).tobytes(), dtype=np.uint8)
    n = min(int(candidates.shape[0]), capacity)
    if n:
        block[_HEADER_BYTES:_HEADER_BYTES + n * capi.IMAGE_CANDIDATE.itemsize] = \
            np.frombuffer(np.ascontiguousarray(candidates[:n]).tobytes(), dtype=np.uint8)
    return block


def _unpack_block(block, capacity):
    count = int(np.frombuffer(block[:8].tobytes(), dtype=np.uint64)[0])
    lo, hi = (float(x) for x in np.frombuffer(block[8:16].tobytes(), dtype=np.float32))
    n = min(count, capacity)
    cand = np.frombuffer(block[_HEADER_BYTES:_HEADER_BYTES + n * capi.IMAGE_CANDIDATE.itemsize].tobytes(),
                         dtype=capi.IMAGE_CANDIDATE).copy()
    return count, lo, hi, cand


def exchange_shard_summaries(candidates, lo, hi, world, device, capacity=None):
    """The ONE control collective of a multi-GPU impulse response: an all-gather of a small fixed-size
    block per rank = (number of valid image-source candidates, min non-zero / max diffuse time of the
    shard, the candidates themselves).  Returns (all candidates in rank order, [(lo, hi)] per rank).
    A shard with more than `capacity` candidates (rare: a few per 100k rays are typical) triggers one
    more all-gather sized for the largest shard; every rank sees the same counts, so all ranks agree
    on whether that second round happens."""
    import torch
    import torch.distributed as dist
    capacity = EXCHANGE_CAPACITY if capacity is None else capacity
    while True:
        mine = torch.from_numpy(_pack_block(candidates, lo, hi, capacity)).to(device)
        blocks = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(blocks, mine)
        parts = [_unpack_block(b.cpu().numpy(), capacity) for b in blocks]
        worst = max(p[0] for p in parts)
        if worst <= capacity:
            return np.concatenate([p[3] for p in parts]), [(p[1], p[2]) for p in parts]
        capacity = worst


def combine_time_ranges(ranges):
    """(min non-zero, max) over (lo, hi) pairs in which lo == 0 means "no non-zero time"."""
    los = [lo for lo, _ in ranges if lo > 0]
    return (min(los) if los else 0.0), max([hi for _, hi in ranges] + [0.0])


class SpeakerModel:
    """Microphone model of reference kernel `attenuate` (cardioid-family speakers, config.h AttenuationModel::SPEAKER)."""

    def __init__(self, directions, coefficients):
        self.directions, self.coefficients = directions, coefficients
        self.nchannels = len(coefficients)

    def configure(self, tracer, mic, which, images):
        tracer.ir_configure_speakers(mic, self.directions, self.coefficients, which, images)


class HrtfModel:
    """Listener model of reference kernel `hrtf` (AttenuationModel::HRTF): table [2][360][180][8], facing and up vectors."""

    def __init__(self, table, facing, up):
        self.table, self.facing, self.up = table, facing, up
        self.nchannels = 2

    def configure(self, tracer, mic, which, images):
        tracer.ir_configure_hrtf(mic, self.table, self.facing, self.up, which, images)


def _p2p_staged(t):
    """gloo carries point-to-point messages for CPU tensors only: a CUDA block goes through host memory there (RCCL takes it as it is)."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo"


def _send_block(view, dst):
    import torch.distributed as dist
    block = view.contiguous()
    dist.send(block.cpu() if _p2p_staged(block) else block, dst=dst)


def _recv_block(view, src):
    import torch
    import torch.distributed as dist
    staged = _p2p_staged(view)
    block = torch.empty(view.shape, dtype=view.dtype, device="cpu" if staged else view.device)
    dist.recv(block, src=src)
    return block.to(view.device) if staged else block


def begin_ir(tracer, mic, source, nreflections, air, ray_offset=0):
    """First half of generate_ir: enqueues the trace of the rays already set on `tracer` (asynchronous on a GPU context)."""
    tracer.trace(mic, source, nreflections, air, ray_offset=ray_offset)


def generate_ir(tracer, mic, source, nreflections, air, speakers_dir=None, speakers_coeff=None, sample_rate=44100.0,
                trim_predelay=True, mode=capi.IR_FAST, rank=0, world=1, ray_offset=0, device="cpu",
                which=capi.IR_ALL, remove_direct=False, on_stage=None, begun=False, model=None, collectives=None, defer=False,
                host_out=None, chain_exact=False, chain_blocks=8):
    """One impulse response from the rays already set on `tracer`.  Returns (hist tensor
    [nchannels][8][nbins] — identical on every rank —, info dict).

    world == 1: trace -> merge image sources -> time range -> bin.
    world > 1: two collectives in all.  (1) exchange_shard_summaries: every rank learns every shard's
    image-source candidates and diffuse time range, merges the candidates itself (deterministic, a few
    dozen records) and so knows the global predelay / length without a further exchange; (2) the
    all-reduce(sum) of the histograms.  Only rank 0 adds the merged image impulses to its histogram.

    begun=True: begin_ir(tracer, ...) has already enqueued this IR's trace.  A caller with two contexts per GPU
    calls begin_ir on the second before generate_ir(..., begun=True) on the first, so that one IR's trace (VALU-bound)
    runs beside the other's record grouping, binning, host work and collectives (IrPipeline below).

    chain_exact=True (several ranks, exact mode): instead of all-reducing the ranks' serial sums — which is the single-GPU histogram
    only up to float re-association — the ranks continue ONE serial sum in ray order: rank r receives the histogram from rank r-1,
    folds its own diffuse impulses on top (rvb_ir_accumulate in exact mode adds to what the histogram holds), hands it to rank r+1;
    the last rank adds the merged image sources (the reference's order: diffuse, then images — rayverb.cpp:708-714) and broadcasts.
    Bit-identical to one GPU and to flattenImpulses.  The traces, and every rank's keying and sorting, run side by side; only the
    fold is a chain, and the histogram travels through it in `chain_blocks` bin-range blocks (rank r folds block k while rank r + 1
    folds block k - 1: (world + blocks - 1) / blocks folds and hops instead of world).

    host_out(shape) -> pinned host tensor: the finished histogram is also copied there (info["host"]) — enqueued behind the binning
    (behind the all-reduce with collectives) on the tracer's export stream, so neither the host nor the tracer's next trace waits for
    the link; info["host"] is complete after tracer.synchronize_exports() (or a device-wide synchronisation), and the caller keeps
    `hist` alive until then.

    defer=True: returns (hist, info, finish) as soon as the binning is ENQUEUED; finish() waits for it (and runs the all-reduce).
    A caller that finishes several IRs enqueues all their binning stages first, so that they run side by side instead of each
    waiting for the one before (IrPipeline: the stages of a group)."""
    import torch
    import torch.distributed as dist

    empty = np.zeros(0, dtype=IMPULSE)
    if collectives is None:                              # True with world == 1 rehearses the multi-rank code path on one rank
        collectives = world > 1
    if model is None:                                    # the two-list form: speakers (what bench.py and the reference demo configs use)
        model = SpeakerModel(speakers_dir, speakers_coeff)
    if not begun:
        begin_ir(tracer, mic, source, nreflections, air, ray_offset)
    candidates = tracer.get_image_candidates()           # small: valid image-source paths only
    if on_stage:
        on_stage("trace", tracer)
    direct = tracer.get_direct()
    want_images = bool(which & capi.IR_IMAGES)
    if not collectives:
        images = capi.merge_images(candidates, direct, remove_direct) if want_images else empty
        model.configure(tracer, mic, which, images)
        lo, hi = tracer.ir_time_range()
        if on_stage:
            on_stage("time_range", tracer)
        contributes = True
    else:
        ranges = []
        lo_d = hi_d = 0.0
        if which & capi.IR_DIFFUSE:                      # this shard's diffuse impulses alone
            model.configure(tracer, mic, capi.IR_DIFFUSE, empty)
            lo_d, hi_d = tracer.ir_time_range()
            if on_stage:
                on_stage("time_range", tracer)
        candidates, shard_ranges = exchange_shard_summaries(candidates, lo_d, hi_d, world, device)
        ranges += shard_ranges
        images = capi.merge_images(candidates, direct, remove_direct) if want_images else empty
        if images.shape[0]:                              # the merged images' own range, the same on every rank
            model.configure(tracer, mic, capi.IR_IMAGES, images)
            ranges.append(tracer.ir_time_range())
        lo, hi = combine_time_ranges(ranges)
        chain = bool(chain_exact) and mode == capi.IR_EXACT and world > 1
        adds_images = rank == (world - 1 if chain else 0)      # one rank adds the merged image impulses: the last of a chain, else rank 0
        mine = which if adds_images else (which & capi.IR_DIFFUSE)
        contributes = mine != 0
        if contributes and not chain:
            model.configure(tracer, mic, mine, images if adds_images else empty)
        if not adds_images and not chain:
            images = empty
    predelay = lo if trim_predelay else 0.0
    nbins = tracer.ir_bins(hi, predelay, sample_rate)
    hist = torch.zeros((model.nchannels, 8, nbins), device=device, dtype=torch.float32)
    chain = collectives and bool(chain_exact) and mode == capi.IR_EXACT and world > 1
    info = {"nbins": nbins, "predelay": predelay, "images": int(images.shape[0]), "max_time": hi}
    host = host_out(tuple(hist.shape)) if host_out is not None else None
    # one call for the binning AND the histogram's way to the host where the tracer offers it (rvb_ir_accumulate_export: in exact
    # mode the bin ranges leave as they become final); otherwise the binning, then the copy behind it in stream order
    fused_export = host is not None and not collectives and contributes and hasattr(tracer, "ir_accumulate_export_tensor")
    if contributes and not chain:                        # (the tracer's stream waits for torch's zero fill by an event)
        if fused_export:
            tracer.ir_accumulate_export_tensor(predelay, sample_rate, nbins, mode, hist, host)
        else:
            tracer.ir_accumulate_tensor(predelay, sample_rate, nbins, mode, hist)
    if host is not None:
        info["host"] = host
        if not collectives and hasattr(tracer, "export_tensor_to_host") and not fused_export:
            if not contributes:                          # (nothing was enqueued on the tracer's stream: order the copy behind torch's zero fill)
                tracer.ir_accumulate_wait_for_torch()
            tracer.export_tensor_to_host(hist, host)     # stream order: behind the binning

    def finish():
        if on_stage:
            on_stage("accumulate", tracer)
        if chain:
            # one serial sum over all ranks, in ray order (see the docstring) — SYSTOLIC: the histogram travels in bin-range blocks, rank r
            # folds block k while rank r + 1 folds block k - 1; a rank keys and sorts its impulses (ir_exact_prepare) before the first
            # block arrives.  Tracers without the two-step form fold the whole histogram as one block.
            two_step = hasattr(tracer, "ir_exact_prepare")
            blocks = max(1, min(int(chain_blocks), (nbins + 15) // 16)) if two_step else 1
            per = ((nbins + blocks - 1) // blocks + 15) & ~15
            folds = bool(which & capi.IR_DIFFUSE)
            if folds:
                model.configure(tracer, mic, capi.IR_DIFFUSE, empty)
                if two_step:
                    tracer.ir_exact_prepare(predelay, sample_rate, nbins)
            for b0 in range(0, nbins, per):
                b1 = min(nbins, b0 + per)
                if rank > 0:
                    hist[:, :, b0:b1] = _recv_block(hist[:, :, b0:b1], rank - 1)
                if folds:
                    if two_step:
                        tracer.ir_exact_fold_tensor(nbins, b0, b1, hist)
                    else:
                        tracer.ir_accumulate_tensor(predelay, sample_rate, nbins, mode, hist)
                    tracer.synchronize()
                if rank < world - 1:
                    _send_block(hist[:, :, b0:b1], rank + 1)
            if rank == world - 1 and (which & capi.IR_IMAGES) and images.shape[0]:
                model.configure(tracer, mic, capi.IR_IMAGES, images)
                tracer.ir_accumulate_tensor(predelay, sample_rate, nbins, mode, hist)
                tracer.synchronize()
            dist.broadcast(hist, src=world - 1)
        else:
            tracer.synchronize()
            if collectives:
                dist.all_reduce(hist, op=dist.ReduceOp.SUM)  # RCCL over xGMI: [channels][8][nbins] floats
        if host is not None and (collectives or not hasattr(tracer, "export_tensor_to_host")):
            if hist.is_cuda and hasattr(tracer, "export_tensor_to_host"):
                tracer.ir_accumulate_wait_for_torch()    # the collective ran on torch's stream: the tracer's stream waits for it by an event
                tracer.export_tensor_to_host(hist, host)
            else:
                host.copy_(hist)

    if defer:
        return hist, info, finish
    finish()
    return hist, info


class IrPipeline:
    """Impulse responses back to back with TWO contexts per GPU: the trace of IR i+1 is enqueued on the other context
    before IR i is finished, so its VALU-bound path kernel runs beside IR i's bandwidth-, atomic- and host-bound stages
    (record grouping, binning, candidate merge, collectives).  Single host thread, collectives in program order: safe
    with any world size.  Measured on one MI355X at workload C2: 6.45 -> 5.6 ms per IR."""

    def __init__(self, tracers):
        assert len(tracers) >= 1
        self.tracers = list(tracers)
        self.next_slot = 0
        self.begun = [False] * len(self.tracers)
        self.fuse_groups = __import__("os").environ.get("RVB_PIPELINE_FUSE", "1") != "0"       # one path-kernel launch per group
        # the traces of a group run side by side (run_jobs): tell the contexts, so that the path kernel is sized for the rays in flight
        for t in self.tracers:
            if hasattr(t, "set_concurrent_traces"):
                t.set_concurrent_traces(self.group_size(len(self.tracers)))

    @staticmethod
    def group_size(n):
        group = max(1, n // 2) if n > 1 else 1
        return min(group, int(__import__("os").environ.get("RVB_PIPELINE_GROUP", group)))       # (an override may only shrink it)

    def run(self, count, trace_args, ir_kwargs, on_result=None):
        """`count` IRs with the same arguments (bench) — trace_args = (mic, source, nreflections, air), ir_kwargs as
        generate_ir takes them.  on_result(hist, info, tracer) is called per IR in order."""
        self.run_jobs([(trace_args, ir_kwargs)] * count, on_result)

    def run_jobs(self, jobs, on_result=None):
        """One IR per job = (trace_args, ir_kwargs), e.g. the source / listener pairs of a hall (BASELINE config C5): every
        context must hold the same scene and rays.  Results arrive in job order."""
        n = len(self.tracers)
        if not jobs:
            return
        first = self.next_slot

        def begin(k):
            slot = (first + k) % n
            trace_args, ir_kwargs = jobs[k]
            begin_ir(self.tracers[slot], *trace_args, ray_offset=ir_kwargs.get("ray_offset", 0))
            self.begun[slot] = True

        def begin_group(ks):
            """The traces of a group in ONE path-kernel launch (Context.trace_group) when the contexts offer it and the jobs agree in
            reflection count and air coefficients; one after the other otherwise."""
            tracers = [self.tracers[(first + k) % n] for k in ks]
            same = all(jobs[k][0][2] == jobs[ks[0]][0][2] and tuple(jobs[k][0][3]) == tuple(jobs[ks[0]][0][3]) for k in ks)
            if len(ks) > 1 and same and all(hasattr(t, "trace_group") for t in tracers) and self.fuse_groups:
                tracers[0].trace_group(tracers, [jobs[k][0][0] for k in ks], [jobs[k][0][1] for k in ks], jobs[ks[0]][0][2], jobs[ks[0]][0][3],
                                       [jobs[k][1].get("ray_offset", 0) for k in ks])
                for k in ks:
                    self.begun[(first + k) % n] = True
            else:
                for k in ks:
                    begin(k)

        # Default schedule: jobs are enqueued in GROUPS of `group` traces (their path kernels then run side by side: more waves
        # per SIMD, see DESIGN.md "rays per launch"), and group j+1 is enqueued before group j is finished.  group =
        # len(tracers) // 2; two contexts: group 1 = the plain alternation.  RVB_PIPELINE_AHEAD=k instead keeps k traces
        # enqueued ahead of the IR being finished (k < len(tracers)).
        group = self.group_size(n)
        ahead = min(n - 1, int(__import__("os").environ.get("RVB_PIPELINE_AHEAD", 0)))
        begun_upto = [0]                                     # jobs [0, begun_upto) have been begun

        def begin_upto(k):
            while begun_upto[0] < min(k, len(jobs)):
                if n > 1 and ahead <= 0 and group > 1:               # (group boundaries: begun_upto is a multiple of group here)
                    ks = list(range(begun_upto[0], min(begun_upto[0] + group, k, len(jobs))))
                    begin_group(ks)
                    begun_upto[0] += len(ks)
                else:
                    begin(begun_upto[0])
                    begun_upto[0] += 1

        begin_upto(group if ahead <= 0 else 1 + ahead)
        if n > 1 and ahead <= 0:
            # group by group: the traces of the group after next are enqueued, then the binning stages of ALL IRs of this group
            # (each needs its own trace's image-source candidates on the host first), and only then the host waits for them — the
            # stages of a group run side by side, beside the next group's path kernels, instead of one after the other
            in_flight = max(2, n // group)                   # groups whose traces are enqueued at a time (each context holds one IR)
            for g0 in range(0, len(jobs), group):
                begin_upto((g0 // group + in_flight) * group)
                pending = []
                for i in range(g0, min(g0 + group, len(jobs))):
                    slot = (first + i) % n
                    trace_args, ir_kwargs = jobs[i]
                    began = self.begun[slot]
                    self.begun[slot] = False
                    hist, info, finish = generate_ir(self.tracers[slot], *trace_args, begun=began, defer=True, **ir_kwargs)
                    pending.append((hist, info, finish, self.tracers[slot]))
                for hist, info, finish, tracer in pending:
                    finish()
                    if on_result:
                        on_result(hist, info, tracer)
            self.next_slot = (first + len(jobs)) % n
            return
        for i, (trace_args, ir_kwargs) in enumerate(jobs):
            slot = (first + i) % n
            if n > 1:
                begin_upto(i + 1 + ahead)
            else:
                begin_upto(i + 1)
            began = self.begun[slot]
            self.begun[slot] = False
            hist, info = generate_ir(self.tracers[slot], *trace_args, begun=began, **ir_kwargs)
            if on_result:
                on_result(hist, info, self.tracers[slot])
        self.next_slot = (first + len(jobs)) % n


def generate_pair_irs(tracers, pairs, nreflections, air, model_for_pair, sample_rate, rank=0, world=1, device="cpu",
                      trim_predelay=True, mode=capi.IR_FAST, which=capi.IR_ALL, remove_direct=False, pairs_per_launch=1):
    """BASELINE config C5: many (source, listener) pairs in one scene.  The PAIRS shard over the ranks (contiguous blocks,
    shard_range) and every rank traces its own pairs with all of its rays — no collective at all.
    `pairs` = [(mic, source)], `model_for_pair(i)` -> SpeakerModel / HrtfModel of pair i; `tracers` = this rank's contexts
    (same scene and rays on each).  Returns {pair index: (hist, info)} for this rank's pairs.

    pairs_per_launch == 1: the pairs are jobs of the two-context pipeline, one trace each.
    pairs_per_launch > 1: that many pairs are traced in ONE launch (Context.trace_pairs: the path kernel runs with more waves per
    SIMD), the next launch is enqueued on the other context before this launch's pairs are binned."""
    import torch
    first, count = shard_range(len(pairs), rank, world)
    out = {}
    if pairs_per_launch <= 1:
        jobs = []
        for i in range(first, first + count):
            mic, source = pairs[i]
            jobs.append(((mic, source, nreflections, air),
                         dict(model=model_for_pair(i), sample_rate=sample_rate, trim_predelay=trim_predelay, mode=mode, device=device,
                              which=which, remove_direct=remove_direct)))    # rank 0 / world 1 inside generate_ir: a pair is not sharded
        order = list(range(first, first + count))
        IrPipeline(tracers).run_jobs(jobs, lambda hist, info, _tracer: out.__setitem__(order[len(out)], (hist, info)))
        return out

    batches = [list(range(b, min(first + count, b + pairs_per_launch))) for b in range(first, first + count, pairs_per_launch)]
    n = len(tracers)

    def begin(j):
        tracers[j % n].trace_pairs([pairs[i][0] for i in batches[j]], [pairs[i][1] for i in batches[j]], nreflections, air)

    if batches:
        begin(0)
    empty = np.zeros(0, dtype=IMPULSE)
    for j, batch in enumerate(batches):
        if j + 1 < len(batches) and n > 1:
            begin(j + 1)
        tracer = tracers[j % n]
        candidates = tracer.get_image_candidates()           # all pairs of the launch, global ray numbers
        for k, i in enumerate(batch):
            tracer.select_pair(k)
            images = empty
            if which & capi.IR_IMAGES:
                images = capi.merge_images(tracer.get_pair_candidates(k, candidates), tracer.get_direct(), remove_direct)
            model = model_for_pair(i)
            model.configure(tracer, pairs[i][0], which, images)
            lo, hi = tracer.ir_time_range()
            predelay = lo if trim_predelay else 0.0
            nbins = tracer.ir_bins(hi, predelay, sample_rate)
            hist = torch.zeros((model.nchannels, 8, nbins), device=device, dtype=torch.float32)
            tracer.ir_accumulate_tensor(predelay, sample_rate, nbins, mode, hist)
            out[i] = (hist, {"nbins": nbins, "predelay": predelay, "images": int(images.shape[0]), "max_time": hi})
        if n == 1 and j + 1 < len(batches):
            tracer.synchronize()                             # the next launch reuses this context's buffers
            begin(j + 1)
        else:
            tracer.synchronize()
    return out
