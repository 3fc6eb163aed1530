"""Multi-GPU impulse-response generation: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by rays: the scene is replicated, rank r traces the contiguous ray
range [r*n, (r+1)*n) of one seeded global ray set, and the only data-path collective is ONE
all-reduce(sum) of the [channels][8][nbins] histograms.  Two tiny control exchanges make the shards
agree on the binning: an all-reduce(max) of (-min_nonzero_time, max_time) — the inputs of findPredelay
(reference rayverb.h:49-74) and MAX_SAMPLE (rayverb.cpp:57) — and an all-gather of the few valid
image-source candidates, merged with the reference's "lowest ray index wins" rule (rayverb.cpp:654-676)
on rank 0, which alone adds the merged image impulses to its histogram.

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


def reduce_time_range(lo, hi, group_world, device):
    """Global (min non-zero, max) of the per-rank attenuated-time ranges; lo == 0 means "none"."""
    if group_world == 1:
        return lo, hi
    import torch
    import torch.distributed as dist
    sentinel = -3.0e38
    r = torch.tensor([-lo if lo > 0 else sentinel, hi], device=device, dtype=torch.float32)
    dist.all_reduce(r, op=dist.ReduceOp.MAX)
    r = r.cpu()
    return (float(-r[0]) if float(r[0]) > sentinel else 0.0), float(r[1])


def gather_candidates(candidates, group_world):
    if group_world == 1:
        return candidates
    import torch.distributed as dist
    gathered = [None] * group_world
    dist.all_gather_object(gathered, candidates)
    return np.concatenate(gathered)


def generate_ir(tracer, mic, source, nreflections, air, speakers_dir, speakers_coeff, sample_rate,
                trim_predelay=True, mode=capi.IR_FAST, rank=0, world=1, ray_offset=0, device="cpu",
                which=capi.IR_ALL, remove_direct=False, on_stage=None):
    """One impulse response from the rays already set on `tracer`.  Returns (hist tensor
    [nchannels][8][nbins] — identical on every rank —, info dict)."""
    import torch
    import torch.distributed as dist

    tracer.trace(mic, source, nreflections, air, ray_offset=ray_offset)
    candidates = tracer.get_image_candidates()           # small: valid image-source paths only
    if on_stage:
        on_stage("trace")
    direct = tracer.get_direct()
    candidates = gather_candidates(candidates, world)
    if rank == 0 and (which & capi.IR_IMAGES):
        images = capi.merge_images(candidates, direct, remove_direct)
    else:
        images = np.zeros(0, dtype=IMPULSE)
    tracer.ir_configure_speakers(mic, speakers_dir, speakers_coeff, which, images)
    lo, hi = tracer.ir_time_range()
    if on_stage:
        on_stage("time_range")
    lo, hi = reduce_time_range(lo, hi, world, device)
    predelay = lo if trim_predelay else 0.0
    nbins = tracer.ir_bins(hi, predelay, sample_rate)
    hist = torch.zeros((len(speakers_coeff), 8, nbins), device=device, dtype=torch.float32)
    if hist.is_cuda:
        torch.cuda.synchronize()                         # the zero fill ran on torch's stream
    tracer.ir_accumulate_tensor(predelay, sample_rate, nbins, mode, hist)
    if on_stage:
        on_stage("accumulate")
    tracer.synchronize()
    if world > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)      # RCCL over xGMI: [channels][8][nbins] floats
    return hist, {"nbins": nbins, "predelay": predelay, "images": int(images.shape[0]), "max_time": hi}
