"""On-disk formats around the hot path, Python side.

* `impulse.dump` — the reference's diagnostic file (reference rayverb/helpers.cpp:19-59): one line of
  JSON per ray, `[{"position":[x,y,z],"volume":mean band volume}, ...]` with one object per reflection;
  read by its Processing viewer (viewer/viewer.pde:55-70).
* `RVBHIST1` — the binary impulse-response dump of include/rayverb/ir_dump.h: histograms
  [channels][8][nbins] + merged image-source impulses, for offline comparison and for shipping a rank's
  partial result.
"""
import json
import struct

import numpy as np

from .dtypes import IMPULSE

MAGIC = b"RVBHIST1"


def write_impulse_dump(fname, impulses, nrays, nreflections):
    """`impulses`: IMPULSE array [nrays * nreflections] in ray-major order (getRawDiffuse)."""
    imp = np.asarray(impulses).reshape(nrays, nreflections)
    pos = imp["position"][..., :3].astype(np.float64)
    average = np.zeros((nrays, nreflections), dtype=np.float32)      # float sum in band order, then / 8
    for k in range(8):
        average = (average + imp["volume"][..., k]).astype(np.float32)
    average = (average / np.float32(8)).astype(np.float32).astype(np.float64)
    with open(fname, "w") as out:
        for i in range(nrays):
            out.write(json.dumps([{"position": [float(x) for x in pos[i, j]], "volume": float(average[i, j])}
                                  for j in range(nreflections)], separators=(",", ":")) + "\n")


def read_impulse_dump(fname):
    """-> (positions [nrays][nreflections][3] float64, mean volumes [nrays][nreflections] float64)"""
    pos, vol = [], []
    with open(fname) as f:
        for line in f:
            if not line.strip():
                continue
            ray = json.loads(line)
            pos.append([r["position"] for r in ray])
            vol.append([r["volume"] for r in ray])
    return np.asarray(pos, dtype=np.float64), np.asarray(vol, dtype=np.float64)


def write_ir_dump(fname, histogram, sample_rate, predelay, images=None):
    hist = np.ascontiguousarray(histogram, dtype="<f4")
    assert hist.ndim == 3 and hist.shape[1] == 8, "histogram must be [channels][8][nbins]"
    img = np.zeros(0, dtype=IMPULSE) if images is None else np.ascontiguousarray(images, dtype=IMPULSE)
    with open(fname, "wb") as out:
        out.write(MAGIC)
        out.write(struct.pack("<IIQffQ", hist.shape[0], 8, hist.shape[2], float(sample_rate), float(predelay), img.shape[0]))
        out.write(hist.tobytes())
        out.write(img.tobytes())


def read_ir_dump(fname):
    """-> dict(histogram [channels][8][nbins] float32, sample_rate, predelay, images IMPULSE[])"""
    with open(fname, "rb") as f:
        head = f.read(8 + struct.calcsize("<IIQffQ"))
        if len(head) < 40 or head[:8] != MAGIC:
            raise ValueError("%s is not an RVBHIST1 file" % fname)
        channels, bands, nbins, sample_rate, predelay, nimages = struct.unpack("<IIQffQ", head[8:])
        if bands != 8:
            raise ValueError("%s: %d bands (8 expected)" % (fname, bands))
        hist = np.frombuffer(f.read(channels * 8 * nbins * 4), dtype="<f4")
        img = np.frombuffer(f.read(nimages * IMPULSE.itemsize), dtype=IMPULSE)
        if hist.shape[0] != channels * 8 * nbins or img.shape[0] != nimages:
            raise ValueError("%s is truncated" % fname)
    return {"histogram": hist.reshape(channels, 8, nbins).copy(), "sample_rate": sample_rate, "predelay": predelay,
            "images": img.copy()}
