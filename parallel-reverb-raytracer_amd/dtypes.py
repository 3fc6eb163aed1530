"""numpy views of the reference's data contracts (reference rayverb/clstructs.h:13-58).

Sizes/alignments as SURVEY.md §8(a) T1-T7: Triangle 32 B, Surface 64 B, Impulse 64 B,
AttenuatedImpulse 64 B, Speaker 32 B, cl_float3 16 B, VolumeType (cl_float8) 32 B.
"""
import numpy as np

NUM_IMAGE_SOURCE = 10        # reference rayverb/clstructs.h:4
SPEED_OF_SOUND = 340.0       # reference rayverb/clstructs.h:5
NUM_BANDS = 8

TRIANGLE = np.dtype([("surface", "<u8"), ("v0", "<u8"), ("v1", "<u8"), ("v2", "<u8")])
SURFACE = np.dtype([("specular", "<f4", (8,)), ("diffuse", "<f4", (8,))])
IMPULSE = np.dtype([("volume", "<f4", (8,)), ("position", "<f4", (4,)), ("time", "<f4"), ("pad", "<f4", (3,))])
ATTENUATED = np.dtype([("volume", "<f4", (8,)), ("time", "<f4"), ("pad", "<f4", (7,))])
SPEAKER = np.dtype([("direction", "<f4", (4,)), ("coefficient", "<f4"), ("pad", "<f4", (3,))])

assert TRIANGLE.itemsize == 32 and SURFACE.itemsize == 64
assert IMPULSE.itemsize == 64 and ATTENUATED.itemsize == 64 and SPEAKER.itemsize == 32

# Air absorption per band, reference rayverb/rayverb.cpp:632-641 (double product, then float).
AIR_COEFFICIENTS = np.array(
    [0.001 * -0.1, 0.001 * -0.2, 0.001 * -0.5, 0.001 * -1.1,
     0.001 * -2.7, 0.001 * -9.4, 0.001 * -29.0, 0.001 * -60.0], dtype=np.float64).astype(np.float32)


def aligned_zeros(n, dtype, align=64):
    """Zero-filled 1-D array of `dtype` whose base address is `align`-byte aligned
    (float8 members are accessed with 32-byte vector moves by host-compiled code)."""
    dtype = np.dtype(dtype)
    nbytes = int(n) * dtype.itemsize
    raw = np.zeros(nbytes + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + nbytes].view(dtype)


def aligned_copy(a, align=64):
    a = np.ascontiguousarray(a)
    out = aligned_zeros(a.size, a.dtype, align).reshape(a.shape)
    out[...] = a
    return out


def float3_array(xyz):
    """[n][3] -> [n][4] float32 (cl_float3 is a 16-byte cl_float4, 4th lane 0)."""
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = aligned_zeros(xyz.shape[0] * 4, np.float32).reshape(-1, 4)
    out[:, :3] = xyz
    return out
