"""ctypes binding of include/rvb_capi.h (librvb_hip.so) for tests and bench.py.

The method names follow the reference's host API for the path (reference rayverb/rayverb.h):
`Raytracer.raytrace / getRawDiffuse / getRawImages / getAllRaw`, `SpeakerAttenuator.attenuate`,
`HrtfAttenuator.attenuate`, `flattenImpulses`.  There is no CPU fallback: if the shared library is
missing or no gfx950 device is usable this module raises.
"""
import ctypes
import os

import numpy as np

from .dtypes import ATTENUATED, IMPULSE, SPEAKER, aligned_zeros

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RVB_LIB", os.path.join(_HERE, "librvb_hip.so"))   # RVB_LIB: A/B builds of the same ABI

IMAGE_CANDIDATE = np.dtype([("ray", "<u8"), ("slot", "<u4"), ("index", "<u4"), ("impulse", IMPULSE)])
assert IMAGE_CANDIDATE.itemsize == 80

IR_DIFFUSE, IR_IMAGES, IR_ALL = 1, 2, 3
IR_FAST, IR_EXACT = 0, 1

# every symbol include/rvb_capi.h declares
SYMBOLS = [
    "rvb_create", "rvb_destroy", "rvb_last_error", "rvb_synchronize", "rvb_wait_for_event", "rvb_device_info",
    "rvb_set_scene", "rvb_share_scene", "rvb_scene_info", "rvb_set_directions", "rvb_set_directions_device", "rvb_set_concurrent_traces", "rvb_set_path_lanes", "rvb_trace", "rvb_trace_group",
    "rvb_trace_pairs", "rvb_ir_select_pair",
    "rvb_get_diffuse", "rvb_diffuse_device", "rvb_get_direct", "rvb_get_image_candidates", "rvb_merge_images",
    "rvb_attenuate_speaker", "rvb_attenuate_speaker_device", "rvb_attenuate_hrtf", "rvb_attenuate_hrtf_device", "rvb_flatten",
    "rvb_ir_configure_speakers", "rvb_ir_configure_hrtf", "rvb_ir_time_range", "rvb_ir_time_range_begin", "rvb_ir_bins", "rvb_ir_accumulate", "rvb_ir_accumulate_export", "rvb_ir_exact_prepare", "rvb_ir_exact_fold", "rvb_record_event",
    "rvb_ir_download", "rvb_last_timings", "rvb_debug_stamps", "rvb_executed_bounces",
    "rvb_device_alloc", "rvb_device_free", "rvb_copy_to_host", "rvb_copy_to_device", "rvb_fix_predelay_device", "rvb_flatten_device",
    "rvb_host_alloc", "rvb_host_free", "rvb_copy_to_pinned_host_async", "rvb_synchronize_exports",
    "rvb_multi_create", "rvb_multi_destroy", "rvb_multi_last_error", "rvb_multi_devices", "rvb_multi_context", "rvb_multi_used_rccl", "rvb_multi_set_chain_blocks", "rvb_multi_peer_links",
    "rvb_multi_set_scene", "rvb_multi_set_directions", "rvb_multi_trace", "rvb_multi_get_diffuse", "rvb_multi_get_images",
    "rvb_multi_ir_speakers", "rvb_multi_ir_hrtf",
    "rvb_device_index", "rvb_pipeline_create", "rvb_pipeline_destroy", "rvb_pipeline_last_error", "rvb_pipeline_configure_speakers",
    "rvb_pipeline_configure_hrtf", "rvb_pipeline_submit", "rvb_pipeline_submit_oriented", "rvb_pipeline_pending", "rvb_pipeline_next",
]

_vp = ctypes.c_void_p
_u64 = ctypes.c_uint64
_lib = None


class RvbError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rvb error %d: %s" % (code, message))
        self.code = code


def load_library():
    """Loads librvb_hip.so (built by `make -C parallel-reverb-raytracer_amd`); raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not built: run __graft_entry__.build() (no CPU fallback exists)" % LIB_PATH)
        # torch brings its own copy of the HIP runtime; if this library's copy initialises the GPU first, torch.cuda later reports
        # "No HIP GPUs are available" (two runtimes in one process, load-order dependent).  Let torch go first when it is there:
        # device tensors handed over by pointer (bench.py, distributed.py) need it anyway.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        lib.rvb_last_error.restype = ctypes.c_char_p
        lib.rvb_last_error.argtypes = [_vp]
        lib.rvb_ir_bins.restype = _u64
        lib.rvb_ir_bins.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float]
        lib.rvb_destroy.restype = None
        lib.rvb_destroy.argtypes = [_vp]
        lib.rvb_multi_destroy.restype = None
        lib.rvb_multi_destroy.argtypes = [_vp]
        lib.rvb_multi_last_error.restype = ctypes.c_char_p
        lib.rvb_multi_last_error.argtypes = [_vp]
        lib.rvb_pipeline_last_error.restype = ctypes.c_char_p
        lib.rvb_pipeline_last_error.argtypes = [_vp]
        lib.rvb_pipeline_destroy.restype = None
        lib.rvb_pipeline_destroy.argtypes = [_vp]
        lib.rvb_pipeline_pending.restype = _u64
        lib.rvb_pipeline_pending.argtypes = [_vp]
        _lib = lib
    return _lib


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def _f8(v):
    return (ctypes.c_float * 8)(*[float(x) for x in v])


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def make_speakers(directions, coefficients):
    sp = aligned_zeros(len(coefficients), SPEAKER)
    sp["direction"][:, :3] = np.asarray(directions, np.float32).reshape(-1, 3)
    sp["coefficient"] = np.asarray(coefficients, np.float32)
    return sp


def merge_images(candidates, direct, remove_direct):
    """Host de-dup of image-source candidates (reference rayverb.cpp:654-676 + :692-706)."""
    lib = load_library()
    cand = np.ascontiguousarray(candidates, dtype=IMAGE_CANDIDATE)
    count = _u64(0)
    d = np.ascontiguousarray(direct, dtype=IMPULSE) if direct is not None else None
    rc = lib.rvb_merge_images(_ptr(cand), _u64(cand.shape[0]), _ptr(d), ctypes.c_int(int(remove_direct)), None, _u64(0),
                              ctypes.byref(count))
    if rc:
        raise RvbError(rc, "rvb_merge_images")
    out = np.zeros(count.value, dtype=IMPULSE)
    rc = lib.rvb_merge_images(_ptr(cand), _u64(cand.shape[0]), _ptr(d), ctypes.c_int(int(remove_direct)), _ptr(out),
                              _u64(out.shape[0]), ctypes.byref(count))
    if rc:
        raise RvbError(rc, "rvb_merge_images")
    return out


class Context:
    """One GPU, one HIP stream (rvb_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.handle = _vp()
        rc = self.lib.rvb_create(ctypes.byref(self.handle), ctypes.c_int(device), ctypes.c_uint(0))
        if rc:
            raise RvbError(rc, self.lib.rvb_last_error(None).decode())
        self.nrays = 0
        self.nreflections = 0
        self.nchannels = 0
        self._keep = []

    def close(self):
        if self.handle:
            self.lib.rvb_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise RvbError(rc, self.lib.rvb_last_error(self.handle).decode())

    # ---- Raytracer ------------------------------------------------------------------------------
    def set_scene(self, scene):
        triangles, vertices, surfaces = scene
        triangles, vertices, surfaces = (np.ascontiguousarray(x) for x in (triangles, vertices, surfaces))
        self._check(self.lib.rvb_set_scene(self.handle, _ptr(triangles), _u64(triangles.shape[0]), _ptr(vertices),
                                           _u64(vertices.shape[0]), _ptr(surfaces), _u64(surfaces.shape[0])))

    def share_scene(self, other):
        """This context reads the scene `other` holds: the same device buffers, no second build or copy (rvb_share_scene)."""
        self._check(self.lib.rvb_share_scene(self.handle, other.handle))

    def scene_info(self):
        nodes, kept, depth = _u64(0), _u64(0), ctypes.c_uint32(0)
        self._check(self.lib.rvb_scene_info(self.handle, ctypes.byref(nodes), ctypes.byref(kept), ctypes.byref(depth)))
        return {"nodes": nodes.value, "kept_triangles": kept.value, "depth": depth.value}

    def device_info(self):
        arch = ctypes.create_string_buffer(64)
        cus, hbm = ctypes.c_int(0), _u64(0)
        self._check(self.lib.rvb_device_info(self.handle, arch, _u64(64), ctypes.byref(cus), ctypes.byref(hbm)))
        return {"arch": arch.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    def set_directions(self, directions):
        d = np.ascontiguousarray(np.asarray(directions, np.float32).reshape(-1, 4))
        self._check(self.lib.rvb_set_directions(self.handle, _ptr(d), _u64(d.shape[0])))
        self.nrays = d.shape[0]

    def set_directions_device(self, device_pointer, nrays):
        self._check(self.lib.rvb_set_directions_device(self.handle, _vp(device_pointer), _u64(nrays)))
        self.nrays = nrays

    @staticmethod
    def trace_group(contexts, mics, sources, nreflections, air, ray_offsets=None):
        """rvb_trace on several contexts of one device with ONE path-kernel launch (rvb_trace_group)."""
        n = len(contexts)
        handles = (ctypes.c_void_p * n)(*[c.handle for c in contexts])
        m = np.ascontiguousarray(np.asarray(mics, dtype=np.float32).reshape(n, 3))
        s = np.ascontiguousarray(np.asarray(sources, dtype=np.float32).reshape(n, 3))
        offs = (ctypes.c_uint64 * n)(*[int(o) for o in (ray_offsets if ray_offsets is not None else [0] * n)])
        contexts[0]._check(contexts[0].lib.rvb_trace_group(handles, _u64(n), _ptr(m), _ptr(s), _u64(nreflections), _f8(air), offs))
        for c in contexts:
            c.nreflections = int(nreflections)
            c.npairs = 1

    def set_concurrent_traces(self, traces):
        """Hint: traces of this size the caller keeps in flight on the device at a time (rvb_set_concurrent_traces)."""
        self._check(self.lib.rvb_set_concurrent_traces(self.handle, ctypes.c_uint32(int(traces))))

    def set_path_lanes(self, lanes):
        """Test / measurement hook: lanes per ray of this context's path kernel (4, 2, 1; 0 = chosen per launch).  Same bytes either way."""
        self._check(self.lib.rvb_set_path_lanes(self.handle, ctypes.c_uint32(int(lanes))))

    def trace(self, mic, source, nreflections, air, ray_offset=0):
        self._check(self.lib.rvb_trace(self.handle, _f3(mic), _f3(source), _u64(nreflections), _f8(air), _u64(ray_offset)))
        self.nreflections = int(nreflections)
        self.npairs = 1

    def trace_pairs(self, mics, sources, nreflections, air, ray_offset=0):
        """Several (source, microphone) pairs in one launch (rvb_trace_pairs): [npairs][3] each; every pair uses the rays set on
        this context.  Raw results are [npairs][nrays][nreflections]; select_pair(p) picks the pair the IR calls work on."""
        m = np.ascontiguousarray(np.asarray(mics, dtype=np.float32).reshape(-1, 3))
        s_ = np.ascontiguousarray(np.asarray(sources, dtype=np.float32).reshape(-1, 3))
        assert m.shape == s_.shape and m.shape[0] >= 1
        self._check(self.lib.rvb_trace_pairs(self.handle, _ptr(m), _ptr(s_), _u64(m.shape[0]), _u64(nreflections), _f8(air), _u64(ray_offset)))
        self.nreflections = int(nreflections)
        self.npairs = int(m.shape[0])

    def select_pair(self, pair):
        self._check(self.lib.rvb_ir_select_pair(self.handle, _u64(pair)))

    def get_pair_candidates(self, pair, candidates=None):
        """Image-source candidates of one pair of a trace_pairs launch, ray numbers relative to the pair."""
        cand = self.get_image_candidates() if candidates is None else candidates
        mine = cand[(cand["ray"] // np.uint64(self.nrays)) == np.uint64(pair)].copy()
        mine["ray"] -= np.uint64(pair * self.nrays)
        return mine

    def raytrace(self, mic, source, directions, nreflections, air):
        """Raytracer::raytrace (reference rayverb.cpp:538-685)."""
        self.set_directions(directions)
        self.trace(mic, source, nreflections, air)

    def synchronize(self):
        self._check(self.lib.rvb_synchronize(self.handle))

    def get_raw_diffuse(self):
        out = np.zeros(getattr(self, "npairs", 1) * self.nrays * self.nreflections, dtype=IMPULSE)
        self._check(self.lib.rvb_get_diffuse(self.handle, _ptr(out)))
        return out

    def diffuse_device(self):
        p, n = _vp(), _u64(0)
        self._check(self.lib.rvb_diffuse_device(self.handle, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def get_direct(self):
        out = np.zeros(1, dtype=IMPULSE)
        self._check(self.lib.rvb_get_direct(self.handle, _ptr(out)))
        return out

    def get_image_candidates(self):
        count = _u64(0)
        self._check(self.lib.rvb_get_image_candidates(self.handle, None, _u64(0), ctypes.byref(count)))
        out = np.zeros(count.value, dtype=IMAGE_CANDIDATE)
        if count.value:
            self._check(self.lib.rvb_get_image_candidates(self.handle, _ptr(out), _u64(out.shape[0]), ctypes.byref(count)))
        return out

    def get_raw_images(self, remove_direct):
        """Raytracer::getRawImages (reference rayverb.cpp:692-706)."""
        direct = self.get_direct() if self.nrays else None
        return merge_images(self.get_image_candidates(), direct, remove_direct)

    def get_all_raw(self, remove_direct):
        return np.concatenate([self.get_raw_diffuse(), self.get_raw_images(remove_direct)])

    def executed_bounces(self):
        v = _u64(0)
        self._check(self.lib.rvb_executed_bounces(self.handle, ctypes.byref(v)))
        return v.value

    def debug_stamps(self):
        out = (ctypes.c_uint64 * 32)()
        self._check(self.lib.rvb_debug_stamps(self.handle, out, _u64(32)))
        return [int(x) for x in out]

    def last_timings(self):
        names = ctypes.create_string_buffer(1024)
        ms = (ctypes.c_float * 32)()
        count = _u64(0)
        self._check(self.lib.rvb_last_timings(self.handle, names, _u64(1024), ms, _u64(32), ctypes.byref(count)))
        keys = names.value.decode().split(";") if count.value else []
        return [(k, float(ms[i])) for i, k in enumerate(keys)]

    # ---- SpeakerAttenuator / HrtfAttenuator (materialised, one channel per call) --------------------
    def attenuate_speaker(self, mic, impulses, direction, coefficient):
        imp = np.ascontiguousarray(impulses, dtype=IMPULSE)
        sp = make_speakers([direction], [coefficient])
        out = np.zeros(imp.shape[0], dtype=ATTENUATED)
        self._check(self.lib.rvb_attenuate_speaker(self.handle, _f3(mic), _ptr(imp), _u64(imp.shape[0]), _ptr(sp), _ptr(out)))
        return out

    def attenuate_speaker_device(self, mic, d_in, n, direction, coefficient, d_out):
        """The materialised `attenuate` kernel on HBM-resident buffers (device addresses), asynchronous."""
        sp = make_speakers([direction], [coefficient])
        self._check(self.lib.rvb_attenuate_speaker_device(self.handle, _f3(mic), _vp(d_in), _u64(n), _ptr(sp), _vp(d_out)))

    def attenuate_hrtf_device(self, mic, d_in, n, table_channel, facing, up, channel, d_out):
        """The materialised `hrtf` kernel on HBM-resident buffers (device addresses), asynchronous."""
        table = np.ascontiguousarray(table_channel, dtype=np.float32).reshape(-1)
        assert table.shape[0] == 360 * 180 * 8
        self._check(self.lib.rvb_attenuate_hrtf_device(self.handle, _f3(mic), _vp(d_in), _u64(n), _ptr(table), _f3(facing), _f3(up),
                                                       _u64(channel), _vp(d_out)))

    def attenuate_hrtf(self, mic, impulses, table_channel, facing, up, channel):
        imp = np.ascontiguousarray(impulses, dtype=IMPULSE)
        table = np.ascontiguousarray(table_channel, dtype=np.float32).reshape(-1)
        assert table.shape[0] == 360 * 180 * 8
        out = np.zeros(imp.shape[0], dtype=ATTENUATED)
        self._check(self.lib.rvb_attenuate_hrtf(self.handle, _f3(mic), _ptr(imp), _u64(imp.shape[0]), _ptr(table), _f3(facing),
                                                _f3(up), _u64(channel), _ptr(out)))
        return out

    def flatten(self, attenuated, sample_rate):
        """flattenImpulses for one channel (reference rayverb.cpp:48-77) -> [8][nbins]."""
        a = np.ascontiguousarray(attenuated, dtype=ATTENUATED)
        nbins = _u64(0)
        self._check(self.lib.rvb_flatten(self.handle, _ptr(a), _u64(a.shape[0]), ctypes.c_float(sample_rate), None, _u64(0),
                                         ctypes.byref(nbins)))
        out = np.zeros((8, nbins.value), dtype=np.float32)
        self._check(self.lib.rvb_flatten(self.handle, _ptr(a), _u64(a.shape[0]), ctypes.c_float(sample_rate), _ptr(out),
                                         _u64(nbins.value), ctypes.byref(nbins)))
        return out

    # ---- fused impulse-response stage ------------------------------------------------------------------
    def ir_configure_speakers(self, mic, directions, coefficients, which=IR_ALL, images=None):
        sp = make_speakers(directions, coefficients)
        img = np.ascontiguousarray(images, dtype=IMPULSE) if images is not None else np.zeros(0, dtype=IMPULSE)
        self._check(self.lib.rvb_ir_configure_speakers(self.handle, _f3(mic), _ptr(sp), _u64(sp.shape[0]), ctypes.c_int(which),
                                                       _ptr(img), _u64(img.shape[0])))
        self.nchannels = sp.shape[0]

    def ir_configure_hrtf(self, mic, table, facing, up, which=IR_ALL, images=None):
        """table: [2][360][180][8] floats, or None = the table of this context's previous ir_configure_hrtf stays on the device (many
        listeners, one table: the 4 MB go up once)."""
        if table is not None:
            t = np.ascontiguousarray(table, dtype=np.float32).reshape(-1)
            assert t.shape[0] == 2 * 360 * 180 * 8
        img = np.ascontiguousarray(images, dtype=IMPULSE) if images is not None else np.zeros(0, dtype=IMPULSE)
        self._check(self.lib.rvb_ir_configure_hrtf(self.handle, _f3(mic), _ptr(t) if table is not None else None, _f3(facing), _f3(up), ctypes.c_int(which),
                                                   _ptr(img), _u64(img.shape[0])))
        self.nchannels = 2

    def ir_time_range(self):
        lo, hi = ctypes.c_float(0), ctypes.c_float(0)
        self._check(self.lib.rvb_ir_time_range(self.handle, ctypes.byref(lo), ctypes.byref(hi)))
        return lo.value, hi.value

    def ir_bins(self, max_time, predelay, sample_rate):
        return int(self.lib.rvb_ir_bins(ctypes.c_float(max_time), ctypes.c_float(predelay), ctypes.c_float(sample_rate)))

    def ir_accumulate(self, predelay, sample_rate, nbins, mode, device_histogram_pointer):
        self._check(self.lib.rvb_ir_accumulate(self.handle, ctypes.c_float(predelay), ctypes.c_float(sample_rate), _u64(nbins),
                                               ctypes.c_int(mode), _vp(device_histogram_pointer)))

    def ir_accumulate_wait_for_torch(self):
        """The context's stream waits (by an event, not the host) for everything enqueued so far on torch's current stream."""
        import torch
        ready = torch.cuda.Event()
        ready.record()
        self._check(self.lib.rvb_wait_for_event(self.handle, _vp(ready.cuda_event)))
        self._keep_event2 = ready

    def ir_accumulate_tensor(self, predelay, sample_rate, nbins, mode, tensor):
        """Adds into a zeroed torch CUDA tensor [nchannels][8][nbins] (plumbing for distributed.py).  The tensor was
        filled on torch's current stream: the context's stream waits for that fill through an event, the host does not."""
        import torch
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() == self.nchannels * 8 * nbins
        ready = torch.cuda.Event()
        ready.record()
        self._check(self.lib.rvb_wait_for_event(self.handle, _vp(ready.cuda_event)))
        self.ir_accumulate(predelay, sample_rate, nbins, mode, tensor.data_ptr())
        self._keep_event = ready                       # alive until the next call (the wait has been enqueued, not executed)

    def ir_accumulate_export_tensor(self, predelay, sample_rate, nbins, mode, tensor, pinned_host_tensor, slices=0):
        """ir_accumulate_tensor + export_tensor_to_host in one call (rvb_ir_accumulate_export): in exact mode with the speaker model
        the histogram leaves for the host bin range by bin range while the later ranges are still being folded."""
        import torch
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() == self.nchannels * 8 * nbins
        assert pinned_host_tensor.is_pinned() and pinned_host_tensor.is_contiguous()
        assert tensor.numel() * tensor.element_size() == pinned_host_tensor.numel() * pinned_host_tensor.element_size()
        ready = torch.cuda.Event()
        ready.record()
        self._check(self.lib.rvb_wait_for_event(self.handle, _vp(ready.cuda_event)))
        self._check(self.lib.rvb_ir_accumulate_export(self.handle, ctypes.c_float(predelay), ctypes.c_float(sample_rate), _u64(nbins),
                                                      ctypes.c_int(mode), _vp(tensor.data_ptr()), _vp(pinned_host_tensor.data_ptr()),
                                                      ctypes.c_uint32(int(slices))))
        self._keep_event = ready

    def ir_exact_prepare(self, predelay, sample_rate, nbins):
        """Exact mode, step 1: keys, sort, run boundaries of this context's impulses (rvb_ir_exact_prepare)."""
        self._check(self.lib.rvb_ir_exact_prepare(self.handle, ctypes.c_float(predelay), ctypes.c_float(sample_rate), _u64(nbins)))

    def ir_exact_fold_tensor(self, nbins, bin_begin, bin_end, tensor):
        """Exact mode, step 2: bins [bin_begin, bin_end) folded on top of what the torch CUDA tensor [nchannels][8][nbins] holds; the
        context's stream first waits (by an event) for what torch's current stream has done to the tensor."""
        import torch
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() == self.nchannels * 8 * nbins
        ready = torch.cuda.Event()
        ready.record()
        self._check(self.lib.rvb_wait_for_event(self.handle, _vp(ready.cuda_event)))
        self._check(self.lib.rvb_ir_exact_fold(self.handle, _u64(nbins), _u64(bin_begin), _u64(bin_end), _vp(tensor.data_ptr())))
        self._keep_event = ready

    def export_tensor_to_host(self, tensor, pinned_host_tensor):
        """Enqueues, behind the binning, the copy of a device tensor (the histogram ir_accumulate_tensor filled) into a PINNED host
        tensor of the same size (rvb_copy_to_pinned_host_async: on the context's export stream); synchronize_exports()
        waits for it, synchronize() does not.  The caller keeps `tensor` alive and untouched until then."""
        assert tensor.is_cuda and tensor.is_contiguous() and pinned_host_tensor.is_pinned() and pinned_host_tensor.is_contiguous()
        assert tensor.numel() * tensor.element_size() == pinned_host_tensor.numel() * pinned_host_tensor.element_size()
        self._check(self.lib.rvb_copy_to_pinned_host_async(self.handle, _vp(pinned_host_tensor.data_ptr()), _vp(tensor.data_ptr()),
                                                           _u64(tensor.numel() * tensor.element_size())))

    def synchronize_exports(self):
        self._check(self.lib.rvb_synchronize_exports(self.handle))

    def ir_download(self, trim_predelay, sample_rate, mode=IR_FAST):
        """attenuate -> fixPredelay -> flattenImpulses (reference cmd/main.cpp:280-298) -> [nch][8][nbins]."""
        nbins = _u64(0)
        self._check(self.lib.rvb_ir_download(self.handle, ctypes.c_int(int(trim_predelay)), ctypes.c_float(sample_rate),
                                             ctypes.c_int(mode), None, _u64(0), ctypes.byref(nbins)))
        out = np.zeros((self.nchannels, 8, nbins.value), dtype=np.float32)
        self._check(self.lib.rvb_ir_download(self.handle, ctypes.c_int(int(trim_predelay)), ctypes.c_float(sample_rate),
                                             ctypes.c_int(mode), _ptr(out), _u64(nbins.value), ctypes.byref(nbins)))
        return out


MULTI_REHEARSE_RCCL = 1


class MultiContext:
    """Several GPUs of one node behind the C-ABI (rvb_multi_*): ray shards, merged image sources, histograms combined on the
    devices (exact mode: one chained serial sum, bit-identical to a single context; fast mode: RCCL all-reduce)."""

    def __init__(self, devices, flags=0):
        self.lib = load_library()
        self.handle = _vp()
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        rc = self.lib.rvb_multi_create(ctypes.byref(self.handle), devs, ctypes.c_int(len(devices)), ctypes.c_uint(flags))
        if rc:
            raise RvbError(rc, self.lib.rvb_last_error(None).decode())
        self.nrays = self.nreflections = 0

    def close(self):
        if self.handle:
            self.lib.rvb_multi_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise RvbError(rc, self.lib.rvb_multi_last_error(self.handle).decode())

    def set_scene(self, scene):
        triangles, vertices, surfaces = (np.ascontiguousarray(x) for x in scene)
        self._check(self.lib.rvb_multi_set_scene(self.handle, _ptr(triangles), _u64(triangles.shape[0]), _ptr(vertices),
                                                 _u64(vertices.shape[0]), _ptr(surfaces), _u64(surfaces.shape[0])))

    def raytrace(self, mic, source, directions, nreflections, air):
        d = np.ascontiguousarray(np.asarray(directions, np.float32).reshape(-1, 4))
        self._check(self.lib.rvb_multi_set_directions(self.handle, _ptr(d), _u64(d.shape[0])))
        self._check(self.lib.rvb_multi_trace(self.handle, _f3(mic), _f3(source), _u64(nreflections), _f8(air)))
        self.nrays, self.nreflections = d.shape[0], int(nreflections)

    def shard(self, index):
        first, count = _u64(0), _u64(0)
        self._check(self.lib.rvb_multi_context(self.handle, ctypes.c_int(index), None, ctypes.byref(first), ctypes.byref(count)))
        return first.value, count.value

    def used_rccl(self):
        return bool(self.lib.rvb_multi_used_rccl(self.handle))

    def set_chain_blocks(self, blocks):
        """Bin-range blocks the exact mode's histogram travels in from device to device (rvb_multi_set_chain_blocks; same bytes for any count)."""
        self._check(self.lib.rvb_multi_set_chain_blocks(self.handle, ctypes.c_uint32(int(blocks))))

    def peer_links(self):
        return int(self.lib.rvb_multi_peer_links(self.handle))

    def get_raw_diffuse(self):
        out = np.zeros(self.nrays * self.nreflections, dtype=IMPULSE)
        self._check(self.lib.rvb_multi_get_diffuse(self.handle, _ptr(out)))
        return out

    def get_raw_images(self, remove_direct):
        count = _u64(0)
        self._check(self.lib.rvb_multi_get_images(self.handle, ctypes.c_int(int(remove_direct)), None, _u64(0), ctypes.byref(count)))
        out = np.zeros(count.value, dtype=IMPULSE)
        self._check(self.lib.rvb_multi_get_images(self.handle, ctypes.c_int(int(remove_direct)), _ptr(out), _u64(out.shape[0]), ctypes.byref(count)))
        return out

    def ir_speakers(self, mic, directions, coefficients, trim_predelay, sample_rate, mode, which=IR_ALL, remove_direct=False):
        sp = make_speakers(directions, coefficients)
        args = [self.handle, _f3(mic), _ptr(sp), _u64(sp.shape[0]), ctypes.c_int(which), ctypes.c_int(int(remove_direct)),
                ctypes.c_int(int(trim_predelay)), ctypes.c_float(sample_rate), ctypes.c_int(mode)]
        nbins = _u64(0)
        self._check(self.lib.rvb_multi_ir_speakers(*args, None, _u64(0), ctypes.byref(nbins)))
        out = np.zeros((sp.shape[0], 8, nbins.value), dtype=np.float32)
        self._check(self.lib.rvb_multi_ir_speakers(*args, _ptr(out), _u64(nbins.value), ctypes.byref(nbins)))
        return out

    def ir_hrtf(self, mic, table, facing, up, trim_predelay, sample_rate, mode, which=IR_ALL, remove_direct=False):
        t = np.ascontiguousarray(table, dtype=np.float32).reshape(-1)
        assert t.shape[0] == 2 * 360 * 180 * 8
        args = [self.handle, _f3(mic), _ptr(t), _f3(facing), _f3(up), ctypes.c_int(which), ctypes.c_int(int(remove_direct)),
                ctypes.c_int(int(trim_predelay)), ctypes.c_float(sample_rate), ctypes.c_int(mode)]
        nbins = _u64(0)
        self._check(self.lib.rvb_multi_ir_hrtf(*args, None, _u64(0), ctypes.byref(nbins)))
        out = np.zeros((2, 8, nbins.value), dtype=np.float32)
        self._check(self.lib.rvb_multi_ir_hrtf(*args, _ptr(out), _u64(nbins.value), ctypes.byref(nbins)))
        return out


class PipelineResult(ctypes.Structure):
    """rvb_pipeline_result of include/rvb_capi.h."""
    _fields_ = [("job", ctypes.c_uint64), ("histogram", ctypes.POINTER(ctypes.c_float)), ("nchannels", ctypes.c_uint64), ("nbins", ctypes.c_uint64),
                ("predelay", ctypes.c_float), ("max_time", ctypes.c_float), ("nimages", ctypes.c_uint64)]


class Pipeline:
    """Impulse responses back to back behind the C-ABI (rvb_pipeline_*, csrc/pipeline.hip): the native form of
    distributed.IrPipeline — groups of traces in one path-kernel launch, the next groups' traces enqueued ahead, the binning stages of a
    group enqueued together, histograms exported to a ring of pinned buffers.  `contexts`: capi.Context objects of one GPU with the
    same scene and rays set."""

    def __init__(self, contexts, group=0):
        self.lib = load_library()
        self.contexts = list(contexts)
        self.handle = _vp()
        handles = (_vp * len(self.contexts))(*[c.handle for c in self.contexts])
        rc = self.lib.rvb_pipeline_create(ctypes.byref(self.handle), handles, _u64(len(self.contexts)), _u64(int(group)))
        if rc:
            raise RvbError(rc, "rvb_pipeline_create failed")

    def close(self):
        if self.handle:
            self.lib.rvb_pipeline_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise RvbError(rc, self.lib.rvb_pipeline_last_error(self.handle).decode())

    def configure_speakers(self, directions, coefficients, nreflections, air, sample_rate=44100.0, trim_predelay=True, mode=IR_EXACT,
                           which=IR_ALL, remove_direct=False):
        sp = make_speakers(directions, coefficients)
        self._check(self.lib.rvb_pipeline_configure_speakers(self.handle, _ptr(sp), _u64(sp.shape[0]), ctypes.c_int(which), ctypes.c_int(int(remove_direct)),
                                                             ctypes.c_int(int(trim_predelay)), ctypes.c_float(sample_rate), ctypes.c_int(mode),
                                                             _u64(nreflections), _f8(air)))

    def configure_hrtf(self, table, facing, up, nreflections, air, sample_rate=44100.0, trim_predelay=True, mode=IR_EXACT, which=IR_ALL,
                       remove_direct=False):
        t = np.ascontiguousarray(table, dtype=np.float32)
        assert t.size == 2 * 360 * 180 * 8
        self._check(self.lib.rvb_pipeline_configure_hrtf(self.handle, _ptr(t), _f3(facing), _f3(up), ctypes.c_int(which), ctypes.c_int(int(remove_direct)),
                                                         ctypes.c_int(int(trim_predelay)), ctypes.c_float(sample_rate), ctypes.c_int(mode),
                                                         _u64(nreflections), _f8(air)))

    def submit(self, mic, source, facing=None, up=None):
        if facing is None:
            self._check(self.lib.rvb_pipeline_submit(self.handle, _f3(mic), _f3(source)))
        else:
            self._check(self.lib.rvb_pipeline_submit_oriented(self.handle, _f3(mic), _f3(source), _f3(facing), _f3(up)))

    def pending(self):
        return int(self.lib.rvb_pipeline_pending(self.handle))

    def next(self, copy=True):
        """The oldest pending job: (histogram [nchannels][8][nbins] — a copy, or a view of the pipeline's pinned buffer that stays valid
        until len(contexts) further results have been taken —, info dict)."""
        res = PipelineResult()
        self._check(self.lib.rvb_pipeline_next(self.handle, ctypes.byref(res)))
        view = np.ctypeslib.as_array(res.histogram, shape=(int(res.nchannels), 8, int(res.nbins)))
        info = {"job": int(res.job), "nbins": int(res.nbins), "predelay": float(res.predelay), "max_time": float(res.max_time), "images": int(res.nimages)}
        return (view.copy() if copy else view), info
