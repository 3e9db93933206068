"""Host-side mirror of the reference's kernel interface, over the C-ABI.

The reference drives its kernels through ComputeApplication::RunOnGPU (src/main.cpp:1307-1730):
bind {output / WeightInfo buffer, image(s)}, push {w, h, sigmas | h}, dispatch.  `Context`
offers the same operators with the same argument meaning; every call goes through
libmi_denoise.so (ctypes) -- there is no NumPy/PyTorch compute path here, and a missing
library or GPU is an error, not a fallback.

NumPy arrays are (h, w, 4): float32 = RGBA32F (.exr path), uint8 = RGBA8 (.png path).
WeightInfo buffers are float32 (h, w, 8): [wc.r, wc.g, wc.b, wc.a, normWeight, pad, pad, pad].
"""
import ctypes
import weakref

import numpy as np

from ._lib import BilateralParams, Image, NlmParams, NormalizeParams, c_void_pp, lib

FMT_RGBA32F, FMT_RGBA8 = 0, 1
LAYOUT_TEXTURE, LAYOUT_LINEAR = 0, 1

# nonlocal.comp:5-6 as shipped, and the 21x21 / 7x7 benchmark configuration (half-open ranges)
NLM_REFERENCE = dict(search=(-7, 7), patch=(-3, 3))
NLM_BENCH = dict(search=(-10, 11), patch=(-3, 4))


class MidError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib.mid_last_error().decode("utf-8", "replace")
        super().__init__(f"{where}: error {code}: {msg}")


def _check(code, where):
    if code != 0:
        raise MidError(code, where)


def load_image(path):
    """LoadImages (src/main.cpp:145-229): '.exr' -> float32 (h,w,4), anything else as PNG -> uint8 (h,w,4).
    Host-only (no GPU)."""
    img = Image()
    _check(lib.mid_image_load(str(path).encode(), ctypes.byref(img)), "mid_image_load")
    try:
        dt = np.float32 if img.format == FMT_RGBA32F else np.uint8
        n = img.width * img.height * 4
        arr = np.ctypeslib.as_array(ctypes.cast(img.data, ctypes.POINTER(ctypes.c_float if dt == np.float32 else ctypes.c_uint8)),
                                    shape=(n,)).reshape(img.height, img.width, 4).copy()
    finally:
        lib.mid_image_free(ctypes.byref(img))
    return arr


def save_image(path, arr):
    """SaveEXR(rgba,w,h,4,0) for float32, lodepng::encode for uint8 (src/main.cpp:1699,1717)."""
    arr = _img(arr)
    _check(lib.mid_image_save(str(path).encode(), arr.ctypes.data, arr.shape[1], arr.shape[0], _fmt_of(arr)),
           "mid_image_save")


class DeviceBuffer:
    """A hipMalloc'd buffer owned by Python (the application owns every buffer, like the reference)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = ctypes.c_void_p()
        _check(lib.mid_alloc(ctx.handle, self.nbytes, ctypes.byref(p)), "mid_alloc")
        self.ptr = p.value
        ctx._live.add(self)

    def free(self):
        # a buffer that outlives its context was already released by Context.close() (ptr is None then)
        if self.ptr and self.ctx.handle:
            lib.mid_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _fmt_of(a):
    if a.dtype == np.float32:
        return FMT_RGBA32F
    if a.dtype == np.uint8:
        return FMT_RGBA8
    raise TypeError(f"image dtype must be float32 or uint8, got {a.dtype}")


def _img(a):
    a = np.ascontiguousarray(a)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError(f"image must be (h, w, 4), got {a.shape}")
    return a


def _same_frames(frames, what):
    """Every frame of a sequence is read with frames[0]'s size and format: refuse anything else up front."""
    frames = [_img(f) for f in frames]
    if not frames:
        raise ValueError(f"{what}: no frames")
    _fmt_of(frames[0])
    for i, f in enumerate(frames):
        if f.shape != frames[0].shape or f.dtype != frames[0].dtype:
            raise ValueError(f"{what}: frame {i} is {f.shape} {f.dtype}, frame 0 is {frames[0].shape} {frames[0].dtype}")
    return frames


class Context:
    """One device + its streams (mid_ctx).  Replaces the reference's Vulkan instance/device/queue."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        _check(lib.mid_ctx_create(int(device), ctypes.byref(h)), "mid_ctx_create")
        self.handle = h
        self.device = device
        self._live = weakref.WeakSet()      # DeviceBuffers allocated through this context and not yet freed

    def close(self):
        if self.handle:
            for buf in list(self._live):    # mid_free needs a live context: release what is still held, then the context
                buf.free()
            lib.mid_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def release_cached(self):
        """Free the device ring / output slots / events the frame pipeline keeps in the context between calls."""
        _check(lib.mid_ctx_release_cached(self.handle), "mid_ctx_release_cached")

    @property
    def name(self):
        buf = ctypes.create_string_buffer(160)
        _check(lib.mid_device_name(self.handle, buf, 160), "mid_device_name")
        return buf.value.decode()

    # ---- memory ---------------------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # NumPy arrays are pageable memory.  The C-ABI itself keeps such memory away from the HIP runtime's pin-on-the-fly path
    # (mid_memcpy_h2d / mid_memcpy_d2h move it through the context's page-locked bounce buffers, csrc/hostcopy.cpp), so
    # these two are plain calls: what the tests exercise here is what any caller of the library gets.
    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        buf = DeviceBuffer(self, max(arr.nbytes, 16))
        if arr.nbytes:
            _check(lib.mid_memcpy_h2d(self.handle, buf.ptr, arr.ctypes.data, arr.nbytes, stream), "mid_memcpy_h2d")
            self.sync(stream)
        return buf

    def download(self, buf, shape, dtype, stream=None):
        out = np.empty(shape, dtype=dtype)
        ptr = buf.ptr if isinstance(buf, DeviceBuffer) else int(buf)
        if out.nbytes:
            _check(lib.mid_memcpy_d2h(self.handle, out.ctypes.data, ptr, out.nbytes, stream), "mid_memcpy_d2h")
            self.sync(stream)
        return out

    def zeros(self, nbytes, stream=None):
        buf = DeviceBuffer(self, nbytes)
        _check(lib.mid_memset(self.handle, buf.ptr, 0, nbytes, stream), "mid_memset")
        return buf

    def sync(self, stream=None):
        _check(lib.mid_stream_sync(self.handle, stream), "mid_stream_sync")

    # ---- raw (device pointer) operators: what bench.py and the pipeline use ------------------
    def record(self, stream=None):
        """`with ctx.record(stream) as rec:` -- the kernel-level calls issued on `stream` inside the block are RECORDED instead of executed
        (vkBeginCommandBuffer ... vkEndCommandBuffer, src/main.cpp:791/846); afterwards `rec.submit(stream)` runs the sequence
        (vkQueueSubmit, :1091) as often as needed."""
        return Recording(self, stream)

    def bilateral_dev(self, in_ptr, out_ptr, w, h, radius, sigma_s, sigma_c, layout, fmt, stream=None):
        p = BilateralParams(w, h, sigma_s, sigma_c, radius, layout, fmt)
        _check(lib.mid_bilateral(self.handle, ctypes.byref(p), in_ptr, out_ptr, stream), "mid_bilateral")

    def bilateral_batch_dev(self, in_ptrs, out_ptrs, w, h, radius, sigma_s, sigma_c, layout, fmt, stream=None):
        if len(out_ptrs) != len(in_ptrs) or not in_ptrs:
            raise ValueError(f"{len(in_ptrs)} input and {len(out_ptrs)} output pointers")
        p = BilateralParams(w, h, sigma_s, sigma_c, radius, layout, fmt)
        fi = (ctypes.c_void_p * len(in_ptrs))(*in_ptrs)
        fo = (ctypes.c_void_p * len(out_ptrs))(*out_ptrs)
        _check(lib.mid_bilateral_batch(self.handle, ctypes.byref(p), fi, fo, len(in_ptrs), stream), "mid_bilateral_batch")

    def nlm_temporal_dev(self, frame_ptrs, out_ptrs, w, h, hparam, search, patch, k, first, count, fmt, stream=None):
        # (the frame range itself is checked by the C side -> MID_ERR_INVALID; what C cannot see is the length of
        # the ctypes output table it is about to read `count` entries of)
        if len(out_ptrs) < count:
            raise ValueError(f"{count} output frames asked for, only {len(out_ptrs)} output pointers given")
        p = NlmParams(w, h, hparam, search[0], search[1], patch[0], patch[1], fmt)
        fr = (ctypes.c_void_p * len(frame_ptrs))(*frame_ptrs)
        ou = (ctypes.c_void_p * len(out_ptrs))(*out_ptrs)
        _check(lib.mid_nlm_temporal(self.handle, ctypes.byref(p), fr, len(frame_ptrs), k, first, count, ou, stream),
               "mid_nlm_temporal")

    # ---- NumPy-level operators (tests, smoke) ----------------------------------------------
    def bilateral(self, img, radius, sigma_s=2.0, sigma_c=0.2, layout="texture"):
        """bialteral.comp (layout='texture') / bialteral_linear.comp (layout='linear')."""
        img = _img(img)
        h, w = img.shape[:2]
        lay = {"texture": LAYOUT_TEXTURE, "linear": LAYOUT_LINEAR}[layout]
        d_in, d_out = self.upload(img), self.alloc(w * h * 16)
        self.bilateral_dev(d_in.ptr, d_out.ptr, w, h, radius, sigma_s, sigma_c, lay, _fmt_of(img))
        return self.download(d_out, (h, w, 4), np.float32)

    def bilateral_batch(self, frames, radius, sigma_s=2.0, sigma_c=0.2, layout="texture"):
        """n independent frames, one launch (mid_bilateral_batch)."""
        frames = _same_frames(frames, "bilateral_batch")
        h, w = frames[0].shape[:2]
        lay = {"texture": LAYOUT_TEXTURE, "linear": LAYOUT_LINEAR}[layout]
        d_in = [self.upload(f) for f in frames]
        d_out = [self.alloc(w * h * 16) for _ in frames]
        self.bilateral_batch_dev([d.ptr for d in d_in], [d.ptr for d in d_out], w, h, radius, sigma_s, sigma_c, lay, _fmt_of(frames[0]))
        return [self.download(d, (h, w, 4), np.float32) for d in d_out]

    def bilateral_layers_accum(self, img, layer, W, radius, sigma_s=2.0, sigma_c=0.2):
        """One dispatch of bialteral_layers.comp: returns W + this layer's sums."""
        img, layer = _img(img), _img(layer)
        if layer.dtype != np.uint8:
            raise TypeError("layers are always RGBA8 (src/main.cpp:1396)")
        h, w = img.shape[:2]
        W = np.ascontiguousarray(W, dtype=np.float32)
        d_in, d_l, d_w = self.upload(img), self.upload(layer), self.upload(W)
        p = BilateralParams(w, h, sigma_s, sigma_c, radius, LAYOUT_TEXTURE, _fmt_of(img))
        _check(lib.mid_bilateral_layers_accum(self.handle, ctypes.byref(p), d_in.ptr, d_l.ptr, d_w.ptr, None),
               "mid_bilateral_layers_accum")
        return self.download(d_w, (h, w, 8), np.float32)

    def bilateral_layers(self, img, layers, radius, sigma_s=2.0, sigma_c=0.2):
        """The per-layer loop + normalize, fused (src/main.cpp:1610-1623,1649-1652)."""
        img = _img(img)
        h, w = img.shape[:2]
        d_in, d_out = self.upload(img), self.alloc(w * h * 16)
        d_layers = [self.upload(_img(l)) for l in layers]
        tbl = (ctypes.c_void_p * max(len(d_layers), 1))(*[d.ptr for d in d_layers])
        p = BilateralParams(w, h, sigma_s, sigma_c, radius, LAYOUT_TEXTURE, _fmt_of(img))
        _check(lib.mid_bilateral_layers(self.handle, ctypes.byref(p), d_in.ptr, tbl, len(d_layers), d_out.ptr, None),
               "mid_bilateral_layers")
        return self.download(d_out, (h, w, 4), np.float32)

    def nlm_accum(self, target, neighbour, W, hparam=0.5, search=(-7, 7), patch=(-3, 3)):
        """One dispatch of nonlocal.comp: returns W + this neighbour frame's sums."""
        target, neighbour = _img(target), _img(neighbour)
        h, w = target.shape[:2]
        W = np.ascontiguousarray(W, dtype=np.float32)
        d_t, d_n, d_w = self.upload(target), self.upload(neighbour), self.upload(W)
        p = NlmParams(w, h, hparam, search[0], search[1], patch[0], patch[1], _fmt_of(target))
        _check(lib.mid_nlm_accum(self.handle, ctypes.byref(p), d_t.ptr, d_n.ptr, d_w.ptr, None), "mid_nlm_accum")
        return self.download(d_w, (h, w, 8), np.float32)

    def nlm_temporal(self, frames, k=0, first=0, count=None, hparam=0.5, search=(-7, 7), patch=(-3, 3)):
        """Fused accumulate-over-neighbour-frames + normalize for `count` output frames."""
        frames = _same_frames(frames, "nlm_temporal")
        count = len(frames) - first if count is None else count
        h, w = frames[0].shape[:2]
        d_fr = [self.upload(f) for f in frames]
        d_out = [self.alloc(w * h * 16) for _ in range(count)]
        self.nlm_temporal_dev([d.ptr for d in d_fr], [d.ptr for d in d_out], w, h, hparam, search, patch,
                              k, first, count, _fmt_of(frames[0]))
        return [self.download(d, (h, w, 4), np.float32) for d in d_out]

    def normalize(self, W):
        """normalize.comp."""
        W = np.ascontiguousarray(W, dtype=np.float32)
        h, w = W.shape[:2]
        d_w, d_out = self.upload(W), self.alloc(w * h * 16)
        p = NormalizeParams(w, h)
        _check(lib.mid_normalize(self.handle, ctypes.byref(p), d_w.ptr, d_out.ptr, None), "mid_normalize")
        return self.download(d_out, (h, w, 4), np.float32)

    def unpack_u8(self, u8, flavour=0):
        u8 = np.ascontiguousarray(u8, dtype=np.uint8)
        n = u8.size
        d_in, d_out = self.upload(u8), self.alloc(max(n * 4, 16))
        _check(lib.mid_unpack_u8(self.handle, d_in.ptr, n, flavour, d_out.ptr, None), "mid_unpack_u8")
        return self.download(d_out, u8.shape, np.float32)

    def pack_u8(self, f32):
        f32 = np.ascontiguousarray(f32, dtype=np.float32)
        n = f32.size
        d_in, d_out = self.upload(f32), self.alloc(max(n, 16))
        _check(lib.mid_pack_u8(self.handle, d_in.ptr, n, d_out.ptr, None), "mid_pack_u8")
        return self.download(d_out, f32.shape, np.uint8)

    def nlm_multiframe(self, target, frames, overlap=True, hparam=0.5, search=(-7, 7), patch=(-3, 3)):
        """The reference's multi-frame mode: one target, neighbour frames streamed (mid_nlm_multiframe)."""
        target = _img(target)
        frames = _same_frames([target] + list(frames), "nlm_multiframe")[1:]
        h, w = target.shape[:2]
        out = np.empty((h, w, 4), np.float32)
        prm = NlmParams(w, h, hparam, search[0], search[1], patch[0], patch[1], _fmt_of(target))
        t = (ctypes.c_float * 3)()
        tbl = (ctypes.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        _check(lib.mid_nlm_multiframe(self.handle, ctypes.byref(prm), target.ctypes.data, tbl, len(frames),
                                      out.ctypes.data, 1 if overlap else 0, t), "mid_nlm_multiframe")
        return out, tuple(t)

    def sequence_nlm_pinned(self, hin, hout, w, h, fmt, k=2, first=0, count=None, overlap=True, hparam=0.5,
                            search=(-7, 7), patch=(-3, 3), out_u8=False):
        """mid_sequence_nlm_range[_u8] on host pointers the caller already holds (PinnedFrames.ptrs): nothing but the C call,
        so a clock around it measures what a C caller sees.  Returns (wall_ms of the whole call, kernel_ms, copy_ms)."""
        n = len(hin)
        count = n - first if count is None else count
        if len(hout) < count:
            raise ValueError(f"{count} outputs asked for, {len(hout)} output buffers given")
        prm = NlmParams(w, h, hparam, search[0], search[1], patch[0], patch[1], fmt)
        t = (ctypes.c_float * 3)()
        entry = lib.mid_sequence_nlm_range_u8 if out_u8 else lib.mid_sequence_nlm_range
        _check(entry(self.handle, ctypes.byref(prm), (ctypes.c_void_p * n)(*hin), n, k, first, count,
                     (ctypes.c_void_p * count)(*hout[:count]), 1 if overlap else 0, t), "mid_sequence_nlm_range")
        return tuple(t)

    def pipe_last_timeline(self, cap=4096):
        """mid_pipe_last_timeline: the device timeline of this context's last mid_sequence_nlm* call, from its own events.
        Returns (uploads, outputs): uploads = [(frame, start_ms, end_ms)], outputs = [(frame, kernel_start, kernel_end,
        download_start, download_end)], ms from the start of the call's first upload."""
        up, out = (ctypes.c_float * (2 * cap))(), (ctypes.c_float * (4 * cap))()
        nu, fu, no, fo = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib.mid_pipe_last_timeline(self.handle, cap, up, ctypes.byref(nu), ctypes.byref(fu), out, ctypes.byref(no),
                                          ctypes.byref(fo)), "mid_pipe_last_timeline")
        return ([(fu.value + i, up[2 * i], up[2 * i + 1]) for i in range(nu.value)],
                [(fo.value + j, out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]) for j in range(no.value)])

    def sequence_nlm(self, frames, k=2, overlap=True, hparam=0.5, search=(-7, 7), patch=(-3, 3), pinned=True,
                     first=0, count=None, out_u8=False, pinned_out=True):
        """Host frames in, host frames out through the overlapped pipeline (mid_sequence_nlm_range[_u8]).
        out_u8: outputs converted to RGBA8 on the device like the reference's read-back (src/main.cpp:97-103).
        pinned / pinned_out = False: the NumPy arrays themselves (pageable memory) are the sources / destinations, which
        the library moves through its bounce buffers (csrc/hostcopy.cpp).
        Returns (outputs for frames first..first+count-1, (wall_ms, kernel_ms, copy_ms))."""
        frames = _same_frames(frames, "sequence_nlm")
        n = len(frames)
        count = n - first if count is None else count
        if not (0 <= first and count >= 1 and first + count <= n):
            raise ValueError(f"outputs [{first}, {first + count}) are not inside the {n} frames given")
        h, w = frames[0].shape[:2]
        out_shape, out_dtype = (h, w, 4), (np.uint8 if out_u8 else np.float32)
        hin = PinnedFrames(self, frames) if pinned else None
        hout = PinnedFrames(self, count, w * h * (4 if out_u8 else 16)) if pinned_out else None
        outs = None if pinned_out else [np.empty(out_shape, out_dtype) for _ in range(count)]
        try:
            t = self.sequence_nlm_pinned(hin.ptrs if pinned else [f.ctypes.data for f in frames],
                                         hout.ptrs if pinned_out else [o.ctypes.data for o in outs], w, h, _fmt_of(frames[0]),
                                         k, first, count, overlap, hparam, search, patch, out_u8)
            return (outs if outs is not None else [hout.array(i, out_shape, out_dtype) for i in range(count)]), t
        finally:
            if hin is not None:
                hin.free()
            if hout is not None:
                hout.free()


class Recording:
    """A recorded sequence of kernel-level calls (mid_record_begin / mid_record_end / mid_recording_submit, csrc/recording.cpp)."""

    def __init__(self, ctx, stream=None):
        self.ctx, self.stream, self.handle = ctx, stream, None

    def __enter__(self):
        _check(lib.mid_record_begin(self.ctx.handle, self.stream), "mid_record_begin")
        return self

    def __exit__(self, exc_type, exc, tb):
        h = ctypes.c_void_p()
        rc = lib.mid_record_end(self.ctx.handle, self.stream, ctypes.byref(h))      # (always: the stream must leave capture mode)
        if exc_type is None:
            _check(rc, "mid_record_end")
            self.handle = h
        elif rc == 0:
            lib.mid_recording_destroy(h)
        return False

    def submit(self, stream=None):
        if self.handle is None:
            raise ValueError("the recording was not completed")
        _check(lib.mid_recording_submit(self.handle, stream), "mid_recording_submit")

    def info(self):
        n, k = ctypes.c_int(), ctypes.c_int()
        _check(lib.mid_recording_info(self.handle, ctypes.byref(n), ctypes.byref(k)), "mid_recording_info")
        return n.value, k.value

    def close(self):
        if self.handle is not None:
            lib.mid_recording_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class PinnedFrames:
    """Page-locked host buffers (mid_alloc_host): the DMA source / destination of the frame pipeline, as the reference's
    mapped staging buffer is (src/main.cpp:247-275).  PinnedFrames(ctx, frames) copies the frames in; PinnedFrames(ctx, n,
    nbytes) allocates n empty buffers."""

    def __init__(self, ctx, frames_or_n, nbytes=None):
        self.ctx, self.ptrs = ctx, []
        frames = None if nbytes is not None else [np.ascontiguousarray(f) for f in frames_or_n]
        sizes = [nbytes] * int(frames_or_n) if frames is None else [f.nbytes for f in frames]
        self.nbytes = sizes[0] if sizes else 0
        try:
            for i, sz in enumerate(sizes):
                p = ctypes.c_void_p()
                _check(lib.mid_alloc_host(ctx.handle, sz, ctypes.byref(p)), "mid_alloc_host")
                self.ptrs.append(p.value)
                if frames is not None:
                    ctypes.memmove(p.value, frames[i].ctypes.data, sz)
        except Exception:
            self.free()
            raise

    def array(self, i, shape, dtype):
        out = np.empty(shape, dtype)
        ctypes.memmove(out.ctypes.data, self.ptrs[i], out.nbytes)
        return out

    def free(self):
        for p in self.ptrs:
            if self.ctx.handle:
                lib.mid_free_host(self.ctx.handle, p)
        self.ptrs = []


# ---- an animation sharded over GPUs: the C++ RCCL path (csrc/sharded.cpp) ----------------------------------------
COMM_ID_BYTES = 128


def shard_block(n_frames, world, rank):
    """(start, count) of rank's contiguous frame block (mid_shard_block) -- host only."""
    a, b = ctypes.c_int(), ctypes.c_int()
    _check(lib.mid_shard_block(n_frames, world, rank, ctypes.byref(a), ctypes.byref(b)), "mid_shard_block")
    return a.value, b.value


def shard_halo_plan(n_frames, world, k, rank):
    """(recv, send): lists of (peer rank, global frame id) in the order csrc/sharded.cpp issues them -- host only."""
    cap = 2 * max(k, 1) * max(world, 1) + 4 * max(k, 1) + 8
    arr = [(ctypes.c_int * cap)() for _ in range(4)]
    nr, ns = ctypes.c_int(), ctypes.c_int()
    _check(lib.mid_shard_halo_plan(n_frames, world, k, rank, cap, arr[0], arr[1], ctypes.byref(nr), arr[2], arr[3], ctypes.byref(ns)),
           "mid_shard_halo_plan")
    return ([(arr[0][i], arr[1][i]) for i in range(nr.value)], [(arr[2][i], arr[3][i]) for i in range(ns.value)])


def shard_launch_plan(n_frames, world, k, rank):
    """[(phase, w_lo, w_hi, first, count, out_offset)] like sharding.block_launch_plan, from the C++ side -- host only."""
    rows = (ctypes.c_int * (6 * 8))()
    n = ctypes.c_int()
    _check(lib.mid_shard_launch_plan(n_frames, world, k, rank, 8, rows, ctypes.byref(n)), "mid_shard_launch_plan")
    return [("interior" if rows[6 * i] else "boundary",) + tuple(rows[6 * i + 1:6 * i + 6]) for i in range(n.value)]


def comm_unique_id():
    """ncclGetUniqueId through the C-ABI: 128 bytes rank 0 hands to the other ranks (file, MPI, torch.distributed)."""
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    _check(lib.mid_comm_unique_id(buf), "mid_comm_unique_id")
    return bytes(buf)


def comm_create_all(ctxs):
    """One process, one rank per context (mid_comm_create_all = ncclCommInitAll): the list of Comm objects, rank i on ctxs[i]."""
    world = len(ctxs)
    tbl = (ctypes.c_void_p * world)(*[c.handle for c in ctxs])
    out = (ctypes.c_void_p * world)()
    _check(lib.mid_comm_create_all(tbl, world, out), "mid_comm_create_all")
    return [Comm(ctxs[i], None, i, world, _handle=ctypes.c_void_p(out[i])) for i in range(world)]


class Comm:
    """One rank of an RCCL communicator bound to a Context (mid_comm_create): one process per GPU."""

    def __init__(self, ctx, unique_id, rank, world, _handle=None):
        self.ctx, self.rank, self.world = ctx, rank, world
        if _handle is not None:                      # made by comm_create_all
            self.handle = _handle
            return
        h = ctypes.c_void_p()
        idb = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _check(lib.mid_comm_create(ctx.handle, idb, rank, world, ctypes.byref(h)), "mid_comm_create")
        self.handle = h

    def close(self):
        if self.handle:
            lib.mid_comm_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def abort(self):
        """ncclCommAbort: the way out of a collective a rank could not enter; the handle then only accepts close()."""
        _check(lib.mid_comm_abort(self.handle), "mid_comm_abort")

    def reserve(self, max_frame_bytes, k):
        """Allocate the 2k halo receive buffers up front (mid_comm_reserve)."""
        _check(lib.mid_comm_reserve(self.handle, int(max_frame_bytes), int(k)), "mid_comm_reserve")

    def loopback(self, src_ptr, dst_ptr, nbytes, stream=None):
        _check(lib.mid_comm_loopback(self.handle, src_ptr, dst_ptr, nbytes, stream), "mid_comm_loopback")

    def last_loopback(self):
        """(start ms, end ms) of the last loopback's send+receive group on the exchange stream, from the moment the caller's
        stream reached the call (mid_comm_last_loopback; waits for it)."""
        t = (ctypes.c_float * 2)()
        _check(lib.mid_comm_last_loopback(self.handle, t), "mid_comm_last_loopback")
        return float(t[0]), float(t[1])

    def nlm_temporal_sharded_dev(self, block_ptrs, out_ptrs, w, h, n_frames, k, hparam, search, patch, fmt, stream=None):
        """This rank's block (device pointers, in order) -> its outputs; halo over RCCL, interior launches meanwhile."""
        _, count = shard_block(n_frames, self.world, self.rank)
        if len(block_ptrs) != count or len(out_ptrs) != count:
            raise ValueError(f"rank {self.rank} owns {count} frames, got {len(block_ptrs)} inputs / {len(out_ptrs)} outputs")
        p = NlmParams(w, h, hparam, search[0], search[1], patch[0], patch[1], fmt)
        fr = (ctypes.c_void_p * max(count, 1))(*block_ptrs)
        ou = (ctypes.c_void_p * max(count, 1))(*out_ptrs)
        _check(lib.mid_nlm_temporal_sharded(self.handle, ctypes.byref(p), fr, n_frames, k, ou, stream), "mid_nlm_temporal_sharded")

    def last_exchange(self):
        """(bytes received, bytes sent, exchange ms) of the last sharded call; waits for that exchange."""
        a, b, ms = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_float()
        _check(lib.mid_comm_last_exchange(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(ms)), "mid_comm_last_exchange")
        return a.value, b.value, ms.value

    def last_timeline(self):
        """Device timeline of the last sharded call (mid_comm_last_timeline; waits for the call), ms from its first event:
        dict(exchange_start, exchange_end, interior_end, end, halo_hidden_frac).  halo_hidden_frac is the share of the
        exchange that ran while interior launches were still executing (None when nothing was exchanged)."""
        t = (ctypes.c_float * 4)()
        _check(lib.mid_comm_last_timeline(self.handle, t), "mid_comm_last_timeline")
        x0, x1, i1, end = (float(v) for v in t)
        hidden = None
        if x1 > x0:
            hidden = max(0.0, min(1.0, (min(x1, i1) - x0) / (x1 - x0)))
        return {"exchange_start_ms": x0, "exchange_end_ms": x1, "interior_end_ms": i1, "end_ms": end, "halo_hidden_frac": hidden}

    def last_issue_order(self):
        """'X' exchange group, 'I' interior launch, 'W' wait for the exchange, 'B' boundary launch -- in host issue order."""
        buf = ctypes.create_string_buffer(64)
        _check(lib.mid_comm_last_issue_order(self.handle, buf, 64), "mid_comm_last_issue_order")
        return buf.value.decode()

    def stream_priority(self):
        """(priority of the exchange stream, least, greatest) -- numerically lower is higher."""
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib.mid_comm_stream_priority(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "mid_comm_stream_priority")
        return a.value, b.value, c.value

    def boundary_priority(self):
        """Priorities of the two boundary streams (the device's lowest: a pool of hardware queues of their own)."""
        pr = (ctypes.c_int * 2)()
        _check(lib.mid_comm_boundary_priority(self.handle, pr), "mid_comm_boundary_priority")
        return pr[0], pr[1]

    def rccl_info(self):
        """(ncclCommCount, ncclCommUserRank, ncclGetVersion) as RCCL reports them; -1 where the loaded library lacks the call."""
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib.mid_comm_rccl_info(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "mid_comm_rccl_info")
        return a.value, b.value, c.value
