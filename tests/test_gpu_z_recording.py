"""GPU suite: kernel-level calls RECORDED once and submitted many times (mid_record_begin / mid_record_end / mid_recording_submit).

The reference works on recorded command buffers: vkBeginCommandBuffer ... vkEndCommandBuffer inside every RecordCommandsOf*
(src/main.cpp:791/846, 855/886, 895/988), submitted by RunCommandBuffer (:1078-1103).  The MI355X counterpart is a captured
hipGraph (csrc/recording.cpp): every kernel-level entry of include/mi_denoise.h only enqueues on the stream it is given -- no
allocation, no host-side wait, no other stream -- so it can be recorded.  Held here, for the dispatch sequences of the reference's
modes: a submitted recording writes the bits the same calls write when issued on a stream, also after the inputs' CONTENT changed
(same buffers); on the context's own stream (NULL) and on a caller's stream; through the library's recording API and inside a
capture the CALLER owns (torch.cuda.CUDAGraph = hipStreamBeginCapture in global mode: any hipMalloc / synchronisation inside would
fail it); calls that cannot be recorded are refused with a message and leave the recording valid; a C++ host records COLD (every
kernel's first launch inside the recording) and gets the same bytes.
(The file name sorts last among the GPU suites on purpose: the driver runs them with -x, and graph capture is the one place where this
round met a runtime misbehaviour that comes and goes with the memory layout -- LABNOTES R6.10.)
"""
import ctypes

import numpy as np
import pytest

from conftest import synth_hdr, synth_ldr

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _t(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


class _Rec:
    """Runs `body(stream_handle)` eagerly (reference bits), then recorded + submitted.  how = "mid": the library's recording API on a
    created stream; "mid_null": the same on the context's own compute stream (stream = NULL); "torch": a capture the caller owns."""

    def __init__(self, torch, ctx, how):
        self.torch, self.ctx, self.how = torch, ctx, how
        self.s = torch.cuda.Stream()
        self.handle = None if how == "mid_null" else self.s.cuda_stream

    def eager(self, body):
        self.torch.cuda.synchronize()
        body(self.handle)
        self.ctx.sync(self.handle)

    def capture(self, body):
        self.torch.cuda.synchronize()
        if self.how == "torch":
            g = self.torch.cuda.CUDAGraph()
            with self.torch.cuda.graph(g, stream=self.s):
                body(self.handle)
            return g.replay
        with self.ctx.record(self.handle) as rec:
            body(self.handle)
        self.rec = rec
        n_nodes, n_kernels = rec.info()
        assert n_kernels >= 1 and n_nodes >= n_kernels
        return lambda: rec.submit(self.handle)

    def memset(self, t, st):
        """clear a torch tensor ON the recorded stream (torch's own fill would go to torch's current stream)"""
        import image_denoising_filter_amd as mid
        from image_denoising_filter_amd.api import _check
        _check(mid.lib.mid_memset(self.ctx.handle, t.data_ptr(), 0, t.numel() * t.element_size(), st), "mid_memset")


HOWS = ["mid", "mid_null", "torch"]


@pytest.mark.parametrize("how", HOWS)
def test_ldr_bilateral_sequence_unpack_filter_pack(ctx, how):
    """The PNG path of the single-image modes: UNORM decode -> bialteral.comp -> u8 encode (src/texture.cpp:16, shaders/bialteral.comp,
    GetImageFromGPU src/main.cpp:91-106) as three launches in one graph."""
    torch = _torch()
    import image_denoising_filter_amd as mid
    from image_denoising_filter_amd.api import _check
    lib = mid.lib
    rng = np.random.default_rng(5)
    h, w, R = 270, 333, 8
    a, b = synth_ldr(rng, h, w), synth_ldr(rng, h, w)[::-1].copy()
    u8 = _t(torch, a)
    f32 = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    flt = torch.empty_like(f32)
    out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    p = mid.BilateralParams(w, h, 2.0, 0.2, R, mid.LAYOUT_TEXTURE, mid.FMT_RGBA32F)

    def body(st):
        _check(lib.mid_unpack_u8(ctx.handle, u8.data_ptr(), h * w * 4, 0, f32.data_ptr(), st), "mid_unpack_u8")
        _check(lib.mid_bilateral(ctx.handle, ctypes.byref(p), f32.data_ptr(), flt.data_ptr(), st), "mid_bilateral")
        _check(lib.mid_pack_u8(ctx.handle, flt.data_ptr(), h * w * 4, out.data_ptr(), st), "mid_pack_u8")

    rec = _Rec(torch, ctx, how)
    rec.eager(body)
    want_a = out.cpu().numpy().copy()
    u8.copy_(_t(torch, b)); rec.eager(body)
    want_b = out.cpu().numpy().copy()
    assert not np.array_equal(want_a, want_b)
    g = rec.capture(body)
    for src, want in ((a, want_a), (b, want_b), (a, want_a)):
        u8.copy_(_t(torch, src)); out.zero_()
        torch.cuda.synchronize()
        g()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
    # and the operator the NumPy-level API offers gives those bits too
    assert np.array_equal(ctx.pack_u8(ctx.bilateral(ctx.unpack_u8(a), R, 2.0, 0.2)), want_a)


@pytest.mark.parametrize("how", HOWS)
@pytest.mark.parametrize("window", ["reference", "bench"])
def test_nlm_literal_mode_accumulate_then_normalize(ctx, window, how):
    """The reference's multi-frame mode (loop src/main.cpp:1577-1606 + normalize :1649-1652): clear WeightInfo, one nonlocal.comp
    dispatch per neighbour frame, normalize -- 1 memset + 5 launches + 1 launch recorded once."""
    torch = _torch()
    import image_denoising_filter_amd as mid
    from image_denoising_filter_amd.api import _check
    lib = mid.lib
    rng = np.random.default_rng(6)
    h, w, n = 97, 141, 5
    win = mid.NLM_REFERENCE if window == "reference" else mid.NLM_BENCH
    fr_a = [synth_hdr(rng, h, w, 3.0) for _ in range(n)]
    fr_b = [np.ascontiguousarray(f[:, ::-1]) for f in fr_a]
    frames = [_t(torch, f) for f in fr_a]
    Wb = torch.empty((h, w, 8), dtype=torch.float32, device="cuda")
    out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    p = mid.NlmParams(w, h, 0.5, win["search"][0], win["search"][1], win["patch"][0], win["patch"][1], mid.FMT_RGBA32F)
    pn = mid.NormalizeParams(w, h)

    def body(st):
        rec.memset(Wb, st)                                            # (a memset node)
        for f in frames:
            _check(lib.mid_nlm_accum(ctx.handle, ctypes.byref(p), frames[2].data_ptr(), f.data_ptr(), Wb.data_ptr(), st), "mid_nlm_accum")
        _check(lib.mid_normalize(ctx.handle, ctypes.byref(pn), Wb.data_ptr(), out.data_ptr(), st), "mid_normalize")

    rec = _Rec(torch, ctx, how)
    rec.eager(body)
    want_a = out.cpu().numpy().copy()
    g = rec.capture(body)
    for f, src in zip(frames, fr_b):
        f.copy_(_t(torch, src))
    rec.eager(body)
    want_b = out.cpu().numpy().copy()
    assert not np.array_equal(want_a, want_b)
    for srcs, want in ((fr_a, want_a), (fr_b, want_b)):
        for f, src in zip(frames, srcs):
            f.copy_(_t(torch, src))
        out.zero_()
        torch.cuda.synchronize()
        g()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
    # the fused entry point (one launch) agrees with the recorded sequence as it does with the eager one
    fused = ctx.nlm_temporal(fr_b, k=n, first=2, count=1, hparam=0.5, **win)[0]
    assert np.array_equal(fused, want_b)


@pytest.mark.parametrize("how", HOWS)
def test_fused_temporal_and_batch_launches_replay(ctx, how):
    """The fused launches -- frame tables by value in kernarg space, a small launch's main part + HALF tail -- keep their arguments
    inside the graph: mid_nlm_temporal (k = 1, every frame an output) and mid_bilateral_batch replayed after the content changed."""
    torch = _torch()
    import image_denoising_filter_amd as mid
    rng = np.random.default_rng(7)
    h, w, n = 120, 200, 4
    fr_a = [synth_hdr(rng, h, w, 2.0) for _ in range(n)]
    fr_b = [np.ascontiguousarray(f[::-1]) for f in fr_a]
    frames = [_t(torch, f) for f in fr_a]
    o_nlm = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(n)]
    o_bil = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(n)]

    def body(st):
        ctx.nlm_temporal_dev([f.data_ptr() for f in frames], [o.data_ptr() for o in o_nlm], w, h, 0.5, mid.NLM_BENCH["search"],
                             mid.NLM_BENCH["patch"], 1, 0, n, mid.FMT_RGBA32F, st)
        ctx.bilateral_batch_dev([f.data_ptr() for f in frames], [o.data_ptr() for o in o_bil], w, h, 4, 2.0, 0.2, mid.LAYOUT_LINEAR,
                                mid.FMT_RGBA32F, st)

    rec = _Rec(torch, ctx, how)
    rec.eager(body)                                                   # (also: every kernel's LDS limit is raised outside the capture)
    g = rec.capture(body)
    for srcs in (fr_b, fr_a):
        for f, src in zip(frames, srcs):
            f.copy_(_t(torch, src))
        for o in o_nlm + o_bil:
            o.zero_()
        torch.cuda.synchronize()
        g()
        torch.cuda.synchronize()
        got_n = [o.cpu().numpy() for o in o_nlm]
        got_b = [o.cpu().numpy() for o in o_bil]
        want_n = ctx.nlm_temporal(srcs, k=1, hparam=0.5, **mid.NLM_BENCH)
        want_b = ctx.bilateral_batch(srcs, 4, 2.0, 0.2, layout="linear")
        for a, b in zip(got_n + got_b, want_n + want_b):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("how", ["mid", "torch"])
def test_layer_loop_recorded_equals_fused(ctx, how):
    """The per-layer loop (src/main.cpp:1610-1623: one bialteral_layers.comp dispatch per guide layer) + normalize as one graph, against
    the fused mid_bilateral_layers launch."""
    torch = _torch()
    import image_denoising_filter_amd as mid
    from image_denoising_filter_amd.api import _check
    lib = mid.lib
    rng = np.random.default_rng(8)
    h, w, R, L = 150, 190, 8, 3
    img = synth_hdr(rng, h, w, 2.0)
    layers = [synth_ldr(rng, h, w) for _ in range(L)]
    d_img = _t(torch, img)
    d_l = [_t(torch, l) for l in layers]
    Wb = torch.empty((h, w, 8), dtype=torch.float32, device="cuda")
    out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    p = mid.BilateralParams(w, h, 2.0, 0.2, R, mid.LAYOUT_TEXTURE, mid.FMT_RGBA32F)
    pn = mid.NormalizeParams(w, h)

    def body(st):
        rec.memset(Wb, st)
        for l in d_l:
            _check(lib.mid_bilateral_layers_accum(ctx.handle, ctypes.byref(p), d_img.data_ptr(), l.data_ptr(), Wb.data_ptr(), st),
                   "mid_bilateral_layers_accum")
        _check(lib.mid_normalize(ctx.handle, ctypes.byref(pn), Wb.data_ptr(), out.data_ptr(), st), "mid_normalize")

    rec = _Rec(torch, ctx, how)
    rec.eager(body)
    g = rec.capture(body)
    out.zero_()
    torch.cuda.synchronize()
    g()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ctx.bilateral_layers(img, layers, R, 2.0, 0.2))


def test_calls_that_cannot_be_recorded_are_refused_and_the_recording_survives(ctx):
    """mid_stream_sync, the frame pipeline, timers and pageable copies wait on the host or drive several streams: inside a recording
    they return MID_ERR_INVALID with a message -- before touching the runtime, so the recording stays valid and still submits."""
    torch = _torch()
    import image_denoising_filter_amd as mid
    from image_denoising_filter_amd.api import MidError, _check
    lib = mid.lib
    rng = np.random.default_rng(9)
    h, w = 64, 80
    img = synth_hdr(rng, h, w, 2.0)
    d_in, d_out = _t(torch, img), torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    pageable = np.zeros((h, w, 4), np.float32)
    pinned = mid.PinnedFrames(ctx, [img])
    p = mid.BilateralParams(w, h, 2.0, 0.2, 4, mid.LAYOUT_TEXTURE, mid.FMT_RGBA32F)
    prm = mid.NlmParams(w, h, 0.5, -7, 7, -3, 3, mid.FMT_RGBA32F)
    want = ctx.bilateral(img, 4, 2.0, 0.2)
    s = torch.cuda.Stream()
    st = s.cuda_stream
    torch.cuda.synchronize()
    try:
        with ctx.record(st) as rec:
            _check(lib.mid_memcpy_h2d(ctx.handle, d_in.data_ptr(), pinned.ptrs[0], h * w * 16, st), "h2d pinned")   # recordable
            _check(lib.mid_bilateral(ctx.handle, ctypes.byref(p), d_in.data_ptr(), d_out.data_ptr(), st), "mid_bilateral")
            for what, call in (
                    ("mid_stream_sync", lambda: lib.mid_stream_sync(ctx.handle, st)),
                    ("pageable", lambda: lib.mid_memcpy_d2h(ctx.handle, pageable.ctypes.data, d_out.data_ptr(), h * w * 16, st)),
                    ("pageable", lambda: lib.mid_memcpy_h2d(ctx.handle, d_in.data_ptr(), pageable.ctypes.data, h * w * 16, st))):
                rc = call()
                assert rc == 1 and "cannot be part of a recording" in lib.mid_last_error().decode() and what in lib.mid_last_error().decode()
            with pytest.raises(MidError, match="already recording"):
                ctx.record(st).__enter__()
        assert rec.info() == (2, 1)                                 # the copy and the launch; nothing of the refused calls
        d_in.zero_()
        torch.cuda.synchronize()
        rec.submit(st)
        ctx.sync(st)
        assert np.array_equal(d_out.cpu().numpy(), want)
        # a recording on the context's own stream: the frame pipeline (which drives that stream) refuses, a later call works
        with ctx.record(None) as rec0:
            _check(lib.mid_bilateral(ctx.handle, ctypes.byref(p), d_in.data_ptr(), d_out.data_ptr(), None), "mid_bilateral")
            outp = (ctypes.c_void_p * 1)(pinned.ptrs[0])
            rc = lib.mid_sequence_nlm(ctx.handle, ctypes.byref(prm), outp, 1, 0, outp, 1, None)
            assert rc == 1 and "mid_sequence_nlm" in lib.mid_last_error().decode()
        rec0.submit(None)
        ctx.sync(None)
        assert np.array_equal(d_out.cpu().numpy(), want)
        seq, _ = ctx.sequence_nlm([img], k=0)
        assert np.array_equal(seq[0], ctx.nlm_temporal([img], k=0)[0])
        with pytest.raises(MidError, match="not recording"):
            h_ = ctypes.c_void_p()
            _check(lib.mid_record_end(ctx.handle, st, ctypes.byref(h_)), "mid_record_end")
    finally:
        pinned.free()


def test_cpp_host_records_cold_and_gets_the_bytes_of_the_call_by_call_sequence(ctx, tmp_path):
    """The recording API from compiled host code (the reference's language), with NO warm-up: every kernel's first launch -- and with it
    the library's one-time hipFuncSetAttribute for its LDS limit -- happens inside the recording.  Bytes against the Python-driven calls."""
    import json
    import os
    import subprocess
    from conftest import ROOT
    src = os.path.join(ROOT, "tests", "recording_host.cpp")
    exe = tmp_path / "recording_host"
    libdir = os.path.join(ROOT, "image_denoising_filter_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lmi_denoise", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, timeout=300)
    w, h, n = 150, 100, 5
    r = subprocess.run([str(exe), str(w), str(h), str(n), "20", str(tmp_path / "out.raw")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["same"] and rep["kernels"] == n + 2 and rep["nodes"] == rep["kernels"]      # the clear (a kernel while recording), n accumulates, normalize
    got = np.frombuffer((tmp_path / "out.raw").read_bytes(), np.float32).reshape(h, w, 4)
    frames = []
    for i in range(n):                                               # the host program's LCG frames
        s = np.uint32(12345 + i)
        v = np.empty(w * h * 4, np.float32)
        state = int(s)
        for j in range(v.size):
            state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
            v[j] = 1.0 if (j & 3) == 3 else np.float32(state >> 8) * np.float32(1.0 / 16777216.0)
        frames.append(v.reshape(h, w, 4))
    Wacc = np.zeros((h, w, 8), np.float32)
    for f in frames:
        Wacc = ctx.nlm_accum(frames[n // 2], f, Wacc, 0.5, (-7, 7), (-3, 3))
    assert np.array_equal(got, ctx.normalize(Wacc))
