"""GPU suite: host <-> device copies through the C-ABI (csrc/hostcopy.cpp).

The library classifies every host pointer it is given: page-locked memory (mid_alloc_host, mid_host_register,
mid_image_load_pinned) is DMA'd in place; anything else is moved through the context's own page-locked bounce buffers in
8 MiB chunks, so the HIP runtime's pin-on-the-fly path for large pageable copies is never entered (LABNOTES R5.1).  The
reference only ever copies from / to memory it mapped for the device (src/main.cpp:1105-1142, :91-123); these tests hold
the replacement to "whatever memory the caller passes, the bytes arrive", at the sizes the product uses: 1080p RGBA32F
frames (33,177,600 B), WeightInfo buffers (66 MB), sizes that are not a multiple of the chunk, unaligned addresses,
several threads on one context, and the frame pipeline with pageable frames on both sides."""
import ctypes
import threading

import numpy as np
import pytest

import image_denoising_filter_amd as mid
from conftest import synth_hdr

pytestmark = pytest.mark.gpu
CHUNK = 8 << 20


def _roundtrip(ctx, src_ptr, dst_ptr, nbytes):
    dev = ctx.alloc(max(nbytes, 16))
    try:
        assert mid.lib.mid_memcpy_h2d(ctx.handle, dev.ptr, src_ptr, nbytes, None) == 0, mid.lib.mid_last_error()
        assert mid.lib.mid_memcpy_d2h(ctx.handle, dst_ptr, dev.ptr, nbytes, None) == 0, mid.lib.mid_last_error()
        ctx.sync()
    finally:
        dev.free()


@pytest.mark.parametrize("nbytes", [1, 4095, (1 << 20) + 1, CHUNK - 1, CHUNK, CHUNK + 1, 2 * CHUNK, 33_177_600, 66_355_200 + 12345])
@pytest.mark.parametrize("misalign", [0, 3])
def test_pageable_memory_round_trips_bit_exact(ctx, nbytes, misalign):
    rng = np.random.default_rng(nbytes % 9973 + misalign)
    raw = rng.integers(0, 256, nbytes + misalign + 64, dtype=np.uint8)
    back = np.full(nbytes + misalign + 64, 0xA5, np.uint8)
    src, dst = raw[misalign:misalign + nbytes], back[misalign:misalign + nbytes]
    _roundtrip(ctx, src.ctypes.data, dst.ctypes.data, nbytes)
    assert np.array_equal(src, dst)
    assert np.all(back[:misalign] == 0xA5) and np.all(back[misalign + nbytes:] == 0xA5)      # nothing written past either end


def test_pinned_registered_and_straddling_sources_all_arrive(ctx):
    """mid_alloc_host memory and memory registered in place are recognised as pinned (one DMA, no bounce); a range that
    starts in a registered region and ends beyond it is treated as pageable -- and every variant delivers the same bytes."""
    n = 33_177_600
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, n, dtype=np.uint8)
    # (a) mid_alloc_host on both sides
    pin = mid.PinnedFrames(ctx, 2, n)
    ctypes.memmove(pin.ptrs[0], data.ctypes.data, n)
    _roundtrip(ctx, pin.ptrs[0], pin.ptrs[1], n)
    assert np.array_equal(pin.array(1, (n,), np.uint8), data)
    pin.free()
    # (b) registered in place, (c) a range that runs past the registered part
    big = np.empty(2 * n, np.uint8)
    big[:n] = data
    big[n:] = data[::-1]
    assert mid.lib.mid_host_register(ctx.handle, big.ctypes.data, n) == 0
    try:
        out = np.empty(n, np.uint8)
        _roundtrip(ctx, big.ctypes.data, out.ctypes.data, n)                       # registered source, pageable destination
        assert np.array_equal(out, data)
        out2 = np.empty(n, np.uint8)
        _roundtrip(ctx, big.ctypes.data + n // 2, out2.ctypes.data, n)             # half registered, half not
        assert np.array_equal(out2, big[n // 2:n // 2 + n])
    finally:
        assert mid.lib.mid_host_unregister(ctx.handle, big.ctypes.data) == 0
    # after unregistering, the same addresses are pageable again and still work
    out3 = np.empty(n, np.uint8)
    _roundtrip(ctx, big.ctypes.data, out3.ctypes.data, n)
    assert np.array_equal(out3, data)


def test_a_device_pointer_in_the_host_slot_is_refused_not_dereferenced(ctx):
    """The bounce path reads / writes the "host" buffer with the CPU; handed a device pointer by mistake it must say so."""
    a, b = ctx.alloc(1 << 20), ctx.alloc(1 << 20)
    try:
        assert mid.lib.mid_memcpy_h2d(ctx.handle, a.ptr, b.ptr, 1 << 20, None) != 0
        assert b"device memory" in mid.lib.mid_last_error()
        assert mid.lib.mid_memcpy_d2h(ctx.handle, b.ptr, a.ptr, 1 << 20, None) != 0
        assert mid.lib.mid_memcpy_h2d(ctx.handle, a.ptr, None, 16, None) != 0                 # NULL stays an error as well
    finally:
        a.free(); b.free()


def test_freshly_mapped_and_freed_frames_one_after_another(ctx):
    """The allocation pattern of the run that aborted in round 4 (LABNOTES R5.1): a 33 MB NumPy array is mmap'd, copied to the
    device, freed (munmap), and the next one lands on the same addresses.  With the bounce buffers the runtime never maps
    those pages for the device, so nothing of one array's life can leak into the next; the test pins the bytes."""
    h, w = 1080, 1920
    addrs = set()
    for i in range(12):
        a = np.full((h, w, 4), np.float32(i + 0.5), np.float32)
        addrs.add(a.ctypes.data)
        buf = ctx.upload(a)
        del a
        got = ctx.download(buf, (h, w, 4), np.float32)
        buf.free()
        assert got[0, 0, 0] == np.float32(i + 0.5) and got[-1, -1, -1] == np.float32(i + 0.5) and np.all(got == got[0, 0, 0])
    # (glibc hands the munmap'd range straight back, so the 12 arrays usually share one or two addresses -- the pattern of the
    # aborted run -- but nothing guarantees it, so it is not asserted)


def test_threads_share_one_contexts_bounce_buffers(ctx):
    """The bounce sets are per context and serialised per direction: four threads copying different pageable frames through ONE
    context (each on its own stream-ordered buffer) all get their own bytes back."""
    n = 20_000_003
    errs = []

    def work(seed):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(3):
                data = rng.integers(0, 256, n, dtype=np.uint8)
                out = np.empty(n, np.uint8)
                _roundtrip(ctx, data.ctypes.data, out.ctypes.data, n)
                if not np.array_equal(out, data):
                    errs.append(f"thread {seed}: bytes differ")
        except Exception as e:                      # noqa: BLE001
            errs.append(f"thread {seed}: {e!r}")

    ts = [threading.Thread(target=work, args=(s,)) for s in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errs and not any(t.is_alive() for t in ts), errs


@pytest.mark.parametrize("k,n", [(2, 9), (0, 3), (1, 6)])
@pytest.mark.parametrize("overlap", [True, False])
def test_pipeline_with_pageable_frames_on_both_sides_equals_pinned(ctx, k, n, overlap):
    """mid_sequence_nlm with ordinary arrays as sources AND destinations (inputs bounced on the upload stream, outputs copied
    out one iteration behind the launch) == the same call on mid_alloc_host buffers, bit for bit; 270p frames of 2 MB, so
    every copy is beyond the 1 MiB from which the runtime would have pinned on the fly."""
    rng = np.random.default_rng(100 + k)
    frames = [synth_hdr(rng, 270, 480) * 0.3 for _ in range(n)]
    want, _ = ctx.sequence_nlm(frames, k=k, overlap=overlap)
    got, t = ctx.sequence_nlm(frames, k=k, overlap=overlap, pinned=False, pinned_out=False)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert t[0] > 0 and t[1] > 0 and t[2] > 0
    got8, _ = ctx.sequence_nlm(frames, k=k, overlap=overlap, pinned=False, pinned_out=False, out_u8=True)
    want8, _ = ctx.sequence_nlm(frames, k=k, overlap=overlap, out_u8=True)
    assert all(np.array_equal(a, b) for a, b in zip(got8, want8))
    mixed, _ = ctx.sequence_nlm(frames, k=k, overlap=overlap, pinned=True, pinned_out=False, first=1, count=n - 1)
    assert all(np.array_equal(a, b) for a, b in zip(mixed, want[1:]))


def test_pipeline_1080p_pageable_frames(ctx):
    """The same at the product's frame size (33 MB per copy, 4 chunks + a remainder each way)."""
    rng = np.random.default_rng(3)
    base = synth_hdr(rng, 1080, 1920) * 0.3
    frames = [np.roll(base, 2 * i, axis=1).copy() for i in range(4)]
    want, _ = ctx.sequence_nlm(frames, k=1)
    got, _ = ctx.sequence_nlm(frames, k=1, pinned=False, pinned_out=False)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    ref, _ = ctx.nlm_multiframe(frames[0], frames)                                   # pageable target, frames and result
    W = np.zeros((1080, 1920, 8), np.float32)
    for f in frames:
        W = ctx.nlm_accum(frames[0], f, W, 0.5, (-7, 7), (-3, 3))
    assert np.array_equal(ref, ctx.normalize(W))


def test_two_threads_run_the_pipeline_on_one_context_while_a_third_releases_its_cache(ctx):
    """the per-context pipeline cache is guarded by one lock (calls on a context are serialised) and
    mid_ctx_release_cached may arrive from another thread at any time.  Outputs must not depend on the interleaving."""
    rng = np.random.default_rng(8)
    seqs = [[synth_hdr(rng, 96, 160) * 0.3 for _ in range(7)], [synth_hdr(rng, 120, 200) * 0.3 for _ in range(5)]]
    want = [ctx.sequence_nlm(s, k=2)[0] for s in seqs]
    errs, stop = [], threading.Event()

    def run(i):
        try:
            for rep in range(6):
                got, _ = ctx.sequence_nlm(seqs[i], k=2, pinned=bool(rep & 1), pinned_out=bool(rep & 2))
                if not all(np.array_equal(a, b) for a, b in zip(got, want[i])):
                    errs.append(f"thread {i} rep {rep}: outputs differ")
        except Exception as e:                      # noqa: BLE001
            errs.append(f"thread {i}: {e!r}")

    def release():
        while not stop.is_set():
            try:
                ctx.release_cached()
            except Exception as e:                  # noqa: BLE001
                errs.append(f"release: {e!r}")
                return
            stop.wait(0.002)

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    r = threading.Thread(target=release)
    for t in ts + [r]:
        t.start()
    for t in ts:
        t.join(300)
    stop.set()
    r.join(60)
    assert not errs and not any(t.is_alive() for t in ts + [r]), errs


def test_an_error_on_the_pipelines_helper_thread_ends_the_call_with_that_error_not_a_hang(ctx):
    """Pageable outputs are copied out by a helper thread of the call.  One of seven output pointers is a DEVICE pointer: the
    helper's copy refuses it, the call returns that error on the calling thread (not a hang, not a crash), and the context
    runs the next call as if nothing had happened."""
    rng = np.random.default_rng(4)
    frames = [synth_hdr(rng, 270, 480) * 0.3 for _ in range(7)]
    want, _ = ctx.sequence_nlm(frames, k=1)
    outs = [np.empty((270, 480, 4), np.float32) for _ in range(7)]
    dev = ctx.alloc(270 * 480 * 16)
    ptrs = [o.ctypes.data for o in outs]
    ptrs[4] = dev.ptr
    with pytest.raises(mid.MidError) as e:
        ctx.sequence_nlm_pinned([f.ctypes.data for f in frames], ptrs, 480, 270, mid.FMT_RGBA32F, k=1)
    assert "device memory" in str(e.value)
    dev.free()
    assert all(np.array_equal(outs[i], want[i]) for i in range(4))           # the frames before the bad one were delivered
    got, _ = ctx.sequence_nlm(frames, k=1, pinned=False, pinned_out=False)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
