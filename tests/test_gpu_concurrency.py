"""GPU suite: the C-ABI under concurrent use and on caller-owned streams.

The reference records one command buffer per submit and fences it (src/main.cpp:1092); a drop-in
library is also called from thread pools and on streams it does not own, so these tests pin the
two things that differ from the reference's single-queue world: contexts are independent under
threads, and a non-zero stream handle orders the work after what the caller enqueued on it.
"""
import threading

import numpy as np
import pytest

import image_denoising_filter_amd as mid
from conftest import synth_hdr

pytestmark = pytest.mark.gpu


def _work(c, frames):
    out = [c.bilateral(frames[0], 8, 3.0, 0.2), c.bilateral(frames[1], 5, 2.0, 0.1, layout="linear")]
    out += c.nlm_temporal(frames, k=1, hparam=0.4, search=(-10, 11), patch=(-3, 4))
    out += c.nlm_temporal(frames, k=0, hparam=0.5, search=(-7, 7), patch=(-3, 3))
    return out


def test_two_contexts_on_two_threads_match_serial_results(ctx):
    rng = np.random.default_rng(77)
    frames = [synth_hdr(rng, 96, 150) for _ in range(3)]
    want = _work(ctx, frames)
    results, errors = {}, []

    def run(i):
        try:
            with mid.Context(0) as c:
                for _ in range(3):
                    results[i] = _work(c, frames)
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append(e)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for got, ref in zip(results[i], want):
            assert np.array_equal(got, ref)


def test_one_context_shared_by_threads_is_serialised_not_corrupted(ctx):
    rng = np.random.default_rng(78)
    frames = [synth_hdr(rng, 64, 130) for _ in range(2)]
    want = _work(ctx, frames)
    results, errors = {}, []

    def run(i):
        try:
            results[i] = _work(ctx, frames)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(4):
        for got, ref in zip(results[i], want):
            assert np.array_equal(got, ref)


def test_caller_stream_orders_after_the_callers_own_work(ctx):
    """Inputs are produced by torch kernels on a side stream and consumed without a host sync."""
    import torch
    rng = np.random.default_rng(79)
    h, w = 270, 480
    frames = [synth_hdr(rng, h, w) for _ in range(3)]
    want_b = ctx.bilateral(frames[0], 8, 3.0, 0.2)
    want_n = ctx.nlm_temporal(frames, k=1, hparam=0.4, search=(-10, 11), patch=(-3, 4))

    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    host = [torch.from_numpy(f).pin_memory() for f in frames]
    with torch.cuda.stream(side):
        # a long-ish producer chain: garbage first, the real frame last, all on `side`
        d = [torch.full((h, w, 4), 123.0, device=dev) for _ in frames]
        big = torch.randn(4096, 4096, device=dev)
        for _ in range(8):
            big = big @ big * 1e-4
        for t, src in zip(d, host):
            t.copy_(src, non_blocking=True)
        out_b = torch.empty((h, w, 4), device=dev)
        out_n = [torch.empty((h, w, 4), device=dev) for _ in frames]
        s = side.cuda_stream
        assert s != 0
        ctx.bilateral_dev(d[0].data_ptr(), out_b.data_ptr(), w, h, 8, 3.0, 0.2, mid.LAYOUT_TEXTURE, mid.FMT_RGBA32F, stream=s)
        ctx.nlm_temporal_dev([t.data_ptr() for t in d], [t.data_ptr() for t in out_n], w, h, 0.4, (-10, 11), (-3, 4),
                             1, 0, len(frames), mid.FMT_RGBA32F, stream=s)
        got_b = out_b.to("cpu", non_blocking=True)
        got_n = [t.to("cpu", non_blocking=True) for t in out_n]
    side.synchronize()
    assert float(big.abs().sum().isfinite())
    assert np.array_equal(got_b.numpy(), want_b)
    for g, r in zip(got_n, want_n):
        assert np.array_equal(g.numpy(), r)


def test_no_device_memory_is_leaked_by_the_entry_points(ctx):
    """The library owns no caller-visible state beyond a context.  The frame pipeline keeps its device ring, output slots and
    events IN the context between calls (round 4: a steady stream of sequences allocates nothing); that cache is bounded by
    the largest call so far, is returned by mid_ctx_release_cached, and goes with the context.  Everything else is back
    where it started after many pipeline / operator calls and the release of the caller's buffers (allocator caches of
    torch are not involved: the NumPy-level operators use mid_alloc / mid_free)."""
    import torch
    rng = np.random.default_rng(81)
    frames = [synth_hdr(rng, 120, 200) * 0.3 for _ in range(5)]

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info(0)[0]

    def mixed_calls():
        for i in range(25):
            ctx.sequence_nlm(frames, k=i % 3, overlap=bool(i % 2), out_u8=False)
            ctx.nlm_multiframe(frames[0], frames[:3])
            ctx.bilateral(frames[0], 8, 2.0, 0.2)
            ctx.bilateral_layers(frames[0], [np.zeros((120, 200, 4), np.uint8)] * 2, 4)
            with mid.Context(0) as c2:
                c2.nlm_temporal(frames[:2], k=1)

    ctx.release_cached()
    ctx.bilateral(frames[0], 8, 2.0, 0.2)                  # code objects, streams
    empty = free_bytes()
    mixed_calls()                                          # fills the cache to its largest shape (k = 2: 8 ring frames + 4 outputs + multiframe's 6 buffers)
    filled = free_bytes()
    frame_bytes = 120 * 200 * 16
    assert 0 <= empty - filled < 24 * frame_bytes + (8 << 20), f"cache of {(empty - filled) >> 10} KiB for {frame_bytes >> 10} KiB frames"
    mixed_calls()                                          # the same calls again allocate nothing more
    assert abs(free_bytes() - filled) < 2 << 20
    ctx.release_cached()
    assert empty - free_bytes() < 8 << 20, f"{(empty - free_bytes()) >> 20} MiB of device memory not returned by mid_ctx_release_cached"
    # a context of its own at 1080p: about 400 MB of ring + outputs while it lives, nothing afterwards
    big = [np.zeros((1080, 1920, 4), np.float32) for _ in range(6)]
    base = free_bytes()
    with mid.Context(0) as c3:
        c3.sequence_nlm(big, k=2, search=(-2, 3), patch=(-1, 2))
        held = base - free_bytes()
        assert held > 8 * 1080 * 1920 * 16, "the ring stays resident between calls"
        c3.sequence_nlm(big, k=2, search=(-2, 3), patch=(-1, 2))
        assert abs((base - free_bytes()) - held) < 8 << 20
    assert base - free_bytes() < 8 << 20, f"{(base - free_bytes()) >> 20} MiB not returned with the context"
