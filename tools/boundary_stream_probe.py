"""One rank's launches of mid_nlm_temporal_sharded at N = 8 (BASELINE configs[4]: 8 frames per rank, k = 2): 4 interior outputs in
one launch (9.03 rounds of workgroups), then the two boundary launches of 2 outputs each (4.5 rounds each).  On ONE stream the three
launches pay three tails (10 + 5 + 5 rounds for 18.06 rounds of work); with the boundary launches on a SECOND stream their workgroups
fill the interior launch's tail.  Times the two arrangements on resident 1080p frames (LABNOTES R5.5).
Round 6 adds the second stream at the device's LOWEST priority (its own pool of hardware queues in the runtime, so that it can never
share an in-order queue with the caller's streams; boundary workgroups are dispatched only where no interior workgroup waits): LABNOTES R6.3."""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
import image_denoising_filter_amd as mid
torch.cuda.set_device(0); dev = torch.device("cuda", 0); ctx = mid.Context(0)
fr = bench.synth_frames(12, 100, dev)
outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(8)]
fp, op = [f.data_ptr() for f in fr], [o.data_ptr() for o in outs]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])          # the copy the process has mapped already (torch's)
hip = hip_runtime()
lo, hi = ctypes.c_int(), ctypes.c_int()
assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
def raw_stream(prio):
    raw = ctypes.c_void_p()
    assert hip.hipStreamCreateWithPriority(ctypes.byref(raw), 1, prio) == 0                     # 1 = hipStreamNonBlocking
    return torch.cuda.ExternalStream(raw.value)
sLow, sLow2 = raw_stream(lo.value), raw_stream(lo.value)
sB2 = torch.cuda.Stream()
print("stream priority range: least %d, greatest %d; boundary stream variants: default (0) and least" % (lo.value, hi.value), flush=True)
W, H, S, P = bench.W, bench.H, (-10, 11), (-3, 4)
def interior(s): ctx.nlm_temporal_dev(fp[2:10], op[2:6], W, H, 0.5, S, P, 2, 2, 4, 0, s)       # outputs 4..7 of the 12: windows of 5
def edge_lo(s):  ctx.nlm_temporal_dev(fp[0:6], op[0:2], W, H, 0.5, S, P, 2, 2, 2, 0, s)
def edge_hi(s):  ctx.nlm_temporal_dev(fp[6:12], op[6:8], W, H, 0.5, S, P, 2, 2, 2, 0, s)
def one_stream():
    interior(sA.cuda_stream); edge_lo(sA.cuda_stream); edge_hi(sA.cuda_stream)
def two_streams(sB=sB):
    e = torch.cuda.Event(); e.record(sA); sB.wait_event(e)
    interior(sA.cuda_stream); edge_lo(sB.cuda_stream); edge_hi(sB.cuda_stream)
    e2 = torch.cuda.Event(); e2.record(sB); sA.wait_event(e2)
def two_streams_low(): two_streams(sLow)
def three_streams(s1=sB, s2=sB2):                      # each boundary launch on a stream of its own
    e = torch.cuda.Event(); e.record(sA); s1.wait_event(e); s2.wait_event(e)
    interior(sA.cuda_stream); edge_lo(s1.cuda_stream); edge_hi(s2.cuda_stream)
    for s_ in (s1, s2):
        e2 = torch.cuda.Event(); e2.record(s_); sA.wait_event(e2)
def three_streams_low(): three_streams(sLow, sLow2)
def clock(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(sA)
    for _ in range(n): fn()
    b.record(sA); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rep in range(3):
    print("one stream %.3f ms | boundary launches on a second stream %.3f ms | on a second stream of the LOWEST priority %.3f ms | "
          "each boundary launch on its own stream %.3f ms | ... both of the LOWEST priority %.3f ms | one launch of 8 outputs %.3f ms" % (
        clock(one_stream), clock(two_streams), clock(two_streams_low), clock(three_streams), clock(three_streams_low), clock(lambda: ctx.nlm_temporal_dev(fp, op, W, H, 0.5, S, P, 2, 2, 8, 0, sA.cuda_stream))), flush=True)
ref = [o.clone() for o in outs]; two_streams(); torch.cuda.synchronize()
one_stream(); torch.cuda.synchronize()
print("outputs equal between the arrangements:", all(torch.equal(a, b) for a, b in zip(ref, outs)))
for fn in (two_streams_low, three_streams, three_streams_low):
    for o in outs: o.zero_()
    fn(); torch.cuda.synchronize()
    print("... and", fn.__name__, all(torch.equal(a, b) for a, b in zip(ref, outs)))
