"""REAL RCCL send/receive kernels beside an interior launch -- ONE device, NO wire.
A 1-rank communicator of the real librccl (mid_comm_create), mid_comm_loopback of 2 x 33 MB (the k = 2 halo of one boundary) on the
communicator's highest-priority exchange stream:
  a.  on an idle device;
  b1. issued BEFORE a 4-output k = 2 interior launch (the issue order of mid_nlm_temporal_sharded: X I), interior on another stream;
  b2. issued right AFTER that interior launch has been queued (the device is already full of NLM workgroups when RCCL's kernel arrives).
Durations from the library's own events on the exchange stream (mid_comm_last_loopback) and torch events on the probe's streams.
   python tools/rccl_loopback_probe.py                     (events)
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/rccl_loopback_probe.py --reps 2     (kernel names, grids, registers, LDS)
   python tools/rccl_loopback_probe.py --summarise DIR     (condenses that trace: no GPU needed)
What this cannot show: the transport between two devices (xGMI), its channel count there, and RCCL's proxy threads."""
import argparse, csv, glob, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--summarise", default=None)
args = ap.parse_args()

if args.summarise:
    tr = sorted(glob.glob(os.path.join(args.summarise, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(tr[-1])))
    t00 = min(int(r["Start_Timestamp"]) for r in rows)
    def short(n):
        n = n.replace("void ", "")
        return (n[:n.index("(")] if "(" in n else n)[:100]
    seen = {}
    for r in rows:
        key = (short(r["Kernel_Name"]), r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r.get("Accum_VGPR_Count", "?"), r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
        seen.setdefault(key, []).append((int(r["Start_Timestamp"]) - t00, int(r["End_Timestamp"]) - t00))
    print("# kernels of the run: name | grid x | workgroup x | VGPR | AGPR | SGPR | LDS B | scratch B | launches | avg ms | min ms | max ms")
    for k, v in sorted(seen.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        d = [(e - s) / 1e6 for s, e in v]
        print(" | ".join(k), "|", len(v), "| %.3f | %.3f | %.3f" % (sum(d) / len(d), min(d), max(d)))
    print("# every launch in start order (ms from the first kernel of the trace): start end name grid")
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        n = short(r["Kernel_Name"])
        if "nlm" in n or "ccl" in n.lower():
            print("%10.3f %10.3f  %s  grid %s" % ((int(r["Start_Timestamp"]) - t00) / 1e6, (int(r["End_Timestamp"]) - t00) / 1e6, n, r["Grid_Size_X"]))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import image_denoising_filter_amd as mid
torch.cuda.set_device(0); dev = torch.device("cuda", 0); ctx = mid.Context(0)
comm = mid.Comm(ctx, mid.comm_unique_id(), 0, 1)
print("RCCL: ncclCommCount %d, rank %d, version %d | exchange stream priority %s (least, greatest = %s)" % (*comm.rccl_info(), comm.stream_priority()[0], comm.stream_priority()[1:]), flush=True)
fr = bench.synth_frames(12, 100, dev)
outs = [torch.empty((bench.H, bench.W, 4), device=dev) for _ in range(4)]
fp, op = [f.data_ptr() for f in fr], [o.data_ptr() for o in outs]
W, H, S, P = bench.W, bench.H, (-10, 11), (-3, 4)
FRAME = W * H * 16
sI, sL = torch.cuda.Stream(), torch.cuda.Stream()
def interior(): ctx.nlm_temporal_dev(fp[2:10], op, W, H, 0.5, S, P, 2, 2, 4, 0, sI.cuda_stream)      # the N = 8 rank's interior launch: 4 outputs, windows of 5
ev = lambda: torch.cuda.Event(enable_timing=True)

for nfr in (2, 4):
    nbytes = nfr * FRAME
    src = torch.rand(nbytes // 4, device=dev); dst = torch.zeros(nbytes // 4, device=dev)
    comm.loopback(src.data_ptr(), dst.data_ptr(), nbytes, sL.cuda_stream); interior(); torch.cuda.synchronize()        # warm both
    assert torch.equal(src, dst)
    print(f"== {nfr} frames = {nbytes / 1e6:.1f} MB sent and received in one group (a boundary of k = {nfr // 2}{'; both boundaries of an inner rank at k = 2' if nfr == 4 else ''})", flush=True)
    for rep in range(args.reps):
        # a. idle device
        comm.loopback(src.data_ptr(), dst.data_ptr(), nbytes, sL.cuda_stream); a0, a1 = comm.last_loopback(); torch.cuda.synchronize()
        # interior alone
        i0, i1 = ev(), ev(); i0.record(sI); interior(); i1.record(sI); torch.cuda.synchronize(); alone = i0.elapsed_time(i1)
        # b1. X then I
        t0, i0, i1 = ev(), ev(), ev()
        t0.record(sL); comm.loopback(src.data_ptr(), dst.data_ptr(), nbytes, sL.cuda_stream)
        i0.record(sI); interior(); i1.record(sI)
        x0, x1 = comm.last_loopback(); torch.cuda.synchronize()
        b1 = (x0, x1, t0.elapsed_time(i0), t0.elapsed_time(i1))
        # b2. I then X
        t0, i0, i1 = ev(), ev(), ev()
        i0.record(sI); interior(); i1.record(sI)
        t0.record(sL); comm.loopback(src.data_ptr(), dst.data_ptr(), nbytes, sL.cuda_stream)
        x0, x1 = comm.last_loopback(); torch.cuda.synchronize()
        b2 = (x0, x1, -i0.elapsed_time(t0), t0.elapsed_time(i1))
        print(f"  rep {rep}: a. idle {a1 - a0:.3f} ms ({2 * nbytes / (a1 - a0) / 1e6:.0f} GB/s read+write) | interior alone {alone:.3f} ms | "
              f"b1. X,I: loopback {b1[0]:.3f}..{b1[1]:.3f} = {b1[1] - b1[0]:.3f} ms, interior {b1[2]:.3f}..{b1[3]:.3f} = {b1[3] - b1[2]:.3f} ms | "
              f"b2. I,X: loopback {b2[0]:.3f}..{b2[1]:.3f} = {b2[1] - b2[0]:.3f} ms, interior {b2[2]:.3f}..{b2[3]:.3f} = {b2[3] - b2[2]:.3f} ms", flush=True)
    assert torch.equal(src, dst)
comm.close()
