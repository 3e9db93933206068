"""Does work on the exchange stream delay the interior launches?  Two ranks (threads) on cuda:0 against the stand-in transport
with its wire slowed down (STANDIN_RCCL_DELAY_MS): per rank the device timeline of one mid_nlm_temporal_sharded call.
   python tools/halo_overlap_probe.py [lib.so ...]      (fresh process per library, MID_LIB_PATH; "" = the shipped one)
Used for LABNOTES R5.3: exchange stream at the highest priority (shipped) vs at the default priority (a build with
-DMID_XS_DEFAULT_PRIORITY; the switch left csrc/sharded.cpp after the measurement, see tools/experiments/README.md)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import json, sys, threading
sys.path.insert(0, sys.argv[1])
import numpy as np
import image_denoising_filter_amd as mid
CFG = dict(search=(-10, 11), patch=(-3, 4))
world, n, k, h, w = 2, int(sys.argv[2]), 2, int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(1)
seq = [(rng.random((h, w, 4), dtype=np.float32) * 0.8).astype(np.float32) for _ in range(n)]
ctxs = [mid.Context(0) for _ in range(world)]
comms = mid.comm_create_all(ctxs)
rep, errs = {}, []
def rank_main(r):
    try:
        c, comm = ctxs[r], comms[r]
        start, count = mid.shard_block(n, world, r)
        d_in = [c.upload(seq[start + i]) for i in range(count)]
        d_out = [c.alloc(h * w * 16) for _ in range(count)]
        comm.reserve(h * w * 16, k)
        for _ in range(2):
            comm.nlm_temporal_sharded_dev([d.ptr for d in d_in], [d.ptr for d in d_out], w, h, n, k, 0.5, CFG["search"], CFG["patch"], mid.FMT_RGBA32F)
            tl = comm.last_timeline()
        rep[r] = dict(tl, order=comm.last_issue_order(), priority=comm.stream_priority())
    except Exception as e:
        errs.append(f"rank {r}: {e!r}")
th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
for t in th: t.start()
for t in th: t.join(timeout=120)
print("PROBE " + json.dumps({"rep": rep, "errs": errs}))
'''
standin = os.path.join(ROOT, "tests", "standin_rccl", "libstandin_rccl.so")
# (wire delay ms per frame, frames, h, w, copy-kernel workgroups [0 = hipMemcpyAsync])
CASES = (("0", 16, 270, 480, 0), ("30", 16, 270, 480, 0), ("10", 16, 1080, 1920, 0),
         # the copy as a KERNEL, like RCCL's send/receive: first with nothing else to do (4 frames on 2 ranks at k = 2: every output is a
         # boundary output, the exchange runs on an idle device), then beside the interior launches of 8 frames per rank
         ("0", 4, 1080, 1920, 8), ("0", 16, 1080, 1920, 8), ("0", 4, 1080, 1920, 32), ("0", 16, 1080, 1920, 32))
for lib in (sys.argv[1:] or [""]):
    for delay, n, h, w, wgs in CASES:
        env = dict(os.environ, MID_RCCL_LIBRARY=standin, STANDIN_RCCL_DELAY_MS=delay)
        if os.environ.get("PROBE_HW_QUEUES"):       # round 5 ran this probe with 24 hardware queues per level; since round 6 the runtime's default (4) is the case of interest
            env["GPU_MAX_HW_QUEUES"] = os.environ["PROBE_HW_QUEUES"]
        if wgs:
            env["STANDIN_RCCL_COPY_WGS"] = str(wgs)
        if lib:
            env["MID_LIB_PATH"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", code, ROOT, str(n), str(h), str(w)], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("PROBE ")]
        if not line:
            print(os.path.basename(lib) or "shipped", "delay", delay, "FAILED", r.stderr[-500:])
            continue
        d = json.loads(line[0][6:])
        for rk, t in sorted(d["rep"].items()):
            print(f"{os.path.basename(lib) or 'shipped':24s} {n:2d} frames {w}x{h} wire delay {delay:>2s} ms/frame copy {('kernel x%d WGs' % wgs) if wgs else 'hipMemcpyAsync':15s} rank {rk}: exchange {t['exchange_start_ms']:.2f}..{t['exchange_end_ms']:.2f}  "
                  f"interior_end {t['interior_end_ms']:.2f}  end {t['end_ms']:.2f}  hidden {t['halo_hidden_frac']}  order {t['order']}  prio {t['priority']}", flush=True)
        if d["errs"]:
            print("   errors:", d["errs"])
