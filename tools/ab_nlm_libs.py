"""tools/ab_nlm.py 0 for a list of library builds, each in a fresh process, last timing line and checksum only:
   python tools/ab_nlm_libs.py [lib.so ...]     ("" = the shipped library)"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib:
        env["MID_LIB_PATH"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, os.path.join(here, "ab_nlm.py"), "0"], env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("variant")]
    timing = [l for l in lines if "batch8" in l]
    check = [l for l in lines if "check" in l]
    print(f"{os.path.basename(lib) or 'shipped':28s} {timing[-1][10:] if timing else r.stderr[-300:]}  | {check[-1].split('px')[0][10:] if check else ''}", flush=True)
