"""tools/ab_bil.py for a list of library builds (fresh process each), warm line only:  python tools/ab_bil_libs.py [lib.so ...]"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib:
        env["MID_LIB_PATH"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, os.path.join(here, "ab_bil.py"), "x"], env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("variant")]
    print(f"{os.path.basename(lib) or 'shipped':26s} {lines[-1][10:] if lines else r.stderr[-400:]}", flush=True)
