"""A/B of two builds of the kernel library on ONE box (measurement tool): bench.py's timed region with each library in turn,
alternating, one process per run.
    python tools/ab_lib.py [--rounds 3] tools/_build/libvqa_hip_other.so        # against the product library
Build the other library by compiling a saved copy of a source file (see DESIGN 6b-3 for the runs that used this)."""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib, extra):
    sys.path.insert(0, REPO)
    import importlib
    pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
    if lib != "product":
        pkg._lib.LIB_PATH = lib
        pkg._lib._lib = None
    import bench
    real = os.dup(1)
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)              # bench.py's own JSON line is not wanted here
    r = bench.main(["--no-cpu-baseline", "--no-extras"] + extra)
    os.write(real, (json.dumps({"ms": r["ms_per_step"], "frac": r["roofline"]["frac"], "dom_us": r["roofline"].get("avg_launch_us")}) + "\n").encode())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("libs", nargs="*")
    args, extra = ap.parse_known_args()
    if args.child:
        child(args.child, extra)
        sys.exit(0)
    libs = ["product"] + [os.path.abspath(l) for l in args.libs]
    res = {l: [] for l in libs}
    for _ in range(args.rounds):
        for l in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l] + extra, capture_output=True, text=True)
            line = [x for x in out.stdout.splitlines() if x.startswith('{"ms"')]
            if not line:
                print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
            res[l].append(json.loads(line[-1]))
            print(os.path.basename(l), res[l][-1], flush=True)
    for l in libs:
        ms = sorted(r["ms"] for r in res[l])
        print(f"{os.path.basename(l):40s} ms/step median {ms[len(ms)//2]:.3f}  min {ms[0]:.3f}  max {ms[-1]:.3f}")
