#!/usr/bin/env python3
"""The global-map gather of tools/bench_photon.py under another build of the library:  python tools/ab_gather.py LIB.so [...]"""
import subprocess
import sys

CHILD = r"""
import importlib, os, runpy, sys
sys.path.insert(0, os.getcwd())
importlib.import_module("go-raytracer_amd._capi").LIB_PATH = sys.argv[1]
sys.argv = ["bench_photon.py", "--cpu-queries", "0", "--rounds", "7", "--render-spp", "0", "--only", "global"] + sys.argv[2:]
runpy.run_path("tools/bench_photon.py", run_name="__main__")
"""
import json
extra = [a for a in sys.argv[1:] if a.startswith("--") or a.isdigit()]
for lib in [a for a in sys.argv[1:] if a not in extra]:
    r = subprocess.run([sys.executable, "-c", CHILD, lib] + extra, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-2000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(lib, {k: round(v["ms"], 4) for k, v in d["gather"].items()}, flush=True)
