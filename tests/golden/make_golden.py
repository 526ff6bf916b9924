#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_*.npz from the float64 C restatement (oracle/), CPU only:
    python tests/golden/make_golden.py
The vectors pin the ORACLE (a change of its arithmetic or of the shared counter RNG shows up as a diff of these
files) and give the GPU tests expected outputs that do not depend on running the oracle on the GPU box.  They
are not outputs of the Go reference: no Go toolchain exists here (tests/golden/README.md)."""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-raytracer_amd")
from oracle import binding as ob  # noqa: E402

W, H = 96, 54


def main():
    for variant in ("cornell8", "cornell8d"):
        s = ob.OracleScene(pkg.scenes.cornell_flat(W, H, variant))
        out = {}
        for jitter in (0, 1):
            ids, t, n = s.PrimaryHits(jitter, 3)
            out[f"ids_j{jitter}"], out[f"t_j{jitter}"], out[f"n_j{jitter}"] = ids, t, n
        rs = np.random.RandomState(2)
        xy = np.stack([rs.randint(0, W, 160), rs.randint(0, H, 160)], 1).astype(np.int32)
        rgb, nr = s.TraceSamples(xy, 6, seed=9)
        out["xy"], out["sample_rgb"], out["sample_rays"] = xy, rgb, nr
        out["sum_8spp"] = s.RenderAccum(8, seed=4)
        np.savez_compressed(os.path.join(HERE, f"oracle_{variant}.npz"), **out)
        print(variant, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
