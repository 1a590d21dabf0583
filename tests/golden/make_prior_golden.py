"""Generate tests/golden/prior_tables.npz from the REFERENCE's own engine/prior.py.

This is the only piece of the reference that is importable in the build
container (TensorFlow is absent).  The module is
loaded by file path (never via the `engine` package, whose __init__ imports
tensorflow) with the removed alias `numpy.int` shimmed, exactly as SURVEY.md
records.  Only the OUTPUT TABLES are committed (data, not source).

Run in the build container only (needs /root/reference):
    python tests/golden/make_prior_golden.py
"""
import importlib.util
import os

import numpy as np

import sys

sys.dont_write_bytecode = True      # /root/reference is read-only input: leave no __pycache__ beside the module loaded from it

REF = "/root/reference/engine/prior.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prior_tables.npz")

CASES = {
    # name: (strides, sizes, scales, ratios) -- default config first
    "default": ([8, 16, 32, 64, 128], [32, 64, 128, 256, 512],
                [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)], [1 / 3, 1 / 2, 1, 2, 3]),
    "three_level": ([8, 16, 32], [32, 64, 128],
                    [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)], [1 / 2, 1, 2]),
    "two_scale": ([16, 32, 64, 128], [64, 128, 256, 512],
                  [1.0, 1.5], [1 / 3, 1, 3]),
    "unsorted_strides": ([32, 8, 16], [128, 32, 64], [1.0], [0.5, 1.0, 2.0]),
}


def main():
    np.int = int  # alias removed in numpy >= 1.24; reference uses it at prior.py:60-66
    spec = importlib.util.spec_from_file_location("ref_prior", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for name, (strides, sizes, scales, ratios) in CASES.items():
        pb = mod.PriorBoxes(strides, sizes, scales, ratios)
        out[name + "/table"] = pb.boxes.values.astype(np.int64)
        out[name + "/len"] = np.int64(len(pb))
        out[name + "/strides"] = np.asarray(strides, np.int64)
        out[name + "/sizes"] = np.asarray(sizes, np.int64)
        out[name + "/scales"] = np.asarray(scales, np.float64)
        out[name + "/ratios"] = np.asarray(ratios, np.float64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("table")})


if __name__ == "__main__":
    main()
