#!/usr/bin/env python3
"""Offline converter: Keras .h5 (what the reference trains and saves, `model.save(...)` /
`save_weights(...)`, loaded by engine/retinamasklab.py:498-508) -> the name-keyed .npz that
`masklab_hip.retinamasklab.load_masklab_inference_model_from_weights` reads.

The reference re-wires a loaded Keras model by layer-name regexes (:515-586); here the same names key
the tensors directly: `<layer>/<sub-layer>/.../<weight>` with the TF suffix `:0` dropped.  Keras stores
  /[model_weights/]<top layer>  attrs['weight_names'] = [b'<scope>/<weight>:0', ...]  + one dataset each
so no graph has to be rebuilt: walk the groups, strip the suffix, and check every tensor the target model
declares (name, shape) is present.

Needs `h5py` to open real files (not installed in the build container: the file walking is written
against the small mapping protocol h5py implements -- `.attrs`, `[]`, `in` -- and is tested with an
in-memory stand-in).  Usage:
    python tools/convert_keras_h5.py weights.h5 weights.npz [--backbone resnext50]
"""
import argparse
import os
import sys

import numpy as np


def _text(x):
    return x.decode("utf8") if isinstance(x, (bytes, np.bytes_)) else str(x)


def collect_h5_weights(root):
    """{name: ndarray} from an open Keras h5 file (or any object with the same mapping protocol).
    Handles `model.save` files (weights under /model_weights) and `save_weights` files (at the root),
    nested models (a top-level layer that is itself a Model stores its sub-layers' full names)."""
    grp = root["model_weights"] if "model_weights" in root else root
    if "layer_names" not in grp.attrs:
        raise ValueError("not a Keras weight file: no 'layer_names' attribute")
    out = {}
    for lname in grp.attrs["layer_names"]:
        layer = grp[_text(lname)]
        for wname in layer.attrs.get("weight_names", []):
            wname = _text(wname)
            node = layer
            for part in wname.split("/"):
                node = node[part]
            key = wname[:-2] if wname.endswith(":0") else wname
            if key in out:
                raise ValueError(f"duplicate weight name in the h5 file: {key}")
            out[key] = np.asarray(node)
    return out


def match_to_model(weights, specs):
    """Map file tensors onto the model's declared weights.  specs: {name: shape}.  Exact names first;
    otherwise a unique file key that ends with '/<name>' (an outer model scope such as 'backbone/').
    Returns (matched {name: array}, report dict with missing / unexpected / shape_mismatch lists)."""
    matched, missing, mismatch, used = {}, [], [], set()
    keys = list(weights)
    # an outer scope shared by the file's names ("inference/...") is found from the names that match
    # by suffix unambiguously, then applied to the ambiguous ones (e.g. 'conv1/kernel')
    votes = {}
    for name in specs:
        if name not in weights:
            cands = [k for k in keys if k.endswith("/" + name)]
            if len(cands) == 1:
                scope = cands[0][:-len(name)]
                votes[scope] = votes.get(scope, 0) + 1
    scope = max(votes, key=votes.get) if votes else ""
    for name, shape in specs.items():
        src = name if name in weights else (scope + name if scope + name in weights else None)
        if src is None:
            cands = [k for k in keys if k.endswith("/" + name)]
            if len(cands) == 1:
                src = cands[0]
        if src is None:
            missing.append(name)
            continue
        arr = np.asarray(weights[src], np.float32)
        if shape is not None and tuple(arr.shape) != tuple(shape):
            mismatch.append((name, tuple(arr.shape), tuple(shape)))
            continue
        matched[name] = arr
        used.add(src)
    unexpected = [k for k in keys if k not in used]
    return matched, {"missing": missing, "unexpected": unexpected, "shape_mismatch": mismatch}


def model_specs(backbone_type):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "..", "instance-segmentation-road-project_amd"))
    from masklab_hip import ModelConfiguration, retinamasklab as R
    cfg = ModelConfiguration()
    cfg.backbone.backbone_type = backbone_type
    _, model = R.construct_masklab_networks(cfg)
    return {k: tuple(v.shape) for k, v in model.weight_specs().items()}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("h5_path")
    ap.add_argument("npz_path")
    ap.add_argument("--backbone", default="resnext50")
    ap.add_argument("--allow-missing", action="store_true")
    args = ap.parse_args(argv)
    try:
        import h5py
    except ImportError:
        raise SystemExit("convert_keras_h5.py needs h5py to read .h5 files (pip install h5py on the machine that "
                         "holds the checkpoint); the .npz it writes has no such dependency")
    with h5py.File(args.h5_path, "r") as f:
        weights = collect_h5_weights(f)
    matched, report = match_to_model(weights, model_specs(args.backbone))
    for k in ("missing", "shape_mismatch"):
        for item in report[k]:
            print(f"{k}: {item}", file=sys.stderr)
    print(f"{len(matched)} tensors matched, {len(report['unexpected'])} file tensors unused "
          f"(training-only layers, optimizer state)", file=sys.stderr)
    if (report["missing"] or report["shape_mismatch"]) and not args.allow_missing:
        raise SystemExit("the checkpoint does not cover the model (see above); --allow-missing to write anyway")
    np.savez(args.npz_path, **matched)


if __name__ == "__main__":
    main()
