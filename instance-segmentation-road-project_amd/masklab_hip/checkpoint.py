"""Keras .h5 checkpoint -> this package's name-keyed weights.

The reference saves its models with Keras (`model.save(...)` / `save_weights(...)`, engine/callbacks.py:143-158) and
re-wires a loaded model by layer-name regexes (engine/retinamasklab.py:498-586).  Here the same names key the tensors
directly: `<layer>/<sub-layer>/.../<weight>` with the TF suffix `:0` dropped; the sub-layers the reference leaves to
Keras' automatic numbering (`conv2d_7`, `group_normalization_3`, ...) are mapped onto this package's hierarchical names
by creation order (rename_keras_auto_names).  Keras stores
  /[model_weights/]<top layer>  attrs['weight_names'] = [b'<scope>/<weight>:0', ...]  + one dataset each
so no graph has to be rebuilt: walk the groups, strip the suffix, and check that every tensor the target model declares
(name, shape) is present.

Opening a real .h5 needs `h5py` (imported lazily, only by load_keras_h5; not installed in the build container: the file
walking is written against the small mapping protocol h5py implements -- `.attrs`, `[]`, `in` -- and is tested with an
in-memory stand-in).  tools/convert_keras_h5.py is the offline command-line form (h5 -> npz)."""
import sys

import numpy as np


import re


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


# ------------------------------------------------------------------ Keras auto-numbered sub-layer names
# The reference does not name the sub-layers of its custom layers: FeaturePyramid's laterals (detection.py:42), the
# tower / mask-head / decoder convs and GroupNormalizations (detection.py:109-130,179-202, instance.py:177-201,
# semantic.py:205-219) and SqueezeExcite's Dense pair (misc.py:34-40) get Keras' automatic names
# `conv2d_N`, `group_normalization_N`, `conv2d_transpose_N`, `squeeze_excite_N`, `mobile_separable_conv2d_N`,
# `dense_N`, with N from ONE counter per class for the whole session -- so N itself depends on what was built before,
# but inside one custom layer the ORDER of the N's is the constructor's creation order.  This package gives the same
# sub-layers hierarchical names (block{b}/conv{i} ...); the two are matched by that order, per scope and per class.
_SCOPES = ("feature_pyramid", "classification_sub_net", "box_regression_sub_net", "mask_sub_net",
           "segmentation_sub_net")
_AUTO = ("conv2d_transpose", "conv2d", "group_normalization", "squeeze_excite", "mobile_separable_conv2d",
         "depthwise_conv2d", "dense")
_AUTO_RE = re.compile(r"^(%s)(?:_(\d+))?$" % "|".join(_AUTO))
_SEP_PARTS = ("expand_conv", "expand_GN", "depthwise_GN", "depthwise", "squeeze_conv", "squeeze_GN")


def _our_sublayers(specs):
    """{scope: {keras class: [our sub-layer prefix, ...] in the reference's creation order}} from this package's
    weight names (the naming convention of masklab_hip/layers/*.py)."""
    out = {}

    def add(scope, cls, order, prefix):
        lst = out.setdefault(scope, {}).setdefault(cls, [])
        if (order, prefix) not in lst:
            lst.append((order, prefix))

    for name in specs:
        parts = name.split("/")
        scope = parts[0]
        if scope not in _SCOPES or len(parts) < 3:
            continue
        if scope == "feature_pyramid":
            m = re.match(r"^C(\d+)_lateral$", parts[1])
            if m:                                   # created for strides in DESCENDING order (detection.py:39-43)
                add(scope, "conv2d", (-int(m.group(1)),), f"{scope}/{parts[1]}")
            continue
        b = 0
        rest = parts[1:-1]
        if re.match(r"^block\d+$", rest[0]):
            b = int(rest[0][5:])
            base = f"{scope}/{rest[0]}"
            rest = rest[1:]
        else:
            base = scope
        if not rest:
            continue
        sub = rest[0]
        m = re.match(r"^(conv|gn|se)(\d+)$", sub)
        ms = re.match(r"^sep(\d+)_(%s)$" % "|".join(_SEP_PARTS), sub)
        if m:
            kind, i = m.group(1), int(m.group(2))
            cls = {"conv": "conv2d", "gn": "group_normalization", "se": "squeeze_excite"}[kind]
            add(scope, cls, (b, i, 0), f"{base}/{sub}")
        elif ms:
            add(scope, "mobile_separable_conv2d", (b, int(ms.group(1)), 0), f"{base}/sep{ms.group(1)}")
        elif sub == "deconv":
            add(scope, "conv2d_transpose", (b, 0, 0), f"{base}/deconv")
        elif sub == "output":                      # created after the block's tower (and after the deconv)
            add(scope, "conv2d", (b, 1 << 20, 0), f"{base}/output")
    return {sc: {cls: [p for _, p in sorted(lst)] for cls, lst in d.items()} for sc, d in out.items()}


def rename_keras_auto_names(weights, specs, table=None):
    """Rewrite the keys of a reference-named weight dict (`<...>/<scope>[_k]/<auto name>/.../<weight>`) to this
    package's names.  Keys that are not under an auto-named sub-layer are returned unchanged.  Raises ValueError
    when a scope holds a different number of auto-named sub-layers of some class than the model declares.
    `table` (a list) receives one (scope, keras class, N, checkpoint prefix, model prefix) row per matched sub-layer:
    the creation-order pairing that was applied, for a user to audit against their checkpoint."""
    ours = _our_sublayers(specs)
    found = {}                                      # (scope, class) -> {N: file prefix up to the sub-layer}
    located = []                                    # (key, scope, class, N, remainder parts)
    for key in weights:
        parts = key.split("/")
        for pos, comp in enumerate(parts[:-2]):
            sc = re.sub(r"_\d+$", "", comp)
            if sc in _SCOPES and comp in (sc,) + tuple(f"{sc}_{k}" for k in range(1, 1000)):
                m = _AUTO_RE.match(parts[pos + 1])
                if m:
                    n = int(m.group(2) or 0)
                    found.setdefault((sc, m.group(1)), {})[n] = "/".join(parts[:pos + 2])
                    located.append((key, sc, m.group(1), n, parts[pos + 2:]))
                break
    rank = {}
    for (sc, cls), by_n in found.items():
        want = ours.get(sc, {}).get(cls, [])
        if len(by_n) != len(want):
            raise ValueError(f"checkpoint has {len(by_n)} auto-named '{cls}' sub-layers under '{sc}', the model "
                             f"declares {len(want)}: different head configuration (num_depth / num_blocks / "
                             f"use_squeeze_excite / use_separable_conv)?")
        for r, n in enumerate(sorted(by_n)):
            rank[(sc, cls, n)] = want[r]
            if table is not None:
                table.append((sc, cls, n, by_n[n], want[r]))
    out = dict(weights)
    dense_seen = {}
    for key, sc, cls, n, rest in located:
        target = rank[(sc, cls, n)]
        if cls == "squeeze_excite":
            # .../squeeze_excite_k/dense_m/kernel: the two Dense layers are created in build() in order (misc.py:34-40)
            md = _AUTO_RE.match(rest[0])
            if not md or md.group(1) != "dense":
                raise ValueError(f"unexpected tensor inside a SqueezeExcite scope: {key}")
            dense_seen.setdefault(target, set()).add(int(md.group(2) or 0))
            continue
        if cls == "mobile_separable_conv2d":
            # explicit inner names 'SeparableConv2d_<part>' (misc.py:73-93) -> '<block>/sep{i}_<part>'
            inner = re.sub(r"^SeparableConv2d_", "", rest[0])
            new = f"{target}_{inner}/" + "/".join(rest[1:])
        else:
            new = target + "/" + "/".join(rest)
        del out[key]
        out[new] = weights[key]
    for key, sc, cls, n, rest in located:
        if cls != "squeeze_excite":
            continue
        target = rank[(sc, cls, n)]
        order = sorted(dense_seen[target])
        md = _AUTO_RE.match(rest[0])
        which = order.index(int(md.group(2) or 0)) + 1
        del out[key]
        out[f"{target}/dense{which}/" + "/".join(rest[1:])] = weights[key]
    return out


def print_order_table(table, file=None):
    """The pairing rename_keras_auto_names applied, one line per auto-named sub-layer: Keras numbers `conv2d_N`,
    `group_normalization_N`, ... from one session-wide counter per class, so only the ORDER of the N's inside a
    custom layer is meaningful (engine/layers/detection.py:39-43,109-130,179-202, instance.py:177-201,
    semantic.py:205-219); check a few rows against `model.summary()` / the layer's sub-layers of the checkpoint."""
    file = file or sys.stderr
    print("creation-order table (checkpoint sub-layer -> this package's name):", file=file)
    for sc, cls, n, src, dst in sorted(table, key=lambda r: (r[0], r[1], r[2])):
        print(f"  {sc:24s} {cls:24s} N={n:<5d} {src}  ->  {dst}", file=file)


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


def load_keras_h5(path, specs, table=None):
    """Open a Keras .h5 (h5py, imported here) and return {this package's weight name: float32 array} for the model whose
    weight_specs() shapes are `specs` ({name: shape}).  Raises ImportError without h5py, ValueError when the checkpoint
    does not cover the model."""
    try:
        import h5py
    except ImportError as e:
        raise ImportError("reading a Keras .h5 checkpoint needs the `h5py` package (not part of this build's "
                          "environment).  Convert the file once where h5py is available -- "
                          "`python tools/convert_keras_h5.py weights.h5 weights.npz` -- and pass the .npz instead.") from e
    with h5py.File(path, "r") as f:
        weights = collect_h5_weights(f)
    matched, report = match_to_model(rename_keras_auto_names(weights, specs, table), specs)
    if report["missing"] or report["shape_mismatch"]:
        raise ValueError(f"{path} does not cover the model: {len(report['missing'])} tensors missing "
                         f"(first: {report['missing'][:3]}), {len(report['shape_mismatch'])} with another shape "
                         f"(first: {report['shape_mismatch'][:3]})")
    return matched
