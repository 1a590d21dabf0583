"""Generate tests/golden/config_defaults.json from the REFERENCE's own engine/config.py.

engine/config.py is pure Python (argparse + os only); it is loaded by file path -- never via the `engine` package, whose
__init__ imports tensorflow -- exactly like make_prior_golden.py loads engine/prior.py.  Only the OUTPUT of
`ModelConfiguration().to_dict()` (reference engine/config.py:190-199) is committed: data, not source.  Values that are
paths under the reference's own checkout (ROOT_DIR, config.py:7) are stored relative to it as {"__root__": "<rest>"};
tuples become lists in JSON (the comparison in tests/test_host_cpu.py normalises both sides the same way).

Run in the build container only (needs /root/reference):
    python tests/golden/make_config_golden.py
"""
import importlib.util
import json
import os

import sys

sys.dont_write_bytecode = True      # /root/reference is read-only input: leave no __pycache__ beside the module loaded from it

REF = "/root/reference/engine/config.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_defaults.json")


def normalise(value, root):
    """JSON-able form of a config value: tuples -> lists, paths under `root` -> {"__root__": relative}."""
    if isinstance(value, (list, tuple)):
        return [normalise(v, root) for v in value]
    if isinstance(value, dict):
        return {str(k): normalise(v, root) for k, v in value.items()}
    if isinstance(value, str) and root and (value == root or value.startswith(root + os.sep)):
        return {"__root__": os.path.relpath(value, root)}
    return value


def main():
    spec = importlib.util.spec_from_file_location("ref_config", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.ModelConfiguration()
    doc = {"source": "reference engine/config.py:10-188 -> ModelConfiguration().to_dict() (:190-199)",
           "groups": list(dir(cfg)),
           "defaults": {g: {k: normalise(v, mod.ROOT_DIR) for k, v in attrs.items()} for g, attrs in cfg.to_dict().items()}}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print(f"wrote {OUT}: {sum(len(v) for v in doc['defaults'].values())} fields in {len(doc['defaults'])} groups")


if __name__ == "__main__":
    main()
