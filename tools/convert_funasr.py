#!/usr/bin/env python3
"""CLI over asr-2pass_amd/convert.py: FunASR model directory -> <out>.bin + <out>.json.

    python3 tools/convert_funasr.py asr  <model_dir> <out_prefix>     # model.pt + am.mvn + config.yaml
    python3 tools/convert_funasr.py vad  <model_dir> <out_prefix>
    python3 tools/convert_funasr.py punc <model_dir> <out_prefix>
Untested against real checkpoints (none are available offline); see the module docstring.
"""
import os
import sys

import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package  # noqa: E402


def main():
    kind, src, out = sys.argv[1:4]
    pkg = load_package()
    import importlib
    conv = importlib.import_module(pkg.__name__ + ".convert")
    wt = importlib.import_module(pkg.__name__ + ".weights")
    pt = next(os.path.join(src, n) for n in ("model.pt", "model.pb") if os.path.exists(os.path.join(src, n)))
    state = conv.load_state_dict(pt)
    if kind == "punc":
        man, blob = conv.convert_punc(state)
    else:
        shift, rescale = conv.parse_am_mvn(open(os.path.join(src, "am.mvn")).read())
        if kind == "asr":
            with open(os.path.join(src, "config.yaml")) as f:
                cfg = conv.config_from_yaml(yaml.safe_load(f))
            man, blob = conv.convert_paraformer(state, cfg, shift, rescale)
        else:
            man, blob = conv.convert_vad(state, shift, rescale)
    wt.save(out, man, blob)
    print(f"wrote {out}.bin ({blob.nbytes / 1e6:.1f} MB) and {out}.json: {len(man['tensors'])} tensors")


if __name__ == "__main__":
    main()
