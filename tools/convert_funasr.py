#!/usr/bin/env python3
"""CLI over asr-2pass_amd/convert.py: a model directory as the reference's server gets it -> <out>.bin + <out>.json
(+ tokens.json copied beside them for the host adapter).

    python3 tools/convert_funasr.py asr  <model_dir> <out_prefix> [--quant] [--from pt|onnx]
    python3 tools/convert_funasr.py vad  <model_dir> <out_prefix>
    python3 tools/convert_funasr.py punc <model_dir> <out_prefix>

<model_dir> is laid out as onnxruntime/include/com-define.h:52-88 names it: model.onnx (or model_quant.onnx with --quant)
[+ decoder.onnx for the online model, + model_eb.onnx for the hotword embedder], am.mvn, config.yaml, tokens.json — the
`...-onnx` ModelScope directories of websocket/run_server_offline.sh:26-36.  The ONNX files are read by the dependency-free
protobuf reader (asr-2pass_amd/onnx_reader.py); a directory that only holds the PyTorch checkpoint (model.pt) works too.
The layer-name mapping is UPSTREAM FunASR from memory and has never met a real Paraformer file (none is available offline).
"""
import os
import shutil
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package  # noqa: E402


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    quant = "--quant" in argv
    prefer = "auto"
    if "--from" in argv:
        prefer = argv[argv.index("--from") + 1]
        del argv[argv.index("--from"):argv.index("--from") + 2]
    argv = [a for a in argv if a != "--quant"]
    if len(argv) != 3 or argv[0] not in ("asr", "vad", "punc"):
        print(__doc__)
        return 2
    kind, src, out = argv
    pkg = load_package()
    import importlib
    conv = importlib.import_module(pkg.__name__ + ".convert")
    wt = importlib.import_module(pkg.__name__ + ".weights")
    man, blob, files = conv.convert_model_dir(kind, src, prefer=prefer, quantized=quant)
    wt.save(out, man, blob)
    tok = os.path.join(src, "tokens.json")
    if os.path.exists(tok) and os.path.dirname(os.path.abspath(out)) != os.path.abspath(src):
        shutil.copy(tok, os.path.join(os.path.dirname(os.path.abspath(out)), "tokens.json"))
    print(f"read {', '.join(os.path.basename(f) for f in files)}; wrote {out}.bin ({blob.nbytes / 1e6:.1f} MB) and {out}.json: "
          f"{len(man['tensors'])} tensors, contextual={man['config'].get('contextual', 0)} timestamp={man['config'].get('timestamp', 0)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
