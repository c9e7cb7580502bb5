"""CPU: the C++ reader of the reference's own model-file contract (asr-2pass_amd/csrc/model_files.cpp behind
pfhip_read_model_files; SURVEY §8 rows b / f2).  What Paraformer::InitAsr / FsmnVad::InitVad / CTTransformer::InitPunc open at
server start (onnxruntime/src/paraformer.cpp:21-53,56-154,178-241,325-360; fsmn-vad.cpp:10-50; ct-transformer.cpp:14-37;
tokenizer.cpp:130-183) is written here as synthetic directories in that layout (tests/ref_layout.py) and must come back as the
container the weights were taken from — bit for bit, and equal to what the Python converter (convert.py) makes of the same files.
No compute call: runs without a GPU.  Real Paraformer files are not available offline: UPSTREAM names stay unverified."""
import importlib
import json
import os
import time

import numpy as np
import pytest

import ref_layout as RL

REF = "/root/reference/utils"


@pytest.fixture(scope="module")
def mods(pkg):
    return pkg, importlib.import_module(pkg.__name__ + ".convert"), importlib.import_module(pkg.__name__ + ".weights")


@pytest.mark.parametrize("heads", [dict(), dict(contextual=1, timestamp=1)])
def test_offline_directory_comes_back_bit_for_bit(mods, tmp_path, heads):
    pkg, conv, wt = mods
    cfg = wt.small_config(enc_layers=2, dec_layers=2, vocab=97, **heads)
    man, blob = wt.synth_weights(cfg, seed=21)
    d = tmp_path / "asr"
    RL.write_asr_dir(str(d), conv, man, blob, cfg)
    hw = str(d / "model_eb.onnx") if heads else None
    man2, blob2, cached = pkg.read_model_files("asr", str(d / "model.onnx"), hotword=hw, cmvn=str(d / "am.mvn"), config=str(d / "config.yaml"))
    assert not cached and man2["tensors"] == man["tensors"] and man2["total_bytes"] == man["total_bytes"]
    for k in ("vocab", "contextual", "timestamp", "enc_layers", "dec_layers", "d_model", "n_head", "ffn", "dec_ffn", "kernel", "n_mels", "lfr_m", "lfr_n"):
        assert man2["config"].get(k, 0) == cfg.get(k, 0), k
    assert man2["config"]["cif_threshold"] == 1.0 and man2["config"]["tail_threshold"] == 0.45 and man2["config"]["fs"] == 16000
    assert np.array_equal(blob2, blob)               # transposes, LSTM gate order, conv singleton dims, am.mvn rows: all undone
    # the Python converter makes the same container of the same files
    man3, blob3, _ = conv.convert_model_dir("asr", str(d))
    assert man3["tensors"] == man2["tensors"] and np.array_equal(blob3, blob2)
    # second load: served from <dir>/model.pfhip.{bin,json}; a touched source invalidates it
    assert os.path.exists(d / "model.pfhip.bin") and os.path.exists(d / "model.pfhip.json")
    man4, blob4, cached = pkg.read_model_files("asr", str(d / "model.onnx"), hotword=hw, cmvn=str(d / "am.mvn"), config=str(d / "config.yaml"))
    assert cached and np.array_equal(blob4, blob) and man4["tensors"] == man["tensors"]
    os.utime(d / "am.mvn", ns=(time.time_ns(), time.time_ns() + 5_000_000_000))
    assert not pkg.read_model_files("asr", str(d / "model.onnx"), hotword=hw, cmvn=str(d / "am.mvn"), config=str(d / "config.yaml"))[2]


def test_the_strings_the_gpu_flavour_passes_resolve_to_the_onnx_file(mods, tmp_path, monkeypatch):
    """offline-stream.cpp:79-84: with use_gpu the reference passes <dir>/model.torchscript (or model_blade.torchscript); quantize=true
    passes model_quant.onnx (:74-77).  Each resolves to the ONNX file that is there; a directory with neither is a clear error."""
    pkg, conv, wt = mods
    monkeypatch.setenv("PFHIP_MODEL_CACHE", "0")
    cfg = wt.small_config(enc_layers=1, dec_layers=1, vocab=53)
    man, blob = wt.synth_weights(cfg, seed=22)
    d = tmp_path / "asr"
    RL.write_asr_dir(str(d), conv, man, blob, cfg)
    args = dict(cmvn=str(d / "am.mvn"), config=str(d / "config.yaml"))
    for name in ("model.torchscript", "model_blade.torchscript", "model_quant.onnx"):
        man2, blob2, cached = pkg.read_model_files("asr", str(d / name), **args)
        assert not cached and np.array_equal(blob2, blob), name
    assert not os.path.exists(d / "model.pfhip.bin")                     # PFHIP_MODEL_CACHE=0
    os.remove(d / "model.onnx")
    with pytest.raises(pkg.PfhipError, match="no ONNX file"):
        pkg.read_model_files("asr", str(d / "model.torchscript"), **args)
    with pytest.raises(pkg.PfhipError, match="am.mvn|cannot open"):
        RL.write_asr_dir(str(d), conv, man, blob, cfg)
        pkg.read_model_files("asr", str(d / "model.onnx"), cmvn=str(d / "nope.mvn"), config=str(d / "config.yaml"))


def test_online_directory_encoder_and_decoder_files(mods, tmp_path):
    """tpass-stream.cpp:63-64: en_model = <online-dir>/model.onnx, de_model = <online-dir>/decoder.onnx."""
    pkg, conv, wt = mods
    cfg = wt.small_config(enc_layers=2, dec_layers=2, vocab=61)
    man, blob = wt.synth_weights(cfg, seed=23)
    d = tmp_path / "online"
    RL.write_asr_dir(str(d), conv, man, blob, cfg, online=True)
    man2, blob2, _ = pkg.read_model_files("asr", str(d / "model.onnx"), second=str(d / "decoder.onnx"), cmvn=str(d / "am.mvn"), config=str(d / "config.yaml"))
    assert man2["tensors"] == man["tensors"] and np.array_equal(blob2, blob)
    with pytest.raises(pkg.PfhipError, match="decoder.output_layer.weight is not in the model files"):       # the decoder file left out: named, not zero-filled
        pkg.read_model_files("asr", str(d / "model.onnx"), cmvn=str(d / "am.mvn"), config=str(d / "config.yaml"))


def test_export_wrapper_names_and_quantised_file(mods, tmp_path, monkeypatch):
    pkg, conv, wt = mods
    monkeypatch.setenv("PFHIP_MODEL_CACHE", "0")
    cfg = wt.small_config(enc_layers=2, dec_layers=1, vocab=41)
    man, blob = wt.synth_weights(cfg, seed=24)
    d = tmp_path / "wrapped"
    RL.write_asr_dir(str(d), conv, man, blob, cfg, wrapper_model_component=True)      # encoder.model.encoders0.0...
    _, blob2, _ = pkg.read_model_files("asr", str(d / "model.onnx"), cmvn=str(d / "am.mvn"), config=str(d / "config.yaml"))
    assert np.array_equal(blob2, blob)
    q = tmp_path / "quant"
    RL.write_asr_dir(str(q), conv, man, blob, cfg, quantize=True)
    os.rename(q / "model.onnx", q / "model_quant.onnx")
    man3, blob3, _ = pkg.read_model_files("asr", str(q / "model_quant.onnx"), cmvn=str(q / "am.mvn"), config=str(q / "config.yaml"))
    w, w3 = RL.view(man, blob, "enc.1.ffn1.w"), RL.view(man3, blob3, "enc.1.ffn1.w")
    assert np.abs(w3 - w).max() <= np.abs(w).max() / 127 / 2 + 1e-7 and not np.array_equal(w3, w)       # int8 grid, folded back
    assert np.array_equal(RL.view(man3, blob3, "enc.1.ffn1.b"), RL.view(man, blob, "enc.1.ffn1.b"))
    man4, blob4, _ = conv.convert_model_dir("asr", str(q), quantized=True)
    assert np.array_equal(blob4, blob3)


def test_vad_and_punc_directories(mods, tmp_path):
    pkg, conv, wt = mods
    man, blob = wt.synth_vad_weights(seed=25)
    v = tmp_path / "vad"
    RL.write_vad_dir(str(v), conv, man, blob)
    man2, blob2, _ = pkg.read_model_files("vad", str(v / "model.onnx"), cmvn=str(v / "am.mvn"), config=str(v / "config.yaml"))
    assert man2["tensors"] == man["tensors"] and np.array_equal(blob2, blob)
    for k in ("n_mels", "lfr_m", "lfr_n", "input_dim", "affine", "linear", "proj", "lorder", "layers", "out_affine", "n_out"):
        assert man2["config"][k] == man["config"][k], k
    assert man2["config"]["max_end_silence_time"] == 800 and man2["config"]["speech_noise_thres"] == 0.9
    pc = dict(wt.CT_TRANSFORMER, vocab=300, sanm_shift=5)
    man, blob = wt.synth_punc_weights(pc, seed=26)
    p = tmp_path / "punc_realtime"
    RL.write_punc_dir(str(p), conv, man, blob, [f"t{i}" for i in range(300)])
    man2, blob2, _ = pkg.read_model_files("punc", str(p / "model.onnx"), config=str(p / "config.yaml"))
    assert man2["tensors"] == man["tensors"] and np.array_equal(blob2, blob)
    assert man2["config"]["punc_list"] == ["<unk>", "_", "，", "。", "？", "、"] and man2["config"]["sanm_shift"] == 5
    assert man2["config"]["vocab"] == 300 and man2["config"]["n_punc"] == 6


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree (real .onnx files) is not on this machine")
@pytest.mark.parametrize("rel", ["DNSMOS/bak_ovr.onnx", "DNSMOS/model_v8.onnx", "DNSMOS/sig.onnx", "DNSMOS/sig_bak_ovr.onnx", "pDNSMOS/sig_bak_ovr.onnx"])
def test_cpp_reader_on_the_reference_s_real_onnx_files(mods, rel):
    """The five genuine .onnx files the reference ships (utils/DNSMOS, keras2onnx exports: raw_data AND float_data tensors, int64
    shape constants): the C++ wire-format walk and the Python one agree on counts, bytes, graph closure and the values."""
    pkg, conv, wt = mods
    R = importlib.import_module(pkg.__name__ + ".onnx_reader")
    path = os.path.join(REF, rel)
    m = R.read_model(path)
    got = pkg.onnx_summary(path)
    assert got["initializers"] == len(m.initializers) and got["nodes"] == len(m.nodes) and got["open_inputs"] == 0
    assert got["initializer_bytes"] == sum(a.nbytes for a in m.initializers.values())
    st = R.torch_style_state(m)
    assert got["state_tensors"] == len(st)
    want = float(sum(np.asarray(a, np.float64).sum() for a in st.values()))
    assert abs(got["float_sum"] - want) <= 1e-5 * max(1.0, abs(want))


def test_malformed_files_are_errors_not_crashes(mods, tmp_path):
    pkg, conv, wt = mods
    cfg = wt.small_config(enc_layers=1, dec_layers=1, vocab=31)
    man, blob = wt.synth_weights(cfg, seed=27)
    d = tmp_path / "asr"
    RL.write_asr_dir(str(d), conv, man, blob, cfg)
    args = dict(cmvn=str(d / "am.mvn"), config=str(d / "config.yaml"))
    raw = open(d / "model.onnx", "rb").read()
    for cut in (len(raw) // 2, len(raw) - 3, 7):
        with open(d / "model.onnx", "wb") as f:
            f.write(raw[:cut])
        with pytest.raises(pkg.PfhipError):
            pkg.read_model_files("asr", str(d / "model.onnx"), **args)
    with open(d / "model.onnx", "wb") as f:
        f.write(raw)
    with open(d / "am.mvn", "w") as f:
        f.write("<Nnet>\n</Nnet>\n")
    with pytest.raises(pkg.PfhipError, match="AddShift"):
        pkg.read_model_files("asr", str(d / "model.onnx"), **args)
    RL.write_mvn(str(d / "am.mvn"), np.zeros(10), np.ones(10))
    with pytest.raises(pkg.PfhipError, match="cmvn.mean: 10 values, the model needs 560"):
        pkg.read_model_files("asr", str(d / "model.onnx"), **args)
    RL.write_mvn(str(d / "am.mvn"), RL.view(man, blob, "cmvn.mean"), RL.view(man, blob, "cmvn.istd"))
    with open(d / "config.yaml", "w") as f:
        f.write(RL.asr_config_yaml(dict(cfg, enc_layers=2)))            # one layer more than the file holds
    with pytest.raises(pkg.PfhipError, match="lack 13 tensors: enc.1.norm1.g <- encoder.encoders.0.norm1.weight"):
        pkg.read_model_files("asr", str(d / "model.onnx"), **args)
    with pytest.raises(pkg.PfhipError, match="kind"):
        pkg.read_model_files("tts", str(d / "model.onnx"), **args)


def test_damaged_files_under_address_sanitizer(mods, tmp_path):
    """The reader built for the CPU with -fsanitize=address,undefined (tools/fuzz/model_files_fuzz.cpp) loads 400 damaged copies of
    a valid directory — truncations, flipped bytes, huge varints, shifted framing, in the ONNX file, am.mvn and config.yaml: every
    one is either loaded or refused with a message; the sanitizers report nothing (a report ends the process with a non-zero code)."""
    import shutil
    import subprocess
    pkg, conv, wt = mods
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "fuzz"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        os.path.join(root, "tools", "fuzz", "model_files_fuzz.cpp"), os.path.join(root, "asr-2pass_amd", "csrc", "model_files.cpp"),
                        "-o", str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    cfg = wt.small_config(enc_layers=2, dec_layers=1, vocab=23, contextual=0, timestamp=1, d_model=32, ffn=64, dec_ffn=64, n_head=2)     # tiny: the reader needs no device
    man, blob = wt.synth_weights(cfg, seed=29)
    d = tmp_path / "asr"
    RL.write_asr_dir(str(d), conv, man, blob, cfg)
    out = subprocess.run([str(exe), "asr", str(d), "400", "7"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["iterations"] == 400 and res["loaded"] + res["refused"] == 400 and res["refused"] >= 100 and res["loaded"] >= 20, res
