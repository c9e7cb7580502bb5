"""asr-2pass_amd — MI355X-native Paraformer acoustic-model forward behind the FunASR `Model` seam.

The product is the gfx950 shared library `libpfhip.so` (C ABI in include/pfhip.h, sources in csrc/).
This Python module is only the ctypes binding used by the tests and bench.py, plus `ParaformerHip`,
a thin mirror of the reference's plug-in interface (`funasr::Model`, onnxruntime/include/model.h:13-46;
batched contract onnxruntime/src/paraformer-torch.cpp:301-475) with the same method names and error
behaviour.  There is NO CPU fallback: if the HIP library is missing or a HIP call fails, this raises.
"""
from __future__ import annotations

import ctypes
import json
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PFHIP_LIB: another build of the same library (A/B timing of two builds inside one GPU session; tools/ab_bench.sh)
LIB_PATH = os.environ.get("PFHIP_LIB") or os.path.join(_HERE, "libpfhip.so")

PFHIP_NUM_KCLASS = 8
KCLASS_NAMES = ["gemm", "attention", "layernorm", "fsmn", "fbank", "cif", "head", "other"]

# every symbol include/pfhip.h declares (tests check the built library exports exactly these)
ABI_SYMBOLS = [
    "pfhip_last_error", "pfhip_create", "pfhip_create_from_memory", "pfhip_create_from_files", "pfhip_read_model_files", "pfhip_container_blob",
    "pfhip_container_manifest", "pfhip_container_from_cache", "pfhip_container_free", "pfhip_onnx_summary", "pfhip_vad_create_from_files", "pfhip_punc_create_from_files", "pfhip_create_group", "pfhip_group_size", "pfhip_group_stats", "pfhip_destroy",
    "pfhip_sample_rate", "pfhip_vocab_size", "pfhip_feat_dim", "pfhip_d_model",
    "pfhip_offline_forward", "pfhip_offline_enqueue", "pfhip_offline_fetch", "pfhip_offline_forward_resident",
    "pfhip_set_batching", "pfhip_set_inflight", "pfhip_warm_up", "pfhip_get_inflight", "pfhip_inflight_stats", "pfhip_is_contextual", "pfhip_has_timestamp_head", "pfhip_hotword_embed", "pfhip_set_hotwords",
    "pfhip_extract_feats", "pfhip_get_tensor", "pfhip_debug_poke", "pfhip_profile_enable", "pfhip_profile_read",
    "pfhip_stream_create", "pfhip_stream_destroy", "pfhip_stream_reset", "pfhip_stream_forward", "pfhip_stream_last_path", "pfhip_stream_forward_batch", "pfhip_set_stream_batching",
    "pfhip_stream_set_debug", "pfhip_stream_get_tensor",
    "pfhip_vad_create_from_memory", "pfhip_vad_destroy", "pfhip_vad_reset", "pfhip_vad_num_classes", "pfhip_vad_forward",
    "pfhip_vad_forward_sil", "pfhip_vad_stream_create", "pfhip_vad_stream_destroy", "pfhip_vad_stream_reset",
    "pfhip_vad_stream_infer", "pfhip_vad_stream_infer_batch", "pfhip_set_vad_stream_batching", "pfhip_vadseg_create", "pfhip_vadseg_destroy", "pfhip_vadseg_reset", "pfhip_vadseg_feed",
    "pfhip_timestamp_onnx", "pfhip_post_process",
    "pfhip_punc_create_from_memory", "pfhip_punc_destroy", "pfhip_punc_num_classes", "pfhip_punc_infer",
    "pfhip_punc_infer_online", "pfhip_punc_infer_batch", "pfhip_set_punc_batching", "pfhip_punc_add_punc",
]


class PfhipError(RuntimeError):
    pass


class _Out(ctypes.Structure):
    _fields_ = [
        ("token_ids", ctypes.POINTER(ctypes.c_int32)),
        ("token_num", ctypes.POINTER(ctypes.c_int32)),
        ("n_fires", ctypes.POINTER(ctypes.c_int32)),
        ("n_frames", ctypes.POINTER(ctypes.c_int32)),
        ("logp", ctypes.POINTER(ctypes.c_float)),
        ("max_tokens", ctypes.c_int32),
        ("us_alphas", ctypes.POINTER(ctypes.c_float)),
        ("us_peaks", ctypes.POINTER(ctypes.c_float)),
        ("us_len", ctypes.POINTER(ctypes.c_int32)),
        ("max_us", ctypes.c_int32),
    ]


class _SlotStats(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("context", ctypes.c_int32), ("forwards", ctypes.c_int64),
                ("calls", ctypes.c_int64), ("utterances", ctypes.c_int64)]


class _Profile(ctypes.Structure):
    _fields_ = [
        ("ms", ctypes.c_double * PFHIP_NUM_KCLASS),
        ("launches", ctypes.c_int64 * PFHIP_NUM_KCLASS),
        ("flops", ctypes.c_double * PFHIP_NUM_KCLASS),
        ("bytes", ctypes.c_double * PFHIP_NUM_KCLASS),
    ]


_lib = None


def load_lib() -> ctypes.CDLL:
    """Loads libpfhip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PfhipError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.pfhip_last_error.restype = ctypes.c_char_p
    lib.pfhip_create.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ci, ctypes.POINTER(vp)]
    lib.pfhip_create_from_memory.argtypes = [vp, ctypes.c_size_t, ctypes.c_char_p, ci, ctypes.POINTER(vp)]
    lib.pfhip_destroy.argtypes = [vp]
    lib.pfhip_destroy.restype = None
    cs = ctypes.c_char_p
    lib.pfhip_create_from_files.argtypes = [cs, cs, cs, cs, cs, ci, ctypes.POINTER(vp)]
    lib.pfhip_vad_create_from_files.argtypes = [cs, cs, cs, ci, ctypes.POINTER(vp)]
    lib.pfhip_punc_create_from_files.argtypes = [cs, cs, ci, ctypes.POINTER(vp)]
    lib.pfhip_read_model_files.argtypes = [cs, cs, cs, cs, cs, cs, ctypes.POINTER(vp)]
    lib.pfhip_container_blob.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
    lib.pfhip_container_blob.restype = vp
    lib.pfhip_container_manifest.argtypes = [vp]
    lib.pfhip_container_manifest.restype = cs
    lib.pfhip_container_from_cache.argtypes = [vp]
    lib.pfhip_container_free.argtypes = [vp]
    lib.pfhip_container_free.restype = None
    lib.pfhip_onnx_summary.argtypes = [cs, ctypes.c_char_p, ctypes.c_size_t]
    lib.pfhip_create_group.argtypes = [vp, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ci), ci, ctypes.POINTER(vp)]
    lib.pfhip_group_size.argtypes = [vp]
    lib.pfhip_group_stats.argtypes = [vp, vp, vp, vp, vp, ci]
    for f in ("pfhip_sample_rate", "pfhip_vocab_size", "pfhip_feat_dim", "pfhip_d_model"):
        getattr(lib, f).argtypes = [vp]
    lib.pfhip_offline_forward.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ci), ci, vp, ci, ctypes.POINTER(_Out)]
    lib.pfhip_offline_enqueue.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ci), ci, vp]
    lib.pfhip_offline_fetch.argtypes = [vp, ctypes.POINTER(_Out)]
    lib.pfhip_offline_forward_resident.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ci), ci, ctypes.POINTER(_Out)]
    lib.pfhip_extract_feats.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ci), ci, vp, ctypes.c_size_t, vp]
    lib.pfhip_get_tensor.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    lib.pfhip_set_batching.argtypes = [vp, ci, ci]
    lib.pfhip_set_inflight.argtypes = [vp, ci]
    lib.pfhip_get_inflight.argtypes = [vp]
    lib.pfhip_warm_up.argtypes = [vp, ci, ci]
    lib.pfhip_inflight_stats.argtypes = [vp, ctypes.POINTER(_SlotStats), ci, ctypes.POINTER(ci)]
    lib.pfhip_is_contextual.argtypes = [vp]
    lib.pfhip_hotword_embed.argtypes = [vp, vp, vp, ci, vp]
    lib.pfhip_set_hotwords.argtypes = [vp, vp, ci]
    lib.pfhip_stream_create.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(vp)]
    lib.pfhip_stream_destroy.argtypes = [vp]
    lib.pfhip_stream_destroy.restype = None
    lib.pfhip_stream_reset.argtypes = [vp]
    lib.pfhip_stream_forward.argtypes = [vp, vp, ci, ci, vp, ci, ctypes.POINTER(ci)]
    lib.pfhip_stream_last_path.argtypes = [vp]
    lib.pfhip_stream_forward_batch.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
    lib.pfhip_set_stream_batching.argtypes = [vp, ci, ci]
    lib.pfhip_stream_set_debug.argtypes = [vp, ci]
    lib.pfhip_stream_get_tensor.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    lib.pfhip_vad_create_from_memory.argtypes = [vp, ctypes.c_size_t, ctypes.c_char_p, ci, ctypes.POINTER(vp)]
    lib.pfhip_vad_destroy.argtypes = [vp]
    lib.pfhip_vad_destroy.restype = None
    lib.pfhip_vad_reset.argtypes = [vp]
    lib.pfhip_vad_num_classes.argtypes = [vp]
    lib.pfhip_vad_forward.argtypes = [vp, vp, ci, ci, vp, ctypes.c_size_t, ctypes.POINTER(ci)]
    lib.pfhip_vad_forward_sil.argtypes = [vp, vp, ci, ci, vp, ctypes.c_size_t, ctypes.POINTER(ci)]
    lib.pfhip_vad_stream_create.argtypes = [vp, ctypes.POINTER(vp)]
    lib.pfhip_vad_stream_destroy.argtypes = [vp]
    lib.pfhip_vad_stream_destroy.restype = None
    lib.pfhip_vad_stream_reset.argtypes = [vp]
    lib.pfhip_vad_stream_infer.argtypes = [vp, vp, ci, ci, vp, ctypes.c_size_t, ctypes.POINTER(ci), vp, ctypes.c_size_t, ctypes.POINTER(ci)]
    lib.pfhip_vad_stream_infer_batch.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.pfhip_set_vad_stream_batching.argtypes = [vp, ci, ci]
    lib.pfhip_vadseg_create.argtypes = [ctypes.POINTER(vp)]
    lib.pfhip_vadseg_destroy.argtypes = [vp]
    lib.pfhip_vadseg_destroy.restype = None
    lib.pfhip_vadseg_reset.argtypes = [vp]
    lib.pfhip_vadseg_feed.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, ci, ctypes.c_float, ci, vp, ci, ctypes.POINTER(ci)]
    lib.pfhip_timestamp_onnx.argtypes = [vp, vp, ci, ci, ctypes.c_float, ctypes.c_float, vp, ci, ctypes.POINTER(ci)]
    lib.pfhip_punc_create_from_memory.argtypes = [vp, ctypes.c_size_t, ctypes.c_char_p, ci, ctypes.POINTER(vp)]
    lib.pfhip_punc_destroy.argtypes = [vp]
    lib.pfhip_punc_destroy.restype = None
    lib.pfhip_punc_num_classes.argtypes = [vp]
    lib.pfhip_punc_infer.argtypes = [vp, vp, ci, vp, vp]
    lib.pfhip_punc_infer_online.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.pfhip_punc_add_punc.argtypes = [vp, vp, ci, vp, ci, ctypes.POINTER(ci)]
    lib.pfhip_punc_infer_batch.argtypes = [vp, vp, vp, vp, ci, vp]
    lib.pfhip_set_punc_batching.argtypes = [vp, ci, ci]
    lib.pfhip_debug_poke.argtypes = [vp, ctypes.c_char_p, ci]
    lib.pfhip_profile_enable.argtypes = [vp, ci]
    lib.pfhip_profile_read.argtypes = [vp, ctypes.POINTER(_Profile), ci]
    _lib = lib
    return lib


def _check(lib, st):
    if st != 0:
        raise PfhipError(f"pfhip status {st}: {lib.pfhip_last_error().decode()}")


def read_model_files(kind, model, second=None, hotword=None, cmvn=None, config=None):
    """The reference's own file contract (com-define.h:52-88) -> (manifest dict, float32 blob, from_cache) through the C++ reader
    in libpfhip.so (pfhip_read_model_files); kind in {"asr", "vad", "punc"}.  No device needed."""
    lib = load_lib()
    enc = lambda s: None if s is None else str(s).encode()
    h = ctypes.c_void_p()
    _check(lib, lib.pfhip_read_model_files(kind.encode(), enc(model), enc(second), enc(hotword), enc(cmvn), enc(config), ctypes.byref(h)))
    try:
        nbytes = ctypes.c_size_t()
        ptr = lib.pfhip_container_blob(h, ctypes.byref(nbytes))
        blob = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), shape=(nbytes.value // 4,)).copy() if nbytes.value else np.zeros(0, np.float32)
        man = json.loads(lib.pfhip_container_manifest(h).decode())
        return man, blob, bool(lib.pfhip_container_from_cache(h))
    finally:
        lib.pfhip_container_free(h)


def onnx_summary(path):
    """pfhip_onnx_summary: what the C++ wire-format walk saw in one .onnx file (dict)."""
    lib = load_lib()
    buf = ctypes.create_string_buffer(1024)
    _check(lib, lib.pfhip_onnx_summary(str(path).encode(), buf, len(buf)))
    return json.loads(buf.value.decode())


class ParaformerHip:
    """Host-side mirror of `funasr::Model` for the offline Paraformer path.

    InitAsr   <-> Paraformer::InitAsr          (onnxruntime/src/paraformer.cpp:21-53)
    Forward   <-> Model::Forward(float** din, int* len, bool input_finished, hw_emb, decoder, batch_in)
                  (model.h:31-32; paraformer.cpp:463-589; paraformer-torch.cpp:301-475)
    GetAsrSampleRate / SetBatchSize / GetBatchSize as in model.h:40-42.
    Token ids -> text (Vocab::Vector2StringV2, vocab.cpp:164) is host string handling outside the
    hot path: Forward joins vocabulary entries when a token list is given, else
    returns space-separated ids.
    """

    def __init__(self):
        self._lib = load_lib()
        self._h = ctypes.c_void_p()
        self._batch_size = 1
        self._vocab: Optional[List[str]] = None
        self.cfg = None

    # -- lifetime ----------------------------------------------------------------------------------
    def InitAsr(self, am_model, am_cmvn=None, am_config=None, token_file=None, thread_num=1, device=0, devices=None,
                second_model=None, hw_model=None):
        """am_model: path prefix of `<prefix>.bin/.json`, a (manifest dict, float32 blob) pair, or — with am_cmvn and am_config —
        the string the reference passes (`<dir>/model.onnx`, `model_quant.onnx`, `model.torchscript`; second_model = the online
        model's `decoder.onnx`, hw_model = `model_eb.onnx`): pfhip_create_from_files reads the reference's own files.
        devices=[0, 1, ...]: one replica per listed device behind this one handle (pfhip_create_group)."""
        if self._h:
            self._lib.pfhip_destroy(self._h)
            self._h = ctypes.c_void_p()
        if isinstance(am_model, str) and am_cmvn and am_config and not os.path.exists(am_model + ".bin"):
            enc = lambda s: None if s is None else str(s).encode()
            _check(self._lib, self._lib.pfhip_create_from_files(enc(am_model), enc(second_model), enc(hw_model), enc(am_cmvn), enc(am_config),
                                                                device, ctypes.byref(self._h)))
            self.cfg = read_model_files("asr", am_model, second_model, hw_model, am_cmvn, am_config)[0]["config"]
        elif isinstance(am_model, (tuple, list)):
            man, blob = am_model
            blob = np.ascontiguousarray(blob, dtype=np.float32)
            if devices is not None:
                devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
                _check(self._lib, self._lib.pfhip_create_group(
                    blob.ctypes.data, blob.nbytes, json.dumps(man).encode(), devs, len(devices), ctypes.byref(self._h)))
            else:
                _check(self._lib, self._lib.pfhip_create_from_memory(
                    blob.ctypes.data, blob.nbytes, json.dumps(man).encode(), device, ctypes.byref(self._h)))
            self.cfg = man["config"]
        else:
            _check(self._lib, self._lib.pfhip_create(
                (am_model + ".bin").encode(), (am_model + ".json").encode(), device, ctypes.byref(self._h)))
            with open(am_model + ".json") as f:
                self.cfg = json.load(f)["config"]
        if token_file:
            with open(token_file) as f:
                self._vocab = json.load(f)
        return self

    def close(self):
        if self._h:
            self._lib.pfhip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def GetAsrSampleRate(self):
        return self._lib.pfhip_sample_rate(self._h)

    def SetBatchSize(self, n):
        self._batch_size = int(n)

    def GetBatchSize(self):
        return self._batch_size

    def group_stats(self):
        """Per replica of the handle: dict(devices, calls, utterances, open_streams) (pfhip_group_stats)."""
        n = self._lib.pfhip_group_size(self._h)
        dev = np.zeros(n, np.int32); calls = np.zeros(n, np.int64); utts = np.zeros(n, np.int64); streams = np.zeros(n, np.int32)
        _check(self._lib, self._lib.pfhip_group_stats(self._h, dev.ctypes.data, calls.ctypes.data, utts.ctypes.data, streams.ctypes.data, n))
        return dict(devices=dev.tolist(), calls=calls.tolist(), utterances=utts.tolist(), open_streams=streams.tolist())

    def set_stream_batching(self, wait_us, max_streams=128):
        """Merge concurrent ParaformerOnlineHip.Forward callers (one thread per connection) into batched forwards."""
        _check(self._lib, self._lib.pfhip_set_stream_batching(self._h, int(wait_us), int(max_streams)))

    def set_batching(self, wait_us, max_utterances=32):
        """Merge concurrent Forward callers into one packed device batch (pfhip_set_batching)."""
        _check(self._lib, self._lib.pfhip_set_batching(self._h, int(wait_us), int(max_utterances)))

    def set_inflight(self, n):
        """n execution contexts per device over the ONE weight set of this handle (pfhip_set_inflight)."""
        _check(self._lib, self._lib.pfhip_set_inflight(self._h, int(n)))

    def get_inflight(self):
        return self._lib.pfhip_get_inflight(self._h)

    def inflight_stats(self):
        """Per execution slot: dict(device, context, forwards, calls, utterances) (pfhip_inflight_stats)."""
        arr = (_SlotStats * 64)()
        n = ctypes.c_int(0)
        _check(self._lib, self._lib.pfhip_inflight_stats(self._h, arr, 64, ctypes.byref(n)))
        return [dict(device=a.device, context=a.context, forwards=a.forwards, calls=a.calls, utterances=a.utterances)
                for a in arr[:n.value]]

    def StartUtterance(self):  # paraformer.cpp:297-307: stateless
        pass

    def EndUtterance(self):
        pass

    def Reset(self):
        pass

    @property
    def vocab_size(self):
        return self._lib.pfhip_vocab_size(self._h)

    # -- the hot path --------------------------------------------------------------------------------
    def CompileHotwordEmbedding(self, hotword_ids: Sequence[Sequence[int]]):
        """Paraformer::CompileHotwordEmbedding from the id stage on (paraformer.cpp:629-693): each hotword's token ids are
        truncated / zero-padded to 10, the blank row [1,0,...] is appended, the embedder runs on the GPU and row len-1
        is taken.  (Hotword string -> token ids is host text handling, :601-628.)  Returns [H+1, d] float32; a plain
        model returns one zero row (:594-599)."""
        d = self._lib.pfhip_d_model(self._h)
        if not self._lib.pfhip_is_contextual(self._h):
            return np.zeros((1, d), np.float32)
        rows, lens = [], []
        for ids in hotword_ids:
            ids = list(ids)[:10]
            if not ids:
                continue
            rows.append(ids + [0] * (10 - len(ids)))
            lens.append(len(ids))
        rows.append([1] + [0] * 9)
        lens.append(1)
        mat = np.ascontiguousarray(rows, np.int32)
        ln = np.ascontiguousarray(lens, np.int32)
        out = np.zeros((len(rows), d), np.float32)
        _check(self._lib, self._lib.pfhip_hotword_embed(self._h, mat.ctypes.data, ln.ctypes.data, len(rows), out.ctypes.data))
        return out

    def forward_ids(self, din: Sequence[np.ndarray], want_logp=False, max_tokens=None, hw_emb=None, want_timestamps=False):
        """Batched forward.  Returns dict(token_num, n_fires, n_frames, ids=list of int arrays,
        logp=list of [n_fires, V] arrays or None[, us_alphas, us_peaks = lists of [3*T_b] arrays])."""
        B = len(din)
        if B == 0:
            raise PfhipError("empty batch")
        bufs = [np.ascontiguousarray(x, dtype=np.float32) for x in din]
        lens = (ctypes.c_int * B)(*[int(b.shape[0]) for b in bufs])
        ptrs = (ctypes.c_void_p * B)(*[b.ctypes.data if b.shape[0] else None for b in bufs])
        if max_tokens is None:
            max_tokens = max(1, max(int(b.shape[0]) for b in bufs) // 960 + 2)   # <= T+1 fires per utterance
        V = self.vocab_size
        ids = np.zeros((B, max_tokens), np.int32)
        tn = np.zeros(B, np.int32)
        nf = np.zeros(B, np.int32)
        fr = np.zeros(B, np.int32)
        logp = np.zeros((B, max_tokens, V), np.float32) if want_logp else None
        out = _Out()
        out.token_ids = ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.token_num = tn.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.n_fires = nf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.n_frames = fr.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.logp = logp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if want_logp else None
        out.max_tokens = max_tokens
        if want_timestamps:
            max_us = 3 * max(1, max(int(b.shape[0]) for b in bufs) // 960 + 2)
            usa = np.zeros((B, max_us), np.float32)
            usp = np.zeros((B, max_us), np.float32)
            usl = np.zeros(B, np.int32)
            out.us_alphas = usa.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            out.us_peaks = usp.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            out.us_len = usl.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
            out.max_us = max_us
        hw = np.ascontiguousarray(hw_emb, dtype=np.float32) if hw_emb is not None else None
        _check(self._lib, self._lib.pfhip_offline_forward(self._h, ptrs, lens, B, hw.ctypes.data if hw is not None else None,
                                                          int(hw.shape[0]) if hw is not None else 0, ctypes.byref(out)))
        res = dict(token_num=tn, n_fires=nf, n_frames=fr,
                   ids=[ids[b, :min(tn[b], nf[b])].copy() for b in range(B)],
                   logp=[logp[b, :nf[b]].copy() for b in range(B)] if want_logp else None)
        if want_timestamps:
            res["us_alphas"] = [usa[b, :usl[b]].copy() for b in range(B)]
            res["us_peaks"] = [usp[b, :usl[b]].copy() for b in range(B)]
        return res

    def Forward(self, din, len_=None, input_finished=True, hw_emb=None, decoder_handle=None, batch_in=1):
        """Same contract as Model::Forward: returns batch_in strings; an utterance that yields no
        feature frame gives "" (paraformer.cpp:477-480)."""
        din = list(din)[:batch_in]
        if len_ is not None:
            din = [np.asarray(x)[:n] for x, n in zip(din, len_)]
        r = self.forward_ids(din, hw_emb=hw_emb if self._lib.pfhip_is_contextual(self._h) else None)
        res = []
        for ids in r["ids"]:
            if self._vocab is not None:
                res.append("".join(self._vocab[i] for i in ids))
            else:
                res.append(" ".join(str(int(i)) for i in ids))
        return res

    def extract_feats(self, din: Sequence[np.ndarray]):
        """FbankKaldi + LfrCmvn (paraformer.cpp:309-323, 421-461) on the GPU; list of [T_b, 560]."""
        B = len(din)
        bufs = [np.ascontiguousarray(x, dtype=np.float32) for x in din]
        lens = (ctypes.c_int * B)(*[int(b.shape[0]) for b in bufs])
        ptrs = (ctypes.c_void_p * B)(*[b.ctypes.data if b.shape[0] else None for b in bufs])
        fd = self._lib.pfhip_feat_dim(self._h)
        cap = sum(max(0, (int(b.shape[0]) - 400) // 160 + 1 + 5) // 6 + 1 for b in bufs) * fd
        feats = np.zeros(cap, np.float32)
        nfr = np.zeros(B, np.int32)
        _check(self._lib, self._lib.pfhip_extract_feats(self._h, ptrs, lens, B, feats.ctypes.data, cap, nfr.ctypes.data))
        out, o = [], 0
        for b in range(B):
            out.append(feats[o * fd:(o + int(nfr[b])) * fd].reshape(int(nfr[b]), fd).copy())
            o += int(nfr[b])
        return out

    def get_tensor(self, name: str, cap_floats: int) -> np.ndarray:
        buf = np.zeros(cap_floats, np.float32)
        n = ctypes.c_size_t(0)
        _check(self._lib, self._lib.pfhip_get_tensor(self._h, name.encode(), buf.ctypes.data, cap_floats, ctypes.byref(n)))
        return buf[:n.value]

    # -- device-resident form (bench.py) -------------------------------------------------------------
    def enqueue_device(self, d_pcm_ptr: int, sample_off: np.ndarray, n_samples: np.ndarray, stream: int = 0):
        B = len(n_samples)
        so = np.ascontiguousarray(sample_off, np.int64)
        ns = np.ascontiguousarray(n_samples, np.int32)
        _check(self._lib, self._lib.pfhip_offline_enqueue(
            self._h, ctypes.c_void_p(d_pcm_ptr), so.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
            ns.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), B, ctypes.c_void_p(stream) if stream else None))

    def fetch(self, B: int, max_tokens: int):
        ids = np.zeros((B, max_tokens), np.int32)
        tn = np.zeros(B, np.int32)
        nf = np.zeros(B, np.int32)
        fr = np.zeros(B, np.int32)
        out = _Out()
        out.token_ids = ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.token_num = tn.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.n_fires = nf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.n_frames = fr.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.logp = None
        out.max_tokens = max_tokens
        _check(self._lib, self._lib.pfhip_offline_fetch(self._h, ctypes.byref(out)))
        return dict(token_num=tn, n_fires=nf, n_frames=fr, ids=[ids[b, :min(tn[b], nf[b])].copy() for b in range(B)])

    def forward_resident(self, d_pcm_ptr: int, sample_off: np.ndarray, n_samples: np.ndarray, max_tokens: int):
        """pfhip_offline_forward_resident: PCM already in HBM, routed over the handle's execution contexts like Forward."""
        B = len(n_samples)
        so = np.ascontiguousarray(sample_off, np.int64)
        ns = np.ascontiguousarray(n_samples, np.int32)
        ids = np.zeros((B, max_tokens), np.int32)
        tn = np.zeros(B, np.int32)
        nf = np.zeros(B, np.int32)
        fr = np.zeros(B, np.int32)
        out = _Out()
        out.token_ids = ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.token_num = tn.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.n_fires = nf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.n_frames = fr.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out.logp = None
        out.max_tokens = max_tokens
        _check(self._lib, self._lib.pfhip_offline_forward_resident(
            self._h, ctypes.c_void_p(d_pcm_ptr), so.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
            ns.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), B, ctypes.byref(out)))
        return dict(token_num=tn, n_fires=nf, n_frames=fr, ids=[ids[b, :min(tn[b], nf[b])].copy() for b in range(B)])

    def debug_poke(self, what, value=0):
        """pfhip_debug_poke: test hooks and read-outs ("range_flag", "range_fallbacks", "blstm_fallbacks", "plane_forwards")."""
        return int(self._lib.pfhip_debug_poke(self._h, what.encode(), int(value)))

    def profile_enable(self, on=True):
        """on: False/0 off, True/1 every kernel class, other int = bit mask of classes (see pfhip.h)."""
        _check(self._lib, self._lib.pfhip_profile_enable(self._h, int(on)))

    def profile_read(self, reset=True):
        p = _Profile()
        _check(self._lib, self._lib.pfhip_profile_read(self._h, ctypes.byref(p), 1 if reset else 0))
        return {KCLASS_NAMES[i]: dict(ms=p.ms[i], launches=p.launches[i], flops=p.flops[i], bytes=p.bytes[i])
                for i in range(PFHIP_NUM_KCLASS)}


class ParaformerOnlineHip:
    """Host-side mirror of `funasr::ParaformerOnline` (onnxruntime/src/paraformer-online.cpp): one object per
    connection, built from the (online) model handle like `ParaformerOnline(Model* offline_handle, chunk_size)`
    (:12-62).  Forward(din, len, input_finished) keeps the reference's name and meaning (:525-601) and returns
    the token ids emitted by the call (the reference returns their text)."""

    def __init__(self, offline_handle: ParaformerHip, chunk_size=(5, 10, 5)):
        self._lib = load_lib()
        self._model = offline_handle          # keeps the model alive
        self._h = ctypes.c_void_p()
        cs = (ctypes.c_int * 3)(*chunk_size)
        _check(self._lib, self._lib.pfhip_stream_create(offline_handle.handle, cs, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            if self._model.handle:            # (a model closed first must not be touched through its stream)
                self._lib.pfhip_stream_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Reset(self):
        _check(self._lib, self._lib.pfhip_stream_reset(self._h))

    def set_debug(self, on=True):
        _check(self._lib, self._lib.pfhip_stream_set_debug(self._h, 1 if on else 0))

    def last_path(self):
        """Which branch of ParaformerOnline::Forward the last call took (pfhip_stream_last_path): 2 = the one whose non-empty
        text the reference ends with a blank (paraformer-online.cpp:585-587)."""
        return int(self._lib.pfhip_stream_last_path(self._h))

    def Forward(self, din, len_=None, input_finished=False, cap=256):
        x = np.ascontiguousarray(din if len_ is None else np.asarray(din)[:len_], dtype=np.float32)
        ids = np.zeros(max(cap, 1), np.int32)
        n = ctypes.c_int(0)
        _check(self._lib, self._lib.pfhip_stream_forward(self._h, x.ctypes.data if x.size else None, int(x.size),
                                                         1 if input_finished else 0, ids.ctypes.data, int(cap), ctypes.byref(n)))
        return [int(v) for v in ids[:n.value]]

    @staticmethod
    def forward_batch(streams, dins, input_finished, caps=None):
        """ParaformerOnline::Forward for many connections of one model in one call (pfhip_stream_forward_batch): the chunk
        windows that are ready are packed into one forward.  Returns one id list per stream.  With `caps` (token-buffer
        capacities per stream) returns (status, id lists, n_tokens) instead of raising: a stream whose buffer is too small
        gets no ids and n_tokens = the count it needed, the others are served."""
        B = len(streams)
        if B == 0:
            return []
        lib = streams[0]._lib
        bufs = [np.ascontiguousarray(x, dtype=np.float32) for x in dins]
        hs = (ctypes.c_void_p * B)(*[s._h for s in streams])
        ptrs = (ctypes.c_void_p * B)(*[b.ctypes.data if b.size else None for b in bufs])
        lens = (ctypes.c_int * B)(*[int(b.size) for b in bufs])
        fin = (ctypes.c_int * B)(*[1 if f else 0 for f in input_finished])
        ids = np.zeros((B, 256), np.int32)
        idp = (ctypes.c_void_p * B)(*[ids[i].ctypes.data for i in range(B)])
        ccaps = (ctypes.c_int * B)(*([256] * B if caps is None else [int(c) for c in caps]))
        nt = (ctypes.c_int * B)()
        st = lib.pfhip_stream_forward_batch(hs, B, ptrs, lens, fin, idp, ccaps, nt)
        if caps is not None:
            return st, [[int(v) for v in ids[i, :min(nt[i], ccaps[i])]] for i in range(B)], [int(v) for v in nt]
        _check(lib, st)
        return [[int(v) for v in ids[i, :nt[i]]] for i in range(B)]

    def get_tensor(self, name: str, cap_floats: int) -> np.ndarray:
        buf = np.zeros(max(cap_floats, 1), np.float32)
        n = ctypes.c_size_t(0)
        _check(self._lib, self._lib.pfhip_stream_get_tensor(self._h, name.encode(), buf.ctypes.data, cap_floats, ctypes.byref(n)))
        return buf[:n.value]


class FsmnVadHip:
    """Host-side mirror of `funasr::FsmnVad` up to the frame scores (onnxruntime/src/fsmn-vad.cpp): InitVad,
    Forward(waves, is_final) -> probs [T, classes], InitCache.  The E2EVadModel post-processing state machine
    (e2e-vad.h) that turns scores into segments is host logic above this path."""

    def __init__(self):
        self._lib = load_lib()
        self._h = ctypes.c_void_p()

    def InitVad(self, vad_model, vad_cmvn=None, vad_config=None, thread_num=1, device=0):
        man, blob = vad_model
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        _check(self._lib, self._lib.pfhip_vad_create_from_memory(blob.ctypes.data, blob.nbytes, json.dumps(man).encode(),
                                                                 device, ctypes.byref(self._h)))
        return self

    def close(self):
        if self._h:
            self._lib.pfhip_vad_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def InitCache(self):
        _check(self._lib, self._lib.pfhip_vad_reset(self._h))

    def set_stream_batching(self, wait_us, max_streams):
        """Merge concurrent FsmnVadOnlineHip.Infer callers (one thread per connection) into batched device passes."""
        _check(self._lib, self._lib.pfhip_set_vad_stream_batching(self._h, int(wait_us), int(max_streams)))

    def ForwardSil(self, waves, is_final=False):
        """Frame-wise silence posterior only (what E2EVadModel reads)."""
        x = np.ascontiguousarray(waves, dtype=np.float32)
        cap = max(0, (x.size - 400) // 160 + 1) + 1
        sil = np.zeros(cap, np.float32)
        n = ctypes.c_int(0)
        _check(self._lib, self._lib.pfhip_vad_forward_sil(self._h, x.ctypes.data if x.size else None, int(x.size),
                                                          1 if is_final else 0, sil.ctypes.data, cap, ctypes.byref(n)))
        return sil[:n.value]

    def Forward(self, waves, is_final=False):
        x = np.ascontiguousarray(waves, dtype=np.float32)
        C = self._lib.pfhip_vad_num_classes(self._h)
        cap = (max(0, (x.size - 400) // 160 + 1) + 1) * C
        probs = np.zeros(cap, np.float32)
        n = ctypes.c_int(0)
        _check(self._lib, self._lib.pfhip_vad_forward(self._h, x.ctypes.data if x.size else None, int(x.size),
                                                      1 if is_final else 0, probs.ctypes.data, cap, ctypes.byref(n)))
        return probs[:n.value * C].reshape(n.value, C)


class CTTransformerHip:
    """Host-side mirror of `funasr::CTTransformer` for the forward only (onnxruntime/src/ct-transformer.cpp):
    InitPunc, Infer(input_data) -> punctuation ids.  Tokenisation / sentence re-segmentation (AddPunc) are host
    string logic above this path."""

    def __init__(self):
        self._lib = load_lib()
        self._h = ctypes.c_void_p()

    def InitPunc(self, punc_model, punc_config=None, token_file=None, thread_num=1, device=0):
        man, blob = punc_model
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        _check(self._lib, self._lib.pfhip_punc_create_from_memory(blob.ctypes.data, blob.nbytes, json.dumps(man).encode(),
                                                                  device, ctypes.byref(self._h)))
        return self

    def close(self):
        if self._h:
            self._lib.pfhip_punc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def AddPuncIds(self, input_data):
        """The id-level part of CTTransformer::AddPunc (ct-transformer.cpp:39-155): punctuation id per token (+ one appended
        period when the text does not end a sentence)."""
        ids = np.ascontiguousarray(input_data, dtype=np.int32)
        out = np.zeros(ids.size + 1, np.int32)
        n = ctypes.c_int(0)
        _check(self._lib, self._lib.pfhip_punc_add_punc(self._h, ids.ctypes.data, int(ids.size), out.ctypes.data, int(out.size), ctypes.byref(n)))
        return out[:n.value].copy()

    def Infer(self, input_data, want_logits=False, nCacheSize=None):
        """nCacheSize=None: CTTransformer::Infer; an int: CTTransformerOnline::Infer(input_data, nCacheSize)."""
        ids = np.ascontiguousarray(input_data, dtype=np.int32)
        n = int(ids.size)
        punc = np.zeros(n, np.int32)
        C = self._lib.pfhip_punc_num_classes(self._h)
        logits = np.zeros((n, C), np.float32) if want_logits else None
        lp = logits.ctypes.data if want_logits else None
        if nCacheSize is None:
            _check(self._lib, self._lib.pfhip_punc_infer(self._h, ids.ctypes.data, n, punc.ctypes.data, lp))
        else:
            _check(self._lib, self._lib.pfhip_punc_infer_online(self._h, ids.ctypes.data, n, int(nCacheSize), punc.ctypes.data, lp))
        return (punc, logits) if want_logits else punc

    def InferBatch(self, sequences, nCacheSizes=None):
        """Several Infer calls as one packed device pass (pfhip_punc_infer_batch); nCacheSizes: per sequence, realtime model."""
        seqs = [np.ascontiguousarray(x, dtype=np.int32) for x in sequences]
        B = len(seqs)
        outs = [np.zeros(x.size, np.int32) for x in seqs]
        P = ctypes.c_void_p * B
        pi = P(*[x.ctypes.data for x in seqs])
        po = P(*[x.ctypes.data for x in outs])
        ns = (ctypes.c_int * B)(*[int(x.size) for x in seqs])
        cs = (ctypes.c_int * B)(*[int(c) for c in nCacheSizes]) if nCacheSizes is not None else None
        _check(self._lib, self._lib.pfhip_punc_infer_batch(self._h, pi, ns, cs, B, po))
        return outs

    def set_batching(self, wait_us, max_sequences):
        _check(self._lib, self._lib.pfhip_set_punc_batching(self._h, int(wait_us), int(max_sequences)))


class FsmnVadOnlineHip:
    """`funasr::FsmnVadOnline` (onnxruntime/src/fsmn-vad-online.cpp): one per connection, built on an FsmnVadHip like
    `FsmnVadOnline(FsmnVad*)`.  Infer(waves, input_finished) returns the (start_ms, end_ms) pairs of the online detector
    (-1 = open), as `FsmnVadOnline::Infer` does; InferScores stops before the scorer (scores + the waveform it would get)."""

    def __init__(self, vad: "FsmnVadHip", vad_silence_duration=800, vad_max_len=15000, vad_speech_noise_thres=0.8):
        self._lib = load_lib()
        self._vad = vad
        self._h = ctypes.c_void_p()
        _check(self._lib, self._lib.pfhip_vad_stream_create(vad._h, ctypes.byref(self._h)))
        self._scorer = E2EVadModelHost()
        self.vad_silence_duration_, self.vad_max_len_, self.vad_speech_noise_thres_ = vad_silence_duration, vad_max_len, vad_speech_noise_thres

    def close(self):
        if self._h:
            if self._vad._h:                  # a parent closed first (garbage collection order) took its streams' device state with it
                self._lib.pfhip_vad_stream_destroy(self._h)
            self._h = ctypes.c_void_p()
            self._scorer.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def SetConfig(self, vad_tail_sil, vad_max_len):
        self.vad_silence_duration_, self.vad_max_len_ = vad_tail_sil, vad_max_len

    def InferScores(self, waves, input_finished=False):
        x = np.ascontiguousarray(waves, dtype=np.float32)
        cap = x.size // 160 + 16
        sil = np.zeros(cap, np.float32)
        wv = np.zeros(x.size + 2048, np.float32)
        nf, nw = ctypes.c_int(0), ctypes.c_int(0)
        _check(self._lib, self._lib.pfhip_vad_stream_infer(self._h, x.ctypes.data if x.size else None, int(x.size),
                                                           1 if input_finished else 0, sil.ctypes.data, cap, ctypes.byref(nf),
                                                           wv.ctypes.data, wv.size, ctypes.byref(nw)))
        return sil[:nf.value].copy(), wv[:nw.value].copy()

    @staticmethod
    def InferScoresBatch(streams, waves, input_finished):
        """InferScores of several connections (of one FsmnVadHip) as one device pass; returns [(sil, waveform), ...]."""
        n = len(streams)
        lib = streams[0]._lib
        xs = [np.ascontiguousarray(w, dtype=np.float32) for w in waves]
        sils = [np.zeros(x.size // 160 + 16, np.float32) for x in xs]
        wvs = [np.zeros(x.size + 2048, np.float32) for x in xs]
        P = ctypes.c_void_p * n
        h = P(*[s._h.value for s in streams])
        px = P(*[x.ctypes.data if x.size else None for x in xs])
        ns = (ctypes.c_int * n)(*[int(x.size) for x in xs])
        fin = (ctypes.c_int * n)(*[1 if f else 0 for f in input_finished])
        ps = P(*[a.ctypes.data for a in sils])
        caps = (ctypes.c_size_t * n)(*[a.size for a in sils])
        pw = P(*[a.ctypes.data for a in wvs])
        wcaps = (ctypes.c_size_t * n)(*[a.size for a in wvs])
        nf, nw = (ctypes.c_int * n)(), (ctypes.c_int * n)()
        _check(lib, lib.pfhip_vad_stream_infer_batch(h, n, px, ns, fin, ps, caps, nf, pw, wcaps, nw))
        return [(sils[i][:nf[i]].copy(), wvs[i][:nw[i]].copy()) for i in range(n)]

    def Infer(self, waves, input_finished=False):
        sil, wv = self.InferScores(waves, input_finished)
        if sil.size == 0:
            return []                                                          # fsmn-vad-online.cpp:140-146
        return self._scorer(sil, wv, input_finished, True, self.vad_silence_duration_, self.vad_max_len_,
                            self.vad_speech_noise_thres_, 16000)

    def Reset(self):
        _check(self._lib, self._lib.pfhip_vad_stream_reset(self._h))


class E2EVadModelHost:
    """ctypes handle on the host-side end-point detector (csrc/host/vad_segmenter.cpp), call-compatible with
    `funasr::E2EVadModel::operator()` (onnxruntime/src/e2e-vad.h:303-362) except that it takes the class-0 score
    column instead of the whole score matrix."""

    def __init__(self):
        self._lib = load_lib()
        self._h = ctypes.c_void_p()
        _check(self._lib, self._lib.pfhip_vadseg_create(ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.pfhip_vadseg_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, score_sil, waveform, is_final=False, online=False, max_end_sil=800, max_single_segment_time=15000,
                 speech_noise_thres=0.8, sample_rate=16000):
        sc = np.ascontiguousarray(score_sil, dtype=np.float32)
        wv = np.ascontiguousarray(waveform, dtype=np.float32)
        cap = sc.size + 8
        segs = np.zeros((cap, 2), np.int32)
        n = ctypes.c_int(0)
        _check(self._lib, self._lib.pfhip_vadseg_feed(self._h, sc.ctypes.data if sc.size else None, int(sc.size),
                                                      wv.ctypes.data if wv.size else None, int(wv.size), int(is_final),
                                                      int(online), int(max_end_sil), int(max_single_segment_time),
                                                      float(speech_noise_thres), int(sample_rate), segs.ctypes.data, cap,
                                                      ctypes.byref(n)))
        return [[int(a), int(b)] for a, b in segs[:n.value]]


def timestamp_onnx(us_alphas, us_cif_peak, n_chars, begin_time=0.0, total_offset=-1.5):
    """`funasr::TimestampOnnx` (onnxruntime/src/util.cpp:838-963) through the C ABI: [(begin_s, end_s, is_sil)]."""
    lib = load_lib()
    a = np.ascontiguousarray(us_alphas, dtype=np.float32).copy()
    p = np.ascontiguousarray(us_cif_peak, dtype=np.float32)
    cap = 2 * (n_chars + 4) + 8
    spans = np.zeros((cap, 3), np.float32)
    n = ctypes.c_int(0)
    _check(lib, lib.pfhip_timestamp_onnx(a.ctypes.data, p.ctypes.data, int(p.size), int(n_chars), float(begin_time),
                                         float(total_offset), spans.ctypes.data, cap, ctypes.byref(n)))
    return [(float(s[0]), float(s[1]), bool(s[2])) for s in spans[:n.value]]
