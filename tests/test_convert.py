"""Real-weight loader (f2): the UPSTREAM-name mapping is self-consistent — synthetic weights pushed through the upstream
state_dict naming and layouts come back as the same container.  (No real checkpoint exists offline: names unverified.)"""
import importlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def mods(pkg):
    return importlib.import_module(pkg.__name__ + ".convert"), importlib.import_module(pkg.__name__ + ".weights")


def to_state(man, blob, name_map, expand):
    state = {}
    for name, meta in man["tensors"].items():
        if name not in name_map:
            continue
        n = int(np.prod(meta["shape"]))
        arr = blob[meta["offset"] // 4: meta["offset"] // 4 + n].reshape(meta["shape"]).copy()
        if name.endswith("fsmn.w"):
            arr = expand(arr)
        state[name_map[name]] = arr
    return state


def test_paraformer_round_trip(mods):
    conv, wt = mods
    cfg = wt.small_config()
    man, blob = wt.synth_weights(cfg, seed=5)
    state = to_state(man, blob, conv.paraformer_name_map(cfg), lambda a: a[:, None, :])      # Conv1d depthwise [d,1,k]
    t = man["tensors"]
    get = lambda n: blob[t[n]["offset"] // 4: t[n]["offset"] // 4 + int(np.prod(t[n]["shape"]))]
    man2, blob2 = conv.convert_paraformer(state, cfg, get("cmvn.mean"), get("cmvn.istd"))
    assert man2["tensors"] == man["tensors"] and man2["config"]["vocab"] == cfg["vocab"]
    assert np.array_equal(blob2, blob)
    del state["decoder.decoders.1.src_attn.linear_k_v.bias"]
    with pytest.raises(KeyError):
        conv.convert_paraformer(state, cfg, get("cmvn.mean"), get("cmvn.istd"))


def test_vad_and_punc_round_trip(mods):
    conv, wt = mods
    man, blob = wt.synth_vad_weights(seed=6)
    cfg = man["config"]
    t = man["tensors"]
    get = lambda n: blob[t[n]["offset"] // 4: t[n]["offset"] // 4 + int(np.prod(t[n]["shape"]))]
    state = to_state(man, blob, conv.vad_name_map(cfg), lambda a: a[:, None, :, None])      # Conv2d [p,1,lorder,1]
    man2, blob2 = conv.convert_vad(state, get("cmvn.mean"), get("cmvn.istd"))
    assert man2["tensors"] == t and np.array_equal(blob2, blob)
    pc = dict(wt.CT_TRANSFORMER, vocab=500)
    man, blob = wt.synth_punc_weights(pc, seed=7)
    state = to_state(man, blob, conv.punc_name_map(pc), lambda a: a[:, None, :])
    man2, blob2 = conv.convert_punc(state)
    assert man2["tensors"] == man["tensors"] and man2["config"]["vocab"] == 500 and np.array_equal(blob2, blob)


def test_parse_am_mvn(mods):
    conv, _ = mods
    text = "<Nnet>\n<Splice> 560 560\n[ 0 ]\n<AddShift> 560 560\n<LearnRateCoef> 0 [ -8.5 -9.25 -7 ]\n<Rescale> 560 560\n<LearnRateCoef> 0 [ 0.25 0.5 0.125 ]\n</Nnet>\n"
    shift, rescale = conv.parse_am_mvn(text)
    assert shift.tolist() == [-8.5, -9.25, -7.0] and rescale.tolist() == [0.25, 0.5, 0.125]
    with pytest.raises(ValueError):
        conv.parse_am_mvn("<Nnet>\n</Nnet>\n")
