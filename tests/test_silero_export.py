"""Silero-VAD weight ingestion (whisperlive_amd/silero_export.py): the stdlib protobuf walk and the ONNX -> kernel weight
mapping, on synthetic ONNX files this test writes itself with a minimal protobuf WRITER (no onnx package offline, no
real silero_vad.onnx either). Two exporter forms: PyTorch-named initializers inside ``If`` branches with the recurrent
cell kept as named tensors, and an ``LSTM`` node with ONNX's gate order and ``Constant``-node weights."""
import struct

import numpy as np
import pytest

from oracle import silero_vad as osv
from whisperlive_amd import silero_export as sx
from whisperlive_amd import vad


# ---- minimal protobuf writer ------------------------------------------------------------------------------------------
def _vi(x):
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(num, payload):
    return _vi(num << 3 | 2) + _vi(len(payload)) + payload


def _iv(num, x):
    return _vi(num << 3) + _vi(x)


def _tensor(name, a, raw=True, packed_dims=False):
    a = np.ascontiguousarray(a, np.float32)
    dims = b"".join(_vi(d) for d in a.shape)
    msg = (_ld(1, dims) if packed_dims else b"".join(_iv(1, d) for d in a.shape)) + _iv(2, 1)
    msg += _ld(9, a.tobytes()) if raw else _ld(4, a.tobytes())           # raw_data / packed float_data
    return msg + _ld(8, name.encode())


def _node(op, ins, outs, name="", attrs=()):
    msg = b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs)
    msg += _ld(3, name.encode()) + _ld(4, op.encode()) + b"".join(_ld(5, a) for a in attrs)
    return msg


def _attr_graph(name, g):
    return _ld(1, name.encode()) + _ld(6, g) + _iv(20, 5)


def _attr_tensor(name, t):
    return _ld(1, name.encode()) + _ld(5, t) + _iv(20, 4)


def _graph(nodes=(), inits=(), name="g"):
    return b"".join(_ld(1, n) for n in nodes) + _ld(2, name.encode()) + b"".join(_ld(5, t) for t in inits)


def _model(graph):
    return _iv(1, 8) + _ld(2, b"test-writer") + _ld(7, graph)


def _branch_8k(seed):
    """the 8 kHz sub-network: same conv shapes except the first, and its own 130x128 filter bank — must be ignored"""
    r = np.random.default_rng(seed)
    t = lambda *s: r.standard_normal(s).astype(np.float32)       # noqa: E731
    inits = [_tensor("_model.stft.forward_basis_buffer", t(130, 1, 128)),
             _tensor("_model.encoder.0.reparam_conv.weight", t(128, 65, 3)), _tensor("_model.encoder.0.reparam_conv.bias", t(128)),
             _tensor("_model.encoder.1.reparam_conv.weight", t(64, 128, 3)), _tensor("_model.encoder.1.reparam_conv.bias", t(64)),
             _tensor("_model.encoder.2.reparam_conv.weight", t(64, 64, 3)), _tensor("_model.encoder.2.reparam_conv.bias", t(64)),
             _tensor("_model.encoder.3.reparam_conv.weight", t(128, 64, 3)), _tensor("_model.encoder.3.reparam_conv.bias", t(128)),
             _tensor("_model.decoder.rnn.weight_ih", t(512, 128)), _tensor("_model.decoder.rnn.weight_hh", t(512, 128)),
             _tensor("_model.decoder.rnn.bias_ih", t(512)), _tensor("_model.decoder.rnn.bias_hh", t(512)),
             _tensor("_model.decoder.decoder.2.weight", t(1, 128, 1)), _tensor("_model.decoder.decoder.2.bias", t(1))]
    nodes = [_node("Conv", ["x", "_model.stft.forward_basis_buffer"], ["s"], "stft8")]
    for i in range(4):
        nodes.append(_node("Conv", [f"h{i}", f"_model.encoder.{i}.reparam_conv.weight", f"_model.encoder.{i}.reparam_conv.bias"], [f"h{i + 1}"], f"c8_{i}"))
    nodes.append(_node("Conv", ["r", "_model.decoder.decoder.2.weight", "_model.decoder.decoder.2.bias"], ["o"], "out8"))
    return _graph(nodes, inits, "else8k")


def _write_named_form(path, w):
    """If(sr == 16000) { 16 kHz net } else { 8 kHz net }; recurrent cell as PyTorch-named tensors; mixed encodings"""
    inits = [_tensor("_model.stft.forward_basis_buffer", w["stft_basis"].reshape(258, 1, 256)),
             _tensor("_model.decoder.rnn.weight_ih", w["lstm_w_ih"], raw=False), _tensor("_model.decoder.rnn.weight_hh", w["lstm_w_hh"]),
             _tensor("_model.decoder.rnn.bias_ih", w["lstm_b_ih"], packed_dims=True), _tensor("_model.decoder.rnn.bias_hh", w["lstm_b_hh"]),
             _tensor("_model.decoder.decoder.2.weight", w["out_w"].reshape(1, 128, 1)), _tensor("_model.decoder.decoder.2.bias", w["out_b"])]
    nodes = [_node("Conv", ["x", "_model.stft.forward_basis_buffer"], ["s"], "stft")]
    for i in range(4):
        inits += [_tensor(f"_model.encoder.{i}.reparam_conv.weight", w[f"enc{i}_w"], packed_dims=bool(i & 1)),
                  _tensor(f"_model.encoder.{i}.reparam_conv.bias", w[f"enc{i}_b"])]
        nodes.append(_node("Conv", [f"h{i}", f"_model.encoder.{i}.reparam_conv.weight", f"_model.encoder.{i}.reparam_conv.bias"], [f"h{i + 1}"], f"c{i}"))
    nodes.append(_node("Conv", ["r", "_model.decoder.decoder.2.weight", "_model.decoder.decoder.2.bias"], ["o"], "out"))
    then16 = _graph(nodes, inits, "then16k")
    top = _graph([_node("Equal", ["sr", "c16000"], ["is16"], "eq"),
                  _node("If", ["is16"], ["output", "stateN"], "If_0", [_attr_graph("then_branch", then16), _attr_graph("else_branch", _branch_8k(99))])],
                 [_tensor("unrelated", np.zeros((3, 3), np.float32))], "main")
    with open(path, "wb") as f:
        f.write(_model(top))


def _write_lstm_form(path, w):
    """flat graph, anonymous tensor names, weights as Constant nodes, recurrent cell as an ONNX LSTM node (gate order iofc)"""
    inv = np.concatenate([np.arange(0, 128), np.arange(384, 512), np.arange(128, 256), np.arange(256, 384)])   # ifgo -> iofc
    W = w["lstm_w_ih"][inv][None]
    R = w["lstm_w_hh"][inv][None]
    B = np.concatenate([w["lstm_b_ih"][inv], w["lstm_b_hh"][inv]])[None]
    consts = {"onnx::Conv_1": w["stft_basis"].reshape(258, 1, 256), "onnx::LSTM_W": W, "onnx::LSTM_R": R, "onnx::LSTM_B": B,
              "onnx::Conv_o": w["out_w"].reshape(1, 128, 1), "onnx::Conv_ob": w["out_b"]}
    nodes = []
    for i in range(4):
        consts[f"onnx::Conv_w{i}"] = w[f"enc{i}_w"]
        consts[f"onnx::Conv_b{i}"] = w[f"enc{i}_b"]
    for k, v in consts.items():
        nodes.append(_node("Constant", [], [k], "", [_attr_tensor("value", _tensor("", v))]))
    nodes.append(_node("Conv", ["x", "onnx::Conv_1"], ["s"]))
    for i in range(4):
        nodes.append(_node("Conv", [f"h{i}", f"onnx::Conv_w{i}", f"onnx::Conv_b{i}"], [f"h{i + 1}"]))
    nodes.append(_node("LSTM", ["h4", "onnx::LSTM_W", "onnx::LSTM_R", "onnx::LSTM_B", "", "h0", "c0"], ["y", "hn", "cn"]))
    nodes.append(_node("Conv", ["r", "onnx::Conv_o", "onnx::Conv_ob"], ["o"]))
    with open(path, "wb") as f:
        f.write(_model(_graph(nodes, [], "flat")))


@pytest.mark.parametrize("form", ["named_if_branches", "lstm_node_constants"])
def test_onnx_to_kernel_weights_roundtrip(tmp_path, form):
    w = osv.random_weights(5)
    path = str(tmp_path / "silero_vad.onnx")
    (_write_named_form if form == "named_if_branches" else _write_lstm_form)(path, w)
    got = vad.check_silero_weights(sx.silero_weights_from_onnx(path))
    assert set(got) == set(vad.SILERO_SHAPES)
    for k in vad.SILERO_SHAPES:
        np.testing.assert_array_equal(got[k], np.asarray(w[k], np.float32).reshape(vad.SILERO_SHAPES[k]), err_msg=k)
    # the exported weights drive the restated network to the same probabilities as the originals
    pcm = (0.1 * np.random.default_rng(1).standard_normal(512 * 12)).astype(np.float32)
    np.testing.assert_array_equal(osv.speech_probs(got, pcm), osv.speech_probs(w, pcm))
    # CLI: .onnx -> .npz that load_silero_npz accepts
    out = str(tmp_path / "silero.npz")
    assert sx.main(["prog", path, out]) == 0
    z = vad.load_silero_npz(out)
    np.testing.assert_array_equal(z["lstm_w_hh"], w["lstm_w_hh"])


def test_refuses_what_it_cannot_map(tmp_path):
    p = str(tmp_path / "x.onnx")
    with open(p, "wb") as f:
        f.write(_model(_graph([], [_tensor("a", np.zeros((4, 4)))])))
    with pytest.raises(ValueError, match="STFT filter bank"):
        sx.silero_weights_from_onnx(p)
    with open(p, "wb") as f:
        f.write(b"\x00\x01garbage")
    with pytest.raises(ValueError):
        sx.silero_weights_from_onnx(p)
    # decomposed recurrent cell with anonymous weights: ambiguous -> explicit error, no guess
    w = osv.random_weights(2)
    inits = [_tensor("b", w["stft_basis"].reshape(258, 1, 256)), _tensor("m1", w["lstm_w_ih"]), _tensor("m2", w["lstm_w_hh"]),
             _tensor("ow", w["out_w"].reshape(1, 128, 1)), _tensor("ob", w["out_b"])]
    nodes = [_node("Conv", ["r", "ow", "ob"], ["o"])]
    for i in range(4):
        inits += [_tensor(f"w{i}", w[f"enc{i}_w"]), _tensor(f"b{i}", w[f"enc{i}_b"])]
        nodes.append(_node("Conv", ["h", f"w{i}", f"b{i}"], ["h"]))
    with open(p, "wb") as f:
        f.write(_model(_graph(nodes, inits)))
    with pytest.raises(ValueError, match="recurrent cell"):
        sx.silero_weights_from_onnx(p)


def test_vad_default_is_refusal_not_a_silent_stand_in(monkeypatch):
    for k in ("WLX_SILERO_VAD_NPZ", "WLX_SILERO_VAD_ONNX", "WLX_ALLOW_VAD_STANDIN"):
        monkeypatch.delenv(k, raising=False)
    vad.set_default_model(None)
    with pytest.raises(vad.VadUnavailable, match="WLX_SILERO_VAD_ONNX"):
        vad.get_default_model()
    with pytest.raises(vad.VadUnavailable):
        vad.get_speech_timestamps(np.zeros(16000, np.float32))
    monkeypatch.setenv("WLX_ALLOW_VAD_STANDIN", "1")
    assert isinstance(vad.get_default_model(), vad.EnergyGateModel)
    vad.set_default_model(None)
