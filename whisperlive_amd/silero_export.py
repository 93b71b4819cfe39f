"""Silero-VAD weight ingestion: ``silero_vad.onnx`` -> the fifteen float32 arrays ``csrc/vad.hip`` consumes.

The reference obtains its voice-activity model as an ONNX file (download at whisper_live/vad.py:112-128; faster-whisper
bundles the same network) and runs it through onnxruntime. Neither ``onnx`` nor ``onnxruntime`` exists in this
deployment, and the HIP kernels only need the weights, so this module reads the file with a ~100-line protobuf walk
(stdlib only): every ``TensorProto`` of the model — graph initializers and ``Constant`` node values, recursively through
the ``If`` branches that hold the 16 kHz and 8 kHz sub-networks — together with the ``Conv`` / ``LSTM`` nodes that
consume them, and maps them to ``vad.SILERO_SHAPES``:

* the scope that owns the 258x256 STFT filter bank is the 16 kHz network (the 8 kHz branch has a 130x128 one);
* ``Conv`` nodes of that scope pair each weight with its bias: (128,129,3), (64,128,3), (64,64,3), (128,64,3) are the
  encoder, (1,128,1) the output layer;
* the recurrent cell is taken from an ``LSTM`` node (ONNX gate order i,o,f,c -> re-ordered to the i,f,g,o rows the
  kernel and ``torch.nn.LSTMCell`` use) or, when the exporter kept PyTorch parameter names, from the tensors named
  ``*weight_ih* / *weight_hh* / *bias_ih* / *bias_hh*``.

``python -m whisperlive_amd.silero_export silero_vad.onnx silero_vad.npz`` writes the archive ``WLX_SILERO_VAD_NPZ`` /
``--vad_weights`` accept; ``vad.configure()`` and ``WLX_SILERO_VAD_ONNX`` call ``silero_weights_from_onnx`` directly.
Anything ambiguous raises with the candidate list instead of guessing. Tested on synthetic ONNX files written by
tests/test_silero_export.py in both forms (no real Silero file exists offline — the mapping of a real file is the one
thing this cannot prove, and the error messages are written for that day).
"""
from __future__ import annotations

import struct
import sys
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np


# ---- protobuf wire format -------------------------------------------------------------------------------------------
def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf: memoryview) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) — value: int for varint / fixed, memoryview for length-delimited"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
            if pos > n:
                raise ValueError("truncated protobuf field")
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, v


def _packed_varints(v) -> List[int]:
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def _sint64(x: int) -> int:
    return x - (1 << 64) if x >= 1 << 63 else x


# ---- ONNX messages (only what is needed) ------------------------------------------------------------------------------
_DTYPES = {1: np.float32, 7: np.int64, 10: np.float16, 11: np.float64, 6: np.int32}


def _tensor(buf: memoryview) -> Tuple[str, Optional[np.ndarray]]:
    dims: List[int] = []
    dtype = 0
    name = ""
    raw = None
    floats: List[float] = []
    doubles: List[float] = []
    ints: List[int] = []
    for num, wt, v in _fields(buf):
        if num == 1:
            dims.extend(_packed_varints(v) if wt == 2 else [v])
        elif num == 2:
            dtype = v
        elif num == 8:
            name = bytes(v).decode("utf-8", "replace")
        elif num == 9:
            raw = bytes(v)
        elif num == 4:
            floats.extend(np.frombuffer(bytes(v), "<f4").tolist() if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]])
        elif num == 10:
            doubles.extend(np.frombuffer(bytes(v), "<f8").tolist() if wt == 2 else [struct.unpack("<d", struct.pack("<Q", v))[0]])
        elif num in (5, 7):
            ints.extend([_sint64(x) for x in _packed_varints(v)] if wt == 2 else [_sint64(v)])
    if dtype not in _DTYPES:
        return name, None                              # strings, bools, ... : nothing the VAD needs
    np_dtype = _DTYPES[dtype]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(np_dtype).newbyteorder("<")).astype(np_dtype)
    elif floats:
        arr = np.asarray(floats, np_dtype)
    elif doubles:
        arr = np.asarray(doubles, np_dtype)
    elif dtype == 10 and ints:                          # fp16 bit patterns travel in int32_data
        arr = np.asarray(ints, np.uint16).view(np.float16)
    else:
        arr = np.asarray(ints, np_dtype)
    n = int(np.prod(dims)) if dims else arr.size
    if arr.size != n:
        return name, None                              # externally stored or malformed
    return name, arr.reshape(dims) if dims else arr.reshape(())


@dataclass
class Node:
    op: str
    inputs: List[str]
    outputs: List[str]
    name: str
    scope: str


@dataclass
class Model:
    tensors: Dict[Tuple[str, str], np.ndarray] = field(default_factory=dict)      # (scope, name) -> array
    nodes: List[Node] = field(default_factory=list)

    def lookup(self, scope: str, name: str) -> Optional[np.ndarray]:
        """name resolution of ONNX sub-graphs: innermost scope first, then the enclosing graphs"""
        while True:
            a = self.tensors.get((scope, name))
            if a is not None or not scope:
                return a
            scope = scope.rsplit("/", 1)[0] if "/" in scope else ""


def _graph(buf: memoryview, scope: str, model: Model):
    for num, _wt, v in _fields(buf):
        if num == 5:                                                   # initializer
            name, arr = _tensor(v)
            if arr is not None:
                model.tensors[(scope, name)] = arr
        elif num == 1:                                                 # node
            op = nname = ""
            ins: List[str] = []
            outs: List[str] = []
            attrs = []
            for n2, _w2, v2 in _fields(v):
                if n2 == 1:
                    ins.append(bytes(v2).decode())
                elif n2 == 2:
                    outs.append(bytes(v2).decode())
                elif n2 == 3:
                    nname = bytes(v2).decode()
                elif n2 == 4:
                    op = bytes(v2).decode()
                elif n2 == 5:
                    attrs.append(v2)
            model.nodes.append(Node(op, ins, outs, nname, scope))
            for ai, a in enumerate(attrs):
                aname = ""
                t = None
                graphs = []
                for n3, _w3, v3 in _fields(a):
                    if n3 == 1:
                        aname = bytes(v3).decode()
                    elif n3 == 5:
                        t = v3
                    elif n3 in (6, 11):
                        graphs.append(v3)
                if t is not None and op == "Constant" and outs:
                    _name, arr = _tensor(t)
                    if arr is not None:
                        model.tensors[(scope, outs[0])] = arr
                for gi, g in enumerate(graphs):
                    sub = f"{scope}/{nname or op}{len(model.nodes)}.{aname or ai}.{gi}".lstrip("/")
                    _graph(g, sub, model)


def read_onnx(path: str) -> Model:
    with open(path, "rb") as f:
        data = memoryview(f.read())
    model = Model()
    found = False
    for num, wt, v in _fields(data):
        if num == 7 and wt == 2:                                       # ModelProto.graph
            _graph(v, "", model)
            found = True
    if not found:
        raise ValueError(f"{path}: no graph in the file (not an ONNX ModelProto?)")
    return model


# ---- mapping to the kernel's weight set --------------------------------------------------------------------------------
_ENC_SHAPES = [(128, 129, 3), (64, 128, 3), (64, 64, 3), (128, 64, 3)]


def _in_scope(scope: str, root: str) -> bool:
    """is `scope` the 16 kHz scope or one of the graphs around it (whose tensors it can see)?"""
    return scope == root or root.startswith(scope + "/") or scope == ""


def silero_weights(model: Model) -> Dict[str, np.ndarray]:
    stft = [(sc, nm, a) for (sc, nm), a in model.tensors.items() if a.size == 258 * 256 and a.shape[0] == 258]
    if not stft:
        shapes = sorted({a.shape for a in model.tensors.values() if a.ndim >= 2})
        raise ValueError(f"no 258x256 STFT filter bank in the file — not Silero VAD v5/v6 at 16 kHz? (tensor shapes: {shapes[:12]})")
    root = stft[0][0]
    if any(sc != root for sc, _n, _a in stft):
        raise ValueError(f"several STFT filter banks in different scopes: {[(sc, nm) for sc, nm, _ in stft]}")
    out: Dict[str, np.ndarray] = {"stft_basis": stft[0][2].reshape(258, 256).astype(np.float32)}
    nodes = [n for n in model.nodes if n.scope == root or n.scope.startswith(root + "/")]

    # Conv nodes pair a weight with its bias
    convs = []
    for n in nodes:
        if n.op == "Conv" and len(n.inputs) >= 2:
            w = model.lookup(n.scope, n.inputs[1])
            b = model.lookup(n.scope, n.inputs[2]) if len(n.inputs) > 2 and n.inputs[2] else None
            if w is not None:
                convs.append((tuple(w.shape), w, b, n))
    for i, shape in enumerate(_ENC_SHAPES):
        hits = [c for c in convs if c[0] == shape]
        if len(hits) != 1:
            by_name = _named(model, root, f"encoder.{i}.", shape)
            if by_name is None:
                raise ValueError(f"encoder conv {i}: expected one Conv with weight {shape}, found {len(hits)} "
                                 f"(Conv weights in scope: {[c[0] for c in convs]})")
            w, b = by_name
        else:
            _s, w, b, _n = hits[0]
        if b is None:
            raise ValueError(f"encoder conv {i} has no bias input")
        out[f"enc{i}_w"], out[f"enc{i}_b"] = w.astype(np.float32), b.reshape(-1).astype(np.float32)
    hits = [c for c in convs if c[0] == (1, 128, 1)]
    if len(hits) != 1 or hits[0][2] is None:
        raise ValueError(f"output layer: expected one Conv with weight (1,128,1) and a bias, found {len(hits)}")
    out["out_w"], out["out_b"] = hits[0][1].reshape(128).astype(np.float32), hits[0][2].reshape(1).astype(np.float32)

    # recurrent cell
    lstm = [n for n in nodes if n.op == "LSTM"]
    if len(lstm) == 1:
        n = lstm[0]
        W, R = model.lookup(n.scope, n.inputs[1]), model.lookup(n.scope, n.inputs[2])
        B = model.lookup(n.scope, n.inputs[3]) if len(n.inputs) > 3 and n.inputs[3] else None
        if W is None or R is None or W.shape != (1, 512, 128) or R.shape != (1, 512, 128):
            raise ValueError(f"LSTM node: W / R are {None if W is None else W.shape} / {None if R is None else R.shape}, expected (1,512,128)")
        perm = np.concatenate([np.arange(0, 128), np.arange(256, 384), np.arange(384, 512), np.arange(128, 256)])   # iofc -> ifgo
        out["lstm_w_ih"], out["lstm_w_hh"] = W[0][perm].astype(np.float32), R[0][perm].astype(np.float32)
        if B is None:
            out["lstm_b_ih"] = np.zeros(512, np.float32)
            out["lstm_b_hh"] = np.zeros(512, np.float32)
        else:
            if B.shape != (1, 1024):
                raise ValueError(f"LSTM node: B is {B.shape}, expected (1,1024)")
            out["lstm_b_ih"], out["lstm_b_hh"] = B[0, :512][perm].astype(np.float32), B[0, 512:][perm].astype(np.float32)
    elif not lstm:
        want = {"lstm_w_ih": ("weight_ih", (512, 128)), "lstm_w_hh": ("weight_hh", (512, 128)),
                "lstm_b_ih": ("bias_ih", (512,)), "lstm_b_hh": ("bias_hh", (512,))}
        for key, (frag, shape) in want.items():
            hits = [(sc, nm, a) for (sc, nm), a in model.tensors.items()
                    if frag in nm and tuple(a.shape) == shape and _in_scope(sc, root)]
            if len(hits) != 1:
                cands = [(nm, a.shape) for (sc, nm), a in model.tensors.items() if a.size in (512, 512 * 128) and _in_scope(sc, root)]
                raise ValueError(f"recurrent cell: no LSTM node and {len(hits)} tensors named *{frag}* of shape {shape}; "
                                 f"candidates by size: {cands}. If the exporter decomposed the cell into anonymous MatMul/Gemm "
                                 "weights, pass an .npz with the names of vad.SILERO_SHAPES instead.")
            out[key] = hits[0][2].astype(np.float32)
    else:
        raise ValueError(f"{len(lstm)} LSTM nodes in the 16 kHz scope")
    return out


def _named(model: Model, root: str, frag: str, shape) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """fallback when Conv nodes do not resolve uniquely: PyTorch parameter names (…encoder.N.….weight / .bias)"""
    ws = [(nm, a) for (sc, nm), a in model.tensors.items() if frag in nm and nm.endswith("weight") and tuple(a.shape) == shape and _in_scope(sc, root)]
    if len(ws) != 1:
        return None
    bname = ws[0][0][: -len("weight")] + "bias"
    bs = [a for (sc, nm), a in model.tensors.items() if nm == bname and _in_scope(sc, root)]
    return (ws[0][1], bs[0]) if len(bs) == 1 else None


def silero_weights_from_onnx(path: str) -> Dict[str, np.ndarray]:
    return silero_weights(read_onnx(path))


def main(argv: List[str]) -> int:
    if len(argv) != 3:
        print("usage: python -m whisperlive_amd.silero_export silero_vad.onnx out.npz", file=sys.stderr)
        return 2
    from .vad import check_silero_weights
    w = check_silero_weights(silero_weights_from_onnx(argv[1]))
    np.savez(argv[2], **w)
    print(f"wrote {argv[2]}: " + ", ".join(f"{k}{list(v.shape)}" for k, v in w.items()))
    return 0


if __name__ == "__main__":
    raise SystemExit(main(sys.argv))
