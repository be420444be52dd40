"""Builder-authored writers of the reference's model FILE FORMATS, for the import tests (tests/test_import.py).

No real `deberta.onnx`, `model_<name>.onnx` or `.sbv2` exists in the build environment and neither `onnx` nor `zstandard` is installed, so
these functions emit the formats by hand from their specifications:
  * ONNX ModelProto (protobuf wire format; only what an importer of WEIGHTS looks at: graph.initializer, graph.node with op type / inputs /
    outputs / attributes), following what `torch.onnx.export` + `onnxsim` (scripts/convert/convert_model.py:113-156,
    convert_deberta.py:40-52) are documented to produce:
      - parameters that reach an op unchanged keep their state-dict name (Conv weight / bias, Embedding tables, LayerNorm gamma / beta);
      - a Linear applied to a 3-D input becomes MatMul(x, W^T) + Add(bias): W^T is an anonymous constant `onnx::MatMul_<n>`;
      - weight_norm'ed convolutions (every HiFi-GAN conv) either keep their `weight_g` / `weight_v` pair (`folded=False`) or have been
        constant-folded by onnxsim into an anonymous `onnx::Conv_<n>` whose Conv node still takes the named bias (`folded=True`).
  * `.sbv2` = zstd(tar{version.txt, model.onnx, style_vectors.json}) exactly as convert_model.py:157-175 writes it (zstd through ctypes on
    the image's libzstd.so.1).
PARITY STATUS: unpinned (no real file to compare with); the files are as faithful to the exporters' conventions as their documentation allows.
"""
import ctypes
import io
import json
import struct
import tarfile

import numpy as np


def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wt: int) -> bytes:
    return _varint((field << 3) | wt)


def _ld(field: int, payload: bytes) -> bytes:
    return _key(field, 2) + _varint(len(payload)) + payload


def tensor_proto(name: str, a: np.ndarray, raw=True) -> bytes:
    a = np.ascontiguousarray(a)
    dt = {np.dtype("float32"): 1, np.dtype("int64"): 7, np.dtype("float16"): 10}[a.dtype]
    out = b"".join(_key(1, 0) + _varint(int(d)) for d in a.shape)       # dims (unpacked, like most exporters write them)
    out += _key(2, 0) + _varint(dt)
    if raw or dt != 1:
        out += _ld(9, a.tobytes())
    else:
        out += _ld(4, a.astype("<f4").tobytes())                          # packed float_data
    return out + _ld(8, name.encode())


def attr_ints(name: str, vals) -> bytes:
    return _ld(1, name.encode()) + b"".join(_key(8, 0) + _varint(int(v)) for v in vals) + _key(20, 0) + _varint(7)


def attr_float(name: str, v: float) -> bytes:
    return _ld(1, name.encode()) + _key(2, 5) + struct.pack("<f", v) + _key(20, 0) + _varint(1)


def node_proto(op: str, inputs, outputs, attrs=(), name="") -> bytes:
    out = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    if name:
        out += _ld(3, name.encode())
    out += _ld(4, op.encode())
    return out + b"".join(_ld(5, a) for a in attrs)


def model_proto(nodes, initializers, metadata=None) -> bytes:
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"main_graph") + b"".join(_ld(5, t) for t in initializers)
    out = _key(1, 0) + _varint(8) + _ld(2, b"pytorch") + _ld(3, b"2.5.0") + _ld(7, graph)
    for k, v in (metadata or {}).items():      # ModelProto.metadata_props = 14: StringStringEntryProto {key = 1, value = 2}
        out += _ld(14, _ld(1, k.encode()) + _ld(2, v if isinstance(v, bytes) else v.encode()))
    return out


def add_metadata(model: bytes, metadata: dict) -> bytes:
    """Appends metadata_props entries to a serialized ModelProto (protobuf messages concatenate)."""
    return model + model_proto([], [], metadata)[len(model_proto([], [])):]


def aivm_style_vectors(a, fortran=False, version=1) -> str:
    """The value AivisSpeech's converter stores under "aivm_style_vectors": base64 of np.save(style_vectors) (tts.rs:94-108 reads it back
    with npyz).  Written by hand so that both memory orders and both header versions can be produced."""
    import base64
    a = np.asarray(a, np.float32)
    hdr = "{'descr': '<f4', 'fortran_order': %s, 'shape': (%s), }" % ("True" if fortran else "False", ", ".join(str(d) for d in a.shape) + ("," if a.ndim == 1 else ""))
    pre = 10 if version == 1 else 12
    pad = 64 - (pre + len(hdr) + 1) % 64
    hdr = hdr + " " * pad + "\n"
    head = b"\x93NUMPY" + bytes([version, 0]) + (struct.pack("<H", len(hdr)) if version == 1 else struct.pack("<I", len(hdr)))
    return base64.b64encode(head + hdr.encode("latin1") + a.tobytes(order="F" if fortran else "C")).decode()


class _Anon:
    def __init__(self):
        self.n = 1000

    def __call__(self, op):
        self.n += 1
        return f"onnx::{op}_{self.n}"


def vits_onnx(W: dict, cfg: dict, folded: bool, raw=True, fold_affine=None) -> bytes:
    """model_<name>.onnx of the JP-Extra synthesizer holding the weights `W` (state-dict names).  folded: the onnxsim form (convert_model.py:156):
    weight_norm products and (fold_affine, default = folded) exp(-sdp.flows.0.logs) are anonymous constants."""
    fold_affine = folded if fold_affine is None else fold_affine
    anon = _Anon()
    inits, nodes = [], []
    nk = len(cfg["res_kernels"])
    wn = {"dec.conv_pre", "dec.conv_post"} | {f"dec.ups.{i}" for i in range(len(cfg["up_rates"]))}
    for name in W:
        if name.startswith("dec.resblocks.") and name.endswith(".weight"):
            wn.add(name[:-7])
    ti = 0
    for name, a in W.items():
        a = np.asarray(a, np.float32)
        base = name[:-7] if name.endswith(".weight") else None
        if fold_affine and name == "sdp.flows.0.logs":
            # onnxsim folds exp(-log_scale) of the first flow (ElementwiseAffine, reverse) into an anonymous constant behind the Sub of the translate vector
            en = anon("Mul")
            inits.append(tensor_proto(en, np.exp(-a).astype(np.float32), raw))
            nodes.append(node_proto("Sub", ["/sdp/z", "sdp.flows.0.m"], ["/sdp/sub"]))
            nodes.append(node_proto("Mul", ["/sdp/sub", en], ["/sdp/aff"]))
            continue
        if base in wn:
            is_t = base.startswith("dec.ups.")
            i_up = int(base.split(".")[2]) if is_t else 0
            attrs = [attr_ints("dilations", [1]), attr_ints("group", [1])]
            if is_t:
                attrs.append(attr_ints("strides", [cfg["up_rates"][i_up]]))
            elif base.startswith("dec.resblocks.") and ".convs1." in base:
                j, q = int(base.split(".")[2]) % nk, int(base.split(".")[4])
                attrs[0] = attr_ints("dilations", [cfg["res_dilations"][j][q]])
            bias = base + ".bias" if base + ".bias" in W else ""
            if folded:
                wname = anon("ConvTranspose" if is_t else "Conv")
                inits.append(tensor_proto(wname, a, raw))
            else:
                # weight_norm (dim 0): weight = g * v / ||v||; any (g, v) with that product is a valid file: scale v per channel
                rows = a.shape[0]
                scale = (1.0 + 0.25 * np.cos(np.arange(rows))).astype(np.float32).reshape((rows,) + (1,) * (a.ndim - 1))
                v = a * scale
                g = np.sqrt((a.astype(np.float64) ** 2).sum(axis=tuple(range(1, a.ndim)), keepdims=True)).astype(np.float32)
                inits.append(tensor_proto(base + ".weight_g", g, raw))
                inits.append(tensor_proto(base + ".weight_v", v, raw))
                wname = f"/dec/w_{ti}"          # the in-graph product; not an initializer
            ti += 1
            nodes.append(node_proto("ConvTranspose" if is_t else "Conv", [f"/x_{ti}", wname] + ([bias] if bias else []), [f"/y_{ti}"], attrs))
        else:
            inits.append(tensor_proto(name, a, raw))
    # noise of the kind onnxsim leaves behind: integer shape constants and a scalar
    inits.append(tensor_proto(anon("Reshape"), np.array([0, -1, 2], np.int64)))
    inits.append(tensor_proto("/Constant_7_output_0", np.array(0.5, np.float32)))
    return model_proto(nodes, inits)


def folded_positions(W: dict, cfg: dict) -> dict:
    """What onnxsim leaves of DeBERTa's relative-position subgraph (modeling_deberta_v2.py:595-599, 292-299): per layer key_proj / query_proj of
    LayerNorm(rel_embeddings), rows [2 * buckets][hidden] in f32 (the arithmetic of the constant folder: numpy here)."""
    H, eps = cfg["hidden"], np.float32(cfg["ln_eps"])
    R = 2 * cfg["position_buckets"]
    rel = np.asarray(W["deberta.encoder.rel_embeddings.weight"], np.float32)[:R]
    mean = rel.mean(-1, keepdims=True, dtype=np.float32)
    var = ((rel - mean) ** 2).mean(-1, keepdims=True, dtype=np.float32)
    ln = ((rel - mean) / np.sqrt(var + eps) * W["deberta.encoder.LayerNorm.weight"] + W["deberta.encoder.LayerNorm.bias"]).astype(np.float32)
    out = {}
    for i in range(cfg["layers"]):
        p = f"deberta.encoder.layer.{i}.attention.self."
        for kind, dst in (("key_proj", "pos_key"), ("query_proj", "pos_query")):
            out[p + dst] = (ln @ np.asarray(W[p + kind + ".weight"], np.float32).T + W[p + kind + ".bias"]).astype(np.float32)
    return out


def deberta_onnx(W: dict, cfg: dict, prefix="", raw=True, folded=None) -> bytes:
    """deberta.onnx: every nn.Linear exported as MatMul(x, W^T) + Add(named bias); embeddings, LayerNorms and the ConvLayer keep names.
    folded = None: rel_embeddings / encoder.LayerNorm are in the file (an export without onnxsim).  "direct" / "tiled": the onnxsim form
    (convert_deberta.py:52): those three tensors are gone and every layer has two anonymous constants, the projected positions, feeding its c2p / p2c
    MatMuls: as [heads, d, R] directly ("direct") or as [1, heads, R, d] behind Tile (the batch repeat) + Transpose ("tiled")."""
    anon = _Anon()
    inits, nodes = [], []
    k = 0
    lin_out = {}
    gone = {"deberta.encoder.rel_embeddings.weight", "deberta.encoder.LayerNorm.weight", "deberta.encoder.LayerNorm.bias"} if folded else set()
    for name, a in W.items():
        if name in gone:
            continue
        a = np.asarray(a, np.float32)
        is_linear = name.endswith(".weight") and a.ndim == 2 and (name[:-7] + ".bias") in W and "embeddings" not in name
        if is_linear:
            wt = anon("MatMul")
            inits.append(tensor_proto(wt, np.ascontiguousarray(a.T), raw))
            k += 1
            nodes.append(node_proto("MatMul", [f"/h_{k}", wt], [f"/mm_{k}"]))
            nodes.append(node_proto("Add", [prefix + name[:-7] + ".bias", f"/mm_{k}"], [f"/lin_{k}"]))
            lin_out[name[:-7]] = f"/lin_{k}"
        else:
            inits.append(tensor_proto(prefix + name, a, raw))
    if folded:
        heads, H = cfg["heads"], cfg["hidden"]
        d, R = H // heads, 2 * cfg["position_buckets"]
        inits.append(tensor_proto(anon("Reshape"), np.array([0, -1, heads, d], np.int64)))
        shape_name = f"onnx::Reshape_{anon.n}"
        inits.append(tensor_proto("/reps", np.array([1, 1, 1, 1], np.int64)))
        for name, P in folded_positions(W, cfg).items():
            layer = name.rsplit(".", 1)[0] + "."
            dyn = lin_out[layer + ("query_proj" if name.endswith("pos_key") else "key_proj")]      # c2p: query x position keys; p2c: key x position queries
            tag = name.replace(".", "_")
            nodes.append(node_proto("Reshape", [dyn, shape_name], [f"/r_{tag}"]))
            nodes.append(node_proto("Transpose", [f"/r_{tag}"], [f"/t_{tag}"], [attr_ints("perm", [0, 2, 1, 3])]))
            P3 = P.reshape(R, heads, d).transpose(1, 0, 2)                                             # [heads, R, d]
            cn = anon("MatMul")
            if folded == "direct":
                inits.append(tensor_proto(cn, np.ascontiguousarray(P3.transpose(0, 2, 1)), raw))      # [heads, d, R]
                nodes.append(node_proto("MatMul", [f"/t_{tag}", cn], [f"/att_{tag}"]))
            else:
                inits.append(tensor_proto(cn, np.ascontiguousarray(P3[None]), raw))                   # [1, heads, R, d]
                nodes.append(node_proto("Tile", [cn, "/reps"], [f"/tile_{tag}"]))
                nodes.append(node_proto("Transpose", [f"/tile_{tag}"], [f"/pt_{tag}"], [attr_ints("perm", [0, 1, 3, 2])]))
                nodes.append(node_proto("MatMul", [f"/t_{tag}", f"/pt_{tag}"], [f"/att_{tag}"]))
    nodes.append(node_proto("LayerNormalization", ["/h_0", prefix + "deberta.embeddings.LayerNorm.weight", prefix + "deberta.embeddings.LayerNorm.bias"],
                            ["/ln_0"], [attr_ints("axis", [-1]), attr_float("epsilon", cfg["ln_eps"])]))
    nodes.append(node_proto("Tanh" if cfg.get("conv_act") == "tanh" and cfg.get("conv_kernel_size", 0) > 0 else "Erf", ["/c"], ["/a"]))
    return model_proto(nodes, inits)


def zstd_compress(data: bytes, level=19) -> bytes:
    l = ctypes.CDLL("libzstd.so.1")
    l.ZSTD_compressBound.restype = ctypes.c_size_t
    l.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    l.ZSTD_compress.restype = ctypes.c_size_t
    l.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    cap = l.ZSTD_compressBound(len(data))
    buf = ctypes.create_string_buffer(cap)
    n = l.ZSTD_compress(buf, cap, data, len(data), level)
    assert not l.ZSTD_isError(n)
    return buf.raw[:n]


def style_json(sv: np.ndarray) -> bytes:
    return json.dumps({"data": np.asarray(sv).tolist(), "shape": list(sv.shape)}).encode()      # convert_model.py:38-45


def sbv2_file(onnx_bytes: bytes, style_bytes: bytes, compress=True, entries=("version.txt", "model.onnx", "style_vectors.json")) -> bytes:
    bio = io.BytesIO()
    with tarfile.open(fileobj=bio, mode="w") as w:      # convert_model.py:160-170
        for name in entries:
            b = {"version.txt": b"1", "model.onnx": onnx_bytes, "style_vectors.json": style_bytes}[name]
            t = tarfile.TarInfo(name)
            t.size = len(b)
            w.addfile(t, io.BytesIO(b))
    return zstd_compress(bio.getvalue()) if compress else bio.getvalue()
