"""Synthetic weights / inputs and the weight-blob container (numpy only).

There is no network and no model file in the build or bench environment, so tests and
bench.py feed the library procedurally generated weights of the exact JP-Extra / DeBERTa-v2-large
shapes.  Values come from a counter-based hash (splitmix64 of tensor-name hash + element index), so a
tensor is reproducible from (seed, name, shape) alone.

Container ("SBV2W001"): what `sbv2_bert_create` / `sbv2_vits_create` accept in place of the ONNX bytes
the reference hands to `load_model` (crates/sbv2_core/src/model.rs:6).  Layout, little endian:

    magic[8] "SBV2W001" | u32 kind (1 bert, 2 vits) | u32 n_tensors | u64 json_len | json (config)
    n_tensors x { u16 name_len | name | u32 ndim | u64 dims[ndim] | u64 byte_offset }   (f32 data)
    data (each tensor 64-byte aligned, offsets from file start)

Tensor names are the upstream PyTorch state-dict names, so a converter from a real checkpoint / ONNX
initializer table only has to emit the same names.
"""
from __future__ import annotations

import json
import math
import struct

import numpy as np

MAGIC = b"SBV2W001"
KIND_BERT, KIND_VITS = 1, 2
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for c in s.encode():
        h = ((h ^ c) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_uniform(key: int, n: int, start: int = 0) -> np.ndarray:
    """n floats in [0, 1) with 24-bit resolution: splitmix64(key + index) >> 40."""
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22
    with np.errstate(over="ignore"):
        for s in range(0, n, step):
            e = min(n, s + step)
            z = np.arange(start + s, start + e, dtype=np.uint64) + np.uint64(key)
            z = z * np.uint64(0x9E3779B97F4A7C15)
            z ^= z >> np.uint64(30)
            z *= np.uint64(0xBF58476D1CE4E5B9)
            z ^= z >> np.uint64(27)
            z *= np.uint64(0x94D049BB133111EB)
            z ^= z >> np.uint64(31)
            out[s:e] = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return out


def hash_normal(key: int, n: int) -> np.ndarray:
    """Box-Muller on two hash streams; used for injected noise and style vectors (float32)."""
    u1 = hash_uniform(key, n).astype(np.float64)
    u2 = hash_uniform(key ^ 0x5851F42D4C957F2D, n).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * math.pi * u2)).astype(np.float32)


def _tensor(seed, name, shape, amp, center=0.0):
    n = int(np.prod(shape))
    u = hash_uniform(_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF), n)
    return ((u - np.float32(0.5)) * np.float32(2.0 * amp) + np.float32(center)).reshape(shape)


class _Builder:
    def __init__(self, seed):
        self.seed = seed
        self.w = {}

    def fan(self, name, shape, fan_in, gain=1.0):
        """uniform with variance gain^2 / fan_in."""
        self.w[name] = _tensor(self.seed, name, shape, gain * math.sqrt(3.0 / fan_in))

    def bias(self, name, n, amp=0.05):
        self.w[name] = _tensor(self.seed, name, (n,), amp)

    def ln(self, wname, bname, n):
        self.w[wname] = _tensor(self.seed, wname, (n,), 0.1, center=1.0)
        self.w[bname] = _tensor(self.seed, bname, (n,), 0.05)

    def conv(self, prefix, cout, cin, k, gain=1.0, bias=True):
        self.fan(prefix + ".weight", (cout, cin, k), cin * k, gain)
        if bias:
            self.bias(prefix + ".bias", cout)

    def lin(self, prefix, cout, cin, gain=1.0):
        self.fan(prefix + ".weight", (cout, cin), cin, gain)
        self.bias(prefix + ".bias", cout)


def make_deberta_weights(cfg, seed=0x5B72):
    """All tensors `deberta_forward` reads, state-dict names of transformers DebertaV2Model under `deberta.`."""
    b = _Builder(seed)
    H, I = cfg["hidden"], cfg["intermediate"]
    span = cfg["position_buckets"] if cfg["position_buckets"] > 0 else cfg["max_relative_positions"]
    b.w["deberta.embeddings.word_embeddings.weight"] = _tensor(seed, "deberta.embeddings.word_embeddings.weight",
                                                               (cfg["vocab_size"], H), 1.0)
    b.ln("deberta.embeddings.LayerNorm.weight", "deberta.embeddings.LayerNorm.bias", H)
    b.w["deberta.encoder.rel_embeddings.weight"] = _tensor(seed, "deberta.encoder.rel_embeddings.weight",
                                                           (2 * span, H), 1.0)
    b.ln("deberta.encoder.LayerNorm.weight", "deberta.encoder.LayerNorm.bias", H)
    if cfg.get("conv_kernel_size", 0) > 0:     # DebertaV2Encoder.conv (ConvLayer after layer 0)
        b.conv("deberta.encoder.conv.conv", H, H, cfg["conv_kernel_size"])
        b.ln("deberta.encoder.conv.LayerNorm.weight", "deberta.encoder.conv.LayerNorm.bias", H)
    for i in range(cfg["layers"]):
        p = f"deberta.encoder.layer.{i}."
        for n in ("query_proj", "key_proj", "value_proj"):
            b.lin(p + "attention.self." + n, H, H, gain=1.5)
        b.lin(p + "attention.output.dense", H, H)
        b.ln(p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias", H)
        b.lin(p + "intermediate.dense", I, H)
        b.lin(p + "output.dense", H, I)
        b.ln(p + "output.LayerNorm.weight", p + "output.LayerNorm.bias", H)
    return b.w


def _encoder_weights(b, p, cfg, n_layers, kernel):
    H, Fc, heads, w = cfg["hidden"], cfg["filter"], cfg["heads"], cfg["window"]
    dk = H // heads
    b.lin(p + "spk_emb_linear", H, cfg["gin"])
    for i in range(n_layers):
        a = f"{p}attn_layers.{i}."
        for n in ("conv_q", "conv_k", "conv_v"):
            b.conv(a + n, H, H, 1, gain=1.5)
        b.conv(a + "conv_o", H, H, 1)
        b.w[a + "emb_rel_k"] = _tensor(b.seed, a + "emb_rel_k", (1, 2 * w + 1, dk), math.sqrt(3.0 / dk) * 2)
        b.w[a + "emb_rel_v"] = _tensor(b.seed, a + "emb_rel_v", (1, 2 * w + 1, dk), math.sqrt(3.0 / dk) * 2)
        b.ln(f"{p}norm_layers_1.{i}.gamma", f"{p}norm_layers_1.{i}.beta", H)
        b.conv(f"{p}ffn_layers.{i}.conv_1", Fc, H, kernel)
        b.conv(f"{p}ffn_layers.{i}.conv_2", H, Fc, kernel)
        b.ln(f"{p}norm_layers_2.{i}.gamma", f"{p}norm_layers_2.{i}.beta", H)


def _dds_weights(b, p, C, cfg):
    k = cfg["sdp_kernel"]
    for i in range(cfg["sdp_dds_layers"]):
        b.fan(f"{p}convs_sep.{i}.weight", (C, 1, k), k)
        b.bias(f"{p}convs_sep.{i}.bias", C)
        b.conv(f"{p}convs_1x1.{i}", C, C, 1)
        b.ln(f"{p}norms_1.{i}.gamma", f"{p}norms_1.{i}.beta", C)
        b.ln(f"{p}norms_2.{i}.gamma", f"{p}norms_2.{i}.beta", C)


def make_vits_weights(cfg, seed=0x5B72):
    """All tensors `vits_forward` reads, state-dict names of style_bert_vits2 `SynthesizerTrn` (JP-Extra),
    weight-norm already folded (what torch.onnx.export + onnxsim leave in the graph)."""
    b = _Builder(seed)
    H, I, G = cfg["hidden"], cfg["inter"], cfg["gin"]
    b.w["emb_g.weight"] = _tensor(seed, "emb_g.weight", (cfg["n_speakers"], G), 1.0)
    # text encoder
    b.w["enc_p.emb.weight"] = _tensor(seed, "enc_p.emb.weight", (cfg["n_vocab"], H), H ** -0.5 * 1.7)
    b.w["enc_p.tone_emb.weight"] = _tensor(seed, "enc_p.tone_emb.weight", (cfg["n_tones"], H), H ** -0.5 * 1.7)
    b.w["enc_p.language_emb.weight"] = _tensor(seed, "enc_p.language_emb.weight", (cfg["n_langs"], H), H ** -0.5 * 1.7)
    b.conv("enc_p.bert_proj", H, cfg["bert_dim"], 1, gain=H ** -0.5)
    b.lin("enc_p.style_proj", H, cfg["style_dim"], gain=H ** -0.5)
    _encoder_weights(b, "enc_p.encoder.", cfg, cfg["enc_layers"], cfg["enc_kernel"])
    b.conv("enc_p.proj", 2 * I, H, 1, gain=0.5)
    # duration predictors
    Fd = cfg["dp_filter"]
    b.conv("dp.cond", H, G, 1, gain=0.3)
    b.conv("dp.conv_1", Fd, H, cfg["dp_kernel"])
    b.ln("dp.norm_1.gamma", "dp.norm_1.beta", Fd)
    b.conv("dp.conv_2", Fd, Fd, cfg["dp_kernel"])
    b.ln("dp.norm_2.gamma", "dp.norm_2.beta", Fd)
    b.conv("dp.proj", 1, Fd, 1, gain=0.8)
    b.w["dp.proj.bias"] = np.full((1,), 0.9, np.float32)   # exp(0.9) ~ 2.5 frames per symbol on average
    b.conv("sdp.pre", H, H, 1)
    b.conv("sdp.proj", H, H, 1)
    b.conv("sdp.cond", H, G, 1, gain=0.3)
    _dds_weights(b, "sdp.convs.", H, cfg)
    b.w["sdp.flows.0.m"] = _tensor(seed, "sdp.flows.0.m", (2, 1), 0.3, center=-0.9)
    b.w["sdp.flows.0.logs"] = _tensor(seed, "sdp.flows.0.logs", (2, 1), 0.3)
    for i in range(2, cfg["sdp_flows"] + 1):           # ConvFlow 1 is dropped in reverse mode
        p = f"sdp.flows.{2 * i - 1}."
        b.conv(p + "pre", H, 1, 1)
        _dds_weights(b, p + "convs.", H, cfg)
        b.conv(p + "proj", 3 * cfg["sdp_bins"] - 1, H, 1, gain=3.0)
    # flow
    for i in range(cfg["flow_n"]):
        p = f"flow.flows.{2 * i}."
        b.conv(p + "pre", H, I // 2, 1)
        _encoder_weights(b, p + "enc.", cfg, cfg["flow_layers"], cfg["flow_kernel"])
        b.conv(p + "post", I // 2, H, 1, gain=0.5)
    # HiFi-GAN
    C = cfg["up_initial"]
    b.conv("dec.conv_pre", C, I, 7)
    b.conv("dec.cond", C, G, 1, gain=0.3)
    nk = len(cfg["res_kernels"])
    for i, (r, k) in enumerate(zip(cfg["up_rates"], cfg["up_kernels"])):
        # ConvTranspose1d weight is [Cin, Cout, k]; each output sample sees Cin * k / r taps
        b.fan(f"dec.ups.{i}.weight", (C, C // 2, k), C * k / r)
        b.bias(f"dec.ups.{i}.bias", C // 2)
        C //= 2
        for j, (rk, dils) in enumerate(zip(cfg["res_kernels"], cfg["res_dilations"])):
            for n in range(len(dils)):
                b.conv(f"dec.resblocks.{i * nk + j}.convs1.{n}", C, C, rk, gain=1.0)
                b.conv(f"dec.resblocks.{i * nk + j}.convs2.{n}", C, C, rk, gain=0.6)
    b.conv("dec.conv_post", 1, C, 7, gain=0.25, bias=False)
    return b.w


# --------------------------------------------------------------------------------------------------
# container
# --------------------------------------------------------------------------------------------------

def pack_blob(kind: int, cfg: dict, weights: dict) -> bytes:
    js = json.dumps(cfg).encode()
    names = list(weights.keys())
    head = bytearray(MAGIC + struct.pack("<IIQ", kind, len(names), len(js)) + js)
    table_len = sum(2 + len(n.encode()) + 4 + 8 * weights[n].ndim + 8 for n in names)
    off = (len(head) + table_len + 63) // 64 * 64
    offs = []
    for n in names:
        offs.append(off)
        off += (weights[n].size * 4 + 63) // 64 * 64
    for n, o in zip(names, offs):
        a = weights[n]
        nb = n.encode()
        head += struct.pack("<H", len(nb)) + nb + struct.pack("<I", a.ndim)
        head += struct.pack(f"<{a.ndim}Q", *a.shape) + struct.pack("<Q", o)
    buf = bytearray(off)
    buf[: len(head)] = head
    for n, o in zip(names, offs):
        a = np.ascontiguousarray(weights[n], dtype=np.float32)
        buf[o:o + a.nbytes] = a.tobytes()
    return bytes(buf)


def unpack_blob(blob: bytes):
    assert blob[:8] == MAGIC
    kind, n, jl = struct.unpack_from("<IIQ", blob, 8)
    pos = 24
    cfg = json.loads(blob[pos:pos + jl])
    pos += jl
    w = {}
    for _ in range(n):
        (nl,) = struct.unpack_from("<H", blob, pos); pos += 2
        name = blob[pos:pos + nl].decode(); pos += nl
        (nd,) = struct.unpack_from("<I", blob, pos); pos += 4
        dims = struct.unpack_from(f"<{nd}Q", blob, pos); pos += 8 * nd
        (off,) = struct.unpack_from("<Q", blob, pos); pos += 8
        cnt = int(np.prod(dims)) if nd else 1
        w[name] = np.frombuffer(blob, dtype=np.float32, count=cnt, offset=off).reshape(dims)
    return kind, cfg, w


# --------------------------------------------------------------------------------------------------
# synthetic utterances (SURVEY.md §8d "Synthetic inputs")
# --------------------------------------------------------------------------------------------------

def make_utterance(n_phones: int, bert_cfg: dict, vits_cfg: dict, seed: int = 0, chars: int | None = None):
    """One utterance with `n_phones` phone symbols (incl. the two `_` pads).

    Mirrors the front end's shapes: T_text = 2N+1 after intersperse with 0 (tts_util.rs:106-108, utils.rs:1-12),
    BERT tokens S = chars + 2 with ids [1] + chars + [2] (tokenizer.rs:10-19), word2ph sums to T_text
    (tts_util.rs:109-112), tones in {6,7} and language 1 at phone positions, 0 at blanks (nlp.rs:17-23).
    """
    N = n_phones
    T = 2 * N + 1
    S = (chars if chars is not None else max(1, N // 2 - 2)) + 2
    key = (seed * 0x9E3779B97F4A7C15 + 0x1234567) & 0xFFFFFFFFFFFFFFFF
    u = hash_uniform(key, 4 * T + S + 8)
    ids = np.empty(S, np.int64)
    ids[0], ids[-1] = 1, 2
    ids[1:-1] = 3 + np.floor(u[: S - 2] * (bert_cfg["vocab_size"] - 3)).astype(np.int64)
    phones = np.zeros(T, np.int64)
    tones = np.zeros(T, np.int64)
    langs = np.zeros(T, np.int64)
    phones[1::2] = 1 + np.floor(u[S:S + N] * (vits_cfg["n_vocab"] - 1)).astype(np.int64)
    tones[1::2] = 6 + (u[S + N:S + 2 * N] > 0.5)
    langs[1::2] = 1
    # word2ph: spread T over S tokens, every token >= 1 while T >= S
    base = np.full(S, T // S, np.int64)
    base[: T % S] += 1
    word2ph = base
    style = hash_normal(key ^ 0xABCDEF, vits_cfg["style_dim"]) * np.float32(0.1)
    # benchmark durations: blank 1 frame, phone 6 frames => T_frames = 7N + 1
    forced = np.where(np.arange(T) % 2 == 1, 6, 1).astype(np.int64)
    return dict(input_ids=ids, attention_mask=np.ones(S, np.int64), word2ph=word2ph, phones=phones, tones=tones,
                langs=langs, style=style, sid=0, forced_durations=forced, T_text=T, S=S)
