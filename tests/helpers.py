"""Shared builders for the GPU parity tests (weights are regenerated from (config, seed), never stored)."""
import functools

import numpy as np

import sbv2_oracle as O
from sbv2_api_amd import synth


@functools.lru_cache(maxsize=None)
def weights(kind: str, size: str, seed: int = 0x5B72):
    cfg = {("bert", "tiny"): O.DEBERTA_TINY, ("bert", "full"): O.DEBERTA_FULL,
           ("vits", "tiny"): O.VITS_TINY, ("vits", "full"): O.VITS_FULL}[(kind, size)]
    W = (synth.make_deberta_weights if kind == "bert" else synth.make_vits_weights)(cfg, seed)
    return cfg, W


def blob(kind: str, size: str, seed: int = 0x5B72) -> bytes:
    cfg, W = weights(kind, size, seed)
    return synth.pack_blob(synth.KIND_BERT if kind == "bert" else synth.KIND_VITS, cfg, W)


def noise_key(seed: int, utt: int, stream: int) -> int:
    """Same key schedule as csrc/ops.hip::noise_key."""
    return (seed + (2 * utt + stream) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


def oracle_noise_w(seed, utt, T, scale):
    return synth.hash_normal(noise_key(seed, utt, 0), 2 * T).reshape(2, T) * np.float32(scale)


def oracle_noise_z(seed, utt, C, scale):
    return lambda Tf: synth.hash_normal(noise_key(seed, utt, 1), C * Tf).reshape(C, Tf) * np.float32(scale)


def make_utts(ns, bert_cfg, vits_cfg, seed0=0, with_bert=True):
    utts = []
    for i, n in enumerate(ns):
        u = synth.make_utterance(n, bert_cfg, vits_cfg, seed=seed0 + i)
        if with_bert:
            u["bert"] = synth.hash_normal(1000 + seed0 + i, vits_cfg["bert_dim"] * u["T_text"]).reshape(vits_cfg["bert_dim"], -1)
        utts.append(u)
    return utts
