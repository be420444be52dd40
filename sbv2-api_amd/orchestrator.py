"""Host-side mirror of the reference's per-request orchestration around the hot path.

Reference behaviour restated here (crates/sbv2_core/src):
  style.rs:11-17    load_style          {"shape": [n, 256], "data": [[...], ...]} JSON -> [n, 256] f32
  style.rs:19-28    get_style_vector    mean + (style_vectors[style_id] - mean) * weight, mean = row 0
  tts.rs:280-349    easy_synthesize     split on '\\n', skip empty lines, synthesize each sentence with noise_scale 0.677 and
                                        noise_scale_w 0.8, append 22050 zero samples after every sentence that is not the LAST
                                        LINE of the request (empty trailing lines included in that test), concatenate
  tts_util.rs:163-180 array_to_vec      44.1 kHz mono 32-bit IEEE-float WAV through hound 3.5.1 (Cargo.lock:848)

What is different, on purpose: the sentence loop of the reference calls bert::predict and model::synthesize once per sentence;
here all sentences of a request go through ONE batched pipeline call (DeBERTa -> word2ph repeat -> VITS2, device resident), which
is where the MI355X path gets its throughput.  Every sentence's PCM equals its batch-1 result bit for bit (packed-batch design,
tests/test_gpu_parity.py), so the WAV is the same as the sequential loop's given the same noise seeds.

The text front-end (G2P, tokenizer: tts_util.rs:14-155) is out of scope: callers pass per-sentence
dicts {input_ids, word2ph, phones, tones, langs} exactly as parse_text produces them, or None for an empty line.
"""
import json
import struct

import numpy as np

from . import model

SAMPLE_RATE = 44100
SENTENCE_GAP = 22050        # tts.rs:321 Array3::zeros((1, 1, 22050))
NOISE_SCALE = 0.677         # tts.rs:313 / :343
NOISE_SCALE_W = 0.8         # tts.rs:314 / :344


class SynthesizeOptions:
    """tts.rs:359-375 (same defaults)."""

    def __init__(self, sdp_ratio=0.0, length_scale=1.0, style_weight=1.0, split_sentences=True):
        self.sdp_ratio, self.length_scale, self.style_weight, self.split_sentences = sdp_ratio, length_scale, style_weight, split_sentences


def load_style(data: bytes) -> np.ndarray:
    d = json.loads(bytes(data).decode("utf-8"))
    shape = tuple(int(v) for v in d["shape"])
    flat = np.asarray([v for row in d["data"] for v in row], np.float32)
    if len(shape) != 2 or flat.size != shape[0] * shape[1]:
        raise model.Sbv2Error(f"style vectors: {flat.size} values do not fill shape {list(shape)}")   # ndarray ShapeError in the reference
    return flat.reshape(shape)


def get_style_vector(style_vectors: np.ndarray, style_id: int, weight: float) -> np.ndarray:
    sv = np.asarray(style_vectors, np.float32)
    if not 0 <= int(style_id) < sv.shape[0]:
        raise IndexError(f"style_id {style_id} out of range (the reference panics on the slice)")
    mean = sv[0]
    return (mean + (sv[int(style_id)] - mean) * np.float32(weight)).astype(np.float32)


def array_to_wav(audio: np.ndarray) -> bytes:
    """[B, 1, L] f32 -> WAV bytes.  hound writes WAVE_FORMAT_EXTENSIBLE for anything but <= 16-bit integer PCM: 40-byte fmt chunk,
    sub-format KSDATAFORMAT_SUBTYPE_IEEE_FLOAT, channel mask = the lowest `channels` bits, no fact chunk.  (Layout restated from
    the crate's documented behaviour; the crate itself is not available here, so the header bytes are unpinned.  The payload,
    sizes and rate are checked by reading the file back with an independent WAV reader in tests/.)"""
    a = np.ascontiguousarray(np.asarray(audio, np.float32))
    if a.ndim != 3:
        raise ValueError("audio must be [B, 1, L]")
    samples = a[:, 0, :].reshape(-1).astype("<f4")
    data = samples.tobytes()
    channels, bits = 1, 32
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHHHHI", 0xFFFE, channels, SAMPLE_RATE, SAMPLE_RATE * block, block, bits, 22, bits, (1 << channels) - 1)
    fmt += bytes([0x03, 0x00, 0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71])
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


def easy_synthesize(pipe: "model.Pipeline", sentences, style_vectors, style_id=0, speaker_id=0, options=None, noise_seed=None,
                    noise_scale=NOISE_SCALE, noise_scale_w=NOISE_SCALE_W) -> bytes:
    """tts.rs:280-349 for one request whose lines are already parsed: `sentences` is the list obtained from text.split('\\n'),
    each entry a dict {input_ids, word2ph, phones, tones, langs} (parse_text's products) or None / {} for an empty line.
    With options.split_sentences False the caller passes the single parsed text as a one-element list."""
    options = options or SynthesizeOptions()
    if noise_seed is None:      # the reference draws fresh noise per request; tests pass an explicit seed
        noise_seed = model.fresh_noise_seed()
    style = get_style_vector(style_vectors, style_id, options.style_weight)
    live = [(i, s) for i, s in enumerate(sentences) if s]
    if not live:
        raise model.Sbv2Error("nothing to synthesize (the reference's concatenate fails on an empty list)")
    utts = [dict(s, style=style, sid=speaker_id) for _, s in live]
    b = pipe.prepare(utts, sdp_ratio=options.sdp_ratio, length_scale=options.length_scale, noise_scale=noise_scale,
                     noise_scale_w=noise_scale_w, noise_seed=noise_seed)
    pipe.run(b)
    pcm = pipe.fetch(b)
    parts = []
    for (i, _), wav in zip(live, pcm):
        parts.append(wav)
        if options.split_sentences and i != len(sentences) - 1:
            parts.append(np.zeros(SENTENCE_GAP, np.float32))
    return array_to_wav(np.concatenate(parts).reshape(1, 1, -1))
