"""TEST INFRASTRUCTURE ONLY (like sbv2_oracle.py): numpy restatement of the reference's per-request orchestration, one sentence at a
time exactly as the reference loops (crates/sbv2_core/src/tts.rs:280-349, style.rs:19-28, tts_util.rs:163-180), with the synthesis
itself delegated to a callable so that tests can plug the oracle model (or batch-1 calls of the device path) in.

PARITY UNPINNED for the WAV header bytes: hound 3.5.1 is a cargo dependency that is not vendored under /root/reference; the
header is restated from the RIFF/WAVE_FORMAT_EXTENSIBLE specification the crate follows and checked with scipy's reader."""
import struct

import numpy as np


def get_style_vector(style_vectors, style_id, weight):
    mean = style_vectors[0].astype(np.float32)
    diff = (style_vectors[style_id].astype(np.float32) - mean) * np.float32(weight)
    return mean + diff


def easy_synthesize(sentences, synth_one, split_sentences=True):
    """sentences: parsed lines (None for an empty line); synth_one(sentence) -> 1-D f32 PCM.  Returns [1, 1, L]."""
    audios = []
    for i, s in enumerate(sentences):
        if not s:
            continue
        audios.append(np.asarray(synth_one(s), np.float32).reshape(1, 1, -1))
        if split_sentences and i != len(sentences) - 1:
            audios.append(np.zeros((1, 1, 22050), np.float32))
    return np.concatenate(audios, axis=2)


def array_to_wav(audio):
    out = bytearray()
    samples = []
    for i in range(audio.shape[0]):
        samples.extend(float(v) for v in audio[i, 0, :])
    payload = b"".join(struct.pack("<f", v) for v in samples)
    out += b"RIFF" + struct.pack("<I", 4 + 8 + 40 + 8 + len(payload)) + b"WAVE"
    out += b"fmt " + struct.pack("<I", 40)
    out += struct.pack("<H", 0xFFFE) + struct.pack("<H", 1) + struct.pack("<I", 44100) + struct.pack("<I", 44100 * 4)
    out += struct.pack("<H", 4) + struct.pack("<H", 32) + struct.pack("<H", 22) + struct.pack("<H", 32) + struct.pack("<I", 1)
    out += bytes.fromhex("0300000000001000800000aa00389b71")
    out += b"data" + struct.pack("<I", len(payload)) + payload
    return bytes(out)
