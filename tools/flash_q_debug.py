"""Builder tool: the flow attention on pre-split keys / values (k_vits_flash_x3q) against the converting kernel, waveform by waveform, over frame counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sbv2_oracle as O
from helpers import blob, make_utts, weights
from sbv2_api_amd import _lib, model

lib = _lib.lib()
size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
frames = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5, 13, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 200, 257]
cfg, W = weights("vits", size)
s = model.load_model(blob("vits", size), False)
for f in frames:
    u = make_utts([4], O.DEBERTA_FULL if size == "full" else O.DEBERTA_TINY, cfg, seed0=177)[0]
    d = np.ones_like(u["forced_durations"]); d[0] = max(1, f - (d.size - 1)); u["forced_durations"] = d
    prev = lib.sbv2_debug_set_flash_parts(2)
    a = model.synthesize_batch(s, [u], forced=True)[0]
    lib.sbv2_debug_set_flash_parts(0)
    b = model.synthesize_batch(s, [u], forced=True)[0]
    lib.sbv2_debug_set_flash_parts(prev)
    print(f, int(d.sum()), "max abs diff", float(np.abs(a - b).max()), "peak", float(np.abs(b).max()), flush=True)
s.close()
