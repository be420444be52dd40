import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sbv2_api_amd import _lib, configs, model, synth
# Builder tool (GPU box): per-launch time of the small-grid kernels inside a single-utterance call, from the library's own HIP-event
# profile (sbv2_prof_begin / end); used with SBV2_CLS_ABL / SBV2_SKINNY_MAX to see what the time is made of.
import ctypes as C, json
bc, vc = configs.DEBERTA_FULL, configs.VITS_FULL
bs = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, synth.make_deberta_weights(bc)), True)
vs = model.load_model(synth.pack_blob(synth.KIND_VITS, vc, synth.make_vits_weights(vc)), False)
u = synth.make_utterance(128, bc, vc, seed=1)
pipe = model.Pipeline(bs, vs)
b = pipe.prepare([u], forced=True)
for _ in range(3):
    pipe.run(b); pipe.sync()
l = _lib.lib()
_lib.check(l.sbv2_prof_begin())
pipe.run(b); pipe.sync()
buf = C.create_string_buffer(1 << 16)
_lib.check(l.sbv2_prof_end(buf, len(buf)))
for r in json.loads(buf.value.decode()):
    print(f"{r['kernel']:28s} {r['launches']:4d} launches  {r['ms']:7.3f} ms  {r['ms'] / r['launches'] * 1e3:8.1f} us each")
