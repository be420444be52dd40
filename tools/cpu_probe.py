"""Builder tool (GPU box): how many host CPUs may this process really use, and how does oracle/sbv2_ref.c scale with the thread count?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e)
print("loadavg", open("/proc/loadavg").read().strip())
import sbv2_ref as R
from sbv2_api_amd import synth, configs
lib = R.load(native=True)
vc = configs.VITS_FULL
vw = synth.make_vits_weights(vc)
m = R.Model(None, synth.pack_blob(synth.KIND_VITS, vc, vw), lib=lib)
u = synth.make_utterance(32, configs.DEBERTA_FULL, vc, seed=1)
bert = synth.hash_normal(5, vc["bert_dim"] * u["T_text"]).reshape(vc["bert_dim"], -1)
for th in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    lib.sbv2c_set_threads(th)
    t = time.time()
    pcm = m.vits(bert, u["phones"], u["tones"], u["langs"], 0, u["style"], forced_durations=u["forced_durations"])
    dt = time.time() - t
    print(f"threads {th}: {dt:.2f} s for {len(pcm)/44100:.2f} s audio -> {len(pcm)/44100/dt:.2f} audio-s/s", flush=True)
