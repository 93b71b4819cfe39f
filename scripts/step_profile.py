"""One decode step at `rows` decoder rows: captured-graph time and the per-kernel table (wlx_debug_profile_step).
usage: [WLX_ROWTILE=0 ...] python scripts/step_profile.py MODEL ROWS [T] [--json out.json]
Env A/B switches are read by libwlx at first use, so one process = one configuration."""
import json
import os
import sys

sys.path.insert(0, ".")
from oracle import logmel as olm          # synthetic PCM generator only
from whisperlive_amd.engine import HipWhisperEngine
from whisperlive_amd.specs import get_spec
from whisperlive_amd.weights import random_weights

args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "small.en"
rows = int(args[1]) if len(args) > 1 else 60
t = int(args[2]) if len(args) > 2 else 33
out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
beams = 5 if rows % 5 == 0 else 1
batch = rows // beams
spec = get_spec(name)
eng = HipWhisperEngine(spec, random_weights(spec, seed=0))
sl = eng.create_slot(batch, beams)
pcm = olm.speech_like_pcm(30.0, seed=1234)
for b in range(batch):
    T = sl.logmel(pcm, item=b) if b else sl.logmel(pcm)
sl.encode(batch, seek=[0] * batch, seg=[min(T - 1, 3000)] * batch)
ms = sl.debug_time_decode_step(rows, t, 30)
prof = sl.debug_profile_step(rows, t, 10)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("WLX_") and k != "WLX_LIB")
print(f"== {name} rows={rows} t={t} [{tag}] step graph {1e3 * ms:.1f} us; sum of kernel chains {sum(k['total_us'] for k in prof):.1f} us")
for k in sorted(prof, key=lambda k: -k["total_us"]):
    gbs = k["bytes_per_launch"] / (k["avg_us"] * 1e-6) / 1e9 if k["avg_us"] > 0 else 0
    print(f"   {k['name']:<48s} x{k['launches']:<4.0f} {k['avg_us']:7.2f} us  {k['total_us']:8.1f} us  {k['bytes_per_launch'] / 1e6:8.2f} MB  {gbs:7.0f} GB/s")
if out:
    json.dump({"model": name, "rows": rows, "t": t, "env": tag, "graph_us": 1e3 * ms, "kernels": prof}, open(out, "w"), indent=1)
sys.stdout.flush()
sl.close()
eng.close()
os._exit(0)
