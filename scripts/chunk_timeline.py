#!/usr/bin/env python
"""The GPU timeline of ONE streaming chunk through the product transcriber (VAD on the GPU, log-mel, encoder, 16 decode steps): run under
`rocprofv3 --kernel-trace --output-format csv -d DIR`, then `chunk_timeline.py --analyse DIR` lists the idle gaps between consecutive kernels of the
last call (everything above 4 us), i.e. where the host — not a kernel — is what the GPU waits for.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python scripts/chunk_timeline.py ;  python scripts/chunk_timeline.py --analyse DIR"""
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch  # noqa: F401 — FIRST: libwlx.so then binds to the HIP runtime torch bundles; under the image's own ROCm 7.2 runtime rocprofv3 segfaults
    #              inside hipGraphLaunch after ~170 graph launches (scripts/rocprof_graph_probe.py; bench.py imports torch and is not affected)
    import numpy as np
    from bench import make_bench_transcriber, stream_pcm, token_ids
    from whisperlive_amd import vad
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import get_spec
    from whisperlive_amd.synthetic import energy_following_vad_weights
    from whisperlive_amd.weights import random_weights
    spec = get_spec("small.en")
    eng = HipWhisperEngine(spec, random_weights(spec, seed=0))
    vm = vad.SileroHIPModel(energy_following_vad_weights(3), device=eng.device)
    tr = make_bench_transcriber(eng, spec, token_ids(spec.vocab), 16, vad_model=vm)
    pcm = stream_pcm(8.0, 77)
    for i in range(int(os.environ.get("CALLS", "12"))):
        t0 = time.perf_counter()
        segs, info = tr.transcribe(pcm, language="en", vad_filter=True, initial_prompt=None)
        list(segs)
        print("call", i, "wall %.3f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
        time.sleep(float(os.environ.get("PAUSE", "0.02")))   # calls are separated by > 10 ms on the GPU timeline
    del tr
    vm.close(); eng.close()


def analyse(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    calls, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 5_000_000:
            calls.append(cur); cur = []
        cur.append(b)
    calls.append(cur)
    c = calls[-2] if len(calls) > 2 else calls[-1]
    t0, t1 = int(c[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in c)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in c)
    print("last full call: %d kernels, first start -> last end %.1f us, sum of kernel durations %.1f us" % (len(c), (t1 - t0) / 1e3, busy / 1e3))
    short = lambda n: n.split("(")[0].replace("void wlx::", "").replace("wlx::", "")[-44:]
    tot = 0.0
    end = int(c[0]["End_Timestamp"])
    for a, b in zip(c, c[1:]):
        gap = (int(b["Start_Timestamp"]) - end) / 1e3
        if gap > 4.0:
            tot += gap
            print("  gap %7.1f us at +%8.1f us  after %-44s before %s" % (gap, (int(b["Start_Timestamp"]) - t0) / 1e3, short(a["Kernel_Name"]), short(b["Kernel_Name"])))
        end = max(end, int(b["End_Timestamp"]))
    print("gaps above 4 us: %.1f us in all" % tot)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse": analyse(sys.argv[2])
    else: run()
