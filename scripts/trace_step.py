#!/usr/bin/env python
"""In-kernel timeline of one decode step on MI355X (profiling tool, not a test).

Builds libwlx_trace.so (same sources, -DWLX_TRACE) HERE if missing (hipcc cross-compiles), runs one window through it on
the GPU box and prints, per launch of the captured decode-step graph: when its first workgroup started and its last one
ended (100 MHz chip-wide clock, relative to the first launch), the gap to the previous launch, and the mean time wave 0
of a workgroup spent up to each interior mark (shader clock, ns at the measured clock ratio).
usage: WLX_LIB=whisperlive_amd/libwlx_trace.so python scripts/trace_step.py [--model small.en] [--t 33] [--csv out.csv]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WLX_LIB", os.path.join(ROOT, "whisperlive_amd", "libwlx_trace.so"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="small.en")
    ap.add_argument("--t", type=int, default=33)
    ap.add_argument("--csv", default=None)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--items", type=int, default=1, help="audio items in the slot (rows = items x 5 beams: 8 -> the batch worker's 40 rows)")
    ap.add_argument("--waves", action="store_true", help="per-wave detail for the lean GEMV launches of the first layer")
    args = ap.parse_args()
    from whisperlive_amd import _lib
    _lib.build_trace()
    if args.build_only:
        return
    import bench
    from oracle import logmel as olm
    from whisperlive_amd.engine import HipWhisperEngine, TokenIds
    from whisperlive_amd.specs import get_spec
    from whisperlive_amd.weights import random_weights

    spec = get_spec(args.model)
    eng = HipWhisperEngine(spec, random_weights(spec, seed=0), device=0)
    slot = eng.create_slot(args.items, 5)
    ids = bench.token_ids(spec.vocab)
    B = args.items
    Ts = []
    for i in range(B):
        slot.pcm_put(olm.speech_like_pcm(30.0, seed=1234 + i), item=i)
        Ts.append(slot.logmel_resident(item=i))
    slot.encode(B, seek=[0] * B, seg=[min(t - 1, 3000) for t in Ts])
    slot.generate([[ids["sot"]]] * B, TokenIds(**ids), beam_size=5, patience=1.0, max_length=65 if B == 1 else 9,
                  suppress_tokens=bench.suppress_list(ids, True))
    names, rec = slot.debug_trace_step(5 * B, args.t, True)
    n = len(names)
    t0 = None
    prev_end = None
    rows = []
    print(f"{'#':>3} {'launch':<28} {'wgs':>5} {'start':>8} {'gap':>6} {'span':>6} {'wg_mean':>7} | marks (ns from wg entry, mean over wgs): m1 m2 m3 m4 m5")
    for i in range(n):
        wg = rec[i, 1:].astype(np.int64)
        wg = wg[wg[:, 1] > 0]
        cnt = wg.shape[0]
        if cnt == 0:
            print(f"{i:3d} {names[i]:<28} (no records)")
            continue
        start = int(wg[:, 0].min()); end = int(wg[:, 1].max())
        if t0 is None:
            t0 = start
        dur_rt = (wg[:, 1] - wg[:, 0]) * 10.0          # ns (100 MHz)
        m0 = wg[:, 2]
        # shader-clock ticks -> ns using this launch's own ratio (last mark that is set vs realtime duration)
        marks = []
        last = None
        for j in range(1, 6):
            col = wg[:, 2 + j]
            ok = col > 0
            marks.append(float(((col - m0)[ok]).mean()) if ok.any() else float("nan"))
            if ok.any():
                last = ((col - m0)[ok]).mean()
        ratio = (dur_rt.mean() / last) if last else float("nan")   # ns per tick (approximate: END is after the last mark)
        gap = (start - prev_end) * 10.0 if prev_end is not None else 0.0
        span = (end - start) * 10.0
        rows.append((i, names[i], cnt, (start - t0) * 10.0, gap, span, dur_rt.mean(), marks, ratio))
        print(f"{i:3d} {names[i]:<28} {cnt:5d} {(start - t0) * 10.0:8.0f} {gap:6.0f} {span:6.0f} {dur_rt.mean():7.0f} | "
              + " ".join(f"{m:7.0f}" for m in marks) + f"  (ticks; {ratio:.3f} ns/tick)")
        prev_end = end
        if args.waves and names[i].startswith("gemv2") and i <= 9:
            # per-wave records (index = workgroup * 16 + wave): when did each wave's loads land (m1), when did it reach
            # the K-reduction barrier (m3), relative to its own entry; and its entry relative to the launch's first entry
            raw = rec[i, 1:].astype(np.int64)
            idx = np.nonzero(raw[:, 1] > 0)[0]
            wv = idx % 16
            for w_ in sorted(set(wv.tolist())):
                sel = raw[idx[wv == w_]]
                ent = (sel[:, 0] - start).mean() * 10.0
                m1 = (sel[:, 3] - sel[:, 2]); m3 = (sel[:, 5] - sel[:, 2])
                print(f"      wave {w_:2d}: n={sel.shape[0]:4d} entry +{ent:5.0f} ns | m1 {m1[m1 > 0].mean() * ratio if (m1 > 0).any() else float('nan'):6.0f} ns  "
                      f"m3 {m3[m3 > 0].mean() * ratio if (m3 > 0).any() else float('nan'):6.0f} ns  end {(sel[:, 1] - sel[:, 0]).mean() * 10.0:6.0f} ns")
    total = (prev_end - t0) * 10.0
    print(f"step total {total / 1000.0:.1f} us over {n} launches; sum of spans {sum(r[5] for r in rows) / 1000.0:.1f} us, "
          f"sum of gaps {sum(r[4] for r in rows) / 1000.0:.1f} us")
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("idx,launch,wgs,start_ns,gap_ns,span_ns,wg_mean_ns,m1_ticks,m2_ticks,m3_ticks,m4_ticks,m5_ticks,ns_per_tick\n")
            for r in rows:
                f.write(f"{r[0]},\"{r[1]}\",{r[2]},{r[3]:.0f},{r[4]:.0f},{r[5]:.0f},{r[6]:.0f}," + ",".join(f"{m:.0f}" for m in r[7]) + f",{r[8]:.4f}\n")
    slot.close(); eng.close()


if __name__ == "__main__":
    main()
