#!/bin/bash
# Round-2 checkpoint: every GPU test, smoke(), default bench line (stream leg, CPU baseline, live PMC pass), rocprofv3
# kernel stats of the same bench command, encoder-only stats, large-v3 line, config 5, 4-client stream legs.
set -u
TAG=${1:-r2p}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "p50_chunk_latency_ms", "stage_ms", "parity_prefix")})
print("step:", {k: v for k, v in d["decode_step"].items() if k != "kernels"})
for k in d["decode_step"]["kernels"]: print("   ", k["name"], k["launches"], round(k["avg_us"], 2))
print("roofline:", {k: v for k, v in d["roofline"].items() if k != "largest_launch"})
print("cpu:", d.get("cpu_baseline")); print("stream:", d.get("stream"))
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-pmc > "$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof_enc" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 4 > "$OUT/rocprof_enc.log" 2>&1; echo "rocprof enc rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && head -24 "$F" | cut -c1-170
F=$(find "$OUT/rocprof_enc" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && head -8 "$F" | cut -c1-170
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete; find "$OUT" -name '*agent_info.csv' -delete
timeout 600 python bench.py --model large-v3 --steps 5 --warmup 2 --no-cpu-baseline --no-stream --no-pmc > "$OUT/bench_large_v3.json" 2> "$OUT/bench_large_v3.err"
python -c "import json; d=json.loads(open('$OUT/bench_large_v3.json').read().strip().splitlines()[-1]); print('large-v3 xRT', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), d['stage_ms'], 'step graph ms', d['decode_step']['graph_replay_ms'])" || tail -3 "$OUT/bench_large_v3.err"
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"; echo "config5 rc=$?"
python -c "import json; d=json.loads(open('$OUT/bench_config5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['ms_per_step'], d.get('decode_step'))"
for extra in "" "--stream-batch"; do
  timeout 600 python bench.py --model small --stream-clients 4 $extra --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > "$OUT/bench_small_4clients$extra.json" 2> "$OUT/bench_small_4clients$extra.err"
  python -c "import json; d=json.loads(open('$OUT/bench_small_4clients$extra.json').read().strip().splitlines()[-1]); print('small 4 clients $extra', round(d['value'],1), d.get('stream'))" || tail -3 "$OUT/bench_small_4clients$extra.err"
done
for B in 4 6; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stream --no-pmc --batch $B > "$OUT/bench_batch$B.json" 2> "$OUT/bench_batch$B.err"
  python -c "import json; d=json.loads(open('$OUT/bench_batch$B.json').read().strip().splitlines()[-1]); print('batch', $B, 'xRT', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2))" || tail -3 "$OUT/bench_batch$B.err"
done
du -sh "$OUT"
