#!/bin/bash
# round 3, call D: persistent-layer prototype on both phase tables; dedicated-queue cap; configs[2] with unlocked client phases and
# the two-lane batch worker; config 5 with two lanes
set -u
TAG=r3d; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
( cd scripts/ubench && timeout 120 ./persist_layer 12 30 && timeout 120 ./persist_layer 12 30 fused ) > "$OUT/ubench_persist_layer.txt" 2>&1; echo "ubench rc=$?"; grep -E "table|chain|persistent|engine" "$OUT/ubench_persist_layer.txt"
python - <<'PY'
import ctypes as C
h = C.CDLL("/opt/rocm/lib/libamdhip64.so")
st = C.c_void_p(); mask = (C.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
print("hipExtStreamCreateWithCUMask rc", h.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask))
fl = C.c_uint(99); print("hipStreamGetFlags rc", h.hipStreamGetFlags(st, C.byref(fl)), "flags", fl.value, "(0 = hipStreamDefault: blocking w.r.t. the null stream, 1 = non-blocking)")
PY
run_b() {  # name, bench args..., -- env...
  name=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 600 python bench.py "${args[@]}" --no-cpu-baseline --no-pmc > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "p50", round(d.get("p50_chunk_latency_ms", d.get("p50_step_ms", 0)), 2))
    st = d.get("stream")
    if st: print("   stream:", {k: (round(v["p50_chunk_latency_ms"], 2), round(v["p95_chunk_latency_ms"], 2), round(v["xrt"], 1)) for k, v in st.items() if isinstance(v, dict) and "xrt" in v})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run_b s4 --streams 4 --steps 10 --warmup 2 --no-stream -- A=1
run_b s8 --streams 8 --steps 6 --warmup 2 --no-stream -- A=1
run_b small_4clients --model small --stream-clients 4 --steps 3 --warmup 1 -- A=1
run_b small_4clients_batch --model small --stream-clients 4 --stream-batch --steps 3 --warmup 1 -- A=1
run_b config5 --config 5 --steps 2 --warmup 1 -- A=1
du -sh "$OUT"
