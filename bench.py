#!/usr/bin/env python
"""bench.py — the reference's headline metric on MI355X: real-time factor (xRT) + p50 chunk latency,
Whisper-small, 30 s window (BASELINE.json), measured on the HIP hot path through the C-ABI.

A "step" = one pass of the per-chunk hot path over one 30 s window of synthetic PCM already resident in HBM:
log-mel (480 000 samples -> [80, 3001]) -> encoder ([1, 80, 3000] -> [1, 1500, 768] + cross-attention K/V of all
decoder layers) -> beam-search decode (beam 5, patience 1, T = 0, Whisper timestamp rules, exactly
`--decode-steps` generated tokens: EOT is suppressed so the length is fixed; weights are seeded random values in the
exact Whisper-small shapes — there are no checkpoints offline; timing is value-independent at fixed length).
That is what whisper_live/backend/base.py:123-131 times per chunk (the chunk is zero-padded to one full window,
transcriber_faster_whisper.py:1125-1127), so xRT = 30 s x steps / wall and p50 chunk latency = median step time.

Multi-GPU: streams are independent — one process per GPU, one engine + one stream each, no
data-path collective; barrier + max-over-ranks timing; value = aggregate audio seconds / wall ("weak" scaling).

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant decode kernel,
timed inside the engine on the slot's stream: a captured graph of just that kernel's launches of one step between
one HIP-event pair) and `cpu_baseline` (the torch-fp32 CPU oracle on a
bounded sample of the same workload, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from whisperlive_amd.synthetic import energy_following_vad_weights, speech_like_pcm  # noqa: E402  (numpy-only generators)


HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
WINDOW_S = 30.0
COLL_DEVICE = "cuda"            # device of the tensors handed to torch.distributed (RCCL needs device tensors; the gloo rehearsal uses host ones)


def token_ids(vocab: int):
    tb = vocab - 1501
    return dict(sot=tb - 106, eot=tb - 107, no_timestamps=tb - 1, timestamp_begin=tb, no_speech=tb - 2, blank=220)


def suppress_list(ids, with_eot: bool):
    tb = ids["timestamp_begin"]
    base = sorted({1, 2, 7, 8, 9, 10, 14, 25, tb - 5, tb - 6, ids["sot"], tb - 3, tb - 4})
    return base + ([ids["eot"]] if with_eot else [])


def encoder_flops(spec) -> float:
    """algorithmic FLOPs of one 30 s window through the encoder + the cross-attention K/V projection:
    conv1 + conv2 as GEMMs, per layer 4 d^2 + 2 d F projections and 4 T^2 d of attention, cross K/V of every decoder layer"""
    d, F, T = spec.d_model, spec.ffn, spec.n_audio_ctx
    conv = 2.0 * 2 * T * d * 3 * spec.n_mels + 2.0 * T * d * 3 * d
    layer = 2.0 * T * (4 * d * d + 2 * d * F) + 4.0 * T * T * d
    return conv + spec.enc_layers * layer + 2.0 * T * d * 2 * d * spec.dec_layers


MFMA_PEAK_TFLOPS = 2500.0      # dense fp16 / bf16 MFMA peak of an MI355X (MI355X_MICROARCH.md; AMD's 5 PF figure includes 2:1 sparsity)


def decode_step_bytes(spec, beams: int, t: int) -> int:
    """fp16 weights + cross-attention K/V + self-attention KV read per decode step."""
    d, F, L, V, T = spec.d_model, spec.ffn, spec.dec_layers, spec.vocab, spec.n_audio_ctx
    return 2 * (L * (6 * d * d + 2 * d * F) + V * d) + 2 * L * 2 * T * d + 2 * L * 2 * t * d * beams


class pinned_threads:
    """torch on exactly `n` threads, the process pinned to the first `n` CPUs it may use (restored on exit): the 16-thread figure moved
    2.5x between rounds with the host's other load (3.2 / 7.0 / 8.0 xRT for the same code) — unpinned threads migrate."""

    def __init__(self, n: int):
        self.n, self.old_aff, self.old_thr = n, None, None

    def __enter__(self):
        import torch
        self.old_thr = torch.get_num_threads()
        torch.set_num_threads(self.n)
        try:
            self.old_aff = os.sched_getaffinity(0)
            os.sched_setaffinity(0, set(sorted(self.old_aff)[: self.n]))
        except (AttributeError, OSError):
            self.old_aff = None
        return self

    def __exit__(self, *exc):
        import torch
        if self.old_aff is not None:
            try:
                os.sched_setaffinity(0, self.old_aff)
            except OSError:
                pass
        torch.set_num_threads(self.old_thr)
        return False


def cpu_baseline(spec, weights, pcm, ids, decode_steps: int, full_steps: int, threads: int, repeats: int = 1, int8: str = None):
    """The CPU oracle (numpy log-mel + torch network + numpy beam search) on a BOUNDED sample of the same workload: the whole
    front-end and encoder of one 30 s window and `decode_steps` of the `full_steps` beam-5 decode steps (the decode time is scaled
    to `full_steps` when fewer are run; per-step cost is flat in t at this length), `repeats` times on pinned threads — the
    MEDIAN window time is reported. int8 = "fbgemm": every linear layer as torch's dynamic-quantised int8 Linear (kind
    "port-int8": the arithmetic class of the reference's CPU default, CTranslate2 int8, faster_whisper_backend.py:93 — CT2 itself
    cannot be installed offline). Evaluated on the engine's fp16-rounded weights, so the fp32 tokens are also the parity reference
    of the timed path (`parity_prefix`). Returns (baseline dict, generated tokens of the last bounded decode)."""
    from oracle import decoding as odec
    from oracle import logmel as olm
    from oracle import model as omodel
    from oracle.provider import NetProvider

    with pinned_threads(threads):
        oracle = omodel.WhisperOracle(omodel.Spec(spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers, spec.dec_layers,
                                                  spec.ffn, spec.vocab), weights, int8=int8)
        ests, parts, res = [], [], None
        if repeats > 1:
            # untimed warm-up (thread pool start, first-touch of the weights, allocator): the first of three runs was 2x the others
            # (7.99 / 3.54 / 3.66 s, profiles/r5final_bench_default.json) — the median hid it, the reported spread did not
            enc_w = oracle.encode(olm.pad_or_trim(olm.log_mel_spectrogram(pcm, spec.n_mels, precise=False)[:, :-1])[None])
            odec.generate(NetProvider(oracle, enc_w), [ids["sot"]], odec.GenOptions(ids=odec.TokenIds(**ids), beam_size=5, patience=1.0, max_length=1 + 2,
                                                                                   suppress_tokens=suppress_list(ids, True)))
        max_runs = max(1, repeats) if repeats <= 1 else 2 * repeats
        for _ in range(max_runs):
            # `repeats` > 1: run until the LAST `repeats` runs agree to 15 % (a shared host's first runs carry other tenants' noise: the
            # driver's round-5 run spread 38 % over three), at most 2 x `repeats` runs; every run's time is reported
            if repeats > 1 and len(ests) >= repeats:
                tail = ests[-repeats:]
                if (max(tail) - min(tail)) / sorted(tail)[len(tail) // 2] <= 0.15:
                    break
            t0 = time.perf_counter()
            feats = olm.log_mel_spectrogram(pcm, spec.n_mels, precise=False)
            t1 = time.perf_counter()
            enc = oracle.encode(olm.pad_or_trim(feats[:, :-1])[None])
            t2 = time.perf_counter()
            o = odec.GenOptions(ids=odec.TokenIds(**ids), beam_size=5, patience=1.0, max_length=1 + decode_steps,
                                suppress_tokens=suppress_list(ids, True))
            res = odec.generate(NetProvider(oracle, enc), [ids["sot"]], o)
            t3 = time.perf_counter()
            per_step = (t3 - t2) / max(1, res.steps)
            ests.append((t1 - t0) + (t2 - t1) + per_step * full_steps)
            parts.append((t1 - t0, t2 - t1, per_step, res.steps, t3 - t0))
    first = max(0, len(ests) - max(1, repeats))              # the runs the figure is taken from: the last `repeats`
    order = sorted(range(first, len(ests)), key=lambda i: ests[i])
    mid = order[len(order) // 2]
    lm, en, per_step, nst, tot = parts[mid]
    arith = "torch-fp32" if int8 is None else "torch dynamic-int8 (fbgemm) linears, fp32 elsewhere"
    scaled = "measured in full" if nst >= full_steps else f"measured for {nst} steps and scaled to {full_steps}"
    return dict(value=WINDOW_S / ests[mid], unit="xRT (audio s / wall s)", cores=threads, kind="port" if int8 is None else "port-int8",
                runs=len(ests), window_s_all_runs=[round(e, 3) for e in ests],
                spread=(max(ests[first:]) - min(ests[first:])) / ests[mid],
                sample=f"one 30 s window, median of the last {len(ests) - first} of {len(ests)} run{'s' if len(ests) != 1 else ''} on {threads} pinned thread{'s' if threads != 1 else ''}: "
                       f"numpy log-mel {lm:.2f} s + {arith} encoder {en:.2f} s measured in full; beam-5 decode {per_step * 1e3:.0f} ms/step, {scaled}; "
                       f"{sum(p[4] for p in parts):.1f} s of CPU work in total. CTranslate2-int8 (the reference's CPU backend) cannot be installed "
                       f"offline, so this is the repo's own port"), res.sequences_ids[0]


def usable_cpus(cap: int = 16) -> int:
    """Host threads this process may really use: the affinity mask and the cgroup CPU quota, not os.cpu_count() (a
    container on a 256-thread host reports 256 and is scheduled on far fewer — torch with 256 threads then crawls)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def note(msg: str):
    """progress on stderr (the JSON line owns stdout)"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def f16_rounded(weights):
    """the engine stores projection matrices in fp16: the CPU leg evaluates the same rounded values"""
    return {k: (v.astype(np.float16).astype(np.float32) if v.ndim >= 2 and "embed_positions" not in k else v)
            for k, v in weights.items()}


FETCH_KIB_TO_BYTES = 1024.0 * 2.0     # FETCH_SIZE is in KiB; on gfx950 it tallies the 128-B requests of wide (16 B per lane)
                                      # coalesced reads at 64 B (MI355X guide, HBM section) — checked per run, see below


def measured_traffic(kernel_name: str, model: str, timeout_s: int = 300, child_args=()):
    """HBM bytes per launch of `kernel_name`, measured BY THIS RUN: a child process of this script (a few decode steps of
    the same workload) under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` — counters in their own pass, as the MI355X guide
    prescribes — and its counter_collection.csv reduced here. Returns (bytes, source, extra): `extra` carries the RAW
    counter mean (KiB) and a calibration of the KiB -> bytes factor inside the SAME pass: the vocabulary projection
    streams its V x d fp16 weight image exactly once per launch (tests pin that it reads nothing else of size), so
    factor = 2 V d / (raw KiB x 1024) must come out at ~2.0 for the x 1024 x 2 conversion to be right on this box."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found", {}
    d = tempfile.mkdtemp(prefix="wlx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [rp, "--pmc", "FETCH_SIZE", "--kernel-trace", "-d", d, "-o", "wlx", "--output-format", "csv", "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child", "--model", model] + list(child_args)
    try:
        proc = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        tot = n = 0.0
        per = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    if r["Counter_Name"] != "FETCH_SIZE":
                        continue
                    a = per.setdefault(r["Kernel_Name"], [0.0, 0])
                    a[0] += float(r["Counter_Value"]); a[1] += 1
                    if kernel_name in r["Kernel_Name"]:
                        tot += float(r["Counter_Value"])
                        n += 1
        if not n:
            return None, f"no FETCH_SIZE rows for {kernel_name} (rocprofv3 rc {proc.returncode}: {proc.stderr[-200:]})", {}
        from whisperlive_amd.specs import get_spec
        sp = get_spec(model)
        return reduce_fetch_pass(per, kernel_name, 2.0 * sp.vocab * sp.d_model)
    except Exception as e:  # noqa: BLE001 — the headline line must survive a failed counter pass
        return None, f"{type(e).__name__}: {e}", {}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def reduce_fetch_pass(per, kernel_name: str, vocab_bytes: float):
    """`per`: {kernel name: [sum of raw FETCH_SIZE (KiB), launches]} of ONE counter pass -> (bytes per launch of `kernel_name`, source, detail).
    The KiB -> bytes factor is calibrated inside the SAME pass: the vocabulary projection (`dec_vocab_kernel`) streams its V x d fp16 weight
    image exactly once per launch and nothing else of size, so known bytes / (raw KiB x 1024) must come out at ~2.0 for the x 1024 x 2
    conversion (MI355X guide, HBM section) to hold on this box. Outside [1.8, 2.2] the converted figure is NOT reported (None + why)."""
    hit = [(k, v) for k, v in per.items() if kernel_name in k]
    tot, n = sum(v[0] for _, v in hit), sum(v[1] for _, v in hit)
    if not n:
        return None, f"no FETCH_SIZE rows for {kernel_name}", {}
    extra = {"raw_fetch_size_kib_per_launch": tot / n, "kib_to_bytes_factor_used": FETCH_KIB_TO_BYTES}
    voc = {k: v for k, v in per.items() if "dec_vocab_kernel" in k}
    if not voc:
        extra["calibration"] = {"error": "no dec_vocab_kernel rows in the pass"}
        return None, "FETCH_SIZE factor not calibrated: no dec_vocab_kernel launch in the counter pass", extra
    big = max(voc.items(), key=lambda kv: kv[1][1])
    raw_big = big[1][0] / big[1][1]
    factor = vocab_bytes / (raw_big * 1024.0)
    extra["calibration"] = {"kernel": big[0][:80], "launches": big[1][1], "raw_kib_per_launch": raw_big, "known_bytes": vocab_bytes,
                            "bytes_per_raw_kib_over_1024": factor, "accepted_range": [1.8, 2.2]}
    if not 1.8 <= factor <= 2.2:
        return None, (f"FETCH_SIZE factor calibrated at {factor:.2f} on {big[0][:40]} — outside [1.8, 2.2]: the x1024x2 conversion "
                      f"does not hold in this pass, traffic withheld"), extra
    return tot / n * FETCH_KIB_TO_BYTES, f"rocprofv3 --pmc FETCH_SIZE pass of this run ({int(n)} launches)", extra


def rocprof_kernel_avg(kernel_name: str, model: str, timeout_s: int = 240, child_args=()):
    """Average duration (us) of `kernel_name` from a `rocprofv3 --kernel-trace --stats` pass of the SAME child workload as
    measured_traffic() (no counters in this pass), so that roofline.frac can be reproduced from a rocprofv3 summary without the
    HIP-event vs. profiler ambiguity: returns (avg_us, calls, source) or (None, 0, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, 0, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="wlx_kt_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [rp, "--kernel-trace", "--stats", "-d", d, "-o", "wlx", "--output-format", "csv", "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child", "--model", model] + list(child_args)
    try:
        proc = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    if kernel_name in r["Name"]:
                        keep = os.path.join(ROOT, "gpurun_out")
                        if os.path.isdir(keep):
                            shutil.copy(f, os.path.join(keep, "bench_kernel_stats.csv"))
                        return float(r["AverageNs"]) / 1e3, int(r["Calls"]), "rocprofv3 --kernel-trace --stats pass of this run"
        return None, 0, f"no kernel_stats row for {kernel_name} (rocprofv3 rc {proc.returncode}: {proc.stderr[-200:]})"
    except Exception as e:  # noqa: BLE001
        return None, 0, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def respawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` started by hand (no RANK/WORLD_SIZE in the environment): launch the N ranks ourselves,
    exactly the way the driver does — one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def ws_client_main(spec_json: str) -> int:
    """`bench.py --ws-client '<json>'`: ONE stock-client stand-in in its own process (whisper_live/client.py:433,547: float32
    16 kHz PCM in 4096-sample binary packets, then END_OF_AUDIO): connect, wait for SERVER_READY, build the PCM, start on the
    common `go` instant (paced clients offset by k / clients of the packet period: independent clients are not phase-locked),
    count the segment messages that come back. Prints one JSON line."""
    import threading
    from whisperlive_amd import ws
    a = json.loads(spec_json)
    res = {"segments": 0, "error": None}
    try:
        c = ws.connect(f"ws://127.0.0.1:{a['port']}")
        c.send(json.dumps(dict(uid=f"bench{a['k']}", language=a["language"], task="transcribe", model=a["model"], use_vad=True,
                               no_speech_thresh=1.0, same_output_threshold=10)))
        if json.loads(c.recv(timeout=30)).get("message") != "SERVER_READY":
            raise RuntimeError("no SERVER_READY")
        pcm = stream_pcm(a["secs"], a["seed"])
        packets = [pcm[i: i + 4096].tobytes() for i in range(0, pcm.shape[0], 4096)]

        def drain():
            try:
                while True:
                    if "segments" in json.loads(c.recv()):
                        res["segments"] += 1
            except Exception:  # noqa: BLE001 — closed
                return
        rd = threading.Thread(target=drain, daemon=True)
        rd.start()
        pace_s = a["pace_s"]
        time.sleep(max(0.0, a["go"] - time.time()) + (pace_s * a["k"] / max(1, a["clients"]) if pace_s else 0.0))
        t0 = time.perf_counter()
        for i, pk in enumerate(packets):
            c.send(pk)
            if pace_s:
                time.sleep(max(0.0, t0 + (i + 1) * pace_s - time.perf_counter()))
        time.sleep(a["settle_s"])
        c.send(b"END_OF_AUDIO")
        rd.join(5)
    except Exception as e:  # noqa: BLE001
        res["error"] = f"{type(e).__name__}: {e}"
    print(json.dumps(res), flush=True)
    return 0


def stream_through_server(make_transcriber, seconds: float = 40.0, paced_seconds: float = 4.0, settle_s: float = 3.0,
                          pcm_fn=None, clients: int = 1, batch: bool = False, model_name: str = "small.en", language="en"):
    """BASELINE configs[1] (and, with clients=4, configs[2]) in their literal form: WebSocket streams against the
    TranscriptionServer shell (whisperlive_amd/server.py), VAD on, float32 16 kHz PCM in 4096-sample packets (the stock
    client's packet, whisper_live/client.py:433,547), first unpaced (throughput) and then paced at 256 ms per packet
    (interactive latency). Numbers are the reference's own counters (whisper_live/backend/base.py:123-131,
    whisper_live/metrics.py:100-107): xrt = sum(audio s) / sum(latency) over every chunk any session thread transcribed
    (a per-stream rate: with N clients the aggregate is up to N x that), p50 = median chunk latency. The transcriber is
    the product one (WhisperModelHIP: VAD gate -> log-mel -> encoder -> beam search -> segments), decode length pinned.
    batch=True starts the per-GPU BatchInferenceWorker (the reference's --batch_inference mode)."""
    import subprocess
    import threading
    from whisperlive_amd import metrics
    from whisperlive_amd.serve_client import ServeClientHIP
    from whisperlive_amd.server import TranscriptionServer

    tr = make_transcriber()
    ServeClientHIP.MODELS.clear()
    srv, ready = TranscriptionServer(), threading.Event()
    th = threading.Thread(target=srv.run, args=("127.0.0.1",), daemon=True,
                          kwargs=dict(port=0, ready=ready, single_model=True, max_clients=clients, batch_enabled=batch,
                                      batch_max_size=max(clients, 1), batch_window_ms=5, model_factory=lambda m, d: tr))
    th.start()
    if not ready.wait(10):
        raise RuntimeError("server did not start")
    out = {}

    def run(tag, secs, seed, pace_s):
        # the clients are SEPARATE PROCESSES (`bench.py --ws-client ...`), as real clients are: four unpaced senders inside
        # this interpreter held its GIL and charged their own packet loops to the server's chunk latency
        metrics.snapshot(reset=True)
        go = time.time() + 2.5                         # every client connects, builds its PCM, then starts on this instant
        cmd = [sys.executable, os.path.abspath(__file__), "--ws-client"]
        procs = [subprocess.Popen(cmd + [json.dumps(dict(port=srv.port, k=k, clients=clients, secs=secs, seed=seed + k, pace_s=pace_s,
                                                          settle_s=settle_s, go=go, model=model_name, language=language))],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(clients)]
        t0 = time.perf_counter()
        time.sleep(max(0.0, go - time.time()) + secs * (pace_s / 0.256 if pace_s else 0.0) + settle_s * 0.9)   # snapshot while every client is still connected
        snap = metrics.snapshot()
        got, errs = [], []
        for pr in procs:
            try:
                o, e = pr.communicate(timeout=90)
                r = json.loads(o.strip().splitlines()[-1])
                got.append(r.get("segments", 0))
                if r.get("error"):
                    errs.append(r["error"])
            except Exception as ex:  # noqa: BLE001
                pr.kill()
                errs.append(f"{type(ex).__name__}: {ex}")
        out[tag] = dict(clients=clients, audio_sent_s_per_client=secs, wall_s=time.perf_counter() - t0, chunks=snap["chunks"],
                        audio_processed_s=snap["audio_s"], xrt=snap["xrt"],
                        p50_chunk_latency_ms=None if snap["p50_latency_s"] is None else 1e3 * snap["p50_latency_s"],
                        p95_chunk_latency_ms=None if snap["p95_latency_s"] is None else 1e3 * snap["p95_latency_s"],
                        segment_messages=sum(got), errors=snap["errors"], client_errors=errs)

    try:
        run("unpaced", seconds, 4321, 0.0)
        run("paced_256ms", paced_seconds, 4400, 0.256)
    finally:
        srv.shutdown()
        th.join(5)
        ServeClientHIP.MODELS.clear()
        try:
            tr.close()                 # its pooled engine slots: a live slot counts against the device's dedicated hardware queues
        except Exception:  # noqa: BLE001
            pass
    return out


def make_bench_transcriber(eng, spec, ids, decode_steps, vad_model=None, max_batch=1):
    """The product transcriber on an already-built engine with the decode length pinned like the window benchmark: EOT
    suppressed, max_length = prompt + decode_steps, quality fallbacks off (they never trigger on real speech; random
    weights would trigger all five re-decodes)."""
    from whisperlive_amd.tokenizer import synthetic_tokenizer
    from whisperlive_amd.transcriber import WhisperModelHIP, _EngineModel

    class FixedLengthModel(_EngineModel):
        def generate(self, encoder_output, prompts, **kw):
            kw["max_length"] = max(len(p) for p in prompts) + decode_steps
            kw["suppress_tokens"] = list(kw.get("suppress_tokens") or (-1,)) + [ids["eot"]]
            return super().generate(encoder_output, prompts, **kw)

    class BenchTranscriber(WhisperModelHIP):
        stage_log = []                 # per transcribe() call: wall ms and the device stages (HIP events on the slot stream; VAD: its own events)

        def transcribe(self, audio, **kw):
            kw.update(temperature=0.0, compression_ratio_threshold=None, log_prob_threshold=None, no_speech_threshold=None)
            t0 = time.perf_counter()
            segs, info = super().transcribe(audio, **kw)
            wall = 1e3 * (time.perf_counter() - t0)
            if info is not None:      # random weights give a flat language distribution; real speech locks the language (> 0.5)
                info.language_probability = max(info.language_probability, 0.99)
                try:
                    tm = self._slot().timings()
                    vm = self.vad_model
                    BenchTranscriber.stage_log.append(dict(wall_ms=wall, vad_ms=float(getattr(vm, "last_device_ms", 0.0)) if kw.get("vad_filter") else 0.0,
                                                           logmel_ms=tm["logmel_ms"], encode_ms=tm["encode_ms"], generate_ms=tm["generate_ms"],
                                                           decode_steps=tm["decode_steps"]))
                except Exception:  # noqa: BLE001 — an extra; the leg's own counters do not depend on it
                    pass
            return segs, info

    tr = BenchTranscriber("bench", engine=eng, hf_tokenizer=synthetic_tokenizer(spec.vocab), vad_model=vad_model, max_batch=max_batch)
    tr.model = FixedLengthModel(tr)
    return tr


def stream_pcm(seconds: float, seed: int = 1234) -> np.ndarray:
    """The stream leg's audio: `speech_like_pcm` (2.5 s phrases / 1.0 s pauses) with a 3 s noise-only stretch in every 12 s,
    i.e. longer than the gate's min_silence_duration_ms = 2000 — audio the VAD has to CUT, not only to look at."""
    pcm = speech_like_pcm(seconds, seed).copy()
    t = np.arange(pcm.shape[0]) / 16000.0
    quiet = (t % 12.0) >= 9.0
    noise = np.random.default_rng(seed + 7).normal(0.0, 0.003, pcm.shape[0]).astype(np.float32)
    pcm[quiet] = noise[quiet]
    return pcm


def stream_leg(eng, spec, ids, decode_steps, pcm_fn, clients=1, batch=False, model_name="small.en"):
    """configs[1] / configs[2] through the server shell on the already-built engine (make_bench_transcriber). VAD: the
    Silero network on the GPU (libwlx.so wlx_vad_*) with seeded weights whose probabilities FOLLOW the audio's energy
    (whisperlive_amd/synthetic.py::energy_following_vad_weights — no Silero weight file exists offline): phrases pass, the 3 s
    noise-only stretches of `stream_pcm` are cut, so speech segmentation, `collect_chunks` and `restore_speech_timestamps`
    all run with real effect inside the timed path, and the network costs exactly what the real one costs."""
    from whisperlive_amd import vad

    w = energy_following_vad_weights(3)
    vm = vad.SileroHIPModel(w, device=eng.device)
    probe = pcm_fn(24.0, 4321)
    spans = vad.get_speech_timestamps(probe, vad.VadOptions(threshold=0.5), model=vm)
    kept = sum(c["end"] - c["start"] for c in spans) / float(probe.shape[0])

    def make():
        return make_bench_transcriber(eng, spec, ids, decode_steps, vad_model=vm, max_batch=max(1, clients if batch else 1))
    english_only = model_name.endswith("en")
    from whisperlive_amd.batching import BatchInferenceWorker
    saved_t = BatchInferenceWorker.TEMPERATURES
    BatchInferenceWorker.TEMPERATURES = (0.0,)        # the batch worker's own fallback ladder, off like the transcriber's
    try:
        res = stream_through_server(make, pcm_fn=pcm_fn, clients=clients, batch=batch, model_name=model_name,
                                    language="en" if english_only else None)
    finally:
        BatchInferenceWorker.TEMPERATURES = saved_t
        vm.close()
    try:       # where a chunk's latency goes (all chunks of both runs): medians of the device stages, the rest is host + synchronisation
        from whisperlive_amd.transcriber import WhisperModelHIP as _W
        logs = [c for k in _W.__subclasses__() for c in getattr(k, "stage_log", [])]
        if logs:
            med = {k: float(np.median([c[k] for c in logs])) for k in ("wall_ms", "vad_ms", "logmel_ms", "encode_ms", "generate_ms", "decode_steps")}
            med["host_and_sync_ms"] = med["wall_ms"] - med["vad_ms"] - med["logmel_ms"] - med["encode_ms"] - med["generate_ms"]
            med["chunks"] = len(logs)
            res["stage_ms_per_chunk"] = med
        for k in _W.__subclasses__():
            if hasattr(k, "stage_log"):
                k.stage_log = []
    except Exception as e:  # noqa: BLE001
        res["stage_ms_per_chunk"] = {"error": f"{type(e).__name__}: {e}"}
    res["vad"] = dict(model="Silero architecture on the GPU, seeded energy-following weights", threshold=0.5,
                      speech_spans_in_24s_probe=len(spans), audio_kept_fraction=kept)
    res["config"] = (f"configs[{1 if clients == 1 else 2}]: {clients} WebSocket stream{'s' if clients > 1 else ''} -> TranscriptionServer -> "
                     f"ServeClientHIP{' -> BatchInferenceWorker' if batch else ''} -> WhisperModelHIP.transcribe, Whisper-{model_name}, "
                     f"VAD on (Silero on GPU), beam 5, {decode_steps} tokens per window"
                     + ("" if english_only else ", language detected on the first chunk"))
    return res


def config5(args, rank, world, local, dist, torch):
    """BASELINE configs[4] ("config 5"): batch_inference.py's batched mode — `--clips` pre-recorded 30 s clips,
    Whisper-large-v3 shapes, sharded in contiguous blocks over the ranks (whisperlive_amd/sharding.py); every rank drives
    its block through its own BatchInferenceWorker(max_batch_size=8) exactly as the reference's server would for N
    clients (whisper_live/server.py:665-673, batch_inference.py:155-438: per-item log-mel, ONE batched encode, ONE
    batched beam-5 decode per batch), then ONE fixed-size all_gather of 2 KiB result records over RCCL. A "step" = one
    pass over all clips; value = clips x 30 s x steps / max-over-ranks wall."""
    from whisperlive_amd import sharding as sh
    from whisperlive_amd.batching import BatchInferenceWorker, BatchRequest
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import get_spec
    from whisperlive_amd.weights import random_weights

    spec = get_spec(args.model)
    eng = HipWhisperEngine(spec, random_weights(spec, seed=0), device=local)
    ids = token_ids(spec.vocab)
    MB = max(1, min(64, args.max_batch))
    tr = make_bench_transcriber(eng, spec, ids, args.decode_steps, vad_model=None, max_batch=MB)
    BatchInferenceWorker.TEMPERATURES = (0.0,)
    worker = BatchInferenceWorker(tr, max_batch_size=MB, batch_window_ms=50, lanes=max(1, args.lanes))
    worker.start()
    n = args.clips
    lo, hi = sh.shard_range(n, rank, world)
    clips = [speech_like_pcm(WINDOW_S, seed=2000 + i) if lo <= i < hi else None for i in range(n)]
    lang = "en"
    process = sh.worker_block_processor(worker, lambda c: BatchRequest(audio=c, language=lang, use_vad=False))
    device = f"cuda:{local}" if COLL_DEVICE == "cuda" else "cpu"     # where the gathered records live (host tensors in the gloo rehearsal)

    def step():
        t0 = time.perf_counter()
        res = sh.transcribe_clips_sharded(clips, process, rank=rank, world=world, dist=dist, device=device)
        return time.perf_counter() - t0, res

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    lat, res = [], None
    for _ in range(args.steps):
        dt, res = step()
        lat.append(dt)
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=COLL_DEVICE)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    out = None
    if rank == 0:
        n_tok = [len(r[0]) for r in res]
        slot = tr._slots[0] if tr._slots else None
        rows = MB * 5
        step_ms = None
        try:
            if slot is not None:
                step_ms = slot.debug_time_decode_step(rows=rows, t=1 + args.decode_steps // 2, iters=20)
        except Exception as e:  # noqa: BLE001 — the step probe is an extra; the measured line must survive it
            note(f"decode-step probe at {rows} rows unavailable: {e}")
        sb = decode_step_bytes(spec, rows, 1 + args.decode_steps // 2) + 2 * spec.dec_layers * 2 * spec.n_audio_ctx * spec.d_model * (MB - 1)
        out = {
            "metric": "real-time factor (xRT), batched mode: pre-recorded 30 s clips through batch_inference's worker",
            "value": n * WINDOW_S * args.steps / wall, "unit": "xRT (audio s / wall s)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * wall / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16 (MFMA operands; f32 accumulate, f32 residual stream, f32 log-mel)", "data": "synthetic",
            "config": {"workload": f"configs[4]: batch_inference.py batched mode, Whisper-{args.model} shapes, {n} clips x 30 s "
                                   f"(seeds 2000..{2000 + n - 1}), contiguous blocks over {world} GPU(s), per-GPU "
                                   f"BatchInferenceWorker(max_batch_size={MB}, lanes={max(1, args.lanes)}) -> per-item log-mel + one batched encode + one batched "
                                   f"beam-5 decode of {args.decode_steps} tokens per batch -> one all_gather of 2 KiB records; "
                                   f"seeded random weights",
                       "clips": n, "clips_per_gpu": hi - lo, "max_batch_size": MB, "worker_lanes": max(1, args.lanes), "beam_size": 5, "decode_steps": args.decode_steps,
                       "window_s": WINDOW_S},
            "tokens_per_clip": {"min": min(n_tok), "max": max(n_tok)},
            "p50_step_ms": 1000.0 * float(np.median(lat)),
        }
        if step_ms is not None:
            out["decode_step"] = {"rows": rows, "graph_replay_ms": step_ms, "algorithmic_bytes": sb,
                                  "hbm_frac_of_peak": sb / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # roofline of the DOMINANT kernel of the 40-row step (HIP-event timed inside the engine, as in the headline run),
            # its HBM traffic from a live FETCH_SIZE pass of the same batched decode
            try:
                prof = slot.debug_profile_step(rows=rows, t=1 + args.decode_steps // 2, iters=10)
                dom = max(prof, key=lambda k: k["total_us"])
                roof = dict(bound="hbm", kernel=dom["name"], achieved=dom["bytes_per_launch"] / (dom["avg_us"] * 1e-6) / 1e9,
                            peak=HBM_PEAK_GBS, unit="GB/s", traffic=None, launches_per_decode_step=dom["launches"],
                            avg_us=dom["avg_us"], algorithmic_bytes_per_launch=dom["bytes_per_launch"], rows=rows)
                roof["frac"] = roof["achieved"] / roof["peak"]
                out["decode_step"]["kernels"] = prof
                if world == 1 and not args.no_pmc:
                    note("rocprofv3 FETCH_SIZE pass (batched decode, 8 x 5 rows)")
                    roof["traffic"], roof["traffic_source"], roof["traffic_detail"] = measured_traffic(
                        dom["name"], args.model, timeout_s=420, child_args=["--batch", str(MB)])
                out["roofline"] = roof
            except Exception as e:  # noqa: BLE001
                note(f"per-kernel probe at {rows} rows unavailable: {e}")
                out["roofline"] = dict(bound="hbm", kernel=f"decode step (all launches, {rows} rows)", achieved=sb / (step_ms * 1e-3) / 1e9,
                                       peak=HBM_PEAK_GBS, unit="GB/s", frac=sb / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None)
    worker.stop()
    tr.close()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if COLL_DEVICE == "cpu":
            out["rehearsal"] = True          # WLX_BENCH_REHEARSAL: ranks shared a GPU over gloo — not a measurement
        if dist is not None:
            out["collectives"] = {"backend": "gloo (host tensors)" if COLL_DEVICE == "cpu" else "nccl (RCCL, device tensors)", "world_size": world}
        print(json.dumps(out))


def throughput_leg(eng, spec, torch, ids, eids, gen_kw, decode_steps, TS=3, TB=48, tsteps=2):
    """The throughput configuration of ONE GPU, timed by the same driver run (VERDICT r03 task 5): TS slots on their own hardware
    queues x TB windows batched into every decode — what a --batch_inference server with TS lanes and --batch_max_size TB
    runs (round 5: 3 x 48; a slot held at most 12 windows until then). Called after the headline's own slot is closed: a fifth live
    slot would send every slot back to the shared queue pool."""
    from concurrent.futures import ThreadPoolExecutor
    tslots = [eng.create_slot(TB, 5) for _ in range(TS)]
    try:
        for i, sl in enumerate(tslots):
            for b in range(TB):
                sl.pcm_put(speech_like_pcm(WINDOW_S, seed=5000 + 100 * i + b), b)

        def tstep(sl):
            Ts_ = [sl.logmel_resident(b) for b in range(TB)]
            sl.encode(TB, seek=[0] * TB, seg=[min(T - 1, 3000) for T in Ts_])
            return sl.generate([[ids["sot"]]] * TB, eids, **gen_kw)[0]
        with ThreadPoolExecutor(max_workers=TS) as tp:
            list(tp.map(tstep, tslots))                       # warm-up (graph capture per slot)
            torch.cuda.synchronize()
            tt0 = time.perf_counter()
            for _ in range(tsteps):
                list(tp.map(tstep, tslots))
            torch.cuda.synchronize()
            twall = time.perf_counter() - tt0
        tm12 = tslots[0].timings()
        # one slot alone: the batched encoder's MFMA fraction without the other slots' work beside it
        Ts_ = [tslots[0].logmel_resident(b) for b in range(TB)]
        tslots[0].encode(TB, seek=[0] * TB, seg=[min(T - 1, 3000) for T in Ts_])
        enc12 = tslots[0].timings()["encode_ms"]
        step12 = tslots[0].debug_time_decode_step(rows=5 * TB, t=1 + decode_steps // 2, iters=20)
        return dict(xrt=TS * TB * tsteps * WINDOW_S / twall, streams=TS, batch_per_stream=TB, steps=tsteps,
                    ms_per_step=1e3 * twall / tsteps, windows_per_step=TS * TB,
                    stage_ms_slot0=tm12, encode_ms_one_slot=enc12,
                    encoder_frac_of_mfma_peak=encoder_flops(spec) * TB / (enc12 * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                    decode_step_rows=5 * TB, decode_step_ms=step12,
                    note=f"aggregate of {TS} concurrent slots (own hardware queues) x {TB} windows batched per decode, {decode_steps} tokens each; "
                         "engine-level (PCM resident in HBM), no server / VAD in this leg")
    finally:
        for sl in tslots:
            sl.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="small.en")
    ap.add_argument("--decode-steps", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-decode-steps", type=int, default=32)
    ap.add_argument("--no-stream", action="store_true", help="skip the WebSocket-server stream leg (configs[1] literal form)")
    ap.add_argument("--stream-clients", type=int, default=1, help="WebSocket clients of the stream leg (configs[2]: 4, with --model small)")
    ap.add_argument("--stream-batch", action="store_true", help="stream leg through the per-GPU BatchInferenceWorker")
    ap.add_argument("--batch", type=int, default=1,
                    help="windows per step batched into ONE decode (batch_inference.py's batched mode on one GPU): "
                         "one slot, B items, encoder and every decode step shared by the B x 5 beam rows")
    ap.add_argument("--streams", type=int, default=1,
                    help="concurrent streams on this GPU (BASELINE configs[2] = 4): one engine, one slot + HIP stream + "
                         "host thread per stream, the same step each; value = aggregate over streams")
    ap.add_argument("--config", type=int, default=1, choices=[1, 5],
                    help="BASELINE.json configs index: 1 = the headline single-stream window benchmark (default; with "
                         "--model/--streams/--batch also configs 2-4), 5 = batch_inference.py's batched mode: 64 pre-recorded "
                         "30 s clips, Whisper-large-v3 shapes, one BatchInferenceWorker(max_batch_size=8) per GPU, clips "
                         "sharded in contiguous blocks, ONE all_gather of 2 KiB result records over RCCL")
    ap.add_argument("--clips", type=int, default=64, help="--config 5: number of 30 s clips per step")
    ap.add_argument("--max-batch", type=int, default=8, help="--config 5: max_batch_size of the BatchInferenceWorker (8 = the reference's default, batch_inference.py:100; up to 64 clips x 5 beams = 320 decoder rows since round 5)")
    ap.add_argument("--lanes", type=int, default=4, help="--config 5: lanes of the BatchInferenceWorker (1 = the reference's single worker thread, 2 = the library default; measured 1086 / 1513 / 1618 / 1708 xRT at 1 / 2 / 3 / 4 lanes, profiles/r3k_*, r3d_*)")
    ap.add_argument("--free-run", action="store_true", help="--streams S: every stream runs its steps back to back, started 1/S of a step apart, instead of a barrier per step")
    ap.add_argument("--no-throughput", action="store_true", help="skip the throughput leg of the default run")
    ap.add_argument("--throughput-shape", default="3x48", help="throughput leg: SLOTSxWINDOWS batched per decode (3x48 = 14.9k xRT, profiles/r5b_*; 4x12 = the round-4 shape, 12.4k; 4x24 14.3k; 1x48 11.6k)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 FETCH_SIZE pass that fills roofline.traffic")
    ap.add_argument("--rccl", action="store_true", help="--gpus 1: initialise torch.distributed over RCCL (backend nccl, world size 1) anyway and run the "
                    "barriers, the max-over-ranks all_reduce, the latency gather and (config 5) the record all_gather through it with device tensors "
                    "— the N-rank code path's collectives executed on the one GPU a box has (tests/test_gpu_rccl.py)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--ws-client", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.ws_client:
        raise SystemExit(ws_client_main(args.ws_client))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand: launch the ranks ourselves (the driver does the same thing around this script)
        raise SystemExit(respawn_ranks(args.gpus, sys.argv[1:]))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` or under "
                         f"torch.distributed.run with --nproc-per-node equal to --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # WLX_BENCH_REHEARSAL=1: a rehearsal of the N-rank code path on a box with FEWER GPUs than ranks (the build pool has one-GPU boxes only):
    # the ranks share the visible GPUs and rendezvous over gloo with host tensors. Everything else — rank roles, barriers, the max-over-ranks
    # reduction, the latency gather, rank 0's extra legs while the others wait — runs as it does under RCCL. Its output line says
    # "rehearsal": true and is not a measurement (RCCL refuses two ranks on one device, so the nccl backend cannot be rehearsed this way).
    rehearsal = world > 1 and os.environ.get("WLX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = local % max(1, torch.cuda.device_count())
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank} needs GPU {local} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local)
    dist = None
    global COLL_DEVICE
    if world > 1 or args.rccl:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))     # (--rccl started by hand: no launcher set one)
        if rehearsal:
            COLL_DEVICE = "cpu"
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from whisperlive_amd.engine import HipWhisperEngine, TokenIds
    from whisperlive_amd.specs import get_spec
    from whisperlive_amd.weights import random_weights

    if args.config == 5 and not args.pmc_child:
        if args.model == "small.en":
            args.model = "large-v3"
        return config5(args, rank, world, local, dist, torch)
    spec = get_spec(args.model)
    weights = random_weights(spec, seed=0)
    eng = HipWhisperEngine(spec, weights, device=local)
    if args.pmc_child:
        # the counter pass of measured_traffic(): a few decode steps of the same workload, nothing else
        Bc = max(1, args.batch)
        sl = eng.create_slot(Bc, 5)
        ids_ = token_ids(spec.vocab)
        for b in range(Bc):
            sl.pcm_put(speech_like_pcm(WINDOW_S, seed=1234 + b), b)
        Ts = [sl.logmel_resident(b) for b in range(Bc)]
        sl.encode(Bc, seek=[0] * Bc, seg=[min(T - 1, 3000) for T in Ts])
        sl.generate([[ids_["sot"]]] * Bc, TokenIds(**ids_), beam_size=5, patience=1.0, max_length=1 + 12,
                    suppress_tokens=suppress_list(ids_, True))
        sl.close()
        eng.close()
        return
    S = max(1, args.streams)
    B = max(1, args.batch)
    slots = [eng.create_slot(B, 5) for _ in range(S)]
    slot = slots[0]
    ids = token_ids(spec.vocab)
    eids = TokenIds(**ids)
    pcm = speech_like_pcm(WINDOW_S, seed=1234 + rank)
    gen_kw = dict(beam_size=5, patience=1.0, max_length=1 + args.decode_steps, suppress_tokens=suppress_list(ids, True))

    for i, sl in enumerate(slots):     # inputs resident in HBM before the timed region
        for b in range(B):
            sl.pcm_put(pcm if (i == 0 and b == 0) else speech_like_pcm(WINDOW_S, seed=1234 + rank + 100 * i + b), b)

    def step_on(sl):
        t0 = time.perf_counter()
        Ts = [sl.logmel_resident(b) for b in range(B)]
        sl.encode(B, seek=[0] * B, seg=[min(T - 1, 3000) for T in Ts])
        r = sl.generate([[ids["sot"]]] * B, eids, **gen_kw)[0]
        return time.perf_counter() - t0, r

    if S == 1:
        def step():
            return step_on(slot)
    else:
        # one host thread per stream (the reference runs one thread per client, faster_whisper_backend.py:121); ctypes
        # releases the GIL for the duration of every engine call, the slots' HIP streams overlap on the GPU
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=S)

        def step():
            t0 = time.perf_counter()
            res = list(pool.map(step_on, slots))
            return time.perf_counter() - t0, res[0][1]

    for _ in range(args.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    note("timed region")
    barrier()
    t0 = time.perf_counter()
    lat = []
    stage = dict(logmel_ms=0.0, encode_ms=0.0, generate_ms=0.0)
    last = None
    if S > 1 and args.free_run:
        # the streams run their K steps back to back WITHOUT a barrier per step, started a fraction of a step apart: clients
        # are not phase-locked (the stream leg's clients are separate processes for the same reason), and in lockstep all S
        # encoders — the launches that fill the whole GPU — collide while the S latency-bound decode chains leave it half idle
        tw0 = time.perf_counter(); step(); est = time.perf_counter() - tw0      # (one more untimed lockstep step: the stagger unit)
        barrier()
        t0 = time.perf_counter()

        def run_stream(i):
            time.sleep(i * est / S)
            out_l, r = [], None
            for _ in range(args.steps):
                dt, r = step_on(slots[i])
                out_l.append(dt)
            return out_l, r
        res = list(pool.map(run_stream, range(S)))
        for out_l, _ in res:
            lat.extend(out_l)
        last = res[0][1]
        tm = slot.timings()
        for k in stage:
            stage[k] = tm[k]
    else:
        for _ in range(args.steps):
            dt, last = step()
            lat.append(dt)
            tm = slot.timings()
            for k in stage:
                stage[k] += tm[k] / args.steps
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=COLL_DEVICE)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        lt = torch.tensor(lat, dtype=torch.float64, device=COLL_DEVICE)
        gathered = [torch.zeros_like(lt) for _ in range(world)]
        dist.all_gather(gathered, lt)
        lat = torch.cat(gathered).cpu().tolist()

    out = None
    if rank == 0:
        n_tok = len(last.sequences_ids[0])
        xrt = world * S * B * args.steps * WINDOW_S / wall
        # ---- roofline of the dominant kernel, HIP-event timed inside the engine on the slot stream
        # one slot's decode step as this run executes it: B windows x 5 beams rows per launch (--streams: S such steps run
        # concurrently on S slots; the probe is one slot's step alone)
        prows = 5 * B
        prof = slot.debug_profile_step(rows=prows, t=1 + args.decode_steps // 2, iters=20)
        dom = max(prof, key=lambda k: k["total_us"])
        roof = dict(bound="hbm", kernel=dom["name"], achieved=dom["bytes_per_launch"] / (dom["avg_us"] * 1e-6) / 1e9,
                    peak=HBM_PEAK_GBS, unit="GB/s", traffic=None,
                    launches_per_decode_step=dom["launches"], avg_us=dom["avg_us"],
                    algorithmic_bytes_per_launch=dom["bytes_per_launch"])
        roof["frac"] = roof["achieved"] / roof["peak"]
        if world == 1 and not args.no_pmc:
            note("rocprofv3 FETCH_SIZE pass")
            roof["traffic"], roof["traffic_source"], roof["traffic_detail"] = measured_traffic(
                dom["name"], args.model, child_args=["--batch", str(B)] if B > 1 else ())
            note(f"traffic: {roof['traffic']} ({roof['traffic_source']}) {roof['traffic_detail'].get('calibration')}")
            note("rocprofv3 --kernel-trace --stats pass")
            ravg, rcalls, rsrc = rocprof_kernel_avg(dom["name"], args.model, child_args=["--batch", str(B)] if B > 1 else ())
            roof["rocprof_avg_us"], roof["rocprof_calls"], roof["rocprof_source"] = ravg, rcalls, rsrc
            if ravg:
                roof["frac_rocprof"] = dom["bytes_per_launch"] / (ravg * 1e-6) / 1e9 / HBM_PEAK_GBS
        # the one launch of the step that is bandwidth- rather than latency-sized: the vocabulary projection (80 MB)
        big = max(prof, key=lambda k: k["bytes_per_launch"])
        roof["largest_launch"] = dict(kernel=big["name"], algorithmic_bytes=big["bytes_per_launch"], avg_us=big["avg_us"],
                                      achieved=big["bytes_per_launch"] / (big["avg_us"] * 1e-6) / 1e9,
                                      frac=big["bytes_per_launch"] / (big["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
        step_us = sum(k["total_us"] for k in prof)
        step_graph_ms = slot.debug_time_decode_step(rows=prows, t=1 + args.decode_steps // 2, iters=50)
        sb = decode_step_bytes(spec, prows, 1 + args.decode_steps // 2) + 2 * spec.dec_layers * 2 * spec.n_audio_ctx * spec.d_model * (B - 1)
        # the whole step next to its dominant kernel: algorithmic bytes of ALL launches of a step / the captured graph's replay time / peak
        roof["step_frac"] = sb / (step_graph_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["step_us"] = 1e3 * step_graph_ms
        roof["step_algorithmic_bytes"] = sb
        h2d = []
        for _ in range(7):                  # one window's PCM (1.92 MB) host -> HBM; wlx_pcm_put returns when the copy is complete
            th = time.perf_counter()
            slot.pcm_put(pcm, 0)
            h2d.append(time.perf_counter() - th)
        h2d_ms = 1e3 * float(np.median(h2d))
        out = {
            "metric": "real-time factor (xRT), Whisper-small 30 s window (p50 chunk latency in p50_chunk_latency_ms)",
            "value": xrt, "unit": "xRT (audio s / wall s)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * wall / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 (MFMA operands; f32 accumulate, f32 residual stream, f32 log-mel)", "data": "synthetic",
            "p50_chunk_latency_ms": 1000.0 * float(np.median(lat)),
            # what `value` times and what it leaves out: the PCM is resident in HBM when the timed region starts
            # (wlx_pcm_put outside it); the upload of one window, measured below, is NOT in `value` — the stream leg, which goes
            # through the server with host buffers, includes it
            "timed_region": "per step: wlx_logmel_resident (PCM already in HBM) + wlx_encode + wlx_generate; excludes the host-to-device copy of the window's PCM (h2d_excluded_ms)",
            "h2d_excluded_ms": h2d_ms,
            "config": {"workload": f"configs[{1 if S == 1 else 2}]: Whisper-{args.model}, {S} stream{'s' if S > 1 else ''} per GPU, one 30 s window per stream per step "
                                   f"(480000 samples 16 kHz f32 resident in HBM -> log-mel -> encoder -> beam-5 decode, "
                                   f"{n_tok} generated tokens forced by suppressing EOT), seeded random weights",
                       "streams_per_gpu": S, "batch_per_stream": B, "beam_size": 5, "decode_steps": n_tok, "window_s": WINDOW_S},
            "stage_ms": stage,
            "decode_step": {"rows": prows, "graph_replay_ms": step_graph_ms, "sum_kernel_us": step_us, "algorithmic_bytes": sb,
                            "hbm_frac_of_peak": sb / (step_graph_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "kernels": prof},
            "roofline": roof,
        }
        # the MFMA-regime roofline (SURVEY.md §8d asks for both regimes): encoder + cross-K/V of the timed run
        ef = encoder_flops(spec) * B
        out["roofline_encoder"] = dict(bound="mfma", flops=ef, ms=stage["encode_ms"], achieved=ef / (stage["encode_ms"] * 1e-3) / 1e12,
                                       peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac_of_mfma_peak=ef / (stage["encode_ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                       windows_per_launch=B)
        if S == 1 and B == 1:
            # every window after the first of a stream is CONDITIONED on up to 223 previous tokens
            # (transcriber_faster_whisper.py:1480-1513): the same window with the reference's full prompt [sot_prev] + 223 + [sot]
            prev = np.random.default_rng(5).integers(0, ids["eot"], size=223).tolist()
            cprompt = [ids["timestamp_begin"] - 4] + prev + [ids["sot"]]
            ckw = dict(gen_kw, max_length=len(cprompt) + args.decode_steps)
            ct = []
            for i in range(2 + 5):
                tc0 = time.perf_counter()
                Tc = slot.logmel_resident(0)
                slot.encode(1, seek=[0], seg=[min(Tc - 1, 3000)])
                slot.generate([cprompt], eids, **ckw)
                if i >= 2:
                    ct.append(time.perf_counter() - tc0)
            out["conditioned_window"] = dict(prompt_tokens=len(cprompt), decode_steps=args.decode_steps, ms_per_window=1e3 * float(np.median(ct)),
                                             xrt=WINDOW_S / float(np.median(ct)), generate_ms=slot.timings()["generate_ms"],
                                             note="the headline window with the reference's full 225-token conditioning prompt: prompt prefill + the same 64 steps at positions 225..288")
            # every window of a stream after its first is conditioned (condition_on_previous_text=True, transcriber_faster_whisper.py:1480-1513):
            # the streaming figure next to the headline
            out["value_conditioned"] = out["conditioned_window"]["xrt"]
        if world == 1 and S == 1 and B == 1 and not args.no_stream:
            note("stream leg")
            try:
                out["stream"] = stream_leg(eng, spec, ids, args.decode_steps, stream_pcm, clients=max(1, args.stream_clients),
                                           batch=args.stream_batch, model_name=args.model)
            except Exception as e:  # noqa: BLE001 — the headline line must survive a failure of this leg
                out["stream"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            # self-check first: the SAME bounded decode on the GPU (untimed), then on the CPU port; the common prefix of
            # the two token sequences travels with the line
            T0 = slot.logmel_resident(0)
            slot.encode(B, seek=[0] * B, seg=[min(T0 - 1, 3000)] * B)
            gpu_toks = slot.generate([[ids["sot"]]] * B, eids, **gen_kw)[0].sequences_ids[0]
            w16 = f16_rounded(weights)
            nproc = usable_cpus()
            note(f"cpu baseline on {nproc} threads (3 runs, {n_tok} steps each)")
            base, cpu_toks = cpu_baseline(spec, w16, pcm, ids, n_tok, n_tok, threads=nproc, repeats=3)
            note("cpu baseline on 1 thread")
            one, _ = cpu_baseline(spec, w16, pcm, ids, n_tok, n_tok, threads=1)          # all steps: nothing scaled
            base["single_thread"] = {k: one[k] for k in ("value", "unit", "cores", "sample")}
            base["single_thread"]["note"] = "OMP_NUM_THREADS=1 is the reference server's default (run_server.py:36-39,118-119)"
            note(f"cpu baseline, int8 linears, on {nproc} threads")
            try:
                q8, _ = cpu_baseline(spec, w16, pcm, ids, max(8, args.cpu_decode_steps // 2), n_tok, threads=nproc, int8="fbgemm")
                base["int8"] = {k: q8[k] for k in ("value", "unit", "cores", "kind", "sample")}
            except Exception as e:  # noqa: BLE001 — an extra figure; the fp32 port is the baseline of record
                base["int8"] = {"error": f"{type(e).__name__}: {e}"}
            out["cpu_baseline"] = base
            n = 0
            while n < min(len(gpu_toks), len(cpu_toks)) and gpu_toks[n] == cpu_toks[n]:
                n += 1
            out["parity_prefix"] = n
            out["parity"] = dict(compared_tokens=min(len(gpu_toks), len(cpu_toks)), common_prefix=n,
                                 note="beam-5 decode of the benchmarked window, GPU (fp16 MFMA) vs CPU port (fp32) on the same "
                                      "fp16-rounded seeded weights; random weights give flat distributions, so the first "
                                      "fp16-vs-fp32 near-tie ends the common prefix")
    for sl in slots:
        sl.close()
    if rank == 0 and world == 1 and S == 1 and B == 1 and not args.no_throughput:
        ts_, tb_ = (int(v) for v in args.throughput_shape.lower().split("x"))
        note(f"throughput leg ({ts_} streams x {tb_} windows per decode)")
        try:
            out["throughput"] = throughput_leg(eng, spec, torch, ids, eids, gen_kw, args.decode_steps, TS=ts_, TB=tb_)
        except Exception as e:  # noqa: BLE001 — the headline line must survive a failure of this leg
            out["throughput"] = {"error": f"{type(e).__name__}: {e}"}
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if COLL_DEVICE == "cpu":
            out["rehearsal"] = True          # WLX_BENCH_REHEARSAL: ranks shared a GPU over gloo — not a measurement
        if dist is not None:
            out["collectives"] = {"backend": "gloo (host tensors)" if COLL_DEVICE == "cpu" else "nccl (RCCL, device tensors)", "world_size": world}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
