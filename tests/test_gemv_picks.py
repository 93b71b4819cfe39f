"""Host-only (-m "not gpu"): which `dec_gemv2_kernel` instantiation the launcher picks for every decode projection of the Whisper
family — the rules of `csrc/decoder.hip gemv2_cfg` as round 6 left them (DESIGN.md §7.3 logs G5, G7-G10), pinned so that a change of
the search loop shows up here before it shows up as a slower step or, as it did once this round, as a launch past its bound.

`scripts/gemv_pick_probe.cpp` is compiled for the HOST (hipcc, no device code) and linked against the production `libwlx.so`,
whose `wlx::dec_gemv_kernel_name` prints the template arguments <CH, LNV, IN, OUT, NTB, MT, XS> the launcher would use:
CH = k-tiles per wave (so K / 32 / CH waves stream the weights), IN 0 = LayerNorm-fronted, 1 = fp16 rows in, 2 = split combine.
No kernel is launched and no GPU is needed."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

# what one stream's step (<= 8 rows, Mtot == 0) runs, per projection
EXPECTED = {
    # fp16 rows in (attention output projection; K-split MLP output projection; the one that also adds the slabs): few waves, long K slices (G5)
    "small o-proj M5": "dec_gemv2_kernel<12, 1, 1, 3, 1, 1, 0>",
    "small fc2 slab M5": "dec_gemv2_kernel<12, 1, 1, 5, 1, 1, 0>",
    "small o-proj slabs M5": "dec_gemv2_kernel<12, 1, 1, 3, 1, 1, 1>",
    "medium o-proj M5": "dec_gemv2_kernel<8, 1, 1, 3, 1, 1, 0>",
    "medium fc2 slab M5": "dec_gemv2_kernel<8, 1, 1, 5, 1, 1, 0>",
    "large o-proj M5": "dec_gemv2_kernel<10, 1, 1, 3, 1, 1, 0>",
    "large fc2 slab M5": "dec_gemv2_kernel<10, 1, 1, 5, 1, 1, 0>",
    "base o-proj M5": "dec_gemv2_kernel<8, 1, 1, 3, 1, 1, 0>",
    "tiny o-proj M5": "dec_gemv2_kernel<12, 1, 1, 3, 1, 1, 0>",
    # the split combine keeps four waves of six / eight of five (G5: wider lost)
    "small xattn M5": "dec_gemv2_kernel<6, 1, 2, 3, 1, 1, 0>",
    "large xattn M5": "dec_gemv2_kernel<5, 1, 2, 3, 1, 1, 0>",
    # LayerNorm-fronted on PLAIN rows: four waves (G7, G8, G10)
    "small mlp-up M5": "dec_gemv2_kernel<6, 3, 0, 1, 1, 1, 0>",
    "medium mlp-up M5": "dec_gemv2_kernel<8, 4, 0, 1, 1, 1, 0>",
    "large mlp-up M5": "dec_gemv2_kernel<10, 5, 0, 1, 2, 1, 0>",
    "large q-proj M5": "dec_gemv2_kernel<10, 5, 0, 0, 1, 1, 0>",
    # a layer's first projection (rows + slabs / embedding rows): one row per wave for K = 768, four waves with both rows of a wave
    # requested together for K = 1024 / 1280 (G9)
    "small qkv slabs M5": "dec_gemv2_kernel<4, 3, 0, 4, 1, 1, 1>",
    "small qkv embed M5": "dec_gemv2_kernel<4, 3, 0, 4, 1, 1, 2>",
    "medium qkv slabs M5": "dec_gemv2_kernel<8, 4, 0, 4, 1, 1, 1>",
    "large qkv slabs M5": "dec_gemv2_kernel<10, 5, 0, 4, 1, 1, 1>",
    # batched rows (row tiles, Mtot > 0) and 9..16 rows of the LayerNorm-fronted launches keep the narrow slices
    "small o-proj M60": "dec_gemv2_kernel<6, 1, 1, 3, 1, 1, 0>",
    "small fc2 slab M60": "dec_gemv2_kernel<6, 1, 1, 5, 1, 1, 0>",
    "small mlp-up M60": "dec_gemv2_kernel<6, 3, 0, 1, 4, 1, 0>",
    "large mlp-up M16": "dec_gemv2_kernel<5, 5, 0, 1, 2, 1, 0>",
    "large qkv slabs M16": "dec_gemv2_kernel<5, 5, 0, 4, 1, 1, 1>",
    # 16 rows of an UNSPLIT K = 4096 projection: eight waves of eight k-tiles twice over (sixteen waves would pass the 512-thread bound
    # of the wide instantiations: 'unspecified launch failure', profiles/r6ap_*)
    "medium fc2 resid M16": "dec_gemv2_kernel<8, 1, 1, 3, 1, 1, 0>",
}


@pytest.fixture(scope="module")
def picks(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not found")
    from whisperlive_amd import _lib
    lib = _lib.build()                                       # (fresh in-tree library: returned as it is)
    exe = tmp_path_factory.mktemp("probe") / "gemv_pick_probe"
    cmd = [hipcc, "-std=c++17", "-O1", str(ROOT / "scripts" / "gemv_pick_probe.cpp"), "-o", str(exe),
           f"-L{lib.parent}", f"-l:{lib.name}", f"-Wl,-rpath,{lib.parent}"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-2000:]
    got = {}
    for line in out.stdout.splitlines():
        what, rest = line[:24].strip(), line[24:]
        got[what] = (rest.split("  ")[-1].strip(), "lean=1" in rest, rest)
    return got


def test_every_probed_projection_runs_on_the_lean_kernel(picks):
    assert len(picks) >= len(EXPECTED)
    for what, (name, lean, rest) in picks.items():
        assert lean and name.startswith("dec_gemv2_kernel<"), (what, rest)


@pytest.mark.parametrize("what", sorted(EXPECTED))
def test_pick(picks, what):
    assert what in picks, sorted(picks)
    assert picks[what][0] == EXPECTED[what], (what, picks[what][2])


def test_k_split_of_the_mlp_output_projection(picks):
    """two K slices for one stream's rows and for row tiles; none for the 16-row unsplit case the probe lists"""
    assert "slab_split=2" in picks["small fc2 slab M5"][2] and "slab_split=2" in picks["large fc2 slab M5"][2]
    assert "slab_split=0" in picks["medium fc2 resid M16"][2]
