#!/usr/bin/env python
"""VGPR / SGPR / scratch / LDS of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
usage: kernel_resources.py whisperlive_amd/csrc/decoder.hip [substring]"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-c", src,
                        "-o", os.path.join(d, "o.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
cur = None
for l in r.stderr.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", l)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur and pat in cur["n"]: print(cur)
        n = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"n": re.sub(r"\(.*", "", n).replace("void wlx::", "")}
    elif cur is not None:
        cur[k.split(" ")[0]] = int(v)
if cur and pat in cur["n"]: print(cur)
