#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py mrgcn_amd/csrc/rgcn_fused.hip [name-substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-x", "hip", src, "-o",
                      "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
if "error:" in out:
    print(out)
    sys.exit(1)
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.rsplit(":", 1)
        rows[cur][k.strip()] = v.strip()
demangle = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>5} {'scratch':>7} {'occ':>3} {'LDS':>6}  kernel")
for name, pretty in zip(rows, demangle):
    if flt and flt not in pretty:
        continue
    r = rows[name]
    pretty = re.sub(r"\(anonymous namespace\)::", "", pretty).split("(")[0].replace("void ", "")
    print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('VGPRs Spill','?'):>5} {r.get('ScratchSize [bytes/lane]','?'):>7} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>3} {r.get('LDS Size [bytes/block]','?'):>6}  {pretty}")
