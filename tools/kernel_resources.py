#!/usr/bin/env python3
"""Registers, scratch bytes and waves per SIMD of every kernel in rt_hip.hip, from hipcc's own resource remarks
(no GPU needed; about two minutes).  A shading kernel of one material type that shows scratch bytes or > 200 VGPRs has
stopped folding its material switch: some access keeps the material record in memory (see matte_albedo)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import HIP_FLAGS  # noqa: E402

src = os.path.join(ROOT, "raytracing_amd", "csrc", "rt_hip.hip")
with tempfile.TemporaryDirectory() as tmp:
    r = subprocess.run(["hipcc"] + HIP_FLAGS + sys.argv[1:] + ["-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "rt.so"), src],
                       cwd=tmp, stderr=subprocess.PIPE, text=True)
if r.returncode:
    sys.exit(r.stderr)
print(f"{'kernel':44s} {'VGPR':>5s} {'spill':>5s} {'scratch':>7s} {'LDS':>6s} {'waves':>5s}")
for block in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    mangled = block.split()[0]
    name = subprocess.run(["c++filt", mangled], stdout=subprocess.PIPE, text=True).stdout.split("(")[0].replace("void ", "")

    def field(key):
        m = re.search(re.escape(key) + r": (\d+)", block)
        return m.group(1) if m else "?"

    print(f"{name:44s} {field(' VGPRs'):>5s} {field('VGPRs Spill'):>5s} {field('ScratchSize [bytes/lane]'):>7s} "
          f"{field('LDS Size [bytes/block]'):>6s} {field('Occupancy [waves/SIMD]'):>5s}")
