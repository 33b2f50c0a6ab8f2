#!/usr/bin/env python3
"""Register budget of every kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only): VGPRs, SGPRs, spills, LDS.
   python tools/kernel_resources.py /tmp/x.s [--spills-only]   (development aid: scratch spills are memory traffic)"""
import re
import subprocess
import sys

cur = {}
rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "name" and not v.endswith(".kd") and "cur_name" not in cur:
        cur["cur_name"] = v
    elif k != "name":
        cur[k] = v
    if k == "vgpr_spill_count":   # (last field of a kernel's block in the metadata order)
        rows.append(cur); cur = {}
names = [r.get("cur_name", "?") for r in rows]
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()
except Exception:  # noqa: BLE001
    dem = names
for r, d in zip(rows, dem):
    if "--spills-only" in sys.argv and r.get("vgpr_spill_count") == "0":
        continue
    print(f"{d.replace('cs::', '')[:90]:90s} vgpr {r.get('vgpr_count'):>4s} spill {r.get('vgpr_spill_count'):>3s} | sgpr {r.get('sgpr_count'):>4s} spill {r.get('sgpr_spill_count'):>3s} | scratch {r.get('private_segment_fixed_size', '?'):>5s} B")
