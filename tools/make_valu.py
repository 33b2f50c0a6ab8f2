#!/usr/bin/env python3
"""profiles/<run>/pmc_sq1.txt + pmc_sq2.txt + kernel_trace.txt -> profiles/pmc_valu.json: the VECTOR-ISSUE side of a kernel's
roofline -- what bench.py reports as `roofline.valu` next to the HBM figures.

    python tools/make_valu.py <run dir> <key> <kernel name substring> [<source .hip> <mangled-name regex>]

Per kernel (means per dispatch of the committed PMC passes; the instruction and cycle counters do not depend on the box):
  valu / salu / lds per wave   SQ_INSTS_* / SQ_WAVES
  simd_busy                    SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs  over  GRBM_GUI_ACTIVE / 8 XCDs: the share of the kernel's
                               duration during which a SIMD's vector pipe is occupied, in the counter's quad-cycle granularity (it
                               over-states the 2-cycle classes, so it is an upper bound of the busy share)
  wait_any / wait_inst         SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES
  issue_floor_ms               (with a source file) the kernel's static mix of vector instructions, priced with the cycle classes
                               tools/ubench measured on this chip at full occupancy (profiles/r02_ubench_oprate.txt: 2.7 cycles per
                               wave-instruction for plain float32 / integer add, mul, fma, mov, and, or; 4.4 for the other
                               single-rate operations, conversions, compares, shifts, DPP, lane reads; 4.7 for float64 and packed
                               float32; 8.5 for rcp / sqrt / rsq / exp / log), times the DYNAMIC instruction count per wave, times
                               the waves, over 1024 SIMDs at the clock the counters imply (GRBM cycles per XCD / kernel time):
                               the time the kernel would take if nothing but vector issue bounded it
  binding_roof                 "valu" when that floor exceeds the memory floor (algorithmic bytes at the 6.29 TB/s copy ceiling of
                               MI355X_MICROARCH.md), else "hbm"
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = (
    (8.5, re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)")),
    (4.7, re.compile(r"^v_(pk_|[a-z0-9_]*f64|cvt_f64|cvt_f32_f64|cvt_[iu]32_f64)")),
    (2.7, re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad)_f32|^v_(add|sub|subrev)_(u32|i32|co_u32)|^v_(mov_b32_e32|and_b32|or_b32|xor_b32|not_b32)")),
)
DEFAULT_CYCLES = 4.4


def counters(path, kernel):
    out = {}
    for line in open(path):
        if kernel in line:
            m = re.search(r"\b([A-Z][A-Z0-9_]+)\s+n=\s*\d+\s+mean=\s*([0-9.]+)", line)
            if m:
                out[m.group(1)] = out.get(m.group(1), 0.0) + float(m.group(2))
    return out


def kernel_us(path, kernel):
    for line in open(path):
        if kernel in line:
            f = line.split()
            return float(f[-2])
    raise SystemExit(f"{kernel} not found in {path}")


def census(source, name_re):
    """Static class mix of the vector instructions of the kernel(s) whose mangled name matches: mean cycles per instruction."""
    asm = f"/tmp/make_valu_{os.path.basename(source)}.s"
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16"
    if "cs_polypoint" in source:
        flags += " -fno-slp-vectorize"
    subprocess.check_call(f"/opt/rocm/bin/hipcc {flags} -S --cuda-device-only {source} -o {asm}", shell=True, stderr=subprocess.DEVNULL)
    pat, inside, hist = re.compile(name_re), False, {}
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside = bool(pat.search(m.group(1)))
            continue
        if inside and "s_endpgm" in line:
            inside = False
        if inside:
            t = line.strip().split()
            if t and t[0].startswith("v_"):
                hist[t[0]] = hist.get(t[0], 0) + 1
    total = sum(hist.values())
    if not total:
        raise SystemExit(f"no kernel matching {name_re} in {source}")
    cyc, by_class = 0.0, {}
    for op, cnt in hist.items():
        c = next((c for c, rx in CLASSES if rx.search(op)), DEFAULT_CYCLES)
        cyc += c * cnt
        by_class[str(c)] = by_class.get(str(c), 0) + cnt
    return cyc / total, total, {k: round(v / total, 3) for k, v in sorted(by_class.items())}


def main():
    run, key, kernel = sys.argv[1], sys.argv[2], sys.argv[3]
    c = counters(os.path.join(run, "pmc_sq1.txt"), kernel)
    c.update(counters(os.path.join(run, "pmc_sq2.txt"), kernel))
    us = kernel_us(os.path.join(run, "kernel_trace.txt"), kernel)
    waves = c["SQ_WAVES"]
    xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    entry = {
        "kernel": kernel, "profile": run.replace("gpurun_out/", "profiles/", 1), "kernel_us_profile": us, "waves": waves,
        "valu_per_wave": c["SQ_INSTS_VALU"] / waves, "salu_per_wave": c.get("SQ_INSTS_SALU", 0.0) / waves,
        "lds_per_wave": c.get("SQ_INSTS_LDS", 0.0) / waves,
        "simd_busy": (c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0) / xcd_cycles,
        "wait_any": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], "wait_inst": c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
        "clock_ghz_implied": xcd_cycles / us / 1e3,
    }
    if len(sys.argv) > 5:
        mean_cyc, static_n, mix = census(os.path.join(ROOT, sys.argv[4]), sys.argv[5])
        floor_cycles = entry["valu_per_wave"] * mean_cyc * waves / 1024.0
        entry.update({"static_valu": static_n, "static_mix_by_cycles": mix, "mean_cycles_per_valu": mean_cyc,
                      "issue_floor_us": floor_cycles / (xcd_cycles / us), "frac_of_issue_floor": floor_cycles / xcd_cycles})
    try:   # frames of one dispatch of that session (tools/make_traffic.py wrote them under the same key)
        entry["frames_per_dispatch"] = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[key]["frames_per_dispatch"]
    except Exception:  # noqa: BLE001
        pass
    out = os.path.join(ROOT, "profiles", "pmc_valu.json")
    try:
        d = json.load(open(out))
    except Exception:  # noqa: BLE001
        d = {}
    d[key] = entry
    json.dump(d, open(out, "w"), indent=1)
    print(key, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in entry.items()})


if __name__ == "__main__":
    main()
