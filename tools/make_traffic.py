#!/usr/bin/env python3
"""profiles/<run>/pmc_fetch.txt + pmc_write.txt -> profiles/pmc_traffic.json (HBM bytes per frame of the dominant
kernel).  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE counts 128-byte
requests at 64 bytes for wide coalesced loads (MI355X_MICROARCH.md, HBM section), so the read side is doubled."""
import json
import re
import sys

run, frames, key = sys.argv[1], int(sys.argv[2]), sys.argv[3]
kernel = sys.argv[4] if len(sys.argv) > 4 else "k_polypoint"


def mean(path, kernel, counter):
    for line in open(path):
        if kernel in line and counter in line:
            return float(re.search(r"mean=\s*([0-9.]+)", line).group(1))
    raise SystemExit(f"{counter} of {kernel} not found in {path}")


# several kernels of one bracket: "k_a+k_b+k_c" (their dispatches are summed)
fetch_kib = sum(mean(f"{run}/pmc_fetch.txt", k, "FETCH_SIZE") for k in kernel.split("+"))
write_kib = sum(mean(f"{run}/pmc_write.txt", k, "WRITE_SIZE") for k in kernel.split("+"))
out = "profiles/pmc_traffic.json"
kept = run.replace("gpurun_out/", "profiles/", 1)  # where the session copies the run (gpurun_out/ is scratch)
try:
    d = json.load(open(out))
except Exception:  # noqa: BLE001
    d = {}
d[key] = {"bytes_per_frame": (2 * fetch_kib + write_kib) * 1024 / frames, "fetch_kib_raw_per_dispatch": fetch_kib,
          "write_kib_per_dispatch": write_kib, "frames_per_dispatch": frames, "read_correction": 2.0, "source": kept, "profile": kept, "kernel": kernel}
json.dump(d, open(out, "w"), indent=1)
print(d[key])
