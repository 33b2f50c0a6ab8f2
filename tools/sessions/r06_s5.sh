#!/bin/bash
# round-6 session 5: k_gpuwarp_q after the prologue diet (per-frame constants from k_gpuwarp_flags by scalar loads, LDS carve-up per
# phase): tests, fuzz, A/B against k_gpuwarp, phase counters (dev build)
bash tools/sessions/r06_s3.sh r06_s5
bash tools/sessions/r06_s4.sh r06_s5
