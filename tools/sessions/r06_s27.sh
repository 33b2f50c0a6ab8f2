#!/bin/bash
# round-6 session 27: where the time of the tie paths goes -- kernel traces of 16 x 4K polylines_sharp / polylines_soft on clipped (saturated)
# depth and 4 x 4K polylines_soft on 8-bit noise, blur off; D64 soft on stepped depth (the SW kernel)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06_s27
bash tools/gpu_trace.sh r06_s27/sharp_clipped tools/quick_bench.py --n 16 --fill polylines_sharp --kind clipped --iters 4 | cut -c1-160
bash tools/gpu_trace.sh r06_s27/soft_clipped tools/quick_bench.py --n 16 --fill polylines_soft --kind clipped --iters 4 | cut -c1-160
bash tools/gpu_trace.sh r06_s27/soft_random8 tools/quick_bench.py --n 4 --fill polylines_soft --kind random8 --iters 3 | cut -c1-160
bash tools/gpu_trace.sh r06_s27/soft_d64 tools/quick_bench.py --n 16 --fill polylines_soft --kind stepped --dialect D64 --iters 4 | cut -c1-160
bash tools/gpu_trace.sh r06_s27/sharp_d64 tools/quick_bench.py --n 16 --fill polylines_sharp --kind stepped --dialect D64 --iters 4 | cut -c1-160
