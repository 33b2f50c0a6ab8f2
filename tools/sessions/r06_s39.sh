#!/bin/bash
# round-6 session 39: kernel traces of the D64 paths on the final binary (16 x 4K): polylines_soft / polylines_sharp under full D64 on stepped depth,
# polylines_soft on scene8 with the blur (second tier + stretch replay in the dialect row kernel), the metric's workload under D64 (64 frames, blur on)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06_s39
bash tools/gpu_trace.sh r06_s39/soft_d64_stepped tools/quick_bench.py --n 16 --fill polylines_soft --kind stepped --dialect D64 --iters 4 | cut -c1-150
bash tools/gpu_trace.sh r06_s39/sharp_d64_stepped tools/quick_bench.py --n 16 --fill polylines_sharp --kind stepped --dialect D64 --iters 4 | cut -c1-150
bash tools/gpu_trace.sh r06_s39/soft_d64_scene8_blur1 tools/quick_bench.py --n 16 --fill polylines_soft --kind scene8 --blur 1 --dialect D64 --iters 4 | cut -c1-150
bash tools/gpu_trace.sh r06_s39/soft_d64_metric tools/quick_bench.py --n 64 --fill polylines_soft --kind stepped --blur 1 --dialect D64 --iters 3 | cut -c1-150
