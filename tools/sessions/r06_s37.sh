#!/bin/bash
# round-6 session 37: kernel traces of 16 x 4K on scene8 depth -- polylines_sharp blur off / on, polylines_soft blur on (where does the distance to
# stepped depth go after the second tier and the lean pass?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06_s37
bash tools/gpu_trace.sh r06_s37/sharp_scene8_blur0 tools/quick_bench.py --n 16 --fill polylines_sharp --kind scene8 --iters 4 | cut -c1-150
bash tools/gpu_trace.sh r06_s37/sharp_scene8_blur1 tools/quick_bench.py --n 16 --fill polylines_sharp --kind scene8 --blur 1 --iters 4 | cut -c1-150
bash tools/gpu_trace.sh r06_s37/soft_scene8_blur1 tools/quick_bench.py --n 16 --fill polylines_soft --kind scene8 --blur 1 --iters 4 | cut -c1-150
bash tools/gpu_trace.sh r06_s37/sharp_stepped_blur0 tools/quick_bench.py --n 16 --fill polylines_sharp --kind stepped --iters 4 | cut -c1-150
