#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s19
timeout 900 python -m pytest tests/test_gpu_dialect.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/s19/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/s19/tests.log
timeout 300 python - <<'PY' 2>&1 | tail -5
import time, torch, numpy as np, sys
sys.path.insert(0, "tools")
import synth
from comfystereo_amd import engine
n, h, w = 2, 2160, 3840
img = torch.from_numpy(synth.image_f32(1, h, w, seed=1)).expand(n, -1, -1, -1).contiguous().cuda()
dep = torch.from_numpy(synth.depth_batch("stepped", n, h, w, channels=3)).cuda()
for d in ("D32", "f64-disparity", "D64"):
    engine.DIALECT = d
    p = engine.make_params(n, h, w, h, w, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 2.0, 6, 4)
    plan = engine.Plan(p, torch.device("cuda"))
    plan.run(img, dep); torch.cuda.synchronize()
    t0 = time.perf_counter(); plan.run(img, dep); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"polylines_soft 4K SBS blur off, dialect {d}: {n / dt:.1f} frames/s")
PY
