"""How much of a flagged row is order-dependent?  (development aid, round 5)  k_polypoint records, per row and eye, which of its tiles
raised a hazard (tile hints in the flagged-row block of the workspace); this reads them back after one call and prints the
distribution: flagged row-eyes, tiles per flagged row-eye, the span from the first to the last flagged tile."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd import engine
n, h, w = 8, 2160, 3840
for kind in sys.argv[1:] or ["clipped", "stepped", "blobs", "random8"]:
    img = torch.from_numpy(synth.image_f32(n, h, w, seed=5)).cuda()
    depth = torch.from_numpy(synth.depth_batch(kind, n, h, w, channels=3)).cuda()
    p = engine.make_params(n, h, w, h, w, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 2.0, 6, 12)
    plan = engine.Plan(p, torch.device("cuda"))
    plan.run(img, depth); torch.cuda.synchronize()
    st = plan.stats()
    al = lambda x: (x + 255) & ~255
    rows = n * h
    off = al(st.shape[0] * st.shape[1] * 4) + 2 * al(rows) + 512
    hints = plan.ws[off:off + rows * 8].cpu().numpy().view(np.uint32).reshape(rows, 2)
    fl = hints[hints != 0]
    pop = np.array([bin(int(v)).count("1") for v in fl]) if len(fl) else np.zeros(0)
    span = np.array([int(v).bit_length() - (int(v) & -int(v)).bit_length() + 1 for v in fl]) if len(fl) else np.zeros(0)
    print(f"{kind}: {len(fl)} of {2 * rows} row-eyes flagged ({100.0 * len(fl) / (2 * rows):.1f} %); tiles per flagged row-eye: mean {pop.mean() if len(pop) else 0:.2f} "
          f"hist {np.bincount(pop.astype(int), minlength=6)[1:6].tolist() if len(pop) else []}; span first..last: mean {span.mean() if len(span) else 0:.2f} "
          f"hist {np.bincount(span.astype(int), minlength=6)[1:6].tolist() if len(span) else []}; flagged rows per stats word 11: {int(st[:, 11].sum())}")
