"""The sharp fixture case that fails only in test order (development aid): same call sequence as
test_f64_disparity_chain_matches_the_reference_fixture, details of the first mismatch, repeats, and the row kernel alone."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from comfystereo_amd import engine, _native
from test_gpu_dialect import FILLS
z = np.load(os.path.join(ROOT, "tests", "golden", "dialect_f64.npz"))
cases = json.loads(str(z["meta"]))["cases"]

def run(c, fill):
    return engine.apply_stereo_divergence(torch.from_numpy(z[f"{c['id']}/img"]).cuda(), torch.from_numpy(z[f"{c['id']}/depth"]).cuda(),
                                          c["divergence"], c["separation"], c["exponent"], fill, c["convergence"], dialect="f64-disparity").cpu().numpy()
for rep in range(12):
    for c in cases:
        for fill in FILLS:
            got, want = run(c, fill), z[f"{c['id']}/{fill}"]
            bad = np.argwhere(got != want)
            if len(bad):
                print("rep", rep, "case", c["id"], fill, z[f"{c['id']}/img"].shape, "mismatches", len(bad),
                      [(int(a), int(b), int(ch), int(got[a, b, ch]), int(want[a, b, ch])) for a, b, ch in bad[:9]])
                for k in range(3):
                    g2 = run(c, fill)
                    print("   again:", int((g2 != want).sum()), "differs from first:", int((g2 != got).sum()))
                _native.debug_set("no_tile", 1)
                g3 = run(c, fill)
                _native.debug_set("no_tile", 0)
                print("   row kernel alone:", int((g3 != want).sum()))
print("done")
