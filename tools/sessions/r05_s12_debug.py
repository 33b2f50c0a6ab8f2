"""Uninitialised device memory? (development aid)  Session 9's sharp fixture failure appeared once, in the first GPU process of a fresh box.
Pre-fill the caching allocator's blocks with a byte pattern, then run the fixture sequence: results must not depend on the pattern."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from comfystereo_amd import engine, _native
from test_gpu_dialect import FILLS
z = np.load(os.path.join(ROOT, "tests", "golden", "dialect_f64.npz"))
cases = json.loads(str(z["meta"]))["cases"]
pattern = int(sys.argv[1])

def poison():
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    big = torch.empty(3 << 30, dtype=torch.uint8, device="cuda"); big.fill_(pattern)
    small = [torch.empty(s, dtype=torch.uint8, device="cuda").fill_(pattern) for s in (1 << 20, 1 << 20, 1 << 20, 1 << 20, 8 << 20, 64 << 20)]
    torch.cuda.synchronize(); del big, small

def run(c, fill, dialect):
    return engine.apply_stereo_divergence(torch.from_numpy(z[f"{c['id']}/img"]).cuda(), torch.from_numpy(z[f"{c['id']}/depth"]).cuda(),
                                          c["divergence"], c["separation"], c["exponent"], fill, c["convergence"], dialect=dialect).cpu().numpy()
nbad = 0
for rep in range(2):
    for c in cases:
        for fill in FILLS:
            for dialect in ("f64-disparity", "D32"):
                poison()
                got = run(c, fill, dialect)
                if dialect == "f64-disparity":
                    want = z[f"{c['id']}/{fill}"]
                    bad = np.argwhere(got != want)
                    if len(bad):
                        nbad += 1
                        print("pattern", pattern, "rep", rep, "case", c["id"], fill, "mismatches", len(bad),
                              [(int(a), int(b), int(ch), int(got[a, b, ch]), int(want[a, b, ch])) for a, b, ch in bad[:9]])
                        _native.debug_set("no_tile", 1); poison(); g3 = run(c, fill, dialect); _native.debug_set("no_tile", 0)
                        print("   row kernel alone:", int((g3 != want).sum()))
print("pattern", pattern, "done, failing calls:", nbad)
