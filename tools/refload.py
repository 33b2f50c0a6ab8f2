"""Load the upstream reference (read-only, /root/reference) for golden-vector generation.

This module is build-container-only tooling: the reference never travels to the GPU box, and nothing
under tests/, bench.py or the product package imports it at run time.  It only exists so that
tools/make_goldens.py can run the reference's own functions on seeded inputs and commit the
inputs/outputs as data fixtures under tests/golden/.

The reference needs two modules this image lacks (cv2: imported, never used on the path;
comfy.utils.ProgressBar: a progress counter).  Both get empty stand-ins in sys.modules *for the import
only*; no reference arithmetic is replaced.
"""
import importlib
import importlib.util
import sys
import types
import warnings

REF = "/root/reference"


def load_sig():
    """stereoimage_generation.py alone (numpy, PIL, scipy, torch only)."""
    if "ref_sig" in sys.modules:
        return sys.modules["ref_sig"]
    spec = importlib.util.spec_from_file_location("ref_sig", REF + "/stereoimage_generation.py")
    sig = importlib.util.module_from_spec(spec)
    sys.modules["ref_sig"] = sig
    spec.loader.exec_module(sig)
    return sig


def load_node():
    """GenerateStereo.py as a submodule of a synthetic package (its __init__.py is never executed)."""
    if "refpkg.GenerateStereo" in sys.modules:
        return sys.modules["refpkg.GenerateStereo"]
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    comfy, cu = types.ModuleType("comfy"), types.ModuleType("comfy.utils")

    class ProgressBar:
        def __init__(self, total):
            self.total, self.n = total, 0

        def update(self, k):
            self.n += k

    cu.ProgressBar = ProgressBar
    comfy.utils = cu
    sys.modules.setdefault("comfy", comfy)
    sys.modules.setdefault("comfy.utils", cu)
    pkg = types.ModuleType("refpkg")
    pkg.__path__ = [REF]
    sys.modules["refpkg"] = pkg
    return importlib.import_module("refpkg.GenerateStereo")


def quiet():
    """The D32 dialect (no numba, NumPy 2) emits an expected uint8-overflow warning per pixel sum."""
    warnings.filterwarnings("ignore", category=RuntimeWarning)
