"""oracle/ref_import.py -- TEST INFRASTRUCTURE ONLY.

Imports the *reference's own* Python implementation of the hot path
(/root/reference/coremasic/mywork/MASIC.py + compressai/{entropy_models,layers,ops,models/utils})
on CPU in THIS container, so that (i) oracle/hsic_oracle.py can be pinned against it and
(ii) tests/golden/make_goldens.py can emit golden vectors.  Nothing here travels to the
GPU box as a dependency: /root/reference does not exist there, and no test, smoke() or
bench leg imports this module at run time.

Recipe:
  1. oracle/Makefile builds compressai._CXX / compressai.ans from the reference sources
     into oracle/_ref/ (import-time deps of entropy_models.py:8-9, MASIC.py:20).
  2. a synthetic `compressai` package whose __path__ points at the reference package dir
     plus oracle/_ref, bypassing the reference's eager __init__ (which imports cv2).
  3. empty stubs for off-path third-party modules absent here (cv2, imageio, range_coder,
     torchvision, PIL is present).
  4. `kornia` stub exposing warp_perspective = OUR restatement of kornia 0.5.0
     (oracle/hsic_oracle.py:warp_perspective) -- kornia is not in the container and not under
     /root/reference, so the warp convention is "parity unpinned" (DESIGN.md).
"""
import importlib
import os
import sys
import types

REF = os.environ.get("MASIC_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
_REFBIN = os.path.join(_HERE, "_ref")


def available():
    return os.path.isdir(os.path.join(REF, "coremasic", "mywork")) and os.path.isdir(_REFBIN)


def load():
    """Returns the reference `MASIC` module (HSIC, Independent_EN, mask, ...)."""
    if "MASIC_reference" in sys.modules:
        return sys.modules["MASIC_reference"]
    if not available():
        raise RuntimeError("reference not available (need %s and oracle/_ref; run `make -C oracle ref`)" % REF)
    from oracle import hsic_oracle

    saved = {k: sys.modules.get(k) for k in list(sys.modules)
             if k == "compressai" or k.startswith("compressai.") or k in ("MASIC", "kornia")}
    for k in saved:
        del sys.modules[k]

    pkg = types.ModuleType("compressai")
    pkg.__path__ = [os.path.join(REF, "compressai"), _REFBIN]
    pkg.available_entropy_coders = lambda: ["ans"]
    pkg.get_entropy_coder = lambda: "ans"
    pkg.set_entropy_coder = lambda name: None
    sys.modules["compressai"] = pkg

    ds = types.ModuleType("compressai.datasets")
    ds.ImageFolder = type("ImageFolder", (), {})
    sys.modules["compressai.datasets"] = ds

    for name in ("cv2", "imageio", "range_coder", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                m = types.ModuleType(name)
                sys.modules[name] = m
    rc = sys.modules["range_coder"]
    for n in ("RangeEncoder", "RangeDecoder", "prob_to_cum_freq"):
        if not hasattr(rc, n):
            setattr(rc, n, None)
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "transforms"):
        tv.transforms = sys.modules["torchvision.transforms"]

    kornia = types.ModuleType("kornia")
    kornia.warp_perspective = lambda src, M, dsize, **kw: hsic_oracle.warp_perspective(src, M, dsize)
    sys.modules["kornia"] = kornia

    sys.path.insert(0, os.path.join(REF, "coremasic", "mywork"))
    try:
        mod = importlib.import_module("MASIC")
    finally:
        sys.path.pop(0)
    # The reference resolves `from compressai import ...` lazily (entropy_models.py:20,46), so
    # its package stays in sys.modules: a process that called load() must not import the
    # product's own `compressai`/`MASIC` afterwards (tests that need both use a subprocess).
    sys.modules["MASIC_reference"] = mod
    del saved
    return mod
