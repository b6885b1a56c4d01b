"""Bind the HIP drop-in under the reference's UNCHANGED entry points.

Every entry point of the reference puts its own checkout FIRST on `sys.path` before it imports the model
(`training/train.py:41-46`, `training/evaluate.py:32-36`, `api/inference.py:25-29`, `api/main.py:35-39`:
`sys.path.insert(0, str(PROJECT_ROOT))` then `from models.vqa_model import ...`; `reproduce_issue.py:5` is run from the
root, which is `sys.path[0]`), so a `PYTHONPATH=.../dropin` in front of it never wins -- the caller would silently train
the reference's own ATen model.  What does win is `sys.modules`: an import statement consults it before any path entry.

`bind()` therefore
  1. imports the drop-in's `models` package and `models.vqa_model` under exactly those names (so they sit in
     `sys.modules['models']` / `sys.modules['models.vqa_model']`),
  2. takes `.../dropin` off `sys.path` again, so `utils.config`, `utils.metrics`, `data.dataset`, ... stay the
     reference's own (`training/train.py:47-49`; the drop-in's `utils/` and `data/` hold only the device-side pieces and
     must not shadow them),
  3. appends the reference's `models/` directory to the bound package's `__path__`, so `models.cnn_backbone` & co. (not
     used by any entry point, but importable in the reference) still resolve to the reference's files.
Nothing here touches the GPU (no HIP call, no `torch.cuda.is_available()`): `run()` may be followed by anything.
"""
from __future__ import annotations

import importlib
import os
import runpy
import sys
from typing import List, Optional

_BOUND_NAMES = ("models", "models.vqa_model")


def dropin_dir() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def is_bound() -> bool:
    m = sys.modules.get("models.vqa_model")
    return m is not None and os.path.abspath(getattr(m, "__file__", "")).startswith(dropin_dir() + os.sep)


def bind(project_root: Optional[str] = None, dtype: Optional[str] = None):
    """Make `from models.vqa_model import VQAModel, create_vqa_model, load_vqa_model` resolve to the HIP drop-in for the rest
    of this process, whatever the caller later puts on `sys.path`.  Returns the bound `models.vqa_model` module.

    project_root: the reference checkout (its `models/` is appended to the package path for the other submodules).
    dtype: "bf16" | "fp32" -> VQA_HIP_DTYPE, the drop-in's compute dtype when the constructor is not told."""
    if dtype is not None:
        if dtype not in ("bf16", "fp32"):
            raise ValueError(f"dtype must be 'bf16' or 'fp32', got {dtype!r}")
        os.environ["VQA_HIP_DTYPE"] = dtype
    if not is_bound():
        for name in _BOUND_NAMES:
            old = sys.modules.get(name)
            if old is not None:
                raise RuntimeError(f"bind(): {name!r} is already imported from {getattr(old, '__file__', '?')}; bind the drop-in "
                                   "BEFORE anything imports the reference's models package")
        d = dropin_dir()
        sys.path.insert(0, d)
        try:
            importlib.invalidate_caches()
            mod = importlib.import_module("models.vqa_model")
        finally:
            while d in sys.path:                       # utils.* / data.* must stay the caller's own packages
                sys.path.remove(d)
        if not is_bound():                             # a stale finder cache or a `models` earlier on the path: refuse loudly
            raise RuntimeError(f"bind(): models.vqa_model resolved to {getattr(mod, '__file__', '?')}, not to the drop-in")
    if project_root is not None:
        ref_models = os.path.join(os.path.abspath(project_root), "models")
        pkg = sys.modules["models"]
        if os.path.isdir(ref_models) and ref_models not in list(pkg.__path__):
            pkg.__path__.append(ref_models)
    return sys.modules["models.vqa_model"]


def find_project_root(entry: str) -> str:
    """The reference checkout an entry point belongs to: the nearest ancestor directory that holds `models/vqa_model.py`
    (what `PROJECT_ROOT = Path(__file__).parent.parent` evaluates to for training/*.py and api/*.py, the script's own
    directory for reproduce_issue.py); the script's directory if there is none."""
    d = os.path.dirname(os.path.abspath(entry))
    probe = d
    for _ in range(4):
        if os.path.isfile(os.path.join(probe, "models", "vqa_model.py")):
            return probe
        up = os.path.dirname(probe)
        if up == probe:
            break
        probe = up
    return d


def run(entry: str, argv: Optional[List[str]] = None, run_name: str = "__main__", dtype: Optional[str] = None):
    """`python <entry> <argv...>` with the drop-in bound: what `run_reference.py` does.  Returns the script's globals
    (with `run_name != "__main__"` the script's `if __name__ == "__main__":` block does not run -- tests use that to get
    at `Trainer` & co. exactly as the entry point defined them)."""
    entry = os.path.abspath(entry)
    if not os.path.isfile(entry):
        raise FileNotFoundError(entry)
    bind(find_project_root(entry), dtype)
    script_dir = os.path.dirname(entry)
    old_argv = sys.argv
    sys.argv = [entry] + list(argv or [])
    sys.path.insert(0, script_dir)                     # what `python script.py` does
    try:
        return runpy.run_path(entry, run_name=run_name)
    finally:
        sys.argv = old_argv
        if script_dir in sys.path:
            sys.path.remove(script_dir)
