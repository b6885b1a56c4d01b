"""MI355X-native forward/backward path of the VQA classifier (drop-in for models.vqa_model.VQAModel).

Import with importlib (the directory name is not a Python identifier):
    pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
"""
from . import _lib, kernels  # noqa: F401
from . import layout, engine, trainer, flops  # noqa: F401,E402


def dropin_path() -> str:
    """Directory to put on sys.path so that `from models.vqa_model import VQAModel` resolves to the HIP drop-in."""
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def load_dropin():
    """Import the drop-in `models.vqa_model` under a private name, without disturbing an already imported `models` package
    (tests and the harness use this; a CALLER of the reference reaches the drop-in through binding.bind() / run_reference.py)."""
    return _load_dropin_file("vqa_hip_dropin_models_vqa_model", "models", "vqa_model.py")


def load_dropin_metrics():
    """The drop-in `utils.metrics` (device-side VQAAccuracy), without disturbing an already imported `utils` package."""
    return _load_dropin_file("vqa_hip_dropin_utils_metrics", "utils", "metrics.py")


def _load_dropin_file(name, *rel):
    import importlib.util
    import os
    import sys
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(dropin_path(), *rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_dropin_tokenizer():
    """The drop-in `utils.tokenizer` (batch encoding packed on the GPU), without disturbing an imported `utils` package."""
    return _load_dropin_file("vqa_hip_dropin_utils_tokenizer", "utils", "tokenizer.py")


def load_dropin_preprocess():
    """The drop-in `data.preprocess` (ToTensor + Normalize on the GPU, gpu_collate_fn)."""
    return _load_dropin_file("vqa_hip_dropin_data_preprocess", "data", "preprocess.py")
