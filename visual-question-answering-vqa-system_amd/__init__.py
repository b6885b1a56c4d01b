"""MI355X-native forward/backward path of the VQA classifier (drop-in for models.vqa_model.VQAModel).

Import with importlib (the directory name is not a Python identifier):
    pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
"""
from . import _lib, kernels  # noqa: F401
