#!/usr/bin/python3
"""Run one of the reference's UNCHANGED entry points on the MI355X drop-in.

    cd <reference checkout>
    python /root/repo/run_reference.py [--dtype bf16|fp32] training/train.py --demo --no-amp
    python /root/repo/run_reference.py reproduce_issue.py
    python /root/repo/run_reference.py training/evaluate.py --checkpoint checkpoints/best_model.pth --demo

The entry points put their own checkout first on sys.path before `from models.vqa_model import ...`
(training/train.py:44-46, training/evaluate.py:33-36, api/inference.py:26-29), so PYTHONPATH cannot redirect that import;
this launcher pre-imports the drop-in's `models.vqa_model` into sys.modules, leaves `utils.*` / `data.*` to the reference and
then runs the script as `__main__` in THIS process (no re-exec, nothing touches the GPU before the script does).
Logic: visual-question-answering-vqa-system_amd/binding.py.
"""
import importlib
import os
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    dtype = None
    while argv and argv[0].startswith("--"):
        if argv[0] == "--dtype" and len(argv) >= 2:
            dtype, argv = argv[1], argv[2:]
        elif argv[0].startswith("--dtype="):
            dtype, argv = argv[0].split("=", 1)[1], argv[1:]
        elif argv[0] in ("-h", "--help"):
            print(__doc__)
            return 0
        else:
            break
    if not argv:
        print(__doc__, file=sys.stderr)
        return 2
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.append(here)            # behind everything of the caller's: only to find the hyphenated package directory
    binding = importlib.import_module("visual-question-answering-vqa-system_amd.binding")
    binding.run(argv[0], argv[1:], run_name="__main__", dtype=dtype)
    return 0


if __name__ == "__main__":
    sys.exit(main())
