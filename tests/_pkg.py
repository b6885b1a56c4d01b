"""Load the hyphen-named product package."""
import importlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
PKG_NAME = "visual-question-answering-vqa-system_amd"


def pkg():
    return importlib.import_module(PKG_NAME)


def sub(name):
    return importlib.import_module(PKG_NAME + "." + name)
