"""pytest configuration: registers the `gpu` marker and puts the product package and the repo root on
sys.path (the package directory name contains a hyphen, so it is reached through its parent)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, "stopthepop-rasterization_amd")
for p in (ROOT, PKG_PARENT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
