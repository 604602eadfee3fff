import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """the state of the diffusers pin is a loud line in every test log (CPU and GPU runs), not a
    silent skip of tests/test_diffusers_pin.py.  Nothing is installed from here: the wheelhouse is only listed."""
    import glob
    import importlib.util
    if importlib.util.find_spec("diffusers") is not None:
        import diffusers
        terminalreporter.write_line(f"PARITY PIN: diffusers {diffusers.__version__} importable -- tests/test_diffusers_pin.py "
                                    "compares the oracle's leaf math with the real classes")
        return
    wheels = sorted(os.path.basename(p) for p in glob.glob("/opt/wheelhouse/diffusers*"))
    terminalreporter.write_line("PARITY UNPINNED: diffusers not importable (pinned 0.27.0 by the reference's requirements.txt:3; "
                                f"/opt/wheelhouse has {wheels if wheels else 'no diffusers wheel'}) -- oracle leaf math, the four "
                                "schedulers and the VAE are checked against torch.nn.functional / float64 NumPy only", red=True)
