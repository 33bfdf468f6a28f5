import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _library_defaults():
    """trinerflet_amd.install_dropin() (tests of the drop-in boundary call it) turns the windowed rebuild under autograd on
    for encoders built afterwards; every test starts from the library's own default."""
    mod = sys.modules.get("trinerflet_amd.triplaneencoder.triplane_encoder")
    if mod is not None:
        mod.WINDOWED_AUTOGRAD = False
    opt = sys.modules.get("trinerflet_amd.optim")
    if opt is not None:
        opt.unpatch_torch_adam()
    fld = sys.modules.get("trinerflet_amd.nerf.field")
    if fld is not None:
        fld._FusedField.deterministic = False
        fld._FusedField.early_sort = True
        fld._FusedField.binned_backward = True
    yield
