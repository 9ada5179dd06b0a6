import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The HIP library is a build artefact (git-ignored): compile it if this checkout has none yet, so the suite never
# "passes" on a missing extension (hipcc cross-compiles gfx950 without a GPU).
if not os.path.exists(os.path.join(ROOT, "dns_slam_amd", "libdns_hip.so")):
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "dns_slam_amd", "csrc"), "-j8"], check=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests never run without a visible device, whatever -m says
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
