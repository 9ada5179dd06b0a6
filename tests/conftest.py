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


def pytest_sessionfinish(session, exitstatus):
    """Parity figures of every assert_close of the session (scale-relative error AND worst element-wise ratio)."""
    try:
        import json
        import util
        if not util.REPORT:
            return
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        rows = [{"what": w, "scale_rel_err": e, "elementwise_ratio": r, "rtol": t} for w, e, r, t in util.REPORT]
        rows.sort(key=lambda d: -(d["elementwise_ratio"] if d["elementwise_ratio"] == d["elementwise_ratio"] else 0))
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump({"n": len(rows), "rows": rows}, f, indent=1)
    except Exception:
        pass
