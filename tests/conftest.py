"""pytest configuration: the `gpu` marker (tests that need a real MI355X) and shared dataset/index builders."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    # some tests open the drop-in library or the HIP runtime directly and others import torch later in the same process:
    # whichever comes first, there must be one HIP runtime (usearch_amd/index.py `_share_hip_runtime`)
    from usearch_amd.index import _share_hip_runtime
    _share_hip_runtime()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference (oracle/_ref). Built here when /root/reference is mounted; prebuilt on the GPU box."""
    from oracle import refbind
    if not refbind.available():
        if os.path.isdir("/root/reference/include/usearch"):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/libusearch_ref.so is not built and /root/reference is not mounted")
    return refbind
