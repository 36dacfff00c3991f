"""CPU: damaged index images never take the process down — the loader validates before it trusts (image.hpp), the drop-in's
host-side entry points walk only what was validated. Mutations: random byte flips, flips in the header / level region, truncation,
flips in the matrix dimensions; golden images of the reference as the starting point."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script,seed", [("engine_images.py", 1), ("engine_images.py", 2), ("dropin_images.py", 1),
                                         ("dropin_images.py", 2)])
def test_mutated_images_are_survived(script, seed):
    out = subprocess.run([sys.executable, os.path.join(HERE, "fuzz", script), str(seed), "250"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert "survived" in out.stdout
