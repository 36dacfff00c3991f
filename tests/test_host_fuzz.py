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


ROOT = os.path.dirname(HERE)
ASAN_DIR = os.path.join(ROOT, "usearch_amd", "lib_asan")


@pytest.mark.parametrize("script", ["engine_images.py", "dropin_images.py"])
def test_mutated_images_under_the_sanitizers(script):
    """The same fuzzers against the host side built with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C usearch_amd/csrc
    sanitize`): an out-of-bounds read the plain build happens to survive is a report — and a failed test — here."""
    runtime = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True,
                             text=True).stdout.strip()
    if not (os.path.exists(os.path.join(ASAN_DIR, "libusearch_c.so")) and os.path.exists(runtime)):
        pytest.skip("no sanitizer build (make -C usearch_amd/csrc sanitize)")
    env = dict(os.environ, LD_PRELOAD=runtime, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               USEARCH_AMD_LIBRARY=os.path.join(ASAN_DIR, "libusearch_amd.so"),
               USEARCH_AMD_DROPIN_LIBRARY=os.path.join(ASAN_DIR, "libusearch_c.so"))
    out = subprocess.run([sys.executable, os.path.join(HERE, "fuzz", script), "3", "200"], capture_output=True, text=True, timeout=900,
                         env=env)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-4000:])
    assert "survived" in out.stdout and "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
