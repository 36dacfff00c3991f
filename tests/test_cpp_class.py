"""include/usearch_amd.hpp (the `index_dense_gt`-shaped C++ veneer over the drop-in C ABI): compiles and links everywhere,
runs its scenario on an MI355X."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = "/tmp/usearch_amd_class_test"


def build():
    lib = os.path.join(ROOT, "usearch_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "class_test.cpp"), "-L", lib, "-lusearch_c",
                           f"-Wl,-rpath,{lib}", "-o", BINARY])


def test_header_compiles_and_links():
    build()
    out = subprocess.check_output([BINARY, "link"]).decode()
    assert "usearch 2.21.0" in out


@pytest.mark.gpu
def test_class_scenario_on_the_device():
    build()
    out = subprocess.run([BINARY, "run"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "class test passed" in out.stdout, out.stdout + out.stderr
