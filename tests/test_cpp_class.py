"""include/usearch/index_dense.hpp — the reference's own class surface, `unum::usearch::index_dense_gt`, over the engine: compiles and
links everywhere, runs the reference's loops on an MI355X."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- include/usearch/index_dense.hpp: the reference's own class surface, `unum::usearch::index_dense_gt`, over the engine
LOOP_BINARY = "/tmp/usearch_amd_bench_loop"
CLASS_DIR = os.path.join(ROOT, "oracle", "_ref", "class")


def build_loop():
    lib = os.path.join(ROOT, "usearch_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fopenmp", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "bench_loop.cpp"), "-L", lib, "-l:libusearch_c.so",
                           f"-Wl,-rpath,{lib}", "-o", LOOP_BINARY])


def test_reference_shaped_caller_compiles_against_the_class_surface():
    """cpp/bench.cpp's add / search loops over `unum::usearch::index_dense_t` from include/usearch/index_dense.hpp."""
    build_loop()
    assert "behind index_dense_t" in subprocess.check_output([LOOP_BINARY, "link"]).decode()


def test_the_references_c_binding_compiled_against_the_class_surface_exports_the_abi():
    """oracle/Makefile `class_lib`: /root/reference/c/lib.cpp with `<usearch/index_dense.hpp>` resolving to THIS repository's
    header. The library must carry the reference's 38 symbols, defined by the reference's binding code, and lean on the drop-in."""
    library = os.path.join(CLASS_DIR, "libusearch_c_over_class.so")
    if not os.path.exists(library):
        pytest.skip("oracle/_ref/class was not built (`make -C oracle class_lib` needs /root/reference)")
    symbols = subprocess.check_output(["nm", "-D", "--defined-only", library]).decode()
    exported = {line.split()[-1] for line in symbols.splitlines() if " T usearch_" in line}
    assert len(exported) == 38 and {"usearch_init", "usearch_search", "usearch_filtered_search", "usearch_exact_search"} <= exported
    needed = subprocess.check_output(["readelf", "-d", library]).decode()
    assert "libusearch_c.so" in needed, "the class surface must reach the engine through the drop-in library"
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", library]).decode()
    assert "usearch_amd_c_api" in undefined


@pytest.mark.gpu
def test_bench_loop_on_the_device():
    build_loop()
    out = subprocess.run([LOOP_BINARY, "run"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "bench loop passed" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.gpu
def test_the_references_own_c_test_passes_through_its_own_binding_over_the_class_surface():
    """/root/reference/c/test.c → (the reference's c/lib.cpp compiled against include/usearch/index_dense.hpp) → drop-in → MI355X."""
    binary = os.path.join(CLASS_DIR, "reference_test_c")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/class was not built (`make -C oracle class_lib` needs /root/reference)")
    result = subprocess.run([binary], capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout[-3000:] + result.stderr[-3000:]
