"""Mutated index images against the engine's loader entry point (no device needed: parsing and validation come first). Run by
tests/test_host_fuzz.py in a child process: a crash is a failed test, not a dead test session."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import usearch_amd
library = usearch_amd.library()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
golden = os.path.join(ROOT, "tests", "golden")
names = [n for n in os.listdir(golden) if n.endswith(".npz")]
accepted = rejected = 0
for round_ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 400):
    image = np.load(os.path.join(golden, names[round_ % len(names)]))["image"].copy()
    kind = rng.integers(0, 4)
    if kind == 0:      # flip a few random bytes anywhere
        for _ in range(int(rng.integers(1, 6))):
            image[rng.integers(0, len(image))] ^= np.uint8(rng.integers(1, 256))
    elif kind == 1:    # flip bytes in the header region (dimensions, sizes, levels)
        rows, cols = np.frombuffer(image[:8].tobytes(), dtype=np.uint32)
        head = 8 + int(rows) * int(cols)
        for _ in range(int(rng.integers(1, 4))):
            image[head + rng.integers(0, min(400, len(image) - head))] ^= np.uint8(rng.integers(1, 256))
    elif kind == 2:    # truncate
        image = image[: int(rng.integers(0, len(image)))]
    else:              # header fields of the matrix
        image[rng.integers(0, 8)] ^= np.uint8(rng.integers(1, 256))
    image = np.ascontiguousarray(image)
    err = C.c_char_p()
    handle = library.usearch_amd_snapshot_from_buffer(C.c_void_p(image.ctypes.data), image.size, 0, C.byref(err))
    if handle:  # a device is present and the damage was harmless (or caught on the device and refused there)
        library.usearch_amd_snapshot_free(C.c_void_p(handle), C.byref(err))
        accepted += 1
        continue
    assert err.value
    if b"device" in err.value.lower() or b"hip" in err.value.lower():
        accepted += 1  # passed validation, failed for want of a device
    else:
        rejected += 1
print("survived", accepted, "passed validation,", rejected, "rejected")
