"""Mutated index images through the drop-in's host-side entry points (metadata / view / load, then size, contains, count, get,
serialized_length, save_buffer, remove on whatever opened). Run by tests/test_host_fuzz.py in a child process."""
import ctypes as C, os, sys
import numpy as np
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.environ.get("USEARCH_AMD_DROPIN_LIBRARY") or os.path.join(ROOT, "usearch_amd", "lib", "libusearch_c.so"))
err_t = C.POINTER(C.c_char_p)
L.usearch_init.restype = C.c_void_p; L.usearch_init.argtypes = [C.c_void_p, err_t]
L.usearch_view_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_t]
L.usearch_load_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_t]
L.usearch_size.restype = C.c_size_t; L.usearch_size.argtypes = [C.c_void_p, err_t]
L.usearch_contains.restype = C.c_bool; L.usearch_contains.argtypes = [C.c_void_p, C.c_uint64, err_t]
L.usearch_count.restype = C.c_size_t; L.usearch_count.argtypes = [C.c_void_p, C.c_uint64, err_t]
L.usearch_get.restype = C.c_size_t; L.usearch_get.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_void_p, C.c_int, err_t]
L.usearch_serialized_length.restype = C.c_size_t; L.usearch_serialized_length.argtypes = [C.c_void_p, err_t]
L.usearch_save_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, err_t]
L.usearch_remove.restype = C.c_size_t; L.usearch_remove.argtypes = [C.c_void_p, C.c_uint64, err_t]
L.usearch_free.argtypes = [C.c_void_p, err_t]
L.usearch_metadata_buffer.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, err_t]
golden = os.path.join(ROOT, "tests", "golden")
names = [n for n in os.listdir(golden) if n.endswith(".npz")]
opened = refused = 0
for round_ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    image = np.load(os.path.join(golden, names[round_ % len(names)]))["image"].copy()
    kind = rng.integers(0, 4)
    if kind == 0:
        for _ in range(int(rng.integers(1, 6))):
            image[rng.integers(0, len(image))] ^= np.uint8(rng.integers(1, 256))
    elif kind == 1:
        rows, cols = np.frombuffer(image[:8].tobytes(), dtype=np.uint32)
        head = 8 + int(rows) * int(cols)
        for _ in range(int(rng.integers(1, 4))):
            image[head + rng.integers(0, min(400, len(image) - head))] ^= np.uint8(rng.integers(1, 256))
    elif kind == 2:
        image = image[: int(rng.integers(0, len(image)))]
    else:
        image[rng.integers(0, 8)] ^= np.uint8(rng.integers(1, 256))
    image = np.ascontiguousarray(image)
    err = C.c_char_p()
    options = (C.c_uint8 * 128)()
    L.usearch_metadata_buffer(C.c_void_p(image.ctypes.data), image.size, options, C.byref(err))
    err = C.c_char_p()
    index = L.usearch_init(None, C.byref(err))
    (L.usearch_view_buffer if round_ % 2 else L.usearch_load_buffer)(index, C.c_void_p(image.ctypes.data), image.size, C.byref(err))
    if err.value:
        refused += 1
    else:
        opened += 1
        err = C.c_char_p()
        n = L.usearch_size(index, C.byref(err))
        for key in (0, 1, 5, int(rng.integers(0, 1 << 40))):
            L.usearch_contains(index, key, C.byref(err)); L.usearch_count(index, key, C.byref(err))
            out = np.zeros(1 << 16, dtype=np.uint8)
            L.usearch_get(index, key, 1, C.c_void_p(out.ctypes.data), 1, C.byref(err))
        length = L.usearch_serialized_length(index, C.byref(err))
        if length and length < (1 << 28):
            out = np.zeros(length, dtype=np.uint8)
            L.usearch_save_buffer(index, C.c_void_p(out.ctypes.data), length, C.byref(err))
        L.usearch_remove(index, 1, C.byref(err))
    L.usearch_free(index, C.byref(err))
print("survived:", opened, "opened,", refused, "refused")
