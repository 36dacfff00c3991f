"""usearch_amd — MI355X-native batched HNSW search behind the USearch C ABI (search path only).

`usearch_amd.Index.restore(path_or_image)` uploads a serialized USearch index to HBM and `Index.search(batch, k)`
walks it with hand-written gfx950 kernels. See DESIGN.md.
"""
from .index import (BatchMatches, BuildConfig, BuildStats, BuiltIndex, Index, build, Matches, Stats, Tuning, cast, device_count, exact_search,  # noqa: F401
                    library, merge_many, note_device_free, settle, LIBRARY_PATH, EXPORTED_SYMBOLS)

__all__ = ["Index", "BuiltIndex", "build", "Matches", "BatchMatches", "Tuning", "Stats", "cast", "device_count", "library", "note_device_free", "settle"]
