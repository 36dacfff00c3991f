/**
 *  usearch_amd/csrc/image.hpp — host-side reader of a serialized USearch v2 index image.
 *
 *  Byte layout, as written by the reference (`/root/reference/include/usearch/…`):
 *    index_dense.hpp:1004-1030   [u32 rows][u32 cols = bytes per vector][rows × cols vector bytes]   (or two u64 when
 *                                saved with `use_64_bit_dimensions`; sniffed like index_dense.hpp:321-371 does)
 *    index_dense.hpp:42-79       64-byte head: "usearch" magic(7) | u16×3 version | u8 metric | u8 scalar | u8 key kind |
 *                                u8 slot kind | u64 present | u64 deleted | u64 dimensions | u8 multi | zero padding
 *    index.hpp:1863-1869         5 × u64: size, connectivity, connectivity_base, max_level, entry_slot
 *    index.hpp:3298-3305         i16 level per node
 *    index.hpp:3308-3314         node tapes back to back, each (index.hpp:2085, 3731-3748):
 *                                u64 key | i16 level | {u32 count, slot × M0} | level × {u32 count, slot × M}
 *                                slot = u32 (`index_dense_t`) or the 5-byte `uint40_t` of index.hpp:969-1031 (the compressed-slot
 *                                kind of `index_dense_big_t`, index_dense.hpp:2230) — accepted while the index has fewer than 2³²
 *                                members, i.e. while every slot fits the device's 32-bit cells; 128-bit `uuid_t` keys (the other
 *                                half of `index_dense_big_t`) have no counterpart in the 64-bit key ABI and are refused by name
 *  Everything after the matrix is unaligned, hence the memcpy loads. The image is borrowed, never copied.
 */
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace usearch_amd {

struct image_t {
    const std::uint8_t* bytes = nullptr;
    std::size_t length = 0;

    std::uint64_t rows = 0, cols = 0;
    const std::uint8_t* vectors = nullptr;
    std::uint64_t vector_stride = 0; ///< bytes between rows of `vectors` (= cols inside an image)
    /// For images saved with `exclude_vectors` (index_dense.hpp:1004: the file starts with the 64-byte head): the caller's own
    /// matrix, one row per slot in slot order, set BEFORE `open`. The reference keeps such vectors outside the index as well
    /// (`copy_vector = false`, index_dense.hpp:2038-2039).
    const std::uint8_t* external_vectors = nullptr;
    std::uint64_t external_stride = 0;

    std::uint16_t version[3] = {0, 0, 0};
    metric_kind_t metric = metric_unknown_k;
    scalar_kind_t scalar = scalar_unknown_k;
    std::uint64_t count_present = 0, count_deleted = 0, dimensions = 0;
    bool multi = false;

    std::uint64_t size = 0, connectivity = 0, connectivity_base = 0, max_level = 0, entry_slot = 0;
    std::uint32_t slot_bytes = 4; ///< bytes per neighbour slot on the tapes: 4 (u32) or 5 (uint40_t)
    const std::uint8_t* levels = nullptr;
    const std::uint8_t* tapes = nullptr; ///< first node tape
    std::size_t tapes_length = 0;

    template <typename scalar_at> static scalar_at load(const std::uint8_t* p) {
        scalar_at v;
        std::memcpy(&v, p, sizeof(v));
        return v;
    }

    std::size_t node_bytes(std::int16_t level) const {
        return 10 + (4 + slot_bytes * connectivity_base) + (std::size_t)level * (4 + slot_bytes * connectivity);
    }

    /// Parses the fixed-size parts. Returns nullptr on success or a static message (the reference's wording where
    /// the reference has one: index_dense.hpp:1102-1146, index.hpp:3330-3370).
    const char* open(const void* image, std::size_t image_length) {
        bytes = static_cast<const std::uint8_t*>(image);
        length = image_length;
        const std::uint8_t* p = bytes;
        const std::uint8_t* const end = bytes + length;
        if (length < 8)
            return "Failed to read 32-bit dimensions of the matrix";
        // Which of the three layouts of index_dense.hpp:1004-1030 is this? The reference sniffs it the same way
        // (`index_dense_metadata_from_buffer`, index_dense.hpp:321-371): the 64-byte head right away = saved with
        // `exclude_vectors`; else the magic behind a matrix announced by two u32 — or by two u64 (`use_64_bit_dimensions`).
        const bool without_vectors = length >= 7 && std::memcmp(p, "usearch", 7) == 0;
        if (without_vectors && !external_vectors)
            return "The image was saved without its vectors (exclude_vectors): hand the matrix in alongside "
                   "(usearch_amd_snapshot_from_parts)";
        std::size_t dimensions_length = without_vectors ? 0 : 8;
        rows = without_vectors ? 0 : load<std::uint32_t>(p);
        cols = without_vectors ? 0 : load<std::uint32_t>(p + 4);
        const auto magic_behind = [&](std::uint64_t r, std::uint64_t c, std::size_t header) {
            if (c && r > (std::uint64_t)length / c)
                return false;
            const std::uint64_t offset = r * c + header;
            return offset + 64 <= (std::uint64_t)length && std::memcmp(bytes + offset, "usearch", 7) == 0;
        };
        if (!without_vectors && !magic_behind(rows, cols, 8) && length >= 16) {
            const std::uint64_t rows64 = load<std::uint64_t>(p), cols64 = load<std::uint64_t>(p + 8);
            if (magic_behind(rows64, cols64, 16))
                rows = rows64, cols = cols64, dimensions_length = 16;
        }
        p += dimensions_length;
        if ((std::uint64_t)(end - p) < rows * cols)
            return "Failed to read vectors";
        vectors = p;
        vector_stride = cols;
        p += rows * cols;
        if ((std::size_t)(end - p) < 64)
            return "Failed to read the index ";
        if (std::memcmp(p, "usearch", 7) != 0)
            return "Magic header mismatch - the file isn't an index";
        version[0] = load<std::uint16_t>(p + 7);
        version[1] = load<std::uint16_t>(p + 9);
        version[2] = load<std::uint16_t>(p + 11);
        if (version[0] != 2)
            return "File format may be different, please rebuild";
        metric = (metric_kind_t)p[13];
        scalar = (scalar_kind_t)p[14];
        if (p[15] == 3 /* uuid_k */)
            return "128-bit keys (the uuid_t of index_dense_big_t) have no counterpart in the 64-bit key ABI of the device index";
        if (p[15] != scalar_u64_k)
            return "Key type doesn't match, consider rebuilding";
        if (p[16] == 2 /* u40_k: index.hpp:969-1031 */)
            slot_bytes = 5;
        else if (p[16] != scalar_u32_k)
            return "Slot type doesn't match, consider rebuilding";
        count_present = load<std::uint64_t>(p + 17);
        count_deleted = load<std::uint64_t>(p + 25);
        dimensions = load<std::uint64_t>(p + 33);
        multi = p[41] != 0;
        p += 64;
        if ((std::size_t)(end - p) < 40)
            return "Failed to pull the header from the stream";
        size = load<std::uint64_t>(p);
        connectivity = load<std::uint64_t>(p + 8);
        connectivity_base = load<std::uint64_t>(p + 16);
        max_level = load<std::uint64_t>(p + 24);
        entry_slot = load<std::uint64_t>(p + 32);
        p += 40;
        if (size && max_level > 0x7FFF) // levels are i16 on the tapes (index.hpp:2116-2137): nothing larger can be a node's level
            return "Failed to pull the header from the stream";
        if (without_vectors) { // the matrix is the caller's: one row of the head's scalar kind and dimensions per slot
            rows = size;
            cols = bytes_per_vector(scalar, dimensions);
            vectors = external_vectors;
            vector_stride = external_stride ? external_stride : cols;
            if (vector_stride < cols)
                return "Stride is smaller than one vector";
        }
        if (size != rows)
            return "Index size and the number of vectors doesn't match";
        if (size >= none_slot_k)
            return "Index is too large for 32-bit slots";
        if ((std::uint64_t)(end - p) < size * 2)
            return "Failed to pull nodes levels from the stream";
        levels = p;
        p += size * 2;
        tapes = p;
        tapes_length = (std::size_t)(end - p);
        if (size && (entry_slot >= size || connectivity == 0 || connectivity_base == 0))
            return "Failed to pull the header from the stream";
        if (cols != bytes_per_vector(scalar, dimensions))
            return "Vector size doesn't match the scalar kind and dimensions";
        // The node tapes are variable-length (one list per level): everything that later walks them — key lookup, `get`,
        // the flattener — relies on this one pass having seen that every level is sane and that the tapes fit the image
        // (a truncated or corrupt file fails here, like the reference's "Failed to pull nodes from the stream").
        if (connectivity >= none_slot_k || connectivity_base >= none_slot_k)
            return "Failed to pull the header from the stream";
        std::uint64_t needed = 0;
        for (std::uint64_t i = 0; i < size; ++i) {
            const std::int16_t node_level = level(i);
            if (node_level < 0 || (std::uint64_t)node_level > max_level)
                return "Failed to pull nodes from the stream";
            needed += node_bytes(node_level);
            if (needed > tapes_length)
                return "Failed to pull nodes from the stream";
        }
        return nullptr;
    }

    std::int16_t level(std::uint64_t slot) const { return load<std::int16_t>(levels + 2 * slot); }
};

} // namespace usearch_amd
