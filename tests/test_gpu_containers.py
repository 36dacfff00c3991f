"""GPU: the device-side containers (`next` binary heap, `top` sorted buffer) against a literal Python restatement of the
reference containers (max_heap_gt index.hpp:664-835, sorted_buffer_gt index.hpp:845-956) on tie-heavy scripts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class RefHeap:  # max-heap on key, reference sift rules
    def __init__(self):
        self.e = []

    def push(self, key, slot):
        self.e.append((key, slot))
        i = len(self.e) - 1
        while i and self.e[(i - 1) // 2][0] < self.e[i][0]:
            self.e[(i - 1) // 2], self.e[i] = self.e[i], self.e[(i - 1) // 2]
            i = (i - 1) // 2

    def pop(self):
        top = self.e[0]
        self.e[0] = self.e[-1]
        self.e.pop()
        i, n = 0, len(self.e)
        while True:
            best, left, right = i, 2 * i + 1, 2 * i + 2
            if left < n and self.e[best][0] < self.e[left][0]:
                best = left
            if right < n and self.e[best][0] < self.e[right][0]:
                best = right
            if best == i:
                break
            self.e[i], self.e[best] = self.e[best], self.e[i]
            i = best
        return top


def ref_sorted_insert(buf, d, slot, limit):
    lo = 0
    while lo < len(buf) and buf[lo][0] < d:
        lo += 1
    if lo == limit:
        return
    if len(buf) == limit:
        buf.pop()
    buf.insert(lo, (d, slot))


@pytest.mark.parametrize("seed,levels,count,limit", [(0, 4, 300, 16), (1, 1000, 500, 64), (2, 2, 700, 100),
                                                     (3, 8, 64, 1), (4, 3, 1000, 200)])
def test_containers_match_reference_rules(seed, levels, count, limit):
    from usearch_amd import index as ua
    rng = np.random.default_rng(seed)
    kinds = rng.choice([0, 0, 0, 1, 2, 2], size=count).astype(np.uint32)
    keys = rng.integers(0, levels, size=count).astype(np.float32) * 0.5 - 1.0  # few distinct values ⇒ ties everywhere
    slots = np.arange(count, dtype=np.uint32)
    heap, top, popped = RefHeap(), [], []
    for kind, key, slot in zip(kinds, keys, slots):
        if kind == 0:
            heap.push(float(key), int(slot))
        elif kind == 1 and heap.e:
            popped.append(heap.pop())
        elif kind == 2:
            ref_sorted_insert(top, float(key), int(slot), limit)
    (pk, ps), (tk, ts) = ua.test_containers(kinds, keys, slots, limit)
    assert [(float(a), int(b)) for a, b in zip(pk, ps)] == popped
    assert [(float(a), int(b)) for a, b in zip(tk, ts)] == top
