"""The pipelined team's early naming (usearch_amd/csrc/kernels.hpp, `search_one`, the `pipelined` beam): the leader names the next
member to expand BEFORE it commits the hop it has just measured. The kernel's claim is that the rule is exact whenever it does not
fall back to "commit first, look again". This file restates both sides in plain Python over the reference's container semantics
(index.hpp:928-939 `sorted_buffer_gt::insert`: lower-bound placement — a new element lands in front of its equals; index.hpp:4233-
4240: a newcomer is taken while the buffer has room or it is strictly closer than the worst kept member) and checks the claim on
a hundred thousand random states full of ties. No GPU: the device side of the same claim is tests/test_gpu_search_parity.py
(test_team_and_one_wave_agree, test_team_settles_ties_like_one_wave)."""
import random


class Top:
    """`top` with the frontier riding in it: ascending (distance, slot, closed) cells, at most `limit` of them."""

    def __init__(self, limit):
        self.limit = limit
        self.cells = []

    def radius(self):
        return self.cells[-1][0]

    def accepts(self, distance):
        return len(self.cells) < self.limit or distance < self.radius()

    def insert(self, distance, slot, closed=False):
        place = 0
        while place < len(self.cells) and self.cells[place][0] < distance:  # the first cell that is not smaller
            place += 1
        self.cells.insert(place, (distance, slot, closed))
        del self.cells[self.limit:]

    def first_open(self):
        for index, (distance, slot, closed) in enumerate(self.cells):
            if not closed:
                return index, distance, slot
        return None

    def close(self, index):
        distance, slot, _ = self.cells[index]
        self.cells[index] = (distance, slot, True)

    def copy(self):
        other = Top(self.limit)
        other.cells = list(self.cells)
        return other


def commit_then_look(top, newcomers):
    """The one-wave kernel's order: every newcomer in list order with the reference's tests, then the closest open member."""
    for distance, slot in newcomers:
        if top.accepts(distance):
            top.insert(distance, slot)
    found = top.first_open()
    if found is None:
        return None
    top.close(found[0])
    return found[2]


def name_then_commit(top, newcomers):
    """The pipelined leader's order. → (expanded slot or None, fell_back)."""
    found = top.first_open()
    from_index = None
    room = top.limit - len(top.cells)
    exact = room == 0 or room >= len(newcomers)  # else the buffer fills up mid-commit
    lands = [room != 0 or distance < top.radius() for distance, _ in newcomers]
    if exact and any(lands):
        best = min(distance for (distance, _), ok in zip(newcomers, lands) if ok)
        at_best = [i for i, ((distance, _), ok) in enumerate(zip(newcomers, lands)) if ok and distance == best]
        if found is None or best < found[1]:
            if len(at_best) > 1:
                exact = False  # two newcomers at the smallest distance: their order in the array decides
            else:
                from_index = at_best[0]
        elif best == found[1]:
            exact = False  # a newcomer lands in front of its equal
    if not exact:
        return commit_then_look(top, newcomers), True
    if from_index is None and found is None:
        for distance, slot in newcomers:  # nothing of it lands
            assert not top.accepts(distance)
        return None, False
    if from_index is None:
        top.close(found[0])  # the flag travels with the cell while the commit shifts it
        expanded = found[2]
    else:
        expanded = newcomers[from_index][1]
    for index, (distance, slot) in enumerate(newcomers):  # the commit, in the shadow of the helpers' work
        if top.accepts(distance):
            top.insert(distance, slot, closed=index == from_index)
    return expanded, False


def random_state(rng):
    limit = rng.randint(1, 12)
    top = Top(limit)
    spread = rng.choice((3, 6, 40))  # few distinct distances: ties everywhere
    slot = 0
    for _ in range(rng.randint(1, limit)):
        top.insert(float(rng.randint(0, spread)), slot, closed=rng.random() < 0.6)
        slot += 1
    newcomers = []
    for _ in range(rng.randint(0, 7)):
        newcomers.append((float(rng.randint(0, spread + 2)), slot))
        slot += 1
    return top, newcomers


def test_early_naming_is_exact_or_falls_back():
    rng = random.Random(20260924)
    named_early = fell_back = newcomer_named = 0
    for _ in range(100_000):
        top, newcomers = random_state(rng)
        plain, piped = top.copy(), top.copy()
        expected = commit_then_look(plain, newcomers)
        got, fallback = name_then_commit(piped, newcomers)
        assert got == expected, (top.cells, newcomers, expected, got)
        assert piped.cells == plain.cells, (top.cells, newcomers, plain.cells, piped.cells)  # closed flags included
        fell_back += fallback
        named_early += not fallback
        newcomer_named += (not fallback) and got is not None and got >= len(top.cells) and any(got == s for _, s in newcomers)
    # the rule must carry real weight on both sides of the choice even in a tie-heavy population
    assert named_early > 40_000 and fell_back > 5_000 and newcomer_named > 5_000, (named_early, fell_back, newcomer_named)


def test_distinct_distances_never_fall_back_once_the_buffer_is_full():
    """Float distances of real data do not tie: with the buffer full (the steady state of a walk) the leader never has to commit first."""
    rng = random.Random(7)
    for _ in range(20_000):
        limit = rng.randint(2, 16)
        values = rng.sample(range(10_000), limit + 8)
        top = Top(limit)
        for slot in range(limit):
            top.insert(float(values[slot]), slot, closed=rng.random() < 0.7)
        newcomers = [(float(values[limit + i]), limit + i) for i in range(rng.randint(0, 8))]
        plain, piped = top.copy(), top.copy()
        expected = commit_then_look(plain, newcomers)
        got, fallback = name_then_commit(piped, newcomers)
        assert not fallback and got == expected and piped.cells == plain.cells
