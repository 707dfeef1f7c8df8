"""CPU: executable statements of two workgroup algorithms of nann_amd/csrc/nann_device.h, checked
against the reference semantics they must reproduce.  They are models (numpy / plain Python with
the atomic operations of a phase applied in random lane order), kept next to the parity tests so
that a change to the device code's logic can be tried here first:

* wg_filter_chunk / wg_filter_chunk_packed -- BitmapRefDifference's ordered first-occurrence scan
  (UO/bitmap_op/bitmap_ops.cc:224-232) done 2048 ids at a time with set-bits-then-arbitrate;
* wg_topk_impl's radix search for the k-th largest key, on raw keys (common-prefix skip) and on
  key - min(key) (NANN_TOPK_MINSUB)."""
import numpy as np
import pytest

EMPTY = 0xFFFFFFFF


def serial_scan(xs, visited):
    out = []
    for x in map(int, xs):
        if x not in visited:
            visited.add(x)
            out.append(x)
    return out


def _thread_interleaving(n, per, rng):
    """Positions in an order that is random across threads but in program order inside one."""
    pend = {t: list(range(t * per, min(n, (t + 1) * per))) for t in range((n + per - 1) // per)}
    keys, seq = list(pend), []
    while keys:
        t = keys[rng.integers(len(keys))]
        seq.append(pend[t].pop(0))
        if not pend[t]:
            keys.remove(t)
    return seq


def filter_with_preread(xs, visited, rng, per=2, slots=2048):
    """wg_filter_chunk: pre-read, barrier, set, losers record min(position), winners join."""
    n = len(xs)
    h0 = lambda x: ((x * 2654435761) & 0xFFFFFFFF) >> 21
    fresh = [int(x) not in visited for x in xs]
    for t in range(0, n, per):  # the same new id twice inside one thread
        for e in range(1, per):
            for e2 in range(e):
                if t + e < n and fresh[t + e] and fresh[t + e2] and xs[t + e] == xs[t + e2]:
                    fresh[t + e] = False
    won = [False] * n
    for p in rng.permutation(n):
        if fresh[p] and int(xs[p]) not in visited:
            visited.add(int(xs[p]))
            won[p] = True
    H, slot = [EMPTY] * slots, [None] * n
    for p in rng.permutation(n):  # contested copies
        if fresh[p] and not won[p]:
            h = h0(int(xs[p]))
            while True:
                if H[h] == EMPTY:
                    H[h] = p
                    break
                if xs[H[h]] == xs[p]:
                    H[h] = min(H[h], p)
                    break
                h = (h + 1) % slots
            slot[p] = h
    for p in rng.permutation(n):  # winners look their id up
        if won[p]:
            h = h0(int(xs[p]))
            while H[h] != EMPTY:
                if xs[H[h]] == xs[p]:
                    H[h] = min(H[h], p)
                    slot[p] = h
                    break
                h = (h + 1) % slots
    keep = [(won[p] and slot[p] is None) or (slot[p] is not None and H[slot[p]] == p) for p in range(n)]
    return [int(xs[p]) for p in range(n) if keep[p]]


def filter_packed(xs, visited, rng, per=2, slots=4096):
    """wg_filter_chunk_packed: set, winners publish id<<11|pos, barrier, losers join, barrier."""
    n = len(xs)
    h0 = lambda x: ((x * 2654435761) & 0xFFFFFFFF) >> 20
    entry = [(int(x) << 11) | p for p, x in enumerate(xs)]
    won, lost, slot = [False] * n, [False] * n, [0] * n
    for p in _thread_interleaving(n, per, rng):
        if int(xs[p]) in visited:
            lost[p] = True
        else:
            visited.add(int(xs[p]))
            won[p] = True
    H = [EMPTY] * slots
    for p in rng.permutation(n):
        if won[p]:
            h = h0(int(xs[p]))
            while H[h] != EMPTY:
                h = (h + 1) % slots
            H[h], slot[p] = entry[p], h
    cont = [False] * n
    for p in rng.permutation(n):
        if lost[p]:
            h = h0(int(xs[p]))
            while H[h] != EMPTY:
                if H[h] >> 11 == int(xs[p]):
                    H[h] = min(H[h], entry[p])
                    slot[p], cont[p] = h, True
                    break
                h = (h + 1) % slots
    return [int(xs[p]) for p in range(n) if (won[p] or cont[p]) and H[slot[p]] == entry[p]]


@pytest.mark.parametrize("model", [filter_with_preread, filter_packed])
def test_chunk_filter_equals_serial_scan(model):
    rng = np.random.default_rng(5)
    for _ in range(150):
        n = int(rng.integers(1, 2049))
        hi = int(rng.choice([8, 300, 3000, 1_000_000]))
        xs = rng.integers(0, hi, size=n)
        pre = set(int(v) for v in rng.integers(0, hi, size=int(rng.integers(0, 2000))))
        v1, v2 = set(pre), set(pre)
        assert model(xs, v1, rng) == serial_scan(xs, v2) and v1 == v2


def score_key(s):
    u = (np.asarray(s, np.float32) + np.float32(0)).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)


def radix_select(keys, k, minsub):
    """-> (T, #keys >= T, #keys > T [valid when the search ran to the last bit], passes, largest bin per pass)"""
    keys = keys.astype(np.uint64)
    n = len(keys)
    if minsub:
        kbase = int(keys.min())
        diff = int(keys.max()) - kbase
    else:
        kbase = 0
        diff = int(np.bitwise_or.reduce(keys)) ^ int(np.bitwise_and.reduce(keys))
    if diff == 0:
        return int(keys[0]), n, 0, 0, []
    hb = diff.bit_length() - 1
    T = 0 if minsub else int(np.bitwise_and.reduce(keys)) & ~((1 << (hb + 1)) - 1) & 0xFFFFFFFF
    top, kk, exact, c_gt, c_ge, passes, biggest = hb + 1, k, False, 0, n, 0, []
    while top > 0 and not exact:
        nb = min(top, 8)
        shift = top - nb
        kq = keys - kbase
        m = (kq >> top) == (T >> top) if top < 32 else np.ones(n, bool)
        h = np.bincount(((kq[m] >> shift) & ((1 << nb) - 1)).astype(np.int64), minlength=256)
        biggest.append(int(h.max()))
        above = 0
        for b in range(255, -1, -1):
            if above < kk <= above + h[b]:
                break
            above += h[b]
        T |= b << shift
        c_gt += above
        kk -= above
        c_ge = c_gt + int(h[b])
        exact = h[b] == kk
        top = shift
        passes += 1
    return T + kbase, c_ge, c_gt, passes, biggest


@pytest.mark.parametrize("minsub", [False, True])
def test_radix_select_threshold(minsub):
    rng = np.random.default_rng(0)
    for trial in range(200):
        n = int(rng.integers(1, 5000))
        k = int(rng.integers(1, min(n, 1024) + 1))
        if trial % 3 == 0:
            s = -(rng.random(n).astype(np.float32) * 2.2 + 0.3)       # -||q - x||^2 of nearby rows
        elif trial % 3 == 1:
            s = rng.standard_normal(n).astype(np.float32)
        else:
            s = -(np.round(rng.random(n) * 20) / 8).astype(np.float32)  # many exact ties
        keys = score_key(s)
        kth = int(np.sort(keys)[::-1][k - 1])
        T, c_ge, c_gt, _, _ = radix_select(keys, k, minsub)
        assert c_ge == int((keys >= T).sum()) and c_ge >= k
        if c_ge > k:   # T is the exact k-th key; ties at T are admitted in position order afterwards
            assert T == kth and c_gt == int((keys > T).sum())
        else:
            assert T <= kth


def test_min_subtraction_spreads_the_leading_digit():
    """Why NANN_TOPK_MINSUB exists: scores within a few binades share most of their top undecided
    bits, so the first histogram pass piles the keys into two or three bins (same-bin LDS atomics
    serialise); on key - min(key) the same pass is spread out and usually ends the search."""
    rng = np.random.default_rng(1)
    keys = score_key(-(rng.random(2500).astype(np.float32) * 2.2 + 0.3))
    _, _, _, passes_raw, big_raw = radix_select(keys, 128, False)
    _, _, _, passes_sub, big_sub = radix_select(keys, 128, True)
    assert big_raw[0] > 10 * big_sub[0] and passes_sub <= passes_raw


# ---- NANN_COMPACT: the visited set as an open-addressing table (vis_contains / vis_insert) ----------
class VisTable:
    """Each probe step of an insert is one atomic LDS operation; inserts of different lanes are
    interleaved at that granularity (generator per insert, scheduled in random order)."""

    def __init__(self, slots=64):
        self.v = [0] * slots
        self.slots = slots

    def hash(self, x):
        return ((x * 2654435761) & 0xFFFFFFFF) % self.slots

    def contains(self, x):
        h = self.hash(x)
        while True:
            cur = self.v[h]
            if cur == x + 1:
                return True
            if cur == 0:
                return False
            h = (h + 1) % self.slots

    def insert_steps(self, x, result, key):
        h = self.hash(x)
        while True:
            cur = self.v[h]
            yield
            if cur == 0:
                cur = self.v[h]          # atomicCAS(&V[h], 0, x + 1): returns the old value
                if cur == 0:
                    self.v[h] = x + 1
                    result[key] = True
                    return
                yield
            if cur == x + 1:
                result[key] = False
                return
            h = (h + 1) % self.slots


def test_visited_hash_set_has_one_winner_per_id():
    rng = np.random.default_rng(11)
    for _ in range(200):
        t = VisTable(64)
        before = set(int(v) for v in rng.integers(0, 40, size=int(rng.integers(0, 20))))
        for x in before:
            res = {}
            for _ in t.insert_steps(x, res, 0):
                pass
        xs = [int(v) for v in rng.integers(0, 40, size=int(rng.integers(1, 30)))]
        if len(before | set(xs)) > 48:
            continue
        res = {}
        gens = [t.insert_steps(x, res, i) for i, x in enumerate(xs)]
        live = list(range(len(gens)))
        while live:
            i = live[rng.integers(len(live))]
            try:
                next(gens[i])
            except StopIteration:
                live.remove(i)
        for x in set(xs):
            winners = [i for i, y in enumerate(xs) if y == x and res[i]]
            assert len(winners) == (0 if x in before else 1)
        assert all(t.contains(x) for x in before | set(xs))
        assert sum(1 for v in t.v if v) == len(before | set(xs))   # no id stored twice
