"""CPU: executable statements of two workgroup algorithms of nann_amd/csrc/nann_device.h, checked
against the reference semantics they must reproduce.  They are models (numpy / plain Python with
the atomic operations of a phase applied in random lane order), kept next to the parity tests so
that a change to the device code's logic can be tried here first:

* wg_filter_chunk -- BitmapRefDifference's ordered first-occurrence scan
  (UO/bitmap_op/bitmap_ops.cc:224-232) done 2048 ids at a time with set-bits-then-arbitrate;
* wg_expand_hash's positional hash set: the same scan with the visited set as an open-addressing
  table of (id << PB | position) entries, CAS to claim, ds_min to keep the first occurrence;
* wg_topk_impl's radix search for the k-th largest key on key - min(key) (the shipped form) and on
  raw keys with the common prefix skipped (the form it replaced);
* l2_rows8_reduce_scatter -- the L2 scorer's xor butterfly over the 16 lanes of a DPP row for eight rows at once as a
  reduce-scatter (round 5): instruction by instruction as the asm block issues them, against the butterfly, bit for bit."""
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


@pytest.mark.parametrize("model", [filter_with_preread])
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


# ---- wg_expand_hash: the visited set as a positional open-addressing table ---------------------
class PosHashSet:
    """Model of the LDS table: every LDS operation of an insert (read, CAS, min) is one atomic step;
    the inserts of a piece are interleaved at that granularity in random order.

    A slot holds (tag << pos_bits) | pos with tag = (remainder << step_bits) | probe step (nann_device.h, vis_key):
    a bijection p of the id space (odd multiply, fold, odd multiply), its top bits the home slot, its low 14 bits
    stored (they cover the remainder and drive the probe stride) -- so (slot, tag) names exactly one id (`decode`)."""
    ODD, ODD2 = 2654435761, 0x85EBCA6B

    def __init__(self, slots, pos_bits, id_bits, step_bits=6, tag_bits=14):
        self.v = [EMPTY] * slots
        self.slots, self.pb, self.ib, self.sb, self.tb = slots, pos_bits, id_bits, step_bits, tag_bits
        self.hb = slots.bit_length() - 1
        assert self.hb < self.ib <= self.hb + tag_bits and tag_bits + step_bits + pos_bits <= 32

    def key(self, x):
        m = (1 << self.ib) - 1
        p = (x * self.ODD) & m
        p ^= p >> (self.ib >> 1)
        p = (p * self.ODD2) & m
        return p >> (self.ib - self.hb), p & ((1 << self.tb) - 1)

    def stride(self, t):  # odd: the probe sequence h, h + s, h + 2s, ... visits every slot of the 2^n table
        return (((t * 0x85EBCA6B) & 0xFFFFFFFF) >> (32 - self.hb)) | 1

    def decode(self, slot, value):
        """the id an entry stands for: undo the probe steps (the stride is a function of stored bits), then the
        three bijections"""
        tag = value >> self.pb
        tt, k = tag >> self.sb, tag & ((1 << self.sb) - 1)
        h0 = (slot - k * self.stride(tt)) % self.slots
        m = (1 << self.ib) - 1
        rb = self.ib - self.hb
        p = (((h0 << rb) | (tt & ((1 << rb) - 1))) * pow(self.ODD2, -1, 1 << self.ib)) & m
        q, sh = p, self.ib >> 1
        for _ in range(4):                                    # y = x ^ (x >> sh)  ->  x
            q = p ^ (q >> sh)
        return (q * pow(self.ODD, -1, 1 << self.ib)) & m

    def insert_steps(self, x, pos, slot_out, key):
        h, r = self.key(x)
        st = self.stride(r)
        k = 0
        while True:
            val = (((r << self.sb) | k) << self.pb) | pos
            cur = self.v[h]                                   # ds_read
            yield
            if cur == EMPTY:
                cur = self.v[h]                               # ds_cmpst_rtn: returns the old value
                if cur == EMPTY:
                    self.v[h] = val
                yield
            if cur == EMPTY or (cur >> self.pb) == (val >> self.pb):
                if cur != EMPTY:
                    self.v[h] = min(self.v[h], val)           # ds_min_u32
                    yield
                slot_out[key] = (h, val)
                return
            assert k + 2 < (1 << self.sb), "probe sequence longer than the tag can name (the kernel hands the query back)"
            h = (h + st) % self.slots
            k += 1

    def filter_piece(self, xs, rng):
        """One piece (len(xs) <= 2^pb - 1): insert all, barrier, keep iff own value survived; the
        keeper resets the position field to 0 ("visited before")."""
        slot = {}
        gens = [self.insert_steps(int(x), p + 1, slot, p) for p, x in enumerate(xs)]
        live = list(range(len(gens)))
        while live:
            i = live[rng.integers(len(live))]
            try:
                next(gens[i])
            except StopIteration:
                live.remove(i)
        keep = []
        for p in rng.permutation(len(xs)):                    # check + reset race freely after the barrier
            h, val = slot[p]
            if self.v[h] == val:
                self.v[h] = val & ~((1 << self.pb) - 1)
                keep.append(p)
        return [int(xs[p]) for p in sorted(keep)]             # ordered compaction by position


@pytest.mark.parametrize("pos_bits,slots,id_bits", [(12, 16384, 21), (12, 16384, 22), (12, 32768, 27), (6, 128, 12)])
def test_positional_hash_set_equals_serial_scan(pos_bits, slots, id_bits):
    rng = np.random.default_rng(21)
    piece = (1 << pos_bits) - 1
    for trial in range(60 if pos_bits == 6 else 8):
        hi = int(rng.choice([8, 40, 3000, 1_000_000, 1 << 27]))
        hi = min(hi, 1 << id_bits)
        table, visited, expect_all, got_all = PosHashSet(slots, pos_bits, id_bits, tag_bits=14 if slots > 128 else 8), set(), [], []
        cap = slots - 64 if slots > 128 else slots // 2
        for _ in range(int(rng.integers(1, 5))):              # several pieces (rounds) against one set
            n = int(rng.integers(1, piece + 1))
            xs = rng.integers(0, hi, size=n)
            if len(visited | set(int(v) for v in xs)) > cap:
                break
            expect_all += serial_scan(xs, visited)
            got_all += table.filter_piece(xs, rng)
        assert got_all == expect_all
        stored = [table.decode(i, v) for i, v in enumerate(table.v) if v != EMPTY]
        assert sorted(stored) == sorted(visited)               # every id once (decoded from slot + tag), positions reset
        assert all((v & ((1 << pos_bits) - 1)) == 0 for v in table.v if v != EMPTY)


def test_hash_set_tag_is_a_bijection():
    """(home slot, remainder) <-> id for every id of a small id space, and distinct strides keep sequences apart."""
    t = PosHashSet(128, 6, 12, tag_bits=8)
    seen = set()
    for x in range(1 << 12):
        h, r = t.key(x)
        assert (h, r) not in seen
        seen.add((h, r))
        for k in (0, 1, 5, 62):
            slot = (h + k * t.stride(r)) % t.slots
            assert t.decode(slot, (((r << t.sb) | k) << t.pb) | 3) == x


# ---------------------------------------------------------------------------------------------------------------------
# l2_rows8_reduce_scatter (nann_device.h): v_cndmask_b32 dst = vcc ? src1 : src0; v_add_f32_dpp dst = dpp(src0) + src1.

def _dpp(src, kind):
    lanes = np.arange(len(src))
    if kind == "xor1":
        return src[(lanes & ~3) | np.array([1, 0, 3, 2])[lanes & 3]]
    if kind == "xor2":
        return src[(lanes & ~3) | np.array([2, 3, 0, 1])[lanes & 3]]
    if kind == "half_mirror":
        return src[(lanes & ~7) | (7 - (lanes & 7))]
    if kind == "mirror":
        return src[(lanes & ~15) | (15 - (lanes & 15))]
    raise ValueError(kind)


def _lane_mask(m32, n_lanes=64):
    m = m32 | (m32 << 32)
    return np.array([(m >> l) & 1 for l in range(n_lanes)], bool)


def _butterfly(x):
    for kind in ("xor1", "xor2", "half_mirror", "mirror"):
        x = (x + _dpp(x, kind)).astype(np.float32)
    return x


def _rows8_reduce_scatter(p):
    """p[u][lane]: the asm block of l2_rows8_reduce_scatter, one numpy statement per instruction."""
    a = [p[u].copy() for u in range(8)]
    t = [None] * 4
    vcc = _lane_mask(0x5A5A5A5A)  # b0 ^ b2
    for k in range(4):
        t[k] = np.where(vcc, a[2 * k + 1], a[2 * k])          # v_cndmask t, a_even, a_odd   (kept)
        a[2 * k] = np.where(vcc, a[2 * k], a[2 * k + 1])      # v_cndmask a_even, a_odd, a_even (sent)
    for k in range(4):
        t[k] = (_dpp(a[2 * k], "xor1") + t[k]).astype(np.float32)
    vcc = _lane_mask(0x3C3C3C3C)  # b1 ^ b2
    a[1] = np.where(vcc, t[0], t[1])
    a[3] = np.where(vcc, t[2], t[3])
    a[0] = np.where(vcc, t[1], t[0])
    a[2] = np.where(vcc, t[3], t[2])
    a[0] = (_dpp(a[1], "xor2") + a[0]).astype(np.float32)
    a[2] = (_dpp(a[3], "xor2") + a[2]).astype(np.float32)
    vcc = _lane_mask(0x0FF00FF0)  # b2 ^ b3
    t[0] = np.where(vcc, a[0], a[2])
    t[1] = np.where(vcc, a[2], a[0])
    t[1] = (_dpp(t[0], "half_mirror") + t[1]).astype(np.float32)
    return (_dpp(t[1], "mirror") + t[1]).astype(np.float32)


def test_rows8_reduce_scatter_is_the_butterfly_bit_for_bit():
    rng = np.random.default_rng(7)
    for trial in range(20):
        p = (rng.standard_normal((8, 64)) * 10.0 ** rng.integers(-3, 4)).astype(np.float32) ** 2
        got = _rows8_reduce_scatter(p)
        ref = [_butterfly(p[u]) for u in range(8)]
        for lane in range(64):
            s = lane & 15
            b0, b1, b2, b3 = s & 1, (s >> 1) & 1, (s >> 2) & 1, (s >> 3) & 1
            u = ((b2 ^ b3) << 2) | ((b1 ^ b2) << 1) | (b0 ^ b2)
            assert got[lane].view(np.uint32) == ref[u][lane].view(np.uint32), (trial, lane, u)
            if s < 8:  # the lanes that store: wg_score_l2_part's out_at
                assert u == (s & 7) ^ (3 if s & 4 else 0)
        assert sorted(((l & 7) ^ (3 if l & 4 else 0)) for l in range(8)) == list(range(8))
