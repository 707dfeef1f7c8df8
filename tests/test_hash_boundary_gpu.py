"""-m gpu: the visited set at its direct <-> tag boundary (bitmap_ops.cc:224-232).

A hash-set entry is (tag << 12) | position-in-piece with pieces of 4095 ids; up to 2^20 - 1 items the tag is the id
itself, from 2^20 items on it is cut from a bijection of the id space (nann_device.h, vis_key; plan_search's
bit_length(n_items)).  ADVICE r3's bug lived exactly here: id 2^20 - 1 at piece position 4095 encoded as the empty
value.  These tests put the LAST id of a shard of 2^20 - 1, 2^20 and 2^20 + 1 items at the last position of a full
piece of a level-0 round, copies of it at the head and the tail of the pieces behind, other boundary ids around it, and
require every plan to answer like the oracle's serial scan bit for bit (ids, scores, per-round counters)."""
import numpy as np
import pytest
import torch

from gpu_util import MODES, bits, cuda, require_gpu, traversal_mode

pytestmark = pytest.mark.gpu

PIECE = 4095
D = 64


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    require_gpu()


def _crafted_index(n, seed):
    """n items; nodes 0..15 are the 'core' (x = ((i + 1) / 256, 0, ...): with q[0] = 0 node i beats node i + 1), the rest
    the bulk (random rows far from the origin).  Level 1 leads entry nodes 0..3 to 8..12; the level-0 rows of the
    first frontier [0, 1, 2, 3, 8, 9, 10, 11] concatenate to a crafted list of 3 pieces + 100 ids; every other node
    has four pseudo-random neighbours."""
    rng = np.random.default_rng(seed)
    embs = (rng.standard_normal((n, D)) * 0.3).astype(np.float16)
    embs[:16] = 0
    embs[:16, 0] = (np.arange(16) + 1) / 256.0
    item_ids = (rng.permutation(n) + 1).astype(np.int64)
    # level 1: rows of the entry nodes only
    rows1 = {0: [8, 9], 1: [10, 11], 3: [12]}
    len1 = np.zeros(n, np.int64)
    for k, v in rows1.items():
        len1[k] = len(v)
    rs1 = np.concatenate([[0], np.cumsum(len1)]).astype(np.int64)
    nb1 = np.array([x for k in sorted(rows1) for x in rows1[k]], np.int32)
    # the crafted list of the first level-0 round
    total = 3 * PIECE + 100
    bulk = rng.integers(16, n, size=total).astype(np.int64)
    dup = rng.random(total) < 0.3                       # copies of earlier list entries
    src = (rng.random(total) * np.maximum(np.arange(total), 1)).astype(np.int64)
    for i in np.nonzero(dup)[0]:
        bulk[i] = bulk[src[i]]
    visited = rng.random(total) < 0.02                  # ids the marks already hold
    bulk[visited] = rng.choice([0, 1, 2, 3, 8, 9, 10, 11], size=int(visited.sum()))
    last = n - 1
    L = bulk
    L[PIECE - 1] = last            # the last position of a full piece: (tag << 12) | 4095
    L[PIECE] = last                # its copy heads the next piece ("visited before this piece")
    L[PIECE - 2] = n - 2
    L[PIECE + 1] = n - 2
    L[2 * PIECE - 1] = last        # and again at the last position of piece 1
    L[2 * PIECE] = n - 3
    L[2 * PIECE - 2] = (1 << 20) - 1 if n > (1 << 20) - 1 else n - 4   # id 2^20 - 1 where the shard has it
    L[total - 1] = last
    L[0] = n - 5
    cuts = [0, 1000, 4000, 4095, 4097, 8097, 8190, 10000, total]       # eight rows, two of them across piece seams
    frontier = [0, 1, 2, 3, 8, 9, 10, 11]
    len0 = np.full(n, 4, np.int64)
    for j, node in enumerate(frontier):
        len0[node] = cuts[j + 1] - cuts[j]
    rs0 = np.concatenate([[0], np.cumsum(len0)]).astype(np.int64)
    nb0 = np.empty(int(rs0[-1]), np.int32)
    i = np.arange(n, dtype=np.int64)
    default = np.stack([(i * 7 + 1) % n, (i * 13 + 5) % n, (i * 101 + 17) % n, (i * 31 + 3) % n], 1).astype(np.int32)
    plain = len0 == 4
    pos = rs0[:-1][plain]
    nb0[(pos[:, None] + np.arange(4)[None, :]).ravel()] = default[plain].ravel()
    for j, node in enumerate(frontier):
        nb0[rs0[node]:rs0[node + 1]] = L[cuts[j]:cuts[j + 1]]
    enter = np.arange(8, dtype=np.int32)
    return dict(item_embs=embs, item_ids=item_ids, nb_values=[nb0, nb1], nb_row_splits=[rs0, rs1], enter_points=enter,
                crafted=L.copy())


@pytest.mark.parametrize("n", [(1 << 20) - 1, 1 << 20, (1 << 20) + 1])
def test_last_id_at_the_last_position_of_a_full_piece(oracle, n):
    from nann_amd import ops, retrieval
    g = _crafted_index(n, seed=n & 0xffff)
    oix = oracle.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    dix = retrieval.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    rng = np.random.default_rng(3)
    q = (rng.standard_normal((6, D)) * 0.05).astype(np.float32)
    q[:, 0] = 0.0  # the core nodes keep their order
    topn = [4, 8, 8, 8, 8, 10]
    exp = oracle.search_batch(oix, oracle.Scorer("l2", D, oracle.EMB_F16), q, topn)
    assert (exp[0] == 0).all(), exp[0]
    ctr = exp[4].reshape(len(q), 3, -1)
    # round 2 (the first level-0 round) walked the crafted list: 8 rows, 3 pieces + 100 gathered ids
    assert (ctr[:, 0, 2] == 8).all() and (ctr[:, 1, 2] == 3 * PIECE + 100).all(), ctr[0]
    # what the serial scan keeps of it, from the stand-alone oracle op on the same list and marks
    bm = np.zeros((n + 31) // 32, np.int32)
    for m in (0, 1, 2, 3, 8, 9, 10, 11):
        bm[m >> 5] |= np.int32(1 << (m & 31)) if (m & 31) < 31 else np.int32(-(1 << 31))
    rc, _, kept, _ = oracle.bitmap_ref_difference(g["crafted"].astype(np.int32), [0, len(g["crafted"])], bm)
    assert rc == 0 and (ctr[:, 2, 2] == len(kept)).all()
    assert (n - 1) in kept.tolist() and kept.tolist().count(n - 1) == 1
    sc = ops.Scorer("l2", D, torch.float16)
    for mode in MODES + ["auto"]:
        with traversal_mode(mode):
            r = retrieval.search(dix, sc, cuda(q), topn)
            torch.cuda.synchronize()
        assert (r.status.cpu().numpy() == 0).all(), (mode, r.status.cpu().numpy())
        assert (r.index.cpu().numpy() == exp[3]).all(), mode
        assert (r.item_ids.cpu().numpy() == exp[1]).all(), mode
        assert (bits(r.scores.cpu().numpy()) == bits(exp[2])).all(), mode
        assert (r.counters.cpu().numpy() == exp[4]).all(), mode


def _regular_random_index(n, degree, n_enter, seed):
    """Every level-0 row holds `degree` random neighbours (so nearly every gathered id is new), level 1 leads each of
    the n_enter entry nodes to 8 random nodes."""
    rng = np.random.default_rng(seed)
    embs = (rng.standard_normal((n, D)) * 0.3).astype(np.float16)
    item_ids = (rng.permutation(n) + 1).astype(np.int64)
    nb0 = rng.integers(0, n, size=n * degree).astype(np.int32)
    rs0 = (np.arange(n + 1, dtype=np.int64) * degree)
    len1 = np.zeros(n, np.int64)
    len1[:n_enter] = 8
    rs1 = np.concatenate([[0], np.cumsum(len1)]).astype(np.int64)
    nb1 = rng.integers(0, n, size=int(rs1[-1])).astype(np.int32)
    return dict(item_embs=embs, item_ids=item_ids, nb_values=[nb0, nb1], nb_row_splits=[rs0, rs1],
                enter_points=np.arange(n_enter, dtype=np.int32))


def _spread(g, n_items):
    """The same graph with node k renamed k * (n_items // n): an id space of n_items in which only every stride-th id has a
    row and neighbours (ids beyond 2^20 put the hash set into its tag form)."""
    n = len(g["item_ids"])
    stride = n_items // n
    at = np.arange(n, dtype=np.int64) * stride
    embs = np.zeros((n_items, D), np.float16)
    embs[at] = g["item_embs"]
    out = dict(item_embs=embs, item_ids=np.arange(1, n_items + 1, dtype=np.int64), nb_values=[], nb_row_splits=[],
               enter_points=(g["enter_points"].astype(np.int64) * stride).astype(np.int32))
    for l in (0, 1):
        lens = np.zeros(n_items, np.int64)
        lens[at] = np.diff(g["nb_row_splits"][l])
        out["nb_row_splits"].append(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
        out["nb_values"].append((g["nb_values"][l].astype(np.int64) * stride).astype(np.int32))
    return out


def test_a_set_that_fills_up_cuts_its_pieces(oracle):
    """Round 5.  A piece of the insert loop may add as many ids as it has positions (4095); the test `count + piece >
    slots - 64` therefore gave a query up with its set at 12.2 k of 16.3 k entries.  Now a piece that could pass the capacity
    is cut to the room left.  On degree-64 random graphs small enough that a round's list repeats itself: (1) beams of 128
    over 20 k items start the second full piece of the last round with ~12.6 k ids in the 16K-slot set and end at ~14.1 k:
    no query is handed to the bitmap kernel, the answer is the oracle's bit for bit; (2) beams of 200 (~17.1 k ids) do
    overflow: every query is rerun, same parity; (3, 4) the same on the 32K-slot set (31.4 k of 32.7 k; 43 k); (5, 6) cases 1
    and 3 again with the nodes renamed into an id space beyond 2^20, where the set's entries are tags."""
    from nann_amd import ops, retrieval
    rng = np.random.default_rng(5)
    q = (rng.standard_normal((12, D)) * 0.3).astype(np.float32)
    sc = ops.Scorer("l2", D, torch.float16)
    for mode, n, ef, want_reruns, n_items in (("lds_hash", 20_000, 128, False, 0), ("lds_hash", 20_000, 200, True, 0),
                                              ("lds_hash32", 40_000, 320, False, 0), ("lds_hash32", 80_000, 320, True, 0),
                                              ("lds_hash", 20_000, 128, False, 1_300_000),    # the same with 21-bit ids: tag entries
                                              ("lds_hash32", 40_000, 320, False, 2_400_000)):
        g = _regular_random_index(n, 64, 320, seed=77)
        if n_items:
            g = _spread(g, n_items)
        oix = oracle.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
        dix = retrieval.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
        topn = [ef] * 5 + [100]
        exp = oracle.search_batch(oix, oracle.Scorer("l2", D, oracle.EMB_F16), q, topn, n_threads=8)
        assert (exp[0] == 0).all(), exp[0]
        ctr = exp[4].reshape(len(q), 3, -1)
        visited = ef + ctr[:, 2, 2:5].sum(1)  # marks of the level + new nodes of its three rounds
        cap = (16384 if mode == "lds_hash" else 32768) - 64
        if want_reruns:
            assert (visited > cap).all(), (mode, ef, visited)
        else:
            # the last round's list is >= 2 full pieces and the set passes `capacity - a piece` before the last full one
            assert (ctr[:, 1, 4] >= 2 * PIECE).all()
            before_last_round = ef + ctr[:, 2, 2:4].sum(1)
            n_full = ctr[:, 1, 4] // PIECE
            at_last_full_piece = before_last_round + ctr[:, 2, 4] * (n_full - 1) * PIECE // ctr[:, 1, 4]  # (new ids spread evenly)
            assert (at_last_full_piece > cap - PIECE + 100).all() and (visited < cap - 512).all(), (mode, ef, visited, at_last_full_piece)
        r = retrieval.search(dix, sc, cuda(q), topn, options=retrieval.search_options(traversal=mode))
        torch.cuda.synchronize()
        assert r.plan["visited_set"] == mode, r.plan
        if n_items:  # tag entries: at 87-96 % load a probe sequence may outrun the 62 steps a tag can name -> that query is rerun
            print("tag entries:", mode, ef, "reruns", r.reruns(), "of", len(q))  # (1 and 12 of 12 when written; plan_search keeps tag sets below 0.80)
        else:
            assert r.reruns() == (len(q) if want_reruns else 0), (mode, ef, r.reruns(), visited)
        assert (r.status.cpu().numpy() == 0).all()
        assert (r.index.cpu().numpy() == exp[3]).all(), (mode, ef)
        assert (r.item_ids.cpu().numpy() == exp[1]).all(), (mode, ef)
        assert (bits(r.scores.cpu().numpy()) == bits(exp[2])).all(), (mode, ef)
        assert (r.counters.cpu().numpy() == exp[4]).all(), (mode, ef)


def test_table_beyond_4_gib_takes_the_wide_row_address(oracle):
    """Round 5's scoring loop forms a row's address with 32-bit arithmetic on a scalar base when the whole table sits below
    4 GiB and ids below 2^24; above, the 64-bit form runs.  9 M x 256-d f16 = 4.6 GB, of which only a subgraph of 20 k nodes
    in ten runs over the whole id range -- 40 % of them behind the 4 GiB mark (id 8 388 608) -- has neighbours and non-zero rows
    (the host copy is calloc'ed: untouched pages never become resident; the device copy is a zeroed tensor + those rows).
    Every plan answers like the oracle bit for bit; the ids also need 24-bit tags in the hash set."""
    from nann_amd import ops, retrieval
    n, d, m = 9_000_000, 256, 20_000
    rng = np.random.default_rng(11)
    mark = (1 << 32) // (2 * d)  # the first row that starts at or behind 4 GiB
    # ten runs of 2 000 consecutive ids (the host array's pages are 2 MB: scattered ids would touch all of them), one of
    # them across the mark, four behind it
    bases = [5_000, 1_000_000, 2_100_000, 4_200_000, 6_300_000, mark - 600, 8_500_000, 8_700_000, 8_900_000, n - 2_000]
    nodes = np.concatenate([np.arange(b, b + m // len(bases)) for b in bases]).astype(np.int64)
    assert len(nodes) == m and (nodes >= mark).sum() > m // 3
    rows = (rng.standard_normal((m, d)) * 0.3).astype(np.float16)
    embs = np.zeros((n, d), np.float16)
    embs[nodes] = rows
    dev_embs = torch.zeros((n, d), dtype=torch.float16, device="cuda")
    dev_embs[torch.as_tensor(nodes).cuda()] = torch.as_tensor(rows).cuda()
    item_ids = np.arange(1, n + 1, dtype=np.int64)
    deg = 24
    len0 = np.zeros(n, np.int64)
    len0[nodes] = deg
    rs0 = np.concatenate([[0], np.cumsum(len0)]).astype(np.int64)
    nb0 = nodes[rng.integers(0, m, size=m * deg)].astype(np.int32)
    n_enter = 64
    len1 = np.zeros(n, np.int64)
    len1[nodes[:n_enter]] = 8
    rs1 = np.concatenate([[0], np.cumsum(len1)]).astype(np.int64)
    nb1 = nodes[rng.integers(0, m, size=n_enter * 8)].astype(np.int32)
    enter = nodes[:n_enter].astype(np.int32)
    oix = oracle.Index(embs, item_ids, [nb0, nb1], [rs0, rs1], enter)
    dix = retrieval.Index(dev_embs, item_ids, [nb0, nb1], [rs0, rs1], enter)
    q = (rng.standard_normal((16, d)) * 0.3).astype(np.float32)
    topn = [32] * 5 + [50]
    exp = oracle.search_batch(oix, oracle.Scorer("l2", d, oracle.EMB_F16), q, topn, n_threads=8)
    assert (exp[0] == 0).all(), exp[0]
    assert (exp[3] >= mark).sum() > exp[3].size // 10  # results from behind the 4 GiB mark
    sc = ops.Scorer("l2", d, torch.float16)
    for mode in MODES + ["auto"]:
        r = retrieval.search(dix, sc, cuda(q), topn, options=retrieval.search_options(traversal=mode))
        torch.cuda.synchronize()
        assert (r.status.cpu().numpy() == 0).all(), (mode, r.status.cpu().numpy())
        assert (r.index.cpu().numpy() == exp[3]).all(), mode
        assert (r.item_ids.cpu().numpy() == exp[1]).all(), mode
        assert (bits(r.scores.cpu().numpy()) == bits(exp[2])).all(), mode
        assert (r.counters.cpu().numpy() == exp[4]).all(), mode
    del dix, dev_embs
    torch.cuda.empty_cache()
