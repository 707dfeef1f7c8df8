"""Lifecycle of the scorers' pre-projected tables (nann_amd/csrc/host/nann_projcache.h) under concurrency, on the CPU:
the cache template nann_hip.hip instantiates with HIP runs here against a mock device (libnann_host.so,
nann_projcache_c.cpp) that counts every violation of: a table freed under a launch that reads it, freed while a
thread holds it between acquire and launch (the round-3 use-after-free), a freed table handed out again,
device memory beyond capacity.  3 indices x 4 threads on one scorer, with prepare / release pins and index
destruction mixed in."""
import ctypes as C

import pytest

from nann_amd import index_build


def _stress(threads, indices, iters, capacity_tables, seed=1):
    L = C.CDLL(index_build.build_host_lib())
    out = (C.c_int64 * 6)()
    rc = L.nann_projcache_stress(C.c_int32(threads), C.c_int32(indices), C.c_int32(iters), C.c_int64(1 << 20),
                                 C.c_int32(capacity_tables), C.c_uint64(seed), out)
    assert rc == 0
    return dict(zip(("violations", "builds", "frees", "peak_bytes", "no_table", "leaked_bytes"), list(out)))


@pytest.mark.parametrize("capacity_tables", [8, 3, 2])
def test_three_indices_four_threads(capacity_tables):
    """capacity 8: room for every retired table; 3: evictions must actually free memory for the third index; 2: some
    searches find no room and are served without a table (the embedding-row kernels) -- never an error."""
    r = _stress(4, 3, 1500, capacity_tables)
    assert r["violations"] == 0, r
    assert r["leaked_bytes"] == 0 and r["builds"] == r["frees"], r
    assert r["peak_bytes"] <= capacity_tables << 20, r
    assert r["builds"] >= 3, r
    if capacity_tables >= 8:
        assert r["no_table"] == 0, r  # with room, every search gets its table


def test_single_index_is_built_once():
    r = _stress(4, 1, 400, 4, seed=7)
    # one index, nothing to evict: built once, plus once after each release of a pin (a released table is retired at
    # once: every 16th search of worker 0) and each destruction of the index (every 64th)
    assert r["violations"] == 0 and r["no_table"] == 0, r
    assert r["builds"] <= 1 + 400 // 16 + 400 // 64 + 1, r


def test_more_indices_than_kept():
    r = _stress(6, 5, 800, 6, seed=3)
    assert r["violations"] == 0 and r["leaked_bytes"] == 0, r


def test_a_slow_build_does_not_stall_the_other_tables():
    """acquire() used to hold the cache's mutex across hipMalloc + the pre-projection kernel + a stream wait, so
    the first search of a new (scorer, index) pair stalled hits on every other index's table.  One thread builds index 1
    for 300 ms; hits on index 2 go on meanwhile (each well under the build's duration), and a second caller of index 1
    waits for the first's table instead of building another."""
    L = C.CDLL(index_build.build_host_lib())
    out = (C.c_int64 * 4)()
    assert L.nann_projcache_slow_build(C.c_int32(300), C.c_int32(2000), out) == 0
    hits_during, worst_us, builds1, bad = list(out)
    assert bad == 0 and builds1 == 1, list(out)
    assert hits_during >= 100, list(out)          # (under the old lock: 0 -- the first hit would return after the build)
    assert worst_us < 100_000, list(out)          # no hit waited for the 300 ms build


def test_a_call_without_preprojection_never_gets_a_cached_table():
    """nann_search_options.preprojection = 0 means "this call runs without tables" whatever earlier calls left
    cached for the pair (acquire() used to look for a hit before it looked at `enabled`)."""
    L = C.CDLL(index_build.build_host_lib())
    out = (C.c_int64 * 3)()
    assert L.nann_projcache_disabled_call(out) == 0
    builds, got, bad = list(out)
    assert builds == 1 and got == 0 and bad == 0, list(out)
