"""CPU: the request-batching front end with a fake backend: shapes of the
reference's signature, aggregation, per-request failures, launch failures."""
import threading
import time

import numpy as np
import pytest

from nann_amd import serving


def fake_backend(log):
    def run(seqs, level_topn):
        log.append(len(seqs))
        time.sleep(0.002)  # a launch
        lt = np.asarray(level_topn)
        k = int(lt[5]) if lt.ndim == 1 else int(lt[:, 5].max())  # mixed batch: rows as wide as the largest k
        first = seqs[:, 0, 0].astype(np.float32)          # request tag
        ids = (first[:, None] * 1000 + np.arange(k)[None, :]).astype(np.int64)
        status = (first < 0).astype(np.int32) * 4          # negative tag -> TopKV2 failure code
        return ids, status
    return run


def test_batches_and_routes_results():
    log = []
    srv = serving.BatchingServer(fake_backend(log), 50, 64, [8] * 5 + [5], max_batch=16, max_wait_us=20000)
    futs = []
    for i in range(40):
        seq = np.zeros(3200, np.float16); seq[0] = i + 1
        futs.append(srv.submit(seq.reshape(1, 3200)))      # reference shape [1, 3200]
    for i, f in enumerate(futs):
        out = f.result(5)
        assert out.shape == (1, 5) and out.dtype == np.int64
        assert out[0, 0] == (i + 1) * 1000                 # each caller gets ITS result
    srv.close()
    assert sum(log) == 40 and max(log) <= 16 and len(log) < 40  # aggregated, capped


def test_failed_request_fails_only_its_caller():
    srv = serving.BatchingServer(fake_backend([]), 50, 64, [8] * 5 + [5], max_batch=8, max_wait_us=20000)
    good = np.zeros(3200, np.float16); good[0] = 3
    bad = np.zeros(3200, np.float16); bad[0] = -1
    f1, f2 = srv.submit(good), srv.submit(bad)
    assert f1.result(5)[0, 0] == 3000
    with pytest.raises(serving.RequestFailed) as e:
        f2.result(5)
    assert e.value.status == 4
    with pytest.raises(ValueError):
        srv.submit(np.zeros(10, np.float16))
    srv.close()


def test_launch_failure_and_closed_loop():
    def boom(seqs, level_topn):
        raise RuntimeError("device lost")
    srv = serving.BatchingServer(boom, 50, 64, [8] * 5 + [5])
    with pytest.raises(RuntimeError):
        srv.predict(np.zeros(3200, np.float16), timeout=5)
    srv.close()
    srv = serving.BatchingServer(fake_backend([]), 50, 64, [8] * 5 + [5], max_batch=64, max_wait_us=500)
    req = np.ones(3200, np.float16)
    stats = serving.closed_loop(srv, lambda cid: req, n_clients=16, duration_s=0.3)
    srv.close()
    assert stats["requests"] > 16 and stats["failures"] == 0 and stats["mean_batch"] > 1.5
    assert stats["latency_us"]["p50"] >= 2000  # at least one launch


def test_level_topn_per_request_and_admission_control():
    """`level_topn` is a per-request feed of the reference's signature (build_opt_graph.py:75,151-159): requests with
    different values share a launch and each reply has ITS k; values beyond the server's are refused at submit; a full
    queue refuses (BlazeXlaOp: "waiting pool is full") and a request that waited past the deadline fails unsearched."""
    seen = []

    def backend(seqs, level_topn):
        seen.append(np.asarray(level_topn).copy())
        return fake_backend([])(seqs, level_topn)

    srv = serving.BatchingServer(backend, 50, 64, [8] * 5 + [6], max_batch=16, max_wait_us=30000)
    seq = np.zeros(3200, np.float16); seq[0] = 2
    fa, fb, fc = srv.submit(seq), srv.submit(seq, [4] * 5 + [3]), srv.submit(seq, [8, 8, 4, 4, 4, 6])
    assert fa.result(5).shape == (1, 6) and fb.result(5).shape == (1, 3) and fc.result(5).shape == (1, 6)
    assert fb.result(5)[0, 0] == 2000
    assert any(t.ndim == 2 and t.shape[1] == 6 and {tuple(r) for r in t} == {(8,) * 5 + (6,), (4,) * 5 + (3,), (8, 8, 4, 4, 4, 6)} for t in seen)
    with pytest.raises(ValueError):
        srv.submit(seq, [9] * 5 + [6])
    srv.close()
    slow = serving.BatchingServer(lambda s, t: (time.sleep(0.05), fake_backend([])(s, t))[1], 50, 64, [8] * 5 + [5],
                                  max_batch=1, max_wait_us=100, max_queue=2, deadline_ms=20.0)
    futs = [slow.submit(seq) for _ in range(12)]
    outcomes = []
    for f in futs:
        try:
            f.result(5); outcomes.append("ok")
        except serving.Overloaded as e:
            outcomes.append(str(e))
    slow.close()
    assert "ok" in outcomes and "waiting pool is full" in outcomes and slow.refused >= 1
    assert slow.expired >= 1 and "request waited too long" in outcomes
