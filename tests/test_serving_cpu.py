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
        k = level_topn[5]
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
