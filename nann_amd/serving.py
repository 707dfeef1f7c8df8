"""Request batching in front of the fused traversal.

The reference serves ONE query per `Session::Run` and gets concurrency from
sessions x virtual GPUs x MPS (blaze-benchmark model.cc:192-235, README.md:175-177).  On the
MI355X a launch wants hundreds of queries (one workgroup per CU), so the front end keeps the
reference's request signature --

    comm_seq  f16[1, seq_len * emb_dim]     (build_opt_graph.py:75-79)
    level_topn i32[6]
    -> top_k  i64[1, level_topn[5]]         (:151-159)

-- and aggregates concurrent requests into one `nann_search` launch: a dispatcher thread drains
the queue up to `max_batch` requests or `max_wait_us` after the first one, whichever comes
first, runs the batch, and completes each caller's future.  A request the reference would have
failed (TopKV2 with n < k, ...) raises the same InvalidArgument in its caller only.

The backend is any callable `(comm_seq f16[B, seq_len, d], level_topn) -> (top_k i64[B, k],
status i32[B])`; `device_backend()` builds the real one.
"""
import os
import queue
import threading
import time
from concurrent.futures import Future

import numpy as np


class RequestFailed(RuntimeError):
    def __init__(self, status):
        super().__init__(f"request failed with nann_status {status}")
        self.status = status


class Overloaded(RuntimeError):
    """admission control, as BlazeXlaOp has it (blaze_xla_kernel.cc:229-258): 'waiting pool is full' / 'blaze wait too long'"""


class BatchingServer:
    """level_topn: the server's DEFAULT and per-entry MAXIMUM.  A request may carry its own `level_topn` -- the reference
    feeds it per request (build_opt_graph.py:75,151-159) --, each entry within the server's; a batch whose requests all
    use one value is launched uniformly, a mixed batch hands the backend an i32[B, 6] array (nann_search_v).
    max_queue / deadline_ms: refuse a request while that many wait / fail one that waited longer, without searching."""

    def __init__(self, backend, seq_len, emb_dim, level_topn, max_batch=1024, max_wait_us=200, max_queue=0, deadline_ms=0.0):
        self.backend, self.seq_len, self.emb_dim = backend, seq_len, emb_dim
        self.level_topn = [int(x) for x in level_topn]
        self.max_batch, self.max_wait = int(max_batch), max_wait_us * 1e-6
        self.max_queue, self.deadline = int(max_queue), deadline_ms * 1e-3
        self.refused = self.expired = 0
        self._q = queue.Queue()
        self._stop = threading.Event()
        self.batches = 0
        self.requests = 0
        self._thread = threading.Thread(target=self._run, name="nann-dispatch", daemon=True)
        self._thread.start()

    def submit(self, comm_seq, level_topn=None):
        """comm_seq: f16 array of seq_len*emb_dim elements (any shape); level_topn: this request's i32[6] (default: the
        server's).  -> Future of i64[1, level_topn[5]]."""
        a = np.asarray(comm_seq, dtype=np.float16).reshape(-1)
        if a.size != self.seq_len * self.emb_dim:
            raise ValueError(f"comm_seq must hold {self.seq_len}x{self.emb_dim} values, got {a.size}")
        t = self.level_topn if level_topn is None else [int(x) for x in level_topn]
        if len(t) != 6 or any(v < 0 or v > m for v, m in zip(t, self.level_topn)):
            raise ValueError(f"level_topn must be six values within the server's {self.level_topn}, got {t}")
        f = Future()
        if self.max_queue and self._q.qsize() >= self.max_queue:
            self.refused += 1
            f.set_exception(Overloaded("waiting pool is full"))
            return f
        self._q.put((a, f, t, time.perf_counter()))
        return f

    def predict(self, comm_seq, level_topn=None, timeout=None):
        """Blocking call with the reference's request/response shapes."""
        return self.submit(comm_seq, level_topn).result(timeout)

    def close(self):
        self._stop.set()
        self._q.put(None)
        self._thread.join()

    def _run(self):
        while not self._stop.is_set():
            first = self._q.get()
            if first is None:
                break
            batch = [first]
            deadline = time.perf_counter() + self.max_wait
            while len(batch) < self.max_batch:
                left = deadline - time.perf_counter()
                try:
                    item = self._q.get(timeout=max(left, 0)) if left > 0 else self._q.get_nowait()
                except queue.Empty:
                    break
                if item is None:
                    self._stop.set()
                    break
                batch.append(item)
            if self.deadline > 0:
                now, kept = time.perf_counter(), []
                for item in batch:
                    if now - item[3] > self.deadline:
                        self.expired += 1
                        item[1].set_exception(Overloaded("request waited too long"))
                    else:
                        kept.append(item)
                batch = kept
                if not batch:
                    continue
            seqs = np.stack([b[0] for b in batch]).reshape(len(batch), self.seq_len, self.emb_dim)
            topns = [b[2] for b in batch]
            uniform = all(t == topns[0] for t in topns)
            try:
                ids, status = self.backend(seqs, topns[0] if uniform else np.asarray(topns, np.int32))
                ids, status = np.asarray(ids), np.asarray(status)
                for i, (_, fut, t, _) in enumerate(batch):
                    if status[i]:
                        fut.set_exception(RequestFailed(int(status[i])))
                    else:
                        fut.set_result(ids[i:i + 1, :t[5]].copy())
            except Exception as e:  # a failed launch fails every request of the batch
                for item in batch:
                    if not item[1].done():
                        item[1].set_exception(e)
            self.batches += 1
            self.requests += len(batch)


def device_backend(index, scorer):
    """The real backend: user_seq_mean + fused traversal on the current GPU."""
    import torch
    from . import ops, retrieval

    def run(seqs, level_topn):
        q = ops.user_seq_mean(torch.as_tensor(seqs).to(index.device))
        r = retrieval.search(index, scorer, q, level_topn, want_counters=False)
        torch.cuda.synchronize()
        return r.item_ids.cpu().numpy(), r.status.cpu().numpy()

    return run


def closed_loop(server, make_request, n_clients, duration_s):
    """Closed-loop load (blaze-benchmark's consumer threads, predict_request_consumer.cc:17-53):
    every client issues its next request when the previous one returns.  Returns throughput and
    latency percentiles in microseconds."""
    lat, failures = [], [0]
    lock = threading.Lock()
    stop = time.perf_counter() + duration_s

    def client(cid):
        my = []
        while time.perf_counter() < stop:
            t0 = time.perf_counter()
            try:
                server.predict(make_request(cid))
            except RequestFailed:
                with lock:
                    failures[0] += 1
            my.append((time.perf_counter() - t0) * 1e6)
        with lock:
            lat.extend(my)

    threads = [threading.Thread(target=client, args=(i,)) for i in range(n_clients)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    a = np.sort(np.asarray(lat)) if lat else np.zeros(1)
    pct = lambda p: float(a[min(len(a) - 1, int(p * len(a)))])
    return {"requests": len(lat), "failures": failures[0], "throughput_qps": len(lat) / wall,
            "latency_us": {"p50": pct(0.5), "p90": pct(0.9), "p99": pct(0.99), "max": float(a[-1])},
            "mean_batch": server.requests / max(server.batches, 1)}


# ---------------------------------------------------------------------------------------------
# the C++ twin: csrc/host/nann_serve.cpp, a serving host over the C ABI alone
_SERVE_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "host", "nann_serve.cpp")
_SERVE_BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "nann_serve")


def build_serve_host(force=False):
    """g++ the C++ serving host (no HIP headers: it only sees include/nann_hip.h) against the in-tree
    libnann_hip.so -> nann_amd/_build/nann_serve."""
    import subprocess
    from . import build as _build
    lib = _build.build()
    if (force or not os.path.exists(_SERVE_BIN) or os.path.getmtime(_SERVE_SRC) > os.path.getmtime(_SERVE_BIN)
            or os.path.getmtime(lib) > os.path.getmtime(_SERVE_BIN)):
        inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-mavx2", "-mf16c", "-I", inc, _SERVE_SRC, "-o", _SERVE_BIN,
                               "-L", os.path.dirname(lib), "-lnann_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib",
                               "-Wl,--allow-shlib-undefined"])
    return _SERVE_BIN


def run_serve_host(index_dir, item_embs_dir, dim, clients=64, seconds=3.0, max_batch=256, max_wait_us=200, ef=128,
                   topk=200, seq_len=50, model_dir=None, probe_out=None, lanes=2, client_threads=0, mixed_topn=False,
                   probe_topn=None, max_queue=0, deadline_ms=0.0):
    """Run the C++ host's closed-loop load test; returns its JSON line as a dict.  probe_out: file that receives
    the reply to one fixed request (item row 0 as the history): status, then top_k ids, one per line."""
    import json
    import subprocess
    cmd = [build_serve_host(), index_dir, item_embs_dir, str(dim), "--clients", str(clients), "--seconds", str(seconds),
           "--max-batch", str(max_batch), "--max-wait-us", str(max_wait_us), "--ef", str(ef), "--topk", str(topk),
           "--seq-len", str(seq_len), "--lanes", str(lanes), "--client-threads", str(client_threads)]
    if model_dir:
        cmd += ["--model-dir", model_dir]
    if probe_out:
        cmd += ["--probe-out", probe_out]
    if mixed_topn:
        cmd += ["--mixed-topn", "1"]
    if probe_topn is not None:
        cmd += ["--probe-topn", ",".join(str(int(x)) for x in probe_topn)]
    if max_queue:
        cmd += ["--max-queue", str(int(max_queue))]
    if deadline_ms:
        cmd += ["--deadline-ms", str(float(deadline_ms))]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=120 + 2 * seconds).stdout
    return json.loads(out.strip().splitlines()[-1])
