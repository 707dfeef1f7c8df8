"""A GEMM-backed CPU port of the serving schedule with the MLP scorer -- BENCH / TEST INFRASTRUCTURE ONLY.

Why it exists ("What's weak" 5): the parity oracle (oracle/nann_oracle.c) scores with scalar fmaf chains in a
canonical order, ~30 queries/s on 16 cores with the 256-128-1 MLP.  The REFERENCE's CPU path does not score like that:
BlazeXlaOp runs the frozen scorer through XLA / Eigen GEMMs (UO/blaze_op/blaze_xla_predictor.cc:360-459).  As a "what
would the reference's CPU path do on these cores" number the oracle's rate is 10-30x too low, so bench.py reports this
port beside it: the same schedule (build_opt_graph.py:109-149, SURVEY.md Appendix A: GroupGather = CSR row concat,
BitmapRefDifference = first occurrence of every unvisited id in input order, TopKV2 = value desc / position asc), the
candidates of a round scored as ONE batch through f32 GEMMs (numpy on OpenBLAS), one query per thread as
blaze-benchmark runs sessions x threads (gen_benchmark_conf.py:22-30).

It is NOT a parity checker: GEMM summation order differs from the canonical chains, so its scores agree with the oracle
within ~1e-6 and its id lists up to near-ties (tests/test_gemm_baseline_cpu.py holds it to that).  Nothing under
nann_amd/ imports it.
"""
import threading
import time

import numpy as np

try:  # one BLAS thread per query thread
    from threadpoolctl import threadpool_limits
except ImportError:  # pragma: no cover
    threadpool_limits = None


def _to_f32(rows, dtype_code):
    if rows.dtype == np.float16:
        return rows.astype(np.float32)
    if rows.dtype == np.uint16:  # bf16 bit patterns
        return (rows.astype(np.uint32) << 16).view(np.float32)
    return rows.astype(np.float32, copy=False)


class GemmSearcher:
    def __init__(self, g, w):
        self.embs = g["item_embs"]
        self.item_ids = np.asarray(g["item_ids"], np.int64)
        self.nbv = [np.asarray(v, np.int32) for v in g["nb_values"]]
        self.nbrs = [np.asarray(r, np.int64) for r in g["nb_row_splits"]]
        self.enter = np.asarray(g["enter_points"], np.int32)
        self.n, self.d = self.embs.shape
        d = self.d
        self.w1q = np.ascontiguousarray(np.asarray(w["w1"], np.float32)[:d])   # [d, 256] query half
        self.w1e = np.ascontiguousarray(np.asarray(w["w1"], np.float32)[d:])   # [d, 256] item half
        self.b1, self.a1 = np.asarray(w["b1"], np.float32), np.asarray(w["alpha1"], np.float32)
        self.w2 = np.ascontiguousarray(np.asarray(w["w2"], np.float32))
        self.b2, self.a2 = np.asarray(w["b2"], np.float32), np.asarray(w["alpha2"], np.float32)
        self.w3 = np.asarray(w["w3"], np.float32)

    def _score(self, u, idx):
        e = _to_f32(self.embs[idx], None)
        h = e @ self.w1e + u
        h = np.where(h > 0, h, h * self.a1)
        h = h @ self.w2 + self.b2
        h = np.where(h > 0, h, h * self.a2)
        return h @ self.w3

    def _gather(self, level, frontier):  # GroupGather, one group, duplicates kept
        rs = self.nbrs[level]
        starts, ends = rs[frontier], rs[frontier.astype(np.int64) + 1]
        lens = (ends - starts).astype(np.int64)
        total = int(lens.sum())
        if total == 0:
            return np.empty(0, np.int32)
        offs = np.repeat(starts - np.concatenate(([0], np.cumsum(lens)[:-1])), lens) + np.arange(total)
        return self.nbv[level][offs]

    @staticmethod
    def _diff(c, visited):  # BitmapRefDifference: first occurrence of every unvisited id, input order; marks them
        if c.size == 0:
            return c
        _, first = np.unique(c, return_index=True)
        first.sort()
        c = c[first]
        c = c[~visited[c]]
        visited[c] = True
        return c

    @staticmethod
    def _topk(ids, s, k):  # TopKV2 sorted: value desc, ties -> lower position
        if s.size < k:
            return None
        order = np.argsort(-s, kind="stable")[:k]
        return ids[order], s[order]

    def search_one(self, q, t, visited):
        """-> (status, item_ids i64[t5], scores f32[t5]); visited: a per-thread bool[n] scratch, returned all-False"""
        u = q @ self.w1q + self.b1
        touched = []
        try:
            if self.enter.size == 0:
                return 6, None, None
            r = self._topk(self.enter, self._score(u, self.enter), t[0])
            if r is None:
                return 4, None, None
            R, sR = r
            C = self._gather(1, R)
            visited[R] = True; touched.append(R)
            C = self._diff(C, visited); touched.append(C)
            if C.size == 0:
                return 6, None, None
            if C.size == 1:
                return 8, None, None
            r = self._topk(np.concatenate([R, C]), np.concatenate([sR, self._score(u, C)]), t[1])
            if r is None:
                return 4, None, None
            P, sP = r
            for a in touched:
                visited[a] = False
            touched = [P]
            visited[P] = True
            B = P
            for i in range(3):
                C = self._diff(self._gather(0, B), visited); touched.append(C)
                if C.size == 0:
                    return 6, None, None
                if C.size == 1:
                    return 8, None, None
                r = self._topk(C, self._score(u, C), t[2 + i])
                if r is None:
                    return 4, None, None
                B, sB = r
                P, sP = np.concatenate([P, B]), np.concatenate([sP, sB])
            r = self._topk(P, sP, t[5])
            if r is None:
                return 4, None, None
            return 0, self.item_ids[r[0]], r[1].astype(np.float32)
        finally:
            for a in touched:
                visited[a] = False


def search_batch(g, w, q, level_topn, n_threads=1):
    """-> (status i32[B], item_ids i64[B, k], scores f32[B, k], seconds): one query per thread."""
    s = GemmSearcher(g, w)
    q = np.asarray(q, np.float32)
    b, k = q.shape[0], int(level_topn[5])
    t = [int(x) for x in level_topn]
    status = np.zeros(b, np.int32)
    ids = np.zeros((b, k), np.int64)
    scores = np.zeros((b, k), np.float32)
    nxt = [0]
    lock = threading.Lock()

    def worker():
        visited = np.zeros(s.n, bool)
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= b:
                return
            st, ii, ss = s.search_one(q[i], t, visited)
            status[i] = st
            if st == 0:
                ids[i], scores[i] = ii, ss

    t0 = time.perf_counter()
    ctx = threadpool_limits(limits=1) if threadpool_limits is not None else None
    try:
        th = [threading.Thread(target=worker) for _ in range(max(1, int(n_threads)))]
        for x in th:
            x.start()
        for x in th:
            x.join()
    finally:
        if ctx is not None:
            ctx.unregister() if hasattr(ctx, "unregister") else ctx.restore_original_limits()
    return status, ids, scores, time.perf_counter() - t0
