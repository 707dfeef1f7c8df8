"""-m gpu, collected last: TopKV2 row lengths between 8 193 and 16 384, which the kernel keeps as
16 keys per thread (wg_topk_impl<16>); the other tests only reach the 8-key and the
re-read-from-memory variants.  Not yet run on hardware when it was added."""
import numpy as np
import pytest
import torch

from gpu_util import bits, cuda, require_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,k", [(12000, 256), (16384, 128), (8193, 1000)])
def test_topk_sixteen_keys_per_thread(oracle, n, k):
    require_gpu()
    from nann_amd import ops
    rng = np.random.default_rng(n + k)
    for x in (rng.standard_normal(n).astype(np.float32),
              rng.integers(0, 60, size=n).astype(np.float32) - 30.0):
        rc, ev, ei = oracle.topk(x, k)
        v, i = ops.top_k(cuda(x), k)
        assert rc == 0
        assert (i.cpu().numpy() == ei).all()
        assert (bits(v.cpu().numpy()) == bits(ev)).all()
