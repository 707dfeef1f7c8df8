"""CPU: the oracle's restatement of the reference's scorer model (attention over
the 50 x 64 user sequence + DNN 128-64-32-1, model.py:189-233, model_util.py:70-97) against an
independent float64 numpy restatement written from the same lines.  PARITY UNPINNED: TensorFlow is
not in the image, so neither has been compared with a run of the frozen graph."""
import numpy as np
import pytest


def np_model(w, u, rows):
    f = np.float64
    prelu = lambda x, a: np.maximum(0.0, x) + a * np.minimum(0.0, x)   # model_util.py:9-11
    q = prelu(rows @ w["wq1"].astype(f) + w["bq1"], w["aq"])            # :81
    q_ = q @ w["wq2"].astype(f) + w["bq2"]                              # :82
    k = prelu(u @ w["wk1"].astype(f) + w["bk1"], w["ak"])               # :84
    k_ = k @ w["wk2"].astype(f) + w["bk2"]                              # :85
    att = q_ @ k_.T / np.sqrt(q_.shape[-1])                             # :90-91
    att = np.exp(att - att.max(-1, keepdims=True))
    p = att / att.sum(-1, keepdims=True)                                # :93
    a = p @ u                                                           # :95 + model.py:206
    x = np.concatenate([a, rows], axis=-1)                              # model.py:211
    for i in range(3):                                                  # :213-216
        x = prelu((x @ w["w"][i].astype(f) + w["b"][i]) * w["bn_scale"][i] + w["bn_shift"][i], w["alpha"][i])
    return x @ w["w"][3].astype(f)                                      # :218-219


@pytest.mark.parametrize("d,dtype", [(64, "f16"), (128, "f32"), (64, "bf16")])
def test_attn_model_matches_numpy(oracle, d, dtype):
    from nann_amd import synth
    E, L, n = 64, 50, 300
    w = synth.make_attn_weights(d, E)
    rng = np.random.default_rng(d)
    u = (rng.standard_normal((L, E)) / 8).astype(np.float16).astype(np.float32)
    u[37:] = 0.0                                                        # zero-padded tail of the history
    x = (rng.standard_normal((n, d)) / 8).astype(np.float32)
    if dtype == "f16":
        rows, code = x.astype(np.float16), oracle.EMB_F16
        xf = rows.astype(np.float64)
    elif dtype == "bf16":
        bits = (x.view(np.uint32) >> 16).astype(np.uint16)              # truncation is fine for a test input
        rows, code = bits, oracle.EMB_BF16
        xf = (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    else:
        rows, code, xf = x, oracle.EMB_F32, x.astype(np.float64)
    m = oracle.AttnModel(d, E, L, code, w)
    rc, got = oracle.attn_score_rows(m, u, rows)
    exp = np_model(w, u.astype(np.float64), xf)
    assert rc == 0
    assert np.abs(got - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max()), np.abs(got - exp).max()
    assert np.std(exp) > 1e-3                                           # the logits do depend on the row
    rc, _ = oracle.attn_score_rows(m, u, rows[:0])
    assert rc == 6                                                      # empty batch (blaze_xla_predictor.cc:259-263)
