"""CPU: a lane-level emulation of k_score_attn (nann_amd/csrc/nann_attn_kernels.h) for one
wavefront -- the same sequence of staged slices, in-place v_mfma_f32_32x32x2_f32 steps, fills,
activations and the register softmax, with the MFMA operand/result layout the hardware-verified
MLP kernel relies on (A: lane l -> row l&31, k-slot l>>5; B: lane l -> column l&31, k-slot l>>5;
D: lane l holds column l&31, rows (r&3) + 8(r>>2) + 4(l>>5)) -- against the oracle's scorer
model.  It checks the index arithmetic of the kernel (which weight row meets which register),
not the hardware; the kernel itself still has to be run on an MI355X."""
import numpy as np
import pytest

LANES = np.arange(64)
UNIT, SLOT = LANES & 31, LANES >> 5


def cd_row(r, slot):
    return (r & 3) + 8 * (r >> 2) + 4 * slot


def mfma(a, b, acc):
    """acc: [64 lanes, 16]; a, b: [64].  D[i][n] += A[i][0] B[0][n] + A[i][1] B[1][n]."""
    A = np.zeros((32, 2), np.float64)
    B = np.zeros((2, 32), np.float64)
    A[UNIT, SLOT] = a
    B[SLOT, UNIT] = b
    D = A @ B                                   # [32 rows, 32 cols]
    out = acc.copy()
    for r in range(16):
        out[:, r] += D[cd_row(r, SLOT), UNIT]
    return out


def attn_mma(slice_, nc, kt, mt, inp, in0, out, out0):
    for t in range(kt):
        for r in range(16):
            krow = 32 * t + cd_row(r, SLOT)
            for m in range(mt):
                a = slice_[krow * nc + 32 * m + UNIT]
                out[out0 + m] = mfma(a, inp[in0 + t][:, r], out[out0 + m])


def stage(src2d, rows, cols, row0=0, col0=0):
    return np.ascontiguousarray(src2d[row0:row0 + rows, col0:col0 + cols]).reshape(-1).astype(np.float64)


def fill(vec, tile):
    out = np.zeros((64, 16))
    if vec is not None:
        for r in range(16):
            out[:, r] = vec[32 * tile + cd_row(r, SLOT)]
    return out


def act(x, tile, scale, shift, alpha):
    for r in range(16):
        j = 32 * tile + cd_row(r, SLOT)
        v = x[:, r]
        if scale is not None:
            v = v * scale[j] + shift[j]
        x[:, r] = np.maximum(0.0, v) + alpha[j] * np.minimum(0.0, v)


def emulate_wave(w, u, rows, L):
    """rows: [32, d] float64 candidate rows of one wavefront -> logits [32]."""
    d = rows.shape[1]
    et = d // 32
    prelu = lambda x, a: np.maximum(0.0, x) + a * np.minimum(0.0, x)
    # k_attn_prepare
    upad = np.zeros((64, 64))
    upad[:L] = u
    k1 = prelu(u @ w["wk1"] + w["bk1"], w["ak"])
    kt = np.zeros((256, 64))
    kt[:, :L] = (k1 @ w["wk2"] + w["bk2"]).T
    # candidate rows into the C/D layout: e[t][lane, 4g + k] = row[cand][32t + 8g + 4slot + k]
    e = [np.zeros((64, 16)) for _ in range(et)]
    for t in range(et):
        for g in range(4):
            for k in range(4):
                e[t][:, 4 * g + k] = rows[UNIT, 32 * t + 8 * g + 4 * SLOT + k]
    q1 = [fill(w["bq1"], m) for m in range(4)]
    for half in range(2):
        attn_mma(stage(w["wq1"], d, 64, 0, 64 * half), 64, et, 2, e, 0, q1, 2 * half)
    for m in range(4):
        act(q1[m], m, None, None, w["aq"])
    att = [fill(None, 0), fill(None, 1)]
    for t in range(8):
        qt = [fill(w["bq2"], t)]
        attn_mma(stage(w["wq2"], 128, 32, 0, 32 * t), 32, 4, 1, q1, 0, qt, 0)
        attn_mma(stage(kt, 32, 64, 32 * t, 0), 64, 1, 2, qt, 0, att, 0)
    mx = np.full(64, -np.inf)
    for p in range(2):
        for r in range(16):
            l = 32 * p + cd_row(r, SLOT)
            att[p][:, r] = np.where(l < L, att[p][:, r] / 16.0, -np.inf)
            mx = np.maximum(mx, att[p][:, r])
    mx = np.maximum(mx, mx[LANES ^ 32])
    s = np.zeros(64)
    for p in range(2):
        att[p] = np.exp(att[p] - mx[:, None])
        s += att[p].sum(1)
    s = s + s[LANES ^ 32]
    for p in range(2):
        att[p] = att[p] / s[:, None]
    x = [fill(None, 0), fill(None, 1)]
    attn_mma(stage(upad, 64, 64), 64, 2, 2, att, 0, x, 0)
    h1 = [fill(w["b"][0], m) for m in range(4)]
    attn_mma(stage(w["w"][0], 64, 128), 128, 2, 4, x, 0, h1, 0)
    for part in range(et // 2):
        attn_mma(stage(w["w"][0], 64, 128, 64 + 64 * part, 0), 128, 2, 4, e, 2 * part, h1, 0)
    for m in range(4):
        act(h1[m], m, w["bn_scale"][0], w["bn_shift"][0], w["alpha"][0])
    h2 = [fill(w["b"][1], 0), fill(w["b"][1], 1)]
    attn_mma(stage(w["w"][1], 128, 64), 64, 4, 2, h1, 0, h2, 0)
    for m in range(2):
        act(h2[m], m, w["bn_scale"][1], w["bn_shift"][1], w["alpha"][1])
    h3 = [fill(w["b"][2], 0)]
    attn_mma(stage(w["w"][2], 64, 32), 32, 2, 1, h2, 0, h3, 0)
    act(h3[0], 0, w["bn_scale"][2], w["bn_shift"][2], w["alpha"][2])
    part = np.zeros(64)
    for r in range(16):
        part += h3[0][:, r] * w["w"][3][cd_row(r, SLOT)]
    part = part + part[LANES ^ 32]
    return part[:32]


@pytest.mark.parametrize("d", [64, 128])
def test_lane_level_model_of_the_attention_kernel(oracle, d):
    from nann_amd import synth
    E, L = 64, 50
    w = synth.make_attn_weights(d, E)
    rng = np.random.default_rng(7 + d)
    u = (rng.standard_normal((L, E)) / 8).astype(np.float16).astype(np.float32)
    u[44:] = 0
    rows = (rng.standard_normal((32, d)) / 8).astype(np.float16)
    m = oracle.AttnModel(d, E, L, oracle.EMB_F16, w)
    rc, exp = oracle.attn_score_rows(m, u, rows)
    w64 = {k: ([np.asarray(a, np.float64) for a in v] if isinstance(v, list) else np.asarray(v, np.float64))
           for k, v in w.items()}
    got = emulate_wave(w64, u.astype(np.float64), rows.astype(np.float64), L)
    assert rc == 0
    assert np.abs(got - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max()), np.abs(got - exp).max()
