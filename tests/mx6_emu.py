"""numpy emulation of the vocabulary normaliser's product forms (test infrastructure; round 6, verdict item 1a).

The device computes a logit  t . b  (t: a hypothesis' projected state, b: a word's output embedding, both scaled by powers of two)
from the split  x = hi + lo,  hi = f16(x):

    split   t_hi.b_hi + t_hi.b_lo + t_lo.b_hi       three f16 matrix passes (csrc/jlm_split.hip)
    mixed   t_hi.b_hi in f16, the two cross terms as int8 x int8 with ONE scale per (row, segment) / per segment   (csrc/jlm_mixed.hip)
    mx6     t_hi.b_hi in f16, the two cross terms as FP6 (e2m3) x FP6 with an E8M0 scale per 32 k-values of every row -- one
            v_mfma_scale_f32_32x32x64_f8f6f4 per 32 k-values, accumulated into the f16 pass's f32 accumulator (csrc/jlm_mx6*.h)

The functions below restate the three in float64 over the SAME quantised operands the packers produce, so that what a form costs in
log-normaliser error (what a path score sees: reference decoder/model.py:15-20, decoder.py:43-49) can be measured -- and the load-time
gate predicted -- without a GPU.  Operand layout of the scaled instruction as decoded on the device by tools/probes/mfma_fp6_probe.hip
(profiles/r06_a_fp6_probe.txt)."""
import numpy as np

LOG2E = 1.4426950408889634


# ------------------------------------------------------------------------------------------------ FP6 e2m3 with a block scale
def e2m3_round(x):
    """nearest e2m3 value (sign, 2 exponent bits with bias 1, 3 mantissa bits: 0, 0.125 .. 0.875, 1 .. 1.875, 2 .. 3.75, 4 .. 7.5;
    ties to even; saturating at 7.5) of every element of x (already divided by its block scale)"""
    a = np.abs(x)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.maximum(a, 1e-300)))
    step = np.exp2(np.clip(e, 0, 2)) * 0.125
    q = np.minimum(np.rint(a / step) * step, 7.5)
    return np.copysign(q, x)


def e2m3_code(q):
    """6-bit codes of e2m3 values (as produced by e2m3_round)"""
    a = np.abs(q)
    e = np.where(a >= 1.0, np.floor(np.log2(np.maximum(a, 1.0))) + 1, 0).astype(np.int64)
    m = np.where(e == 0, np.rint(a * 8.0), np.rint((a / np.exp2(np.maximum(e - 1, 0)) - 1.0) * 8.0)).astype(np.int64)
    return ((np.signbit(q).astype(np.int64) << 5) | (e << 3) | m).astype(np.uint8)


def e2m3_value(code):
    code = np.asarray(code, dtype=np.int64)
    e, m = (code >> 3) & 3, code & 7
    v = np.where(e == 0, m * 0.125, (1.0 + m * 0.125) * np.exp2(np.maximum(e - 1, 0)))
    return np.where((code >> 5) & 1, -v, v)


def block_exponent(x, block=32):
    """per block of `block` consecutive k-values the E8M0 exponent: the smallest power of two s with max|x| <= 7.5 s; all-zero blocks
    get 2^-127 (byte 0).  x [rows, k] with k a multiple of block -> int exponents [rows, k / block]"""
    r, k = x.shape
    amax = np.abs(x).reshape(r, k // block, block).max(axis=2)
    with np.errstate(divide="ignore"):
        e = np.ceil(np.log2(np.maximum(amax, 1e-300) / 7.5))
    e = np.where(amax > 0, e, -127)
    return np.clip(e, -127, 127).astype(np.int64)


def quant6(x, block=32):
    """x [rows, k] (float64 / float32) -> (values as float64 after FP6 quantisation with per-block scales, exponents [rows, k / block])"""
    r, k = x.shape
    e = block_exponent(x, block)
    s = np.exp2(e.astype(np.float64)).repeat(block, axis=1)
    return e2m3_round(x / s) * s, e


# ------------------------------------------------------------------------------------------------ the three product forms
def split_hi_lo(x32):
    hi = x32.astype(np.float16)
    lo = x32 - hi.astype(np.float32)
    return hi.astype(np.float64), lo.astype(np.float64)


def pad32(x):
    r, k = x.shape
    kp = (k + 31) // 32 * 32
    if kp == k:
        return x
    out = np.zeros((r, kp), dtype=x.dtype)
    out[:, :k] = x
    return out


def logits_exact(t32, b32):
    return t32.astype(np.float64) @ b32.astype(np.float64).T


def logits_split(t32, b32):
    th, tl = split_hi_lo(t32)
    bh, bl = split_hi_lo(b32)
    return th @ bh.T + th @ bl.T + tl @ bh.T


def logits_mixed(t32, b32):
    """int8 cross terms: one power-of-two scale per T row (from its largest |hi|) and one per vocabulary block"""
    th, tl = split_hi_lo(t32)
    bh, bl = split_hi_lo(b32)

    def p2(amax):
        with np.errstate(divide="ignore"):
            return np.where(amax > 0, np.exp2(np.ceil(np.log2(np.maximum(amax, 1e-300) / 127.0))), 1.0)
    s_t = p2(np.abs(th).max(axis=1))[:, None]
    s_b = p2(np.abs(bh).max())
    q = lambda v: np.clip(np.rint(v), -127, 127)
    t8, tl8 = q(th / s_t), q(tl / (s_t / 2048.0))
    b8, bl8 = q(bh / s_b), q(bl / (s_b / 2048.0))
    cross = (t8 @ bl8.T + tl8 @ b8.T) * (s_t * s_b / 2048.0)
    return th @ bh.T + cross


def logits_mx6(t32, b32):
    """FP6 cross terms with a scale per 32 k-values of every row, both sides"""
    th, tl = split_hi_lo(pad32(t32))
    bh, bl = split_hi_lo(pad32(b32))
    th6, _ = quant6(th)
    tl6, _ = quant6(tl)
    bh6, _ = quant6(bh)
    bl6, _ = quant6(bl)
    return th @ bh.T + th6 @ bl6.T + tl6 @ bh6.T


FORMS = dict(split=logits_split, mixed=logits_mixed, mx6=logits_mx6)


# ------------------------------------------------------------------------------------------------ a model's probe rows
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lm_probe_rows(cfg, weights, rows=256, steps=3, seed=20240929, words=None):
    """hidden states [rows, H] (float64) after `steps` LSTM steps from the zero state over seeded word ids (reference
    decoder/model.py:125-139), as DeviceModel._calibrate_mixed takes them; ``words`` [steps, rows] overrides the uniform draw"""
    H = cfg["hidden_size"]
    w = weights
    if cfg["V_table"]:
        segs = cfg["embedding_seg"]
        emb = np.concatenate([w["LM0"]] + [w["LM%d" % i] @ w["VT%d" % i] for i in range(1, len(segs))], axis=0)
    elif cfg["D_softmax"]:
        segs = cfg["embedding_seg"]
        V = w["b2"].shape[0]
        emb = np.zeros((V, sum(s[0] for s in segs)))
        c0 = 0
        for i, (size, s, e) in enumerate(segs):
            e = V if e is None else e
            emb[s:e, c0:c0 + size] = w["LM"][i]
            c0 += size
    else:
        emb = w["LM"]
    V = emb.shape[0]
    rng = np.random.RandomState(seed)
    if words is None:
        words = rng.randint(0, V, size=steps * rows).reshape(steps, rows)
    h = np.zeros((rows, H))
    c = np.zeros((rows, H))
    for s in range(len(words)):
        x = emb[words[s]]
        z = {g: h @ w["HM" + g] + x @ w["IM" + g] + w["b" + g] for g in "ifog"}
        i, f, o, g = sigmoid(z["i"]), sigmoid(z["f"]), sigmoid(z["o"]), np.tanh(z["g"])
        c = c * f + g * i
        h = np.tanh(c) * o
    return h


def output_segments(cfg, weights):
    """[(T columns as a function of h: matrix [H, k], block B [V_i, k] f32, v_start, v_end)] of a tied / D-softmax / V-table model
    (reference decoder/model.py:141-186; the V-table projections folded into the panel as jlm_amd/model.py does)"""
    w = weights
    V = w["b2"].shape[0]
    if not cfg["share_embedding"]:
        return [(np.eye(cfg["hidden_size"]), np.ascontiguousarray(w["UM"].T, dtype=np.float32), 0, V)]
    PM = np.asarray(w["PM"], dtype=np.float64)
    if cfg["V_table"]:
        out = []
        for i, (size, s, e) in enumerate(cfg["embedding_seg"]):
            e = V if e is None else e
            panel = PM if i == 0 else (np.asarray(w["VT%d" % i], dtype=np.float64) @ PM.T).astype(np.float32).astype(np.float64).T
            out.append((panel, np.asarray(w["LM%d" % i], dtype=np.float32), s, e))
        return out
    if cfg["D_softmax"]:
        out, c0 = [], 0
        for i, (size, s, e) in enumerate(cfg["embedding_seg"]):
            e = V if e is None else e
            out.append((PM[:, c0:c0 + size], np.asarray(w["LM"][i], dtype=np.float32), s, e))
            c0 += size
        return out
    return [(PM, np.asarray(w["LM"], dtype=np.float32), 0, V)]


def pow2_below(limit, value):
    if not (value > 0.0) or not np.isfinite(value):
        return 0
    return int(np.clip(np.floor(np.log2(limit / value)), -40, 40))


def lse_by_form(cfg, weights, h, forms=("split", "mixed", "mx6")):
    """log-normalisers (base e) of the probe rows: {'exact': [rows], form: [rows]} plus per-form worst logit error relative to the
    rows' logit scale.  Scales as DeviceModel._build_mixed chooses them: 2^eB puts max|B| at <= 2^14, 2^eT the T bound x log2 e at <= 2^15"""
    segs = output_segments(cfg, weights)
    b2 = np.asarray(weights["b2"], dtype=np.float64)
    ys = {f: [] for f in ("exact",) + tuple(forms)}
    for panel, B, s, e in segs:
        t32 = (h @ panel).astype(np.float32)
        tb = max(float(np.abs(panel).sum(axis=0).max()), 1.0)
        eT = pow2_below(2.0 ** 15, tb * LOG2E)
        eB = pow2_below(2.0 ** 14, max(float(np.abs(B).max()), float(np.abs(b2[s:e]).max()) * LOG2E))
        ts = (t32 * np.float32(2.0 ** eT * LOG2E)).astype(np.float32)        # base-2 logit units, as jlm_pack_t_mixed
        bs = (B * np.float32(2.0 ** eB)).astype(np.float32)
        de = 2.0 ** -(eT + eB) / LOG2E
        ys["exact"].append(logits_exact(ts, bs) * de + b2[s:e])
        for f in forms:
            ys[f].append(FORMS[f](ts, bs) * de + b2[s:e])
    out, logit_err = {}, {}
    y0 = np.concatenate(ys["exact"], axis=1)
    scale = np.abs(y0).max(axis=1)
    for f, parts in ys.items():
        y = np.concatenate(parts, axis=1)
        mx = y.max(axis=1)
        out[f] = mx + np.log(np.exp(y - mx[:, None]).sum(axis=1))
        if f != "exact":
            logit_err[f] = float((np.abs(y - y0).max(axis=1) / scale).max())
    return out, logit_err


def form_errors(cfg, weights, rows=256, steps=3, seed=20240929, forms=("split", "mixed", "mx6"), words=None):
    """{form: (rms, max) of lse(form) - lse(exact), 'logit': {form: worst relative logit error}}"""
    h = lm_probe_rows(cfg, weights, rows, steps, seed, words)
    lse, lerr = lse_by_form(cfg, weights, h, forms)
    res = {f: (float(np.sqrt(np.mean((lse[f] - lse["exact"]) ** 2))), float(np.abs(lse[f] - lse["exact"]).max())) for f in forms}
    res["logit"] = lerr
    res["lse_mean"] = float(lse["exact"].mean())
    return res
