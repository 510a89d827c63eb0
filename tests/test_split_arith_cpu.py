"""The arithmetic of the split-bf16 convolution kernels (csrc/conv_split.hip, wgrad_split.hip, conv1x1_split.hip,
wgrad1x1_split.hip), restated in numpy so that its two claims are checked without a GPU:

  1. the 3-way split x = hi + mid + lo (each term a bf16, round-to-nearest-even) is EXACT for every finite fp32 below the top
     binade (hi would round to infinity there) whose terms do not underflow -- the kernels lose nothing by staging operands as three bf16 planes;
  2. a dot product evaluated as the six cross terms hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi (products exact, fp32 accumulation)
     stays in the error class of a plain fp32 fmaf chain -- the dropped terms are below half an ulp of each product.  The model here
     is pessimistic: it rounds the accumulator after EVERY product (6 roundings per element), the matrix core once per 16-product
     instruction, which is why the measured distance on the hardware equals the fp32 instruction's.

The measured counterpart on the matrix cores is profiles/mfma_split_exp_r02.txt and tests/test_conv_gpu.py."""
import numpy as np


def bf16_rne(x):
    """fp32 -> the nearest bf16 (ties to even), returned as fp32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rne(r1)
    lo = bf16_rne((r1 - mid).astype(np.float32))
    return hi, mid, lo


def test_three_way_split_is_exact():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.standard_normal(200000).astype(np.float32),
        (rng.standard_normal(200000) * np.exp(rng.uniform(-60, 60, 200000))).astype(np.float32),   # 1e-26 .. 1e26
        np.array([0.0, -0.0, 1.0, -1.0, np.float32(1) + np.float32(2 ** -23), np.float32(3.3e38), np.float32(-3.3e38), np.float32(1.1754944e-38) * 2 ** 20], np.float32),
        np.frombuffer(rng.integers(0x00800000 + (20 << 23), 0x7e800000, 100000, dtype=np.uint32).tobytes(), np.float32),   # random bit patterns
    ])
    hi, mid, lo = split3(x)
    assert np.array_equal((hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))   # exact, not just to fp32
    for t in (hi, mid, lo):   # each term really is a bf16
        assert not np.any(t.view(np.uint32) & 0xffff)
    # the terms shrink by at least 2^-8 each (what bounds the dropped cross products)
    nz = x != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_products_match_the_fp32_chain():
    rng = np.random.default_rng(1)
    worst = {"split": 0.0, "fp32": 0.0}
    for dist in range(3):
        for K in (32, 256, 2048):
            a = rng.standard_normal((64, K))
            b = rng.standard_normal((64, K))
            if dist == 1:
                a, b = np.abs(a), np.abs(b)          # long same-sign sums (post-ReLU activations)
            if dist == 2:
                a, b = a * np.exp(4 * rng.standard_normal(a.shape)), b * np.exp(4 * rng.standard_normal(b.shape))
            a, b = a.astype(np.float32), b.astype(np.float32)
            ref = np.sum(a.astype(np.float64) * b.astype(np.float64), axis=1)
            mag = np.sum(np.abs(a.astype(np.float64) * b.astype(np.float64)), axis=1)
            # plain fp32 chain (what v_mfma_f32_*_f32 computes)
            acc = np.zeros(64, np.float32)
            for k in range(K):
                acc = (acc.astype(np.float64) + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32)   # fmaf: one rounding
            # six exact bf16 x bf16 products per element, fp32 accumulation, smallest first (the kernels' order)
            sa, sb = split3(a), split3(b)
            acc6 = np.zeros(64, np.float32)
            for k in range(K):
                for p, q in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
                    prod = sa[p][:, k].astype(np.float64) * sb[q][:, k].astype(np.float64)   # exact: 8 x 8 bit mantissas
                    assert np.array_equal(prod.astype(np.float32).astype(np.float64), prod)
                    acc6 = (acc6.astype(np.float64) + prod).astype(np.float32)
            e32, e6 = np.max(np.abs(acc - ref) / mag), np.max(np.abs(acc6 - ref) / mag)
            worst["fp32"], worst["split"] = max(worst["fp32"], e32), max(worst["split"], e6)
            assert e6 <= 4 * e32 + 2.0 ** -23, (dist, K, e6, e32)
    assert worst["split"] < 1e-5
