"""The product-quantiser scan's oracle — CPU only.  The scan is a DEFINITION on top of pkg/distancepq (the package
that used it, pkg/hnswpq, is absent from the reference): what can be pinned is (i) the leaf arithmetic against an INDEPENDENT
restatement of asm/dot.s / asm/euclidean.s with exact-rational FMA (oracle/pyref.py), (ii) the definition's two restatements
against each other and against the committed fixture tests/golden/pq.npz."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import oracle as O, pyref as P  # noqa: E402
import make_golden_pq as MG  # noqa: E402
from util import bits  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "pq.npz"))


@pytest.mark.parametrize("d", [1, 2, 7, 8, 31, 32, 33, 40, 64, 96, 100])
def test_distancepq_kernels_equal_the_exact_rational_restatement(d):
    """std::fmaf in the C++ oracle == one rounding of the exact product-sum, in the 4 x 8-lane order of the avo-generated assembly
    (pkg/distancepq/asm/dot.s:7-55, euclidean.s:7-65) — incl. tiny and huge magnitudes (subnormal partial sums, cancellations)."""
    for t in range(6):
        x = O.fill_normal(31000 + 97 * d + t, (d,)); y = O.fill_normal(32000 + 97 * d + t, (d,))
        if t == 1: x = x * np.float32(1e-22); y = y * np.float32(1e-20)
        if t == 2: y = -x.copy(); y[0] = np.float32(y[0] * np.float32(1.0000001))
        if t == 3: x = x * np.float32(2e18); y = y * np.float32(1e18)
        if t == 4: y = x.copy()
        assert bits(O.pq_dot(x, y)) == bits(P.pq_dot(x, y)), (d, t)
        assert bits(O.pq_l2sq(x, y)) == bits(P.pq_l2sq(x, y)), (d, t)


def test_fma32_rounds_once():
    # a*b + c where the double-rounded float64 path differs from the fused result
    a, b, c = np.float32(1 + 2**-12), np.float32(1 + 2**-12), np.float32(-(1 + 2**-11))
    assert P.fma32(a, b, c) == np.float32(2.0**-24)          # exact: 2^-24; a separately rounded product would give 0
    assert bits(P.fma32(np.float32(-0.0), np.float32(3), np.float32(-0.0))) == 0x80000000
    assert bits(P.fma32(np.float32(-1), np.float32(1), np.float32(1))) == 0           # exact cancellation -> +0
    assert P.fma32(np.float32(3e38), np.float32(2), np.float32(0)) == np.float32(np.inf)
    assert P.fma32(np.float32(1e-30), np.float32(1e-30), np.float32(0)) == np.float32(0)   # underflow to zero
    tiny = P.fma32(np.float32(2**-75), np.float32(2**-74), np.float32(0))                   # exactly the smallest subnormal
    assert bits(tiny) == 1


@pytest.mark.parametrize("cfg", [(12, 4, 5), (24, 3, 17), (70, 2, 9), (64, 8, 16)])
def test_oracle_scan_equals_the_independent_python(cfg):
    dim, m, c = cfg
    X = O.fill_normal(40 + dim, (60, dim)); Q = O.fill_normal(41 + dim, (2, dim)); T = O.fill_normal(42 + dim, (max(c, 40), dim))
    ids = np.arange(60, dtype=np.uint64)[::-1].copy() * np.uint64(7)
    cb = O.pq_train(T, m, c, 2)
    codes = O.pq_encode(cb, X)
    for metric in (O.PQ_COSINE, O.PQ_EUCLIDEAN, O.PQ_DOT):
        oi, os_, oc, _ = O.pq_search(metric, cb, codes, Q, 7, ids=ids)
        for qi in range(len(Q)):
            pcodes, plut, pi, ps = P.pq_search(metric, cb, X, ids, Q[qi], 7)
            assert np.array_equal(pcodes, codes)
            assert np.array_equal(bits(plut), bits(O.pq_lut(metric, cb, Q[qi])))
            assert np.array_equal(pi, oi[qi]) and np.array_equal(bits(ps), bits(os_[qi])) and oc[qi] == 7


def test_oracle_reproduces_the_golden_fixture():
    X, Q, T, ids = MG.inputs(MG.A)
    cb = O.pq_train(T, MG.A["m"], MG.A["c"], 2)
    assert np.array_equal(cb.view(np.uint32), GOLD["a_codebooks_bits"])
    codes = O.pq_encode(cb, X)
    assert np.array_equal(codes, GOLD["a_codes"])
    for metric in (0, 1, 2):
        oi, os_, _, _ = O.pq_search(metric, cb, codes, Q, MG.A["k"], ids=ids)
        for qi in range(MG.A["nq"]):
            assert np.array_equal(bits(O.pq_lut(metric, cb, Q[qi])), GOLD[f"a_lut_{metric}_{qi}"])
            assert np.array_equal(oi[qi], GOLD[f"a_ids_{metric}_{qi}"]) and np.array_equal(bits(os_[qi]), GOLD[f"a_scores_{metric}_{qi}"])
    X, Q, T, ids = MG.inputs(MG.B)
    cb = O.pq_train(T, MG.B["m"], MG.B["c"], 1)
    codes = O.pq_encode(cb, X)
    assert np.array_equal(codes, GOLD["b_codes"])
    for metric in (0, 1, 2):
        oi, os_, _, _ = O.pq_search(metric, cb, codes, Q, MG.B["k"], ids=ids, threads=2)
        assert np.array_equal(oi, GOLD[f"b_ids_{metric}"]) and np.array_equal(bits(os_), GOLD[f"b_scores_{metric}"])


def test_encode_ties_take_the_lowest_centroid_and_nan_is_never_chosen():
    cb = np.zeros((2, 4, 3), np.float32)
    cb[0, 1] = cb[0, 3] = [1, 2, 3]            # duplicate centroids 1 and 3
    cb[0, 2] = [np.nan, 0, 0]
    cb[1, :] = [[5, 5, 5], [np.nan] * 3, [np.nan] * 3, [np.nan] * 3]
    x = np.array([[1, 2, 3, 9, 9, 9]], np.float32)
    assert O.pq_encode(cb, x).tolist() == [[1, 0]]
    allnan = np.full((2, 4, 3), np.nan, np.float32)
    assert O.pq_encode(allnan, x).tolist() == [[0, 0]]      # no distance ever beats MaxFloat32: the code stays 0


def test_score_is_the_sequential_sum_in_subvector_order():
    m, c = 40, 256
    lut = (O.fill_normal(77, (m, c)) * np.float32(1e3)).astype(np.float32)
    codes = np.random.default_rng(5).integers(0, c, (500, m), dtype=np.uint8)
    want = np.zeros(500, np.float32)
    for j in range(m):
        want = (want + lut[j, codes[:, j]]).astype(np.float32)
    assert np.array_equal(bits(O.pq_adc(lut, codes)), bits(want))
    # a different association order gives different bits on this data: the order is part of the definition
    pairwise = (lut[np.arange(m)[None, :], codes].astype(np.float32).reshape(500, m // 2, 2).sum(axis=2, dtype=np.float32)).sum(axis=1, dtype=np.float32)
    assert not np.array_equal(bits(pairwise), bits(want))


def test_topk_order_is_score_bits_then_id():
    cb = np.zeros((2, 2, 1), np.float32); cb[:, 1, 0] = 1
    codes = np.array([[0, 0], [1, 1], [0, 0], [0, 1], [0, 0]], np.uint8)
    ids = np.array([50, 40, 30, 20, 10], np.uint64)
    q = np.zeros((1, 2), np.float32)
    oi, os_, oc, _ = O.pq_search(O.PQ_EUCLIDEAN, cb, codes, q, 4, ids=ids)
    assert oi[0].tolist() == [10, 30, 50, 20] and os_[0].tolist() == [0, 0, 0, 1] and oc[0] == 4
    oi, os_, oc, _ = O.pq_search(O.PQ_DOT, cb, codes, q, 5, ids=ids)      # every score is 0 + (-0) + (-0) = +0: ids decide
    assert oi[0].tolist() == [10, 20, 30, 40, 50] and np.all(bits(os_[0]) == 0)


def test_train_is_deterministic_and_centroids_are_means_of_their_members():
    T = O.fill_normal(88, (400, 16))
    cb1 = O.pq_train(T, 4, 8, 3); cb2 = O.pq_train(T, 4, 8, 3)
    assert np.array_equal(cb1.view(np.uint32), cb2.view(np.uint32))
    cb0 = O.pq_train(T, 4, 8, 0)
    assert np.array_equal(cb0, T[:8].reshape(8, 4, 4).transpose(1, 0, 2))          # iteration 0 = the first C sample vectors
    prev = O.pq_train(T, 4, 8, 2)
    codes = O.pq_encode(prev, T)                                                   # the assignment the third iteration averages
    for j in range(4):
        for c in range(8):
            mem = T[codes[:, j] == c][:, 4 * j:4 * j + 4]
            if len(mem):
                s = np.zeros(4, np.float32)
                for v in mem:
                    s = (s + v).astype(np.float32)
                assert np.array_equal(bits(cb1[j, c]), bits((s / np.float32(len(mem))).astype(np.float32)))
            else:
                assert np.array_equal(bits(cb1[j, c]), bits(prev[j, c]))
    with pytest.raises(ValueError):
        O.pq_train(T[:5], 4, 8, 1)
