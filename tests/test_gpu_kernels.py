"""GPU leaf kernels vs the CPU oracle, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O
from util import bits

pytestmark = pytest.mark.gpu
DIMS = [4, 8, 12, 32, 36, 128, 384, 768, 1536]


@pytest.mark.parametrize("metric", [O.COSINE, O.L2])
@pytest.mark.parametrize("order", [O.ORDER_AVX, O.ORDER_SSE, O.ORDER_NATIVE])
def test_distance_pairs(gpu, metric, order):
    for d in DIMS + ([1, 7, 9, 31, 33] if order != O.ORDER_AVX else []):
        a = O.fill_normal(11 + d, (64, d)); b = O.fill_normal(977 + d, (64, d))
        got = gpu.kernels.distance_pairs(metric, a, b, order)
        want = np.array([(O.cosine if metric == O.COSINE else O.l2)(a[i], b[i], order) for i in range(64)], np.float32)
        assert np.array_equal(bits(got), bits(want)), (metric, order, d)


def test_normalize(gpu):
    for d in [1, 7, 128, 768]:
        v = O.fill_normal(5, (33, d)); v[3] = 0
        assert np.array_equal(bits(gpu.kernels.normalize(v)), bits(O.normalize(v)))


def test_f16_codec_exhaustive(gpu):
    codes = np.arange(65536, dtype=np.uint16)
    got = gpu.kernels.quant_raise(gpu.Q_F16, codes); want = O.f16_decode(codes)
    assert np.array_equal(bits(got), bits(want))
    assert np.array_equal(bits(gpu.kernels.quant_raise(gpu.Q_BF16, codes)), bits(want))  # "bf16" == binary16 in the reference
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-30, 14, 200000))).astype(np.float32)
    edge = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 65504.0, 65520.0, 65519.99, 6e-8, 5.96e-8, 2.98e-8, 2.9802322e-8,
                     6.1035156e-05, 6.0975552e-05, 1e-45, -1e-45, 1.0009766, 1.0004883, 1.0014648], np.float32)
    x = np.concatenate([x, edge, -edge])
    for q in (gpu.Q_F16, gpu.Q_BF16):
        assert np.array_equal(gpu.kernels.quant_lower(q, x), O.f16_encode(x))


def test_f8_codec(gpu):
    codes = np.arange(256, dtype=np.uint8)
    assert np.array_equal(bits(gpu.kernels.quant_raise(gpu.Q_F8, codes)), bits(O.f8_decode(codes)))
    x = np.concatenate([O.fill_normal(9, 100000), np.array([0, np.inf, -np.inf, np.nan, 1e-8, 70000.0, 3e-5], np.float32)])
    assert np.array_equal(gpu.kernels.quant_lower(gpu.Q_F8, x), O.f8_encode(x))


def test_shard_vertex(gpu):
    ids = (np.arange(5000, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(12345)
    for c in (16, 8, 7):
        want = np.array([O.shard_vertex(int(i), c) for i in ids], np.uint64)
        assert np.array_equal(gpu.kernels.shard_vertex(ids, c), want)


def test_distancepq(gpu):
    for d in [1, 5, 31, 32, 33, 64, 100, 768]:
        q = O.fill_normal(1, d); rows = O.fill_normal(2, (37, d))
        for kind, f in ((0, lambda y: O.pq_dot(q, y)), (1, lambda y: O.pq_l2sq(q, y)), (2, lambda y: np.float32(1) - O.pq_dot(q, y)),
                        (3, lambda y: -O.pq_dot(q, y))):
            got = gpu.kernels.pq_float_scan(kind, q, rows)
            want = np.array([f(rows[i]) for i in range(37)], np.float32)
            assert np.array_equal(bits(got), bits(want)), (d, kind)
    rng = np.random.default_rng(0)
    q = rng.integers(0, 2**63, 12, dtype=np.uint64); rows = rng.integers(0, 2**63, (50, 12), dtype=np.uint64); rows[7] = 0
    assert np.array_equal(bits(gpu.kernels.pq_bit_scan(0, q, rows)), bits(np.array([O.pq_hamming(q, r) for r in rows], np.float32)))
    assert np.array_equal(bits(gpu.kernels.pq_bit_scan(1, q, rows)), bits(np.array([O.pq_jaccard(q, r) for r in rows], np.float32)))
    z = np.zeros(12, np.uint64)
    assert gpu.kernels.pq_bit_scan(1, z, rows[7:8])[0] == 0.0


def test_manhattan_pairs_all_three_orders(gpu):
    """distance.SpaceImpl.ManhattanDistance (pkg/distance/space.go:25-29, simd/cpp/avx.cpp:34-49, sse.cpp:35-53, native_impl.go:33-40):
    no store of the reference calls it; the leaf kernel has a device twin, bit-exact in every order incl. magnitudes where sqrt(d * d) != |d|."""
    K = gpu.kernels
    for d in (1, 7, 8, 9, 31, 33, 128, 770):
        a = O.fill_normal(700 + d, (12, d)); b = O.fill_normal(800 + d, (12, d))
        a[3] *= np.float32(1e-25); b[3] *= np.float32(1e-25); a[5] *= np.float32(1e25)
        for order in (0, 1, 2):
            want = np.array([O.manhattan(a[i], b[i], order) for i in range(12)], np.float32)
            got = K.distance_pairs(2, a, b, order)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (d, order)
