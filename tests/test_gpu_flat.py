"""FLAT VertexSearch on the GPU vs the oracle's canonical form: bit-exact ids, ranks and scores
(SURVEY.md §8a rows a8-a11)."""
import numpy as np
import pytest

from oracle import oracle as O
from util import assert_same_results, bits

pytestmark = pytest.mark.gpu


def build_pair(gpu, n, d, metric, quant, seed=1):
    X = O.fill_normal(seed, (n, d))
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(7919) + np.uint64(1000003)) % np.uint64(1 << 40)
    of = O.Flat(d, metric, quant); of.upsert(ids, X)
    gf = gpu.FlatSpace(d, metric, quant); gf.ChangedVertex(ids, X)
    return X, ids, of, gf


@pytest.mark.parametrize("quant", [O.Q_NONE, O.Q_F16, O.Q_F8, O.Q_BF16])
@pytest.mark.parametrize("metric", [O.COSINE, O.L2])
def test_flat_search_parity(gpu, metric, quant):
    n, d = 2048, 128
    X, ids, of, gf = build_pair(gpu, n, d, metric, quant)
    assert gf.LoadSize() == n
    for i in (0, 17, n - 1):  # stored bits == reference Normalize+Lower
        assert np.array_equal(gf.Stored(ids[i]).view(np.uint8), of.get(ids[i]).view(np.uint8))
    Q = O.fill_normal(99, (19, d))
    for k in (1, 10, 100):
        for select in (gpu.SELECT_REFERENCE, gpu.SELECT_NEAREST):
            gi, gs, gc = gf.VertexSearch(Q, k, select)
            for qi in range(len(Q)):
                wi, ws = of.search(Q[qi], k, nearest=bool(select), mode=2)
                assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi} k{k} sel{select}")


def test_flat_768_and_literal_heap(gpu):
    """config-2 shape at test size; also: the canonical answer equals the literal Go-heap emulation."""
    n, d = 4096, 768
    X, ids, of, gf = build_pair(gpu, n, d, O.COSINE, O.Q_NONE, seed=5)
    Q = O.fill_normal(6, (16, d))
    gi, gs, gc = gf.VertexSearch(Q, 10, gpu.SELECT_REFERENCE)
    for qi in range(16):
        for mode in (0, 1, 2):
            wi, ws = of.search(Q[qi], 10, nearest=False, mode=mode)
            assert_same_results(gi[qi], gs[qi], wi, ws, f"mode{mode}")


def test_flat_ragged_and_edge_cases(gpu):
    d = 20  # not a multiple of 8: scalar tail path, padded row stride
    X, ids, of, gf = build_pair(gpu, 333, d, O.L2, O.Q_F16, seed=8)
    Q = O.fill_normal(3, (5, d))
    gi, gs, gc = gf.VertexSearch(Q, 500, gpu.SELECT_NEAREST)  # k > n
    assert (gc == 333).all()
    for qi in range(5):
        wi, ws = of.search(Q[qi], 500, nearest=True, mode=2)
        assert_same_results(gi[qi, :333], gs[qi, :333], wi, ws)
    # empty store
    e = gpu.FlatSpace(d, O.COSINE, O.Q_NONE)
    _, _, c = e.VertexSearch(Q, 10)
    assert (c == 0).all()
    with pytest.raises(ValueError):
        e.ChangedVertex([1], np.zeros((1, d + 1), np.float32))


def test_flat_ties_upsert_remove_filter(gpu):
    n, d = 600, 64
    X = O.fill_normal(21, (n, d)); X[100:140] = X[7]  # 40 exact duplicates -> ties at the boundary
    ids = np.arange(n, dtype=np.uint64) + np.uint64(50)
    of = O.Flat(d, O.COSINE, O.Q_NONE); gf = gpu.FlatSpace(d, O.COSINE, O.Q_NONE)
    of.upsert(ids, X); gf.ChangedVertex(ids, X)
    Q = np.concatenate([X[7:8], O.fill_normal(4, (6, d))])

    def check(tag, cand=None):
        for k in (5, 25, 60):
            for select in (0, 1):
                if cand is None: gi, gs, gc = gf.VertexSearch(Q, k, select)
                else: gi, gs, gc = gf.FilterableVertexSearch(cand, Q, k, select)
                for qi in range(len(Q)):
                    wi, ws = of.search(Q[qi], k, nearest=bool(select), mode=2, cand=cand)
                    assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"{tag} q{qi} k{k} s{select}")
    check("ties")
    # overwrite some ids, add new ones, remove others (incl. unknown ids: no-op like Go's delete)
    up_ids = np.concatenate([ids[10:30], np.arange(5000, 5040, dtype=np.uint64)]); up = O.fill_normal(77, (60, d))
    of.upsert(up_ids, up); gf.ChangedVertex(up_ids, up)
    rm = np.concatenate([ids[200:260], np.array([999999], np.uint64), ids[n - 1:]])
    of.remove(rm); gf.RemoveVertex(rm)
    assert gf.LoadSize() == len(of)
    check("after-mutation")
    cand = np.concatenate([ids[::3], np.array([424242, 5001, 5003], np.uint64)])  # includes removed + unknown ids
    check("filtered", cand)
    check("filtered-empty", np.array([1, 2, 3], np.uint64))


def test_flat_adversarial_order_overflow_path(gpu):
    """rows sorted so every later row beats the threshold: exercises the safe (segmented) selection path."""
    n, d = 70000, 8
    base = O.fill_normal(31, d)
    scale = (1.0 + np.arange(n, dtype=np.float32) / n)[:, None]
    X = (base[None, :] * scale).astype(np.float32)  # L2 distance to 0 grows with the row index
    ids = np.arange(n, dtype=np.uint64)
    of = O.Flat(d, O.L2, O.Q_NONE); gf = gpu.FlatSpace(d, O.L2, O.Q_NONE)
    of.upsert(ids, X); gf.ChangedVertex(ids, X)
    q = np.zeros((1, d), np.float32)
    gi, gs, gc = gf.VertexSearch(q, 10, gpu.SELECT_REFERENCE)
    wi, ws = of.search(q[0], 10, nearest=False, mode=2)
    assert_same_results(gi[0], gs[0], wi, ws)


@pytest.mark.parametrize("quant", [O.Q_F16, O.Q_BF16, O.Q_NONE])
def test_flat_mfma_mode_equals_exact_mode(gpu, quant):
    """COLTT_MODE_MFMA: matrix-core candidate generation + exact re-score returns the SAME ids, ranks and score bits as
    the exact-order scan (and hence as the oracle), for every batch size / k / direction."""
    n, d = 6000, 128
    X, ids, of, gf = build_pair(gpu, n, d, O.COSINE, quant, seed=17)
    for nq in (3, 70, 200, 300):
        Q = O.fill_normal(1000 + nq, (nq, d))
        for k in (1, 10, 100):
            for select in (gpu.SELECT_REFERENCE, gpu.SELECT_NEAREST):
                ei, es, ec = gf.VertexSearch(Q, k, select, gpu.MODE_EXACT)
                mi, ms, mc = gf.VertexSearch(Q, k, select, gpu.MODE_MFMA)
                assert np.array_equal(ec, mc) and np.array_equal(ei, mi), (nq, k, select)
                assert np.array_equal(bits(es), bits(ms)), (nq, k, select)
        wi, ws = of.search(Q[0], 10, nearest=True, mode=2)
        mi, ms, mc = gf.VertexSearch(Q[:1], 10, gpu.SELECT_NEAREST, gpu.MODE_MFMA)
        assert_same_results(mi[0], ms[0], wi, ws, "mfma vs oracle")


def test_flat_mfma_768_duplicates_and_fallbacks(gpu):
    n, d = 9000, 768
    X = O.fill_normal(23, (n, d)); X[500:540] = X[3]       # exact duplicates: boundary ties go through the margin logic
    ids = np.arange(n, dtype=np.uint64) + np.uint64(7)
    gf = gpu.FlatSpace(d, O.COSINE, O.Q_F16); gf.ChangedVertex(ids, X)
    Q = np.concatenate([X[3:4], O.fill_normal(24, (99, d))])
    for k, select in ((10, 1), (25, 1), (60, 0)):
        ei, es, ec = gf.VertexSearch(Q, k, select, gpu.MODE_EXACT)
        mi, ms, mc = gf.VertexSearch(Q, k, select, gpu.MODE_MFMA)
        assert np.array_equal(ei, mi) and np.array_equal(bits(es), bits(ms)), (k, select)
    # combinations the MFMA kernel does not cover (f8 codes, L2, dim % 32 != 0) are served by the exact path: same answer
    for metric, quant, dd in ((O.L2, O.Q_F16, 128), (O.COSINE, O.Q_F8, 128), (O.COSINE, O.Q_F16, 72)):
        Y = O.fill_normal(5, (700, dd)); g2 = gpu.FlatSpace(dd, metric, quant); g2.ChangedVertex(np.arange(700, dtype=np.uint64), Y)
        q = O.fill_normal(6, (5, dd))
        a = g2.VertexSearch(q, 7, 1, gpu.MODE_EXACT); b = g2.VertexSearch(q, 7, 1, gpu.MODE_MFMA)
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))


def test_flat_mfma_large_dims(gpu):
    """dim = 1536 / 2048 through the MFMA path."""
    for d in (1536, 2048):
        n = 3000
        X = O.fill_normal(29, (n, d)); ids = np.arange(n, dtype=np.uint64)
        gf = gpu.FlatSpace(d, O.COSINE, O.Q_F16); gf.ChangedVertex(ids, X)
        Q = O.fill_normal(30, (130, d))
        for k, select in ((10, 1), (33, 0)):
            ei, es, ec = gf.VertexSearch(Q, k, select, gpu.MODE_EXACT)
            mi, ms, mc = gf.VertexSearch(Q, k, select, gpu.MODE_MFMA)
            assert np.array_equal(ei, mi) and np.array_equal(bits(es), bits(ms)), (d, k, select)


@pytest.mark.parametrize("quant", [O.Q_NONE, O.Q_F16, O.Q_F8, O.Q_BF16])
def test_flat_save_load_vertex_streams(gpu, quant):
    """SaveVertex / LoadVertex: the GPU store writes the byte-identical stream the oracle writes (canonical shard/id order),
    and loads the oracle's stream into a store that answers identically."""
    n, d = 700, 24
    X = O.fill_normal(41, (n, d)); ids = (np.arange(n, dtype=np.uint64) * np.uint64(977) + np.uint64(13)) % np.uint64(1 << 30)
    of = O.Flat(d, O.COSINE, quant); of.upsert(ids, X)
    gf = gpu.FlatSpace(d, O.COSINE, quant); gf.ChangedVertex(ids, X)
    stream = of.save_vertex()
    assert gf.SaveVertex() == stream
    g2 = gpu.FlatSpace(d, O.COSINE, quant)
    assert g2.LoadVertex(stream) == n and g2.LoadSize() == n
    assert g2.SaveVertex() == stream
    Q = O.fill_normal(42, (9, d))
    a = gf.VertexSearch(Q, 10, gpu.SELECT_NEAREST); b = g2.VertexSearch(Q, 10, gpu.SELECT_NEAREST)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    o2 = O.Flat(d, O.COSINE, quant); assert o2.load_vertex(stream) == 0
    wi, ws = o2.search(Q[0], 10, nearest=True, mode=2)
    assert_same_results(b[0][0], b[1][0], wi, ws)
    with pytest.raises(gpu.ColttError):
        g2.LoadVertex(stream[:-3])
    assert gpu.FlatSpace(d, O.COSINE, quant).LoadVertex(gpu.FlatSpace(d, O.COSINE, quant).SaveVertex()) == 0


def test_flat_config0_full_size_single_queries(gpu):
    """BASELINE.json configs[0] at full size — 100 000 x 128 f32 cosine, single-query searches, k = 10 — GPU (exact-order
    mode and matrix-core mode) vs the oracle's canonical scan, both select directions; the CPU side finishes in seconds."""
    n, d = 100_000, 128
    X, ids, of, gf = build_pair(gpu, n, d, O.COSINE, O.Q_NONE, seed=41)
    Q = O.fill_normal(42, (6, d))
    for qi in range(len(Q)):
        for select in (gpu.SELECT_REFERENCE, gpu.SELECT_NEAREST):
            wi, ws = of.search(Q[qi], 10, nearest=bool(select), mode=2)
            for mode in (gpu.MODE_EXACT, gpu.MODE_MFMA):
                gi, gs, gc = gf.VertexSearch(Q[qi:qi + 1], 10, select, mode=mode)
                assert_same_results(gi[0, :gc[0]], gs[0, :gc[0]], wi, ws, f"q{qi} sel{select} mode{mode}")


@pytest.mark.parametrize("metric,quant", [(O.COSINE, O.Q_F16), (O.L2, O.Q_F16), (O.COSINE, O.Q_NONE)])
def test_flat_mfma_many_tiles_per_workgroup_smallest_dim(gpu, metric, quant, monkeypatch):
    """200 k x 128: 782 row tiles over 256 persistent workgroups (three or four tiles each, the raw-norm parity buffers flip with
    every tile) at the SMALLEST dim the matrix-core mode takes (4 K steps per tile: the DMA rings run three tiles ahead of the
    epilogue).  Ragged batch, batch 256 and a last tile of 64 rows; == exact mode bit for bit.
    (Round 4: dim 96 / 64 / 32 run with K padded to 128 — `Stats()` shows their matrix-core group too; tests/test_gpu_round4.py.)"""
    import subprocess, sys, os, json
    # (a child process: the comparison used to run once per kernel generation, each read once per process)
    code = f"""
import numpy as np, json, sys
sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})
import coltt_amd as G
from oracle import oracle as O
assert G.lib().coltt_init(0) == 0
n, d = 200_000 + 64, 128
X = O.fill_normal(41, (n, d)); ids = np.arange(n, dtype=np.uint64)
gf = G.FlatSpace(d, {metric}, {quant}); gf.ChangedVertex(ids, X)
ok = True
for nq in (256, 37):
    Q = O.fill_normal(42 + nq, (nq, d))
    for k, sel in ((10, G.SELECT_NEAREST), (33, G.SELECT_REFERENCE)):
        e = gf.VertexSearch(Q, k, sel, G.MODE_EXACT); m = gf.VertexSearch(Q, k, sel, G.MODE_MFMA)
        ok &= bool(np.array_equal(e[0], m[0]) and np.array_equal(e[1].view(np.uint32), m[1].view(np.uint32)) and np.array_equal(e[2], m[2]))
st = gf.Stats()
small = G.FlatSpace(96, {metric}, {quant}); small.ChangedVertex(ids[:3000], O.fill_normal(43, (3000, 96)))
small.VertexSearch(O.fill_normal(44, (8, 96)), 5, G.SELECT_NEAREST, G.MODE_MFMA)
print(json.dumps({{"ok": ok, "groups": st["mfma_groups"], "fallbacks": st["mfma_fallbacks"], "small_groups": small.Stats()["mfma_groups"]}}))
"""
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["ok"] and r["groups"] > 0 and r["small_groups"] == 1, r


@pytest.mark.parametrize("d", [130, 200, 300])
@pytest.mark.parametrize("metric,quant", [(O.COSINE, O.Q_F16), (O.L2, O.Q_BF16), (O.COSINE, O.Q_NONE), (O.L2, O.Q_NONE)])
def test_flat_mfma_dims_not_multiple_of_32(gpu, metric, quant, d):
    """K is padded to whole 32-column steps with zero QUERY columns; the rows' overhang (their zeroed padding and the head of the
    next row) multiplies zeros.  70 000 rows (274 tiles: some workgroups take two), a ragged batch: ids, ranks and score bits equal
    exact mode, and the matrix cores really served the call.  A store that ever held a non-finite row stays on the exact scan
    (its bits could sit in another row's overhang): same answers, no matrix-core group."""
    n = 70_000
    X = O.fill_normal(300 + d, (n, d)).astype(np.float32); ids = np.arange(n, dtype=np.uint64)
    gf = gpu.FlatSpace(d, metric, quant); gf.ChangedVertex(ids, X)
    Q = O.fill_normal(301 + d, (70, d))
    for k, sel in ((10, gpu.SELECT_NEAREST), (40, gpu.SELECT_REFERENCE)):
        e = gf.VertexSearch(Q, k, sel, gpu.MODE_EXACT); m = gf.VertexSearch(Q, k, sel, gpu.MODE_MFMA)
        assert np.array_equal(e[0], m[0]) and np.array_equal(bits(e[1]), bits(m[1])) and np.array_equal(e[2], m[2]), (k, sel)
    served = gf.Stats()["mfma_groups"]
    assert served > 0
    bad = X[:1].copy(); bad[0, 3] = np.inf
    gf.ChangedVertex(np.array([n], dtype=np.uint64), bad)
    e = gf.VertexSearch(Q, 10, gpu.SELECT_NEAREST, gpu.MODE_EXACT); m = gf.VertexSearch(Q, 10, gpu.SELECT_NEAREST, gpu.MODE_MFMA)
    assert np.array_equal(e[0], m[0]) and np.array_equal(bits(e[1]), bits(m[1]))
    assert gf.Stats()["mfma_groups"] == served
