"""Collection groups on the GPU (BASELINE.json configs[4] layout on ONE device: N members share device 0, so the packed
per-shard answers travel through pinned host memory; a 1-member group exercises the RCCL all-gather itself —
ncclCommInitAll refuses the same device twice).  Parity: per-shard oracle search + the oracle-side merge of the union."""
import numpy as np
import pytest

from oracle import oracle as O
from util import assert_same_results, bits

pytestmark = pytest.mark.gpu


def _shards(ids, G):
    return np.array([O.shard_vertex(int(i), G) for i in ids])


@pytest.mark.parametrize("G", [4, 8])
@pytest.mark.parametrize("quant,metric", [(O.Q_NONE, O.COSINE), (O.Q_F16, O.L2)])
def test_flat_group_equals_unsharded_store_and_per_shard_oracle(gpu, G, quant, metric):
    """FLAT is exact, so a sharded collection must answer EXACTLY like the unsharded store, in both select directions."""
    from coltt_amd import group as GG
    n, d, k = 3000, 40, 10
    X = O.fill_normal(1300 + G, (n, d)); ids = (np.arange(n, dtype=np.uint64) * np.uint64(7919) + np.uint64(13))
    grp = gpu.Group([0] * G, d, metric, quant, kind=GG.GROUP_FLAT)
    assert grp.info() == {"n_local": G, "world": G, "exchange": "host", "rank_base": 0}
    assert grp.ChangedVertex(ids, X) == n and grp.Len() == n
    sh = _shards(ids, G)
    for i in (0, 5, 77): assert grp.shard_of(ids[i]) == sh[i]
    whole = O.Flat(d, metric, quant); whole.upsert(ids, X)
    parts = []
    for s in range(G):
        f = O.Flat(d, metric, quant); f.upsert(ids[sh == s], X[sh == s]); parts.append(f)
    Q = O.fill_normal(1301, (21, d))
    for nearest in (True, False):
        gi, gs, gc = grp.Search(Q, k, select=gpu.SELECT_NEAREST if nearest else gpu.SELECT_REFERENCE)
        for qi in range(len(Q)):
            wi, ws = whole.search(Q[qi], k, nearest=nearest, mode=2)
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi} near{nearest}")
            # the reference's own shape: local queues per shard, merged into the global queue
            u = sorted((float(s_), int(i_)) for p in parts for i_, s_ in zip(*p.search(Q[qi], k, nearest=nearest, mode=2)))
            want = u[:k] if nearest else u[-k:]
            assert [(float(gs[qi, j]), int(gi[qi, j])) for j in range(gc[qi])] == want
    # removal is routed too
    grp.Remove(ids[:50]); whole.remove(ids[:50])
    gi, gs, gc = grp.Search(Q[:5], k)
    for qi in range(5):
        wi, ws = whole.search(Q[qi], k, nearest=True, mode=2)
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws)
    assert grp.Len() == n - 50


@pytest.mark.parametrize("G", [4, 8])
def test_hnsw_bf16_ef256_sharded_by_shard_vertex(gpu, G):
    """configs[4] on one device: "bf16" HNSW shards (ShardVertex(id, G)), efSearch 256, per-shard search + exact merge.
    Checker per shard: the oracle's canonical search over that shard's arrays copied out of HBM; then the merged answer must
    be the k best of the union; and against the unsharded exact scan the recall must be high."""
    from coltt_amd import group as GG
    n, d, k, ef = 6000, 64, 10, 256
    X = O.fill_normal(1400 + G, (n, d)); lv = O.levels(1401 + G, n)
    ids = np.arange(n, dtype=np.uint64) * np.uint64(104729) + np.uint64(5)
    grp = gpu.Group([0] * G, d, O.COSINE, O.Q_BF16, kind=GG.GROUP_HNSW, cfg=gpu.HnswCfg.default(ef_construction=80))
    for b in range(0, n, 1000):   # several ingest calls; batch 1 inside a shard == the reference's sequential Insert
        assert grp.Insert(ids[b:b + 1000], X[b:b + 1000], lv[b:b + 1000], batch=1) == 1000
    assert grp.Len() == n
    sh = _shards(ids, G)
    Q = O.fill_normal(1402, (32, d))
    gi, gs, gc = grp.Search(Q, k, ef=ef)
    per_shard = []
    for s in range(G):
        m = gpu.Hnsw.__new__(gpu.Hnsw); m.h = grp.member(s); m.dim = d; m.quantization = O.Q_BF16; m.cfg = gpu.HnswCfg()
        m.Config()
        g = m.ExportRaw(); rows = m.FetchRows(); ex = m.Export()
        assert np.array_equal(np.sort(ex["ids"]), np.sort(ids[sh == s]))             # routing: exactly this shard's vertices
        sl, sc, cn, _, _ = O.csr_search(rows, O.Q_BF16, g["adj0"], g["upper_off"], g["adjU"], d, O.COSINE, g["entry"], g["entry_level"], Q, k, ef, threads=2)
        per_shard.append((ex["ids"], sl, sc, cn))
        m.h = None                                                                    # the group owns the member
    for qi in range(len(Q)):
        u = sorted((float(sc[qi, j]), int(idv[sl[qi, j]])) for idv, sl, sc, cn in per_shard for j in range(cn[qi]))[:k]
        assert [(float(gs[qi, j]), int(gi[qi, j])) for j in range(gc[qi])] == u, qi
        got_bits = bits(gs[qi, :gc[qi]])
        assert np.array_equal(got_bits, bits(np.float32([x[0] for x in u])))
    fl = gpu.FlatSpace(d, O.COSINE, O.Q_BF16); fl.ChangedVertex(ids, X)
    ti, ts, tc = fl.VertexSearch(Q, k, gpu.SELECT_NEAREST)
    rec = np.mean([len(set(gi[q].tolist()) & set(ti[q].tolist())) / k for q in range(len(Q))])
    assert rec > 0.95, rec


def test_group_rccl_allgather_single_member_and_replica_layout(gpu):
    """(1) a 1-member group with exchange = RCCL: ncclCommInitAll + ncclAllGather really run (world 1) and the answers equal
    the plain store's; (2) REPLICA layout: the batch is split over the members, nothing is exchanged, same answers."""
    from coltt_amd import group as GG
    n, d, k = 2000, 32, 10
    X = O.fill_normal(1500, (n, d)); ids = np.arange(n, dtype=np.uint64) + np.uint64(100)
    fl = gpu.FlatSpace(d, O.L2); fl.ChangedVertex(ids, X)
    Q = O.fill_normal(1501, (37, d))
    want = fl.VertexSearch(Q, k, gpu.SELECT_NEAREST)
    g1 = gpu.Group([0], d, O.L2, kind=GG.GROUP_FLAT, exchange=GG.EXCHANGE_RCCL)
    assert g1.info()["exchange"] == "rccl"
    g1.ChangedVertex(ids, X)
    a = g1.Search(Q, k)
    assert np.array_equal(a[0], want[0]) and np.array_equal(bits(a[1]), bits(want[1])) and np.array_equal(a[2], want[2])
    with pytest.raises(gpu.ColttError):
        gpu.Group([0, 0], d, O.L2, kind=GG.GROUP_FLAT, exchange=GG.EXCHANGE_RCCL)   # a communicator cannot hold a device twice
    g3 = gpu.Group([0, 0, 0], d, O.L2, kind=GG.GROUP_FLAT, layout=GG.LAYOUT_REPLICA)
    assert g3.ChangedVertex(ids, X) == n and g3.Len() == n
    b = g3.Search(Q, k)
    assert np.array_equal(b[0], want[0]) and np.array_equal(bits(b[1]), bits(want[1])) and np.array_equal(b[2], want[2])
    # device-resident query batch per member
    import torch
    qd = torch.from_numpy(Q).cuda(); torch.cuda.synchronize()
    c = g3.SearchDevice([qd.data_ptr()] * 3, len(Q), k)
    assert np.array_equal(c[0], want[0]) and np.array_equal(bits(c[1]), bits(want[1]))


def test_group_rccl_init_rank_path_single_process(gpu):
    """the one-process-per-GPU bootstrap (coltt_group_unique_id + ncclCommInitRank), run here with a world of one: the path
    bench.py's sharded leg takes under torch.distributed.run, with torch's own RCCL communicator alive in the same process."""
    from coltt_amd import group as GG
    n, d, k = 1500, 32, 10
    X = O.fill_normal(1600, (n, d)); ids = np.arange(n, dtype=np.uint64)
    fl = gpu.FlatSpace(d, O.COSINE); fl.ChangedVertex(ids, X)
    Q = O.fill_normal(1601, (21, d))
    want = fl.VertexSearch(Q, k, gpu.SELECT_NEAREST)
    uid = GG.unique_id()
    g = gpu.Group([0], d, O.COSINE, kind=GG.GROUP_FLAT, exchange=GG.EXCHANGE_RCCL, world_size=1, rank_base=0, uid=uid)
    assert g.info()["exchange"] == "rccl" and g.info()["world"] == 1
    g.ChangedVertex(ids, X)
    a = g.Search(Q, k)
    assert np.array_equal(a[0], want[0]) and np.array_equal(bits(a[1]), bits(want[1]))
    g.close()


_TWO_PROC = r'''
import json, os, sys, numpy as np
sys.path.insert(0, {root!r})
import coltt_amd as G
from coltt_amd import group as GG
from oracle import oracle as O
rank, world, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
uid = bytes.fromhex(sys.argv[4])
assert G.lib().coltt_init(0) == 0
n, d, k = 4000, 48, 10
X = O.fill_normal(1700, (n, d)); lv = O.levels(1701, n)
ids = np.arange(n, dtype=np.uint64) * np.uint64(104729) + np.uint64(3)
Q = O.fill_normal(1702, (45, d))
if kind == "flat":
    g = G.Group([0], d, O.L2, O.Q_F16, kind=GG.GROUP_FLAT, exchange=GG.EXCHANGE_SHM, world_size=world, rank_base=rank, uid=uid)
    kept = g.ChangedVertex(ids, X)                                   # every process is OFFERED every vertex and keeps its shard's
    a = g.Search(Q, k, select=G.SELECT_NEAREST); b = g.Search(Q, k, select=G.SELECT_REFERENCE)
    g.Remove(ids[:40]); c = g.Search(Q[:7], k)
    res = {{"kept": int(kept), "len": int(g.Len()), "info": g.info(), "near": [a[0].tolist(), a[1].view(np.uint32).tolist(), a[2].tolist()],
           "ref": [b[0].tolist(), b[1].view(np.uint32).tolist(), b[2].tolist()], "after_remove": [c[0].tolist(), c[1].view(np.uint32).tolist()]}}
else:
    g = G.Group([0], d, O.COSINE, O.Q_BF16, kind=GG.GROUP_HNSW, cfg=G.HnswCfg.default(ef_construction=60), exchange=GG.EXCHANGE_SHM,
                world_size=world, rank_base=rank, uid=uid)
    kept = g.Insert(ids, X, lv, batch=1)
    a = g.Search(Q, k, ef=200)
    m = G.Hnsw.__new__(G.Hnsw); m.h = g.member(0); m.dim = d; m.quantization = O.Q_BF16; m.cfg = G.HnswCfg(); m.Config()
    gr = m.ExportRaw(); rows = m.FetchRows(); ex = m.Export()
    sl, sc, cn, _, _ = O.csr_search(rows, O.Q_BF16, gr["adj0"], gr["upper_off"], gr["adjU"], d, O.COSINE, gr["entry"], gr["entry_level"], Q, k, 200, threads=2)
    m.h = None
    mine = [[(float(sc[q, j]), int(ex["ids"][sl[q, j]])) for j in range(cn[q])] for q in range(len(Q))]   # this shard's oracle answers
    res = {{"kept": int(kept), "len": int(g.Len()), "info": g.info(), "near": [a[0].tolist(), a[1].view(np.uint32).tolist(), a[2].tolist()],
           "shard_oracle": mine, "shard_ids": sorted(int(i) for i in ex["ids"])}}
print("RESULT " + json.dumps(res))
g.close()
'''


@pytest.mark.parametrize("kind", ["flat", "hnsw"])
def test_two_processes_one_device_rank_wise_group_over_shared_memory(gpu, kind, monkeypatch):
    """The one-process-per-shard layout (world_size 2, rank_base = rank, every process offered every vertex) with BOTH processes on
    device 0 — RCCL cannot form a communicator there, so the packed per-shard top-k travel through COLTT_EXCHANGE_SHM (same records,
    same merge).  Every process must end with the answers of the unsharded store (FLAT, exact) / of the merged per-shard oracle
    walks (HNSW).  The batch fits one slot here; the CPU twin (tests/test_multi_rank.py) covers the chunked exchange."""
    import json, os, subprocess, sys
    from coltt_amd import group as GG
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    uid = GG.unique_id().hex()
    code = _TWO_PROC.format(root=root)
    env = dict(os.environ, COLTT_SHM_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2", kind, uid], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    n, d, k = 4000, 48, 10
    X = O.fill_normal(1700, (n, d)); ids = np.arange(n, dtype=np.uint64) * np.uint64(104729) + np.uint64(3)
    Q = O.fill_normal(1702, (45, d))
    sh = _shards(ids, 2)
    for r in range(2):
        assert outs[r]["info"] == {"n_local": 1, "world": 2, "exchange": "shm", "rank_base": r}
        assert outs[r]["kept"] == int((sh == r).sum())
    assert outs[0]["near"] == outs[1]["near"]           # an all-gather: both processes hold the merged answer
    gi = np.array(outs[0]["near"][0], np.uint64); gs = np.array(outs[0]["near"][1], np.uint32).view(np.float32); gc = np.array(outs[0]["near"][2])
    if kind == "flat":
        whole = O.Flat(d, O.L2, O.Q_F16); whole.upsert(ids, X)
        ri = np.array(outs[1]["ref"][0], np.uint64); rs = np.array(outs[1]["ref"][1], np.uint32).view(np.float32)
        for qi in range(len(Q)):
            wi, ws = whole.search(Q[qi], k, nearest=True, mode=2)
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"near q{qi}")
            wi, ws = whole.search(Q[qi], k, nearest=False, mode=2)
            assert_same_results(ri[qi], rs[qi], wi, ws, f"ref q{qi}")
        whole.remove(ids[:40])
        ai = np.array(outs[0]["after_remove"][0], np.uint64); as_ = np.array(outs[0]["after_remove"][1], np.uint32).view(np.float32)
        for qi in range(7):
            wi, ws = whole.search(Q[qi], k, nearest=True, mode=2)
            assert_same_results(ai[qi], as_[qi], wi, ws, f"after remove q{qi}")
        assert outs[0]["len"] + outs[1]["len"] == n - 40
    else:
        for r in range(2):
            assert outs[r]["shard_ids"] == sorted(int(i) for i in ids[sh == r])      # routing: exactly this shard's vertices
        for qi in range(len(Q)):
            u = sorted(tuple(x) for r in range(2) for x in outs[r]["shard_oracle"][qi])[:k]
            assert [(float(gs[qi, j]), int(gi[qi, j])) for j in range(gc[qi])] == [(float(np.float32(a)), int(b)) for a, b in u], qi


def test_streamed_shard_search_overlaps_exchange_and_merge_and_answers_identically(gpu):
    """8 members on one device, 10 000 queries per batch.  Batches streamed through coltt_group_search_begin / _end (the
    exchange and the host merge of batch i under the search of batch i+1, slots double-buffered) answer bit for bit like plain
    coltt_group_search calls, the sub-batched synchronous call (COLTT_GROUP_SUBBATCH shape, driven here through begin/end of slices) too,
    and the group reports where a batch's time went: exchange + merge are a small fraction of the members' search."""
    import time
    from coltt_amd import group as GG
    G, n, d, k, nq, ef = 8, 40000, 64, 10, 10000, 64
    X = O.fill_normal(1700, (n, d)); lv = O.levels(1701, n)
    ids = np.arange(n, dtype=np.uint64) * np.uint64(7919) + np.uint64(3)
    grp = gpu.Group([0] * G, d, O.COSINE, O.Q_F16, kind=GG.GROUP_HNSW, cfg=gpu.HnswCfg.default(ef_construction=60))
    assert grp.Insert(ids, X, lv, batch=512) == n
    Qs = [O.fill_normal(1710 + b, (nq, d)) for b in range(4)]
    plain = [grp.Search(q, k, ef=ef) for q in Qs]
    t0 = grp.Timing()
    # streamed: begin b+1 before ending b
    w0 = time.time()
    pend = []
    outs = []
    for q in Qs:
        pend.append(grp.SearchBegin(k, queries=q, ef=ef))
        if len(pend) == 2:
            t, o = pend.pop(0); grp.SearchEnd(t); outs.append(o)
    while pend:
        t, o = pend.pop(0); grp.SearchEnd(t); outs.append(o)
    wall_ms = (time.time() - w0) * 1e3
    for a, b in zip(plain, outs):
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
    t1 = grp.Timing()
    nb = t1["batches"] - t0["batches"]
    assert nb == len(Qs)
    search = (t1["search_ms"] - t0["search_ms"]) / nb; exch = (t1["exchange_ms"] - t0["exchange_ms"]) / nb; merge = (t1["merge_ms"] - t0["merge_ms"]) / nb
    print(f"\n[group pipeline] per batch of {nq} x {G} shards: search {search:.2f} ms, exchange {exch:.2f} ms, merge {merge:.2f} ms; "
          f"streamed wall {wall_ms / nb:.2f} ms per batch")
    assert merge < max(6.0, 0.25 * search)      # the merge is split over host threads (it competes with eight member threads for the box's CPU quota): a few ms for 10 000 x 8 x 10 records
    # what is NOT hidden: the streamed loop's wall per batch exceeds the search stage by less than the un-overlapped exchange + merge would
    assert wall_ms / nb < search + exch + merge + 1.0
    # slices of one batch through the pipeline == the whole batch (the COLTT_GROUP_SUBBATCH shape)
    q = Qs[0]; parts = []; tickets = []
    for lo in range(0, nq, 2500):
        t, o = grp.SearchBegin(k, queries=q[lo:lo + 2500], ef=ef); tickets.append(t); parts.append(o)
        if len(tickets) == 3:
            grp.SearchEnd(tickets.pop(0))
    for t in tickets: grp.SearchEnd(t)
    assert np.array_equal(np.concatenate([p[0] for p in parts]), plain[0][0]) and np.array_equal(bits(np.concatenate([p[1] for p in parts])), bits(plain[0][1]))
    # argument errors do not consume tickets or slots
    with pytest.raises(gpu.ColttError):
        grp.SearchBegin(0, queries=q[:4], ef=ef)
    again = grp.Search(Qs[1], k, ef=ef)
    assert np.array_equal(again[0], plain[1][0])


_FAIL_PROC = r'''
import json, os, sys, time, numpy as np
sys.path.insert(0, {root!r})
import coltt_amd as G
from coltt_amd import group as GG
from oracle import oracle as O
rank, world = int(sys.argv[1]), int(sys.argv[2]); uid = bytes.fromhex(sys.argv[3])
assert G.lib().coltt_init(0) == 0
n, d, k = 1500, 32, 10
X = O.fill_normal(1800, (n, d)); ids = np.arange(n, dtype=np.uint64); Q = O.fill_normal(1801, (20, d))
g = G.Group([0], d, O.L2, kind=GG.GROUP_FLAT, exchange=GG.EXCHANGE_SHM, world_size=world, rank_base=rank, uid=uid)
g.ChangedVertex(ids, X)
a = g.Search(Q, k)                                         # a healthy batch first: both processes answer
os.environ["COLTT_TEST_FAIL_STAGE_A"] = "1"                # from now on rank 1's shard search fails before it searches anything
t0 = time.time(); err = None
try:
    g.Search(Q, k)
except G.ColttError as e:
    err = str(e)
dt = time.time() - t0
os.environ.pop("COLTT_TEST_FAIL_STAGE_A")
b = g.Search(Q, k)                                         # the group is still usable: the failure was that batch's, not the transport's
print("RESULT " + json.dumps({{"err": err, "dt": dt, "same": bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)))}}))
g.close()
'''


def test_a_rank_that_fails_before_the_exchange_releases_its_peer_at_once_over_shared_memory(gpu):
    """two processes, rank 1's shard search fails (test hook) — it still contributes a block of status records, BOTH processes return an
    error for that batch within seconds (nobody waits out the 120 s timeout), and the next batch runs as if nothing had happened."""
    import json, os, subprocess, sys
    from coltt_amd import group as GG
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    uid = GG.unique_id().hex()
    code = _FAIL_PROC.format(root=root)
    env = dict(os.environ, COLTT_SHM_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2", uid], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    for r in range(2):
        assert outs[r]["err"] is not None and outs[r]["dt"] < 5.0, outs[r]
        assert outs[r]["same"], "the batch after the failed one must answer as before"
    assert "rank 1" in outs[0]["err"], outs[0]["err"]                       # the healthy rank names the one that failed
    assert "injected stage-A failure" in outs[1]["err"], outs[1]["err"]     # the failing rank reports its own error


def test_rccl_exchange_survives_a_failed_shard_search_and_bounds_its_wait(gpu, monkeypatch):
    """the same on the RCCL transport, one process (world 1): the failing rank still issues its ncclAllGather — of status records — and returns its
    error; the communicator is intact for the next batch.  (A peer that never issues its all-gather at all is what the deadline on the comm stream is
    for: COLTT_EXCHANGE_TIMEOUT_S; it cannot be provoked with one rank.)"""
    from coltt_amd import group as GG
    n, d, k = 1200, 32, 10
    X = O.fill_normal(1900, (n, d)); ids = np.arange(n, dtype=np.uint64); Q = O.fill_normal(1901, (16, d))
    g = gpu.Group([0], d, O.L2, kind=GG.GROUP_FLAT, exchange=GG.EXCHANGE_RCCL)
    g.ChangedVertex(ids, X)
    a = g.Search(Q, k)
    monkeypatch.setenv("COLTT_TEST_FAIL_STAGE_A", "0")
    with pytest.raises(gpu.ColttError) as ei:
        g.Search(Q, k)
    assert "injected stage-A failure" in str(ei.value)
    t, out = g.SearchBegin(k, queries=Q)                    # streaming form: _begin hands the batch over, _end returns the batch's error
    with pytest.raises(gpu.ColttError):
        g.SearchEnd(t)
    monkeypatch.delenv("COLTT_TEST_FAIL_STAGE_A")
    b = g.Search(Q, k)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    g.close()
