"""The selection scheme of flat_one_kernel (coltt_amd/csrc/flat.hip: one launch for <= 4 queries) modelled in numpy — runs without a
GPU.  The kernel's claim: if every block publishes its k best records and lowers bucket[block % k] to its BEST key, then
    bound = max over the k buckets (an untouched bucket counts as +inf)
is an upper bound of the collection's k-th best key, so the records with key <= bound contain the exact top-k under the canonical
(key, id) order (edge/priority_queue.go:39-69 in closed form: DESIGN.md §4).  The model checks exactly that — with ties, with fewer
blocks than k, with blocks shorter than k — and that the bound is tight enough for the short-list path of the selection.
The GPU twin is tests/test_gpu_round3.py::test_small_batches_in_one_launch_equal_the_oracle."""
import numpy as np
import pytest


def one_launch_model(keys, ids, k, rows_per_block):
    n = len(keys)
    order = np.lexsort((ids, keys))                      # the canonical total order: key, then id
    want = order[:min(k, n)]
    nblocks = max(1, -(-n // rows_per_block))
    bucket = np.full(k, np.iinfo(np.uint64).max, np.uint64)
    records = []
    for b in range(nblocks):
        lo, hi = b * rows_per_block, min(n, (b + 1) * rows_per_block)
        if lo >= hi:
            continue
        loc = lo + np.lexsort((ids[lo:hi], keys[lo:hi]))[:k]      # the block's k best, in order (per-wave lists merged)
        records.extend(loc.tolist())
        bucket[b % k] = min(bucket[b % k], np.uint64(keys[loc[0]]))
    bound = bucket.max()
    passed = np.array([r for r in records if np.uint64(keys[r]) <= bound], np.int64)
    final = passed[np.lexsort((ids[passed], keys[passed]))][:k]  # the ordinary selection over the survivors
    return want, final, len(passed), len(records)


@pytest.mark.parametrize("n,k,rows_per_block", [(100000, 10, 256), (5000, 64, 256), (700, 10, 256), (31, 10, 256), (3, 10, 256),
                                                (70000, 1, 256), (4096, 33, 64), (20000, 10, 2048)])
@pytest.mark.parametrize("distinct", [1 << 30, 50, 1])
def test_bucket_bound_never_loses_a_member_of_the_top_k(n, k, rows_per_block, distinct):
    rng = np.random.default_rng(n * 131 + k * 7 + distinct % 97)
    keys = rng.integers(0, distinct, n).astype(np.uint32)          # distinct = 50 / 1: massive ties, the ids decide
    ids = rng.permutation(n).astype(np.uint64) * np.uint64(2654435761) % np.uint64(1 << 40)   # not in slot order
    want, final, n_pass, n_rec = one_launch_model(keys, ids, k, rows_per_block)
    assert np.array_equal(want, final)
    if distinct > 1 << 20 and n >= 64 * k * 4 and rows_per_block == 256:   # random keys: the survivors are a short list (k ln k-ish, not ~1 % of n)
        assert n_pass <= 512, (n_pass, n_rec)


def test_blocks_own_kth_best_alone_would_be_a_loose_bound():
    """Why the buckets exist: the minimum over blocks of each block's k-th best (the first design) lets ~1 % of the rows through —
    1 500 records for 100 k rows, a radix select in the last block (35 us on the GPU); the bucket maximum passes a few dozen."""
    rng = np.random.default_rng(5)
    n, k, rpb = 100000, 10, 256
    keys = rng.integers(0, 1 << 30, n).astype(np.uint32)
    blocks = [np.sort(keys[b:b + rpb])[:k] for b in range(0, n, rpb)]
    loose = min(b[-1] for b in blocks if len(b) == k)
    n_loose = sum(int((b <= loose).sum()) for b in blocks)
    ids = np.arange(n, dtype=np.uint64)
    _, _, n_tight, _ = one_launch_model(keys, ids, k, rpb)
    assert n_loose > 500 and n_tight < 100, (n_loose, n_tight)


# ---- the two register/LDS merges of round 3, modelled lane by lane (what every lane computes, then one scatter) ----------------

def parallel_merge_model(entries, cands, cap=64):
    """flat_one_kernel's per-wave list update: `entries` sorted ascending (<= 64), `cands` unordered; every existing entry counts the
    candidates in front of it, every candidate its rank among the candidates plus the entries in front of it — final positions."""
    out = [None] * cap
    for i, e in enumerate(entries):
        pe = i + sum(1 for c in cands if c < e)
        if pe < cap:
            assert out[pe] is None
            out[pe] = e
    for c in cands:
        pc = sum(1 for d in cands if d < c) + sum(1 for e in entries if e < c)
        if pc < cap:
            assert out[pc] is None          # positions are a permutation: no two writers per slot
            out[pc] = c
    n = min(cap, len(entries) + len(cands))
    return out[:n]


def small_set_merge_model(res, admitted, ef):
    """hnsw_walk2.hpp, ef <= 128: members shift by the number of admitted keys not behind them, an admitted key lands at
    (#members in front of it) + (its rank among the admitted); entries pushed past ef are dropped."""
    out = {}
    for i, e in enumerate(res):
        np_ = i + sum(1 for a in admitted if not (e < a))
        if np_ < ef:
            assert np_ not in out
            out[np_] = e
    for a in admitted:
        np_ = sum(1 for e in res if e < a) + sum(1 for b in admitted if b < a)
        if np_ < ef:
            assert np_ not in out
            out[np_] = a
    n = min(ef, len(res) + len(admitted))
    return [out[i] for i in range(n)]


@pytest.mark.parametrize("seed", range(40))
def test_merge_models_equal_a_sort(seed):
    rng = np.random.default_rng(seed)
    universe = rng.permutation(5000)                      # distinct keys (the kernels compare (key, id) pairs: never equal)
    n_e, n_c = int(rng.integers(0, 65)), int(rng.integers(0, 33))
    entries = sorted(universe[:n_e].tolist()); cands = universe[n_e:n_e + n_c].tolist()
    assert parallel_merge_model(entries, cands) == sorted(entries + cands)[:64]
    ef = int(rng.choice([1, 7, 20, 64, 100, 128]))
    res = sorted(universe[1000:1000 + int(rng.integers(0, ef + 1))].tolist())
    adm = universe[2000:2000 + int(rng.integers(0, 33))].tolist()
    assert small_set_merge_model(res, adm, ef) == sorted(res + adm)[:ef]
