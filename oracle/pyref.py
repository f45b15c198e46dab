"""pyref.py — SECOND, INDEPENDENT CPU restatement of the reference's HNSW and edge queue, in pure Python.  TEST INFRASTRUCTURE ONLY.

Written from the Go text (not from oracle/coltt_oracle.cpp) so that a misreading in one restatement shows up as a
disagreement between the two (VERDICT r1 "Next round" #7): tests/test_oracle.py asserts C++ oracle == this file on random
configurations incl. removals, and tests/golden/hnsw_pyref.npz (made by tests/golden/make_golden_pyref.py FROM THIS FILE)
pins both.  The Go toolchain is absent, so neither restatement can be checked against a run of the reference itself: parity
above the SIMD kernels remains "unpinned by the reference", now with two independent readers instead of one.

Followed line by line (all paths under /root/reference):
  core/vectorindex/hnsw.go             Insert :104-167, Remove :191-241, Search :243-278, greedyClosestNeighbor :320-343,
                                       searchLevel :345-389, selectNeighbors :391-397, selectNeighborsHeuristic :399-447,
                                       pruneNeighbors :449-474
  core/vectorindex/priority_queue.go   min / max queues over container/heap :57-199 (Peek = slice[0], Reverse re-types the SAME
                                       backing array and heap.Init's it :109-122)
  core/vectorindex/hnsw_vertex.go      edge sets, deleted flag :27-127
  core/vectorindex/metadata.go         Normalize :107-123
  pkg/distance/space.go :61-95, simd/avx/AVX_amd64.go :26-52, simd/cpp/avx.cpp :4-75   (8-lane sums, no FMA, hadd tree, tail)
  edge/priority_queue.go :33-69, edge/priorityqueue/priority_queue.go                      (bounded queue: min-heap, pop-min)
  Go 1.23 container/heap: up / down / Init / Push / Pop (stdlib; restated from its published algorithm)

The ONE deliberate substitution: Go's map iteration order is random; every `for x := range someMap` here walks ascending
vertex insertion index (the canonical order of DESIGN.md §4).  Everything else — stale lowerBound, heap sibling order, the
Heuristic path's Reverse()/heap.Init — is literal.  Arithmetic is numpy float32 with the exact operation order.
"""
import numpy as np

f32 = np.float32


# ------------------------------------------------------------------------------------------------ container/heap
class GoHeap:
    """heap.Interface over a Python list `a` of (priority f32, value); less(i, j) decides min or max."""

    def __init__(self, is_max, items=None):
        self.is_max = is_max
        self.a = items if items is not None else []

    def less(self, i, j):
        return self.a[i][0] > self.a[j][0] if self.is_max else self.a[i][0] < self.a[j][0]

    def _up(self, j):
        while True:
            i = int((j - 1) / 2)          # Go integer division truncates toward zero: (0-1)/2 == 0
            if i == j or not self.less(j, i):
                break
            self.a[i], self.a[j] = self.a[j], self.a[i]
            j = i

    def _down(self, i0, n):
        i = i0
        while True:
            j1 = 2 * i + 1
            if j1 >= n or j1 < 0:
                break
            j = j1
            j2 = j1 + 1
            if j2 < n and self.less(j2, j1):
                j = j2
            if not self.less(j, i):
                break
            self.a[i], self.a[j] = self.a[j], self.a[i]
            i = j
        return i > i0

    def init(self):
        n = len(self.a)
        for i in range(n // 2 - 1, -1, -1):
            self._down(i, n)

    def push(self, item):
        self.a.append(item)
        self._up(len(self.a) - 1)

    def pop(self):
        n = len(self.a) - 1
        self.a[0], self.a[n] = self.a[n], self.a[0]
        self._down(0, n)
        return self.a.pop()

    def peek(self):
        return self.a[0]

    def __len__(self):
        return len(self.a)

    def reverse(self):
        """priorityQueue.Reverse (priority_queue.go:109-122): the opposite heap over the SAME backing array, heap.Init'ed."""
        h = GoHeap(not self.is_max, self.a)   # same list object on purpose
        h.init()
        return h


# ------------------------------------------------------------------------------------------------ distances (AVX path)
def _hsum8(v):
    """_sum_vector (avx.cpp:4-9): hadd, hadd, [0] + [4]  ==  ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7)) in f32"""
    return f32(f32(f32(v[0] + v[1]) + f32(v[2] + v[3])) + f32(f32(v[4] + v[5]) + f32(v[6] + v[7])))


def euclidean(a, b):
    n8 = (len(a) // 8) * 8
    acc = np.zeros(8, f32)
    for i in range(0, n8, 8):
        d = a[i:i + 8] - b[i:i + 8]          # vsubps
        acc = acc + d * d                     # vmulps, vaddps (two roundings)
    r = _hsum8(acc)
    for i in range(n8, len(a)):
        d = f32(a[i] - b[i])
        r = f32(r + f32(d * d))
    return f32(np.sqrt(np.float64(r)))        # AVX_amd64.go:31


def cosine(a, b):
    n8 = (len(a) // 8) * 8
    dot = np.zeros(8, f32); na = np.zeros(8, f32); nb = np.zeros(8, f32)
    for i in range(0, n8, 8):
        x = a[i:i + 8]; y = b[i:i + 8]
        dot = dot + x * y
        na = na + x * x
        nb = nb + y * y
    d = _hsum8(dot); sa = _hsum8(na); sb = _hsum8(nb)
    for i in range(n8, len(a)):
        d = f32(d + f32(a[i] * b[i]))
        sa = f32(sa + f32(a[i] * a[i]))
        sb = f32(sb + f32(b[i] * b[i]))
    nsq = f32(sa * sb)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = f32(f32(1.0) - f32(d / f32(np.sqrt(np.float64(nsq)))))   # AVX_amd64.go:51
    return f32(np.abs(np.float64(r)))                                 # Cosine.Distance: gomath.Abs (space.go:93-95)


def normalize(v):
    """metadata.go:107-123: sequential f32 sum of squares, float32(math.Sqrt(float64)), element-wise divide"""
    v = np.asarray(v, f32)
    norm = f32(0)
    for x in v:
        norm = f32(norm + f32(x * x))
    out = np.zeros(len(v), f32)
    if norm == 0:
        return out
    norm = f32(np.sqrt(np.float64(norm)))
    for i in range(len(v)):
        out[i] = f32(v[i] / norm)
    return out


# ------------------------------------------------------------------------------------------------ HNSW
class Vertex:
    __slots__ = ("id", "vector", "level", "deleted", "edges", "index")

    def __init__(self, id_, vector, level, index):
        self.id, self.vector, self.level, self.deleted, self.index = id_, vector, level, False, index
        self.edges = [dict() for _ in range(level + 1)]     # hnswEdgeSet per level: {vertex index: distance}


class Hnsw:
    COSINE, L2 = 0, 1

    def __init__(self, dim, metric, m=16, m_max=-1, m_max0=-1, ef=20, ef_construction=200, algo=0, keep_pruned=True):
        self.dim, self.metric = dim, metric
        self.m, self.ef, self.efc, self.algo, self.keep_pruned = m, ef, ef_construction, algo, keep_pruned
        self.m_max = m if m_max == -1 else m_max            # newHnswConfig (hnsw_config.go:150-160)
        self.m_max0 = 2 * m if m_max0 == -1 else m_max0
        self.v = []                                          # insertion order == canonical order
        self.by_id = {}
        self.entry = None
        self.n_dist = 0

    def dist(self, a, b):
        self.n_dist += 1
        return cosine(a, b) if self.metric == self.COSINE else euclidean(a, b)

    def _nbrs(self, vertex, level):
        """`for neighbor, d := range vertex.edges[level]` in the canonical (ascending insertion index) order"""
        return [(self.v[i], d) for i, d in sorted(vertex.edges[level].items())]

    # hnsw.go:104-167
    def insert(self, id_, value, vertex_level):
        value = np.asarray(value, f32)
        if self.metric == self.COSINE:
            value = normalize(value)
        if id_ in self.by_id:
            return "ItemAlreadyExistsError"                                  # storeVertex :293-295
        if self.entry is None:
            vertex = Vertex(id_, value, 0, len(self.v))                       # :109-110 level forced to 0
            self.v.append(vertex); self.by_id[id_] = vertex
            self.entry = vertex                                               # CAS nil -> vertex succeeds (single thread)
            return None
        vertex = Vertex(id_, value, vertex_level, len(self.v))
        self.v.append(vertex); self.by_id[id_] = vertex
        entrypoint = self.entry
        min_distance = self.dist(vertex.vector, entrypoint.vector)
        for l in range(entrypoint.level, vertex.level, -1):
            entrypoint, min_distance = self.greedy(vertex.vector, entrypoint, min_distance, l)
        for l in range(min(entrypoint.level, vertex.level), -1, -1):
            neighbors = self.search_level(vertex.vector, entrypoint, self.efc, l)
            if self.algo == 0:
                neighbors = self.select_neighbors(neighbors, self.m)
            else:
                neighbors = self.select_heuristic(vertex.vector, neighbors, self.m, l)
            m_max = self.m_max0 if l == 0 else self.m_max
            while len(neighbors) > 0:
                prio, neighbor = neighbors.pop()
                entrypoint = neighbor
                vertex.edges[l][neighbor.index] = prio                        # addEdge both ways
                neighbor.edges[l][vertex.index] = prio
                if len(neighbor.edges[l]) > m_max:
                    self.prune(neighbor, m_max, l)
        if self.entry is not None and vertex.level > self.entry.level:
            self.entry = vertex
        return None

    # hnsw.go:191-241
    def remove(self, id_):
        vertex = self.by_id.pop(id_, None)
        if vertex is None:
            return "ItemNotFoundError"
        vertex.deleted = True
        if self.entry is vertex:
            min_distance = f32(np.finfo(np.float32).max)
            closest = None
            for l in range(vertex.level, -1, -1):
                for neighbor, distance in self._nbrs(vertex, l):
                    if distance < min_distance:
                        min_distance = distance
                        closest = neighbor
                if closest is not None:
                    break
            self.entry = closest
        for l in range(vertex.level, -1, -1):
            m_max = self.m_max0 if l == 0 else self.m_max
            for neighbor, _ in self._nbrs(vertex, l):
                neighbor.edges[l].pop(vertex.index, None)                     # removeEdge
                self.prune(neighbor, m_max, l)
        return None

    # hnsw.go:243-278
    def search(self, query, k, ef_override=0):
        query = np.asarray(query, f32)
        if self.metric == self.COSINE:
            query = normalize(query)
        entrypoint = self.entry
        if entrypoint is None:
            return []
        min_distance = self.dist(query, entrypoint.vector)
        for l in range(entrypoint.level, 0, -1):
            entrypoint, min_distance = self.greedy(query, entrypoint, min_distance, l)
        ef = max(ef_override or self.ef, k)
        neighbors = self.search_level(query, entrypoint, ef, 0)
        if self.algo == 0:
            neighbors = self.select_neighbors(neighbors, k)
        else:
            neighbors = self.select_heuristic(query, neighbors, k, 0)
        n = min(k, len(neighbors))
        result = [None] * n
        for i in range(n - 1, -1, -1):
            prio, vtx = neighbors.pop()
            result[i] = (vtx.id, prio)
        return result

    # hnsw.go:320-343
    def greedy(self, query, entrypoint, min_distance, level):
        while True:
            closest = None
            for neighbor, _ in self._nbrs(entrypoint, level):
                if neighbor.deleted:
                    continue
                distance = self.dist(query, neighbor.vector)
                if distance < min_distance:
                    min_distance = distance
                    closest = neighbor
            if closest is None:
                break
            entrypoint = closest
        return entrypoint, min_distance

    # hnsw.go:345-389
    def search_level(self, query, entrypoint, ef, level):
        ep_distance = self.dist(query, entrypoint.vector)
        item = (ep_distance, entrypoint)
        candidates = GoHeap(False); candidates.push(item)
        results = GoHeap(True); results.push(item)
        visited = {entrypoint.index}
        while len(candidates) > 0:
            c_prio, candidate = candidates.pop()
            lower_bound = results.peek()[0]                                   # read ONCE per popped candidate (:357)
            if c_prio > lower_bound:
                break
            for neighbor, _ in self._nbrs(candidate, level):
                if neighbor.deleted:
                    continue
                if neighbor.index in visited:
                    continue
                visited.add(neighbor.index)
                distance = self.dist(query, neighbor.vector)
                if distance < lower_bound or len(results) < ef:              # stale lowerBound (:374)
                    it = (distance, neighbor)
                    candidates.push(it)
                    results.push(it)
                    if len(results) > ef:
                        results.pop()
        return results

    # hnsw.go:391-397
    @staticmethod
    def select_neighbors(neighbors, k):
        while len(neighbors) > k:
            neighbors.pop()
        return neighbors

    # hnsw.go:399-447 with extendCandidates == false (true is undefined behaviour in the reference: shared backing array)
    def select_heuristic(self, query, neighbors, k, level):
        candidates = neighbors.reverse()                                     # MinPriorityQueue over the same array
        result = GoHeap(True)
        while len(candidates) > 0 and len(result) < k:
            result.push(candidates.pop())
        if self.keep_pruned:                                                  # dead loop: result.Len() >= k or candidates empty
            while len(candidates) > 0:
                if len(result) >= k:
                    break
                result.push(candidates.pop())
        return result

    # hnsw.go:449-474
    def prune(self, vertex, k, level):
        queue = GoHeap(True)
        for neighbor, distance in self._nbrs(vertex, level):
            if neighbor.deleted:
                continue
            queue.push((distance, neighbor))
        if self.algo == 0:
            queue = self.select_neighbors(queue, k)
        else:
            queue = self.select_heuristic(vertex.vector, queue, k, level)
        vertex.edges[level] = {vtx.index: prio for prio, vtx in queue.a}     # setEdges(ToSlice())

    # ---- export in the C++ oracle's layout (for cross-checks)
    def export(self):
        ids = np.array([v.id for v in self.v], np.uint64)
        levels = np.array([v.level for v in self.v], np.int32)
        deleted = np.array([v.deleted for v in self.v], np.uint8)
        offs, nbr, nd = [0], [], []
        for v in self.v:
            for l in range(v.level + 1):
                for i, d in sorted(v.edges[l].items()):
                    nbr.append(i); nd.append(d)
                offs.append(len(nbr))
        return {"ids": ids, "levels": levels, "deleted": deleted, "row_offsets": np.array(offs, np.int64),
                "nbr": np.array(nbr, np.int32), "nbr_dist": np.array(nd, np.float32), "entry": -1 if self.entry is None else self.entry.index}


# ------------------------------------------------------------------------------------------------ algo 2, "diverse"
class DiverseHnsw(Hnsw):
    """COLTT_HNSW_DIVERSE — NOT reference behaviour (the reference's selectNeighborsHeuristic, hnsw.go:399-447, never compares a candidate
    with the neighbours already chosen).  A DEFINITION, restated here independently of oracle/coltt_oracle.cpp (select_diverse, prune,
    hnsw_insert): Insert and pruneNeighbors-on-overflow choose neighbours by the diversity test of the HNSW paper (Algorithm 4, as hnswlib
    runs it).  Everything else — greedy descent, searchLevel (the literal Go-heap form above), Remove, Search — is the class above."""

    def __init__(self, dim, metric, keep_pruned=False, **kw):
        super().__init__(dim, metric, algo=1, keep_pruned=keep_pruned, **kw)   # algo 1: Search's final selection = the k nearest

    @staticmethod
    def _order(item):
        d, vtx = item
        return (int(np.float32(d).view(np.uint32)), vtx.index)                # (distance bits, slot): the canonical result-set order

    def select_diverse(self, cands, k):
        """cands: (distance to the base vertex, vertex), any order.  Walk them ascending while fewer than k are chosen; c is chosen iff no
        already chosen r has dist(c as the query, r) < c's distance; EVERY chosen r is evaluated.  keep_pruned: the rejected, nearest first,
        fill the result up to k.  Returns the chosen list (chosen first, then the re-added)."""
        chosen, pruned = [], []
        for d, c in sorted(cands, key=self._order):
            if len(chosen) >= k:
                break
            good = True
            for _, r in chosen:
                if self.dist(c.vector, r.vector) < d:
                    good = False
            (chosen if good else pruned).append((d, c))
        if self.keep_pruned:
            for it in pruned:
                if len(chosen) >= k:
                    break
                chosen.append(it)
        return chosen

    def insert(self, id_, value, vertex_level):
        value = np.asarray(value, f32)
        if self.metric == self.COSINE:
            value = normalize(value)
        if id_ in self.by_id:
            return "ItemAlreadyExistsError"
        if self.entry is None:
            vertex = Vertex(id_, value, 0, len(self.v))
            self.v.append(vertex); self.by_id[id_] = vertex
            self.entry = vertex
            return None
        vertex = Vertex(id_, value, vertex_level, len(self.v))
        self.v.append(vertex); self.by_id[id_] = vertex
        entrypoint = self.entry
        min_distance = self.dist(vertex.vector, entrypoint.vector)
        for l in range(entrypoint.level, vertex.level, -1):
            entrypoint, min_distance = self.greedy(vertex.vector, entrypoint, min_distance, l)
        for l in range(min(entrypoint.level, vertex.level), -1, -1):
            found = self.search_level(vertex.vector, entrypoint, self.efc, l)
            chosen = self.select_diverse(list(found.a), self.m)
            m_max = self.m_max0 if l == 0 else self.m_max
            for prio, neighbor in chosen:
                vertex.edges[l][neighbor.index] = prio
                neighbor.edges[l][vertex.index] = prio
                if len(neighbor.edges[l]) > m_max:
                    self.prune(neighbor, m_max, l)
            entrypoint = chosen[0][1]                                         # the nearest candidate is always chosen first
        if self.entry is not None and vertex.level > self.entry.level:
            self.entry = vertex
        return None

    def prune(self, vertex, k, level):
        live = [(d, n) for n, d in self._nbrs(vertex, level) if not n.deleted]
        if len(live) > k:                                                     # the selection runs on overflow only (hnswlib's rule)
            live = self.select_diverse(live, k)
        vertex.edges[level] = {n.index: d for d, n in live}


# ------------------------------------------------------------------------------------------------ edge bounded queue
def edge_queue(scores_ids, max_size):
    """edge.PriorityQueue (edge/priority_queue.go:33-69): Add = push into a MIN-heap, pop the minimum when over capacity (keeps
    the K LARGEST scores); ToSlice sorts ascending by Score.  scores_ids: iterable of (score f32, id) in scan order.
    sort.Slice is unstable: ties are returned in ascending (score, id) here (the canonical order)."""
    h = GoHeap(False)
    for s, i in scores_ids:
        h.push((f32(s), i))
        if len(h) > max_size:
            h.pop()
    return sorted(h.a, key=lambda t: (t[0], t[1]))


# ------------------------------------------------------------------------------------------------
# pkg/distancepq and the product-quantiser scan defined on top of it — an INDEPENDENT restatement,
# written from asm/dot.s:7-55, asm/euclidean.s:7-65 and distance.go:30-42 alone, with the fused multiply-adds evaluated in EXACT
# rational arithmetic and rounded once (round-to-nearest-even to binary32), so it shares no code path with the C++ oracle's
# std::fmaf.  Pure Python: small cases only.
# ------------------------------------------------------------------------------------------------
from fractions import Fraction as _Fr


def _round_f32(x):
    """a non-zero exact rational -> the nearest binary32 (ties to even), subnormals and overflow included"""
    neg = x < 0
    a = -x if neg else x
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if _Fr(2) ** e > a:
        e -= 1
    elif _Fr(2) ** (e + 1) <= a:
        e += 1
    q = max(e, -126) - 23                  # the spacing of binary32 around a is 2^q (2^-149 below the normal range)
    n = a / _Fr(2) ** q
    f = n.numerator // n.denominator
    rem = n - f
    if rem > _Fr(1, 2) or (rem == _Fr(1, 2) and (f & 1)):
        f += 1
    v = _Fr(f) * _Fr(2) ** q
    r = np.float32(np.inf) if v >= _Fr(2) ** 128 else np.float32(float(v))   # v is representable: float() is exact
    return -r if neg else r


def fma32(a, b, c):
    """fused multiply-add on binary32 operands: a*b + c exactly, ONE rounding (VFMADD231PS / VFMADD231SS)"""
    a, b, c = np.float32(a), np.float32(b), np.float32(c)
    if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
        return np.float32(np.float64(a) * np.float64(b) + np.float64(c))   # inf / nan propagate the same way through any order
    ex = _Fr(float(a)) * _Fr(float(b)) + _Fr(float(c))
    if ex == 0:
        pneg = bool(np.signbit(a)) != bool(np.signbit(b))
        if (a == 0 or b == 0) and c == 0 and pneg and bool(np.signbit(c)):
            return np.float32(-0.0)        # (-0) + (-0)
        return np.float32(0.0)             # exact cancellation, or zeros of unlike sign: +0 under round-to-nearest
    return _round_f32(ex)


def _pq_reduce(acc, tail):
    f = np.float32
    s = [f(f(f(acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j]) for j in range(8)]      # VADDPS Y0,Y1 ; +Y2 ; +Y3
    t = [f(s[j] + s[j + 4]) for j in range(4)]                                         # VEXTRACTF128 $1 ; VADDPS X0, X1
    t = [f(t[0] + tail), f(t[1] + f(0)), f(t[2] + f(0)), f(t[3] + f(0))]               # VADDPS X0, X4   (X4 = {tail, 0, 0, 0})
    return f(f(t[0] + t[1]) + f(t[2] + t[3]))                                          # VHADDPS ; VHADDPS


def pq_dot(x, y):
    """asm.Dot (pkg/distancepq/asm/dot.s:7-55)"""
    x = np.asarray(x, np.float32); y = np.asarray(y, np.float32)
    acc = [[np.float32(0)] * 8 for _ in range(4)]
    i = 0
    while len(x) - i >= 32:                                     # blockloop: Y0..Y3 += x[i+8r .. i+8r+7] * y[...]
        for r in range(4):
            for j in range(8):
                acc[r][j] = fma32(x[i + 8 * r + j], y[i + 8 * r + j], acc[r][j])
        i += 32
    tail = np.float32(0)
    while i < len(x):                                           # tailloop: VFMADD231SS into lane 0 of X4
        tail = fma32(x[i], y[i], tail); i += 1
    return _pq_reduce(acc, tail)


def pq_l2sq(x, y):
    """asm.SquaredEuclideanDistance (pkg/distancepq/asm/euclidean.s:7-65): VSUBPS then VFMADD231PS d, d, acc"""
    x = np.asarray(x, np.float32); y = np.asarray(y, np.float32)
    acc = [[np.float32(0)] * 8 for _ in range(4)]
    i = 0
    while len(x) - i >= 32:
        for r in range(4):
            for j in range(8):
                d = np.float32(x[i + 8 * r + j] - y[i + 8 * r + j])
                acc[r][j] = fma32(d, d, acc[r][j])
        i += 32
    tail = np.float32(0)
    while i < len(x):
        d = np.float32(x[i] - y[i]); tail = fma32(d, d, tail); i += 1
    return _pq_reduce(acc, tail)


def pq_fn(metric, x, y):
    """distance.go:30-42: 0 cosineDistance = 1 - dot, 1 euclideanDistance = squared L2, 2 dotProductDistance = -dot"""
    if metric == 1:
        return pq_l2sq(x, y)
    d = pq_dot(x, y)
    return np.float32(np.float32(1) - d) if metric == 0 else np.float32(-d)


def pq_search(metric, codebooks, vectors, ids, query, k):
    """the product-quantiser scan as DEFINED in oracle/coltt_oracle.cpp ("Product quantiser"): Encode every vector, build the query's
    table, sum it in sub-vector order, keep the k smallest by (score bits, id).  Returns codes, lut, ids, scores."""
    cb = np.asarray(codebooks, np.float32); m, c, ds = cb.shape
    codes = np.zeros((len(vectors), m), np.uint8)
    for i, v in enumerate(np.asarray(vectors, np.float32)):
        for j in range(m):
            best, md = 0, np.float32(3.4028234663852886e38)
            for cc in range(c):
                d = pq_l2sq(v[j * ds:(j + 1) * ds], cb[j, cc])
                if d < md:
                    best, md = cc, d
            codes[i, j] = best
    q = np.asarray(query, np.float32)
    lut = np.array([[pq_fn(metric, q[j * ds:(j + 1) * ds], cb[j, cc]) for cc in range(c)] for j in range(m)], np.float32)
    scored = []
    for i in range(len(codes)):
        dist = np.float32(0)
        for j in range(m):
            dist = np.float32(dist + lut[j, codes[i, j]])
        u = int(np.float32(dist).view(np.uint32))
        key = (~u & 0xFFFFFFFF) if (u & 0x80000000) else (u | 0x80000000)
        scored.append((key, int(ids[i]), dist))
    scored.sort(key=lambda t: (t[0], t[1]))
    top = scored[:k]
    return codes, lut, np.array([t[1] for t in top], np.uint64), np.array([t[2] for t in top], np.float32)


def csr_search_pq(rows_seen, adj0, upper_off, adj_u, metric, entry, entry_level, codes, codebooks, pq_metric, query_seen, k, ef, rerank=0):
    """Product-quantised Hnsw.Search as DEFINED in oracle/coltt_oracle.cpp ("Product-quantised HNSW"), restated independently in plain
    Python over the padded-array graph (adj0 [n][w0], upper_off [n], adj_u [rows][wu], 0xffffffff padded): table distance
    d = S_lo + S_hi, the two half-row sums of float32(binary16(lut[j][code[j]])) (f32 adds, j order within a half) in place of Distance() for the entrypoint (hnsw.go:253), greedyClosestNeighbor (:320-343)
    and searchLevel(ef) (:345-389, canonical closed form: stale lowerBound per pop, ascending-slot neighbour order, ties by (d, slot));
    then the r = min(max(rerank, k), len) nearest (rerank = 0: all) re-scored with the exact distance (AVX order) and the k smallest by
    (score bits, slot) returned.  rows_seen / query_seen: the f32 values the index's distance sees.  Returns slots, scores, counters."""
    cb = np.asarray(codebooks, f32); m, c, ds = cb.shape
    q = np.asarray(query_seen, f32)
    lut32 = [[f32(pq_fn(pq_metric, q[j * ds:(j + 1) * ds], cb[j, cc])) for cc in range(c)] for j in range(m)]
    big = max([v for r_ in lut32 for v in r_ if v > 0] or [f32(0)])
    sc = f32(1)
    if np.isfinite(big):
        while f32(big * sc) > f32(32768):   # table scale (coltt_oracle.cpp): a power of two, exact
            sc = f32(sc * f32(0.5))
    with np.errstate(over="ignore"):
        lut = [[f32(np.float16(f32(v * sc))) for v in r_] for r_ in lut32]   # entries rounded to binary16 (RNE)
    cnt = {"n_dist": 0, "n_exp": 0, "n_hops": 0, "n_exact": 0}
    NONE = 0xFFFFFFFF

    js = 16 * (((m + 15) // 16 + 1) // 2)     # the first ceil(P / 2) of the row's P = ceil(m / 16) 16-byte pieces

    def d_of(s, count=True):
        if count:
            cnt["n_dist"] += 1
        lo = f32(0); hi = f32(0)               # two half-row sums, each in j order from +0.0, added once (coltt_oracle.cpp: pq_adc_walk)
        for j in range(min(m, js)):
            lo = f32(lo + lut[j][int(codes[s][j])])
        for j in range(js, m):
            hi = f32(hi + lut[j][int(codes[s][j])])
        return f32(lo + hi)

    def bits(x):
        return int(f32(x).view(np.uint32))

    def row(s, level):
        return adj0[s] if level == 0 else adj_u[int(upper_off[s]) + level - 1]

    if entry < 0:
        return [], [], cnt
    ep, min_d = int(entry), d_of(int(entry))
    for level in range(int(entry_level), 0, -1):
        while True:
            closest = None
            for nb in row(ep, level):
                nb = int(nb)
                if nb == NONE:
                    break
                dd = d_of(nb)
                if dd < min_d:
                    min_d, closest = dd, nb
            cnt["n_hops"] += 1
            if closest is None:
                break
            ep = closest
    res = [[d_of(ep), ep, False]]          # ascending by (d bits, slot)
    visited = {ep}
    while True:
        ci = next((i for i, e in enumerate(res) if not e[2]), None)
        if ci is None:
            break
        res[ci][2] = True
        lower_bound = res[-1][0]; free = ef - len(res); cnt["n_exp"] += 1
        full_at_pop = free == 0                 # bounded visiting (coltt_oracle.cpp: csr_search_pq): the set stays full, its worst member only improves
        adm = []
        for nb in row(res[ci][1], 0):
            nb = int(nb)
            if nb == NONE:
                break
            if full_at_pop:
                dd = d_of(nb, count=False)      # a table sum is cheaper than a visited test: the bound first
                if not dd < lower_bound:
                    continue                    # can never be admitted, now or later: neither marked nor counted
                if nb in visited:
                    continue
                visited.add(nb); cnt["n_dist"] += 1
                adm.append([dd, nb, False])
                continue
            if nb in visited:
                continue
            visited.add(nb)
            dd = d_of(nb)
            if free > 0:
                adm.append([dd, nb, False]); free -= 1
            elif dd < lower_bound:
                adm.append([dd, nb, False])
        res = sorted(res + adm, key=lambda e: (bits(e[0]), e[1]))[:ef]
    r = len(res) if rerank == 0 else max(rerank, k)
    r = min(r, len(res))
    ex = []
    for e in res[:r]:
        cnt["n_exact"] += 1
        v = np.asarray(rows_seen[e[1]], f32)
        ex.append((bits(cosine(q, v) if metric == 0 else euclidean(q, v)), e[1]))
    ex.sort()
    top = ex[:k]
    return [t[1] for t in top], [np.uint32(t[0]).view(f32) for t in top], cnt
