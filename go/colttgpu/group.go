package colttgpu

/*
#include <stdlib.h>
#include "coltt_gpu.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"
)

// Collection groups: ONE collection over the GPUs of a node (BASELINE.json configs[4]: "sharded 8 ways, RCCL all-gather").
// The reference partitions with sharding.ShardVertex(id, shardCount) (pkg/sharding/shard.go:34-41: core/core.go fans Insert /
// Search out over its shard slice, edge keeps a local queue per worker and merges them, edge/none_vectorstore.go:148-178); a
// group does the same routing inside the library, runs every member's search on its own GPU and stream, exchanges the packed
// per-shard top-k with one RCCL all-gather over xGMI and merges on the host in (score, id) order.
// NOT COMPILED here (no Go toolchain in the build container); the same ABI is driven by tests/test_gpu_group.py and bench.py.

const (
	GroupFlat = int(C.COLTT_GROUP_FLAT) // members are edge-style FLAT stores
	GroupHnsw = int(C.COLTT_GROUP_HNSW) // members are core/vectorindex HNSW graphs

	LayoutShard   = int(C.COLTT_LAYOUT_SHARD)   // vertex id lives on shard ShardVertex(id, world)
	LayoutReplica = int(C.COLTT_LAYOUT_REPLICA) // every member holds everything; query batches are split

	ExchangeAuto = int(C.COLTT_EXCHANGE_AUTO)
	ExchangeRccl = int(C.COLTT_EXCHANGE_RCCL)
	ExchangeHost = int(C.COLTT_EXCHANGE_HOST)
	ExchangeShm  = int(C.COLTT_EXCHANGE_SHM) // processes of one box: POSIX shared memory + process-shared counters (coltt_shm_*)
)

// GroupOpts mirrors coltt_group_opts.  WorldSize / RankBase / UniqueID are only needed when the collection spans more than one
// process (one process per GPU): every process passes the same UniqueID (GroupUniqueID on one of them, shipped to the others).
type GroupOpts struct {
	Kind, Layout, Exchange int
	WorldSize, RankBase    int
	UniqueID               []byte
}

type Group struct {
	h   Handle
	dim uint32
}

func GroupUniqueID() ([]byte, error) {
	id := make([]byte, int(C.COLTT_UNIQUE_ID_BYTES))
	err := call(func() C.int { return C.coltt_group_unique_id(bptr(id)) })
	return id, err
}

// NewGroup opens one member per entry of devices (HIP device ordinals).  cfg is used by GroupHnsw members only.
func NewGroup(devices []int, dim uint32, metric, quant int, cfg *HnswCfg, o GroupOpts) (*Group, error) {
	if len(devices) == 0 {
		return nil, fmt.Errorf("colttgpu: a group needs at least one device")
	}
	if o.UniqueID != nil && len(o.UniqueID) != int(C.COLTT_UNIQUE_ID_BYTES) {
		return nil, fmt.Errorf("colttgpu: unique id must be %d bytes", int(C.COLTT_UNIQUE_ID_BYTES))
	}
	devs := make([]C.int, len(devices))
	for i, d := range devices {
		devs[i] = C.int(d)
	}
	// the unique id travels through C memory: opts holds a pointer, and no Go pointer to Go memory may sit inside a C struct
	var uid *C.uint8_t
	if o.UniqueID != nil {
		uid = (*C.uint8_t)(C.CBytes(o.UniqueID))
		defer C.free(unsafe.Pointer(uid))
	}
	opts := C.coltt_group_opts{kind: C.int32_t(o.Kind), layout: C.int32_t(o.Layout), exchange: C.int32_t(o.Exchange),
		world_size: C.int32_t(o.WorldSize), rank_base: C.int32_t(o.RankBase), unique_id: uid}
	var cc C.coltt_hnsw_cfg
	var pc *C.coltt_hnsw_cfg
	if cfg != nil {
		cc = cfg.c()
		pc = &cc
	}
	g := &Group{dim: dim}
	err := call(func() C.int {
		return C.coltt_group_create(&devs[0], C.int(len(devs)), C.uint32_t(dim), C.int(metric), C.int(quant), pc, &opts, &g.h)
	})
	if err != nil {
		return nil, err
	}
	return g, nil
}

func (g *Group) Close() { C.coltt_group_destroy(g.h) }

// Info: members in this process, shards in the whole collection, the exchange in use, the shard number of the first member.
func (g *Group) Info() (nLocal, world, exchange, rankBase int, err error) {
	var a, b, c, d C.int32_t
	err = call(func() C.int { return C.coltt_group_info(g.h, &a, &b, &c, &d) })
	return int(a), int(b), int(c), int(d), err
}

// Member hands out the i-th local member's own handle (a coltt_flat_* / coltt_hnsw_* handle) — Commit / SaveVertex and
// device-resident ingest go straight to it.
func (g *Group) Member(i int) (Handle, error) {
	var h Handle
	err := call(func() C.int { return C.coltt_group_member(g.h, C.int(i), &h) })
	return h, err
}

// ShardOf = sharding.ShardVertex(id, world).
func (g *Group) ShardOf(id uint64) (int, error) {
	var s C.int32_t
	err := call(func() C.int { return C.coltt_group_shard_of(g.h, C.uint64_t(id), &s) })
	return int(s), err
}

// Upsert = ChangedVertex on whichever member hosts each id (FLAT groups).  kept = how many of the n vertices this process hosts.
func (g *Group) Upsert(ids []uint64, vecs []float32) (kept uint64, err error) {
	if err = checkDim(vecs, g.dim, len(ids)); err != nil {
		return 0, err
	}
	var k C.uint64_t
	err = call(func() C.int { return C.coltt_group_upsert(g.h, uptr(ids), fptr(vecs), C.size_t(len(ids)), &k) })
	return uint64(k), err
}

// Insert = Hnsw.Insert on whichever member hosts each id (HNSW groups); levels come from the Go side's RandomLevel draws.
func (g *Group) Insert(ids []uint64, vecs []float32, levels []int32, batch uint32) (kept uint64, err error) {
	if err = checkDim(vecs, g.dim, len(ids)); err != nil {
		return 0, err
	}
	if len(levels) != len(ids) {
		return 0, fmt.Errorf("colttgpu: %d levels for %d ids", len(levels), len(ids))
	}
	var k C.uint64_t
	err = call(func() C.int {
		return C.coltt_group_insert(g.h, uptr(ids), fptr(vecs), (*C.int32_t)(unsafe.Pointer(&levels[0])), C.size_t(len(ids)), C.uint32_t(batch), &k)
	})
	return uint64(k), err
}

func (g *Group) Remove(ids []uint64) error {
	if len(ids) == 0 {
		return nil
	}
	return call(func() C.int { return C.coltt_group_remove(g.h, uptr(ids), C.size_t(len(ids))) })
}

func (g *Group) Len() (uint64, error) {
	var n C.uint64_t
	err := call(func() C.int { return C.coltt_group_len(g.h, &n) })
	return uint64(n), err
}

// Search runs the batch over the whole collection.  sel / mode apply to FLAT groups, ef to HNSW groups (0 = the configured
// efSearch).  In a multi-process group every process must make the same call.  Rows come back ascending by (score, id).
func (g *Group) Search(queries []float32, nq int, k uint32, sel, mode int, ef uint32) (ids []uint64, scores []float32, counts []uint32, err error) {
	if k == 0 || nq == 0 {
		return nil, nil, make([]uint32, nq), nil
	}
	if err = checkDim(queries, g.dim, nq); err != nil {
		return
	}
	ids = make([]uint64, nq*int(k))
	scores = make([]float32, nq*int(k))
	counts = make([]uint32, nq)
	err = call(func() C.int {
		return C.coltt_group_search(g.h, fptr(queries), C.size_t(nq), C.uint32_t(k), C.int(sel), C.int(mode), C.uint32_t(ef),
			uptr(ids), fptr(scores), (*C.uint32_t)(unsafe.Pointer(&counts[0])))
	})
	return
}

// PendingSearch is a batch between SearchBegin and End.  The library writes the merged answers AFTER SearchBegin has returned, so
// the out arrays are C memory (cgo: no Go pointer may be retained by C past the call); End copies them into Go slices and frees them.
type PendingSearch struct {
	g      *Group
	ticket C.uint64_t
	nq     int
	k      uint32
	ids    *C.uint64_t
	scores *C.float
	counts *C.uint32_t
}

// SearchBegin — coltt_group_search_begin: every local member searches the batch now; the exchange (pack + one all-gather + D2H on
// the comm streams) and the host merge are queued behind the earlier batches'.  A caller that begins batch i+1 before it Ends batch i
// hides exchange and merge under the next search.  At most 3 batches may be pending.  Multi-process groups: same calls, same order.
func (g *Group) SearchBegin(queries []float32, nq int, k uint32, sel, mode int, ef uint32) (*PendingSearch, error) {
	if k == 0 || nq == 0 {
		return nil, fmt.Errorf("colttgpu: empty batch")
	}
	if err := checkDim(queries, g.dim, nq); err != nil {
		return nil, err
	}
	p := &PendingSearch{g: g, nq: nq, k: k,
		ids:    (*C.uint64_t)(C.malloc(C.size_t(nq * int(k) * 8))),
		scores: (*C.float)(C.malloc(C.size_t(nq * int(k) * 4))),
		counts: (*C.uint32_t)(C.malloc(C.size_t(nq * 4)))}
	err := call(func() C.int {
		return C.coltt_group_search_begin(g.h, fptr(queries), nil, C.size_t(nq), C.uint32_t(k), C.int(sel), C.int(mode), C.uint32_t(ef),
			p.ids, p.scores, p.counts, &p.ticket)
	})
	if err != nil {
		p.free()
		return nil, err
	}
	// A batch that is never Ended must not leak its three C blocks — nor free them while the group's exchange thread may still write into them
	// (ADVICE r5): the finalizer waits for the batch (Close) before it frees.
	runtime.SetFinalizer(p, func(q *PendingSearch) { q.Close() })
	return p, nil
}

func (p *PendingSearch) free() {
	if p.ids == nil {
		return
	}
	C.free(unsafe.Pointer(p.ids))
	C.free(unsafe.Pointer(p.scores))
	C.free(unsafe.Pointer(p.counts))
	p.ids, p.scores, p.counts = nil, nil, nil
	runtime.SetFinalizer(p, nil)
}

// Close abandons a batch: it waits until the library has finished writing the batch's answers (the ticket is consumed, its error ignored) and
// frees the C blocks.  End calls it implicitly; after End or Close it does nothing.
func (p *PendingSearch) Close() {
	if p.ids == nil {
		return
	}
	_ = call(func() C.int { return C.coltt_group_search_end(p.g.h, p.ticket) })
	p.free()
}

// End blocks until the batch's merged answers are complete and returns them (rows ascending by (score, id)).
func (p *PendingSearch) End() (ids []uint64, scores []float32, counts []uint32, err error) {
	defer p.free()
	if err = call(func() C.int { return C.coltt_group_search_end(p.g.h, p.ticket) }); err != nil {
		return
	}
	n := p.nq * int(p.k)
	ids = make([]uint64, n)
	scores = make([]float32, n)
	counts = make([]uint32, p.nq)
	copy(ids, unsafe.Slice((*uint64)(unsafe.Pointer(p.ids)), n))
	copy(scores, unsafe.Slice((*float32)(unsafe.Pointer(p.scores)), n))
	copy(counts, unsafe.Slice((*uint32)(unsafe.Pointer(p.counts)), p.nq))
	return
}

// Timing — cumulative ms of the finished shard-search batches: members' searches, exchange (pack + all-gather + D2H), host merge.
func (g *Group) Timing() (batches uint64, searchMs, exchangeMs, mergeMs float64, err error) {
	var o [4]C.double
	err = call(func() C.int { return C.coltt_group_timing(g.h, &o[0]) })
	return uint64(o[0]), float64(o[1]), float64(o[2]), float64(o[3]), err
}

// GroupBackend plugs a group under the micro-batcher (batcher.go), so single-query RPCs over a sharded collection ride in batches.
func GroupBackend(g *Group, sel, mode int, ef uint32) Backend {
	return func(q []float32, nq int, k uint32) ([]uint64, []float32, []uint32, error) {
		return g.Search(q, nq, k, sel, mode, ef)
	}
}
