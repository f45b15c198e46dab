package colttgpu

/*
#include <stdlib.h>
#include "coltt_gpu.h"
*/
import "C"

import (
	"fmt"
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

// GroupBackend plugs a group under the micro-batcher (batcher.go), so single-query RPCs over a sharded collection ride in batches.
func GroupBackend(g *Group, sel, mode int, ef uint32) Backend {
	return func(q []float32, nq int, k uint32) ([]uint64, []float32, []uint32, error) {
		return g.Search(q, nq, k, sel, mode, ef)
	}
}
