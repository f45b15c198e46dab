package colttgpu

/*
#include "coltt_gpu.h"
*/
import "C"

import (
	"time"
	"unsafe"
)

// Batcher coalesces the reference's one-query-per-RPC calls (core/core.go:633-695, edge/edge.go:610-690) into GPU batches:
// goroutines enqueue a query and block on a channel; the collector flushes when MaxBatch queries are waiting or
// MaxWait elapsed and issues ONE coltt_hnsw_search with nq = len(batch).
type Batcher struct {
	x        *Hnsw
	MaxBatch int
	MaxWait  time.Duration
	in       chan *pending
}

type pending struct {
	q    []float32
	k    uint
	done chan batchResult
}
type batchResult struct {
	res SearchResult
	err error
}

func NewBatcher(x *Hnsw, maxBatch int, maxWait time.Duration) *Batcher {
	b := &Batcher{x: x, MaxBatch: maxBatch, MaxWait: maxWait, in: make(chan *pending, 4*maxBatch)}
	go b.loop()
	return b
}

func (b *Batcher) Search(q []float32, k uint) (SearchResult, error) {
	p := &pending{q: q, k: k, done: make(chan batchResult, 1)}
	b.in <- p
	r := <-p.done
	return r.res, r.err
}

func (b *Batcher) loop() {
	for first := range b.in {
		batch := []*pending{first}
		timer := time.NewTimer(b.MaxWait)
	collect:
		for len(batch) < b.MaxBatch {
			select {
			case p := <-b.in:
				batch = append(batch, p)
			case <-timer.C:
				break collect
			}
		}
		timer.Stop()
		b.flush(batch)
	}
}

func (b *Batcher) flush(batch []*pending) {
	var kmax uint
	for _, p := range batch {
		if p.k > kmax {
			kmax = p.k
		}
	}
	nq, dim := len(batch), int(b.x.dim)
	flat := make([]float32, nq*dim) // the library copies inputs before returning: no Go pointer is retained
	for i, p := range batch {
		copy(flat[i*dim:], p.q)
	}
	ids := make([]uint64, nq*int(kmax))
	sc := make([]float32, nq*int(kmax))
	cnt := make([]uint32, nq)
	rc := C.coltt_hnsw_search(b.x.h, (*C.float)(unsafe.Pointer(&flat[0])), C.size_t(nq), C.uint32_t(kmax), 0,
		(*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&sc[0])), (*C.uint32_t)(unsafe.Pointer(&cnt[0])), nil)
	err := toErr(rc)
	for i, p := range batch {
		if err != nil {
			p.done <- batchResult{nil, err}
			continue
		}
		n := int(cnt[i])
		if n > int(p.k) {
			n = int(p.k) // identical to a single-query call whenever kmax <= cfg.ef (then ef = cfg.ef for everyone); group by k otherwise
		}
		res := make(SearchResult, n)
		b.x.mu.RLock()
		for j := 0; j < n; j++ {
			id := ids[i*int(kmax)+j]
			res[j] = SearchResultItem{Id: id, Score: sc[i*int(kmax)+j], Metadata: b.x.meta[id]}
		}
		b.x.mu.RUnlock()
		p.done <- batchResult{res, nil}
	}
}
