package colttgpu

import (
	"fmt"
	"sync"
	"time"
)

// Batcher coalesces the reference's one-query-per-RPC calls (core/core.go:633-695, edge/edge.go:610-690) into GPU batches:
// goroutines enqueue a query and block on a channel; the collector flushes when MaxBatch queries are waiting or MaxWait
// elapsed since the first of them and issues ONE batched search per distinct k.  Same semantics as the compiled C++ twin
// include/coltt_batcher.hpp (which is the one exercised on the GPU here — no Go toolchain in the build container):
//   - queries are GROUPED BY k: HNSW searches with ef = max(cfg.ef, k), so answers for different k are not prefixes of one
//     another; a k = 10 query batched with a k = 500 query must get exactly what a single-query call returns;
//   - k == 0 answers immediately with an empty result (hnsw.go:243-278 returns no rows), it never reaches the backend;
//   - len(query) != dim is an error for THAT caller only (C never reads past a short Go slice).
//
// Backend is any batched search: HnswBackend / FlatBackend below, or a test double.
type Backend func(queries []float32, nq int, k uint32) (ids []uint64, scores []float32, counts []uint32, err error)

type BatchItem struct {
	Id    uint64
	Score float32
}

type Batcher struct {
	dim      int
	backend  Backend
	MaxBatch int
	MaxWait  time.Duration
	in       chan *pending
	quit     chan struct{}
	closeOne sync.Once
}

type pending struct {
	q    []float32
	k    uint32
	done chan batchResult
}
type batchResult struct {
	items []BatchItem
	err   error
}

func NewBatcher(dim int, backend Backend, maxBatch int, maxWait time.Duration) *Batcher {
	if maxBatch < 1 {
		maxBatch = 1
	}
	b := &Batcher{dim: dim, backend: backend, MaxBatch: maxBatch, MaxWait: maxWait, in: make(chan *pending, 4*maxBatch),
		quit: make(chan struct{})}
	go b.loop()
	return b
}

// Close stops the collector.  Queries still queued (or racing with Close) are answered with "batcher closed": no caller is left
// blocked on its done channel.  Safe to call more than once.
func (b *Batcher) Close() { b.closeOne.Do(func() { close(b.quit) }) }

var errClosed = fmt.Errorf("batcher closed")

// HnswBackend / FlatBackend: the two searches that need batching (BASELINE.json configs 2-5)
func HnswBackend(h Handle, dim uint32, ef uint32) Backend {
	return func(q []float32, nq int, k uint32) ([]uint64, []float32, []uint32, error) {
		return HnswSearch(h, dim, q, nq, k, ef)
	}
}
func FlatBackend(h Handle, dim uint32, sel, mode int) Backend {
	return func(q []float32, nq int, k uint32) ([]uint64, []float32, []uint32, error) {
		return FlatSearch(h, dim, q, nq, k, sel, mode, nil, false)
	}
}

// Search blocks until the batch this query rode in has been answered.  The query is copied before it is queued.
func (b *Batcher) Search(q []float32, k uint) ([]BatchItem, error) {
	if len(q) != b.dim {
		return nil, fmt.Errorf("Dim Length UnmatchdError: expect dimension: [%d], but got [%d]", b.dim, len(q))
	}
	if k == 0 {
		return []BatchItem{}, nil
	}
	p := &pending{q: append([]float32(nil), q...), k: uint32(k), done: make(chan batchResult, 1)}
	select {
	case b.in <- p:
	case <-b.quit:
		return nil, errClosed
	}
	select {
	case r := <-p.done:
		return r.items, r.err
	case <-b.quit:
		// the collector may still answer (it drains on quit) — prefer its answer, never block
		select {
		case r := <-p.done:
			return r.items, r.err
		default:
			return nil, errClosed
		}
	}
}

func (b *Batcher) loop() {
	for {
		var first *pending
		select {
		case first = <-b.in:
		case <-b.quit:
			b.drain(nil)
			return
		}
		batch := []*pending{first}
		timer := time.NewTimer(b.MaxWait)
		closed := false
	collect:
		for len(batch) < b.MaxBatch {
			select {
			case p := <-b.in:
				batch = append(batch, p)
			case <-timer.C:
				break collect
			case <-b.quit:
				closed = true
				break collect
			}
		}
		timer.Stop()
		if closed {
			b.drain(batch)
			return
		}
		b.flush(batch)
	}
}

// drain answers the batch in hand and everything still queued with errClosed (done has capacity 1: never blocks)
func (b *Batcher) drain(batch []*pending) {
	for _, p := range batch {
		p.done <- batchResult{nil, errClosed}
	}
	for {
		select {
		case p := <-b.in:
			p.done <- batchResult{nil, errClosed}
		default:
			return
		}
	}
}

func (b *Batcher) flush(batch []*pending) {
	byK := map[uint32][]*pending{}
	var order []uint32
	for _, p := range batch {
		if _, ok := byK[p.k]; !ok {
			order = append(order, p.k)
		}
		byK[p.k] = append(byK[p.k], p)
	}
	for _, k := range order {
		grp := byK[k]
		nq := len(grp)
		flat := make([]float32, nq*b.dim) // the library copies inputs before returning: no Go pointer is retained
		for i, p := range grp {
			copy(flat[i*b.dim:], p.q)
		}
		ids, sc, cnt, err := b.backend(flat, nq, k)
		for i, p := range grp {
			if err != nil {
				p.done <- batchResult{nil, err}
				continue
			}
			n := int(cnt[i])
			items := make([]BatchItem, n)
			for j := 0; j < n; j++ {
				items[j] = BatchItem{Id: ids[i*int(k)+j], Score: sc[i*int(k)+j]}
			}
			p.done <- batchResult{items, nil}
		}
	}
}
