// Package colttgpu — cgo shim that puts libcoltt_gpu.so behind the reference's Go interfaces.
//
// NOT COMPILED in the build container (no Go toolchain there); shipped as source for the maintainer.  It mirrors
//   edge.vectorspace      (edge/vectorstore.go:30-49)        -> GpuVecSpace
//   *vectorindex.Hnsw     (core/vectorindex/hnsw.go:43-54)   -> Hnsw
// Metadata never crosses the boundary: the shim keeps id -> Metadata and re-attaches it after each call.
package colttgpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../coltt_amd -lcoltt_gpu -Wl,-rpath,${SRCDIR}/../../coltt_amd
#include "coltt_gpu.h"
*/
import "C"

import (
	"context"
	"errors"
	"fmt"
	"math/rand"
	"sync"
	"unsafe"
)

var (
	ItemNotFoundError      = errors.New("Item not found")      // core/vectorindex/hnsw.go:39
	ItemAlreadyExistsError = errors.New("Item already exists") // core/vectorindex/hnsw.go:40
)

func toErr(rc C.int) error {
	switch rc {
	case C.COLTT_OK:
		return nil
	case C.COLTT_E_NOT_FOUND:
		return ItemNotFoundError
	case C.COLTT_E_EXISTS:
		return ItemAlreadyExistsError
	}
	return errors.New(C.GoString(C.coltt_last_error()))
}

type Metadata map[string]any

type SearchResultItem struct {
	Id       uint64
	Metadata map[string]any
	Score    float32
}
type SearchResult []SearchResultItem

// ---------------------------------------------------------------------------------------------- HNSW
type Hnsw struct {
	h    C.coltt_handle_t
	dim  uint
	mu   sync.RWMutex
	meta map[uint64]Metadata
}

// NewHnsw(dim, distancer, options...) — core/vectorindex/hnsw.go:56.  metric: 0 cosine-dot, 1 l2.
func NewHnsw(dim uint, metric int, cfg *C.coltt_hnsw_cfg) (*Hnsw, error) {
	x := &Hnsw{dim: dim, meta: map[uint64]Metadata{}}
	if err := toErr(C.coltt_hnsw_create(C.uint32_t(dim), C.int(metric), C.COLTT_Q_NONE, cfg, &x.h)); err != nil {
		return nil, err
	}
	return x, nil
}

// Insert(id, value, metadata, vertexLevel) — hnsw.go:104
func (x *Hnsw) Insert(id uint64, value []float32, metadata Metadata, vertexLevel int) error {
	if uint(len(value)) != x.dim {
		return fmt.Errorf("Dim Length UnmatchdError: expect dimension: [%d], but got [%d]", x.dim, len(value))
	}
	if err := toErr(C.coltt_hnsw_insert(x.h, C.uint64_t(id), (*C.float)(unsafe.Pointer(&value[0])), C.int32_t(vertexLevel))); err != nil {
		return err
	}
	x.mu.Lock()
	x.meta[id] = metadata
	x.mu.Unlock()
	return nil
}

// Remove(id) — hnsw.go:191
func (x *Hnsw) Remove(id uint64) error {
	if err := toErr(C.coltt_hnsw_remove(x.h, C.uint64_t(id))); err != nil {
		return err
	}
	x.mu.Lock()
	delete(x.meta, id)
	x.mu.Unlock()
	return nil
}

// Search(ctx, query, k) — hnsw.go:243.  One query per call as in the reference; see batcher.go for coalescing.
func (x *Hnsw) Search(_ context.Context, query []float32, k uint) (SearchResult, error) {
	if k == 0 {
		return SearchResult{}, nil
	}
	ids := make([]uint64, k)
	sc := make([]float32, k)
	var cnt C.uint32_t
	rc := C.coltt_hnsw_search(x.h, (*C.float)(unsafe.Pointer(&query[0])), 1, C.uint32_t(k), 0,
		(*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&sc[0])), &cnt, nil)
	if err := toErr(rc); err != nil {
		return nil, err
	}
	res := make(SearchResult, int(cnt))
	x.mu.RLock()
	for i := range res {
		res[i] = SearchResultItem{Id: ids[i], Score: sc[i], Metadata: x.meta[ids[i]]}
	}
	x.mu.RUnlock()
	return res, nil
}

func (x *Hnsw) Len() int { var n C.uint64_t; C.coltt_hnsw_len(x.h, &n); return int(n) }

// RandomLevel mirrors (*vectorindex.Hnsw).RandomLevel (hnsw.go:280-282): the uniform draw stays on the Go side
// (math/rand, as in the reference), the library applies gomath.Floor(-gomath.Log(u) * levelMultiplier).
func (x *Hnsw) RandomLevel() int {
	u := rand.Float32()
	for u <= 0 { // the reference would produce Floor(+Inf); redraw instead
		u = rand.Float32()
	}
	var lv C.int32_t
	C.coltt_hnsw_random_level(x.h, C.float(u), &lv)
	return int(lv)
}
func (x *Hnsw) Dim() uint32 { return uint32(x.dim) }
func (x *Hnsw) Close()    { C.coltt_hnsw_destroy(x.h) }

// ---------------------------------------------------------------------------------------------- edge FLAT
type ENode struct {
	Vector   []float32
	Metadata map[string]interface{}
}

// GpuVecSpace satisfies the vector half of edge.vectorspace (edge/vectorstore.go:30-49); the inverted index, metadata
// analyzers and persistence stay in package edge and call into this type.
type GpuVecSpace struct {
	h        C.coltt_handle_t
	dim      uint32
	distance int
	quant    int
	mu       sync.RWMutex
	meta     map[uint64]map[string]interface{}
}

func NewGpuVecSpace(dim uint32, distance, quantization int) (*GpuVecSpace, error) {
	s := &GpuVecSpace{dim: dim, distance: distance, quant: quantization, meta: map[uint64]map[string]interface{}{}}
	if err := toErr(C.coltt_flat_create(C.uint32_t(dim), C.int(distance), C.int(quantization), &s.h)); err != nil {
		return nil, err // "not support quantization type" for unknown enums (edge/vectorstore.go:79)
	}
	return s, nil
}

// ChangedVertex — edge/none_vectorstore.go:66-103 (primary-key lookup / analyzers / inverted.Add happen in the caller)
func (s *GpuVecSpace) ChangedVertex(commitId uint64, data ENode) error {
	if s.dim != uint32(len(data.Vector)) {
		return fmt.Errorf("Dim Length UnmatchdError: expect dimension: [%d], but got [%d]", s.dim, len(data.Vector))
	}
	id := C.uint64_t(commitId)
	if err := toErr(C.coltt_flat_upsert(s.h, &id, (*C.float)(unsafe.Pointer(&data.Vector[0])), 1)); err != nil {
		return err
	}
	s.mu.Lock()
	s.meta[commitId] = data.Metadata
	s.mu.Unlock()
	return nil
}

// RemoveVertex — edge/none_vectorstore.go:118-124, after SearchMultiFilter resolved dropFilter to ids
func (s *GpuVecSpace) RemoveVertex(dropIds []uint64) error {
	if len(dropIds) == 0 {
		return nil
	}
	if err := toErr(C.coltt_flat_remove(s.h, (*C.uint64_t)(unsafe.Pointer(&dropIds[0])), C.size_t(len(dropIds)))); err != nil {
		return err
	}
	s.mu.Lock()
	for _, id := range dropIds {
		delete(s.meta, id)
	}
	s.mu.Unlock()
	return nil
}

// VertexSearch — edge/none_vectorstore.go:129-180.  COLTT_SELECT_REFERENCE reproduces the reference's queue exactly
// (it keeps the K LARGEST distances, edge/priority_queue.go:46-55).
func (s *GpuVecSpace) VertexSearch(target []float32, topK int, _ bool) ([]*SearchResultItem, error) {
	return s.search(target, topK, nil)
}

// FilterableVertexSearch — edge/none_vectorstore.go:182-253, candidates = invertedIndex.SearchWithExpression(filter)
func (s *GpuVecSpace) FilterableVertexSearch(candidates []uint64, target []float32, topK int, _ bool) ([]*SearchResultItem, error) {
	if candidates == nil {
		candidates = []uint64{}
	}
	return s.search(target, topK, candidates)
}

func (s *GpuVecSpace) search(target []float32, topK int, cand []uint64) ([]*SearchResultItem, error) {
	if topK <= 0 {
		return []*SearchResultItem{}, nil
	}
	ids := make([]uint64, topK)
	sc := make([]float32, topK)
	var cnt C.uint32_t
	var rc C.int
	if cand == nil {
		rc = C.coltt_flat_search(s.h, (*C.float)(unsafe.Pointer(&target[0])), 1, C.uint32_t(topK), C.COLTT_SELECT_REFERENCE,
			C.COLTT_MODE_EXACT, (*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&sc[0])), &cnt)
	} else {
		var cp *C.uint64_t
		if len(cand) > 0 {
			cp = (*C.uint64_t)(unsafe.Pointer(&cand[0]))
		}
		rc = C.coltt_flat_search_ids(s.h, (*C.float)(unsafe.Pointer(&target[0])), 1, C.uint32_t(topK), C.COLTT_SELECT_REFERENCE,
			cp, C.size_t(len(cand)), (*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&sc[0])), &cnt)
	}
	if err := toErr(rc); err != nil {
		return nil, err
	}
	out := make([]*SearchResultItem, int(cnt))
	s.mu.RLock()
	for i := range out {
		out[i] = &SearchResultItem{Id: ids[i], Score: sc[i], Metadata: s.meta[ids[i]]}
	}
	s.mu.RUnlock()
	return out, nil
}

func (s *GpuVecSpace) Dim() uint32     { return s.dim }
func (s *GpuVecSpace) LoadSize() int64 { var n C.uint64_t; C.coltt_flat_len(s.h, &n); return int64(n) }
func (s *GpuVecSpace) Close()          { C.coltt_flat_destroy(s.h) }
