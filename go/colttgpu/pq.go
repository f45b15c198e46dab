// pq.go — cgo binding of the product-quantised store (coltt_pq_*, include/coltt_gpu.h; coltt_amd/csrc/pq.hip).
//
// What a maintainer binds it to: the quantiser of pkg/hnswpq — the package playground/hnswpq_verification.go:29 imports and the
// reference's tree does not contain — with models.ProductQuantizerParameters (pkg/models/hnsw_common.go:20-33) as its
// configuration.  Method names follow that call shape: Fit (PreTrainProductQuantizer / Fit, :97-98,154), Insert (:115), Search on
// codes only (:190-199).  NOT COMPILED here (no Go toolchain in the build image).
package colttgpu

/*
#include "coltt_gpu.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

const (
	PQCosine    = int(C.COLTT_PQ_COSINE)    // distancepq.cosineDistance = 1 - Dot        (pkg/distancepq/distance.go:40-42)
	PQEuclidean = int(C.COLTT_PQ_EUCLIDEAN) // distancepq.euclideanDistance = squared L2  (:30-32)
	PQDot       = int(C.COLTT_PQ_DOT)       // distancepq.dotProductDistance = -Dot       (:36-38)
)

// ProductQuantizer — one quantised collection shard on one GPU.
type ProductQuantizer struct {
	h                   Handle
	dim                 uint32
	subVectors, centers uint32
}

// NewProductQuantizer: numSubVectors >= 2, 2 <= numCentroids <= 256 (hnsw_common.go:25,28), dim % numSubVectors == 0.
func NewProductQuantizer(dim uint32, metric int, numSubVectors, numCentroids int) (*ProductQuantizer, error) {
	var h Handle
	if err := call(func() C.int {
		return C.coltt_pq_create(C.uint32_t(dim), C.int(metric), C.uint32_t(numSubVectors), C.uint32_t(numCentroids), &h)
	}); err != nil {
		return nil, err
	}
	return &ProductQuantizer{h: h, dim: dim, subVectors: uint32(numSubVectors), centers: uint32(numCentroids)}, nil
}

func (p *ProductQuantizer) Close() { C.coltt_pq_destroy(p.h) }

// SetCodebooks installs codebooks trained elsewhere: [numSubVectors][numCentroids][dim/numSubVectors], row-major.
func (p *ProductQuantizer) SetCodebooks(cb []float32) error {
	if uint64(len(cb)) != uint64(p.centers)*uint64(p.dim) {
		return fmt.Errorf("codebooks: expect %d floats, got %d", uint64(p.centers)*uint64(p.dim), len(cb))
	}
	return call(func() C.int { return C.coltt_pq_set_codebooks(p.h, fptr(cb)) })
}

// Fit trains the quantiser on a sample of n vectors (TriggerThreshold, hnsw_common.go:29-32) with deterministic Lloyd iterations.
func (p *ProductQuantizer) Fit(sample []float32, n int, iterations int) error {
	if err := checkDim(sample, p.dim, n); err != nil {
		return err
	}
	return call(func() C.int { return C.coltt_pq_train(p.h, fptr(sample), C.size_t(n), C.uint32_t(iterations)) })
}

// Encode returns the numSubVectors codes of each of the n vectors without storing them.
func (p *ProductQuantizer) Encode(vecs []float32, n int) ([]byte, error) {
	if err := checkDim(vecs, p.dim, n); err != nil {
		return nil, err
	}
	out := make([]byte, n*int(p.subVectors))
	err := call(func() C.int { return C.coltt_pq_encode(p.h, fptr(vecs), C.size_t(n), bptr(out)) })
	return out, err
}

// Insert encodes and stores n vectors under ids (existing ids are overwritten).
func (p *ProductQuantizer) Insert(ids []uint64, vecs []float32) error {
	if err := checkDim(vecs, p.dim, len(ids)); err != nil {
		return err
	}
	return call(func() C.int { return C.coltt_pq_upsert(p.h, uptr(ids), fptr(vecs), C.size_t(len(ids))) })
}

// InsertCodes stores ready codes ([len(ids)][numSubVectors]); a code >= numCentroids is refused.
func (p *ProductQuantizer) InsertCodes(ids []uint64, codes []byte) error {
	if len(codes) != len(ids)*int(p.subVectors) {
		return fmt.Errorf("codes: expect %d bytes, got %d", len(ids)*int(p.subVectors), len(codes))
	}
	return call(func() C.int { return C.coltt_pq_upsert_codes(p.h, uptr(ids), bptr(codes), C.size_t(len(ids))) })
}

func (p *ProductQuantizer) Remove(ids []uint64) error {
	if len(ids) == 0 {
		return nil
	}
	return call(func() C.int { return C.coltt_pq_remove(p.h, uptr(ids), C.size_t(len(ids))) })
}

func (p *ProductQuantizer) Len() int {
	var n C.uint64_t
	if call(func() C.int { return C.coltt_pq_len(p.h, &n) }) != nil {
		return 0
	}
	return int(n)
}

// Search: the k nearest of every query by the asymmetric distance (the query's per-sub-space table summed over each row's codes),
// ascending by (score, id).  ids / scores are [nq][k]; counts[q] = min(k, Len()).
func (p *ProductQuantizer) Search(queries []float32, nq int, k uint32) (ids []uint64, scores []float32, counts []uint32, err error) {
	if err = checkDim(queries, p.dim, nq); err != nil {
		return
	}
	ids = make([]uint64, nq*int(k))
	scores = make([]float32, nq*int(k))
	counts = make([]uint32, nq)
	err = call(func() C.int {
		return C.coltt_pq_search(p.h, fptr(queries), C.size_t(nq), C.uint32_t(k), uptr(ids), fptr(scores),
			(*C.uint32_t)(unsafe.Pointer(&counts[0])))
	})
	return
}

// PQBackend lets the micro-batcher (batcher.go) coalesce single-query calls into one scan.
func PQBackend(p *ProductQuantizer) Backend {
	return func(q []float32, nq int, k uint32) ([]uint64, []float32, []uint32, error) { return p.Search(q, nq, k) }
}

// ---- product-quantised HNSW (coltt_hnsw_pq_*): the reference's own PQ call shape — hnswpq.NewProductQuantizationHnsw(), m = 32, 256
// centroids, pre-train, Fit, search on the codes (playground/hnswpq_verification.go:69-105).  The walk runs on table distances over the
// quantiser's codes; the survivors are re-scored with the index's exact distance, which is what the answers carry.

// HnswPqAttach snapshots the trained quantiser into the index and encodes every stored row; later Inserts are encoded as they arrive.
func HnswPqAttach(h Handle, p *ProductQuantizer) error {
	return call(func() C.int { return C.coltt_hnsw_pq_attach(h, p.h) })
}

// HnswPqSearch: ef = 0 -> the configured efSearch; rerank = 0 -> every survivor of the walk is re-scored exactly.
func HnswPqSearch(h Handle, dim uint32, queries []float32, nq int, k, ef, rerank uint32) ([]uint64, []float32, []uint32, error) {
	if nq == 0 || k == 0 {
		return nil, nil, make([]uint32, nq), nil
	}
	if err := checkDim(queries, dim, nq); err != nil {
		return nil, nil, nil, err
	}
	ids := make([]uint64, nq*int(k))
	sc := make([]float32, nq*int(k))
	cnt := make([]uint32, nq)
	err := call(func() C.int {
		return C.coltt_hnsw_pq_search(h, fptr(queries), C.size_t(nq), C.uint32_t(k), C.uint32_t(ef), C.uint32_t(rerank), uptr(ids), fptr(sc),
			(*C.uint32_t)(unsafe.Pointer(&cnt[0])), nil, nil)
	})
	return ids, sc, cnt, err
}
