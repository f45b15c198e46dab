// Package colttgpu — the cgo layer under the two drop-in packages of this directory tree:
//
//	go/vectorindex  replacement body for github.com/sjy-dv/coltt/core/vectorindex   (*Hnsw and friends)
//	go/edge         one extra file for package github.com/sjy-dv/coltt/edge          (gpuVecSpace, an edge.vectorspace)
//
// NOT COMPILED in the build container (no Go toolchain there); shipped as source for the maintainer (INTEGRATION.md).
// Rules kept here once so the callers cannot get them wrong:
//   - no Go pointer is retained by C: every call copies in / out before it returns;
//   - coltt_last_error() is thread-local and a goroutine may migrate between OS threads across cgo calls, so every
//     call + error fetch runs under runtime.LockOSThread (Call);
//   - slices handed to C are validated against `dim` first — C never reads past a short Go slice.
package colttgpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../coltt_amd -lcoltt_gpu -Wl,-rpath,${SRCDIR}/../../coltt_amd
#include <stdlib.h>
#include "coltt_gpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"
)

var (
	ErrNotFound = errors.New("Item not found")      // core/vectorindex/hnsw.go:39 ItemNotFoundError
	ErrExists   = errors.New("Item already exists") // core/vectorindex/hnsw.go:40 ItemAlreadyExistsError
)

type Handle = C.coltt_handle_t

// HnswCfg mirrors coltt_hnsw_cfg (hnswConfig, core/vectorindex/hnsw_config.go:135-162).
type HnswCfg struct {
	M, MMax, MMax0, Ef, EfConstruction, Algo int32
	LevelMultiplier                          float32
	ExtendCandidates, KeepPruned             int32
}

func (c *HnswCfg) c() C.coltt_hnsw_cfg {
	return C.coltt_hnsw_cfg{m: C.int32_t(c.M), m_max: C.int32_t(c.MMax), m_max0: C.int32_t(c.MMax0), ef: C.int32_t(c.Ef),
		ef_construction: C.int32_t(c.EfConstruction), algo: C.int32_t(c.Algo), level_multiplier: C.float(c.LevelMultiplier),
		extend_candidates: C.int32_t(c.ExtendCandidates), keep_pruned: C.int32_t(c.KeepPruned)}
}

// call runs one C entry point and turns its status into a Go error on the SAME OS thread that made the call.
func call(f func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	rc := f()
	switch rc {
	case C.COLTT_OK:
		return nil
	case C.COLTT_E_NOT_FOUND:
		return ErrNotFound
	case C.COLTT_E_EXISTS:
		return ErrExists
	}
	return errors.New(C.GoString(C.coltt_last_error()))
}

func checkDim(v []float32, dim uint32, n int) error {
	if n == 0 || uint64(len(v)) != uint64(dim)*uint64(n) {
		// edge/none_vectorstore.go:86-88
		return fmt.Errorf("Dim Length UnmatchdError: expect dimension: [%d], but got [%d]", dim, len(v)/max1(n))
	}
	return nil
}
func max1(n int) int {
	if n < 1 {
		return 1
	}
	return n
}
func fptr(v []float32) *C.float {
	if len(v) == 0 {
		return nil
	}
	return (*C.float)(unsafe.Pointer(&v[0]))
}
func uptr(v []uint64) *C.uint64_t {
	if len(v) == 0 {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(&v[0]))
}
func bptr(v []byte) *C.uint8_t {
	if len(v) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&v[0]))
}

func Init(device int) error { return call(func() C.int { return C.coltt_init(C.int(device)) }) }

// Normalize — edge.Normalize / vectorindex.Normalize (edge/vectorstore.go:173-189; core/vectorindex/metadata.go:107-123) through
// coltt_normalize_host: the library's HOST copy of the arithmetic (sequential f32 sum, float64 sqrt, per-element divide).  It touches no
// device, stream or allocator, so a call per RPC cannot stall concurrent searches, needs no initialised GPU and — like the reference's
// function — cannot fail; a zero vector comes back as zeros.  Batches (tests, bulk ingest) use NormalizeBatch.
func Normalize(v []float32) []float32 {
	out := make([]float32, len(v))
	if len(v) != 0 {
		C.coltt_normalize_host(fptr(v), C.uint32_t(len(v)), fptr(out))
	}
	return out
}

// NormalizeBatch — n row-major vectors on the device (coltt_normalize); for bulk paths only.
func NormalizeBatch(v []float32, dim uint32) ([]float32, error) {
	out := make([]float32, len(v))
	if len(v) == 0 || dim == 0 {
		return out, nil
	}
	err := call(func() C.int { return C.coltt_normalize(fptr(v), C.size_t(len(v)/int(dim)), C.uint32_t(dim), fptr(out)) })
	return out, err
}

// ------------------------------------------------------------------------------------------------ HNSW
func HnswCreate(dim uint32, metric, quant int, cfg *HnswCfg) (Handle, error) {
	var h Handle
	cc := cfg.c()
	err := call(func() C.int { return C.coltt_hnsw_create(C.uint32_t(dim), C.int(metric), C.int(quant), &cc, &h) })
	return h, err
}
func HnswDestroy(h Handle) { C.coltt_hnsw_destroy(h) }
func HnswGetCfg(h Handle) (HnswCfg, error) {
	var c C.coltt_hnsw_cfg
	err := call(func() C.int { return C.coltt_hnsw_get_cfg(h, &c) })
	return HnswCfg{int32(c.m), int32(c.m_max), int32(c.m_max0), int32(c.ef), int32(c.ef_construction), int32(c.algo),
		float32(c.level_multiplier), int32(c.extend_candidates), int32(c.keep_pruned)}, err
}
func HnswInsert(h Handle, dim uint32, id uint64, v []float32, level int) error {
	if err := checkDim(v, dim, 1); err != nil {
		return err
	}
	return call(func() C.int { return C.coltt_hnsw_insert(h, C.uint64_t(id), fptr(v), C.int32_t(level)) })
}
// HnswReserve sizes every device array of the index for nSlots vertices in one allocation (coltt_hnsw_reserve): optional, but a
// collection whose size is known (a Load, a bulk import) should call it before the inserts — growth re-copies the arrays.
func HnswReserve(h Handle, nSlots uint64) error {
	return call(func() C.int { return C.coltt_hnsw_reserve(h, C.uint64_t(nSlots), 0) })
}

func HnswRemove(h Handle, id uint64) error {
	return call(func() C.int { return C.coltt_hnsw_remove(h, C.uint64_t(id)) })
}
func HnswLen(h Handle) int {
	var n C.uint64_t
	C.coltt_hnsw_len(h, &n)
	return int(n)
}

// HnswSearch: nq queries (row-major), k results each; returns ids, scores [nq*k] and counts [nq].
func HnswSearch(h Handle, dim uint32, queries []float32, nq int, k uint32, ef uint32) ([]uint64, []float32, []uint32, error) {
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
		return C.coltt_hnsw_search(h, fptr(queries), C.size_t(nq), C.uint32_t(k), C.uint32_t(ef), uptr(ids), fptr(sc),
			(*C.uint32_t)(unsafe.Pointer(&cnt[0])), nil)
	})
	return ids, sc, cnt, err
}
func HnswRandomLevel(h Handle, u float32) (int, error) {
	var lv C.int32_t
	err := call(func() C.int { return C.coltt_hnsw_random_level(h, C.float(u), &lv) })
	return int(lv), err
}

// HnswGet: stored (normalised) f32 vector and level of a live vertex (hnsw.go:169-189).
func HnswGet(h Handle, dim uint32, id uint64) ([]float32, int, error) {
	v := make([]float32, dim)
	var lv C.int32_t
	err := call(func() C.int { return C.coltt_hnsw_get(h, C.uint64_t(id), unsafe.Pointer(&v[0]), &lv) })
	return v, int(lv), err
}

// HnswEntryLevel: level of the entrypoint (-1 = empty index); host-side only, no device traffic (what BytesSize needs).
func HnswEntryLevel(h Handle) (int, error) {
	var lv C.int32_t
	err := call(func() C.int { return C.coltt_hnsw_entry_level(h, &lv) })
	return int(lv), err
}

// HnswSlots: ids of every slot in slot order and the deleted flags (what Commit walks).
// The size query and the fill are two cgo calls; an Insert may land between them.  The library treats the counts passed to
// the second call as the CAPACITIES of the slices and refuses (COLTT_E_INVALID, needed sizes returned) instead of writing past
// them, so the loop simply re-sizes and retries.
func HnswSlots(h Handle) (ids []uint64, deleted []byte, err error) {
	for attempt := 0; attempt < 8; attempt++ {
		var ns, nr, ne C.uint64_t
		var ent C.int32_t
		if err = call(func() C.int { return C.coltt_hnsw_export(h, &ns, &nr, &ne, nil, nil, nil, nil, nil, nil, &ent) }); err != nil {
			return
		}
		if ns == 0 {
			return nil, nil, nil
		}
		capSlots := ns
		ids = make([]uint64, int(capSlots))
		deleted = make([]byte, int(capSlots))
		rc := C.int(0)
		err = call(func() C.int {
			rc = C.coltt_hnsw_export(h, &ns, &nr, &ne, uptr(ids), nil, bptr(deleted), nil, nil, nil, &ent)
			return rc
		})
		if err == nil {
			return ids[:int(ns)], deleted[:int(ns)], nil // ns <= capSlots: Remove never shrinks the slot count
		}
		if rc != C.COLTT_E_INVALID || ns <= capSlots {
			return nil, nil, err
		}
	}
	return nil, nil, errors.New("hnsw export: the index kept growing during 8 attempts")
}

// HnswCommit: Hnsw.Commit stream (hnsw_commit.go:69-162); metaBlobs[slot] = that vertex's Metadata in stream encoding
// (metadata.go:31-74), nil = empty map.
func HnswCommit(h Handle, header bool, metaBlobs [][]byte) ([]byte, error) {
	n := len(metaBlobs)
	// C arrays of blob pointers live in C memory for the duration of the call (no Go pointer to Go pointer crosses cgo)
	var cptr **C.uint8_t
	var clen *C.uint32_t
	var pins []unsafe.Pointer
	if n > 0 {
		cptr = (**C.uint8_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))
		clen = (*C.uint32_t)(C.malloc(C.size_t(n) * 4))
		defer C.free(unsafe.Pointer(cptr))
		defer C.free(unsafe.Pointer(clen))
		ps := unsafe.Slice(cptr, n)
		ls := unsafe.Slice(clen, n)
		for i, b := range metaBlobs {
			if len(b) == 0 {
				ps[i], ls[i] = nil, 0
				continue
			}
			p := C.CBytes(b)
			pins = append(pins, p)
			ps[i], ls[i] = (*C.uint8_t)(p), C.uint32_t(len(b))
		}
		defer func() {
			for _, p := range pins {
				C.free(p)
			}
		}()
	}
	hd := C.int(0)
	if header {
		hd = 1
	}
	var need C.uint64_t
	if err := call(func() C.int { return C.coltt_hnsw_commit(h, hd, cptr, clen, C.uint64_t(n), nil, 0, &need) }); err != nil {
		return nil, err
	}
	// slots past n (vertices inserted since the caller built metaBlobs) are written with empty metadata by the library; a
	// stream that outgrew the buffer between the two calls comes back as "buffer too small" and is retried with the new size
	for attempt := 0; attempt < 8; attempt++ {
		capB := need
		out := make([]byte, int(capB))
		err := call(func() C.int { return C.coltt_hnsw_commit(h, hd, cptr, clen, C.uint64_t(n), bptr(out), capB, &need) })
		if err == nil {
			return out[:int(need)], nil
		}
		if need <= capB {
			return nil, err
		}
	}
	return nil, errors.New("hnsw commit: the index kept growing during 8 attempts")
}

// HnswLoad: Hnsw.Load (hnsw_commit.go:164-278) straight into HBM; returns each vertex's id and the position of its
// metadata blob inside data (the caller decodes the msgpack values).
func HnswLoad(h Handle, header bool, data []byte, dim uint32) (ids []uint64, metaOff []uint64, metaLen []uint32, err error) {
	capN := uint64(len(data))/uint64(14+4*dim) + 1 // a vertex record is at least 8 + 4 + 4*dim + 2 bytes
	ids = make([]uint64, capN)
	metaOff = make([]uint64, capN)
	metaLen = make([]uint32, capN)
	hd := C.int(0)
	if header {
		hd = 1
	}
	var n C.uint64_t
	err = call(func() C.int {
		return C.coltt_hnsw_load(h, hd, bptr(data), C.uint64_t(len(data)), &n, uptr(ids), uptr(metaOff),
			(*C.uint32_t)(unsafe.Pointer(&metaLen[0])), C.uint64_t(capN))
	})
	if err != nil {
		return nil, nil, nil, err
	}
	return ids[:n], metaOff[:n], metaLen[:n], nil
}

// ------------------------------------------------------------------------------------------------ FLAT
const (
	SelectReference = int(C.COLTT_SELECT_REFERENCE) // what edge.PriorityQueue does: keeps the K LARGEST distances (priority_queue.go:39-55)
	SelectNearest   = int(C.COLTT_SELECT_NEAREST)
	ModeExact       = int(C.COLTT_MODE_EXACT)
	ModeMFMA        = int(C.COLTT_MODE_MFMA)
)

func FlatCreate(dim uint32, metric, quant int) (Handle, error) {
	var h Handle
	err := call(func() C.int { return C.coltt_flat_create(C.uint32_t(dim), C.int(metric), C.int(quant), &h) })
	return h, err
}
func FlatDestroy(h Handle) { C.coltt_flat_destroy(h) }
func FlatLen(h Handle) int64 {
	var n C.uint64_t
	C.coltt_flat_len(h, &n)
	return int64(n)
}
func FlatUpsert(h Handle, dim uint32, ids []uint64, vecs []float32) error {
	if len(ids) == 0 {
		return nil
	}
	if err := checkDim(vecs, dim, len(ids)); err != nil {
		return err
	}
	return call(func() C.int { return C.coltt_flat_upsert(h, uptr(ids), fptr(vecs), C.size_t(len(ids))) })
}
func FlatRemove(h Handle, ids []uint64) error {
	if len(ids) == 0 {
		return nil
	}
	return call(func() C.int { return C.coltt_flat_remove(h, uptr(ids), C.size_t(len(ids))) })
}

// FlatSearch: cand == nil -> VertexSearch over the whole store; otherwise FilterableVertexSearch over the candidate ids.
func FlatSearch(h Handle, dim uint32, queries []float32, nq int, k uint32, sel, mode int, cand []uint64, filtered bool) ([]uint64, []float32, []uint32, error) {
	if nq == 0 || k == 0 {
		return nil, nil, make([]uint32, nq), nil
	}
	if err := checkDim(queries, dim, nq); err != nil {
		return nil, nil, nil, err
	}
	ids := make([]uint64, nq*int(k))
	sc := make([]float32, nq*int(k))
	cnt := make([]uint32, nq)
	cp := (*C.uint32_t)(unsafe.Pointer(&cnt[0]))
	var err error
	if !filtered {
		err = call(func() C.int {
			return C.coltt_flat_search(h, fptr(queries), C.size_t(nq), C.uint32_t(k), C.int(sel), C.int(mode), uptr(ids), fptr(sc), cp)
		})
	} else {
		err = call(func() C.int {
			// mode: ModeExact = the exact-order gather scan (one query per RPC); ModeMfma = candidates from the gathered rows on the matrix cores
			return C.coltt_flat_search_ids_mode(h, fptr(queries), C.size_t(nq), C.uint32_t(k), C.int(sel), C.int(mode), uptr(cand), C.size_t(len(cand)), uptr(ids), fptr(sc), cp)
		})
	}
	return ids, sc, cnt, err
}

// FlatSaveVertex / FlatLoadVertex: the edge `.vertex` stream (none_vectorstore.go:308-516 and the f16/f8/bf16 twins).
// FlatOneLaunchSearches: how many searches of <= 4 queries the one-launch kernel served (coltt_flat_one_launch_searches) —
// a diagnostic for dashboards: the RPC path of the reference issues exactly this shape (edge/edge.go:610-690).
func FlatOneLaunchSearches(h Handle) (uint64, error) {
	var n C.uint64_t
	err := call(func() C.int { return C.coltt_flat_one_launch_searches(h, &n) })
	return uint64(n), err
}

func FlatSaveVertex(h Handle, metaIds []uint64, metaBlobs [][]byte) ([]byte, error) {
	n := len(metaIds)
	var cptr **C.uint8_t
	var clen *C.uint32_t
	var pins []unsafe.Pointer
	if n > 0 {
		cptr = (**C.uint8_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))
		clen = (*C.uint32_t)(C.malloc(C.size_t(n) * 4))
		defer C.free(unsafe.Pointer(cptr))
		defer C.free(unsafe.Pointer(clen))
		ps := unsafe.Slice(cptr, n)
		ls := unsafe.Slice(clen, n)
		for i, b := range metaBlobs {
			p := C.CBytes(b)
			pins = append(pins, p)
			ps[i], ls[i] = (*C.uint8_t)(p), C.uint32_t(len(b))
		}
		defer func() {
			for _, p := range pins {
				C.free(p)
			}
		}()
	}
	var need C.uint64_t
	if err := call(func() C.int { return C.coltt_flat_save_vertex(h, uptr(metaIds), cptr, clen, C.uint64_t(n), nil, 0, &need) }); err != nil {
		return nil, err
	}
	out := make([]byte, int(need))
	err := call(func() C.int { return C.coltt_flat_save_vertex(h, uptr(metaIds), cptr, clen, C.uint64_t(n), bptr(out), need, &need) })
	return out[:int(need)], err
}
func FlatLoadVertex(h Handle, data []byte, elemBytes int, dim uint32) (ids []uint64, metaOff []uint64, metaLen []uint32, err error) {
	capN := uint64(len(data))/uint64(16+uint32(elemBytes)*dim) + 1 // key u64 + vecLen u32 + codes + metaCount u32
	ids = make([]uint64, capN)
	metaOff = make([]uint64, capN)
	metaLen = make([]uint32, capN)
	var n C.uint64_t
	err = call(func() C.int {
		return C.coltt_flat_load_vertex(h, bptr(data), C.uint64_t(len(data)), &n, uptr(ids), uptr(metaOff),
			(*C.uint32_t)(unsafe.Pointer(&metaLen[0])), C.uint64_t(capN))
	})
	if err != nil {
		return nil, nil, nil, err
	}
	return ids[:n], metaOff[:n], metaLen[:n], nil
}
