// Package vectorindex — GPU-backed REPLACEMENT BODY for github.com/sjy-dv/coltt/core/vectorindex.
//
// `core` stores the concrete type *vectorindex.Hnsw (core/core.go:34,51), so there is no interface to plug into: the
// drop-in is source level — this directory replaces core/vectorindex and exports every identifier core, e2e and playground
// use (usage census SURVEY.md §8b): Hnsw, NewHnsw, HnswOption + the nine option constructors, HnswSearchSimple /
// HnswSearchHeuristic, ProtoConfig, Metadata, SearchResult{,Item}, Normalize, ItemNotFoundError, ItemAlreadyExistsError,
// and GetVertex whose result offers Id / Vector / Metadata / Level.
//
// Division of labour: vectors, the graph and every distance evaluation live in HBM behind libcoltt_gpu.so
// (go/colttgpu); Metadata maps never cross the boundary — this package keeps id -> Metadata and re-attaches it.
// NOT COMPILED in the build container (no Go toolchain); see INTEGRATION.md for the build line.
package vectorindex

import (
	"bytes"
	"context"
	"encoding/binary"
	"errors"
	"fmt"
	"io"
	"math"
	"math/rand"
	"strings"
	"sync"
	"sync/atomic"

	"github.com/sjy-dv/coltt/edge"
	"github.com/sjy-dv/coltt/go/colttgpu"
	"github.com/sjy-dv/coltt/pkg/distance"
	"github.com/vmihailenco/msgpack/v5"
)

var (
	ItemNotFoundError      error = errors.New("Item not found")      // hnsw.go:39
	ItemAlreadyExistsError error = errors.New("Item already exists") // hnsw.go:40
)

// ---------------------------------------------------------------------------------------------- config (hnsw_config.go)
var hnswSearchAlgorithmNames = [...]string{"Simple", "Heuristic", "Diverse"}

type hnswSearchAlgorithm int

const (
	HnswSearchSimple hnswSearchAlgorithm = iota
	HnswSearchHeuristic
	// HnswSearchDiverse is NOT in the reference (COLTT_HNSW_DIVERSE, include/coltt_gpu.h): neighbour selection with the diversity test of
	// the HNSW paper, which the reference's "heuristic" (hnsw.go:399-447) does not apply.  It builds a DIFFERENT graph than the reference
	// would: NewHnsw refuses it unless HnswAllowNonReferenceSelection(true) is passed as well.
	HnswSearchDiverse
)

func (a hnswSearchAlgorithm) String() string { return hnswSearchAlgorithmNames[a] }

type hnswConfig struct {
	searchAlgorithm           hnswSearchAlgorithm
	levelMultiplier           float32
	ef, efConstruction        int
	m, mMax, mMax0            int
	heuristicExtendCandidates bool
	heuristicKeepPruned       bool
	quantization              int // extension (BASELINE.json configs[4]): 0 none, 1 f16, 2 f8, 3 "bf16" — edgepb.Quantization order
	allowNonReference         bool // opt-in for HnswSearchDiverse
}

// HnswOption — functional options, hnsw_config.go:43-109
type HnswOption interface{ apply(*hnswConfig) }
type hnswOption struct{ f func(*hnswConfig) }

func (o *hnswOption) apply(c *hnswConfig) { o.f(c) }

func HnswLevelMultiplier(v float32) HnswOption {
	return &hnswOption{func(c *hnswConfig) { c.levelMultiplier = v }}
}
func HnswEf(v int) HnswOption             { return &hnswOption{func(c *hnswConfig) { c.ef = v }} }
func HnswEfConstruction(v int) HnswOption { return &hnswOption{func(c *hnswConfig) { c.efConstruction = v }} }
func HnswM(v int) HnswOption              { return &hnswOption{func(c *hnswConfig) { c.m = v }} }
func HnswMmax(v int) HnswOption           { return &hnswOption{func(c *hnswConfig) { c.mMax = v }} }
func HnswMmax0(v int) HnswOption          { return &hnswOption{func(c *hnswConfig) { c.mMax0 = v }} }
func HnswSearchAlgorithm(v hnswSearchAlgorithm) HnswOption {
	return &hnswOption{func(c *hnswConfig) { c.searchAlgorithm = v }}
}
func HnswHeuristicExtendCandidates(v bool) HnswOption {
	return &hnswOption{func(c *hnswConfig) { c.heuristicExtendCandidates = v }}
}
func HnswHeuristicKeepPruned(v bool) HnswOption {
	return &hnswOption{func(c *hnswConfig) { c.heuristicKeepPruned = v }}
}

// HnswAllowNonReferenceSelection opts in to HnswSearchDiverse: without it a drop-in user can never get a graph the reference would not build.
func HnswAllowNonReferenceSelection(v bool) HnswOption {
	return &hnswOption{func(c *hnswConfig) { c.allowNonReference = v }}
}

// HnswQuantization is NOT in the reference: stored rows become 2-/1-byte codes scored as the edge quantised stores do.
func HnswQuantization(q int) HnswOption { return &hnswOption{func(c *hnswConfig) { c.quantization = q }} }

// ProtoConfig — hnsw_config.go:123-133
type ProtoConfig struct {
	SearchAlgorithm           string
	LevelMultiplier           float32
	Ef                        int
	EfConstruction            int
	M                         int
	MMax                      int
	MMax0                     int
	HeuristicExtendCandidates bool
	HeuristicKeepPruned       bool
}

func b2i(b bool) int32 {
	if b {
		return 1
	}
	return 0
}

// ---------------------------------------------------------------------------------------------- Metadata (metadata.go)
type Metadata map[string]any

// encode = Metadata.save (metadata.go:31-74): u16 pairs, each {u8 keylen, key, u16 vallen, msgpack(value)}
func (m Metadata) encode() ([]byte, error) {
	var b bytes.Buffer
	if err := binary.Write(&b, binary.BigEndian, uint16(len(m))); err != nil {
		return nil, err
	}
	for k, v := range m {
		if len(k) > 255 {
			return nil, fmt.Errorf("metadata key too long: %s", k)
		}
		b.WriteByte(uint8(len(k)))
		b.WriteString(k)
		vb, err := msgpack.Marshal(v)
		if err != nil {
			return nil, err
		}
		if err := binary.Write(&b, binary.BigEndian, uint16(len(vb))); err != nil {
			return nil, err
		}
		b.Write(vb)
	}
	return b.Bytes(), nil
}

// decodeMetadata = Metadata.load (metadata.go:43-105) over the blob the library located inside the stream
func decodeMetadata(blob []byte) (Metadata, error) {
	r := bytes.NewReader(blob)
	var n uint16
	if err := binary.Read(r, binary.BigEndian, &n); err != nil {
		return nil, err
	}
	m := make(Metadata, n)
	for i := 0; i < int(n); i++ {
		var kl uint8
		if err := binary.Read(r, binary.BigEndian, &kl); err != nil {
			return nil, err
		}
		kb := make([]byte, kl)
		if _, err := io.ReadFull(r, kb); err != nil {
			return nil, err
		}
		var vl uint16
		if err := binary.Read(r, binary.BigEndian, &vl); err != nil {
			return nil, err
		}
		vb := make([]byte, vl)
		if _, err := io.ReadFull(r, vb); err != nil {
			return nil, err
		}
		var v any
		if err := msgpack.Unmarshal(vb, &v); err != nil {
			return nil, err
		}
		m[string(kb)] = v
	}
	return m, nil
}

// Normalize — metadata.go:107-123.  Kept as an identifier for callers (core/core_helper.go).  The arithmetic has ONE copy, in the
// library, and this calls its host-side entry (coltt_normalize_host): no device work, no allocation beyond the result, infallible — as
// the reference's.
func Normalize(v []float32) []float32 { return colttgpu.Normalize(v) }

// ---------------------------------------------------------------------------------------------- results (search.go)
type SearchResult []SearchResultItem
type SearchResultItem struct {
	Id       uint64
	Metadata map[string]any
	Score    float32
}

func (xx SearchResult) Len() int           { return len(xx) }
func (xx SearchResult) Swap(i, j int)      { xx[i], xx[j] = xx[j], xx[i] }
func (xx SearchResult) Less(i, j int) bool { return xx[i].Score < xx[j].Score }

// hnswVertex — what GetVertex hands out (core/core.go:512-517,607 read .Metadata()); a value snapshot, not a live node
type hnswVertex struct {
	id       uint64
	vector   edge.Vector
	metadata Metadata
	level    int
}

func (v *hnswVertex) Id() uint64          { return v.id }
func (v *hnswVertex) Vector() edge.Vector { return v.vector }
func (v *hnswVertex) Metadata() Metadata  { return v.metadata }
func (v *hnswVertex) Level() int          { return v.level }

// ---------------------------------------------------------------------------------------------- Hnsw (hnsw.go:43-54)
const metaShards = 16

type Hnsw struct {
	dim       uint
	distancer distance.Space
	config    *hnswConfig
	h         colttgpu.Handle
	bytesSize uint64 // sum of vector + metadata bytes of live vertices (hnsw_vertex.go:123-127), Go side
	meta      [metaShards]map[uint64]Metadata
	metaMu    [metaShards]sync.RWMutex
	commitMu  sync.RWMutex // Insert / Remove hold it shared, Commit exclusive: a snapshot never interleaves with a mutation
	err       error // creation error, surfaced by the first call (NewHnsw has no error result in the reference)
}

func metric(d distance.Space) int {
	if d.Type() == "cosine-dot" { // pkg/distance/space.go:101
		return 0
	}
	return 1
}

// NewHnsw(dim, distancer, options...) — hnsw.go:56-73; defaults hnsw_config.go:135-162 are filled in by the library (-1).
func NewHnsw(dim uint, distancer distance.Space, option ...HnswOption) *Hnsw {
	c := &hnswConfig{searchAlgorithm: HnswSearchSimple, levelMultiplier: -1, ef: 20, efConstruction: 200, m: 16, mMax: -1, mMax0: -1,
		heuristicKeepPruned: true}
	for _, o := range option {
		o.apply(c)
	}
	x := &Hnsw{dim: dim, distancer: distancer, config: c}
	for i := range x.meta {
		x.meta[i] = make(map[uint64]Metadata)
	}
	if c.searchAlgorithm == HnswSearchDiverse && !c.allowNonReference {
		x.err = fmt.Errorf("HnswSearchDiverse is not reference behaviour: pass HnswAllowNonReferenceSelection(true) to opt in")
		return x
	}
	cfg := colttgpu.HnswCfg{M: int32(c.m), MMax: int32(c.mMax), MMax0: int32(c.mMax0), Ef: int32(c.ef), EfConstruction: int32(c.efConstruction),
		Algo: int32(c.searchAlgorithm), LevelMultiplier: c.levelMultiplier, ExtendCandidates: b2i(c.heuristicExtendCandidates),
		KeepPruned: b2i(c.heuristicKeepPruned)}
	x.h, x.err = colttgpu.HnswCreate(uint32(dim), metric(distancer), c.quantization, &cfg)
	if x.err == nil {
		x.refreshConfig()
	}
	return x
}

func (xx *Hnsw) refreshConfig() {
	if c, err := colttgpu.HnswGetCfg(xx.h); err == nil {
		xx.config.searchAlgorithm = hnswSearchAlgorithm(c.Algo)
		xx.config.levelMultiplier = c.LevelMultiplier
		xx.config.ef, xx.config.efConstruction = int(c.Ef), int(c.EfConstruction)
		xx.config.m, xx.config.mMax, xx.config.mMax0 = int(c.M), int(c.MMax), int(c.MMax0)
		xx.config.heuristicExtendCandidates, xx.config.heuristicKeepPruned = c.ExtendCandidates != 0, c.KeepPruned != 0
	}
}

func (xx *Hnsw) Close() { colttgpu.HnswDestroy(xx.h) }

func (xx *Hnsw) Info() string {
	c := xx.config
	return fmt.Sprintf("HNSW(dim: %d, distancer: %s, config={searchAlgorithm: %s, ef: %d, efConstruction: %d, m: %d, mMax: %d, mMax0: %d, levelMultiplier: %.4f, extendCandidates: %t, keepPruned: %t})",
		xx.dim, xx.distancer.Type(), c.searchAlgorithm, c.ef, c.efConstruction, c.m, c.mMax, c.mMax0, c.levelMultiplier,
		c.heuristicExtendCandidates, c.heuristicKeepPruned)
}
func (xx *Hnsw) Dim() uint32 { return uint32(xx.dim) }
func (xx *Hnsw) Len() int    { return colttgpu.HnswLen(xx.h) }
func (xx *Hnsw) Config() ProtoConfig { // hnsw.go:86-98
	c := xx.config
	return ProtoConfig{SearchAlgorithm: strings.ToLower(c.searchAlgorithm.String()), LevelMultiplier: c.levelMultiplier, Ef: c.ef,
		EfConstruction: c.efConstruction, M: c.m, MMax: c.mMax, MMax0: c.mMax0,
		HeuristicExtendCandidates: c.heuristicExtendCandidates, HeuristicKeepPruned: c.heuristicKeepPruned}
}
func (xx *Hnsw) Distance() string { return xx.distancer.Type() } // hnsw.go:100-102

func (xx *Hnsw) shard(id uint64) int { return int(id % metaShards) }

func mapErr(err error) error {
	switch err {
	case colttgpu.ErrNotFound:
		return ItemNotFoundError
	case colttgpu.ErrExists:
		return ItemAlreadyExistsError
	}
	return err
}

func metaBytes(m Metadata) uint64 { // coarse stand-in for Metadata.byteSize (metadata.go:125-132)
	var n uint64
	for k, v := range m {
		n += uint64(len(k)) + 16
		if s, ok := v.(string); ok {
			n += uint64(len(s))
		}
	}
	return n
}

// Insert(id, value, metadata, vertexLevel) — hnsw.go:104-167
func (xx *Hnsw) Insert(id uint64, value edge.Vector, metadata Metadata, vertexLevel int) error {
	if xx.err != nil {
		return xx.err
	}
	xx.commitMu.RLock()
	defer xx.commitMu.RUnlock()
	s := xx.shard(id)
	// the metadata is in place BEFORE the id becomes searchable (a concurrent Search must never see the id without it);
	// on failure it is taken out again unless the id already existed
	xx.metaMu[s].Lock()
	_, had := xx.meta[s][id]
	if !had {
		xx.meta[s][id] = metadata
	}
	xx.metaMu[s].Unlock()
	if err := colttgpu.HnswInsert(xx.h, uint32(xx.dim), id, value, vertexLevel); err != nil {
		if !had {
			xx.metaMu[s].Lock()
			delete(xx.meta[s], id)
			xx.metaMu[s].Unlock()
		}
		return mapErr(err)
	}
	atomic.AddUint64(&xx.bytesSize, uint64(len(value))*4+metaBytes(metadata))
	return nil
}

// Get(id) — hnsw.go:169-178
func (xx *Hnsw) Get(id uint64) (edge.Vector, error) {
	v, _, err := colttgpu.HnswGet(xx.h, uint32(xx.dim), id)
	if err != nil {
		return nil, mapErr(err)
	}
	return edge.Vector(v), nil
}

// GetVertex(id) — hnsw.go:180-189
func (xx *Hnsw) GetVertex(id uint64) (*hnswVertex, error) {
	v, lv, err := colttgpu.HnswGet(xx.h, uint32(xx.dim), id)
	if err != nil {
		return nil, mapErr(err)
	}
	s := xx.shard(id)
	xx.metaMu[s].RLock()
	m := xx.meta[s][id]
	xx.metaMu[s].RUnlock()
	return &hnswVertex{id: id, vector: edge.Vector(v), metadata: m, level: lv}, nil
}

// Remove(id) — hnsw.go:191-241
func (xx *Hnsw) Remove(id uint64) error {
	xx.commitMu.RLock()
	defer xx.commitMu.RUnlock()
	if err := colttgpu.HnswRemove(xx.h, id); err != nil {
		return mapErr(err)
	}
	s := xx.shard(id)
	xx.metaMu[s].Lock()
	m := xx.meta[s][id]
	delete(xx.meta[s], id)
	xx.metaMu[s].Unlock()
	atomic.AddUint64(&xx.bytesSize, ^(uint64(xx.dim)*4 + metaBytes(m) - 1))
	return nil
}

// Search(ctx, query, k) — hnsw.go:243-278.  One query per call as in the reference; Batcher (batcher.go) coalesces callers.
func (xx *Hnsw) Search(_ context.Context, query edge.Vector, k uint) (SearchResult, error) {
	if xx.err != nil {
		return nil, xx.err
	}
	ids, sc, cnt, err := colttgpu.HnswSearch(xx.h, uint32(xx.dim), query, 1, uint32(k), 0)
	if err != nil {
		return nil, err
	}
	return xx.attach(ids, sc, int(cnt[0])), nil
}

func (xx *Hnsw) attach(ids []uint64, sc []float32, n int) SearchResult {
	res := make(SearchResult, n)
	for i := 0; i < n; i++ {
		s := xx.shard(ids[i])
		xx.metaMu[s].RLock()
		res[i] = SearchResultItem{Id: ids[i], Score: sc[i], Metadata: xx.meta[s][ids[i]]}
		xx.metaMu[s].RUnlock()
	}
	return res
}

// RandomLevel — hnsw.go:280-282: the uniform draw stays with Go's global math/rand (gomath/rand.go:42-44), the library applies
// gomath.Floor(-gomath.Log(u) * levelMultiplier).  u = 0 (the reference's Floor(+Inf)) is redrawn.
func (xx *Hnsw) RandomLevel() int {
	u := rand.Float32()
	for u <= 0 {
		u = rand.Float32()
	}
	lv, err := colttgpu.HnswRandomLevel(xx.h, u)
	if err != nil {
		return 0
	}
	return lv
}

// BytesSize — hnsw.go:476-490 (HNSW_VERTEX_EDGE_BYTES = 12, HNSW_VERTEX_MUTEX_BYTES = 24: the HOST-side estimate the
// reference reports; the HBM footprint is Len() * (row stride + 2 * 4 * mMax0) + upper rows)
func (xx *Hnsw) BytesSize() uint64 {
	maxLevel := 10
	if lv, err := colttgpu.HnswEntryLevel(xx.h); err == nil && lv >= 0 { // host-side field of the index: no adjacency is copied
		maxLevel = lv
	}
	const edgeB, mutB = 12.0, 24.0
	ptr := float64(xx.config.mMax0)*edgeB + mutB
	for i := 1; i < maxLevel; i++ {
		ptr += (float64(xx.config.mMax)*edgeB + mutB) * math.Exp(float64(i)/-float64(xx.config.levelMultiplier))
	}
	return uint64(math.Floor(float64(xx.Len())*ptr)) + atomic.LoadUint64(&xx.bytesSize)
}

// Commit(w, header) — hnsw_commit.go:69-162: the reference's big-endian stream, produced by the library from the HBM layout.
func (xx *Hnsw) Commit(w io.Writer, header bool) error {
	// Snapshot against mutations: the reference's Commit walks the live maps without a lock; here the slot list, the metadata
	// blobs and the stream are produced under commitMu so an Insert / Remove cannot interleave (they take it shared).
	xx.commitMu.Lock()
	defer xx.commitMu.Unlock()
	ids, deleted, err := colttgpu.HnswSlots(xx.h)
	if err != nil {
		return err
	}
	blobs := make([][]byte, len(ids))
	for slot, id := range ids {
		if deleted[slot] != 0 {
			continue
		}
		s := xx.shard(id)
		xx.metaMu[s].RLock()
		m := xx.meta[s][id]
		xx.metaMu[s].RUnlock()
		if blobs[slot], err = m.encode(); err != nil {
			return err
		}
	}
	out, err := colttgpu.HnswCommit(xx.h, header, blobs)
	if err != nil {
		return err
	}
	_, err = w.Write(out)
	return err
}

// Load(r, header) — hnsw_commit.go:164-278: the stream goes straight into HBM; metadata blobs are decoded here.
func (xx *Hnsw) Load(r io.Reader, header bool) error {
	data, err := io.ReadAll(r)
	if err != nil {
		return err
	}
	ids, off, ln, err := colttgpu.HnswLoad(xx.h, header, data, uint32(xx.dim))
	if err != nil {
		return err
	}
	fresh := [metaShards]map[uint64]Metadata{}
	for i := range fresh {
		fresh[i] = make(map[uint64]Metadata)
	}
	var size uint64
	for i, id := range ids {
		m, err := decodeMetadata(data[off[i] : off[i]+uint64(ln[i])])
		if err != nil {
			return err
		}
		fresh[xx.shard(id)][id] = m
		size += uint64(xx.dim)*4 + metaBytes(m)
	}
	for i := range fresh {
		xx.metaMu[i].Lock()
		xx.meta[i] = fresh[i]
		xx.metaMu[i].Unlock()
	}
	atomic.StoreUint64(&xx.bytesSize, size)
	xx.refreshConfig()
	return nil
}
