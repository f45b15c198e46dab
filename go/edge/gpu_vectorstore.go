// gpu_vectorstore.go — ONE EXTRA FILE for package github.com/sjy-dv/coltt/edge: a GPU-backed edge.vectorspace
// (edge/vectorstore.go:30-49, all 16 methods) serving all four quantisations from one type.
//
// `vectorspace` is unexported, so the drop-in lives INSIDE package edge: copy this file to edge/gpu_vectorstore.go and
// route Vectorstore.CreateCollection / FillEmpty to it (two-line patches shown in INTEGRATION.md).  Everything that is
// not vector arithmetic stays the reference's own code and is only CALLED here: Metadata, standardAnalyzer,
// dropKeyAnalyzer, inverted.BitmapIndex, FilterExpression, SearchResultItem, ENode.
//   vectors + distances + top-k ........ libcoltt_gpu.so (go/colttgpu): Normalize and Lower are applied by the library, so
//                                         the stored bits equal {none,f16,f8,bf16}VecSpace's (none_vectorstore.go:95-101,
//                                         f16_vectorstore.go:97-105)
//   id -> metadata maps, inverted index . here (16 shards by sharding.ShardVertex, as the reference keeps them)
// NOT COMPILED in the build container (no Go toolchain).
package edge

import (
	"bytes"
	"encoding/binary"
	"encoding/json"
	"fmt"
	"math"
	"sync"

	"github.com/sjy-dv/coltt/gen/protoc/v4/edgepb"
	"github.com/sjy-dv/coltt/go/colttgpu"
	"github.com/sjy-dv/coltt/pkg/inverted"
	"github.com/sjy-dv/coltt/pkg/sharding"
)

type gpuVecSpace struct {
	vertexMetadata Metadata
	collectionName string
	h              colttgpu.Handle
	created        bool
	createErr      error
	meta           [EDGE_MAP_SHARD_COUNT]map[uint64]map[string]interface{}
	metaMu         [EDGE_MAP_SHARD_COUNT]*sync.RWMutex
	invertedIndex  *inverted.BitmapIndex
	// SelectNearest = useful direction; SelectReference reproduces edge.PriorityQueue bit for bit, which keeps the K
	// FARTHEST vectors (priority_queue.go:39-55; SURVEY.md §0 finding 1).  Default: reference behaviour.
	Select int
	// ModeExact = reference AVX summation order (bit-identical scores); ModeMFMA = matrix-core candidates + exact re-score
	Mode int
}

func newGpuVectorstore(collectionName string, metadata Metadata) *gpuVecSpace {
	s := &gpuVecSpace{vertexMetadata: metadata, collectionName: collectionName, invertedIndex: inverted.NewBitmapIndex(),
		Select: colttgpu.SelectReference, Mode: colttgpu.ModeExact}
	for i := 0; i < EDGE_MAP_SHARD_COUNT; i++ {
		s.meta[i] = make(map[uint64]map[string]interface{})
		s.metaMu[i] = &sync.RWMutex{}
	}
	s.open()
	return s
}

// open creates the device store once the collection metadata (dim, distance, quantisation) is known — at construction, or
// after LoadVertexMetadata for a FillEmpty'd collection.
func (s *gpuVecSpace) open() {
	if s.created || s.vertexMetadata.Dimensional() == 0 {
		return
	}
	s.h, s.createErr = colttgpu.FlatCreate(s.vertexMetadata.Dimensional(), int(s.vertexMetadata.Distancer()),
		int(s.vertexMetadata.Quantizationer())) // edgepb enum order == COLTT_COSINE/EUCLIDEAN, COLTT_Q_*
	s.created = s.createErr == nil
	for i := range s.metaMu {
		if s.metaMu[i] == nil {
			s.meta[i] = make(map[uint64]map[string]interface{})
			s.metaMu[i] = &sync.RWMutex{}
		}
	}
	if s.invertedIndex == nil {
		s.invertedIndex = inverted.NewBitmapIndex()
	}
}

func (s *gpuVecSpace) ready() error {
	if s.createErr != nil {
		return s.createErr // e.g. "not support quantization type" (vectorstore.go:79)
	}
	if !s.created {
		return fmt.Errorf("collection %s: vertex metadata not loaded", s.collectionName)
	}
	return nil
}

func (s *gpuVecSpace) shard(id uint64) uint64 { return sharding.ShardVertex(id, uint64(EDGE_MAP_SHARD_COUNT)) }

// ChangedVertex — none_vectorstore.go:66-103
func (s *gpuVecSpace) ChangedVertex(updateId string, commitId uint64, data ENode) error {
	if err := s.ready(); err != nil {
		return err
	}
	if updateId != "" { // primary-key lookup: an existing key keeps its id (:67-83)
		var primaryIndex string
		for _, indexer := range s.Indexer() {
			if indexer.PrimaryKey {
				primaryIndex = indexer.IndexName
				break
			}
		}
		ids, err := s.invertedIndex.SearchSingleFilter(inverted.NewFilter(primaryIndex, inverted.OpEqual, updateId))
		if err != nil {
			return err
		}
		if len(ids) != 0 {
			commitId = ids[0]
		}
	}
	if s.Dim() != uint32(data.Vector.Dimensions()) {
		return fmt.Errorf("Dim Length UnmatchdError: expect dimension: [%d], but got [%d]", s.Dim(), data.Vector.Dimensions())
	}
	if err := standardAnalyzer(data.Metadata, s.Indexer()); err != nil {
		return err
	}
	if err := s.invertedIndex.Add(commitId, data.Metadata); err != nil {
		return fmt.Errorf("ErrInvertedIndexAddFailed: %s", err.Error())
	}
	sh := s.shard(commitId)
	s.metaMu[sh].Lock() // metadata first: a concurrent search must never return the id without it
	prev, had := s.meta[sh][commitId]
	s.meta[sh][commitId] = data.Metadata
	s.metaMu[sh].Unlock()
	// Normalize (cosine) + Lower happen in the library, on the device
	if err := colttgpu.FlatUpsert(s.h, s.Dim(), []uint64{commitId}, data.Vector); err != nil {
		// the row was not stored: take the metadata and the inverted-index entry back out (or restore the previous ones)
		s.metaMu[sh].Lock()
		if had {
			s.meta[sh][commitId] = prev
		} else {
			delete(s.meta[sh], commitId)
		}
		s.metaMu[sh].Unlock()
		_ = s.invertedIndex.Remove(commitId, data.Metadata)
		if had {
			_ = s.invertedIndex.Add(commitId, prev)
		}
		return err
	}
	return nil
}

// RemoveVertex — none_vectorstore.go:105-127
func (s *gpuVecSpace) RemoveVertex(dropFilter map[string]interface{}) error {
	if err := s.ready(); err != nil {
		return err
	}
	if err := dropKeyAnalyzer(dropFilter, s.Indexer()); err != nil {
		return err
	}
	filters := make([]*inverted.Filter, 0, len(dropFilter))
	for index, indexValue := range dropFilter {
		filters = append(filters, inverted.NewFilter(index, inverted.OpEqual, indexValue))
	}
	dropIds, err := s.invertedIndex.SearchMultiFilter(filters)
	if err != nil {
		return fmt.Errorf("InvertedIndexFindDeleteIdsError: %s", err.Error())
	}
	for _, id := range dropIds {
		sh := s.shard(id)
		s.metaMu[sh].Lock()
		s.invertedIndex.Remove(id, s.meta[sh][id])
		delete(s.meta[sh], id)
		s.metaMu[sh].Unlock()
	}
	return colttgpu.FlatRemove(s.h, dropIds)
}

func (s *gpuVecSpace) results(ids []uint64, sc []float32, n int) []*SearchResultItem {
	out := make([]*SearchResultItem, n)
	for i := 0; i < n; i++ {
		sh := s.shard(ids[i])
		s.metaMu[sh].RLock()
		out[i] = &SearchResultItem{Id: ids[i], Score: sc[i], Metadata: s.meta[sh][ids[i]]}
		s.metaMu[sh].RUnlock()
	}
	return out
}

// VertexSearch — none_vectorstore.go:129-180 (highCpu only chooses how the CPU splits the scan; the answer is the same)
func (s *gpuVecSpace) VertexSearch(target Vector, topK int, highCpu bool) ([]*SearchResultItem, error) {
	if err := s.ready(); err != nil {
		return nil, err
	}
	if topK <= 0 {
		return []*SearchResultItem{}, nil
	}
	ids, sc, cnt, err := colttgpu.FlatSearch(s.h, s.Dim(), target, 1, uint32(topK), s.Select, s.Mode, nil, false)
	if err != nil {
		return nil, err
	}
	return s.results(ids, sc, int(cnt[0])), nil
}

// FilterableVertexSearch — none_vectorstore.go:182-253: candidates from the roaring index, ids not stored are skipped (:201)
func (s *gpuVecSpace) FilterableVertexSearch(filter *inverted.FilterExpression, target Vector, topK int, highCpu bool) (
	[]*SearchResultItem, error) {
	if err := s.ready(); err != nil {
		return nil, err
	}
	candidates, err := s.invertedIndex.SearchWithExpression(filter)
	if err != nil {
		return nil, err
	}
	if topK <= 0 {
		return []*SearchResultItem{}, nil
	}
	ids, sc, cnt, err := colttgpu.FlatSearch(s.h, s.Dim(), target, 1, uint32(topK), s.Select, colttgpu.ModeExact, candidates, true)
	if err != nil {
		return nil, err
	}
	return s.results(ids, sc, int(cnt[0])), nil
}

func (s *gpuVecSpace) SaveVertexMetadata() ([]byte, error) { return json.Marshal(s.vertexMetadata) }

// LoadVertexMetadata — none_vectorstore.go:259-274; the device store can only be created now for a FillEmpty'd collection
func (s *gpuVecSpace) LoadVertexMetadata(collectionName string, data []byte) error {
	var metadata Metadata
	if err := json.Unmarshal(data, &metadata); err != nil {
		return err
	}
	s.collectionName = collectionName
	s.vertexMetadata = metadata
	s.open()
	return s.ready()
}

func (s *gpuVecSpace) SaveVertexInverted() ([]byte, error) { return s.invertedIndex.SerializeBinary() }
func (s *gpuVecSpace) LoadVertexInverted(data []byte) error {
	s.invertedIndex = inverted.NewBitmapIndex()
	return s.invertedIndex.DeserializeBinary(data)
}

// typed metadata pairs of the `.vertex` stream (none_vectorstore.go:338-413): u32 count, each {u16 keylen, key, u8 tag,
// value}: 0 int64, 1 {u16 len, string}, 2 float64 (float32 widened), 3 bool byte
func encodeVertexMeta(m map[string]interface{}) ([]byte, error) {
	var b bytes.Buffer
	binary.Write(&b, binary.BigEndian, uint32(len(m)))
	for k, v := range m {
		if len(k) > 65535 {
			return nil, fmt.Errorf("metadata key too long: %s", k)
		}
		binary.Write(&b, binary.BigEndian, uint16(len(k)))
		b.WriteString(k)
		switch t := v.(type) {
		case int64:
			b.WriteByte(0)
			binary.Write(&b, binary.BigEndian, t)
		case string:
			if len(t) > 65535 {
				return nil, fmt.Errorf("metadata string too long: %s", t)
			}
			b.WriteByte(1)
			binary.Write(&b, binary.BigEndian, uint16(len(t)))
			b.WriteString(t)
		case float32:
			b.WriteByte(2)
			binary.Write(&b, binary.BigEndian, float64(t))
		case float64:
			b.WriteByte(2)
			binary.Write(&b, binary.BigEndian, t)
		case bool:
			b.WriteByte(3)
			if t {
				b.WriteByte(1)
			} else {
				b.WriteByte(0)
			}
		default:
			return nil, fmt.Errorf("unsupported metadata type: %T", v)
		}
	}
	return b.Bytes(), nil
}

func decodeVertexMeta(blob []byte) (map[string]interface{}, error) {
	if len(blob) < 4 {
		return nil, fmt.Errorf("truncated metadata")
	}
	n := binary.BigEndian.Uint32(blob)
	p := 4
	need := func(k int) error {
		if p+k > len(blob) {
			return fmt.Errorf("truncated metadata")
		}
		return nil
	}
	m := make(map[string]interface{}, n)
	for i := uint32(0); i < n; i++ {
		if err := need(2); err != nil {
			return nil, err
		}
		kl := int(binary.BigEndian.Uint16(blob[p:]))
		p += 2
		if err := need(kl + 1); err != nil {
			return nil, err
		}
		key := string(blob[p : p+kl])
		p += kl
		tag := blob[p]
		p++
		switch tag {
		case 0:
			if err := need(8); err != nil {
				return nil, err
			}
			m[key] = int64(binary.BigEndian.Uint64(blob[p:]))
			p += 8
		case 1:
			if err := need(2); err != nil {
				return nil, err
			}
			sl := int(binary.BigEndian.Uint16(blob[p:]))
			p += 2
			if err := need(sl); err != nil {
				return nil, err
			}
			m[key] = string(blob[p : p+sl])
			p += sl
		case 2:
			if err := need(8); err != nil {
				return nil, err
			}
			m[key] = math.Float64frombits(binary.BigEndian.Uint64(blob[p:]))
			p += 8
		case 3:
			if err := need(1); err != nil {
				return nil, err
			}
			m[key] = blob[p] != 0
			p++
		default:
			return nil, fmt.Errorf("unsupported metadata type tag: %d", tag)
		}
	}
	return m, nil
}

// SaveVertex — none_vectorstore.go:308-423 (and the f16/f8/bf16 twins): the library emits the shards, keys and stored codes
// straight from HBM; the typed metadata blobs come from here.
func (s *gpuVecSpace) SaveVertex() ([]byte, error) {
	if err := s.ready(); err != nil {
		return nil, err
	}
	var ids []uint64
	var blobs [][]byte
	for sh := 0; sh < EDGE_MAP_SHARD_COUNT; sh++ {
		s.metaMu[sh].RLock()
		for id, m := range s.meta[sh] {
			b, err := encodeVertexMeta(m)
			if err != nil {
				s.metaMu[sh].RUnlock()
				return nil, err
			}
			ids = append(ids, id)
			blobs = append(blobs, b)
		}
		s.metaMu[sh].RUnlock()
	}
	return colttgpu.FlatSaveVertex(s.h, ids, blobs)
}

// LoadVertex — none_vectorstore.go:425-516: codes go straight into the HBM rows (no re-normalisation, as in the reference)
func (s *gpuVecSpace) LoadVertex(data []byte) error {
	if err := s.ready(); err != nil {
		return err
	}
	eb := 4
	switch s.Quantization() {
	case edgepb.Quantization_F16, edgepb.Quantization_BF16:
		eb = 2
	case edgepb.Quantization_F8:
		eb = 1
	}
	ids, off, ln, err := colttgpu.FlatLoadVertex(s.h, data, eb, s.Dim())
	if err != nil {
		return err
	}
	var fresh [EDGE_MAP_SHARD_COUNT]map[uint64]map[string]interface{}
	for i := range fresh {
		fresh[i] = make(map[uint64]map[string]interface{})
	}
	for i, id := range ids {
		m, err := decodeVertexMeta(data[off[i] : off[i]+uint64(ln[i])])
		if err != nil {
			return err
		}
		fresh[s.shard(id)][id] = m
	}
	for i := range fresh {
		s.metaMu[i].Lock()
		s.meta[i] = fresh[i]
		s.metaMu[i].Unlock()
	}
	return nil
}

func (s *gpuVecSpace) Quantization() edgepb.Quantization { return s.vertexMetadata.Quantizationer() }
func (s *gpuVecSpace) Distance() edgepb.Distance         { return s.vertexMetadata.Distancer() }
func (s *gpuVecSpace) Dim() uint32                       { return s.vertexMetadata.Dimensional() }
func (s *gpuVecSpace) LoadSize() int64 {
	if !s.created {
		return 0
	}
	return colttgpu.FlatLen(s.h)
}
func (s *gpuVecSpace) Indexer() map[string]IndexFeature { return s.vertexMetadata.IndexType }
func (s *gpuVecSpace) Versional() bool                  { return s.vertexMetadata.Versional() }

var _ vectorspace = (*gpuVecSpace)(nil) // all 16 methods of edge/vectorstore.go:30-49
