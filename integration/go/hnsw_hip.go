// UNTESTED SKETCH: no Go toolchain exists in the build image, nothing compiles or runs this file.  The routing rules
// below are those of kektor::hnsw::MicroBatcher (include/kektor_hip.hpp), which is compiled, sanitised and run on the GPU.

//go:build hip

// hnsw_hip.go -- the binding a KektorDB maintainer adds to package hnsw to route SearchWithScores through
// libkektor_hip.so on an MI355X.  It lives INSIDE package hnsw (pkg/core/hnsw/hnsw_hip.go) because the engine
// type-asserts *hnsw.Index at 36 sites (pkg/engine/ops.go:36-42), so the GPU path has to be a field of
// hnsw.Index rather than a new core.VectorIndex implementation.  Selection follows the reference's own
// convention for native code: a build tag (`rust`: pkg/core/distance/distance_rust.go:1,12-17).
//
// NOT COMPILED IN THIS REPOSITORY: the build image has no Go toolchain.  The C ABI it binds is exercised end to
// end by kektordb_amd/index.py (ctypes) and include/kektor_hip.hpp (C++), which follow the same ownership rules.
//
// What the file adds
//   gpuMirror        HBM copy of one index (rows + graph + deleted bits): a derived cache, rebuilt from the
//                    arena rows and Node.Connections when a writer has bumped the epoch; nothing new to persist.
//   hipBatcher       micro-batcher: the reference API is one query per call from many goroutines
//                    (ops.go:1003-1007); concurrent callers that share (k, efSearch, allowList) are grouped
//                    into ONE kdb_search_batch / kdb_flat_scan_batch call.
//   searchHIP        what SearchWithScores calls under the tag (one line in hnsw_index.go:355).
//
// Required hooks in the existing code (all one-liners):
//   Index            gets two fields:  gpu gpuMirror;  writeEpoch atomic.Uint64
//   Add/AddBatch/Delete/Vacuum/optimizer commits:  h.writeEpoch.Add(1)
//   Close (hnsw_index.go:3533-3586), after the activeMu drain:  h.gpu.destroy()
package hnsw

/*
#cgo CFLAGS: -I${SRCDIR}/../../../third_party/kektor-hip/include
#cgo LDFLAGS: -lkektor_hip
#include <stdlib.h>
#include "kektor_hip.h"
*/
import "C"

import (
	"math"
	"errors"
	"fmt"
	"log/slog"
	"sync"
	"time"
	"unsafe"

	"github.com/RoaringBitmap/roaring"
	"github.com/sanonone/kektordb/pkg/core/distance"
	"github.com/sanonone/kektordb/pkg/core/types"
)

// ---------------------------------------------------------------------------------------------------------
// codes of include/kektor_hip.h
// ---------------------------------------------------------------------------------------------------------

func hipMetric(m distance.DistanceMetric) C.uint32_t {
	if m == distance.Cosine {
		return C.KDB_METRIC_COSINE
	}
	return C.KDB_METRIC_L2
}

func hipPrecision(p distance.PrecisionType) C.uint32_t {
	switch p {
	case distance.Float16:
		return C.KDB_PREC_F16
	case distance.Int8:
		return C.KDB_PREC_I8
	}
	return C.KDB_PREC_F32
}

func hipErr(what string) error {
	return fmt.Errorf("%s: %s", what, C.GoString(C.kdb_last_error()))
}

// ---------------------------------------------------------------------------------------------------------
// gpuMirror
// ---------------------------------------------------------------------------------------------------------

type gpuMirror struct {
	mu      sync.Mutex
	h       *C.kdb_index
	epoch   uint64 // Index.writeEpoch the mirror was built from
	valid   bool
	batcher *hipBatcher
}

// refresh makes the HBM copy current.  Called with activeMu.RLock held (so Close cannot unmap the arena while
// rows are read) and WITHOUT metaMu: rows and links are read under the same fine-grained shard locks the
// reference's own snapshot export uses (hnsw_index.go:3064-3251).
func (m *gpuMirror) refresh(h *Index) error {
	m.mu.Lock()
	defer m.mu.Unlock()
	ep := h.writeEpoch.Load()
	if m.valid && m.epoch == ep {
		return nil
	}
	nodes := h.getNodes()
	count := uint32(h.nodeCounter.Load())
	if m.h == nil {
		desc := C.kdb_index_desc{
			dim: C.uint32_t(h.vectorDim), metric: hipMetric(h.metric), precision: hipPrecision(h.precision),
			m: C.uint32_t(h.m), ef_construction: C.uint32_t(h.efConstruction),
			capacity: C.uint32_t(cap(nodes)), device_id: 0,
		}
		if rc := C.kdb_index_create(&desc, &m.h); rc != 0 {
			return hipErr("kdb_index_create")
		}
		C.kdb_index_set_launch_timing(m.h, 0) // nobody reads last_kernel_ms here: two queue packets less per search
	}
	if err := m.uploadRows(h, nodes, count); err != nil {
		return err
	}
	if err := m.uploadGraph(h, nodes, count); err != nil {
		return err
	}
	m.epoch, m.valid = ep, true
	return nil
}

// uploadRows: Node.vec slices point into 64 MiB arena chunks (pkg/storage/mmap/arena.go:378-447), dense and in
// slot order inside a chunk, so consecutive ids are usually one contiguous run; one kdb_index_upload_rows call per
// run.  (When the data dir is local to the GPU host, kdb_index_upload_arena reads the chunk files directly.)
func (m *gpuMirror) uploadRows(h *Index, nodes []*Node, count uint32) error {
	rowBytes := uintptr(h.vectorDim) * map[distance.PrecisionType]uintptr{
		distance.Float32: 4, distance.Float16: 2, distance.Int8: 1}[h.precision]
	rowPtr := func(n *Node) unsafe.Pointer {
		switch h.precision {
		case distance.Float16:
			if v := n.GetVectorF16(); len(v) > 0 {
				return unsafe.Pointer(&v[0])
			}
		case distance.Int8:
			if v := n.GetVectorI8(); len(v) > 0 {
				return unsafe.Pointer(&v[0])
			}
		default:
			if v := n.GetVectorF32(); len(v) > 0 {
				return unsafe.Pointer(&v[0])
			}
		}
		return nil
	}
	var first uint32
	var base unsafe.Pointer
	var n uint32
	flush := func() error {
		if n == 0 {
			return nil
		}
		if rc := C.kdb_index_upload_rows(m.h, C.uint32_t(first), C.uint32_t(n), base); rc != 0 {
			return hipErr("kdb_index_upload_rows")
		}
		n = 0
		return nil
	}
	for id := uint32(1); id <= count && int(id) < len(nodes); id++ {
		node := nodes[id]
		if node == nil {
			if err := flush(); err != nil {
				return err
			}
			continue
		}
		p := rowPtr(node)
		if p == nil {
			if err := flush(); err != nil {
				return err
			}
			continue
		}
		if n > 0 && uintptr(p) == uintptr(base)+uintptr(n)*rowBytes {
			n++
			continue
		}
		if err := flush(); err != nil {
			return err
		}
		first, base, n = id, p, 1
	}
	if err := flush(); err != nil {
		return err
	}
	if h.precision == distance.Int8 {
		norms := h.getNorms()
		if len(norms) > 1 {
			nn := uint32(len(norms) - 1)
			if nn > count {
				nn = count
			}
			if rc := C.kdb_index_upload_norms(m.h, 1, C.uint32_t(nn), (*C.float)(unsafe.Pointer(&norms[1]))); rc != 0 {
				return hipErr("kdb_index_upload_norms")
			}
		}
		if h.quantizer != nil {
			C.kdb_index_set_quantizer(m.h, C.float(h.quantizer.AbsMax))
		}
	}
	return nil
}

// uploadGraph flattens Node.Connections into the per-level CSR of kdb_graph_view.  The pointer TABLES handed to C
// are C-allocated (cgo forbids Go pointers to Go pointers); the arrays they point at are Go slices pinned for the
// duration of the call by runtime.Pinner-free means: they are only read inside the call and the library copies.
func (m *gpuMirror) uploadGraph(h *Index, nodes []*Node, count uint32) error {
	maxLevel := int(h.maxLevel.Load())
	if maxLevel < 0 || count == 0 {
		return nil
	}
	nl := maxLevel + 1
	levels := make([]uint8, count+1)
	deleted := make([]uint64, (count>>6)+1)
	offsets := make([][]uint64, nl)
	neigh := make([][]uint32, nl)
	for l := 0; l < nl; l++ {
		offsets[l] = make([]uint64, count+2)
	}
	for id := uint32(1); id <= count && int(id) < len(nodes); id++ {
		node := nodes[id]
		if node == nil {
			for l := 0; l < nl; l++ {
				offsets[l][id+1] = offsets[l][id]
			}
			continue
		}
		shard := id % NumShards
		h.shardsMu[shard].RLock()
		conns := node.Connections
		lv := len(conns) - 1
		if lv < 0 {
			lv = 0
		}
		levels[id] = uint8(lv)
		for l := 0; l < nl; l++ {
			if l < len(conns) {
				neigh[l] = append(neigh[l], conns[l]...)
			}
			offsets[l][id+1] = uint64(len(neigh[l]))
		}
		h.shardsMu[shard].RUnlock()
		if node.Deleted.Load() {
			deleted[id>>6] |= 1 << (id & 63)
		}
	}
	// C-side pointer tables
	ptrSize := unsafe.Sizeof(uintptr(0))
	offTab := C.malloc(C.size_t(uintptr(nl) * ptrSize))
	nbTab := C.malloc(C.size_t(uintptr(nl) * ptrSize))
	defer C.free(offTab)
	defer C.free(nbTab)
	for l := 0; l < nl; l++ {
		if len(neigh[l]) == 0 {
			neigh[l] = []uint32{0}
		}
		*(*unsafe.Pointer)(unsafe.Add(offTab, uintptr(l)*ptrSize)) = unsafe.Pointer(&offsets[l][0])
		*(*unsafe.Pointer)(unsafe.Add(nbTab, uintptr(l)*ptrSize)) = unsafe.Pointer(&neigh[l][0])
	}
	g := C.kdb_graph_view{
		count: C.uint32_t(count), entry: C.uint32_t(h.entrypointID.Load()), max_level: C.int32_t(maxLevel),
		levels: (*C.uint8_t)(unsafe.Pointer(&levels[0])), offsets: (**C.uint64_t)(offTab), neighbors: (**C.uint32_t)(nbTab),
		deleted_bits: (*C.uint64_t)(unsafe.Pointer(&deleted[0])),
	}
	if rc := C.kdb_index_upload_graph(m.h, &g); rc != 0 {
		return hipErr("kdb_index_upload_graph")
	}
	return nil
}

func (m *gpuMirror) destroy() {
	m.mu.Lock()
	defer m.mu.Unlock()
	if m.batcher != nil {
		m.batcher.stop()
		m.batcher = nil
	}
	if m.h != nil {
		C.kdb_index_destroy(m.h)
		m.h = nil
	}
	m.valid = false
}

// ---------------------------------------------------------------------------------------------------------
// micro-batcher
// ---------------------------------------------------------------------------------------------------------

// Round 5: the LIBRARY serves concurrent callers itself -- a cgo call holds the handle's lock only to enqueue, one-query calls that
// find its slots busy are combined into one launch and every caller returns when ITS walk is done (kektor_hip.h, "Conventions").
// Unfiltered queries therefore go straight to C.kdb_search_batch with B = 1, one cgo call per goroutine (searchHIP below); the
// batcher only groups what the library does not combine: queries that share an allow list (one conversion and one upload of the
// list per group, and the routing of selective filters to the exact scan).  Its compiled twin is kektor::hnsw::MicroBatcher.
const (
	hipBatchWindow = 0 * time.Microsecond // a group's first caller flushes at once when a turn on the device is free
	hipBatchMax    = 8192                 // queries per GPU call
	hipMaxInFlight = 2                    // GPU calls of one batcher on the device at once
	// below this fraction of allowed ids the reference's filtered graph walk is disconnected (it prunes
	// non-allowed neighbours during traversal, hnsw_index.go:2545-2549): such queries take the exact scan
	hipFlatScanSelectivity = 0.1 // measured: below it the filtered walk returns 1-2 answers instead of k (INTEGRATION.md, "Filter routing")
)

type hipRequest struct {
	query []float32
	done  chan []types.SearchResult
}

type hipGroupKey struct {
	k, ef int
	allow *roaring.Bitmap // callers that share the SAME bitmap object share a batch (ops.go builds one per request)
}

type hipGroup struct {
	key   hipGroupKey
	reqs  []*hipRequest
	timer *time.Timer
}

type hipBatcher struct {
	h      *Index
	turns  chan struct{} // hipMaxInFlight tokens: two groups on the device at once, the next one stays open (and grows) meanwhile
	mu     sync.Mutex
	groups map[hipGroupKey]*hipGroup
	closed bool
}

func newHipBatcher(h *Index) *hipBatcher {
	b := &hipBatcher{h: h, groups: make(map[hipGroupKey]*hipGroup), turns: make(chan struct{}, hipMaxInFlight)}
	for i := 0; i < hipMaxInFlight; i++ {
		b.turns <- struct{}{}
	}
	return b
}

func (b *hipBatcher) stop() {
	b.mu.Lock()
	b.closed = true
	gs := b.groups
	b.groups = map[hipGroupKey]*hipGroup{}
	b.mu.Unlock()
	for _, g := range gs {
		for _, r := range g.reqs {
			r.done <- []types.SearchResult{}
		}
	}
}

// submit enqueues one query and blocks until its batch has run.
func (b *hipBatcher) submit(query []float32, k, ef int, allow *roaring.Bitmap) []types.SearchResult {
	req := &hipRequest{query: query, done: make(chan []types.SearchResult, 1)}
	key := hipGroupKey{k: k, ef: ef, allow: allow}
	b.mu.Lock()
	if b.closed {
		b.mu.Unlock()
		return []types.SearchResult{}
	}
	g := b.groups[key]
	if g == nil {
		g = &hipGroup{key: key}
		b.groups[key] = g
		g.timer = time.AfterFunc(hipBatchWindow, func() { b.flush(key, g) })
	}
	g.reqs = append(g.reqs, req)
	full := len(g.reqs) >= hipBatchMax
	b.mu.Unlock()
	if full {
		b.flush(key, g)
	}
	return <-req.done
}

func (b *hipBatcher) flush(key hipGroupKey, g *hipGroup) {
	// the group stays open (joinable) while the calls in flight run: batch size follows the load
	<-b.turns
	defer func() { b.turns <- struct{}{} }()
	b.mu.Lock()
	if b.groups[key] != g { // already flushed by the other trigger
		b.mu.Unlock()
		return
	}
	delete(b.groups, key)
	g.timer.Stop()
	reqs := g.reqs
	b.mu.Unlock()
	out, err := b.h.searchBatchHIP(reqs, key.k, key.ef, key.allow)
	if err != nil {
		slog.Error("Error during HNSW search (hip)", "error", err) // same swallow-to-empty as hnsw_index.go:356-359
	}
	for i, r := range reqs {
		if err != nil || i >= len(out) {
			r.done <- []types.SearchResult{}
		} else {
			r.done <- out[i]
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// the calls
// ---------------------------------------------------------------------------------------------------------

// searchHIP is what SearchWithScores does under the tag, after its isClosed / activeMu.RLock prologue
// (hnsw_index.go:343-351); the caller still holds activeMu.RLock while it waits for its batch.
func (h *Index) searchHIP(query []float32, k int, allowList *roaring.Bitmap, efSearch int) []types.SearchResult {
	if len(query) != h.vectorDim {
		slog.Error("Error during HNSW search", "error", errors.New("query dimension mismatch"))
		return []types.SearchResult{}
	}
	for _, x := range query { // the kernels' contract (kektor_hip.h, DESIGN 5.1): a NaN or an infinity in a query can fault the GPU
		if math.Float32bits(x)&0x7f800000 == 0x7f800000 {
			slog.Error("Error during HNSW search", "error", errors.New("query holds a value that is not finite"))
			return []types.SearchResult{}
		}
	}
	if allowList == nil { // unfiltered: one cgo call per goroutine; the library combines concurrent calls (measured on one MI355X,
		// 1M x 768, ef 60: 64 goroutine-like callers 257-267 k QPS at p50 0.22 ms, 172-193 k with the heap-order flag this file sets -- bench.py micro_batcher)
		out, err := h.searchBatchHIP([]*hipRequest{{query: query}}, k, efSearch, nil)
		if err != nil || len(out) == 0 {
			if err != nil {
				slog.Error("Error during HNSW search (hip)", "error", err)
			}
			return []types.SearchResult{}
		}
		return out[0]
	}
	h.gpu.mu.Lock()
	if h.gpu.batcher == nil {
		h.gpu.batcher = newHipBatcher(h)
	}
	b := h.gpu.batcher
	h.gpu.mu.Unlock()
	return b.submit(query, k, efSearch, allowList)
}

func (h *Index) searchBatchHIP(reqs []*hipRequest, k, ef int, allow *roaring.Bitmap) ([][]types.SearchResult, error) {
	if err := h.gpu.refresh(h); err != nil {
		return nil, err
	}
	B := len(reqs)
	dim := h.vectorDim
	queries := make([]float32, B*dim)
	for i, r := range reqs {
		copy(queries[i*dim:(i+1)*dim], r.query)
	}
	count := h.nodeCounter.Load()
	var allowPtr *C.uint64_t
	var dense []uint64
	useFlat := false
	if allow != nil { // roaring -> dense words over internal ids; an all-zero bitmap is the non-nil EMPTY list
		dense = make([]uint64, (count>>6)+1)
		it := allow.Iterator()
		for it.HasNext() {
			id := uint64(it.Next())
			if int(id>>6) < len(dense) {
				dense[id>>6] |= 1 << (id & 63)
			}
		}
		allowPtr = (*C.uint64_t)(unsafe.Pointer(&dense[0]))
		card := allow.GetCardinality()
		// the exact scan answers k <= 1024 (kdb_flat_scan_batch); larger k keeps the walk, as the C++ twin does
		useFlat = card > 0 && k <= 1024 && float64(card) < hipFlatScanSelectivity*float64(count)
	}
	ids := make([]uint32, B*k)
	// int8 indexes: the distances are float64 in the reference (hnsw_index.go:2429-2454); KDB_SEARCH_DIST_F64 makes the
	// library hand over those doubles (out_dist then points to B*k float64) instead of their float32 rounding
	wide := h.precision == distance.Int8
	dist := make([]float32, B*k)
	var dist64 []float64
	distPtr := (*C.float)(unsafe.Pointer(&dist[0]))
	var flags C.uint32_t
	if wide {
		dist64 = make([]float64, B*k)
		distPtr = (*C.float)(unsafe.Pointer(&dist64[0]))
		flags |= C.KDB_SEARCH_DIST_F64
	}
	cnt := make([]uint32, B)
	var rc C.int
	if useFlat { // the filtered path of north_star: exact scan over the allowed rows
		rc = C.kdb_flat_scan_batch(h.gpu.h, (*C.float)(unsafe.Pointer(&queries[0])), C.uint32_t(B), C.uint32_t(k), allowPtr, flags,
			(*C.uint32_t)(unsafe.Pointer(&ids[0])), distPtr, (*C.uint32_t)(unsafe.Pointer(&cnt[0])))
	} else {
		if h.needsRefine.Load() {
			flags |= C.KDB_SEARCH_NEEDS_REFINE // the ef boost of hnsw_index.go:387-399 is applied by the library
		}
		// duplicate vectors: equal distances pop / evict / sort in the order of the two heaps of hnsw_heap.go; the library walks
		// exactly those queries again with the same two heaps, so ids and their order are this package's own, ties included
		flags |= C.KDB_SEARCH_HEAP_ORDER
		rc = C.kdb_search_batch(h.gpu.h, (*C.float)(unsafe.Pointer(&queries[0])), C.uint32_t(B), C.uint32_t(k), C.uint32_t(ef),
			allowPtr, flags, (*C.uint32_t)(unsafe.Pointer(&ids[0])), distPtr,
			(*C.uint32_t)(unsafe.Pointer(&cnt[0])))
	}
	if rc != 0 {
		return nil, hipErr("kdb_search_batch")
	}
	out := make([][]types.SearchResult, B)
	for b := 0; b < B; b++ {
		n := int(cnt[b])
		res := make([]types.SearchResult, n)
		for i := 0; i < n; i++ {
			raw := float64(dist[b*k+i]) // float64(sum), distance_go.go:67
			if wide {
				raw = dist64[b*k+i]
			}
			score := raw
			if h.metric == distance.Cosine && h.precision == distance.Float32 {
				score = 1.0 - raw // the library returns the raw dot product; distance_go.go:127
			}
			res[i] = types.SearchResult{DocID: ids[b*k+i], Score: score}
		}
		out[b] = res
	}
	return out, nil
}
