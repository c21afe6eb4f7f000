// Package suggesthip plugs libsuggest_hip.so (MI355X engine) into suggest-go behind the reference's own
// extension seam: it implements suggest.Builder (pkg/suggest/ngram_index_builder.go:14-17) and
// suggest.NGramIndex = Suggester + Autocomplete (pkg/suggest/ngram_index.go:7-10), so
// Service.AddIndex / Suggest / Autocomplete callers (pkg/suggest/service.go:78-173) are unchanged.
//
// NOT compiled in this repository (no Go toolchain in the build image); shipped as the binding a
// maintainer adds next to pkg/suggest.  Build: CGO_ENABLED=1, -I<repo>/include, -L<repo>/suggest_amd.
// Written for the Go the reference builds with (go.mod: 1.13; golang:1.14 image): no unsafe.Slice, no generics.
package suggesthip

/*
#cgo LDFLAGS: -lsuggest_hip
#include <stdlib.h>
#include "suggest_hip.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math"
	"reflect"
	"runtime"
	"sync"
	"sync/atomic"
	"unsafe"

	"github.com/suggest-go/suggest/pkg/dictionary"
	"github.com/suggest-go/suggest/pkg/merger"
	"github.com/suggest-go/suggest/pkg/metric"
	"github.com/suggest-go/suggest/pkg/suggest"
)

// Builder builds a GPU index from a dictionary and an IndexDescription (suggest.Builder).
type Builder struct {
	Dict        dictionary.Dictionary
	Description suggest.IndexDescription
	Device      int
	// Devices, when set, gets one replica of the index per listed GPU (sg_index_replicate: ONE host build) and batches
	// are cut into contiguous slices over them (sg_suggest_batch_multi) — SURVEY.md 8e, no collective.
	Devices []int
	// OnDevice builds the index on the GPU (sg_index_build_device: same arrays, ~16x faster at 10 M strings); a dictionary
	// beyond its limits (a document above 65 536 bytes, 2^26 documents) falls back to the host builder.
	OnDevice bool
}

// Build tokenises the dictionary on the host, lays out the CSR and uploads it to HBM.
func (b *Builder) Build() (suggest.NGramIndex, error) {
	var blob []byte
	offs := []C.uint64_t{0}
	err := b.Dict.Iterate(func(_ dictionary.Key, v dictionary.Value) error {
		blob = append(blob, v...)
		offs = append(offs, C.uint64_t(len(blob)))
		return nil
	})
	if err != nil {
		return nil, err
	}
	d := b.Description
	arr, freeAlpha := cstrings(d.Alphabet)
	defer freeAlpha()
	w0, w1, pad := C.CString(d.Wrap[0]), C.CString(d.Wrap[1]), C.CString(d.Pad)
	defer C.free(unsafe.Pointer(w0))
	defer C.free(unsafe.Pointer(w1))
	defer C.free(unsafe.Pointer(pad))
	// the descriptor lives in C memory for the duration of the call (cgo pointer rules)
	desc := (*C.sg_desc)(C.malloc(C.size_t(unsafe.Sizeof(C.sg_desc{}))))
	defer C.free(unsafe.Pointer(desc))
	desc.ngram_size, desc.wrap_start, desc.wrap_end, desc.pad = C.uint32_t(d.NGramSize), w0, w1, pad
	desc.alphabet, desc.n_alphabet = arr, C.uint32_t(len(d.Alphabet))

	var h *C.sg_index
	var bp *C.uint8_t
	if len(blob) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&blob[0]))
	}
	built := false
	if b.OnDevice { // a dictionary beyond the device builder's limits: SG_E_UNSUPPORTED, then the host's
		runtime.LockOSThread()
		rc := C.sg_index_build_device(bp, &offs[0], C.uint32_t(len(offs)-1), desc, C.int(b.Device), &h)
		if rc != 0 && rc != C.SG_E_UNSUPPORTED {
			err := fmt.Errorf("suggest_hip %d: %s", int(rc), C.GoString(C.sg_last_error()))
			runtime.UnlockOSThread()
			return nil, err
		}
		runtime.UnlockOSThread()
		built = rc == 0
	}
	if !built {
		if err := ccall(func() C.int { return C.sg_index_build(bp, &offs[0], C.uint32_t(len(offs)-1), desc, &h) }); err != nil {
			return nil, err
		}
	}
	if err := ccall(func() C.int { return C.sg_index_upload(h, C.int(b.Device)) }); err != nil {
		C.sg_index_release(h)
		return nil, err
	}
	if len(b.Devices) > 0 {
		devs := make([]C.int, len(b.Devices))
		for i, d := range b.Devices {
			devs[i] = C.int(d)
		}
		if err := ccall(func() C.int { return C.sg_index_replicate(h, &devs[0], C.uint32_t(len(devs))) }); err != nil {
			C.sg_index_release(h)
			return nil, err
		}
	}
	e := &engine{h: h, reqs: make(chan *request, 4096), closing: make(chan struct{}), done: make(chan struct{}), multi: len(b.Devices) > 1,
		tables: map[tablesKey]*tableEntry{}}
	var st C.sg_stats
	C.sg_index_stats(h, &st)
	e.segments = int(st.n_segments)
	go e.dispatch() // holds the engine only: the Index stays collectable, its finalizer (or Close) stops the dispatcher
	ix := &Index{e: e}
	runtime.SetFinalizer(ix, func(i *Index) { i.Close() }) // cf. pkg/index/index_reader.go:49-51
	return ix, nil
}

// Index is the GPU-resident NGramIndex.  Single-query calls (the reference's calling pattern: one Suggest per goroutine,
// pkg/suggest/service_test.go:36-79) are coalesced HERE, in Go: a goroutine parks on a channel for a fraction of what a
// blocked cgo call costs, so the C-side queue (sg_suggest_one) is left to C / C++ / Python callers.
//
// Life cycle: Service.AddIndex swaps indexes (service.go:78-91, the SIGHUP re-index path) and drops the old one.  The
// dispatcher goroutine references only the inner engine, never the Index, so a dropped Index is finalized: Close stops
// the dispatcher, which releases the handle — and with it the HBM replicas — once the batches in flight are answered.
// Callers that know when an index goes should call Close themselves instead of waiting for the collector.
type Index struct {
	e    *engine
	once sync.Once
}

type engine struct {
	h        *C.sg_index
	reqs     chan *request
	closing  chan struct{} // closed by Close: senders and the dispatcher select on it (no lock is held across a channel send)
	done     chan struct{}
	mu       sync.RWMutex // `closed` and the retain of a call are read under it: Close cannot slip in between (never held across a channel operation)
	closed   bool
	multi    bool // several replicas: batches go through sg_suggest_batch_multi (synchronous); one: sg_suggest_submit, two tickets in flight
	segments int
	lastK    int32 // atomic: the k the last fuzzy factory turned out to have (probed first next time)
	tmu      sync.Mutex
	tables   map[tablesKey]*tableEntry // at most maxTableSets, least recently used out first; nil once the index is closed
	tclock   uint64
}

// A cached table set and when it was last asked for.  The cache holds ONE reference of the set (sg_metric_tables_retain /
// _release count them); every call that uses it holds its own, so a set evicted or dropped by Close while a call is in flight
// stays in HBM until that call returns.
type tableEntry struct {
	t    *C.sg_metric_tables
	used uint64
}

const (
	maxTableSets  = 8   // table sets kept per index: (metric value, similarity, aMax) triples a service really alternates between
	maxTableTerms = 256 // longest query (n-grams) a table set is built for: (aMax + 1)^2 x segments doubles — 34 MB at 64 segments;
	// a longer query comes back SG_COUNT_TOO_LONG from the engine instead of costing gigabytes of tables
)

var errClosed = errors.New("suggesthip: index is closed")

// Close stops the dispatcher and gives the creator's reference back; queries in flight finish first (they hold their own).
func (i *Index) Close() error {
	i.once.Do(func() {
		e := i.e
		e.mu.Lock()
		e.closed = true
		e.mu.Unlock()
		close(e.closing)
		runtime.SetFinalizer(i, nil)
	})
	return nil
}

// retain takes a reference for the duration of a C call (service.go:85-88 swaps indexes while queries run).  The dispatcher
// gives the creator's reference back only after `closing` is closed, which Close does after setting `closed` under the
// lock: a retain that saw closed == false took its reference from a live handle.
func (e *engine) retain() error {
	e.mu.RLock()
	defer e.mu.RUnlock()
	if e.closed {
		return errClosed
	}
	C.sg_index_retain(e.h)
	return nil
}

func (e *engine) release() { C.sg_index_release(e.h) }

type request struct {
	query      string
	similarity float64
	code       C.int                // the engine's enum, or -1: tabulated
	tables     *C.sg_metric_tables  // an opaque metric.Metric as tables (code == -1)
	k          int
	resp       chan response
}

type response struct {
	cands  []suggest.Candidate
	status uint32 // SG_COUNT_* flag of THIS query (0: answered)
	err    error
}

// pinnedBuf is a grow-only block of pinned host memory (sg_host_alloc): the DMA engine reads the queries from it and writes
// the rows into it directly, and it is C memory — no Go pointer crosses the boundary while a ticket is in flight.
type pinnedBuf struct {
	p   unsafe.Pointer
	cap int
}

func (b *pinnedBuf) need(n int) error {
	if n <= b.cap {
		return nil
	}
	if b.p != nil {
		C.sg_host_free(b.p)
		b.p, b.cap = nil, 0
	}
	c := 1 << 16
	for c < n {
		c <<= 1
	}
	var p unsafe.Pointer
	if err := ccall(func() C.int { return C.sg_host_alloc(C.uint64_t(c), &p) }); err != nil {
		return err
	}
	b.p, b.cap = p, c
	return nil
}

// flight is one batch between sg_suggest_submit and sg_ticket_wait.
type flight struct {
	reqs   []*request
	k      int
	ticket *C.sg_ticket
	in     pinnedBuf // [offsets | query bytes]
	out    pinnedBuf // [scores | ids | counts]
	err    error
}

// dispatch runs whatever single-query requests are pending as one batch per distinct (metric, similarity, k): no timer —
// an idle engine serves a lone request at once, a busy one finds the requests that arrived meanwhile.  With one replica the
// batches go through sg_suggest_submit with TWO tickets in flight: while the kernel of one batch runs, the next batch is
// gathered, staged in pinned memory and copied in, and the previous one's rows come back — from one goroutine the engine
// sees the rate of its device-resident entry point (bench.py `host_buffers_pipelined`), not that minus the PCIe copies.
// Every request gets ITS OWN outcome: a query the reference would panic or dead-lock on, or one past the engine's
// limits, fails its caller only — in the caller's goroutine (see Suggest) — and never the others that shared its launch.
func (e *engine) dispatch() {
	var inflight []*flight
	free := []*flight{{}, {}, {}}
	finish := func() {
		f := inflight[0]
		inflight = inflight[1:]
		e.finishFlight(f)
		free = append(free, f)
	}
	defer func() {
		for len(inflight) > 0 {
			finish()
		}
		for _, f := range free {
			if f.in.p != nil {
				C.sg_host_free(f.in.p)
			}
			if f.out.p != nil {
				C.sg_host_free(f.out.p)
			}
		}
		e.tmu.Lock()
		for _, te := range e.tables { // the cache's references; calls in flight hold their own
			C.sg_metric_tables_release(te.t)
		}
		e.tables = nil
		e.tmu.Unlock()
		C.sg_index_release(e.h) // the creator's reference: the handle goes when the last call in flight returns
		close(e.done)
	}()
	for {
		var first *request
		if len(inflight) > 0 { // something is on the GPU: take what has arrived, else hand the oldest batch back
			select {
			case first = <-e.reqs:
			case <-e.closing:
				return
			default:
				finish()
				continue
			}
		} else {
			select {
			case first = <-e.reqs:
			case <-e.closing:
				return
			}
		}
		batch := []*request{first}
	drain:
		for len(batch) < 65536 {
			select {
			case r := <-e.reqs:
				batch = append(batch, r)
			default:
				break drain
			}
		}
		for len(batch) > 0 {
			head := batch[0]
			var same, rest []*request
			for _, r := range batch {
				if r.code == head.code && r.tables == head.tables && r.similarity == head.similarity && r.k == head.k {
					same = append(same, r)
				} else {
					rest = append(rest, r)
				}
			}
			batch = rest
			if e.multi || head.tables != nil { // several replicas / a tabulated metric: the synchronous entry points
				qs := make([]string, len(same))
				for j, r := range same {
					qs[j] = r.query
				}
				res, status, err := e.suggestBatch(qs, head.similarity, head.code, head.tables, head.k)
				for j, r := range same {
					if err != nil {
						r.resp <- response{nil, 0, err}
					} else {
						r.resp <- response{res[j], status[j], nil}
					}
				}
				continue
			}
			f := free[len(free)-1]
			free = free[:len(free)-1]
			f.reqs, f.k = same, head.k
			e.submitFlight(f, head)
			inflight = append(inflight, f)
			if len(inflight) > 2 {
				finish()
			}
		}
	}
}

// submitFlight stages one batch in the flight's pinned memory and enqueues copy in -> launch -> copy out (sg_suggest_submit).
func (e *engine) submitFlight(f *flight, head *request) {
	n, k := len(f.reqs), f.k
	bytes := 0
	for _, r := range f.reqs {
		bytes += len(r.query)
	}
	offBytes := (n + 1) * 8
	f.err = f.in.need(offBytes + bytes + 16)
	if f.err == nil {
		f.err = f.out.need(n*k*12 + n*4 + 16)
	}
	if f.err != nil {
		return
	}
	offs := (*[1 << 28]C.uint64_t)(f.in.p)[: n+1 : n+1]
	blob := (*[1 << 30]byte)(unsafe.Pointer(uintptr(f.in.p) + uintptr(offBytes)))[:bytes:bytes]
	at := 0
	for j, r := range f.reqs {
		offs[j] = C.uint64_t(at)
		at += copy(blob[at:], r.query)
	}
	offs[n] = C.uint64_t(at)
	scores := (*C.double)(f.out.p)
	ids := (*C.uint32_t)(unsafe.Pointer(uintptr(f.out.p) + uintptr(n*k*8)))
	counts := (*C.uint32_t)(unsafe.Pointer(uintptr(f.out.p) + uintptr(n*k*12)))
	if f.err = e.retain(); f.err != nil {
		return
	}
	f.err = ccall(func() C.int {
		return C.sg_suggest_submit(e.h, (*C.uint8_t)(unsafe.Pointer(uintptr(f.in.p)+uintptr(offBytes))), (*C.uint64_t)(f.in.p), C.uint32_t(n), head.code,
			C.double(head.similarity), C.uint32_t(k), ids, scores, counts, &f.ticket)
	})
	if f.err != nil {
		e.release()
	}
}

// finishFlight waits for the batch's rows and answers its callers.
func (e *engine) finishFlight(f *flight) {
	err := f.err
	if err == nil {
		err = ccall(func() C.int { return C.sg_ticket_wait(f.ticket) })
		e.release()
	}
	n, k := len(f.reqs), f.k
	if err != nil {
		for _, r := range f.reqs {
			r.resp <- response{nil, 0, err}
		}
	} else {
		scores := (*[1 << 27]C.double)(f.out.p)[: n*k : n*k]
		ids := (*[1 << 28]C.uint32_t)(unsafe.Pointer(uintptr(f.out.p) + uintptr(n*k*8)))[: n*k : n*k]
		counts := (*[1 << 28]C.uint32_t)(unsafe.Pointer(uintptr(f.out.p) + uintptr(n*k*12)))[:n:n]
		for q, r := range f.reqs {
			c := uint32(counts[q])
			if c >= C.SG_COUNT_LM_ERROR { // an SG_COUNT_* flag: no row
				r.resp <- response{nil, c, nil}
				continue
			}
			out := make([]suggest.Candidate, c)
			for j := uint32(0); j < c; j++ {
				out[j] = suggest.Candidate{Key: uint32(ids[q*k+int(j)]), Score: float64(scores[q*k+int(j)])}
			}
			r.resp <- response{out, 0, nil}
		}
	}
	f.reqs, f.ticket, f.err = nil, nil, nil
}

var respPool = sync.Pool{New: func() interface{} { return make(chan response, 1) }}

// ccall runs one C-ABI call and, on failure, reads its message: sg_last_error is thread-local, and a goroutine may move to
// another OS thread between two cgo calls — so both happen with the goroutine pinned.
func ccall(f func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := f(); rc != 0 {
		return fmt.Errorf("suggest_hip %d: %s", int(rc), C.GoString(C.sg_last_error()))
	}
	return nil
}

// cstrings lays a []string out as a C array of C strings (no unsafe.Slice: the reference builds with Go 1.13/1.14).
func cstrings(xs []string) (**C.char, func()) {
	arr := (**C.char)(C.malloc(C.size_t(len(xs)+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	sl := (*[1 << 28]*C.char)(unsafe.Pointer(arr))[: len(xs) : len(xs)]
	for i, x := range xs {
		sl[i] = C.CString(x)
	}
	return arr, func() {
		for _, p := range sl {
			C.free(unsafe.Pointer(p))
		}
		C.free(unsafe.Pointer(arr))
	}
}

// statusError is what a query's SG_COUNT_* flag means to its caller.
func statusError(status uint32, q int) error {
	switch status {
	case C.SG_COUNT_REF_DEADLOCK:
		return fmt.Errorf("suggesthip: query %d: the reference dead-locks here (suggester.go:62, zero channel capacity)", q)
	case C.SG_COUNT_TOO_LONG:
		return fmt.Errorf("suggesthip: query %d has more than %d n-grams (or more than its metric was tabulated for)", q, int(C.SG_MAX_QUERY_TERMS))
	}
	return nil
}

// metricCode recovers the metric behaviourally: the Metric interface is opaque at this seam (unexported types), but its
// four methods identify the five implementations of pkg/metric on two probe points.  ok == false: somebody else's
// implementation — it reaches the engine as tables of its four methods (tablesFor).
func metricCode(m metric.Metric) (C.int, bool) {
	probes := []struct {
		code C.int
		m    metric.Metric
	}{{C.SG_JACCARD, metric.JaccardMetric()}, {C.SG_COSINE, metric.CosineMetric()}, {C.SG_DICE, metric.DiceMetric()},
		{C.SG_EXACT, metric.ExactMetric()}, {C.SG_OVERLAP, metric.OverlapMetric()}}
	for _, p := range probes {
		if m.Threshold(0.37, 11, 17) == p.m.Threshold(0.37, 11, 17) && m.MinY(0.37, 11) == p.m.MinY(0.37, 11) &&
			m.MaxY(0.37, 11) == p.m.MaxY(0.37, 11) && m.Distance(5, 11, 17) == p.m.Distance(5, 11, 17) &&
			m.Threshold(0.81, 23, 19) == p.m.Threshold(0.81, 23, 19) && m.Distance(9, 23, 19) == p.m.Distance(9, 23, 19) {
			return p.code, true
		}
	}
	return -1, false
}

type tablesKey struct {
	m    metric.Metric
	sim  float64
	aMax int
}

// tablesFor tabulates an opaque metric.Metric (pkg/metric/metric.go:7-16) at one similarity for queries of up to aMax n-grams
// (rounded up to a power of two, so that a handful of table sets serve every query length; capped at maxTableTerms): MinY, MaxY,
// Threshold and the score 1 - Distance exactly as metricScorer.Score computes it (pkg/suggest/scorer.go:29-31) —
// sg_metric_tables_create.  The caller OWNS one reference of what it gets and gives it back with releaseTables when its call
// has returned.  Sets of metrics whose dynamic type is comparable are cached per (metric value, similarity, aMax), at most
// maxTableSets of them (least recently used out first: a metric built per request, or a similarity that varies per request,
// costs a rebuild, not HBM); the others live for their one call.
func (e *engine) tablesFor(m metric.Metric, sim float64, terms int) (*C.sg_metric_tables, error) {
	aMax := 16
	for aMax < terms && aMax < maxTableTerms {
		aMax <<= 1
	}
	key := tablesKey{m, sim, aMax}
	cacheable := reflect.TypeOf(m).Comparable()
	if cacheable {
		e.tmu.Lock()
		if te := e.tables[key]; te != nil {
			e.tclock++
			te.used = e.tclock
			C.sg_metric_tables_retain(te.t) // the caller's reference, taken under the lock: eviction cannot slip in between
			e.tmu.Unlock()
			return te.t, nil
		}
		e.tmu.Unlock()
	}
	S, nA := e.segments, aMax+1
	minY, maxY := make([]C.int32_t, nA), make([]C.int32_t, nA)
	thr := make([]C.int32_t, nA*S)
	score := make([]C.double, nA*S*nA)
	clamp := func(v int) C.int32_t {
		if v > math.MaxInt32 {
			return math.MaxInt32
		}
		if v < math.MinInt32 {
			return math.MinInt32
		}
		return C.int32_t(v)
	}
	for a := 1; a < nA; a++ {
		lo, hi := m.MinY(sim, a), m.MaxY(sim, a)
		minY[a], maxY[a] = clamp(lo), clamp(hi)
		if lo < 0 {
			lo = 0
		}
		if hi > S-1 {
			hi = S - 1
		}
		for b := lo; b <= hi; b++ {
			t := m.Threshold(sim, a, b)
			thr[a*S+b] = clamp(t)
			if t < 0 {
				t = 0
			}
			for o := t; o <= a; o++ { // (a query that repeats a term: overlaps up to a)
				score[(a*S+b)*nA+o] = C.double(1 - m.Distance(o, a, b))
			}
		}
	}
	var t *C.sg_metric_tables
	if err := e.retain(); err != nil {
		return nil, err
	}
	err := ccall(func() C.int {
		return C.sg_metric_tables_create(e.h, C.uint32_t(aMax), &minY[0], &maxY[0], &thr[0], &score[0], &t)
	})
	e.release()
	if err != nil {
		return nil, err
	}
	if !cacheable {
		return t, nil // the caller's only reference: gone with releaseTables
	}
	e.tmu.Lock()
	defer e.tmu.Unlock()
	if e.tables == nil { // closed meanwhile: nothing is cached any more
		return t, nil
	}
	if old := e.tables[key]; old != nil { // somebody else was faster: theirs is the cached one
		C.sg_metric_tables_release(t)
		e.tclock++
		old.used = e.tclock
		C.sg_metric_tables_retain(old.t)
		return old.t, nil
	}
	if len(e.tables) >= maxTableSets { // the least recently used set gives up the cache's reference
		var oldest tablesKey
		first := true
		for k, te := range e.tables {
			if first || te.used < e.tables[oldest].used {
				oldest, first = k, false
			}
		}
		C.sg_metric_tables_release(e.tables[oldest].t)
		delete(e.tables, oldest)
	}
	e.tclock++
	e.tables[key] = &tableEntry{t: t, used: e.tclock}
	C.sg_metric_tables_retain(t) // the cache's reference beside the caller's
	return t, nil
}

// releaseTables gives a call's reference of a table set back (nil: the metric had a device twin, there were no tables).
func releaseTables(t *C.sg_metric_tables) {
	if t != nil {
		C.sg_metric_tables_release(t)
	}
}

// fuzzyK recovers k from a fuzzy collector-manager factory (the closure hides it, pkg/suggest/collector.go:143-149): a
// manager fed n distinct candidates returns min(n, k) of them.  Service.Suggest makes a fresh factory per call
// (service.go:116-121), so nothing can be remembered per factory; what is remembered is the LAST k: one manager fed k + 1
// candidates that keeps exactly k confirms it — the steady state of a service whose callers ask for the same topK.
// Otherwise doubling, then bisection, up to SG_MAX_TOPK.
func (e *engine) fuzzyK(factory suggest.CollectorManagerFactory) int {
	kept := func(n int) int { // how many of n candidates does a fresh queue keep?
		mgr := factory()
		c := mgr.Create()
		c.SetScorer(constScorer(1))
		for d := 1; d <= n; d++ {
			if c.Collect(merger.NewMergeCandidate(uint32(d), 1)) != nil {
				break
			}
		}
		_ = mgr.Collect(c)
		return len(mgr.GetCandidates())
	}
	if last := int(atomic.LoadInt32(&e.lastK)); last > 0 && kept(last+1) == last {
		return last
	}
	max := int(C.SG_MAX_TOPK)
	hi := 1
	for hi < max && kept(hi*2) == hi*2 {
		hi *= 2
	}
	lo, top := hi, hi*2 // keeps lo, not top (or top is beyond the engine's limit)
	if top > max {
		top = max + 1
	}
	for top-lo > 1 {
		mid := (lo + top) / 2
		if kept(mid) == mid {
			lo = mid
		} else {
			top = mid
		}
	}
	atomic.StoreInt32(&e.lastK, int32(lo))
	return lo
}

type constScorer float64

func (s constScorer) Score(merger.MergeCandidate) float64 { return float64(s) }

// Suggest implements suggest.Suggester (pkg/suggest/suggester.go:17-20) for ANY metric.Metric and ANY collector manager.
//
// The usual case — one of pkg/metric's five metrics and the fuzzy top-k manager (what Service.Suggest passes,
// service.go:116-121) — joins whatever other goroutines are asking at the moment and is answered from their common launch.
// Another Metric implementation is tabulated (tablesFor) and takes the same road.  Another CollectorManager gets what the
// reference gives it (suggester.go:78-99): every document whose overlap reaches its segment's threshold, per segment in
// ascending docID, through a collector of its own with that segment's scorer — replayed from sg_suggest_batch_from's pages.
//
// What the reference does on THIS query it does to THIS caller: the panic of suggester.go:62 (negative channel capacity) is
// raised here, in the caller's goroutine, where it can be recovered like the reference's; the dead-lock case and a query
// past the engine's limits come back as an error.
func (i *Index) Suggest(query string, similarity float64, m metric.Metric, factory suggest.CollectorManagerFactory) ([]suggest.Candidate, error) {
	e := i.e
	defer runtime.KeepAlive(i)
	code, known := metricCode(m)
	var tables *C.sg_metric_tables
	if !known {
		t, err := e.tablesFor(m, similarity, len(query)+16) // (n-grams <= runes + wrap runes <= bytes + 16)
		if err != nil {
			return nil, err
		}
		tables = t
		defer releaseTables(t) // (after the answer is here: the request below is answered before Suggest returns)
	}
	mgr := factory()
	if _, fuzzy := mgr.(*suggest.FuzzyCollectorManager); !fuzzy {
		return e.suggestAny(query, similarity, m, code, tables, mgr)
	}
	r := &request{query: query, similarity: similarity, code: code, tables: tables, k: e.fuzzyK(factory), resp: respPool.Get().(chan response)}
	select {
	case e.reqs <- r:
	case <-e.closing:
		return nil, errClosed
	}
	var out response
	select {
	case out = <-r.resp:
	case <-e.done: // the dispatcher went away with the request still queued
		select {
		case out = <-r.resp:
		default:
			return nil, errClosed
		}
	}
	respPool.Put(r.resp)
	if out.err != nil {
		return nil, out.err
	}
	if out.status == C.SG_COUNT_REF_PANIC {
		panic("makechan: size out of range") // what pkg/suggest/suggester.go:62 does on this query
	}
	if err := statusError(out.status, 0); err != nil {
		return nil, err
	}
	return out.cands, nil
}

// suggestAny serves a collector manager that is not the fuzzy top-k one.  The engine pages through EVERY candidate of the
// query (sg_suggest_batch_from: ascending docID, with the segment and the overlap of each); they are sorted into their
// segments and replayed the way nGramSuggester.Suggest feeds a manager (suggester.go:64-75,78-99): segments inside-out from
// |A|, one collector per segment with NewMetricScorer(metric, |A|, segment), candidates in ascending docID, then
// manager.Collect.  A collector that terminates (merger.ErrCollectionTerminated) ends its own segment only, as
// searcher.Search does.
const suggestPage = 4096

func (e *engine) suggestAny(query string, similarity float64, m metric.Metric, code C.int, tables *C.sg_metric_tables, mgr suggest.CollectorManager) ([]suggest.Candidate, error) {
	blob := []byte(query)
	offs := []C.uint64_t{0, C.uint64_t(len(blob))}
	var bp *C.uint8_t
	if len(blob) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&blob[0]))
	}
	if code < 0 {
		code = C.SG_JACCARD // (ignored: the tables replace it)
	}
	page := suggestPage
	ids, aux, scores := make([]C.uint32_t, page), make([]C.uint32_t, page), make([]C.double, page)
	bySegment := map[int][]merger.MergeCandidate{}
	first, skip := uint32(0), 0
	for {
		var cnt C.uint32_t
		if err := e.retain(); err != nil {
			return nil, err
		}
		err := ccall(func() C.int {
			return C.sg_suggest_batch_from(e.h, bp, &offs[0], 1, code, C.double(similarity), tables, C.uint32_t(first), C.uint32_t(page),
				&ids[0], &scores[0], &aux[0], &cnt)
		})
		e.release()
		if err != nil {
			return nil, err
		}
		if uint32(cnt) == C.SG_COUNT_REF_PANIC {
			panic("makechan: size out of range") // suggester.go:62
		}
		if err := statusError(uint32(cnt), 0); err != nil {
			return nil, err
		}
		n := int(cnt)
		if n > page {
			n = page
		}
		// a document that repeats a term has several entries (cp_merge.go:47-78): a page that ends inside such a run is
		// resumed AT that docID, past the entries already taken
		run := 0
		for j := n - 1; j >= 0 && ids[j] == ids[n-1]; j-- {
			run++
		}
		if n == page && run == n { // a whole page of one document: a larger page
			page *= 2
			ids, aux, scores = make([]C.uint32_t, page), make([]C.uint32_t, page), make([]C.double, page)
			continue
		}
		for j := skip; j < n; j++ {
			seg := int(aux[j] >> 16)
			bySegment[seg] = append(bySegment[seg], merger.NewMergeCandidate(uint32(ids[j]), uint32(aux[j]&0xFFFF)))
		}
		if n < page {
			break
		}
		first, skip = uint32(ids[n-1]), run
	}
	if err := e.retain(); err != nil {
		return nil, err
	}
	sizeA := int(C.sg_tokenize(e.h, bp, C.uint32_t(len(blob)), 0, nil, 0)) // len(tokens), suggester.go:47-53 (no keys wanted: cap 0)
	e.release()
	lo, hi := m.MinY(similarity, sizeA), m.MaxY(similarity, sizeA)
	if hi >= e.segments {
		hi = e.segments - 1
	}
	feed := func(sizeB int) error {
		cands := bySegment[sizeB]
		if len(cands) == 0 {
			return nil
		}
		c := mgr.Create()
		c.SetScorer(suggest.NewMetricScorer(m, sizeA, sizeB))
		for _, mc := range cands {
			if err := c.Collect(mc); err != nil {
				if errors.Is(err, merger.ErrCollectionTerminated) {
					break
				}
				return err
			}
		}
		return mgr.Collect(c)
	}
	for i, j := sizeA, sizeA+1; i >= lo || j <= hi; i, j = i-1, j+1 { // suggester.go:113-121
		if i >= lo {
			if err := feed(i); err != nil {
				return nil, err
			}
		}
		if j <= hi {
			if err := feed(j); err != nil {
				return nil, err
			}
		}
	}
	return mgr.GetCandidates(), nil
}

// suggestBatch is one synchronous call for all queries (sg_suggest_batch_multi over every replica, or sg_suggest_batch_tables
// for a tabulated metric); status[q] carries query q's SG_COUNT_* flag (0: answered, cands[q] valid).
func (e *engine) suggestBatch(queries []string, similarity float64, code C.int, tables *C.sg_metric_tables, k int) ([][]suggest.Candidate, []uint32, error) {
	n := len(queries)
	if n == 0 {
		return nil, nil, nil
	}
	var blob []byte
	offs := []C.uint64_t{0}
	for _, q := range queries {
		blob = append(blob, q...)
		offs = append(offs, C.uint64_t(len(blob)))
	}
	ids := make([]C.uint32_t, n*k)
	scores := make([]C.double, n*k)
	counts := make([]C.uint32_t, n)
	var bp *C.uint8_t
	if len(blob) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&blob[0]))
	}
	if err := e.retain(); err != nil {
		return nil, nil, err
	}
	err := ccall(func() C.int {
		if tables != nil {
			return C.sg_suggest_batch_tables(e.h, bp, &offs[0], C.uint32_t(n), tables, C.uint32_t(k), &ids[0], &scores[0], &counts[0])
		}
		// one replica: the plain call; several: contiguous slices, a worker thread per replica
		return C.sg_suggest_batch_multi(e.h, bp, &offs[0], C.uint32_t(n), code, C.double(similarity), C.uint32_t(k), &ids[0], &scores[0], &counts[0])
	})
	e.release()
	if err != nil {
		return nil, nil, err
	}
	out := make([][]suggest.Candidate, n)
	status := make([]uint32, n)
	for q := 0; q < n; q++ {
		c := uint32(counts[q])
		if c >= C.SG_COUNT_LM_ERROR { // an SG_COUNT_* flag: no row
			status[q] = c
			continue
		}
		out[q] = make([]suggest.Candidate, c)
		for j := uint32(0); j < c; j++ {
			out[q][j] = suggest.Candidate{Key: uint32(ids[q*k+int(j)]), Score: float64(scores[q*k+int(j)])}
		}
	}
	return out, status, nil
}

// SuggestBatch is the additive API the GPU earns its keep on: one kernel launch for all queries.  status[q] != 0 marks a
// query without an answer (C.SG_COUNT_REF_PANIC / _REF_DEADLOCK: the reference panics / dead-locks on it; _TOO_LONG);
// the other rows are valid — one bad query does not fail the batch.
func (i *Index) SuggestBatch(queries []string, similarity float64, m metric.Metric, k int) (cands [][]suggest.Candidate, status []uint32, err error) {
	code, known := metricCode(m)
	var tables *C.sg_metric_tables
	if !known {
		longest := 0
		for _, q := range queries {
			if len(q) > longest {
				longest = len(q)
			}
		}
		if tables, err = i.e.tablesFor(m, similarity, longest+16); err != nil {
			return nil, nil, err
		}
		defer releaseTables(tables)
	}
	cands, status, err = i.e.suggestBatch(queries, similarity, code, tables, k)
	runtime.KeepAlive(i)
	return
}

// Autocomplete implements suggest.Autocomplete (pkg/suggest/autocomplete.go:14-17) for ANY collector manager: the matching
// documents, ascending docID, are replayed through the caller's own collector — first-k (collector.go:48-115) stops after
// its limit, pkg/spellchecker's lmCollectorManager (spellchecker/collector.go:61-79) ranks every one of them with its
// scorer.  The engine hands them over in pages (sg_autocomplete_one_from: the next docIDs at or above first_doc), so a
// prefix with any number of matches is served in full; a collector that terminates ends the paging.  The page grows
// (x4 up to 262 144 entries) while the collector keeps asking: a prefix with M matches costs O(log M) searches, not M / 4096.
const autocompletePage = 4096

func (i *Index) Autocomplete(query string, factory suggest.CollectorManagerFactory) ([]suggest.Candidate, error) {
	e := i.e
	blob := []byte(query)
	page := autocompletePage
	ids := make([]C.uint32_t, page)
	var bp *C.uint8_t
	if len(blob) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&blob[0]))
	}
	mgr := factory()
	c := mgr.Create()
	first, skip := uint32(0), 0
paging:
	for {
		var cnt C.uint32_t
		if err := e.retain(); err != nil {
			return nil, err
		}
		err := ccall(func() C.int {
			return C.sg_autocomplete_one_from(e.h, bp, C.uint32_t(len(blob)), C.uint32_t(first), C.uint32_t(page), &ids[0], &cnt)
		})
		e.release()
		if err != nil {
			return nil, err
		}
		if err := statusError(uint32(cnt), 0); err != nil {
			return nil, err
		}
		n := int(cnt)
		if n > page {
			n = page
		}
		// a document that repeats a term is emitted once per secondary match (list_intersector.go:37-70): a page that ends
		// inside such a run is resumed AT that docID, past the entries already delivered — resuming at last + 1 lost them
		run := 0
		for j := n - 1; j >= 0 && ids[j] == ids[n-1]; j-- {
			run++
		}
		if n == page && run == n { // a whole page of one document
			page *= 2
			ids = make([]C.uint32_t, page)
			continue
		}
		for j := skip; j < n; j++ {
			if err := c.Collect(merger.NewMergeCandidate(uint32(ids[j]), 0)); err != nil {
				if errors.Is(err, merger.ErrCollectionTerminated) {
					break paging
				}
				return nil, err
			}
		}
		if n < page {
			break
		}
		first, skip = uint32(ids[n-1]), run
		if page < 262144 { // the collector wants more than a page: fewer, larger searches from here on
			page *= 4
			ids = make([]C.uint32_t, page)
		}
	}
	runtime.KeepAlive(i)
	if err := mgr.Collect(c); err != nil {
		return nil, err
	}
	return mgr.GetCandidates(), nil
}

// SpellChecker binds pkg/spellchecker.SpellChecker.Predict (spellchecker.go:40-92) to sg_spell_predict_batch: the fuzzy
// index is built over the language model's vocabulary (docID = word id).  NewSpellChecker reads the Google-format count
// files (<dir>/1-gm .. <order>-gm) and numbers the words like `lm build-lm` does (count descending, word ascending,
// pkg/lm/binary.go:101-199); NewSpellCheckerFromBinary is BuildSpellChecker's own path (internal/spellchecker/dep/
// spellchecker.go:13-53): <name>.lm + <name>.cdb through sg_lm_load_binary.
type SpellChecker struct {
	lm     *C.sg_lm
	index  *C.sg_index
	mu     sync.RWMutex
	closed bool
}

// NewSpellChecker mirrors internal/spellchecker/dep.BuildSpellChecker for a model directory.
func NewSpellChecker(dir string, order int, startSymbol, endSymbol string, alphabet []string, d suggest.IndexDescription, device int) (*SpellChecker, error) {
	cdir, cs, ce := C.CString(dir), C.CString(startSymbol), C.CString(endSymbol)
	defer C.free(unsafe.Pointer(cdir))
	defer C.free(unsafe.Pointer(cs))
	defer C.free(unsafe.Pointer(ce))
	lmAlpha, freeLm := cstrings(alphabet)
	defer freeLm()
	sc := &SpellChecker{}
	if err := ccall(func() C.int {
		return C.sg_lm_load_google_ex(cdir, C.uint32_t(order), cs, ce, lmAlpha, C.uint32_t(len(alphabet)), 1, &sc.lm)
	}); err != nil {
		return nil, err
	}
	return sc.finish(d, device)
}

// NewSpellCheckerFromBinary opens what `lm build-lm` left behind — <name>.lm and <name>.cdb — like
// lm.RetrieveLMFromBinary (pkg/lm/binary.go:59-98) inside BuildSpellChecker.
func NewSpellCheckerFromBinary(lmPath, cdbPath, startSymbol, endSymbol string, alphabet []string, d suggest.IndexDescription, device int) (*SpellChecker, error) {
	cl, cc, cs, ce := C.CString(lmPath), C.CString(cdbPath), C.CString(startSymbol), C.CString(endSymbol)
	for _, p := range []*C.char{cl, cc, cs, ce} {
		defer C.free(unsafe.Pointer(p))
	}
	lmAlpha, freeLm := cstrings(alphabet)
	defer freeLm()
	sc := &SpellChecker{}
	if err := ccall(func() C.int {
		return C.sg_lm_load_binary(cl, cc, cs, ce, lmAlpha, C.uint32_t(len(alphabet)), &sc.lm)
	}); err != nil {
		return nil, err
	}
	return sc.finish(d, device)
}

// finish builds the fuzzy index over the model's vocabulary (docID = word id) on `device`.
func (sc *SpellChecker) finish(d suggest.IndexDescription, device int) (*SpellChecker, error) {
	ixAlpha, freeIx := cstrings(d.Alphabet)
	defer freeIx()
	w0, w1, pad := C.CString(d.Wrap[0]), C.CString(d.Wrap[1]), C.CString(d.Pad)
	defer C.free(unsafe.Pointer(w0))
	defer C.free(unsafe.Pointer(w1))
	defer C.free(unsafe.Pointer(pad))
	desc := (*C.sg_desc)(C.malloc(C.size_t(unsafe.Sizeof(C.sg_desc{}))))
	defer C.free(unsafe.Pointer(desc))
	desc.ngram_size, desc.wrap_start, desc.wrap_end, desc.pad = C.uint32_t(d.NGramSize), w0, w1, pad
	desc.alphabet, desc.n_alphabet = ixAlpha, C.uint32_t(len(d.Alphabet))
	if err := ccall(func() C.int { return C.sg_spell_index_build(sc.lm, desc, C.int(device), &sc.index) }); err != nil {
		C.sg_lm_release(sc.lm)
		return nil, err
	}
	runtime.SetFinalizer(sc, func(s *SpellChecker) { s.Close() })
	return sc, nil
}

// Close gives the model and the vocabulary index back (calls in flight hold references of their own).
func (s *SpellChecker) Close() {
	s.mu.Lock()
	defer s.mu.Unlock()
	if s.closed {
		return
	}
	s.closed = true
	C.sg_index_release(s.index)
	C.sg_lm_release(s.lm)
	runtime.SetFinalizer(s, nil)
}

// Predict has the signature of spellchecker.SpellChecker.Predict.
func (s *SpellChecker) Predict(query string, topK int, similarity float64) ([]string, error) {
	if topK < 1 {
		return nil, fmt.Errorf("topK should be greater or equal to 1")
	}
	q := []byte(query)
	offs := []C.uint64_t{0, C.uint64_t(len(q))}
	ids := make([]C.uint32_t, topK+1)
	var count C.uint32_t
	var qp *C.uint8_t
	if len(q) > 0 {
		qp = (*C.uint8_t)(unsafe.Pointer(&q[0]))
	}
	s.mu.RLock() // both handles stay valid while this call is in flight
	if s.closed {
		s.mu.RUnlock()
		return nil, errClosed
	}
	C.sg_index_retain(s.index)
	C.sg_lm_retain(s.lm)
	s.mu.RUnlock()
	err := ccall(func() C.int {
		return C.sg_spell_predict_batch(s.index, s.lm, qp, &offs[0], 1, C.uint32_t(topK), C.double(similarity), &ids[0], &count)
	})
	defer func() { C.sg_lm_release(s.lm); C.sg_index_release(s.index); runtime.KeepAlive(s) }()
	if err != nil {
		return nil, err
	}
	switch uint32(count) {
	case C.SG_COUNT_REF_PANIC:
		panic("makechan: size out of range") // what the reference does (suggester.go:62)
	case C.SG_COUNT_REF_DEADLOCK:
		return nil, fmt.Errorf("suggest_hip: the reference dead-locks on this query (suggester.go:62)")
	case C.SG_COUNT_TOO_LONG:
		return nil, fmt.Errorf("suggest_hip: query word has more than %d n-grams", int(C.SG_MAX_QUERY_TERMS))
	case C.SG_COUNT_LM_ERROR:
		return nil, fmt.Errorf("nGrams length should be less than the nGramModel order") // ngram_model.go:66
	}
	out := make([]string, 0, int(count))
	buf := make([]byte, 512)
	for i := 0; i < int(count); i++ {
		n := int(C.sg_lm_word(s.lm, ids[i], (*C.char)(unsafe.Pointer(&buf[0])), C.uint32_t(len(buf))))
		if n < 0 || n > len(buf) {
			return nil, fmt.Errorf("suggest_hip: bad word id %d", uint32(ids[i]))
		}
		out = append(out, string(buf[:n]))
	}
	return out, nil
}
