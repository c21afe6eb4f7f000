// Package suggesthip plugs libsuggest_hip.so (MI355X engine) into suggest-go behind the reference's own
// extension seam: it implements suggest.Builder (pkg/suggest/ngram_index_builder.go:14-17) and
// suggest.NGramIndex = Suggester + Autocomplete (pkg/suggest/ngram_index.go:7-10), so
// Service.AddIndex / Suggest / Autocomplete callers (pkg/suggest/service.go:78-173) are unchanged.
//
// NOT compiled in this repository (no Go toolchain in the build image); shipped as the binding a
// maintainer adds next to pkg/suggest.  Build: CGO_ENABLED=1, -I<repo>/include, -L<repo>/suggest_amd.
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
	"runtime"
	"sync"
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
	// OnDevice builds the index on the GPU (sg_index_build_device: same arrays, ~16x faster at 10 M strings); documents
	// with more than 128 n-grams make it fall back to the host builder.
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
	alpha := make([]*C.char, len(d.Alphabet))
	for i, a := range d.Alphabet {
		alpha[i] = C.CString(a)
		defer C.free(unsafe.Pointer(alpha[i]))
	}
	w0, w1, pad := C.CString(d.Wrap[0]), C.CString(d.Wrap[1]), C.CString(d.Pad)
	defer C.free(unsafe.Pointer(w0))
	defer C.free(unsafe.Pointer(w1))
	defer C.free(unsafe.Pointer(pad))
	// the descriptor lives in C memory for the duration of the call (cgo pointer rules)
	desc := (*C.sg_desc)(C.malloc(C.size_t(unsafe.Sizeof(C.sg_desc{}))))
	defer C.free(unsafe.Pointer(desc))
	arr := (**C.char)(C.malloc(C.size_t(len(alpha)+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(arr))
	copy(unsafe.Slice(arr, len(alpha)), alpha)
	desc.ngram_size, desc.wrap_start, desc.wrap_end, desc.pad = C.uint32_t(d.NGramSize), w0, w1, pad
	desc.alphabet, desc.n_alphabet = arr, C.uint32_t(len(alpha))

	var h *C.sg_index
	var bp *C.uint8_t
	if len(blob) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&blob[0]))
	}
	rc := C.int(C.SG_E_UNSUPPORTED)
	if b.OnDevice {
		rc = C.sg_index_build_device(bp, &offs[0], C.uint32_t(len(offs)-1), desc, C.int(b.Device), &h)
	}
	if rc == C.SG_E_UNSUPPORTED {
		rc = C.sg_index_build(bp, &offs[0], C.uint32_t(len(offs)-1), desc, &h)
	}
	if rc != 0 {
		return nil, lastError(rc)
	}
	if rc := C.sg_index_upload(h, C.int(b.Device)); rc != 0 {
		C.sg_index_release(h)
		return nil, lastError(rc)
	}
	if len(b.Devices) > 0 {
		devs := make([]C.int, len(b.Devices))
		for i, d := range b.Devices {
			devs[i] = C.int(d)
		}
		if rc := C.sg_index_replicate(h, &devs[0], C.uint32_t(len(devs))); rc != 0 {
			C.sg_index_release(h)
			return nil, lastError(rc)
		}
	}
	ix := &Index{h: h, reqs: make(chan *request, 4096)}
	go ix.dispatch()
	runtime.SetFinalizer(ix, func(i *Index) { C.sg_index_release(i.h) }) // cf. pkg/index/index_reader.go:49-51
	return ix, nil
}

// Index is the GPU-resident NGramIndex.  Single-query calls (the reference's calling pattern: one Suggest per goroutine,
// pkg/suggest/service_test.go:36-79) are coalesced HERE, in Go: a goroutine parks on a channel for a fraction of what a
// blocked cgo call costs, so the C-side queue (sg_suggest_one) is left to C / C++ / Python callers.
type Index struct {
	h    *C.sg_index
	reqs chan *request
}

type request struct {
	query      string
	similarity float64
	m          metric.Metric
	code       C.int
	k          int
	resp       chan response
}

type response struct {
	cands []suggest.Candidate
	err   error
}

// dispatch runs whatever single-query requests are pending as one batch per distinct (metric, similarity, k): no timer —
// an idle engine serves a lone request at once, a busy one finds the requests that arrived meanwhile.
func (i *Index) dispatch() {
	for first := range i.reqs {
		batch := []*request{first}
	drain:
		for len(batch) < 8192 {
			select {
			case r, ok := <-i.reqs:
				if !ok {
					break drain
				}
				batch = append(batch, r)
			default:
				break drain
			}
		}
		for len(batch) > 0 {
			head := batch[0]
			var same, rest []*request
			for _, r := range batch {
				if r.code == head.code && r.similarity == head.similarity && r.k == head.k {
					same = append(same, r)
				} else {
					rest = append(rest, r)
				}
			}
			qs := make([]string, len(same))
			for j, r := range same {
				qs[j] = r.query
			}
			res, err := i.SuggestBatch(qs, head.similarity, head.m, head.k)
			for j, r := range same {
				if err != nil {
					r.resp <- response{nil, err}
				} else {
					r.resp <- response{res[j], nil}
				}
			}
			batch = rest
		}
	}
}

var respPool = sync.Pool{New: func() interface{} { return make(chan response, 1) }}

func lastError(rc C.int) error { return fmt.Errorf("suggest_hip %d: %s", int(rc), C.GoString(C.sg_last_error())) }

// metricCode recovers the metric behaviourally: the Metric interface is opaque at this seam
// (unexported types), but its four methods identify it on two probe points.
func metricCode(m metric.Metric) (C.int, error) {
	probes := []struct {
		code C.int
		m    metric.Metric
	}{{C.SG_JACCARD, metric.JaccardMetric()}, {C.SG_COSINE, metric.CosineMetric()}, {C.SG_DICE, metric.DiceMetric()},
		{C.SG_EXACT, metric.ExactMetric()}, {C.SG_OVERLAP, metric.OverlapMetric()}}
	for _, p := range probes {
		if m.Threshold(0.37, 11, 17) == p.m.Threshold(0.37, 11, 17) && m.MinY(0.37, 11) == p.m.MinY(0.37, 11) &&
			m.MaxY(0.37, 11) == p.m.MaxY(0.37, 11) && m.Distance(5, 11, 17) == p.m.Distance(5, 11, 17) {
			return p.code, nil
		}
	}
	return 0, errors.New("suggesthip: unsupported metric implementation")
}

// topK recovers k from the collector-manager factory (the closure hides it, pkg/suggest/collector.go:143-149): a manager
// fed n distinct candidates returns min(n, k) of them.  Doubling, then bisection: ~2 log2(k) probes, each a fresh manager.
func topK(factory suggest.CollectorManagerFactory) int {
	holds := func(n int) bool { // does the queue keep all of n candidates?
		mgr := factory()
		c := mgr.Create()
		c.SetScorer(constScorer(1))
		for d := 1; d <= n; d++ {
			if c.Collect(merger.NewMergeCandidate(uint32(d), 1)) != nil {
				break
			}
		}
		_ = mgr.Collect(c)
		return len(mgr.GetCandidates()) == n
	}
	hi := 1
	for hi < 1024 && holds(hi*2) {
		hi *= 2
	}
	lo, top := hi, hi*2 // holds(lo), !holds(top) (or top > 1024)
	if top > 1024 {
		top = 1025
	}
	for top-lo > 1 {
		mid := (lo + top) / 2
		if holds(mid) {
			lo = mid
		} else {
			top = mid
		}
	}
	return lo
}

type constScorer float64

func (s constScorer) Score(merger.MergeCandidate) float64 { return float64(s) }

// Suggest implements suggest.Suggester (pkg/suggest/suggester.go:17-20): the request joins whatever other goroutines are
// asking at the moment and is answered from their common launch.
func (i *Index) Suggest(query string, similarity float64, m metric.Metric, factory suggest.CollectorManagerFactory) ([]suggest.Candidate, error) {
	code, err := metricCode(m)
	if err != nil {
		return nil, err
	}
	r := &request{query: query, similarity: similarity, m: m, code: code, k: topK(factory), resp: respPool.Get().(chan response)}
	i.reqs <- r
	out := <-r.resp
	respPool.Put(r.resp)
	return out.cands, out.err
}

// SuggestBatch is the additive API the GPU earns its keep on: one kernel launch for all queries.
func (i *Index) SuggestBatch(queries []string, similarity float64, m metric.Metric, k int) ([][]suggest.Candidate, error) {
	code, err := metricCode(m)
	if err != nil {
		return nil, err
	}
	var blob []byte
	offs := []C.uint64_t{0}
	for _, q := range queries {
		blob = append(blob, q...)
		offs = append(offs, C.uint64_t(len(blob)))
	}
	n := len(queries)
	ids := make([]C.uint32_t, n*k)
	scores := make([]C.double, n*k)
	counts := make([]C.uint32_t, n)
	var bp *C.uint8_t
	if len(blob) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&blob[0]))
	}
	C.sg_index_retain(i.h) // the handle stays valid while this query is in flight (service.go:85-88 swaps indexes)
	rc := C.sg_suggest_batch_multi(i.h, bp, &offs[0], C.uint32_t(n), code, C.double(similarity), C.uint32_t(k), &ids[0], &scores[0], &counts[0]) // one replica: the plain call
	C.sg_index_release(i.h)
	runtime.KeepAlive(i)
	if rc != 0 {
		return nil, lastError(rc)
	}
	out := make([][]suggest.Candidate, n)
	for q := 0; q < n; q++ {
		c := uint32(counts[q])
		switch c {
		case C.SG_COUNT_REF_PANIC:
			panic("makechan: size out of range") // what pkg/suggest/suggester.go:62 does on this query
		case C.SG_COUNT_REF_DEADLOCK, C.SG_COUNT_TOO_LONG:
			return nil, fmt.Errorf("suggesthip: query %d cannot be answered (status %#x)", q, c)
		}
		out[q] = make([]suggest.Candidate, c)
		for j := uint32(0); j < c; j++ {
			out[q][j] = suggest.Candidate{Key: uint32(ids[q*k+int(j)]), Score: float64(scores[q*k+int(j)])}
		}
	}
	return out, nil
}

// Autocomplete implements suggest.Autocomplete (pkg/suggest/autocomplete.go:14-17) for ANY collector manager: the matching
// documents (ascending docID, as the reference's segments deliver them) are replayed through the caller's own collector —
// first-k (collector.go:48-115) stops after its limit, pkg/spellchecker's lmCollectorManager ranks them with its scorer.
// The engine hands over at most SG_MAX_TOPK documents per query; a prefix matched by more cannot be served to a collector
// that wants them all and is reported as an error rather than answered short.
func (i *Index) Autocomplete(query string, factory suggest.CollectorManagerFactory) ([]suggest.Candidate, error) {
	const limit = 1024 // SG_MAX_TOPK
	blob := []byte(query)
	ids := make([]C.uint32_t, limit)
	var cnt C.uint32_t
	var bp *C.uint8_t
	if len(blob) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&blob[0]))
	}
	C.sg_index_retain(i.h)
	rc := C.sg_autocomplete_one(i.h, bp, C.uint32_t(len(blob)), C.uint32_t(limit), &ids[0], &cnt)
	C.sg_index_release(i.h)
	runtime.KeepAlive(i)
	if rc != 0 {
		return nil, lastError(rc)
	}
	if uint32(cnt) == C.SG_COUNT_TOO_LONG {
		return nil, fmt.Errorf("suggesthip: query has more than 128 n-grams")
	}
	mgr := factory()
	c := mgr.Create()
	terminated := false
	for j := 0; j < int(cnt); j++ {
		if err := c.Collect(merger.NewMergeCandidate(uint32(ids[j]), 0)); err != nil {
			if errors.Is(err, merger.ErrCollectionTerminated) {
				terminated = true
				break
			}
			return nil, err
		}
	}
	if int(cnt) == limit && !terminated {
		return nil, fmt.Errorf("suggesthip: more than %d documents complete %q; the collector wants them all", limit, query)
	}
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
	lm    *C.sg_lm
	index *C.sg_index
}

// NewSpellChecker mirrors internal/spellchecker/dep.BuildSpellChecker for a model directory.
func NewSpellChecker(dir string, order int, startSymbol, endSymbol string, alphabet []string, d suggest.IndexDescription, device int) (*SpellChecker, error) {
	cdir, cs, ce := C.CString(dir), C.CString(startSymbol), C.CString(endSymbol)
	defer C.free(unsafe.Pointer(cdir))
	defer C.free(unsafe.Pointer(cs))
	defer C.free(unsafe.Pointer(ce))
	cstrings := func(xs []string) (**C.char, func()) {
		arr := (**C.char)(C.malloc(C.size_t(len(xs)+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
		sl := unsafe.Slice(arr, len(xs))
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
	lmAlpha, freeLm := cstrings(alphabet)
	defer freeLm()
	sc := &SpellChecker{}
	if rc := C.sg_lm_load_google_ex(cdir, C.uint32_t(order), cs, ce, lmAlpha, C.uint32_t(len(alphabet)), 1, &sc.lm); rc != 0 {
		return nil, lastError(rc)
	}
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
	if rc := C.sg_spell_index_build(sc.lm, desc, C.int(device), &sc.index); rc != 0 {
		C.sg_lm_release(sc.lm)
		return nil, lastError(rc)
	}
	runtime.SetFinalizer(sc, func(s *SpellChecker) { C.sg_index_release(s.index); C.sg_lm_release(s.lm) })
	return sc, nil
}

// NewSpellCheckerFromBinary opens what `lm build-lm` left behind — <name>.lm and <name>.cdb — like
// lm.RetrieveLMFromBinary (pkg/lm/binary.go:59-98) inside BuildSpellChecker.
func NewSpellCheckerFromBinary(lmPath, cdbPath, startSymbol, endSymbol string, alphabet []string, d suggest.IndexDescription, device int) (*SpellChecker, error) {
	cl, cc, cs, ce := C.CString(lmPath), C.CString(cdbPath), C.CString(startSymbol), C.CString(endSymbol)
	for _, p := range []*C.char{cl, cc, cs, ce} {
		defer C.free(unsafe.Pointer(p))
	}
	mk := func(xs []string) (**C.char, func()) {
		arr := (**C.char)(C.malloc(C.size_t(len(xs)+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
		sl := unsafe.Slice(arr, len(xs))
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
	lmAlpha, freeLm := mk(alphabet)
	defer freeLm()
	sc := &SpellChecker{}
	if rc := C.sg_lm_load_binary(cl, cc, cs, ce, lmAlpha, C.uint32_t(len(alphabet)), &sc.lm); rc != 0 {
		return nil, lastError(rc)
	}
	ixAlpha, freeIx := mk(d.Alphabet)
	defer freeIx()
	w0, w1, pad := C.CString(d.Wrap[0]), C.CString(d.Wrap[1]), C.CString(d.Pad)
	defer C.free(unsafe.Pointer(w0))
	defer C.free(unsafe.Pointer(w1))
	defer C.free(unsafe.Pointer(pad))
	desc := (*C.sg_desc)(C.malloc(C.size_t(unsafe.Sizeof(C.sg_desc{}))))
	defer C.free(unsafe.Pointer(desc))
	desc.ngram_size, desc.wrap_start, desc.wrap_end, desc.pad = C.uint32_t(d.NGramSize), w0, w1, pad
	desc.alphabet, desc.n_alphabet = ixAlpha, C.uint32_t(len(d.Alphabet))
	if rc := C.sg_spell_index_build(sc.lm, desc, C.int(device), &sc.index); rc != 0 {
		C.sg_lm_release(sc.lm)
		return nil, lastError(rc)
	}
	runtime.SetFinalizer(sc, func(s *SpellChecker) { C.sg_index_release(s.index); C.sg_lm_release(s.lm) })
	return sc, nil
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
	if rc := C.sg_spell_predict_batch(s.index, s.lm, qp, &offs[0], 1, C.uint32_t(topK), C.double(similarity), &ids[0], &count); rc != 0 {
		return nil, lastError(rc)
	}
	switch uint32(count) {
	case 0xFFFFFFFF:
		panic("makechan: size out of range") // what the reference does (suggester.go:62)
	case 0xFFFFFFFE:
		return nil, fmt.Errorf("suggest_hip: the reference dead-locks on this query (suggester.go:62)")
	case 0xFFFFFFFD:
		return nil, fmt.Errorf("suggest_hip: query word has more than 128 n-grams")
	case 0xFFFFFFFC:
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
