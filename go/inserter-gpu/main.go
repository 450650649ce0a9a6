// inserter-gpu: the reference's Kafka consumer (inserter/inserter.go) with its
// Postgres/ClickHouse sink replaced by libflowagg on an MI355X.
//
// SOURCE ONLY: there is no Go toolchain in the build image, so this file has never
// been compiled here.  It is the reference-side binding a maintainer would add; it
// keeps pb-ext/flow.proto (no Go-side decode at all - message.Value goes to the GPU
// as bytes), every inserter flag (inserter/inserter.go:25-42) and the
// ConsumerGroupHandler shape (Setup/Cleanup/ConsumeClaim, inserter.go:167-196).
//
// Differences from the reference, on purpose:
//   - delivery: the reference marks a message before its row is inserted
//     (inserter.go:188: at-most-once on crash).  Here a record sits in HBM until its
//     5-minute window closes, so "the sink accepted it" has two meanings, chosen by
//     -mark.after.close:
//       true  (default): a flush batch is marked only once EVERY window that was open
//              when it was ingested has been emitted (closeWindows).  AT-LEAST-ONCE: a
//              batch that spans a window already emitted and one still open stays
//              unmarked, so a crash (or a rebalance that emits partial windows) replays
//              records whose window the sink has seen - SummingMergeTree then counts
//              them twice.  The sink has to be idempotent per (partition, timeslot)
//              for exactly-once (e.g. ReplacingMergeTree keyed by it, or a dedup on
//              insert); libflowagg reports such records when they come back
//              (fa_stats_t.records_late, ABI 6).  fa_open_timeslots, asked before every
//              flush here, is two scans of the device table (ABI 6), not a copy of it;
//       false: marked as soon as fa_ingest returns (the reference's timing: what the
//              GPU holds and has not emitted is lost with the process);
//     a fatal sink error closes every partition's windows (best effort) before exit,
//     and a claim that ends (rebalance) emits its partition's windows at once;
//   - one fa_ctx per claimed partition (partition p on GPU p % -gpu.devices), no global mutex on the ingest path
//     (inserter.go:84,115): flushes hold a read lock.  The contexts of a session form ONE group (fa_group_*, ABI 7+) and
//     windows are closed for the whole topic under the write lock: flows_5m rows merged over the partitions in HBM (peer
//     copies over xGMI, or RCCL with -gpu.transport=rccl), the heavy hitters of the merged sketches at the end of a
//     session (-out.topk).  Same logic as flow-pipeline_amd/host/inserter_gpu.cpp, which is built and GPU-tested;
//   - insert_count is actually incremented (registered but never Inc()'d at
//     inserter.go:44-49).
package main

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../flow-pipeline_amd -lflowagg
#include <stdlib.h>
#include "flowagg.h"
*/
import "C"

import (
	"context"
	"flag"
	"fmt"
	"net/http"
	"os"
	"os/signal"
	"runtime"
	"strings"
	"sync"
	"sync/atomic"
	"syscall"
	"time"
	"unsafe"

	"github.com/Shopify/sarama"
	"github.com/prometheus/client_golang/prometheus"
	"github.com/prometheus/client_golang/prometheus/promhttp"
	log "github.com/sirupsen/logrus"
)

var (
	// inserter.go:25-42, verbatim names and defaults
	LogLevel = flag.String("loglevel", "info", "Log level")

	MetricsAddr = flag.String("metrics.addr", ":8081", "Metrics address")
	MetricsPath = flag.String("metrics.path", "/metrics", "Metrics path")

	KafkaVersion = flag.String("kafka.version", "2.1.1", "Kafka version")
	KafkaTopic   = flag.String("kafka.topic", "flows-processed", "Kafka topic to consume from")
	KafkaBrk     = flag.String("kafka.brokers", "127.0.0.1:9092,[::1]:9092", "Kafka brokers list separated by commas")
	KafkaGroup   = flag.String("kafka.group", "postgres-inserter", "Kafka group id")
	FlushTime    = flag.Duration("flush.dur", time.Second*5, "Flush duration")
	FlushCount   = flag.Int("flush.count", 100, "Flush count")

	// accepted for command-line compatibility; the GPU sink does not use them
	PostgresUser   = flag.String("postgres.user", "postgres", "Postgres user")
	PostgresPass   = flag.String("postgres.pass", "", "Postgres password")
	PostgresHost   = flag.String("postgres.host", "127.0.0.1", "Postgres host")
	PostgresPort   = flag.Int("postgres.port", 5432, "Postgres port")
	PostgresDbName = flag.String("postgres.dbname", "postgres", "Postgres database")

	// additive flags
	GpuDevices  = flag.Int("gpu.devices", 1, "Number of GPUs; partition p is served by GPU p % gpu.devices")
	ProtoFixed  = flag.Bool("proto.fixedlen", true, "Messages carry the varint length prefix (mocker -proto.fixedlen)")
	WindowSecs  = flag.Int("window.secs", 300, "Rollup window (toStartOfFiveMinute)")
	CloseLagSec = flag.Int("window.lag", 30, "Close a window this many seconds after it ended")
	KeySets     = flag.Int("key.sets", 1, "fa key_sets mask (1 = flows_5m rollup; see include/flowagg.h)")
	OutRowBin   = flag.String("out.rowbinary", "", "Append closed flows_5m rows to this file as ClickHouse RowBinary")
	MarkAfter   = flag.Bool("mark.after.close", true, "Commit a batch's offsets only after every window it touched has been emitted")
	GpuTransport = flag.String("gpu.transport", "peer", "Window-close exchange between the GPUs: peer (hipMemcpyPeer over xGMI) or rccl (sketches by ncclAllReduce)")
	OutTopk      = flag.String("out.topk", "", "At the end of a session write the top -topk.k SrcAddr / DstAddr of the merged Count-Min sketches here (key.sets 2 / 4)")
	TopkK        = flag.Int("topk.k", 100, "Rows of -out.topk per sketch")
	TableLog2    = flag.Int("gpu.table.log2", 0, "log2 slots of a context's flows_5m table (0 = library default)")
	KeysetLog2   = flag.Int("gpu.keyset.log2", 0, "log2 slots of a context's distinct-address sets (0 = library default)")
	WideLog2     = flag.Int("gpu.wide.log2", 0, "log2 slots of a context's (SrcAddr,DstPort,Proto) table (0 = library default)")
	BatchBytes   = flag.Int("gpu.batch.bytes", 64<<20, "A batch that reached -flush.count keeps growing while the claim has messages ready, up to this many bytes (0 = flush exactly at -flush.count)")
	TopkMode     = flag.String("topk.mode", "auto", "Top-k contract (fa_config.topk_mode): exact | candidates | auto (candidates with more than one claimed partition)")
	TopkTrack    = flag.Int("topk.track", 0, "Candidates mode: the rank the admission threshold follows (0 = library default)")

	Inserts = prometheus.NewCounter(prometheus.CounterOpts{Name: "insert_count", Help: "Flow messages aggregated on the GPU."})
)

// one aggregation context per claimed partition (fa_ctx is not thread-safe; distinct ctxs are independent)
type partitionState struct {
	ctx     *C.fa_ctx
	buf     []byte   // message values back to back
	offsets []uint64 // n+1 entries
	pending []*sarama.ConsumerMessage
	// -mark.after.close: flush batches whose records still sit in open windows, oldest first.  Offsets of one
	// partition are monotone and MarkMessage commits "everything up to here", so a batch is one message (its
	// last) plus the timeslots that were open right after it was ingested.
	unmarked []batchMark
}

type batchMark struct {
	last  *sarama.ConsumerMessage
	slots map[uint32]bool
}

type state struct {
	ready chan bool
	lock  sync.Mutex
	parts map[int32]*partitionState
	// the window close of the whole topic (include/flowagg.h, ABI 7): a group call uses EVERY member context, so flushes
	// (fa_ingest on one goroutine's own context) hold closeMu for reading and a close holds it for writing
	group   *C.fa_group
	closeMu sync.RWMutex
	rows5m  []C.fa_row5m // row buffer of the window close (under closeMu): kept between closes
}

func (s *state) metricsHTTP() {
	prometheus.MustRegister(Inserts)
	http.Handle(*MetricsPath, promhttp.Handler())
	log.Fatal(http.ListenAndServe(*MetricsAddr, nil))
}

// candidates: the distinct-address sets hold what can rank (2^16 slots unless -gpu.keyset.log2 says otherwise) instead of every
// address ever seen (2 x 32 B x 2^-gpu.keyset.log2 per partition) - the default once a process serves several partitions
func topkCandidates(claims int) bool {
	return *TopkMode == "candidates" || (*TopkMode == "auto" && claims > 1)
}

func newPartition(partition int32, claims int) *partitionState {
	cfg := C.fa_config{}
	cfg.device = C.int32_t(int(partition) % *GpuDevices)
	cfg.window_secs = C.uint32_t(*WindowSecs)
	cfg.key_sets = C.uint32_t(*KeySets)
	cfg.table_capacity_log2 = C.uint32_t(*TableLog2)
	cfg.topk_capacity_log2 = C.uint32_t(*KeysetLog2)
	cfg.wide_capacity_log2 = C.uint32_t(*WideLog2)
	cfg.topk_track = C.uint32_t(*TopkTrack)
	if topkCandidates(claims) {
		cfg.topk_mode = C.FA_TOPK_CANDIDATES
		if *KeysetLog2 == 0 {
			cfg.topk_capacity_log2 = 16
		}
	}
	if *ProtoFixed {
		cfg.framed = 1
	}
	var ctx *C.fa_ctx
	if rc := C.fa_create(&cfg, &ctx); rc != 0 {
		log.Fatalf("fa_create: %d %s", int(rc), C.GoString(C.fa_last_error(nil))) // sink error is fatal, inserter.go:102-105
	}
	// the staging a batch of -gpu.batch.bytes needs is page-locked here, not inside the consume loop (ABI 8)
	if *BatchBytes > 0 {
		bytes := C.size_t(*BatchBytes + 64<<10)
		recs := bytes / 48
		if recs > 1<<24 {
			recs = 1 << 24
		}
		if rc := C.fa_reserve_ingest(ctx, bytes, recs); rc != 0 {
			log.Fatalf("fa_reserve_ingest: %d %s", int(rc), C.GoString(C.fa_last_error(ctx)))
		}
	}
	return &partitionState{ctx: ctx, offsets: []uint64{0}}
}

// flush = inserter.go:90-111 with the per-row db.Exec loop replaced by one fa_ingest.
func (p *partitionState) flush(s *state, session sarama.ConsumerGroupSession) {
	n := len(p.offsets) - 1
	if n == 0 {
		return
	}
	log.Infof("Processed %d records in the last iteration.", n)
	s.closeMu.RLock() // (not while the group closes a window)
	// fa_ingest copies into library-owned pinned memory before returning (cgo: C keeps no Go pointers)
	rc := C.fa_ingest(p.ctx, (*C.uint8_t)(unsafe.Pointer(&p.buf[0])), C.size_t(len(p.buf)),
		(*C.uint64_t)(unsafe.Pointer(&p.offsets[0])), C.size_t(n))
	var open []uint32
	var openErr error
	if rc == 0 && *MarkAfter {
		open, openErr = p.openTimeslots() // the batch's records can only sit in timeslots that are open now
	}
	s.closeMu.RUnlock() // (released before ANY fatal error: sinkFatal's last-resort close wants the write lock)
	if rc != 0 {
		sinkFatal("fa_ingest: %d %s", int(rc), C.GoString(C.fa_last_error(p.ctx)))
	}
	if openErr != nil {
		sinkFatal("%v", openErr)
	}
	Inserts.Add(float64(n))
	if *MarkAfter {
		set := make(map[uint32]bool, len(open))
		for _, ts := range open {
			set[ts] = true
		}
		p.unmarked = append(p.unmarked, batchMark{last: p.pending[len(p.pending)-1], slots: set})
	} else {
		// after fa_ingest accepted the batch; one mark per batch - offsets of a partition are monotone, so the last message
		// commits what the reference's per-message marks (inserter.go:188) commit
		session.MarkMessage(p.pending[len(p.pending)-1], "")
	}
	p.buf, p.offsets, p.pending = p.buf[:0], p.offsets[:1], p.pending[:0]
}

// openTimeslots lists the windows the context holds; retried with the size the library reports (a backlog
// replay can hold hundreds of windows).  Errors are sink errors - RETURNED, not raised: the callers hold closeMu for reading,
// and sinkFatal's last-resort close needs it for writing (it must be released first).
func (p *partitionState) openTimeslots() ([]uint32, error) {
	slots := make([]C.uint32_t, 64)
	var ns C.size_t
	rc := C.fa_open_timeslots(p.ctx, &slots[0], C.size_t(len(slots)), &ns)
	if rc == C.FA_ERR_CAPACITY {
		slots = make([]C.uint32_t, int(ns))
		rc = C.fa_open_timeslots(p.ctx, &slots[0], C.size_t(len(slots)), &ns)
	}
	if rc != 0 {
		return nil, fmt.Errorf("fa_open_timeslots: %d %s", int(rc), C.GoString(C.fa_last_error(p.ctx)))
	}
	out := make([]uint32, int(ns))
	for i := range out {
		out[i] = uint32(slots[i])
	}
	return out, nil
}

// markEmitted commits the batches (oldest first) none of whose windows is open any more.  The caller holds closeMu for reading
// (or nothing runs beside it: Cleanup); an error comes back to be raised once the lock is released.
func (p *partitionState) markEmitted(session sarama.ConsumerGroupSession) error {
	if len(p.unmarked) == 0 || session == nil {
		return nil
	}
	open := make(map[uint32]bool)
	slots, err := p.openTimeslots()
	if err != nil {
		return err
	}
	for _, ts := range slots {
		open[ts] = true
	}
	done := 0
	for _, b := range p.unmarked {
		still := false
		for ts := range b.slots {
			if open[ts] {
				still = true
				break
			}
		}
		if still {
			break // offsets commit in order: a later batch cannot overtake this one
		}
		session.MarkMessage(b.last, "")
		done++
	}
	p.unmarked = p.unmarked[done:]
	return nil
}

// A sink error is fatal like the reference's failed db.Exec (inserter.go:102-105) - but the contexts hold aggregates.
// The failing goroutine raises `dying`; every ConsumeClaim loop notices it within a second, hands its buffered messages to
// its context and returns; the failing goroutine waits for them (bounded), then - unless the error came out of a group
// call itself - closes every window of the group once more (best effort) and exits the process.
var (
	dying        int32  // 1 once a sink error has been seen
	activeClaims int32  // goroutines inside ConsumeClaim
	inEmergency  int32  // the last-resort close is running: a failure inside it exits at once
	theState     *state // (main sets it)
)

func sinkFatal(format string, args ...interface{}) {
	msg := fmt.Sprintf(format, args...)
	if atomic.LoadInt32(&inEmergency) != 0 {
		log.Fatal(msg)
	}
	if atomic.CompareAndSwapInt32(&dying, 0, 1) {
		log.Error(msg)
		deadline := time.Now().Add(10 * time.Second)
		for atomic.LoadInt32(&activeClaims) > 1 && time.Now().Before(deadline) {
			time.Sleep(50 * time.Millisecond)
		}
		// (TryLock: when this goroutine failed INSIDE a group call it holds the lock already - nothing more to emit then)
		if theState != nil && !strings.HasPrefix(msg, "fa_group_") && theState.closeMu.TryLock() {
			theState.closeMu.Unlock()
			atomic.StoreInt32(&inEmergency, 1)
			theState.closeWindows(time.Now().UTC(), true)
		}
		log.Fatal(msg)
	}
	// a second failure while the process is going down: this goroutine just stops (deferred calls run)
	log.Error(msg)
	runtime.Goexit()
}

// closeWindows emits the finished flows_5m windows of the WHOLE topic (create.sh:70-90), merged over the partitions in
// HBM, as one RowBinary payload per window (`INSERT INTO flows_5m FORMAT RowBinary`) instead of the reference's per-row
// db.Exec (inserter.go:100-106).  One row per key: what flows_5m holds once its SummingMergeTree has merged the
// per-partition inserts, and 1 / partitions of the rows to insert.  Same logic as flow-pipeline_amd/host/inserter_gpu.cpp.
func (s *state) closeWindows(now time.Time, all bool) {
	s.closeMu.Lock()
	defer s.closeMu.Unlock()
	if s.group == nil {
		return
	}
	// any error below is a sink error and fatal, like the reference's failed db.Exec (inserter.go:102-105)
	slots := make([]C.uint32_t, 64)
	var ns C.size_t
	rc := C.fa_group_open_timeslots(s.group, &slots[0], C.size_t(len(slots)), &ns)
	if rc == C.FA_ERR_CAPACITY {
		slots = make([]C.uint32_t, int(ns))
		rc = C.fa_group_open_timeslots(s.group, &slots[0], C.size_t(len(slots)), &ns)
	}
	if rc != 0 {
		sinkFatal("fa_group_open_timeslots: %d %s", int(rc), C.GoString(C.fa_group_last_error(s.group)))
	}
	for _, cts := range slots[:int(ns)] {
		ts := uint32(cts)
		if !all && int64(ts)+int64(*WindowSecs)+int64(*CloseLagSec) > now.Unix() {
			continue
		}
		// (the row buffer is the session's: a buffer that is too small costs the close twice - the library knows the size it needs
		// only after collect + exchange + merge - so it keeps the largest window seen, with room to spare)
		if len(s.rows5m) < 1<<16 {
			s.rows5m = make([]C.fa_row5m, 1<<16)
		}
		rows := s.rows5m
		var nr C.size_t
		rc := C.fa_group_close_window(s.group, C.FA_ROWS_5M, C.uint32_t(ts), unsafe.Pointer(&rows[0]), C.size_t(len(rows)), &nr)
		if rc == C.FA_ERR_CAPACITY { // (nothing was removed: ask again with room)
			s.rows5m = make([]C.fa_row5m, int(nr)+int(nr)/4)
			rows = s.rows5m
			rc = C.fa_group_close_window(s.group, C.FA_ROWS_5M, C.uint32_t(ts), unsafe.Pointer(&rows[0]), C.size_t(len(rows)), &nr)
		}
		if rc != 0 {
			sinkFatal("fa_group_close_window: %d %s", int(rc), C.GoString(C.fa_group_last_error(s.group)))
		}
		// a key set without a sink is closed all the same: (SrcAddr,DstPort,Proto) windows have no output in this shim - kept,
		// the wide table / wide log would grow for the life of the session (inserter_gpu.cpp writes them with -out.app)
		if C.uint32_t(*KeySets)&C.FA_KEYS_ADDR_PORT_PROTO != 0 {
			for _, p := range s.parts {
				if rc := C.fa_drop_window(p.ctx, C.FA_ROWS_APP, C.uint32_t(ts)); rc != 0 {
					sinkFatal("fa_group_close_window: fa_drop_window(FA_ROWS_APP): %d %s", int(rc), C.GoString(C.fa_last_error(p.ctx)))
				}
			}
		}
		log.Infof("flows_5m timeslot %d: %d rows", ts, int(nr))
		if *OutRowBin != "" && nr > 0 {
			buf := make([]byte, int(nr)*C.FA_ROWBINARY_ROW5M_BYTES)
			var nb C.size_t
			if rc := C.fa_rows_to_rowbinary(&rows[0], nr, (*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)), &nb); rc != 0 {
				sinkFatal("fa_rows_to_rowbinary: %d", int(rc))
			}
			appendFile(*OutRowBin, buf[:int(nb)])
		}
	}
}

func appendFile(path string, b []byte) {
	f, err := os.OpenFile(path, os.O_APPEND|os.O_CREATE|os.O_WRONLY, 0644)
	if err != nil {
		sinkFatal("%v", err)
	}
	if _, err = f.Write(b); err != nil {
		sinkFatal("%v", err)
	}
	f.Close()
}

// writeTopk: the heavy hitters of the whole topic (viz-ch.json:233,479) - the members' sketches all-reduced over the GPUs,
// every member's distinct addresses ranked by the MERGED estimate, the first k of their union.
func (s *state) writeTopk() {
	if *OutTopk == "" || s.group == nil {
		return
	}
	for _, ks := range []struct {
		mask C.uint32_t
		name string
	}{{C.FA_KEYS_SRCADDR_CMS, "src"}, {C.FA_KEYS_DSTADDR_CMS, "dst"}} {
		if C.uint32_t(*KeySets)&ks.mask == 0 {
			continue
		}
		rows := make([]C.fa_topk_row, *TopkK)
		var nt C.size_t
		if rc := C.fa_group_topk(s.group, ks.mask, C.size_t(len(rows)), &rows[0], C.size_t(len(rows)), &nt); rc != 0 {
			sinkFatal("fa_group_topk: %d %s", int(rc), C.GoString(C.fa_group_last_error(s.group)))
		}
		var sb strings.Builder
		for _, r := range rows[:int(nt)] {
			fmt.Fprintf(&sb, "%s\t%x\t%d\n", ks.name, C.GoBytes(unsafe.Pointer(&r.key[0]), 16), uint64(r.weight))
		}
		appendFile(*OutTopk, []byte(sb.String()))
	}
}

// Setup: the session knows its claims - one context per claimed partition and ONE group over them for the window close.
func (s *state) Setup(session sarama.ConsumerGroupSession) error {
	s.lock.Lock()
	defer s.lock.Unlock()
	var ctxs []*C.fa_ctx
	claims := 0
	for _, parts := range session.Claims() {
		claims += len(parts)
	}
	for _, parts := range session.Claims() {
		for _, partition := range parts {
			p := newPartition(partition, claims)
			s.parts[partition] = p
			ctxs = append(ctxs, p.ctx)
		}
	}
	if len(ctxs) > 0 {
		flags := C.uint32_t(C.FA_GROUP_PEER)
		if *GpuTransport == "rccl" {
			flags = C.FA_GROUP_RCCL
		}
		// (the array of context pointers is only read during the call: C memory, cgo's pointer rules)
		arr := (**C.fa_ctx)(C.malloc(C.size_t(len(ctxs)) * C.size_t(unsafe.Sizeof(ctxs[0]))))
		copy(unsafe.Slice(arr, len(ctxs)), ctxs)
		rc := C.fa_group_create(arr, C.size_t(len(ctxs)), flags, &s.group)
		C.free(unsafe.Pointer(arr))
		if rc != 0 {
			log.Fatalf("fa_group_create: %d %s", int(rc), C.GoString(C.fa_group_last_error(nil)))
		}
		mode := "exact"
		if topkCandidates(claims) {
			mode = "candidates"
		}
		log.Infof("window close: group of %d context(s) over %d GPU(s), top-k mode %s", len(ctxs), *GpuDevices, mode)
	}
	close(s.ready)
	return nil
}

// Cleanup runs when every ConsumeClaim of the session has returned (rebalance / shutdown): whoever owns the partitions
// next starts from the committed offsets, so everything the GPUs still hold goes to the sink now, the batches are
// committed, and the session's group and contexts go away.
func (s *state) Cleanup(session sarama.ConsumerGroupSession) error {
	s.closeWindows(time.Now().UTC(), true)
	s.writeTopk()
	s.lock.Lock()
	defer s.lock.Unlock()
	if s.group != nil {
		C.fa_group_destroy(s.group)
		s.group = nil
	}
	for part, p := range s.parts {
		if err := p.markEmitted(session); err != nil {
			log.Error(err) // (the windows went out above; the offsets stay uncommitted: at-least-once)
		}
		C.fa_destroy(p.ctx)
		delete(s.parts, part)
	}
	return nil
}

// ConsumeClaim: sarama runs one goroutine per claimed partition (inserter.go:176).
func (s *state) ConsumeClaim(session sarama.ConsumerGroupSession, claim sarama.ConsumerGroupClaim) error {
	s.lock.Lock()
	p, ok := s.parts[claim.Partition()]
	s.lock.Unlock()
	if !ok {
		return fmt.Errorf("partition %d was not claimed in Setup", claim.Partition())
	}
	// one OS thread per partition for the life of the claim: the library selects the ctx's GPU at every entry
	// point anyway, but HIP keeps per-thread state (current device, error state) - a goroutine that hops between OS
	// threads would drag other partitions' state along
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	atomic.AddInt32(&activeClaims, 1)
	defer atomic.AddInt32(&activeClaims, -1)
	timer := time.NewTimer(*FlushTime)
	alive := time.NewTicker(time.Second)
	defer alive.Stop()
	for {
		if atomic.LoadInt32(&dying) != 0 { // another partition hit a sink error: hand over what is buffered, then stop (Cleanup emits)
			p.flush(s, session)
			return nil
		}
		select {
		case <-alive.C:
		case message, open := <-claim.Messages():
			if !open {
				// the claim ends (rebalance / shutdown): the session's Cleanup emits every window of the group and commits
				p.flush(s, session)
				return nil
			}
			p.buf = append(p.buf, message.Value...)
			p.offsets = append(p.offsets, uint64(len(p.buf)))
			p.pending = append(p.pending, message)
			// inserter.go:118-120 - and, beyond the reference: one fa_ingest is a PCIe transfer and a handful of kernel launches
			// whatever its size, so a batch that reached -flush.count keeps growing while the claim has more messages buffered
			// (len of the channel: the next receive would not block), up to -gpu.batch.bytes; a drained claim flushes at once and
			// -flush.dur bounds the wait as before.  (Measured on the C++ twin, inserter_gpu.cpp: profiles/r06_host_phases.json.)
			if len(p.pending) >= *FlushCount && (len(p.buf) >= *BatchBytes || len(claim.Messages()) == 0) {
				p.flush(s, session)
			}
		case <-timer.C: // inserter.go:189-191
			p.flush(s, session)
			s.closeWindows(time.Now().UTC(), false) // (the whole topic's finished windows: whichever goroutine's timer fires first)
			s.closeMu.RLock()
			err := p.markEmitted(session)
			s.closeMu.RUnlock()
			if err != nil {
				sinkFatal("%v", err)
			}
			timer.Reset(*FlushTime)
		}
	}
}

func main() {
	flag.Parse()
	lvl, _ := log.ParseLevel(*LogLevel)
	log.SetLevel(lvl)

	s := &state{ready: make(chan bool), parts: make(map[int32]*partitionState)}
	theState = s
	go s.metricsHTTP()

	config := sarama.NewConfig()
	version, err := sarama.ParseKafkaVersion(*KafkaVersion)
	if err != nil {
		log.Fatal(err)
	}
	config.Version = version
	brokers := strings.Split(*KafkaBrk, ",")
	client, err := sarama.NewConsumerGroup(brokers, *KafkaGroup, config)
	if err != nil {
		log.Fatal(err)
	}
	ctx, cancel := context.WithCancel(context.Background())
	go func() {
		for {
			if err := client.Consume(ctx, strings.Split(*KafkaTopic, ","), s); err != nil {
				log.Fatalf("Error from consumer: %v", err)
			}
			if ctx.Err() != nil {
				return
			}
			s.ready = make(chan bool)
		}
	}()
	<-s.ready
	log.Info("inserter-gpu up and running")
	sigterm := make(chan os.Signal, 1)
	signal.Notify(sigterm, syscall.SIGINT, syscall.SIGTERM)
	<-sigterm
	cancel()
	if err = client.Close(); err != nil {
		log.Fatal(fmt.Sprintf("Error closing client: %v", err))
	}
	// (what the GPUs held went to the sink in the last session's Cleanup, which also destroyed the group and the contexts)
}
