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
//   - one fa_ctx per claimed partition, no global mutex (inserter.go:84,115);
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
}

func (s *state) metricsHTTP() {
	prometheus.MustRegister(Inserts)
	http.Handle(*MetricsPath, promhttp.Handler())
	log.Fatal(http.ListenAndServe(*MetricsAddr, nil))
}

func newPartition(partition int32) *partitionState {
	cfg := C.fa_config{}
	cfg.device = C.int32_t(int(partition) % *GpuDevices)
	cfg.window_secs = C.uint32_t(*WindowSecs)
	cfg.key_sets = C.uint32_t(*KeySets)
	if *ProtoFixed {
		cfg.framed = 1
	}
	var ctx *C.fa_ctx
	if rc := C.fa_create(&cfg, &ctx); rc != 0 {
		log.Fatalf("fa_create: %d %s", int(rc), C.GoString(C.fa_last_error(nil))) // sink error is fatal, inserter.go:102-105
	}
	return &partitionState{ctx: ctx, offsets: []uint64{0}}
}

// flush = inserter.go:90-111 with the per-row db.Exec loop replaced by one fa_ingest.
func (p *partitionState) flush(session sarama.ConsumerGroupSession) {
	n := len(p.offsets) - 1
	if n == 0 {
		return
	}
	log.Infof("Processed %d records in the last iteration.", n)
	// fa_ingest copies into library-owned pinned memory before returning (cgo: C keeps no Go pointers)
	rc := C.fa_ingest(p.ctx, (*C.uint8_t)(unsafe.Pointer(&p.buf[0])), C.size_t(len(p.buf)),
		(*C.uint64_t)(unsafe.Pointer(&p.offsets[0])), C.size_t(n))
	if rc != 0 {
		sinkFatal("fa_ingest: %d %s", int(rc), C.GoString(C.fa_last_error(p.ctx)))
	}
	Inserts.Add(float64(n))
	if *MarkAfter {
		// the batch's records can only sit in timeslots that are open now
		open := p.openTimeslots()
		set := make(map[uint32]bool, len(open))
		for _, ts := range open {
			set[ts] = true
		}
		p.unmarked = append(p.unmarked, batchMark{last: p.pending[len(p.pending)-1], slots: set})
	} else {
		for _, m := range p.pending {
			session.MarkMessage(m, "") // after fa_ingest accepted the batch (the reference's timing)
		}
	}
	p.buf, p.offsets, p.pending = p.buf[:0], p.offsets[:1], p.pending[:0]
}

// openTimeslots lists the windows the context holds; retried with the size the library reports (a backlog
// replay can hold hundreds of windows).  Errors are sink errors.
func (p *partitionState) openTimeslots() []uint32 {
	slots := make([]C.uint32_t, 64)
	var ns C.size_t
	rc := C.fa_open_timeslots(p.ctx, &slots[0], C.size_t(len(slots)), &ns)
	if rc == C.FA_ERR_CAPACITY {
		slots = make([]C.uint32_t, int(ns))
		rc = C.fa_open_timeslots(p.ctx, &slots[0], C.size_t(len(slots)), &ns)
	}
	if rc != 0 {
		sinkFatal("fa_open_timeslots: %d %s", int(rc), C.GoString(C.fa_last_error(p.ctx)))
	}
	out := make([]uint32, int(ns))
	for i := range out {
		out[i] = uint32(slots[i])
	}
	return out
}

// markEmitted commits the batches (oldest first) none of whose windows is open any more.
func (p *partitionState) markEmitted(session sarama.ConsumerGroupSession) {
	if len(p.unmarked) == 0 || session == nil {
		return
	}
	open := make(map[uint32]bool)
	for _, ts := range p.openTimeslots() {
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
}

// A sink error is fatal like the reference's failed db.Exec (inserter.go:102-105) - but the other partitions'
// contexts hold aggregates too.  A context is not thread-safe, so nobody closes another goroutine's windows: the
// failing goroutine raises `dying`, every ConsumeClaim loop notices it within a second, emits what its own context
// holds (closeWindows(all)) and returns; the failing goroutine waits for them (bounded) and then exits the process.
var (
	dying        int32 // 1 once a sink error has been seen
	activeClaims int32 // goroutines inside ConsumeClaim
)

func sinkFatal(format string, args ...interface{}) {
	msg := fmt.Sprintf(format, args...)
	if atomic.CompareAndSwapInt32(&dying, 0, 1) {
		log.Error(msg)
		deadline := time.Now().Add(10 * time.Second)
		for atomic.LoadInt32(&activeClaims) > 1 && time.Now().Before(deadline) {
			time.Sleep(50 * time.Millisecond)
		}
		log.Fatal(msg)
	}
	// a second failure while the process is going down: this goroutine just stops (deferred calls run)
	log.Error(msg)
	runtime.Goexit()
}

// closeWindows emits finished flows_5m rows (create.sh:70-90) as one RowBinary payload per window
// (`INSERT INTO flows_5m FORMAT RowBinary`) instead of the reference's per-row db.Exec
// (inserter.go:100-106).  Same logic as flow-pipeline_amd/host/inserter_gpu.cpp, which is built and tested.
func (p *partitionState) closeWindows(now time.Time, all bool) {
	// any error below is a sink error and fatal, like the reference's failed db.Exec (inserter.go:102-105)
	slots := p.openTimeslots()
	for _, ts := range slots {
		if !all && int64(ts)+int64(*WindowSecs)+int64(*CloseLagSec) > now.Unix() {
			continue
		}
		rows := make([]C.fa_row5m, 1<<16)
		var nr C.size_t
		rc := C.fa_close_window(p.ctx, C.uint32_t(ts), &rows[0], C.size_t(len(rows)), &nr)
		if rc == C.FA_ERR_CAPACITY {
			rows = make([]C.fa_row5m, int(nr))
			rc = C.fa_close_window(p.ctx, C.uint32_t(ts), &rows[0], C.size_t(len(rows)), &nr)
		}
		if rc != 0 {
			sinkFatal("fa_close_window: %d %s", int(rc), C.GoString(C.fa_last_error(p.ctx)))
		}
		log.Infof("flows_5m timeslot %d: %d rows", ts, int(nr))
		if *OutRowBin != "" && nr > 0 {
			buf := make([]byte, int(nr)*C.FA_ROWBINARY_ROW5M_BYTES)
			var nb C.size_t
			if rc := C.fa_rows_to_rowbinary(&rows[0], nr, (*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)), &nb); rc != 0 {
				sinkFatal("fa_rows_to_rowbinary: %d", int(rc))
			}
			f, err := os.OpenFile(*OutRowBin, os.O_APPEND|os.O_CREATE|os.O_WRONLY, 0644)
			if err != nil {
				sinkFatal("%v", err)
			}
			if _, err = f.Write(buf[:int(nb)]); err != nil {
				sinkFatal("%v", err)
			}
			f.Close()
		}
	}
}

func (s *state) Setup(sarama.ConsumerGroupSession) error {
	close(s.ready)
	return nil
}

func (s *state) Cleanup(sarama.ConsumerGroupSession) error { return nil }

// ConsumeClaim: sarama runs one goroutine per claimed partition (inserter.go:176).
func (s *state) ConsumeClaim(session sarama.ConsumerGroupSession, claim sarama.ConsumerGroupClaim) error {
	s.lock.Lock()
	p, ok := s.parts[claim.Partition()]
	if !ok {
		p = newPartition(claim.Partition())
		s.parts[claim.Partition()] = p
	}
	s.lock.Unlock()
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
		if atomic.LoadInt32(&dying) != 0 { // another partition hit a sink error: emit what this context holds, then stop
			p.flush(session)
			p.closeWindows(time.Now().UTC(), true)
			p.markEmitted(session)
			return nil
		}
		select {
		case <-alive.C:
		case message, open := <-claim.Messages():
			if !open {
				// the claim ends (rebalance / shutdown): whoever owns the partition next starts from the committed
				// offsets, so what this context holds goes to the sink now and its batches are committed
				p.flush(session)
				p.closeWindows(time.Now().UTC(), true)
				p.markEmitted(session)
				return nil
			}
			p.buf = append(p.buf, message.Value...)
			p.offsets = append(p.offsets, uint64(len(p.buf)))
			p.pending = append(p.pending, message)
			if len(p.pending) >= *FlushCount { // inserter.go:118-120
				p.flush(session)
			}
		case <-timer.C: // inserter.go:189-191
			p.flush(session)
			p.closeWindows(time.Now().UTC(), false)
			p.markEmitted(session)
			timer.Reset(*FlushTime)
		}
	}
}

func main() {
	flag.Parse()
	lvl, _ := log.ParseLevel(*LogLevel)
	log.SetLevel(lvl)

	s := &state{ready: make(chan bool), parts: make(map[int32]*partitionState)}
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
	for _, p := range s.parts {
		// what the GPU still holds must reach the sink before the contexts go away (the C++ twin: CloseAllAtEnd);
		// claims that ended have emitted already (ConsumeClaim), this is the safety net
		p.closeWindows(time.Now().UTC(), true)
		C.fa_destroy(p.ctx)
	}
}
