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
//   - messages are marked AFTER fa_ingest returns (the reference marks before the
//     insert, inserter.go:188: at-most-once on crash);
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

	Inserts = prometheus.NewCounter(prometheus.CounterOpts{Name: "insert_count", Help: "Flow messages aggregated on the GPU."})
)

// one aggregation context per claimed partition (fa_ctx is not thread-safe; distinct ctxs are independent)
type partitionState struct {
	ctx     *C.fa_ctx
	buf     []byte   // message values back to back
	offsets []uint64 // n+1 entries
	pending []*sarama.ConsumerMessage
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
		log.Fatalf("fa_ingest: %d %s", int(rc), C.GoString(C.fa_last_error(p.ctx)))
	}
	Inserts.Add(float64(n))
	for _, m := range p.pending {
		session.MarkMessage(m, "") // after the sink accepted the batch
	}
	p.buf, p.offsets, p.pending = p.buf[:0], p.offsets[:1], p.pending[:0]
}

// closeWindows emits finished flows_5m rows (create.sh:70-90) as one RowBinary payload per window
// (`INSERT INTO flows_5m FORMAT RowBinary`) instead of the reference's per-row db.Exec
// (inserter.go:100-106).  Same logic as flow-pipeline_amd/host/inserter_gpu.cpp, which is built and tested.
func (p *partitionState) closeWindows(now time.Time, all bool) {
	// every open timeslot: retry with the size the library reports (a backlog replay can hold hundreds of windows);
	// any other error is a sink error and fatal, like the reference's failed db.Exec (inserter.go:102-105)
	slots := make([]C.uint32_t, 64)
	var ns C.size_t
	rc := C.fa_open_timeslots(p.ctx, &slots[0], C.size_t(len(slots)), &ns)
	if rc == C.FA_ERR_CAPACITY {
		slots = make([]C.uint32_t, int(ns))
		rc = C.fa_open_timeslots(p.ctx, &slots[0], C.size_t(len(slots)), &ns)
	}
	if rc != 0 {
		log.Fatalf("fa_open_timeslots: %d %s", int(rc), C.GoString(C.fa_last_error(p.ctx)))
	}
	for i := 0; i < int(ns); i++ {
		ts := uint32(slots[i])
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
			log.Fatalf("fa_close_window: %d %s", int(rc), C.GoString(C.fa_last_error(p.ctx)))
		}
		log.Infof("flows_5m timeslot %d: %d rows", ts, int(nr))
		if *OutRowBin != "" && nr > 0 {
			buf := make([]byte, int(nr)*C.FA_ROWBINARY_ROW5M_BYTES)
			var nb C.size_t
			if rc := C.fa_rows_to_rowbinary(&rows[0], nr, (*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)), &nb); rc != 0 {
				log.Fatalf("fa_rows_to_rowbinary: %d", int(rc))
			}
			f, err := os.OpenFile(*OutRowBin, os.O_APPEND|os.O_CREATE|os.O_WRONLY, 0644)
			if err != nil {
				log.Fatal(err)
			}
			if _, err = f.Write(buf[:int(nb)]); err != nil {
				log.Fatal(err)
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
	timer := time.NewTimer(*FlushTime)
	for {
		select {
		case message, open := <-claim.Messages():
			if !open {
				p.flush(session)
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
		// the offsets of everything ingested are committed: what the GPU still holds must reach the sink before the
		// contexts go away (the C++ twin: CloseAllAtEnd)
		p.closeWindows(time.Now().UTC(), true)
		C.fa_destroy(p.ctx)
	}
}
