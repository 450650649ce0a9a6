// inserter_gpu - C++ host side above the C-ABI: the reference's Kafka consumer
// (inserter/inserter.go) with its per-row Postgres sink replaced by libflowagg on an MI355X.
//
// The reference host is Go (sarama consumer group).  There is no Go toolchain in this image, so the
// host side is written in C++ and mirrors the reference's interface for this path name by name:
//   flags                 inserter.go:25-42  (same names, defaults and meaning; -postgres.* accepted and unused)
//   ConsumerGroupHandler  Setup / Cleanup / ConsumeClaim            inserter.go:167-196
//   buffer -> flush       by -flush.count and by -flush.dur          inserter.go:113-120,189-191
//   error policy          malformed record: counted + dropped (inside the library, inserter.go:125-126);
//                         sink error: fatal (log.Fatal, inserter.go:102-105)
//   insert_count          the Prometheus counter of inserter.go:44-49 (written to -metrics.dump at exit)
// The cgo twin of this file (what a maintainer of the reference would add) is go/inserter-gpu/main.go.
//
// Kafka itself is out of scope (no broker in this image): a claim's message stream is read
// from a partition log file - the message values back to back, which is self-delimiting when the producer
// runs with -proto.fixedlen=true (mocker.go:98-101; one framed record per message value), or with a
// 4-byte little-endian length in front of every value (-input.format=len32) for bare records.
//
// Differences from the reference, on purpose (same as the Go shim):
//   - messages are marked AFTER the sink accepted the batch (the reference marks first, inserter.go:188);
//   - one aggregation context per claimed partition, no global mutex (inserter.go:84,115): partition p lives on GPU
//     p % -gpu.devices; the contexts form ONE group (fa_group_*, include/flowagg.h ABI 7) and windows are closed for the
//     whole topic - flows_5m rows merged over the partitions in HBM, (SrcAddr,DstPort,Proto) rows hash-partitioned over
//     the GPUs, sketches all-reduced, top-k of the merged sketch.  Flushes (fa_ingest, one goroutine's ctx) hold a read
//     lock, a close the write lock: a group call uses every member;
//   - insert_count is actually incremented;
//   - -gpu.table.log2 / -gpu.keyset.log2 / -gpu.wide.log2 size a context's tables (fa_config; 0 = the library's defaults);
//   - batches: -flush.count is the reference's trigger (inserter.go:118-120).  One fa_ingest is a PCIe transfer and a handful of
//     kernel launches whatever its size, so while the claim has more messages READY (sarama: len(claim.Messages()) > 0) a batch
//     that reached -flush.count keeps growing up to -gpu.batch.bytes (default 64 MiB; 0 = flush exactly at -flush.count);
//     a consumer that has drained its claim flushes at once, and -flush.dur bounds the wait as in the reference (:189-191).
//     Messages that lie back to back in the client's fetch buffer (a partition log does; a Kafka record batch's values do not,
//     there are record headers between them) are handed over in place - fa_ingest copies into its pinned staging anyway;
//   - -topk.mode exact|candidates (fa_config.topk_mode): candidates (64 k slots per set) is the default with more than one claimed
//     partition - the exact mode's sets hold EVERY address (2 x 32 B x 2^-gpu.keyset.log2 per partition);
//   - a key set without a sink (-key.sets includes 8, no -out.app) is closed all the same: its windows are dropped, not kept.
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/flowagg.h"

// ---- flags (Go `flag` syntax: -name=value, -name value, --name=value; bools: -name / -name=false) -------
struct Flags {
    std::string LogLevel = "info";
    std::string MetricsAddr = ":8081", MetricsPath = "/metrics";
    std::string KafkaVersion = "2.1.1", KafkaTopic = "flows-processed", KafkaBrk = "127.0.0.1:9092,[::1]:9092",
                KafkaGroup = "postgres-inserter";
    double FlushTime = 5.0;  // -flush.dur (seconds; Go duration syntax accepted)
    long FlushCount = 100;   // -flush.count
    std::string PostgresUser = "postgres", PostgresPass, PostgresHost = "127.0.0.1", PostgresDbName = "postgres";
    long PostgresPort = 5432;
    // additive
    long GpuDevices = 1;
    bool ProtoFixed = true;
    long WindowSecs = 300, CloseLagSec = 30;
    long KeySets = FA_KEYS_AS_PAIR;
    std::string InputFiles, InputFormat = "framed";
    std::string OutRowBinary, OutTsv, OffsetsOut, MetricsDump;
    std::string OutApp, OutTopk, GpuTransport = "peer";  // raw fa_row_app records / "src|dst <hex key> <weight>" lines; peer | rccl
    long TopkK = 100;
    long TableLog2 = 0, KeysetLog2 = 0, WideLog2 = 0;  // fa_config capacities (0 = the library's defaults)
    long BatchBytes = 64l << 20;   // -gpu.batch.bytes: how far a batch may grow beyond -flush.count while messages are ready (0: not at all)
    std::string TopkMode = "auto"; // exact | candidates | auto (candidates with more than one claimed partition)
    long TopkTrack = 0;            // fa_config.topk_track (0 = the library's default)
    std::string PhasesOut;         // -phases.out: where the time of the consume loop went, per partition thread (JSON)
    bool Prefault = true;          // -input.prefault: map partition logs with MAP_POPULATE (a Kafka client's fetch buffers are resident too)
    bool DryRun = false;  // test double for the host logic: batches are logged, nothing is computed
    bool CloseAllAtEnd = true;
};

static int g_loglevel = 2;  // 0 error, 1 warn, 2 info, 3 debug
static std::mutex g_logmu;
static void logf(int lvl, const char* fmt, ...) {
    if (lvl > g_loglevel) return;
    std::lock_guard<std::mutex> g(g_logmu);
    static const char* names[] = {"error", "warning", "info", "debug"};
    fprintf(stderr, "level=%s msg=\"", names[lvl]);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\"\n");
}
[[noreturn]] static void fatal(const char* fmt, ...) {  // log.Fatal
    {
        std::lock_guard<std::mutex> g(g_logmu);
        fprintf(stderr, "level=fatal msg=\"");
        va_list ap;
        va_start(ap, fmt);
        vfprintf(stderr, fmt, ap);
        va_end(ap);
        fprintf(stderr, "\"\n");
    }
    _Exit(1);
}

static bool parse_duration(const std::string& s, double& secs) {  // "5s", "500ms", "1m30s", "2h", plain number = seconds
    secs = 0;
    size_t i = 0;
    bool any = false;
    while (i < s.size()) {
        size_t j = i;
        while (j < s.size() && (isdigit((unsigned char)s[j]) || s[j] == '.')) j++;
        if (j == i) return false;
        double v = atof(s.substr(i, j - i).c_str());
        size_t k = j;
        while (k < s.size() && isalpha((unsigned char)s[k])) k++;
        std::string u = s.substr(j, k - j);
        if (u == "" && k == s.size()) secs += v;
        else if (u == "ns") secs += v * 1e-9;
        else if (u == "us") secs += v * 1e-6;
        else if (u == "ms") secs += v * 1e-3;
        else if (u == "s") secs += v;
        else if (u == "m") secs += v * 60;
        else if (u == "h") secs += v * 3600;
        else return false;
        any = true;
        i = k;
    }
    return any;
}

static void usage_and_exit(const std::string& bad) {
    fprintf(stderr, "flag provided but not defined: -%s\n", bad.c_str());
    exit(2);
}

static Flags parse_flags(int argc, char** argv) {
    Flags f;
    std::map<std::string, std::string*> strs = {
        {"loglevel", &f.LogLevel}, {"metrics.addr", &f.MetricsAddr}, {"metrics.path", &f.MetricsPath},
        {"kafka.version", &f.KafkaVersion}, {"kafka.topic", &f.KafkaTopic}, {"kafka.brokers", &f.KafkaBrk},
        {"kafka.group", &f.KafkaGroup}, {"postgres.user", &f.PostgresUser}, {"postgres.pass", &f.PostgresPass},
        {"postgres.host", &f.PostgresHost}, {"postgres.dbname", &f.PostgresDbName}, {"input.files", &f.InputFiles},
        {"input.format", &f.InputFormat}, {"out.rowbinary", &f.OutRowBinary}, {"out.tsv", &f.OutTsv},
        {"offsets.out", &f.OffsetsOut}, {"metrics.dump", &f.MetricsDump}, {"out.app", &f.OutApp}, {"out.topk", &f.OutTopk},
        {"gpu.transport", &f.GpuTransport}, {"topk.mode", &f.TopkMode}, {"phases.out", &f.PhasesOut}};
    std::map<std::string, long*> ints = {
        {"flush.count", &f.FlushCount}, {"postgres.port", &f.PostgresPort}, {"gpu.devices", &f.GpuDevices},
        {"window.secs", &f.WindowSecs}, {"window.lag", &f.CloseLagSec},
        {"key.sets", &f.KeySets}, {"topk.k", &f.TopkK}, {"gpu.table.log2", &f.TableLog2}, {"gpu.keyset.log2", &f.KeysetLog2},
        {"gpu.wide.log2", &f.WideLog2}, {"gpu.batch.bytes", &f.BatchBytes}, {"topk.track", &f.TopkTrack}};
    std::map<std::string, bool*> bools = {{"proto.fixedlen", &f.ProtoFixed}, {"sink.dryrun", &f.DryRun},
                                          {"window.closeall", &f.CloseAllAtEnd}, {"input.prefault", &f.Prefault}};
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() < 2 || a[0] != '-') fatal("unexpected argument %s", a.c_str());
        a = a.substr(a[1] == '-' ? 2 : 1);
        std::string name = a, val;
        bool has = false;
        size_t eq = a.find('=');
        if (eq != std::string::npos) {
            name = a.substr(0, eq);
            val = a.substr(eq + 1);
            has = true;
        }
        if (bools.count(name)) {
            *bools[name] = !has || val == "true" || val == "1" || val == "t" || val == "T" || val == "TRUE" || val == "True";
            continue;
        }
        if (!has) {
            if (i + 1 >= argc) fatal("flag needs an argument: -%s", name.c_str());
            val = argv[++i];
        }
        if (strs.count(name)) *strs[name] = val;
        else if (ints.count(name)) *ints[name] = atol(val.c_str());
        else if (name == "flush.dur") {
            if (!parse_duration(val, f.FlushTime)) fatal("invalid value \"%s\" for flag -flush.dur", val.c_str());
        } else usage_and_exit(name);
    }
    return f;
}

// ---- the consumer-group surface of the reference (sarama), reduced to what inserter.go uses -------------
struct ConsumerMessage {
    int32_t Partition;
    int64_t Offset;
    const uint8_t* Value;
    size_t Len;
};

struct ConsumerGroupSession {
    std::vector<int32_t> claims;  // sarama: session.Claims()[topic] - the partitions this member of the consumer group owns
    std::mutex mu;
    std::map<int32_t, int64_t> marked;  // partition -> next offset to consume (sarama: offset+1 is committed)
    void MarkMessage(const ConsumerMessage& m, const char* /*metadata*/) {
        std::lock_guard<std::mutex> g(mu);
        int64_t& o = marked[m.Partition];
        if (m.Offset + 1 > o) o = m.Offset + 1;
    }
};

struct MappedLog {  // a read-only view of a partition log file (lives until exit)
    const uint8_t* data = nullptr;
    size_t size = 0;
    const uint8_t& operator[](size_t i) const { return data[i]; }
};

// One claimed partition: a message stream.  next() returns false when the claim is closed.
class ConsumerGroupClaim {
public:
    ConsumerGroupClaim(int32_t part, MappedLog log, bool len32) : part_(part), log_(log), len32_(len32) {}
    int32_t Partition() const { return part_; }
    // more messages buffered on the client side, i.e. next() would not block (sarama: len(claim.Messages()) > 0)
    bool ready() const { return pos_ < log_.size; }
    const uint8_t* peek() const { return pos_ < log_.size ? &log_[pos_] : nullptr; }  // where the next message's value starts
    // The messages that are READY, taken in one go (what a sarama consumer does when it drains len(claim.Messages()) without
    // blocking; librdkafka: consume_batch): as many as fit `max_msgs` / stop once `bytes_so_far` + their bytes reaches
    // `byte_bound`, but only while each one starts where the previous one ended in the client's fetch buffer - the batch then is a
    // byte range, handed to the sink in place.  ends[i] = bytes_so_far + bytes of messages 0..i (the sink's offsets array); `last`
    // = the newest message taken.  Returns the count (0: the next message is not contiguous / not a short frame - take it with
    // next()).  The loop is a dependent chain (a message's length byte tells where the next one starts): one prefetch ahead keeps
    // the lines coming.
    size_t next_run(size_t max_msgs, size_t bytes_so_far, size_t byte_bound, const uint8_t* expect_at, uint64_t* ends, ConsumerMessage& last) {
        if (len32_ || pos_ >= log_.size || (expect_at && expect_at != &log_[pos_])) return 0;
        const uint8_t* const base = log_.data;
        const size_t safe_end = log_.size > 256 ? log_.size - 256 : 0;  // (one-byte prefixes only, whole message inside the log)
        size_t pos = pos_, n = 0, bytes = bytes_so_far, prev = pos_;
        while (n < max_msgs && pos < safe_end && bytes < byte_bound) {
            const uint8_t b = base[pos];
            if (b & 0x80) break;
            __builtin_prefetch(base + pos + 1024);
            prev = pos;
            pos += 1u + (size_t)b;
            bytes += 1u + (size_t)b;
            ends[n++] = bytes;
        }
        if (n) {
            last = ConsumerMessage{part_, off_ + (int64_t)n - 1, base + prev, pos - prev};
            off_ += (int64_t)n;
            pos_ = pos;
        }
        return n;
    }
    bool next(ConsumerMessage& m) {
        if (pos_ >= log_.size) return false;
        size_t start = pos_, len = 0;
        if (!len32_ && log_.size - pos_ > 128 && !(log_[pos_] & 0x80)) {  // a one-byte length prefix, the whole message inside the log
            len = 1 + (size_t)log_[pos_];
            m = ConsumerMessage{part_, off_++, &log_[start], len};
            pos_ = start + len;
            return true;
        }
        if (len32_) {
            if (log_.size - pos_ < 4) fatal("partition %d: truncated length prefix at byte %zu", part_, pos_);
            uint32_t l;
            memcpy(&l, &log_[pos_], 4);
            start = pos_ + 4;
            len = l;
        } else {  // value = varint(len) || payload (mocker.go:98-101): the value includes its prefix
            uint64_t v = 0;
            size_t p = pos_;
            for (int i = 0;; i++) {
                if (i >= 10 || p >= log_.size) fatal("partition %d: bad varint frame at byte %zu", part_, pos_);
                uint8_t b = log_[p++];
                v |= (uint64_t)(b & 0x7f) << (7 * i);
                if (!(b & 0x80)) break;
            }
            len = (p - pos_) + v;
        }
        if (len > log_.size - start) fatal("partition %d: message at byte %zu runs past the end of the log", part_, pos_);
        m = ConsumerMessage{part_, off_++, &log_[start], len};
        pos_ = start + len;
        return true;
    }

private:
    int32_t part_;
    MappedLog log_;
    bool len32_;
    size_t pos_ = 0;
    int64_t off_ = 0;
};

struct ConsumerGroupHandler {
    virtual ~ConsumerGroupHandler() {}
    virtual int Setup(ConsumerGroupSession&) = 0;
    virtual int Cleanup(ConsumerGroupSession&) = 0;
    virtual int ConsumeClaim(ConsumerGroupSession&, ConsumerGroupClaim&) = 0;
};

// ---- the sink -----------------------------------------------------------------------------------------------
static std::atomic<uint64_t> Inserts{0};  // insert_count (inserter.go:44-49)
static std::atomic<uint64_t> RowsOut{0}, Flushes{0};

class RowWriter {  // flows_5m rows of closed windows: RowBinary for clickhouse-client / HTTP, TSV for humans
public:
    void open(const Flags& f) {
        if (!f.OutRowBinary.empty() && !(rb_ = fopen(f.OutRowBinary.c_str(), "wb"))) fatal("cannot open %s", f.OutRowBinary.c_str());
        if (!f.OutTsv.empty() && !(tsv_ = fopen(f.OutTsv.c_str(), "w"))) fatal("cannot open %s", f.OutTsv.c_str());
        if (!f.OutApp.empty() && !(app_ = fopen(f.OutApp.c_str(), "wb"))) fatal("cannot open %s", f.OutApp.c_str());
        if (!f.OutTopk.empty() && !(topk_ = fopen(f.OutTopk.c_str(), "w"))) fatal("cannot open %s", f.OutTopk.c_str());
    }
    bool wantsApp() const { return app_ != nullptr; }
    bool wantsTopk() const { return topk_ != nullptr; }
    void writeApp(const fa_row_app* rows, size_t n) {  // (SrcAddr,DstPort,Proto) rows of a closed window, as they are
        if (!n || !app_) return;
        std::lock_guard<std::mutex> g(mu_);
        if (fwrite(rows, sizeof(fa_row_app), n, app_) != n) fatal("short write on the app-rows sink");
        RowsOut += n;
    }
    void writeTopk(const char* which, const std::vector<fa_topk_row>& rows) {
        if (!topk_) return;
        std::lock_guard<std::mutex> g(mu_);
        for (auto& r : rows) {
            fprintf(topk_, "%s\t", which);
            for (int i = 0; i < 16; i++) fprintf(topk_, "%02x", r.key[i]);
            fprintf(topk_, "\t%llu\n", (unsigned long long)r.weight);
        }
    }
    void write(const fa_row5m* rows, size_t nrows) {
        if (!nrows) return;
        std::lock_guard<std::mutex> g(mu_);
        if (rb_) {
            rbuf_.resize(nrows * FA_ROWBINARY_ROW5M_BYTES);
            size_t n = 0;
            if (fa_rows_to_rowbinary(rows, nrows, rbuf_.data(), rbuf_.size(), &n) != 0) fatal("fa_rows_to_rowbinary failed");
            if (fwrite(rbuf_.data(), 1, n, rb_) != n) fatal("short write on the RowBinary sink");
        }
        if (tsv_)
            for (size_t i = 0; i < nrows; i++) {
                const fa_row5m& r = rows[i];
                fprintf(tsv_, "%u\t%u\t%u\t%u\t%u\t%llu\t%llu\t%llu\n", r.date, r.timeslot, r.src_as, r.dst_as, r.etype,
                        (unsigned long long)r.bytes, (unsigned long long)r.packets, (unsigned long long)r.count);
            }
        RowsOut += nrows;
    }
    void close() {
        if (rb_) fclose(rb_);
        if (tsv_) fclose(tsv_);
        if (app_) fclose(app_);
        if (topk_) fclose(topk_);
        rb_ = tsv_ = app_ = topk_ = nullptr;
    }

private:
    std::mutex mu_;
    std::vector<uint8_t> rbuf_;
    FILE* rb_ = nullptr;
    FILE* tsv_ = nullptr;
    FILE* app_ = nullptr;
    FILE* topk_ = nullptr;
};

// one aggregation context per claimed partition (fa_ctx is not thread-safe; distinct ctxs are independent)
struct Phases {  // seconds of one partition thread, by what it was doing
    double take = 0;     // claim.next() + batch bookkeeping (offsets, the zero-copy run or the copy into the batch buffer)
    double ingest = 0;   // inside fa_ingest: wait for a staging slot, copy into pinned memory, enqueue H2D + kernels
    double lock = 0;     // waiting for the read lock (a window close holds the write lock)
    double mark = 0;     // session.MarkMessage + counters
    double close = 0;    // closeWindows called from this thread's timer
    uint64_t batches = 0, records = 0, bytes = 0, copied_bytes = 0;
    // of fa_ingest, by the library's account (fa_stats_t, ABI 8): waiting for a free staging slot (the transfer / kernels of the
    // call before the last still hold it: the GPU / PCIe side is behind), the copy into page-locked staging; the rest of the
    // call is enqueueing (H2D, kernels, events) and the launch bookkeeping.  device_path: hipEvent time of the launches.
    double lib_ingest = 0, lib_wait = 0, lib_copy = 0, device_path = 0;
    uint64_t launches = 0;
};
struct PartitionState {
    fa_ctx* ctx = nullptr;
    // the pending batch: message values back to back - in place in the client's fetch buffer while every message follows the
    // previous one there (run != nullptr), copied into buf from the first one that does not
    const uint8_t* run = nullptr;
    size_t run_len = 0;
    std::vector<uint8_t> buf;
    // offsets[0 .. pending]: where message i starts in the batch, offsets[pending] = its bytes (a plain array: the bulk path
    // writes into it directly, and a vector would zero what it hands out)
    std::unique_ptr<uint64_t[]> offsets;
    size_t offsets_cap = 0;
    size_t pending = 0;             // messages in the batch
    ConsumerMessage last{};         // the newest of them: offsets of one partition are monotone, marking it commits the batch
    Phases ph;
    size_t bytes() const { return run ? run_len : buf.size(); }
    bool empty() const { return pending == 0; }
    // in place so far (or empty): the bulk path may go on; -> where the next message has to start (nullptr: anywhere)
    bool in_place() const { return pending == 0 || run != nullptr; }
    const uint8_t* run_end() const { return pending == 0 ? nullptr : run + run_len; }
    uint64_t* offsets_room(size_t more) {  // room for `more` offsets behind the batch; -> where they go
        if (pending + 1 + more > offsets_cap) {
            const size_t cap = std::max<size_t>(2 * offsets_cap, pending + 1 + more + 4096);
            std::unique_ptr<uint64_t[]> grown(new uint64_t[cap]);
            if (offsets_cap) memcpy(grown.get(), offsets.get(), (pending + 1) * sizeof(uint64_t));
            else grown[0] = 0;
            offsets = std::move(grown);
            offsets_cap = cap;
        }
        return offsets.get() + pending + 1;
    }
    // `n` contiguous messages, the first at `first`, the newest `lastm`, were written behind the batch by
    // ConsumerGroupClaim::next_run (their ends are in the offsets already)
    void took_run(size_t n, const uint8_t* first, const ConsumerMessage& lastm) {
        if (!n) return;
        if (pending == 0) {
            run = first;
            buf.clear();
        }
        pending += n;
        run_len = (size_t)offsets[pending];
        last = lastm;
    }
    void append(const ConsumerMessage& m) {
        if (pending == 0) {
            run = m.Value;
            run_len = m.Len;
            buf.clear();
        } else if (run && m.Value == run + run_len) {
            run_len += m.Len;
        } else {
            if (run) {
                buf.assign(run, run + run_len);
                ph.copied_bytes += run_len;
                run = nullptr;
                run_len = 0;
            }
            buf.insert(buf.end(), m.Value, m.Value + m.Len);
            ph.copied_bytes += m.Len;
        }
        *offsets_room(1) = bytes();
        last = m;
        pending++;
    }
    void reset() {
        run = nullptr;
        run_len = 0;
        buf.clear();
        pending = 0;
    }
};
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

class State : public ConsumerGroupHandler {
public:
    State(const Flags& f, RowWriter& w) : f_(f), out_(w) {}

    // Setup: the session knows its claims (sarama: session.Claims()) - one ctx per claimed partition, partition p on GPU
    // p % -gpu.devices (north star: 8 partitions, one per GPU), and ONE group over them for the window close
    int Setup(ConsumerGroupSession& session) override {
        std::vector<fa_ctx*> ctxs;
        for (int32_t part : session.claims) {
            auto p = std::make_unique<PartitionState>();
            (void)p->offsets_room(1);  // (offsets[0] = 0)
            if (!f_.DryRun) {
                fa_config cfg;
                memset(&cfg, 0, sizeof cfg);
                cfg.device = (int32_t)(part % (f_.GpuDevices > 0 ? f_.GpuDevices : 1));
                cfg.window_secs = (uint32_t)f_.WindowSecs;
                cfg.subwindow_secs = 0;  // the sink stores tumbling windows (what flows_5m holds, create.sh:96)
                cfg.key_sets = (uint32_t)f_.KeySets;
                cfg.table_capacity_log2 = (uint32_t)f_.TableLog2;
                cfg.topk_capacity_log2 = (uint32_t)f_.KeysetLog2;
                cfg.wide_capacity_log2 = (uint32_t)f_.WideLog2;
                // candidates: the sets hold what can rank (2^16 slots unless -gpu.keyset.log2 says otherwise), not every address
                const bool cand = f_.TopkMode == "candidates" || (f_.TopkMode == "auto" && session.claims.size() > 1);
                cfg.topk_mode = cand ? FA_TOPK_CANDIDATES : FA_TOPK_EXACT;
                cfg.topk_track = (uint32_t)f_.TopkTrack;
                if (cand && !f_.KeysetLog2) cfg.topk_capacity_log2 = 16;
                cfg.framed = f_.ProtoFixed ? 1 : 0;
                int rc = fa_create(&cfg, &p->ctx);
                if (rc != 0) fatal("fa_create: %d %s", rc, fa_last_error(nullptr));
                // the staging a batch of -gpu.batch.bytes needs is page-locked here, not inside the consume loop
                if (f_.BatchBytes > 0) {
                    const size_t bytes = (size_t)f_.BatchBytes + (64u << 10), recs = std::min<size_t>(bytes / 48, (size_t)1 << 24);
                    if ((rc = fa_reserve_ingest(p->ctx, bytes, recs)) != 0) fatal("fa_reserve_ingest: %d %s", rc, fa_last_error(p->ctx));
                }
                ctxs.push_back(p->ctx);
            }
            if (f_.BatchBytes > 0) {  // the batch's offsets: allocated and touched here, not page by page inside the consume loop
                const size_t recs = std::min<size_t>(((size_t)f_.BatchBytes + (64u << 10)) / 48, (size_t)1 << 24);
                memset(p->offsets_room(recs + 1), 0, (recs + 1) * sizeof(uint64_t));
            }
            parts_[part] = std::move(p);
        }
        if (!ctxs.empty()) {
            const uint32_t flags = f_.GpuTransport == "rccl" ? FA_GROUP_RCCL : FA_GROUP_PEER;
            int rc = fa_group_create(ctxs.data(), ctxs.size(), flags, &group_);
            if (rc != 0) fatal("fa_group_create: %d %s", rc, fa_group_last_error(nullptr));
            logf(2, "window close: group of %zu context(s) over %ld GPU(s), transport %s, top-k mode %s", ctxs.size(), f_.GpuDevices,
                 fa_group_transport(group_) == FA_GROUP_RCCL ? "rccl" : "peer copies",
                 f_.TopkMode == "candidates" || (f_.TopkMode == "auto" && session.claims.size() > 1) ? "candidates" : "exact");
        }
        return 0;
    }
    int Cleanup(ConsumerGroupSession&) override { return 0; }

    // ConsumeClaim: the consumer group runs one of these per claimed partition, concurrently (inserter.go:176)
    int ConsumeClaim(ConsumerGroupSession& session, ConsumerGroupClaim& claim) override {
        PartitionState* p = partition(claim.Partition());
        auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(f_.FlushTime);
        const size_t flush_count = (size_t)f_.FlushCount, batch_bytes = (size_t)std::max(0l, f_.BatchBytes);
        ConsumerMessage m;
        double t_mark = now_s();
        auto lap = [&](double& acc) {  // the time since the last lap goes to `acc`
            const double t = now_s();
            acc += t - t_mark;
            t_mark = t;
        };
        for (;;) {
            // the bulk path: every message that is ready and lies right behind the batch in the client's buffer, up to the next
            // decision point - the -flush.count trip, beyond it the byte bound, the clock every 4096 messages
            if (p->in_place()) {
                const bool below = p->pending < flush_count;
                const size_t want = below ? std::min<size_t>(4096, flush_count - p->pending) : 4096;
                const uint8_t* first = claim.peek();
                const size_t got = claim.next_run(want, p->bytes(), below ? (size_t)-1 : std::max<size_t>(batch_bytes, 1), p->run_end(), p->offsets_room(want), m);
                if (got) {
                    p->took_run(got, first, m);
                    if (p->pending >= flush_count && (p->bytes() >= batch_bytes || !claim.ready())) {
                        lap(p->ph.take);
                        flush(*p, session);
                        t_mark = now_s();
                    }
                    if (std::chrono::steady_clock::now() >= deadline) {  // inserter.go:189-191
                        lap(p->ph.take);
                        flush(*p, session);
                        t_mark = now_s();
                        closeWindows((int64_t)time(nullptr), false);
                        lap(p->ph.close);
                        deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(f_.FlushTime);
                    }
                    continue;
                }
            }
            if (!claim.next(m)) break;
            p->append(m);
            // inserter.go:118-120 - and, beyond the reference, a batch keeps growing while the claim has messages ready (header)
            if (p->pending >= flush_count && (p->bytes() >= batch_bytes || !claim.ready())) {
                lap(p->ph.take);
                flush(*p, session);
                t_mark = now_s();
            }
            if ((p->pending & 15) == 0 && std::chrono::steady_clock::now() >= deadline) {  // inserter.go:189-191 (the clock is read every 16 messages)
                lap(p->ph.take);
                flush(*p, session);
                t_mark = now_s();
                closeWindows((int64_t)time(nullptr), false);
                lap(p->ph.close);
                deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(f_.FlushTime);
            }
        }
        lap(p->ph.take);
        flush(*p, session);  // claim closed
        return 0;
    }

    // flush = inserter.go:90-111 with the per-row db.Exec loop replaced by one fa_ingest
    void flush(PartitionState& p, ConsumerGroupSession& session) {
        const size_t n = p.pending;
        if (n == 0) return;
        const uint8_t* data = p.run ? p.run : p.buf.data();
        const size_t len = p.bytes();
        logf(3, "Processed %zu records in the last iteration.", n);
        double t0 = now_s();
        if (f_.DryRun) {
            logf(2, "dryrun flush partition=%d records=%zu bytes=%zu", p.last.Partition, n, len);
        } else {
            std::shared_lock<std::shared_mutex> rd(close_mu_);  // (not while the group closes a window: that uses every ctx)
            const double t1 = now_s();
            p.ph.lock += t1 - t0;
            // fa_ingest copies into library-owned pinned memory before returning
            int rc = fa_ingest(p.ctx, data, len, p.offsets.get(), n);
            if (rc != 0) fatal("fa_ingest: %d %s", rc, fa_last_error(p.ctx));  // sink error is fatal, inserter.go:102-105
            t0 = now_s();
            p.ph.ingest += t0 - t1;
        }
        Inserts += n;
        Flushes += 1;
        session.MarkMessage(p.last, "");  // after the sink accepted the batch; one mark per batch (the reference marks every message,
                                          // inserter.go:188 - same committed offset, without 8 threads meeting on the session's lock per message)
        p.ph.batches += 1;
        p.ph.records += n;
        p.ph.bytes += len;
        p.reset();
        p.ph.mark += now_s() - t0;
    }

    // emits the finished windows of the WHOLE topic to the bulk-load sinks: flows_5m rows (create.sh:70-90) merged over the
    // partitions (one row per key - what flows_5m holds after its SummingMergeTree has merged the per-partition inserts, and
    // an eighth of the rows to insert), (SrcAddr,DstPort,Proto) rows hash-partitioned over the GPUs (-out.app)
    void closeWindows(int64_t now, bool all) {
        if (f_.DryRun || !group_) return;
        std::unique_lock<std::shared_mutex> wr(close_mu_);
        std::vector<uint32_t> slots(64);
        size_t ns = 0;
        int rc = fa_group_open_timeslots(group_, slots.data(), slots.size(), &ns);
        if (rc == FA_ERR_CAPACITY) {
            slots.resize(ns);
            rc = fa_group_open_timeslots(group_, slots.data(), slots.size(), &ns);
        }
        if (rc != 0) fatal("fa_group_open_timeslots: %d %s", rc, fa_group_last_error(group_));
        const uint32_t gran = (uint32_t)f_.WindowSecs;
        for (size_t i = 0; i < ns; i++) {
            const uint32_t ts = slots[i];
            if (!all && (int64_t)ts + gran + f_.CloseLagSec > now) continue;
            if (out_.wantsApp() && ((uint32_t)f_.KeySets & FA_KEYS_ADDR_PORT_PROTO)) {
                // (the row buffer is the session's: a buffer that is too small costs the close twice - the library knows the size it
                // needs only after collect + exchange + merge - so it keeps the largest window seen, with room to spare)
                std::vector<fa_row_app>& app = app_rows_;
                if (app.size() < (1u << 16)) app.resize(1u << 16);
                size_t na = 0;
                rc = fa_group_close_window_partitioned(group_, FA_ROWS_APP, ts, app.data(), app.size(), nullptr, &na);
                if (rc == FA_ERR_CAPACITY) {
                    app.resize(na + na / 4);
                    rc = fa_group_close_window_partitioned(group_, FA_ROWS_APP, ts, app.data(), app.size(), nullptr, &na);
                }
                if (rc != 0) fatal("fa_group_close_window_partitioned: %d %s", rc, fa_group_last_error(group_));
                logf(2, "(SrcAddr,DstPort,Proto) timeslot %u: %zu rows", ts, na);
                out_.writeApp(app.data(), na);
            } else if ((uint32_t)f_.KeySets & FA_KEYS_ADDR_PORT_PROTO) {
                // the key set is on but nobody takes its rows: the window is closed all the same (dropped on every member) -
                // kept, the wide table / wide log would grow for the life of the session
                for (auto& kv : parts_)
                    if (kv.second->ctx && (rc = fa_drop_window(kv.second->ctx, FA_ROWS_APP, ts)) != 0)
                        fatal("fa_drop_window(FA_ROWS_APP): %d %s", rc, fa_last_error(kv.second->ctx));
            }
            std::vector<fa_row5m>& rows = rows5m_;
            if (rows.size() < (1u << 16)) rows.resize(1u << 16);
            size_t nr = 0;
            rc = fa_group_close_window(group_, FA_ROWS_5M, ts, rows.data(), rows.size(), &nr);
            if (rc == FA_ERR_CAPACITY) {
                rows.resize(nr + nr / 4);
                rc = fa_group_close_window(group_, FA_ROWS_5M, ts, rows.data(), rows.size(), &nr);
            }
            if (rc != 0) fatal("fa_group_close_window: %d %s", rc, fa_group_last_error(group_));
            logf(2, "flows_5m timeslot %u: %zu rows", ts, nr);
            out_.write(rows.data(), nr);
        }
    }

    void finish(ConsumerGroupSession&) {
        if (f_.CloseAllAtEnd) closeWindows(0, true);
        std::lock_guard<std::mutex> g(mu_);
        if (group_ && !f_.DryRun) {
            // the heavy hitters of the whole topic (viz-ch.json:233,479): sketches all-reduced over the GPUs, top k of the merged sketch
            if (out_.wantsTopk())
                for (uint32_t ks : {(uint32_t)FA_KEYS_SRCADDR_CMS, (uint32_t)FA_KEYS_DSTADDR_CMS}) {
                    if (!((uint32_t)f_.KeySets & ks)) continue;
                    std::vector<fa_topk_row> top((size_t)std::max(1L, f_.TopkK));
                    size_t nt = 0;
                    int rc = fa_group_topk(group_, ks, top.size(), top.data(), top.size(), &nt);
                    if (rc != 0) fatal("fa_group_topk: %d %s", rc, fa_group_last_error(group_));
                    top.resize(nt);
                    out_.writeTopk(ks == FA_KEYS_SRCADDR_CMS ? "src" : "dst", top);
                }
            for (auto& kv : parts_) {  // the library's own account of its fa_ingest calls (ABI 8)
                fa_stats_t ps;
                if (kv.second->ctx && fa_stats(kv.second->ctx, &ps) == 0) {
                    kv.second->ph.lib_ingest = ps.host_ingest_ns * 1e-9;
                    kv.second->ph.lib_wait = ps.host_stage_wait_ns * 1e-9;
                    kv.second->ph.lib_copy = ps.host_stage_copy_ns * 1e-9;
                    kv.second->ph.launches = ps.kernel_launches;
                    kv.second->ph.device_path = ps.batch_ns_total * 1e-9;
                }
            }
            fa_stats_t st;
            if (fa_group_stats(group_, &st) == 0) {
                logf(2, "topic: records_ok=%llu records_bad=%llu", (unsigned long long)st.records_ok, (unsigned long long)st.records_bad);
                bad_ += st.records_bad;
            }
            fa_group_destroy(group_);
            group_ = nullptr;
        }
        for (auto& kv : parts_)
            if (kv.second->ctx) fa_destroy(kv.second->ctx);
    }
    uint64_t bad() const { return bad_; }
    // where the consume loop's time went, per partition thread (-phases.out; the "phases:" log line carries the sums)
    std::string phasesJson(double setup_s, double consume_s, double finish_s) {
        std::string js = "{\"setup_s\": " + std::to_string(setup_s) + ", \"consume_s\": " + std::to_string(consume_s) + ", \"finish_s\": " + std::to_string(finish_s) +
                         ", \"partitions\": [";
        bool first = true;
        for (auto& kv : parts_) {
            const Phases& h = kv.second->ph;
            char b[1024];
            snprintf(b, sizeof b, "%s{\"partition\": %d, \"take_s\": %.6f, \"fa_ingest_s\": %.6f, \"lock_wait_s\": %.6f, \"mark_s\": %.6f, \"close_s\": %.6f, "
                     "\"batches\": %llu, \"records\": %llu, \"bytes\": %llu, \"copied_bytes\": %llu, \"lib_ingest_s\": %.6f, \"lib_stage_wait_s\": %.6f, "
                     "\"lib_stage_copy_s\": %.6f, \"launches\": %llu, \"device_path_s\": %.6f}", first ? "" : ", ", kv.first, h.take, h.ingest, h.lock, h.mark, h.close,
                     (unsigned long long)h.batches, (unsigned long long)h.records, (unsigned long long)h.bytes, (unsigned long long)h.copied_bytes, h.lib_ingest, h.lib_wait,
                     h.lib_copy, (unsigned long long)h.launches, h.device_path);
            js += b;
            first = false;
        }
        return js + "]}";
    }
    Phases phasesSum() {
        Phases t;
        for (auto& kv : parts_) {
            const Phases& h = kv.second->ph;
            t.take += h.take, t.ingest += h.ingest, t.lock += h.lock, t.mark += h.mark, t.close += h.close;
            t.batches += h.batches, t.records += h.records, t.bytes += h.bytes, t.copied_bytes += h.copied_bytes;
            t.lib_ingest += h.lib_ingest, t.lib_wait += h.lib_wait, t.lib_copy += h.lib_copy, t.device_path += h.device_path, t.launches += h.launches;
        }
        return t;
    }

private:
    PartitionState* partition(int32_t part) {
        std::lock_guard<std::mutex> g(mu_);
        auto it = parts_.find(part);
        if (it == parts_.end()) fatal("partition %d was not claimed in Setup", part);
        return it->second.get();
    }

    const Flags& f_;
    RowWriter& out_;
    std::mutex mu_;
    std::shared_mutex close_mu_;  // flushes: shared; a window close (every member ctx of the group): exclusive
    std::map<int32_t, std::unique_ptr<PartitionState>> parts_;
    fa_group* group_ = nullptr;
    uint64_t bad_ = 0;
    std::vector<fa_row_app> app_rows_;  // row buffers of the window close (under close_mu_): kept between closes
    std::vector<fa_row5m> rows5m_;
};

// a partition log, mapped (the page cache is the only copy; a Kafka client would hand out its fetch buffers the same way)
static MappedLog map_file(const std::string& path, bool prefault) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) fatal("cannot open %s", path.c_str());
    struct stat st;
    if (fstat(fd, &st) != 0) fatal("cannot stat %s", path.c_str());
    MappedLog m;
    m.size = (size_t)st.st_size;
    if (m.size) {
        void* p = mmap(nullptr, m.size, PROT_READ, MAP_PRIVATE | (prefault ? MAP_POPULATE : 0), fd, 0);
        if (p == MAP_FAILED) fatal("cannot map %s", path.c_str());
        madvise(p, m.size, MADV_SEQUENTIAL);
        m.data = (const uint8_t*)p;
    }
    close(fd);
    return m;
}

int main(int argc, char** argv) {
    Flags f = parse_flags(argc, argv);
    g_loglevel = f.LogLevel == "debug" ? 3 : f.LogLevel == "info" ? 2 : f.LogLevel == "warning" || f.LogLevel == "warn" ? 1 : 0;
    if (f.InputFiles.empty())
        fatal("no Kafka client in this build: pass -input.files=p0.log,p1.log (one partition log per file, topic %s)", f.KafkaTopic.c_str());
    if (f.InputFormat != "framed" && f.InputFormat != "len32") fatal("-input.format must be framed or len32");
    if (f.InputFormat == "framed" && !f.ProtoFixed) fatal("-input.format=framed needs -proto.fixedlen=true (bare values are not self-delimiting)");
    if (f.FlushCount < 1) fatal("-flush.count must be >= 1");
    if (f.GpuTransport != "peer" && f.GpuTransport != "rccl") fatal("-gpu.transport must be peer or rccl");
    if (f.TopkMode != "auto" && f.TopkMode != "exact" && f.TopkMode != "candidates") fatal("-topk.mode must be exact, candidates or auto");

    RowWriter out;
    out.open(f);
    State s(f, out);
    ConsumerGroupSession session;
    const auto t_start = std::chrono::steady_clock::now();  // (setup = mapping the partition logs + contexts + group)

    std::vector<std::unique_ptr<ConsumerGroupClaim>> claims;
    int32_t part = 0;
    size_t i = 0;
    while (i <= f.InputFiles.size()) {
        size_t j = f.InputFiles.find(',', i);
        if (j == std::string::npos) j = f.InputFiles.size();
        if (j > i) claims.push_back(std::make_unique<ConsumerGroupClaim>(part++, map_file(f.InputFiles.substr(i, j - i), f.Prefault), f.InputFormat == "len32"));
        i = j + 1;
    }
    for (auto& c : claims) session.claims.push_back(c->Partition());
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    if (s.Setup(session) != 0) fatal("Setup failed");
    const double setup_s = since(t_start);
    logf(2, "inserter-gpu up and running: %zu partition(s), flush.count=%ld flush.dur=%gs", claims.size(), f.FlushCount, f.FlushTime);
    std::vector<std::thread> workers;
    for (auto& c : claims) workers.emplace_back([&s, &session, &c] { s.ConsumeClaim(session, *c); });
    const auto t_consume = std::chrono::steady_clock::now();
    for (auto& t : workers) t.join();
    const double consume_s = since(t_consume);
    const auto t_finish = std::chrono::steady_clock::now();
    s.finish(session);
    if (s.Cleanup(session) != 0) fatal("Cleanup failed");
    out.close();
    const double finish_s = since(t_finish);
    logf(2, "phases: setup %.3f s, consume %.3f s, last close + top-k + teardown %.3f s", setup_s, consume_s, finish_s);
    {
        const Phases t = s.phasesSum();
        const double nthr = claims.empty() ? 1.0 : (double)claims.size();
        logf(2, "consume loop, mean per partition thread: take %.3f s, fa_ingest %.3f s, lock wait %.3f s, mark %.3f s, timer closes %.3f s; %llu batches, %.1f MiB per batch, "
                "%.1f %% of the bytes copied into a batch buffer (the rest handed over in place); of fa_ingest: wait for a staging slot %.3f s, copy into "
                "page-locked staging %.3f s; device path %.3f s in %llu launches",
             t.take / nthr, t.ingest / nthr, t.lock / nthr, t.mark / nthr, t.close / nthr, (unsigned long long)t.batches,
             t.batches ? (double)t.bytes / (double)t.batches / 1048576.0 : 0.0, t.bytes ? 100.0 * (double)t.copied_bytes / (double)t.bytes : 0.0,
             t.lib_wait / nthr, t.lib_copy / nthr, t.device_path / nthr, (unsigned long long)t.launches);
    }
    if (!f.PhasesOut.empty()) {
        FILE* fp = fopen(f.PhasesOut.c_str(), "w");
        if (!fp) fatal("cannot open %s", f.PhasesOut.c_str());
        fprintf(fp, "%s\n", s.phasesJson(setup_s, consume_s, finish_s).c_str());
        fclose(fp);
    }

    if (!f.OffsetsOut.empty()) {
        FILE* fp = fopen(f.OffsetsOut.c_str(), "w");
        if (!fp) fatal("cannot open %s", f.OffsetsOut.c_str());
        for (auto& kv : session.marked) fprintf(fp, "%d\t%lld\n", kv.first, (long long)kv.second);
        fclose(fp);
    }
    if (!f.MetricsDump.empty()) {
        FILE* fp = fopen(f.MetricsDump.c_str(), "w");
        if (!fp) fatal("cannot open %s", f.MetricsDump.c_str());
        fprintf(fp, "# HELP insert_count Inserts made to Postgres.\n# TYPE insert_count counter\ninsert_count %llu\n",
                (unsigned long long)Inserts.load());
        fprintf(fp, "# TYPE flowagg_flushes counter\nflowagg_flushes %llu\n", (unsigned long long)Flushes.load());
        fprintf(fp, "# TYPE flowagg_rows_out counter\nflowagg_rows_out %llu\n", (unsigned long long)RowsOut.load());
        fprintf(fp, "# TYPE flowagg_records_bad counter\nflowagg_records_bad %llu\n", (unsigned long long)s.bad());
        fclose(fp);
    }
    logf(2, "done: insert_count=%llu flushes=%llu rows_out=%llu", (unsigned long long)Inserts.load(), (unsigned long long)Flushes.load(),
         (unsigned long long)RowsOut.load());
    return 0;
}
