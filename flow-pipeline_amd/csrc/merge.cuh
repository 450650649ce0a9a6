// merge.cuh - window close on the device: row sets (flows_5m rows, (SrcAddr,DstPort,Proto) rows, port / minute rows,
// top-k candidates) are sorted, rows with equal keys summed and the result put into the order the read side emits -
// all in HBM.  One code path serves a single ctx's close (the sub-buckets of a sliding window fold here) and the merge
// of several ranks' row sets gathered over RCCL (SummingMergeTree collapse, compose/clickhouse/create.sh:70-90;
// read-side orders: README.md:164-184, compose/grafana/dashboards/viz-ch.json:74,233,358,479,604).
//
// Shape: LSD radix sort over the key words of a row kind (hipcub SortPairs of {key word, row index}; the next word is
// gathered through the permutation between passes), head flags + exclusive scan, one thread per run sums its rows
// (runs are as long as there are ranks x sub-buckets: short), optional second sort into the emit order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "maintenance.cuh"
#include "rowplan.cuh"

namespace fa {

enum { RK_5M = 0, RK_APP = 1, RK_PORT_SRC = 2, RK_PORT_DST = 3, RK_MINUTE = 4, RK_TOPK_SRC = 5, RK_TOPK_DST = 6, RK_COUNT = 7 };

struct RowApp {  // == fa_row_app
    uint32_t date, timeslot;
    uint32_t addr[4];
    uint32_t dst_port, proto;
    unsigned long long bytes, packets, count;
};
static_assert(sizeof(RowApp) == 56, "fa_row_app layout");
struct RowApp48 {  // == fa_row_app48: a window's rows without the (date, timeslot) they share
    uint32_t addr[4];
    uint32_t dst_port, proto;
    unsigned long long bytes, packets, count;
};
static_assert(sizeof(RowApp48) == 48, "fa_row_app48 layout");
struct RowW {  // == fa_port_row == fa_minute_row
    uint32_t key, pad;
    unsigned long long weight, count;
};
static_assert(sizeof(RowW) == 24, "fa_port_row layout");
static_assert(sizeof(TopkRow) == 24, "fa_topk_row layout");

// memcmp order of 8 key bytes held as a little-endian u64
__host__ __device__ __forceinline__ unsigned long long bytes_order(unsigned long long le) {
    return __builtin_bswap64(le);
}

// Per row kind: the words of the MERGE order (equal words <=> same group; least significant word first), how two rows
// of one group combine, and - where the read side wants another order - the words of the EMIT order.
// fold = the window start every row is assigned to (sliding windows: the sub-buckets of a group become one row), ~0: none.
template <int KIND>
struct RowOps;

template <>
struct RowOps<RK_5M> {
    typedef Row5m Row;
    static constexpr int NK = 2, NK2 = 0;
    __device__ static unsigned long long key(const Row& r, int w, uint32_t fold) {
        return w == 0 ? ((unsigned long long)r.dst_as << 32 | r.etype) : ((unsigned long long)(fold != 0xffffffffu ? fold : r.timeslot) << 32 | r.src_as);
    }
    __host__ __device__ static int bits(int) { return 64; }
    __device__ static unsigned long long key2(const Row&, int) { return 0; }
    __device__ static bool same(const Row& a, const Row& b, uint32_t fold) {
        return a.src_as == b.src_as && a.dst_as == b.dst_as && a.etype == b.etype && (fold != 0xffffffffu || a.timeslot == b.timeslot);
    }
    __device__ static void add(Row& a, const Row& b) {
        a.bytes += b.bytes;
        a.packets += b.packets;
        a.count += b.count;
    }
    __device__ static void add_atomic(Row& a, const Row& b) {
        atomicAdd(&a.bytes, b.bytes);
        atomicAdd(&a.packets, b.packets);
        atomicAdd(&a.count, b.count);
    }
    __device__ static void finish(Row& r, uint32_t fold) {
        if (fold != 0xffffffffu) r.timeslot = fold;
        r.date = r.timeslot / 86400u;
        r.pad = 0;
    }
};

template <>
struct RowOps<RK_APP> {
    typedef RowApp Row;
    static constexpr int NK = 4, NK2 = 0;
    // order: date, timeslot, src_addr (bytes), dst_port, proto (fa_read_window_app)
    __device__ static unsigned long long key(const Row& r, int w, uint32_t fold) {
        switch (w) {
        case 0: return (unsigned long long)r.dst_port << 32 | r.proto;
        case 1: return bytes_order((unsigned long long)r.addr[3] << 32 | r.addr[2]);
        case 2: return bytes_order((unsigned long long)r.addr[1] << 32 | r.addr[0]);
        default: return fold != 0xffffffffu ? fold : r.timeslot;
        }
    }
    __host__ __device__ static int bits(int w) { return w == 3 ? 32 : 64; }
    __device__ static unsigned long long key2(const Row&, int) { return 0; }
    __device__ static bool same(const Row& a, const Row& b, uint32_t fold) {
        return a.addr[0] == b.addr[0] && a.addr[1] == b.addr[1] && a.addr[2] == b.addr[2] && a.addr[3] == b.addr[3] &&
               a.dst_port == b.dst_port && a.proto == b.proto && (fold != 0xffffffffu || a.timeslot == b.timeslot);
    }
    __device__ static void add(Row& a, const Row& b) {
        a.bytes += b.bytes;
        a.packets += b.packets;
        a.count += b.count;
    }
    __device__ static void add_atomic(Row& a, const Row& b) {
        atomicAdd(&a.bytes, b.bytes);
        atomicAdd(&a.packets, b.packets);
        atomicAdd(&a.count, b.count);
    }
    __device__ static void finish(Row& r, uint32_t fold) {
        if (fold != 0xffffffffu) r.timeslot = fold;
        r.date = r.timeslot / 86400u;
    }
};

// GROUP BY port ORDER BY sum(Bytes*SamplingRate) DESC, port (viz-ch.json:358,604) / GROUP BY minute ORDER BY minute (:74)
template <bool BY_WEIGHT>
struct RowOpsW {
    typedef RowW Row;
    static constexpr int NK = 1, NK2 = BY_WEIGHT ? 2 : 0;
    __device__ static unsigned long long key(const Row& r, int, uint32_t) { return r.key; }
    __host__ __device__ static int bits(int) { return 32; }
    __device__ static unsigned long long key2(const Row& r, int w) { return w == 0 ? (unsigned long long)r.key : ~r.weight; }
    __host__ __device__ static int bits2(int w) { return w == 0 ? 32 : 64; }
    __device__ static bool same(const Row& a, const Row& b, uint32_t) { return a.key == b.key; }
    __device__ static void add(Row& a, const Row& b) {
        a.weight += b.weight;
        a.count += b.count;
    }
    __device__ static void add_atomic(Row& a, const Row& b) {
        atomicAdd(&a.weight, b.weight);
        atomicAdd(&a.count, b.count);
    }
    __device__ static void finish(Row& r, uint32_t) { r.pad = 0; }
};
template <>
struct RowOps<RK_PORT_SRC> : RowOpsW<true> {};
template <>
struct RowOps<RK_PORT_DST> : RowOpsW<true> {};
template <>
struct RowOps<RK_MINUTE> : RowOpsW<false> {};

// heavy hitters: one row per address (a key stored twice in the distinct set, or reported by several ranks, carries the
// same estimate everywhere: the first one stands), ORDER BY estimate DESC, address bytes (fa_topk)
struct RowOpsTopk {
    typedef TopkRow Row;
    static constexpr int NK = 2, NK2 = 3;
    __device__ static unsigned long long key(const Row& r, int w, uint32_t) { return w == 0 ? bytes_order(r.hi) : bytes_order(r.lo); }
    __host__ __device__ static int bits(int) { return 64; }
    __device__ static unsigned long long key2(const Row& r, int w) { return w == 0 ? bytes_order(r.hi) : w == 1 ? bytes_order(r.lo) : ~r.weight; }
    __host__ __device__ static int bits2(int) { return 64; }
    __device__ static bool same(const Row& a, const Row& b, uint32_t) { return a.lo == b.lo && a.hi == b.hi; }
    __device__ static void add(Row& a, const Row& b) { a.weight = a.weight > b.weight ? a.weight : b.weight; }
    __device__ static void add_atomic(Row& a, const Row& b) { atomicMax(&a.weight, b.weight); }
    __device__ static void finish(Row&, uint32_t) {}
};
template <>
struct RowOps<RK_TOPK_SRC> : RowOpsTopk {};
template <>
struct RowOps<RK_TOPK_DST> : RowOpsTopk {};

// ---- sort keys: only the bits that differ ----------------------------------------------------------------------------
// A row kind's order is up to four 64-bit words (224 bits for (SrcAddr,DstPort,Proto) rows) - but inside ONE row set most of
// those bits are the same in every row: one or a few timeslots, IPv4 addresses in a FixedString(16), ports below 2^16,
// two or three protocol numbers.  A first pass takes the OR and the AND of every key word over the rows; a bit that is
// equal in both is the same everywhere and cannot decide a comparison.  The bits that do vary are packed - in their order of
// significance - into as few 64-bit words as they need (usually one), and the LSD radix sort runs over those, with
// end_bit = the bits in use: config 5's 16.6 M-row window sorts 45 bits in ONE pass of SortPairs instead of 224 bits in four
// passes with gathers through the permutation in between.  Exact for every input: rows compare on the packed bits as they
// do on the full key (the first bit in which two keys differ is a varying bit, and packing keeps the order of bits).
template <int KIND, bool EMIT>
__device__ __forceinline__ unsigned long long row_key_word(const typename RowOps<KIND>::Row& r, int w, uint32_t fold) {
    if constexpr (EMIT) return RowOps<KIND>::key2(r, w);
    else return RowOps<KIND>::key(r, w, fold);
}
// bits[w] |= every row's key word w;  bits[NW + w] &= it   (bits: NW zeros, then NW times ~0)
template <int KIND, bool EMIT>
__global__ void row_bits_kernel(const typename RowOps<KIND>::Row* rows, uint32_t n, uint32_t fold, unsigned long long* bits) {
    constexpr int NW = EMIT ? RowOps<KIND>::NK2 : RowOps<KIND>::NK;
    unsigned long long o[NW > 0 ? NW : 1], a[NW > 0 ? NW : 1];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        o[w] = 0ull;
        a[w] = ~0ull;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const typename RowOps<KIND>::Row r = rows[i];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const unsigned long long k = row_key_word<KIND, EMIT>(r, w, fold);
            o[w] |= k;
            a[w] &= k;
        }
    }
#pragma unroll
    for (int w = 0; w < NW; w++) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            o[w] |= (unsigned long long)__shfl_xor((long long)o[w], d);
            a[w] &= (unsigned long long)__shfl_xor((long long)a[w], d);
        }
        if (__lane_id() == 0) {
            atomicOr(&bits[w], o[w]);
            atomicAnd(&bits[NW + w], a[w]);
        }
    }
}
// packed word p of every row, in the order idx gives (idx == nullptr: identity, which also initialises idx_out)
template <int KIND, bool EMIT>
__global__ void row_pack_kernel(const typename RowOps<KIND>::Row* rows, const uint32_t* idx, uint32_t n, uint32_t fold, RowPlan plan, uint32_t p,
                                unsigned long long* out, uint32_t* idx_out) {
    constexpr int NW = EMIT ? RowOps<KIND>::NK2 : RowOps<KIND>::NK;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t j = idx ? idx[i] : i;
        const typename RowOps<KIND>::Row r = rows[j];
        unsigned long long kw[NW > 0 ? NW : 1];
#pragma unroll
        for (int w = 0; w < NW; w++) kw[w] = row_key_word<KIND, EMIT>(r, w, fold);
        const unsigned long long v = rowplan_pack(plan, kw, NW, p);
        out[i] = v;
        if (!idx) idx_out[i] = i;
    }
}
__global__ void iota_kernel(uint32_t* idx, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) idx[i] = i;
}
// head flags of the sorted sequence
template <int KIND>
__global__ void row_heads_kernel(const typename RowOps<KIND>::Row* rows, const uint32_t* idx, uint32_t n, uint32_t fold, uint32_t* flags) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        flags[i] = (i == 0 || !RowOps<KIND>::same(rows[idx[i]], rows[idx[i - 1]], fold)) ? 1u : 0u;
}
// One thread per run sums its rows and writes the group's row.  Runs are as long as there are ranks x sub-buckets -
// short - but the input of fa_rows_merge_device is the caller's: a head only walks the first RUN_SERIAL rows of its
// run; what lies beyond (row_tail_kernel, launched only when a longer run exists) is added by the rows themselves
// with memory-side atomics, so a degenerate input (every row the same key) costs n atomics, not one thread's n steps.
constexpr uint32_t RUN_SERIAL = 32;
template <int KIND>
__global__ void row_reduce_kernel(const typename RowOps<KIND>::Row* rows, const uint32_t* idx, const uint32_t* flags, const uint32_t* pos,
                                  uint32_t n, uint32_t fold, typename RowOps<KIND>::Row* out, unsigned int* long_runs) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!flags[i]) continue;
        typename RowOps<KIND>::Row acc = rows[idx[i]];
        uint32_t j = i + 1;
        for (; j < n && j < i + RUN_SERIAL && !flags[j]; j++) RowOps<KIND>::add(acc, rows[idx[j]]);
        if (j < n && !flags[j]) atomicAdd(long_runs, 1u);  // (the run goes on: row_tail_kernel adds the rest)
        RowOps<KIND>::finish(acc, fold);
        out[pos[i]] = acc;
    }
}
// rows that sit RUN_SERIAL or more rows behind their run's head (no head among the RUN_SERIAL - 1 rows before them)
template <int KIND>
__global__ void row_tail_kernel(const typename RowOps<KIND>::Row* rows, const uint32_t* idx, const uint32_t* flags, const uint32_t* pos,
                                uint32_t n, typename RowOps<KIND>::Row* out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (flags[i]) continue;
        bool covered = false;
        for (uint32_t b = 1; b < RUN_SERIAL && b <= i && !covered; b++) covered = flags[i - b] != 0u;
        if (covered) continue;
        RowOps<KIND>::add_atomic(out[pos[i] - 1u], rows[idx[i]]);  // (pos = heads before row i: its group is the last of them)
    }
}
template <class Row>
__global__ void row_gather_kernel(const Row* src, const uint32_t* idx, uint32_t n, Row* dst) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}

// ---- hash partition of a row set (multi-GPU window close of large sparse sets) -----------------------------------------
// Rank r of `world` merges the keys with row_dest == r: every rank cuts ITS rows into `world` groups with the same function
// of the key alone, the groups travel with one all-to-all, and each rank sorts and sums 1 / world of the keys - instead of
// every rank receiving and sorting every rank's rows (SURVEY.md 8(e) option (ii); flow-pipeline_amd/dist.py).
// dest = high half of a mix64 chain over the kind's merge-order key words (no fold: fa_rows_device has applied it), scaled
// to [0, world) by multiply-shift.  Restated in numpy by dist.partition_rows_host (the CPU tests compare the two).
constexpr uint32_t RPART_MAX_WORLD = 1024, RPART_BLOCK = 256, RPART_U = 16;
template <int KIND>
__device__ __forceinline__ uint32_t row_dest(const typename RowOps<KIND>::Row& r, uint32_t world) {
    unsigned long long h = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int w = 0; w < RowOps<KIND>::NK; w++) h = mix64(h ^ RowOps<KIND>::key(r, w, 0xffffffffu));
    return (uint32_t)(((h >> 32) * (unsigned long long)world) >> 32);
}
// SCATTER = false: counts[d] += rows of group d.  SCATTER = true: rows to out[starts[d] + ...] (cursor[d]: rows of group d
// placed so far; the order inside a group is arbitrary - the merge behind the exchange sorts).
template <int KIND, bool SCATTER>
__global__ __launch_bounds__(RPART_BLOCK) void row_partition_kernel(const typename RowOps<KIND>::Row* rows, uint32_t n, uint32_t world, unsigned int* counts,
                                                                    const unsigned int* starts, unsigned int* cursor, typename RowOps<KIND>::Row* out) {
    __shared__ unsigned int hist[RPART_MAX_WORLD], base[RPART_MAX_WORLD];
    constexpr uint32_t PER = RPART_BLOCK * RPART_U;
    for (uint32_t b0 = blockIdx.x * PER; b0 < n; b0 += gridDim.x * PER) {  // (n < 2^31: no wrap)
        for (uint32_t d = threadIdx.x; d < world; d += RPART_BLOCK) hist[d] = 0;
        __syncthreads();
        uint32_t dest[RPART_U], lrank[RPART_U];
#pragma unroll
        for (uint32_t q = 0; q < RPART_U; q++) {
            const uint32_t i = b0 + q * RPART_BLOCK + threadIdx.x;
            dest[q] = lrank[q] = 0;
            if (i < n) {
                dest[q] = row_dest<KIND>(rows[i], world);
                lrank[q] = atomicAdd(&hist[dest[q]], 1u);
            }
        }
        __syncthreads();
        if (!SCATTER) {
            for (uint32_t d = threadIdx.x; d < world; d += RPART_BLOCK)
                if (hist[d]) atomicAdd(&counts[d], hist[d]);
        } else {
            for (uint32_t d = threadIdx.x; d < world; d += RPART_BLOCK) base[d] = hist[d] ? atomicAdd(&cursor[d], hist[d]) : 0u;
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < RPART_U; q++) {
                const uint32_t i = b0 + q * RPART_BLOCK + threadIdx.x;
                if (i < n) out[(size_t)starts[dest[q]] + base[dest[q]] + lrank[q]] = rows[i];
            }
        }
        __syncthreads();
    }
}

// ---- public-format rows out of the device state ------------------------------------------------------------------
// wide-table rows (packed keys) -> fa_row_app, in place (both 56 bytes; every thread rewrites its own element)
__global__ void wrows_to_app_kernel(WRow* rows, uint32_t n, uint32_t gran) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const WRow r = rows[i];
        uint32_t kind, tb, port, proto;
        uint64_t lo, hi;
        wkey_unpack(r.w, kind, tb, lo, hi, port, proto);
        RowApp o;
        o.timeslot = tb * gran;
        o.date = o.timeslot / 86400u;
        o.addr[0] = (uint32_t)lo;
        o.addr[1] = (uint32_t)(lo >> 32);
        o.addr[2] = (uint32_t)hi;
        o.addr[3] = (uint32_t)(hi >> 32);
        o.dst_port = port;
        o.proto = proto;
        o.bytes = r.v0;
        o.packets = r.v1;
        o.count = r.v2;
        *reinterpret_cast<RowApp*>(&rows[i]) = o;
    }
}
// wide-table rows of a port / minute kind -> {key, weight, count}; scale: 1 for ports, 60 for minutes
__global__ void wrows_to_w_kernel(const WRow* rows, uint32_t n, uint32_t scale, RowW* out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t kind, tb, port, proto;
        uint64_t lo, hi;
        wkey_unpack(rows[i].w, kind, tb, lo, hi, port, proto);
        out[i] = RowW{port * scale, 0u, rows[i].v0, rows[i].v2};
    }
}
// the dense port histogram's occupied entries, appended behind `base` rows (count in *n_out, starts at base)
__global__ void port_dense_rows_kernel(const ulonglong2* hist, uint32_t nports, RowW* out, unsigned int* n_out) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < nports; p += gridDim.x * blockDim.x) {
        const ulonglong2 e = hist[p];
        if (e.y) out[atomicAdd(n_out, 1u)] = RowW{p, 0u, e.x, e.y};
    }
}

// rows gathered from other ranks are checked before they are trusted: a timeslot off this ctx's bucket grid would be
// folded into the wrong bucket silently
__global__ void rows5m_check_kernel(const Row5m* rows, uint32_t n, uint32_t gran, unsigned int* bad) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (rows[i].timeslot % gran) atomicAdd(bad, 1u);
}

// fa_read_window_app48: the rows of ONE window share date and timeslot - they leave without them (a seventh of the bytes of the
// PCIe copy that a window close of this key set is: 930 MB -> 797 MB for 16.6 M rows)
__global__ __launch_bounds__(256) void row_app48_kernel(const RowApp* rows, uint32_t n, RowApp48* out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const RowApp r = rows[i];
        out[i] = RowApp48{{r.addr[0], r.addr[1], r.addr[2], r.addr[3]}, r.dst_port, r.proto, r.bytes, r.packets, r.count};
    }
}

// fa_read_window_app48 in two halves (rows_host.inc, rows_read_app48): the most significant key word of a window's rows - the first
// eight address bytes in comparison order - sampled for a pivot, and the kernel that cuts the rows at it
__global__ __launch_bounds__(256) void app_key_sample_kernel(const RowApp* rows, uint32_t n, uint32_t stride, unsigned long long* out, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = RowOps<RK_APP>::key(rows[min((unsigned long long)i * stride, (unsigned long long)n - 1ull)], 2, 0xffffffffu);
}
// The cut: rows whose key word lies below the pivot to the front of `out` (cnt[0] of them when the kernel is done), the others to its
// end, backwards (cnt[1]) - in any order inside a half, both are sorted next.  A 1024-thread workgroup takes the places of a round's
// rows with two returning atomics (hipcub::DevicePartition::If took 4.4 ms for 16.6 M 56-byte rows; this is one read and one write).
constexpr int APP_CUT_BLOCK = 1024;
__global__ __launch_bounds__(APP_CUT_BLOCK) void app_cut_kernel(const RowApp* in, uint32_t n, unsigned long long pivot, RowApp* out, unsigned int* cnt) {
    __shared__ uint32_t wcnt[2][APP_CUT_BLOCK / 64];
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * APP_CUT_BLOCK; base < n; base += gridDim.x * APP_CUT_BLOCK) {  // (the trip count is the workgroup's)
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < n;
        RowApp r = in[valid ? i : n - 1u];
        const bool below = valid && RowOps<RK_APP>::key(r, 2, 0xffffffffu) < pivot;
        const unsigned long long ma = __builtin_amdgcn_ballot_w64(below), mb = __builtin_amdgcn_ballot_w64(valid && !below);
        if (lane == 0) {
            wcnt[0][wave] = (uint32_t)__builtin_popcountll(ma);
            wcnt[1][wave] = (uint32_t)__builtin_popcountll(mb);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            constexpr uint32_t NW = APP_CUT_BLOCK / 64;
            const uint32_t half = lane >= 32 ? 1u : 0u, w = lane & 31u;
            const uint32_t t = w < NW ? wcnt[half][w] : 0u;
            uint32_t incl = t;
#pragma unroll
            for (int o = 1; o < (int)NW; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 32);
                if (w >= (uint32_t)o) incl += up;
            }
            const uint32_t tot = (uint32_t)__shfl((int)incl, (int)(NW - 1u), 32);
            uint32_t b = 0;
            if (w == 0 && tot) b = atomicAdd(&cnt[half], tot);
            b = (uint32_t)__shfl((int)b, 0, 32);
            if (w < NW) wcnt[half][w] = b + incl - t;  // where this wave's rows of this half start
        }
        __syncthreads();
        const uint32_t below_lanes = (uint32_t)__builtin_popcountll(ma & ((1ull << lane) - 1ull));
        const uint32_t other_lanes = (uint32_t)__builtin_popcountll(mb & ((1ull << lane) - 1ull));
        if (below) out[wcnt[0][wave] + below_lanes] = r;
        else if (valid) out[n - 1u - (wcnt[1][wave] + other_lanes)] = r;
        __syncthreads();
    }
}

// window close of a group of contexts (group_host.inc): member r's slice of a sketch = its own slice + the same slice of every
// other member, staged back to back (stride words apart) in the member's exchange buffer.  Plain 16-byte loads and stores,
// every word read once: n x slice bytes at the copy rate.  own / stage / out are 16-byte aligned, w is even or the tail is
// taken by the last thread.
__global__ __launch_bounds__(256) void group_sum_kernel(const unsigned long long* own, const unsigned long long* stage, uint32_t nstage, size_t stride, size_t w,
                                                        unsigned long long* out) {
    const size_t pairs = w / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
        ulonglong2 s = reinterpret_cast<const ulonglong2*>(own)[i];
        for (uint32_t q = 0; q < nstage; q++) {
            const ulonglong2 v = reinterpret_cast<const ulonglong2*>(stage + (size_t)q * stride)[i];
            s.x += v.x;
            s.y += v.y;
        }
        reinterpret_cast<ulonglong2*>(out)[i] = s;
    }
    if ((w & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long s = own[w - 1];
        for (uint32_t q = 0; q < nstage; q++) s += stage[(size_t)q * stride + w - 1];
        out[w - 1] = s;
    }
}

}  // namespace fa
