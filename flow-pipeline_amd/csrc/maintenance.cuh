// maintenance.cuh - kernels off the hot path: window close (extract / rebuild), spill replay, row merges, wide-table
// maintenance, top-k rows, synthetic producer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sinks.cuh"

namespace fa {

// ---- window close ---------------------------------------------------------------
struct Row5m {
    uint32_t date, timeslot, src_as, dst_as, etype, pad;
    unsigned long long bytes, packets, count;
};

// Appends rows whose time bucket lies in [tb_lo, tb_hi) to `rows`.  A wave looks at EX_U x 64 slots per round (all
// loads first) and takes the positions of the round's rows with one returning atomic - the scan waits for that round
// trip, not for bandwidth (same as wextract_kernel below).
constexpr int EX_U = 4;
__global__ __launch_bounds__(256) void extract_kernel(const Slot* tab, uint32_t nslots, uint32_t gran, uint32_t tb_lo,
                                                      uint32_t tb_hi, Row5m* rows, uint32_t rows_cap, Counters* ctr) {
    const uint32_t nthr = gridDim.x * blockDim.x, lane = __lane_id();
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < nslots; i0 += EX_U * nthr) {
        ulonglong2 k[EX_U], bp[EX_U];
        unsigned long long cnt[EX_U], m[EX_U];
        bool sel[EX_U];
        uint32_t total = 0;
#pragma unroll
        for (int u = 0; u < EX_U; u++) {
            const Slot* sp = &tab[min(i0 + (uint32_t)u * nthr, nslots - 1u)];
            k[u] = *reinterpret_cast<const ulonglong2*>(&sp->k0);
            bp[u] = *reinterpret_cast<const ulonglong2*>(&sp->bytes);
            cnt[u] = sp->count;
        }
#pragma unroll
        for (int u = 0; u < EX_U; u++) {
            uint32_t tb, sa, da, et;
            unpack_key(k[u].x, k[u].y, tb, sa, da, et);
            sel[u] = i0 + (uint32_t)u * nthr < nslots && k[u].x != 0 && k[u].y != 0 && cnt[u] != 0 && tb >= tb_lo && tb < tb_hi;
            m[u] = __builtin_amdgcn_ballot_w64(sel[u]);
            total += (uint32_t)__builtin_popcountll(m[u]);
        }
        if (total != 0u) {  // (wave-uniform)
            const uint32_t leader = (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
            unsigned int base = 0;
            if (lane == leader) base = atomicAdd(&ctr->rows_count, total);
            base = (unsigned int)__builtin_amdgcn_readlane((int)base, (int)leader);
#pragma unroll
            for (int u = 0; u < EX_U; u++) {
                const unsigned int j = base + (unsigned int)__builtin_popcountll(m[u] & ((1ull << lane) - 1ull));
                base += (unsigned int)__builtin_popcountll(m[u]);
                if (sel[u] && j < rows_cap) {
                    uint32_t tb, sa, da, et;
                    unpack_key(k[u].x, k[u].y, tb, sa, da, et);
                    const uint32_t ts = tb * gran;
                    rows[j] = Row5m{ts / 86400u, ts, sa, da, et, 0, bp[u].x, bp[u].y, cnt[u]};
                }
            }
        }
    }
}

// ---- which time buckets does the table hold?  (fa_open_timeslots: a consumer asks before every flush) -----------------
// Pass 1 (bits == nullptr): range[0] = min, range[1] = max bucket of the live groups.  Pass 2: bit (tb - lo) of `bits`.
// Two scans of the table and a few hundred bytes to the host, instead of every row copied out and sorted there.
__global__ __launch_bounds__(256) void timeslots_kernel(const Slot* tab, uint32_t nslots, uint32_t lo, uint32_t nbits, unsigned int* bits, unsigned int* range) {
    uint32_t mn = 0xffffffffu, mx = 0u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += gridDim.x * blockDim.x) {
        const ulonglong2 k = *reinterpret_cast<const ulonglong2*>(&tab[i].k0);
        if (k.x == 0 || k.y == 0 || tab[i].count == 0) continue;
        uint32_t tb, sa, da, et;
        unpack_key(k.x, k.y, tb, sa, da, et);
        if (bits) {
            if (tb - lo < nbits) atomicOr(&bits[(tb - lo) >> 5], 1u << ((tb - lo) & 31u));
        } else {
            mn = min(mn, tb);
            mx = max(mx, tb);
        }
    }
    if (!bits) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        }
        if (__lane_id() == 0 && mn != 0xffffffffu) {
            atomicMin(&range[0], mn);
            atomicMax(&range[1], mx);
        }
    }
}

// ---- window close, device side: sort the extracted rows by (date, timeslot, src_as, dst_as, etype) ---------
// Two stable radix-sort passes over 64-bit keys (hipcub, flowagg.hip): low key (DstAS, EType) first, then high key
// (Timeslot, SrcAS); Date is a function of Timeslot.  fold_ts != ~0: the rows of a sliding window get the window's
// start as their timeslot (the sub-buckets of one group then sort next to each other and are summed by the host).
__global__ void row_keys_kernel(Row5m* rows, uint32_t n, uint32_t fold_ts, unsigned long long* klo, unsigned long long* khi, uint32_t* idx) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (fold_ts != 0xffffffffu) {
            rows[i].timeslot = fold_ts;
            rows[i].date = fold_ts / 86400u;
        }
        klo[i] = (unsigned long long)rows[i].dst_as << 32 | rows[i].etype;
        khi[i] = (unsigned long long)rows[i].timeslot << 32 | rows[i].src_as;
        idx[i] = i;
    }
}
__global__ void gather_u64_kernel(const unsigned long long* src, const uint32_t* idx, uint32_t n, unsigned long long* dst) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
__global__ void gather_rows_kernel(const Row5m* src, const uint32_t* idx, uint32_t n, Row5m* dst) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}

// Re-inserts every row outside [tb_lo, tb_hi) into a fresh table (window removal / growth).
__global__ void rebuild_kernel(const Slot* old_tab, uint32_t old_slots, uint32_t tb_lo, uint32_t tb_hi,
                               KArgs a) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < old_slots; i += gridDim.x * blockDim.x) {
        const Slot& s = old_tab[i];
        if (s.k0 == 0 || s.k1 == 0 || s.count == 0) continue;
        uint32_t tb, sa, da, et;
        unpack_key(s.k0, s.k1, tb, sa, da, et);
        if (tb >= tb_lo && tb < tb_hi) continue;
        agg_global(a, s.k0, s.k1, key_hash(s.k0, s.k1), s.bytes, s.packets, s.count);
    }
}

__global__ void replay_spill_kernel(const SpillEntry* sp, uint32_t n, KArgs a) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        agg_global(a, sp[i].k0, sp[i].k1, key_hash(sp[i].k0, sp[i].k1), sp[i].bytes, sp[i].packets,
                   sp[i].count);
}

// rows produced elsewhere (another GPU / Kafka partition) folded into this table
__global__ void merge_rows_kernel(const Row5m* rows, uint32_t n, KArgs a) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t k0, k1;
        pack_key(rows[i].timeslot / a.gran, rows[i].src_as, rows[i].dst_as, rows[i].etype, k0, k1);
        agg_global(a, k0, k1, key_hash(k0, k1), rows[i].bytes, rows[i].packets, rows[i].count);
    }
}

// ---- wide table maintenance ---------------------------------------------------------------------
struct WRow {
    unsigned long long w[4], v0, v1, v2;
};
// which rows: kind_mask bit k selects kind k; WK_APP rows additionally need tb in [tb_lo, tb_hi)
__device__ __forceinline__ bool wrow_selected(const unsigned long long w[4], uint32_t kind_mask, uint32_t tb_lo, uint32_t tb_hi) {
    uint32_t kind, tb, port, proto;
    uint64_t lo, hi;
    wkey_unpack(w, kind, tb, lo, hi, port, proto);
    if (!((kind_mask >> kind) & 1u)) return false;
    return kind != WK_APP || (tb >= tb_lo && tb < tb_hi);
}
// A workgroup of 16 waves looks at WX_U x 1024 slots per round (each slot with four independent 16-byte loads) and takes the
// positions of the round's rows with ONE returning atomic: the scan itself runs at 6 TB/s (tools/micro/scan64.hip), what it
// waits for is that round trip - 41 ms for a 16 GiB table with one atomic per selected row; with one per wave and round (rounds
// 2-4) a table that holds a row in most rounds still paid 262 k same-address round trips, 2.3 ms for BASELINE config 5's first
// window (1.2 M rows in 2^26 slots) beside a 0.7 ms scan.
constexpr int WX_U = 4;
constexpr int WX_BLOCK = 1024;
__global__ __launch_bounds__(WX_BLOCK) void wextract_kernel(const WSlot* tab, uint32_t nslots, uint32_t kind_mask, uint32_t tb_lo, uint32_t tb_hi, WRow* rows,
                                                            uint32_t rows_cap, Counters* ctr) {
    __shared__ uint32_t wave_total[WX_BLOCK / 64];
    __shared__ uint32_t wg_base;
    const uint32_t nthr = gridDim.x * blockDim.x, lane = __lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t b0 = blockIdx.x * blockDim.x; b0 < nslots; b0 += WX_U * nthr) {  // (the trip count is the workgroup's: barriers inside)
        const uint32_t i0 = b0 + threadIdx.x;
        uint4 q[WX_U][4];
        bool sel[WX_U];
        unsigned long long m[WX_U];
        uint32_t total = 0;
#pragma unroll
        for (int u = 0; u < WX_U; u++) {
            const uint32_t i = i0 + (uint32_t)u * nthr;
            const uint4* p = reinterpret_cast<const uint4*>(&tab[min(i, nslots - 1u)]);
#pragma unroll
            for (int k = 0; k < 4; k++) q[u][k] = p[k];
        }
#pragma unroll
        for (int u = 0; u < WX_U; u++) {
            const unsigned long long w[4] = {(unsigned long long)q[u][0].y << 32 | q[u][0].x, (unsigned long long)q[u][0].w << 32 | q[u][0].z,
                                             (unsigned long long)q[u][1].y << 32 | q[u][1].x, (unsigned long long)q[u][1].w << 32 | q[u][1].z};
            const unsigned long long v2 = (unsigned long long)q[u][3].y << 32 | q[u][3].x;
            // (i0 + u * nthr < 2^32: nslots <= 2^30 and the last round overshoots by less than WX_U * nthr)
            sel[u] = i0 + (uint32_t)u * nthr < nslots && w[0] != 0 && w[1] != 0 && w[2] != 0 && w[3] != 0 && v2 != 0 &&
                     wrow_selected(w, kind_mask, tb_lo, tb_hi);
            m[u] = __builtin_amdgcn_ballot_w64(sel[u]);
            total += (uint32_t)__builtin_popcountll(m[u]);
        }
        if (lane == 0) wave_total[wave] = total;
        __syncthreads();
        if (threadIdx.x < 64) {  // wave 0: the workgroup's rows of this round, one atomic
            const uint32_t t = lane < WX_BLOCK / 64 ? wave_total[lane] : 0u;
            uint32_t incl = t;
#pragma unroll
            for (int o = 1; o < WX_BLOCK / 64; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                if (lane >= (uint32_t)o) incl += up;
            }
            const uint32_t wg_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, WX_BLOCK / 64 - 1);
            uint32_t base = 0;
            if (lane == 0 && wg_total != 0u) base = atomicAdd(&ctr->wrows_count, wg_total);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
            if (lane < WX_BLOCK / 64) wave_total[lane] = base + incl - t;  // (exclusive: where this wave's rows start)
            if (lane == 0) wg_base = wg_total;
        }
        __syncthreads();
        if (wg_base != 0u && total != 0u) {  // (wave-uniform)
            unsigned int base = wave_total[wave];
#pragma unroll
            for (int u = 0; u < WX_U; u++) {
                const unsigned int j = base + (unsigned int)__builtin_popcountll(m[u] & ((1ull << lane) - 1ull));
                base += (unsigned int)__builtin_popcountll(m[u]);
                if (sel[u] && j < rows_cap) {
                    uint4* d = reinterpret_cast<uint4*>(&rows[j]);  // 56-byte rows: 8-byte aligned only
                    unsigned long long* d8 = reinterpret_cast<unsigned long long*>(d);
                    d8[0] = (unsigned long long)q[u][0].y << 32 | q[u][0].x;
                    d8[1] = (unsigned long long)q[u][0].w << 32 | q[u][0].z;
                    d8[2] = (unsigned long long)q[u][1].y << 32 | q[u][1].x;
                    d8[3] = (unsigned long long)q[u][1].w << 32 | q[u][1].z;
                    d8[4] = (unsigned long long)q[u][2].y << 32 | q[u][2].x;
                    d8[5] = (unsigned long long)q[u][2].w << 32 | q[u][2].z;
                    d8[6] = (unsigned long long)q[u][3].y << 32 | q[u][3].x;
                }
            }
        }
        __syncthreads();  // (wave_total / wg_base are the next round's again)
    }
}
// ---- wide log (FA_WIDE=log): launches whose (SrcAddr,DstPort,Proto) tuples still sit in their scatter segments ----------
// One chunk = one launch's segments wseg[region][workgroup][q] + counts, its time base (device word) and a watermark
// (buckets below it were dropped after the launch).  A wave takes a segment at a time.
struct WChunkArgs {
    const uint4* wseg;
    const uint32_t* counts;
    const uint32_t* tb_base;
    uint32_t nparts, nwg, capq, wm;
    size_t region;
};
template <class F>
__device__ __forceinline__ void wchunk_walk(const WChunkArgs& k, F&& f) {
    const uint32_t lane = __lane_id(), nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t tb_base = *k.tb_base;
    const uint32_t nseg = k.nparts * k.nwg;
    for (uint32_t sgi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); sgi < nseg; sgi += nwaves) {
        const uint32_t p = sgi / k.nwg, w = sgi % k.nwg;
        const uint32_t cnt = min(k.counts[(size_t)p * k.nwg + w], k.capq);
        const uint4* base = k.wseg + 2u * ((size_t)p * k.region + (size_t)w * k.capq);
        for (uint32_t q0 = 0; q0 < cnt; q0 += 64u) {  // (wave-uniform trip count)
            const uint32_t q = q0 + lane;
            const bool valid = q < cnt;
            const uint4 t0 = base[2u * (valid ? q : 0u)], t1 = base[2u * (valid ? q : 0u) + 1u];
            WKey key;
            uint64_t bytes, packets;
            wtup_unpack(t0, t1, tb_base, key, bytes, packets);
            const uint32_t tb = tb_base + (t1.w >> 24);
            f(valid && tb >= k.wm, tb, key, bytes, packets);
        }
    }
}
// The chunk's live tuples of buckets [tb_lo, tb_hi) as packed-key rows (count() = 1 each), appended behind the rows that are
// there already (*row_base).  Two passes and a scan instead of one returning atomic per 64 tuples on ONE counter word
// (round 4, first cut: 4.7 ms per 16.67 M-tuple chunk, 2.6 of them the 262 k same-address atomics): wlog_count_kernel counts
// the selected tuples of every segment, an exclusive scan gives each segment its place, wlog_rows_kernel writes - no atomics,
// no holes.  wlog_bump_kernel then moves *row_base behind the chunk's rows.
template <bool WRITE>
__device__ __forceinline__ void wlog_rows_pass(const WChunkArgs& k, uint32_t tb_lo, uint32_t tb_hi, uint32_t* seg_sel, const uint32_t* seg_pos, WRow* rows,
                                               uint32_t rows_cap, uint32_t base) {
    const uint32_t lane = __lane_id(), nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t tb_base = *k.tb_base;
    const uint32_t nseg = k.nparts * k.nwg;
    for (uint32_t sgi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); sgi < nseg; sgi += nwaves) {
        const uint32_t p = sgi / k.nwg, w = sgi % k.nwg;
        const uint32_t cnt = min(k.counts[(size_t)p * k.nwg + w], k.capq);
        const uint4* seg = k.wseg + 2u * ((size_t)p * k.region + (size_t)w * k.capq);
        uint32_t run = WRITE ? base + seg_pos[sgi] : 0u;  // (wave-uniform)
        for (uint32_t q0 = 0; q0 < cnt; q0 += 64u) {
            const uint32_t q = q0 + lane;
            const bool valid = q < cnt;
            const uint4 t1 = seg[2u * (valid ? q : 0u) + 1u];
            const uint32_t tb = tb_base + (t1.w >> 24);
            const bool sel = valid && tb >= k.wm && tb >= tb_lo && tb < tb_hi;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(sel);
            if (WRITE && sel) {
                const uint4 t0 = seg[2u * q];
                WKey key;
                uint64_t bytes, packets;
                wtup_unpack(t0, t1, tb_base, key, bytes, packets);
                const uint32_t j = run + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (j < rows_cap) rows[j] = WRow{{key.w[0], key.w[1], key.w[2], key.w[3]}, bytes, packets, 1ull};
            }
            run += (uint32_t)__builtin_popcountll(m);
        }
        if (!WRITE && lane == 0) seg_sel[sgi] = run;
    }
}
__global__ __launch_bounds__(256) void wlog_count_kernel(WChunkArgs k, uint32_t tb_lo, uint32_t tb_hi, uint32_t* seg_sel) {
    wlog_rows_pass<false>(k, tb_lo, tb_hi, seg_sel, nullptr, nullptr, 0u, 0u);
}
__global__ __launch_bounds__(256) void wlog_rows_kernel(WChunkArgs k, uint32_t tb_lo, uint32_t tb_hi, const uint32_t* seg_pos, WRow* rows, uint32_t rows_cap,
                                                        const unsigned int* row_base) {
    wlog_rows_pass<true>(k, tb_lo, tb_hi, nullptr, seg_pos, rows, rows_cap, *row_base);
}
__global__ void wlog_bump_kernel(const uint32_t* seg_sel, const uint32_t* seg_pos, uint32_t nseg, unsigned int* row_base) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && nseg) *row_base += seg_pos[nseg - 1] + seg_sel[nseg - 1];
}
// the smallest and the largest bucket among the chunk's live tuples (out[0] starts at ~0, out[1] at 0): what a window
// close compares its range with, and how it knows that nothing of a chunk is left
__global__ __launch_bounds__(256) void wlog_minbucket_kernel(WChunkArgs k, uint32_t* out) {
    uint32_t m = 0xffffffffu, x = 0u;
    wchunk_walk(k, [&](bool live, uint32_t tb, const WKey&, uint64_t, uint64_t) {
        m = live ? min(m, tb) : m;
        x = live ? max(x, tb) : x;
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m = min(m, (uint32_t)__shfl_xor((int)m, o));
        x = max(x, (uint32_t)__shfl_xor((int)x, o));
    }
    if (__lane_id() == 0 && m != 0xffffffffu) {
        atomicMin(&out[0], m);
        atomicMax(&out[1], x);
    }
}
// the chunk folded into the table through the atomic path (the table's geometry changed since the tuples were scattered:
// their regions are not the table's any more, wagg_kernel's ownership does not hold)
__global__ __launch_bounds__(256) void wlog_replay_kernel(WChunkArgs k, KArgs a) {
    const WArgs t = wargs(a);
    wchunk_walk(k, [&](bool live, uint32_t, const WKey& key, uint64_t bytes, uint64_t packets) {
        if (live) wagg_global(t, key, bytes, packets, 1);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
        if (m != 0ull && __lane_id() == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&a.ctr->wfold_n, (unsigned long long)__builtin_popcountll(m));
    });
}

// Window close without a rebuild: the selected rows' sums are zeroed in place.  A slot whose count() is 0 is no row (every
// reader and the rebuild skip it) but keeps its key, so probe sequences stay intact; the same key arriving again simply
// adds to the zeros.  The dead slots go when the table is rebuilt for growth (or, when half of a full table is dead, at
// the same size: settle_wide).  One scan of the table instead of a scan plus the re-insertion of every surviving row
// into a second table of the same size (16 GiB for BASELINE config 5: 54 ms per close in round 3).
__global__ __launch_bounds__(256) void wdrop_kernel(WSlot* tab, uint32_t nslots, uint32_t kind_mask, uint32_t tb_lo, uint32_t tb_hi, unsigned int* dropped) {
    uint32_t mine = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += gridDim.x * blockDim.x) {
        const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(&tab[i].w[0]);
        const ulonglong2 k23 = *reinterpret_cast<const ulonglong2*>(&tab[i].w[2]);
        if (k01.x == 0 || k01.y == 0 || k23.x == 0 || k23.y == 0) continue;
        const unsigned long long w[4] = {k01.x, k01.y, k23.x, k23.y};
        if (!wrow_selected(w, kind_mask, tb_lo, tb_hi) || tab[i].v2 == 0) continue;
        *reinterpret_cast<ulonglong2*>(&tab[i].v0) = make_ulonglong2(0ull, 0ull);
        tab[i].v2 = 0ull;
        mine++;
    }
    const uint32_t tot = wave_sum_u32(mine);
    if (__lane_id() == 0 && tot) atomicAdd(dropped, tot);
}

// Re-inserts every row that is NOT selected into a fresh table (window removal / reset / growth).
__global__ void wrebuild_kernel(const WSlot* old_tab, uint32_t old_slots, uint32_t kind_mask, uint32_t tb_lo, uint32_t tb_hi, KArgs a) {
    const WArgs t = wargs(a);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < old_slots; i += gridDim.x * blockDim.x) {
        const WSlot& s = old_tab[i];
        if (s.w[0] == 0 || s.w[1] == 0 || s.w[2] == 0 || s.w[3] == 0 || s.v2 == 0) continue;
        if (wrow_selected(s.w, kind_mask, tb_lo, tb_hi)) continue;
        WKey k{{s.w[0], s.w[1], s.w[2], s.w[3]}};
        wagg_global(t, k, s.v0, s.v1, s.v2);
    }
}
// parked updates / rows produced elsewhere (another GPU / Kafka partition) folded into this table
__global__ void wmerge_kernel(const WRow* rows, uint32_t n, KArgs a) {
    const WArgs t = wargs(a);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        WKey k{{rows[i].w[0], rows[i].w[1], rows[i].w[2], rows[i].w[3]}};
        wagg_global(t, k, rows[i].v0, rows[i].v1, rows[i].v2);
    }
}

// ---- heavy hitters ---------------------------------------------------------------------------
// One row per stored key: its Count-Min estimate = min over the sketch rows (>= the exact sum(Bytes*SamplingRate),
// viz-ch.json:233).  The merge behind it (rows_merge_t<RK_TOPK_*>) orders by weight and cuts at k.
// A read wants k rows out of millions (BASELINE config 3: 8 M distinct addresses in a set of 2^26 slots), and round 4 handed
// EVERY row to the sort - one returning atomic per row on one counter word (8 M same-address memory-side atomics) and a radix
// sort of 8 M rows to keep 100: 12 ms per call.  Now the set is scanned twice:
//   topk_hist_kernel   estimates -> a histogram over 2048 monotone bins (64 octaves x 32 steps), per workgroup in LDS;
//   topk_thresh_kernel the bin that holds rank k: everything in a lower bin is strictly lighter than k rows above it;
//   topk_rows_kernel   rows of the bins >= that one only - a few hundred more than k on a skewed stream -, positions claimed
//                      once per wave and round (ballot + one atomic).
// Exact: the selection holds every row that can be among the first k in (weight DESC, key) order.  k = 0 (all rows) skips the
// first two kernels.  A stream whose estimates all fall into one bin selects everything: the old cost, the same result.
// The set is sparse (an eighth of its slots hold a key at BASELINE config 3's load) and an estimate is a chain of hashes and
// `depth` random sketch reads: evaluated where the key was loaded, a wave ran that chain 4 x per round for a handful of lanes
// each (1.3 ms per scan of 2 GiB).  Now a round's keys are first compacted into a wave-private LDS list and the estimates
// run densely, one key per lane, the sketch reads of a key issued together.
constexpr int TK_U = 8;
enum { TK_HIST = 0, TK_ROWS = 1, TK_ONE = 2 };
// the key's sketch coordinates and its row-0 counter (an upper bound of the estimate; in set order the row-0 columns of a
// partition come in ascending order - these reads stay in cache - while the rows behind them are random 64-byte fabric requests)
__device__ __forceinline__ unsigned long long topk_row0(unsigned long long lo, unsigned long long hi, const unsigned long long* cms, uint32_t wl2, uint64_t seed, CmsKey& k) {
    uint64_t h, h2;
    cms_hash2(lo, hi, seed, h, h2);
    k = cms_key(h, h2, wl2);
    return cms[cms_column(k, 0, wl2)];
}
__device__ __forceinline__ unsigned long long topk_rest(unsigned long long best, const CmsKey& k, const unsigned long long* cms, uint32_t depth, uint32_t wl2) {
    for (uint32_t r0 = 1; r0 < depth; r0 += 4) {
        unsigned long long v[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {  // (rows beyond depth repeat the last one: four independent loads in flight)
            const uint32_t r = min(r0 + q, depth - 1u);
            v[q] = cms[((size_t)r << wl2) + cms_column(k, r, wl2)];
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) best = v[q] < best ? v[q] : best;
    }
    return best;
}
__device__ __forceinline__ unsigned long long topk_estimate(unsigned long long lo, unsigned long long hi, const unsigned long long* cms, uint32_t depth, uint32_t wl2, uint64_t seed) {
    CmsKey k;
    const unsigned long long u0 = topk_row0(lo, hi, cms, wl2, seed, k);
    return topk_rest(u0, k, cms, depth, wl2);
}
struct TopkLds {
    ulonglong2 key[4][TK_U * 64];  // per wave: the (lo, hi) of the round's keys, compacted
    uint32_t slot[4][TK_U * 64];   // ... and where they sit
    unsigned int wmax[4][TK_U];    // the largest bin of each of the wave's TK_U chunks of 64 slots
};
// One scan of the set (modes):
//   TK_HIST  estimates -> histogram of their bins; the estimate is left in the slot's spare word (KeySlot::pad) and the largest
//            bin of every chunk of 64 slots in chunkmax - what topk_pick_kernel needs to find the selected rows without a second
//            scan (stored as bin + 1; 0 = the chunk holds no estimated key).  lb_bin > 0: a bin the k-th estimate is known to reach (the previous read's k-th row: estimates and sets only
//            grow) - a key whose ROW-0 counter already lies below it is counted in bin 0 and costs one cached read.
//   TK_ONE   the rows whose bin is >= lb_bin, in one pass (no histogram): k = 0 reads (lb_bin = 0), and reads whose lb_bin is
//            known - almost every key stops at its row-0 counter.
template <int M>
__device__ __forceinline__ void topk_scan_body(unsigned int* lh, TopkLds& L, uint32_t block, uint32_t nblocks, KeySlot* ks, uint32_t nslots, const unsigned long long* cms,
                                               uint32_t depth, uint32_t wl2, uint64_t seed, uint32_t lb_bin, unsigned int* hist, unsigned short* chunkmax, TopkRow* rows,
                                               uint32_t rows_cap, Counters* ctr) {
    const uint32_t nthr = nblocks * blockDim.x, lane = __lane_id(), wave = threadIdx.x >> 6;
    if constexpr (M == TK_HIST) {
        for (uint32_t b = threadIdx.x; b < TK_BINS; b += blockDim.x) lh[b] = 0u;
        if (lane < (uint32_t)TK_U) L.wmax[wave][lane] = 0u;
        __syncthreads();
    }
    ulonglong2* mine = L.key[wave];
    uint32_t* mslot = L.slot[wave];
    // (wave-uniform trip count - the ballots below want every lane: nslots is a power of two >= 256, a wave's 64 slots are in or out together)
    for (uint32_t w0 = block * blockDim.x + (threadIdx.x & ~63u); w0 < nslots; w0 += TK_U * nthr) {
        const uint32_t i0 = w0 + lane;
        ulonglong2 tl[TK_U];
        unsigned long long hi[TK_U];
#pragma unroll
        for (int u = 0; u < TK_U; u++) {  // (all loads first: the scan is bound by a load's round trip, not by bandwidth)
            const KeySlot* sp = &ks[min(i0 + (uint32_t)u * nthr, nslots - 1u)];
            tl[u] = *reinterpret_cast<const ulonglong2*>(&sp->tag);
            hi[u] = sp->hi;
        }
        uint32_t total = 0;
#pragma unroll
        for (int u = 0; u < TK_U; u++) {
            const bool ready = i0 + (uint32_t)u * nthr < nslots && (tl[u].x & KS_READY);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(ready);
            if (ready) {
                const uint32_t at = total + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                mine[at] = make_ulonglong2(tl[u].y, hi[u]);
                mslot[at] = i0 + (uint32_t)u * nthr;
            }
            total += (uint32_t)__builtin_popcountll(m);
        }
        for (uint32_t base = 0; base < total; base += 64u) {  // (wave-uniform)
            const bool have = base + lane < total;
            unsigned long long est = 0;
            ulonglong2 key = make_ulonglong2(0, 0);
            uint32_t slot = 0, bin = 0;
            bool full = false;  // the estimate was worked out (its row-0 counter did not already rule the key out)
            if (have) {
                key = mine[base + lane];
                slot = mslot[base + lane];
                CmsKey k;
                est = topk_row0(key.x, key.y, cms, wl2, seed, k);
                full = topk_bin(est) >= lb_bin;
                if (full) est = topk_rest(est, k, cms, depth, wl2);
                bin = full ? topk_bin(est) : 0u;
            }
            if constexpr (M == TK_HIST) {
                if (have) atomicAdd(&lh[bin], 1u);
                if (full) {
                    ks[slot].pad = est;
                    atomicMax(&L.wmax[wave][((slot - w0) / nthr) & (TK_U - 1)], bin + 1u);  // (bin + 1: 0 = no estimated key in the chunk)
                }
            } else {
                const bool sel = full && bin >= lb_bin;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(sel);
                if (m != 0ull) {
                    const uint32_t leader = (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
                    unsigned int at = 0;
                    if (lane == leader) at = atomicAdd(&ctr->ks_rows, (unsigned int)__builtin_popcountll(m));
                    at = (unsigned int)__builtin_amdgcn_readlane((int)at, (int)leader) + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (sel && at < rows_cap) rows[at] = TopkRow{key.x, key.y, est};
                }
            }
        }
        if constexpr (M == TK_HIST) {  // the chunks' largest bins (wave-private LDS: in order behind the atomics above)
            if (lane < (uint32_t)TK_U) {
                const uint32_t c0 = w0 + lane * nthr;
                if (c0 < nslots) chunkmax[c0 >> 6] = (unsigned short)L.wmax[wave][lane];
                L.wmax[wave][lane] = 0u;
            }
        }
    }
    if constexpr (M == TK_HIST) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < TK_BINS; b += blockDim.x)
            if (lh[b]) atomicAdd(&hist[b], lh[b]);
    }
}
template <int M>
__global__ __launch_bounds__(256) void topk_scan_kernel(KeySlot* ks, uint32_t nslots, const unsigned long long* cms, uint32_t depth, uint32_t wl2, uint64_t seed,
                                                        uint32_t lb_bin, unsigned int* hist, unsigned short* chunkmax, TopkRow* rows, uint32_t rows_cap, Counters* ctr) {
    __shared__ unsigned int lh[M == TK_HIST ? TK_BINS : 1];
    __shared__ TopkLds L;
    topk_scan_body<M>(lh, L, blockIdx.x, gridDim.x, ks, nslots, cms, depth, wl2, seed, lb_bin, hist, chunkmax, rows, rows_cap, ctr);
}
// behind TK_HIST + topk_thresh_kernel: the rows of the bins >= min_bin, found through the chunks' largest bins (2 bytes per 64
// slots instead of the 2 KiB they occupy) and the estimates TK_HIST left in the slots
__global__ __launch_bounds__(256) void topk_pick_kernel(const KeySlot* ks, uint32_t nslots, const unsigned short* chunkmax, uint32_t min_bin, TopkRow* rows, uint32_t rows_cap,
                                                        Counters* ctr) {
    const uint32_t lane = __lane_id(), nchunks = nslots >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t cbase = wave * 64u; cbase < nchunks; cbase += nwaves * 64u) {
        const uint32_t cm = cbase + lane < nchunks ? chunkmax[cbase + lane] : 0u;
        // (chunkmax holds bin + 1: a chunk whose keys all estimate to 0 - zero-weight addresses, Bytes * SamplingRate == 0 - is not an
        // empty chunk; with the threshold in bin 0 - fewer than k keys, or a k-th estimate of 0 - those rows are part of the answer)
        unsigned long long todo = __builtin_amdgcn_ballot_w64(cm > min_bin);
        while (todo != 0ull) {  // (wave-uniform)
            const uint32_t c = cbase + (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const KeySlot* sp = &ks[(size_t)c * 64u + lane];
            const ulonglong2 tl = *reinterpret_cast<const ulonglong2*>(&sp->tag);
            const ulonglong2 hp = *reinterpret_cast<const ulonglong2*>(&sp->hi);  // hi, pad (= the estimate)
            const bool sel = (tl.x & KS_READY) && topk_bin(hp.y) >= min_bin;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(sel);
            if (m != 0ull) {
                unsigned int at = 0;
                if (lane == 0) at = atomicAdd(&ctr->ks_rows, (unsigned int)__builtin_popcountll(m));
                at = (unsigned int)__builtin_amdgcn_readfirstlane((int)at) + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (sel && at < rows_cap) rows[at] = TopkRow{tl.y, hp.x, hp.y};
            }
        }
    }
}
// sel[0] = the lowest bin to keep (the bin that holds rank k from the top), sel[1] = rows in the bins >= it, sel[2] = all rows
__device__ __forceinline__ void topk_thresh_body(unsigned int* part, const unsigned int* hist, uint32_t k, unsigned int* sel) {
    constexpr uint32_t PER = TK_BINS / 256;
    unsigned int h[PER], sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        h[q] = hist[threadIdx.x * PER + q];
        sum += h[q];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    // rows in the chunks above this thread's; the chunk that takes the running count to k owns the threshold bin
    unsigned int above = 0, all = 0;
    for (uint32_t t = 0; t < 256; t++) {
        const unsigned int v = part[t];
        all += v;
        if (t > threadIdx.x) above += v;
    }
    if (above < k && above + sum >= k) {
        unsigned int run = above;
        for (int q = (int)PER - 1; q >= 0; q--) {
            run += h[q];
            if (run >= k) {
                sel[0] = threadIdx.x * PER + (uint32_t)q;
                sel[1] = run;
                break;
            }
        }
    }
    if (threadIdx.x == 0) {
        sel[2] = all;
        if (all < k) {  // fewer rows than asked for: everything
            sel[0] = 0u;
            sel[1] = all;
        }
    }
}
__global__ __launch_bounds__(256) void topk_thresh_kernel(const unsigned int* hist, uint32_t k, unsigned int* sel) {
    __shared__ unsigned int part[256];
    topk_thresh_body(part, hist, k, sel);
}

// ---- candidates mode (fa_config.topk_mode = FA_TOPK_CANDIDATES): the launch boundary -----------------------------------------
// The exact mode keeps EVERY address (2 x 2 GiB of sets at BASELINE config 3, one random 64-byte HBM line per address instance
// in the ingest kernel).  The standard Count-Min heavy-hitter contract keeps candidates only; made deterministic per launch:
//   R_t = R_(t-1)  u  { x in launch t : estimate_(t-1)(x) >= theta_(t-1) }
//   theta_t = max( floor of the bin (topk_bin) that holds rank K among the estimates_t of R_t   [0 while |R_t| < K],
//                  N_t >> (log2 slots of the set - 2),   1 )                N_t = total weight = sum of sketch row 0
// estimate_t = the sketch after launch t.  A key joins on its first occurrence AFTER a boundary at which its estimate stood
// above the threshold - independent of the order of records inside a launch, restated in the test oracle
// (topk_candidates).  Behind launch t:  cand_scan_kernel (estimates of R_t -> histogram; sum of row 0) and cand_bits_kernel
// (theta_t; one bit per counter: counter >= theta_t).  The ingest kernel of launch t + 1 tests an address against the bits of
// all its counters - that IS estimate_t(x) >= theta_t - and only then touches the set.
constexpr uint32_t CAND_SCAN_BLOCKS = 256;
// grid (CAND_SCAN_BLOCKS, sketches): y = 0 SrcAddr, 1 DstAddr (a sketch that is off has ks == nullptr)
__global__ __launch_bounds__(256) void cand_scan_kernel(KeySlot* ks0, KeySlot* ks1, uint32_t nslots, const unsigned long long* cms0, const unsigned long long* cms1,
                                                        uint32_t depth, uint32_t wl2, uint64_t seed, CandState* st, unsigned short* chunkmax) {
    __shared__ unsigned int lh[TK_BINS];
    __shared__ TopkLds L;
    KeySlot* ks = blockIdx.y ? ks1 : ks0;
    const unsigned long long* cms = blockIdx.y ? cms1 : cms0;
    if (!ks) return;
    CandState* my = st + blockIdx.y;
    topk_scan_body<TK_HIST>(lh, L, blockIdx.x, gridDim.x, ks, nslots, cms, depth, wl2, seed, 0u, my->hist, chunkmax + (size_t)blockIdx.y * (nslots >> 6), nullptr, 0u, nullptr);
    unsigned long long sum = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ((size_t)1 << wl2); i += (size_t)gridDim.x * blockDim.x) sum += cms[i];
    sum = wave_sum_u64(sum);
    if (__lane_id() == 0 && sum) atomicAdd(&my->total, sum);
}
__global__ __launch_bounds__(256) void cand_bits_kernel(const KeySlot* ks0, const KeySlot* ks1, const unsigned long long* cms0, const unsigned long long* cms1, uint32_t depth,
                                                        uint32_t wl2, uint32_t track, uint32_t floor_shift, CandState* st, uint32_t* bits0, uint32_t* bits1) {
    __shared__ unsigned int part[256];
    __shared__ unsigned int sel[4];
    const KeySlot* ks = blockIdx.y ? ks1 : ks0;
    const unsigned long long* cms = blockIdx.y ? cms1 : cms0;
    uint32_t* bits = blockIdx.y ? bits1 : bits0;
    if (!ks) return;
    CandState* my = st + blockIdx.y;
    topk_thresh_body(part, my->hist, track, sel);  // (every workgroup for itself: 8 KiB of histogram out of L2)
    __syncthreads();
    unsigned long long theta = sel[2] >= track ? topk_bin_floor(sel[0]) : 0ull;
    const unsigned long long fl = my->total >> floor_shift;
    theta = theta > fl ? theta : fl;
    theta = theta ? theta : 1ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        my->theta = theta;
        my->sel[0] = sel[0];
        my->sel[1] = sel[1];
        my->sel[2] = sel[2];
    }
    // one bit per counter; a wave takes 64 consecutive counters per step (coalesced), the ballot is their 64 bits
    const size_t ncnt = (size_t)depth << wl2;
    const uint32_t lane = __lane_id();
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t c0 = wave * 64; c0 < ncnt; c0 += nwaves * 64) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(c0 + lane < ncnt && cms[c0 + lane] >= theta);
        if (lane == 0) *reinterpret_cast<unsigned long long*>(&bits[c0 >> 5]) = m;
    }
}


// keys found by other GPUs / Kafka partitions join this context's candidate set (window close)
__global__ void keyset_merge_kernel(const uint4* keys, uint32_t n, KeySlot* tab, KArgs a) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t k[4] = {keys[i].x, keys[i].y, keys[i].z, keys[i].w};
        keyset_insert(a, tab, k);
    }
}

// ---- synthetic producer ------------------------------------------------------------
__global__ void gen_len_kernel(fa_mock_params g, uint64_t i0, uint32_t n, uint32_t* len) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t tmp[FA_MOCK_MAX_RECORD];
    len[i] = gen_encode(g, i0 + i, tmp);
}
__global__ void gen_write_kernel(fa_mock_params g, uint64_t i0, uint32_t n, const uint32_t* off,
                                 uint8_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t tmp[FA_MOCK_MAX_RECORD];
    uint32_t l = gen_encode(g, i0 + i, tmp);
    uint8_t* p = out + off[i];
    for (uint32_t k = 0; k < l; k++) p[k] = tmp[k];
}

}  // namespace fa
