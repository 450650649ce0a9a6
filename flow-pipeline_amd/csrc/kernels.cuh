// kernels.cuh - the gfx950 kernels of libflowagg.
//
// Hot path (one launch per batch):  tile_kernel<MODE_INGEST,...>
//   wire bytes in HBM --16 B/lane coalesced loads--> LDS tile (32 KiB)
//   -> one record per lane parsed out of LDS (wire.cuh, parse_fast)
//   -> key = (TimeReceived/granule, SrcAS, DstAS, EType)   [create.sh:92-110]
//   -> wave-level duplicate combining (DPP row shifts + readlane)
//   -> per-workgroup LDS hash table (hot keys)  -> device-wide table (HBM/L2, 64-bit atomics)
//   Records the fast parser is not sure about are appended to a deferral list
//   and handled by exotic_kernel with the complete (generic) parser.
//
// Roofline: HBM-bound integer/byte work; algorithmic bytes = wire bytes, read
// once (DESIGN.md "Roofline").  No MFMA anywhere - nothing here is a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gen.cuh"
#include "table.cuh"
#include "wire.cuh"

namespace fa {

constexpr int BLOCK = 256;
constexpr int TILE_BYTES = 32768;  // staged wire bytes per workgroup pass
constexpr int TILE_PAD = 64;       // readable slack behind the staged bytes
constexpr int LDS_SLOTS = 512;     // per-workgroup pre-aggregation slots (20 KiB)
constexpr int LDS_PROBES = 4;

enum { MODE_INGEST = 0, MODE_DECODE = 1 };

struct SpillEntry {
    unsigned long long k0, k1, bytes, packets, count;
};

struct Counters {
    unsigned long long ok, bad, slow, spill_lost, used;
    unsigned int exotic_count, spill_count, rows_count, pad;
};

struct ColumnPtrs {
    uint64_t *time_received, *time_flow_start, *sampling_rate, *bytes, *packets;
    uint32_t *sequence_num, *src_as, *dst_as, *etype, *proto, *src_port, *dst_port;
    uint4 *sampler_address, *src_addr, *dst_addr;
    uint8_t* status;
};

struct KArgs {
    const uint8_t* buf;   // 16-byte aligned device pointer
    const uint32_t* off;  // n+1 offsets
    uint32_t n;
    uint32_t framed;
    uint32_t gran;
    Slot* tab;
    uint32_t mask;
    SpillEntry* spill;
    uint32_t spill_cap;
    Counters* ctr;
    uint32_t* exotic_idx;
    unsigned long long* cms_src;
    unsigned long long* cms_dst;
    uint32_t cms_depth, cms_wl2;
    uint64_t cms_seed;
    ColumnPtrs cols;
};

// ---- sinks ------------------------------------------------------------------
__device__ __forceinline__ void agg_global(const KArgs& a, uint64_t k0, uint64_t k1, uint64_t h,
                                           uint64_t b, uint64_t p, uint64_t c) {
    uint32_t i = (uint32_t)h & a.mask;
    for (int probe = 0; probe < FA_MAX_PROBES; probe++, i = (i + 1) & a.mask) {
        Slot* s = &a.tab[i];
        unsigned long long c0 = s->k0;
        if (c0 == 0) c0 = atomicCAS(&s->k0, 0ull, (unsigned long long)k0);
        if (c0 != 0 && c0 != k0) continue;
        unsigned long long c1 = s->k1;
        if (c1 == 0) {
            c1 = atomicCAS(&s->k1, 0ull, (unsigned long long)k1);
            if (c1 == 0) atomicAdd(&a.ctr->used, 1ull);  // this lane created the group
        }
        if (c1 != 0 && c1 != k1) continue;
        if (b) atomicAdd(&s->bytes, (unsigned long long)b);
        if (p) atomicAdd(&s->packets, (unsigned long long)p);
        atomicAdd(&s->count, (unsigned long long)c);
        return;
    }
    // probe limit: park the partial aggregate; the host grows the table and replays it
    unsigned int j = atomicAdd(&a.ctr->spill_count, 1u);
    if (j < a.spill_cap) {
        a.spill[j] = SpillEntry{k0, k1, b, p, c};
    } else {
        atomicAdd(&a.ctr->spill_lost, 1ull);
    }
}

__device__ __forceinline__ uint64_t cms_hash(uint64_t lo, uint64_t hi, uint64_t seed, uint32_t row) {
    uint64_t h = mix64(lo ^ mix64(seed + 0x9E3779B97F4A7C15ull * (row + 1)));
    return mix64(h ^ hi);
}
__device__ __forceinline__ void cms_add(unsigned long long* cms, uint32_t depth, uint32_t wl2,
                                        uint64_t seed, const uint32_t key[4], uint64_t w) {
    if (w == 0) return;
    uint64_t lo = (uint64_t)key[1] << 32 | key[0], hi = (uint64_t)key[3] << 32 | key[2];
    for (uint32_t r = 0; r < depth; r++) {
        uint64_t h = cms_hash(lo, hi, seed, r);
        atomicAdd(&cms[((size_t)r << wl2) + (size_t)(h >> (64 - wl2))], (unsigned long long)w);
    }
}

__device__ __forceinline__ void store_columns(const ColumnPtrs& c, uint32_t idx, const Rec& r,
                                              uint8_t status) {
    c.time_received[idx] = r.time_received;
    c.time_flow_start[idx] = r.time_flow_start;
    c.sampling_rate[idx] = r.sampling_rate;
    c.bytes[idx] = r.bytes;
    c.packets[idx] = r.packets;
    c.sequence_num[idx] = r.sequence_num;
    c.src_as[idx] = r.src_as;
    c.dst_as[idx] = r.dst_as;
    c.etype[idx] = r.etype;
    c.proto[idx] = r.proto;
    c.src_port[idx] = r.src_port;
    c.dst_port[idx] = r.dst_port;
    c.sampler_address[idx] = make_uint4(r.sampler[0], r.sampler[1], r.sampler[2], r.sampler[3]);
    c.src_addr[idx] = make_uint4(r.src[0], r.src[1], r.src[2], r.src[3]);
    c.dst_addr[idx] = make_uint4(r.dst[0], r.dst[1], r.dst[2], r.dst[3]);
    c.status[idx] = status;
}

template <uint32_t KEYSETS>
constexpr uint32_t cols_for_keysets() {
    uint32_t c = 0;
    if (KEYSETS & FA_KEYS_AS_PAIR) c |= COLS_AS_ROLLUP;
    if (KEYSETS & FA_KEYS_SRCADDR_CMS) c |= COL_SRC_ADDR | COL_BYTES | COL_SAMPLING_RATE;
    if (KEYSETS & FA_KEYS_DSTADDR_CMS) c |= COL_DST_ADDR | COL_BYTES | COL_SAMPLING_RATE;
    return c;
}

// ---- the tile kernel ----------------------------------------------------------
template <int MODE, uint32_t KEYSETS>
__global__ __launch_bounds__(BLOCK) void tile_kernel(KArgs a) {
    constexpr uint32_t COLS = MODE == MODE_DECODE ? (uint32_t)COL_ALL : cols_for_keysets<KEYSETS>();
    __shared__ __attribute__((aligned(16))) uint32_t tile[(TILE_BYTES + TILE_PAD) / 4];
    __shared__ LdsTable<LDS_SLOTS> lt;

    const int tid = threadIdx.x;
    if (MODE == MODE_INGEST && (KEYSETS & FA_KEYS_AS_PAIR)) lds_table_clear(lt);
    __syncthreads();

    uint32_t n_ok = 0;
    const uint32_t ntiles = (a.n + BLOCK - 1) / BLOCK;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t r0 = t * BLOCK;
        const uint32_t nrec = min((uint32_t)BLOCK, a.n - r0);
        const uint32_t last = a.off[r0 + nrec];
        uint32_t cur = 0;
        while (cur < nrec) {
            const uint32_t cbase = a.off[r0 + cur] & ~15u;
            const uint32_t climit = cbase + TILE_BYTES;  // staged window [cbase, climit)
            const uint32_t stage_end = min(last, climit);
            // cooperative, coalesced 16 B/lane copy of the wire bytes into LDS
            {
                const uint4* g = reinterpret_cast<const uint4*>(a.buf + cbase);
                uint4* l = reinterpret_cast<uint4*>(tile);
                const uint32_t nvec = (stage_end - cbase + 15) >> 4;
                for (uint32_t v = tid; v < nvec; v += BLOCK) l[v] = g[v];
            }
            const uint32_t k = cur + tid;
            uint32_t o0 = 0, o1 = 0;
            bool mine = false;
            if (k < nrec) {
                o0 = a.off[r0 + k];
                o1 = a.off[r0 + k + 1];
                mine = o1 <= climit && o1 >= o0 && o0 >= cbase;
            }
            const int nfit = __syncthreads_count(mine);  // offsets are monotone: a prefix fits
            if (nfit == 0) {
                // a single record larger than the LDS tile (or broken offsets): generic path
                if (tid == 0) {
                    unsigned int j = atomicAdd(&a.ctr->exotic_count, 1u);
                    a.exotic_idx[j] = r0 + cur;
                }
                cur += 1;
                __syncthreads();
                continue;
            }
            // ---- parse (divergent: only lanes that own a staged record) ----
            bool sure = false;
            Rec r;
            rec_clear(r);
            if (mine) {
                LdsSrc src{tile};
                uint32_t pos = o0 - cbase, end = o1 - cbase;
                sure = true;
                if (a.framed) {
                    uint32_t pl = 0;
                    sure = frame_fast(window64(src, pos), end - pos, pl);
                    pos += pl;
                }
                if (sure) sure = parse_fast<COLS>(src, pos, end, r);
                if (!sure) {
                    unsigned int j = atomicAdd(&a.ctr->exotic_count, 1u);
                    a.exotic_idx[j] = r0 + k;
                }
            }
            // ---- sink (reconverged: the cross-lane combine needs the whole wave) ----
            if (MODE == MODE_DECODE) {
                if (sure) store_columns(a.cols, r0 + k, r, 0);
            } else {
                n_ok += sure ? 1 : 0;
                if (KEYSETS & FA_KEYS_AS_PAIR) {
                    uint32_t t32 = (uint32_t)r.time_received;  // UInt64 -> DateTime (create.sh:39)
                    uint64_t k0, k1;
                    pack_key(t32 / a.gran, r.src_as, r.dst_as, r.etype, k0, k1);
                    uint64_t h = key_hash(k0, k1);
                    uint64_t b = r.bytes, p = r.packets, c = 1;
                    bool valid = sure;
                    wave_combine<16, 4>(valid, k0, k1, b, p, c);
                    if (valid) {
                        if (!lds_table_add<LDS_SLOTS, LDS_PROBES>(lt, k0, k1, h, b, p, c))
                            agg_global(a, k0, k1, h, b, p, c);
                    }
                }
                if (sure && (KEYSETS & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS))) {
                    uint64_t w = r.bytes * r.sampling_rate;  // viz-ch.json:233 sum(Bytes*SamplingRate)
                    if (KEYSETS & FA_KEYS_SRCADDR_CMS)
                        cms_add(a.cms_src, a.cms_depth, a.cms_wl2, a.cms_seed, r.src, w);
                    if (KEYSETS & FA_KEYS_DSTADDR_CMS)
                        cms_add(a.cms_dst, a.cms_depth, a.cms_wl2, a.cms_seed, r.dst, w);
                }
            }
            cur += nfit;
            __syncthreads();  // tile is overwritten by the next pass
        }
    }
    if (MODE == MODE_INGEST) {
        if (KEYSETS & FA_KEYS_AS_PAIR) {
            __syncthreads();
            for (int i = tid; i < LDS_SLOTS; i += BLOCK) {
                unsigned long long k0 = lt.k0[i], k1 = lt.k1[i], c = lt.count[i];
                if (k0 != 0 && k1 != 0 && c != 0)
                    agg_global(a, k0, k1, key_hash(k0, k1), lt.bytes[i], lt.packets[i], c);
            }
        }
        uint64_t tot = wave_sum_u64(n_ok);
        if (__lane_id() == 0 && tot) atomicAdd(&a.ctr->ok, (unsigned long long)tot);
    }
}

// Deferred records: complete semantics, one record per lane straight from HBM.
template <int MODE, uint32_t KEYSETS>
__global__ __launch_bounds__(BLOCK) void exotic_kernel(KArgs a) {
    const uint32_t cnt = a.ctr->exotic_count;
    for (uint32_t j = blockIdx.x * BLOCK + threadIdx.x; j < cnt; j += gridDim.x * BLOCK) {
        uint32_t idx = a.exotic_idx[j];
        const uint8_t* p = a.buf + a.off[idx];
        const uint8_t* end = a.buf + a.off[idx + 1];
        bool ok = end >= p;
        if (ok && a.framed) ok = frame_generic(p, end);
        Rec r;
        if (ok)
            ok = parse_generic(p, end, r);
        if (!ok) rec_clear(r);
        atomicAdd(&a.ctr->slow, 1ull);
        if (MODE == MODE_DECODE) {
            store_columns(a.cols, idx, r, ok ? 0 : 1);
            continue;
        }
        if (!ok) {
            atomicAdd(&a.ctr->bad, 1ull);
            continue;
        }
        atomicAdd(&a.ctr->ok, 1ull);
        if (KEYSETS & FA_KEYS_AS_PAIR) {
            uint64_t k0, k1;
            pack_key((uint32_t)r.time_received / a.gran, r.src_as, r.dst_as, r.etype, k0, k1);
            agg_global(a, k0, k1, key_hash(k0, k1), r.bytes, r.packets, 1);
        }
        uint64_t w = r.bytes * r.sampling_rate;
        if (KEYSETS & FA_KEYS_SRCADDR_CMS) cms_add(a.cms_src, a.cms_depth, a.cms_wl2, a.cms_seed, r.src, w);
        if (KEYSETS & FA_KEYS_DSTADDR_CMS) cms_add(a.cms_dst, a.cms_depth, a.cms_wl2, a.cms_seed, r.dst, w);
    }
}

// ---- window close ---------------------------------------------------------------
struct Row5m {
    uint32_t date, timeslot, src_as, dst_as, etype, pad;
    unsigned long long bytes, packets, count;
};

// Appends rows whose time bucket lies in [tb_lo, tb_hi) to `rows`.
__global__ void extract_kernel(const Slot* tab, uint32_t nslots, uint32_t gran, uint32_t tb_lo,
                               uint32_t tb_hi, Row5m* rows, uint32_t rows_cap, Counters* ctr) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += gridDim.x * blockDim.x) {
        const Slot& s = tab[i];
        if (s.k0 == 0 || s.k1 == 0 || s.count == 0) continue;
        uint32_t tb, sa, da, et;
        unpack_key(s.k0, s.k1, tb, sa, da, et);
        if (tb < tb_lo || tb >= tb_hi) continue;
        unsigned int j = atomicAdd(&ctr->rows_count, 1u);
        if (j < rows_cap) {
            uint32_t ts = tb * gran;
            rows[j] = Row5m{ts / 86400u, ts, sa, da, et, 0, s.bytes, s.packets, s.count};
        }
    }
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

// ---- synthetic producer ------------------------------------------------------------
__global__ void gen_len_kernel(fa_mock_params g, uint64_t i0, uint32_t n, uint32_t* len) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t tmp[208];
    len[i] = gen_encode(g, i0 + i, tmp);
}
__global__ void gen_write_kernel(fa_mock_params g, uint64_t i0, uint32_t n, const uint32_t* off,
                                 uint8_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t tmp[208];
    uint32_t l = gen_encode(g, i0 + i, tmp);
    uint8_t* p = out + off[i];
    for (uint32_t k = 0; k < l; k++) p[k] = tmp[k];
}

}  // namespace fa
