// ingest.cuh - the ingest kernels: per-record work (parse -> key -> sink), the LDS tuple bins, the wave-tile
// kernel (production), the workgroup-tile kernel (decode path / direct sink), probe and deferred kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sinks.cuh"

#ifndef FA_FOLD_EVERY
#define FA_FOLD_EVERY 4
#endif
#ifndef FA_LT_KEEP
#define FA_LT_KEEP 8u
#endif
namespace fa {

// ---- per-lane work on a staged record (called by every lane of the workgroup) ------
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// after_parse: called by the whole wave between the parse and the sink (the tile buffer is dead from there on,
// unless the sketches use it as scratch)
// Per-wave tallies that reach the device counters once per workgroup (block_counters_add).
struct LaneTally {
    uint32_t ok = 0, direct = 0, second = 0, misfit8 = 0, late = 0;
    uint32_t tick = 0;  // tiles this wave has taken through the sketch path (wave-uniform)
};
// T8: this launch writes compact 8-byte tuples (wave-tile kernel only; table.cuh)
// CANDM: the top-k contract known at compile time (wave-tile variants of their own: 0 exact, 1 candidates) or read from the launch
// arguments (-1)
template <int MODE, uint32_t KEYSETS, uint32_t COLS, bool T8 = false, int CANDM = -1, class Hook = NoHook>
__device__ __forceinline__ void lane_work(const KArgs& a, LdsTable<LDS_SLOTS>& lt, LdsMinutes& lm, uint32_t* part_cnt, const uint32_t* tile,
                                          bool mine, uint32_t pos, uint32_t end, uint32_t rec_idx, uint32_t tb_base,
                                          LaneTally& tally, uint32_t& pmode, uint32_t& lt_seen, uint32_t& lt_hits,
                                          uint4* bins, uint32_t* bin_cnt, uint32_t& fill_out, Hook&& after_parse = Hook(),
                                          CmsLds* cl = nullptr, uint32_t* cms_scratch = nullptr, HotAddrs* hot = nullptr,
                                          uint32_t* wpart_cnt = nullptr, uint32_t* seq = nullptr) {
    constexpr uint32_t TB = bin_cap<T8, bin_line(KEYSETS)>();
    // ---- parse (divergent: only lanes that own a staged record) ----
    bool sure = false, framed_ok = false;
    Rec r;
    rec_clear(r);
    if (mine) {
        LdsSrc src{tile};
        framed_ok = true;
        if (a.framed && !(FA_DBG(a, DBG_NO_FRAME))) {
            uint32_t pl = 0;
            const uint32_t i = pos >> 2;
            framed_ok = frame_short(fa_alignbyte(src.dw(i + 1), src.dw(i), pos), end - pos, pl);
            pos += pl;
        }
        sure = framed_ok && (FA_DBG(a, DBG_NO_PARSE)) != 0;
    }
    // Four tiers, each "exact or not sure", tried from the one the wave has learnt from its previous tiles (pmode, wave-uniform):
    //  0 parse_tmpl<MOCKER>: the template walk over the 13 fields mocker/mocker.go:76-91 sets (+ Proto) - no per-field
    //    question to the wave, no step for fields that producer never emits;
    //  1 parse_tmpl<GOFLOW>: the template walk over the 33 fields GoFlow fills for an sFlow sample (a superset of tier 0);
    //  2 parse_canon<FULL>: every schema field in ascending order and any other field up to number 2047 as generic runs
    //    (pb-ext/flow.pb.go:57-147) - whatever else marshals in field order;
    //  3 parse_fast: any field order, duplicates, unknown fields - IN PLACE, while the bytes still sit in the wave's
    //    LDS tile.  Only what that one cannot decide either (varints above 2^42 in projected fields, groups, 3-byte
    //    tags; broken frames) is deferred to deferred_kernel, which reads it back from HBM one record per lane.
    //  4 parse_seq in front of parse_fast (lean variants: `seq`, 26 words of LDS per wave): a producer that does not marshal
    //    in field order still marshals every record the same way - once most of a tile needed parse_fast the wave takes the
    //    field list of that tile's longest record (seq_learn) and walks the following tiles with it, parse_fast only for what
    //    that refuses; a tile it mostly refuses makes the wave learn again (four times per launch at most).
    // A tier runs only for the lanes the tier before left unsure (wave-uniform branches: a stream of one kind pays one
    // ballot per tile for the tiers it never needs); a wave moves its starting tier up when more than half of a tile
    // needed the next one.
    if (!FA_DBG(a, DBG_NO_PARSE)) {
        // (the walks run on absolute LDS addresses: wire.cuh, LdsAbsSrc)
        LdsAbsSrc src;
        const uint32_t tile_at = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)tile;
        pos += tile_at;
        end += tile_at;
        const uint32_t n_mine = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(framed_ok));
        auto most = [&](bool c) { return 2u * (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c)) > n_mine; };
        if (pmode == 0u && !FA_DBG(a, DBG_LOOP_PARSER)) {
            if (framed_ok) sure = parse_tmpl<COLS, SHAPE_MOCKER>(src, pos, end, r);
        }
        bool need = framed_ok && !sure;
        if (pmode <= 1u && FA_ANY(need) && !FA_DBG(a, DBG_LOOP_PARSER | DBG_NO_SECOND)) {
            if (need) {
                rec_clear(r);
                sure = parse_tmpl<COLS, SHAPE_GOFLOW>(src, pos, end, r);
            }
            if (pmode < 1u && most(need && sure)) pmode = 1u;
        }
        need = framed_ok && !sure;
        if (pmode <= 2u && FA_ANY(need) && !FA_DBG(a, DBG_LOOP_PARSER | DBG_NO_SECOND)) {
            if (need) {
                rec_clear(r);
                sure = parse_canon<COLS, true>(src, pos, end, r);
            }
            if (pmode < 2u && most(need && sure)) pmode = 2u;
        }
        if (pmode == 4u && seq != nullptr && !FA_DBG(a, DBG_NO_SECOND)) {
            if (framed_ok) {
                sure = parse_seq<COLS>(src, pos, end, r, seq + 2, fa_uniform(seq[0]));
                if (!FA_DBG(a, DBG_LOOP_PARSER)) tally.second += sure ? 1u : 0u;
            }
        }
        const bool need_fast = framed_ok && !sure;
        if (FA_ANY(need_fast) && !FA_DBG(a, DBG_NO_SECOND)) {
            if (need_fast) {
                rec_clear(r);
                sure = parse_fast<COLS>(src, pos, end, r);
                if (!FA_DBG(a, DBG_LOOP_PARSER)) tally.second += sure ? 1u : 0u;
                if (!sure) rec_clear(r);
            }
            // most of a tile needed the order-free parser and it worked: start with it from now on
            if (pmode < 3u && most(need_fast && sure)) pmode = 3u;
            if (seq != nullptr && pmode >= 3u && most(need_fast && sure)) {
                // ... behind the field order of the tile's longest record (proto3 omits zero values: the longest has them all)
                pmode = 3u;
                if (fa_uniform(seq[1]) < 4u) {
                    const uint32_t len = (need_fast && sure) ? end - pos : 0u;
                    uint32_t mx = len;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
                    const uint32_t who = (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(len == mx));
                    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)who), e0 = (uint32_t)__builtin_amdgcn_readlane((int)end, (int)who);
                    const uint32_t k = seq_learn(src, p0, e0, seq + 2);  // (every lane, the same record, the same words)
                    seq[0] = k;
                    seq[1] = fa_uniform(seq[1]) + 1u;
                    if (k) pmode = 4u;
                }
            }
        }
    }
    if (mine && !sure) {  // (one counter atomic per wave: the compiler folds the lanes' adds - s_bcnt1 + mbcnt)
        const KArgs ca = cold_args();
        unsigned int j = atomicAdd(&ca.ctr->retry_count[ca.par], 1u);
        ca.retry_idx[j] = rec_idx;
    }
    after_parse();
    // ---- sink ----
    if (MODE == MODE_DECODE) {
        if (sure) store_columns(a.cols, rec_idx, r, 0);
        return;
    }
    tally.ok += sure ? 1 : 0;
    if (FA_DBG(a, DBG_NO_SINK)) {
        tally.ok += (uint32_t)(r.time_received ^ r.bytes ^ r.packets ^ r.src_as ^ r.dst_as ^ r.etype) & 1;
        return;
    }
    const uint32_t t32 = (uint32_t)r.time_received;  // UInt64 -> DateTime (create.sh:39)
    const uint32_t tb = time_bucket(a, t32);
    if (a.late_below != 0u) tally.late += (sure && tb < a.late_below) ? 1u : 0u;  // (its flows_5m window was closed before it arrived; aggregated all the same.  Nothing closed yet: three instructions a record less)
    // state of the sketch path between its two halves (below: in front of and behind the flows_5m sink)
    uint64_t cw = 0, ws = 0, wd = 0, slo = 0, shi = 0, dlo = 0, dhi = 0, sh1 = 0, sh2 = 0, dh1 = 0, dh2 = 0;
    bool on_s = false, on_d = false, vs = false, vd = false;
    const bool keys_on = !(FA_DBG(a, DBG_NO_KEYSET));
    const bool cand = CANDM >= 0 ? CANDM == 1 : (a.cand_src != nullptr || a.cand_dst != nullptr);  // (wave-uniform: kernel arguments)
    uint32_t cw0s = 0, cw0d = 0;
    KsProbe ps{}, pd{};
    if (KEYSETS & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS)) {
        // (this half - folds, hashes, hot-address cache, the distinct sets' home-slot loads ISSUED - runs in front of the
        // flows_5m sink: a streaming batch probes cold lines, and the sink's LDS work is what hides their way to HBM;
        // the sketch tuples and the look at what the probes returned follow behind the sink)
        // lanes of a wave that carry the same address (heavy hitters) are folded first: one sketch update and
        // one distinct-set probe per address and wave (wave-tile kernel: its parsed tile buffer is the scratch).
        // Order of work: fold both addresses, hash them, ISSUE the distinct-set probes of both (global loads), then
        // the sketch updates (LDS work that hides the probes' latency), then look at what the probes returned.
        cw = r.bytes * r.sampling_rate;  // viz-ch.json:233 sum(Bytes*SamplingRate), UInt64 wrap
        on_s = ks_on<KEYSETS>(a, FA_KEYS_SRCADDR_CMS);
        on_d = ks_on<KEYSETS>(a, FA_KEYS_DSTADDR_CMS);
        slo = (uint64_t)r.src[1] << 32 | r.src[0];
        shi = (uint64_t)r.src[3] << 32 | r.src[2];
        dlo = (uint64_t)r.dst[1] << 32 | r.dst[0];
        dhi = (uint64_t)r.dst[3] << 32 | r.dst[2];
        ws = wd = cw;
        vs = sure && on_s;
        vd = sure && on_d;
        // (the folds run on every FA_FOLD_EVERY-th tile of a wave: with the hot-address cache in front of the sink what a
        // fold still finds is mostly an address that deserves an entry there - the admission signal - and that can wait
        // a few tiles; equal addresses of an unfolded tile leave as separate tuples (+1 % tuples), sums commute.  Same-box:
        // every tile 0.999 ms per launch of the config-3 shape, every 4th 0.984, every 8th 0.984.)
        const bool fold_now = bins && (__builtin_amdgcn_readfirstlane((int)tally.tick++) % FA_FOLD_EVERY) == 0;
        if (fold_now && on_s) wave_fold_lds(const_cast<uint32_t*>(tile), vs, slo, shi, ws);
        if (fold_now && on_d) wave_fold_lds(const_cast<uint32_t*>(tile), vd, dlo, dhi, wd);
        if (on_s) cms_hash2(slo, shi, a.cms_seed, sh1, sh2);
        if (on_d) cms_hash2(dlo, dhi, a.cms_seed, dh1, dh2);
        if (hot && !(FA_DBG(a, DBG_NO_HOT))) {  // heavy hitters: one LDS add, nothing else (an address may move in once a wave has seen it twice)
            if (vs && hot_add(*hot, 0u, slo, shi, sh1, ws, ws != cw)) vs = false;
            if (vd && hot_add(*hot, 1u, dlo, dhi, dh1, wd, wd != cw)) vd = false;
        }
        if (cand) {  // candidates mode: the row-0 word of the candidate bits instead of the sets' home slots (L2-resident)
            if (vs && keys_on) cw0s = cand_word(a.cand_src, a.cms_wl2, 0, cms_column(cms_key(sh1, sh2, a.cms_wl2), 0, a.cms_wl2));
            if (vd && keys_on) cw0d = cand_word(a.cand_dst, a.cms_wl2, 0, cms_column(cms_key(dh1, dh2, a.cms_wl2), 0, a.cms_wl2));
        } else {
            if (vs && keys_on) ps = keyset_probe(a, a.ks_src, sh1);
            if (vd && keys_on) pd = keyset_probe(a, a.ks_dst, dh1);
        }
    }
    if (ks_on<KEYSETS>(a, FA_KEYS_AS_PAIR)) {
        // the two key words and the 32-bit key hash (three quarter-rate multiplies) are only needed by the hot-key table, the wide
        // tuple's partition and the direct path: a compact-tuple wave that has given the hot-key table up never computes them
        // (the key words are packed where they are used: hoisted in front of the branches they were six instructions a record)
        uint64_t k0 = 0, k1 = 0;
        const bool lt_on = lt_seen != 0xffffffffu && !(FA_DBG(a, DBG_NO_LDS_TABLE));  // wave-uniform
        uint32_t h = 0;
        if (!T8 || lt_on) {
            pack_key(tb, r.src_as, r.dst_as, r.etype, k0, k1);
            h = key_hash(k0, k1);
        }
        const uint64_t b = r.bytes, p = r.packets, c = 1;
        bool pending = sure;
        // hot-key table: worth its LDS atomics only while it absorbs records.  Every wave keeps score (ballots:
        // wave-uniform, no LDS traffic) and stops offering records once fewer than 1 in 8 of its first 256 stuck
        // (64 k uniform AS pairs never do; the mocker's 9 groups always do).  lt_seen == ~0u: switched off.
        if (lt_on) {
            if (pending) pending = !lds_table_add<LDS_SLOTS, LDS_PROBES>(lt, k0, k1, h, b, p, c);
            lt_seen += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sure));
            lt_hits += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sure && !pending));
            if (lt_seen >= 256u) {
                if (lt_hits * FA_LT_KEEP < lt_seen) lt_seen = 0xffffffffu;
                else lt_seen = lt_hits = 0;
            }
        }
        // tuple path: one tuple to this workgroup's private segment of the key's partition
        uint32_t fill_part = 0xffffffffu;  // wave-tile kernel: the bin this lane has just filled
        bool fits = false;
        uint32_t part = 0;
        uint4 tv = make_uint4(0, 0, 0, 0);  // (compact tuples: .x, .y)
        if (pending && a.seg) {
            const uint32_t tbr = tb - tb_base;
            if (T8) {
                fits = t8_fits(r.src_as, r.dst_as, tbr, b, p, r.etype);
                const uint2 tc = t8_pack(r.src_as, r.dst_as, (uint32_t)b, (uint32_t)p, tbr, r.etype, tb_base, part);
                tv.x = tc.x;
                tv.y = tc.y;
                if (FA_ANY(!fits)) tally.misfit8 += (!fits && tup16_fits(tbr, b, p, r.etype)) ? 1u : 0u;  // (format feedback; rare)
            } else {
                fits = tup16_fits(tbr, b, p, r.etype);
                part = h >> (32 - a.plog2);
                tv = tup16_pack(r.src_as, r.dst_as, (uint32_t)b, (uint32_t)p, tbr, r.etype);
            }
        }
        if (bins) {
            // wave-tile kernel: the tuple waits in the workgroup's LDS bin of its partition; the lane that takes the last
            // slot of a bin sends the whole bin off as one full store unit (bins_flush)
            // (acquire: the tuple write must not move above the claim - the previous occupants of the bin are read by
            // the flusher until it resets the word; release: the tuple is written before it counts)
            auto claim = [&]() {
                const uint32_t slot = __hip_atomic_fetch_add(&bin_cnt[part], 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & 0xffffu;  // low half: slots taken, high half: slots written
                if (slot < TB) {
                    if (T8) reinterpret_cast<uint2*>(bins)[part * TB + slot] = make_uint2(tv.x, tv.y);
                    else bins[part * TB + slot] = tv;
                    __hip_atomic_fetch_add(&bin_cnt[part], 0x10000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    fill_part = slot == TB - 1 ? part : fill_part;
                    pending = false;
                }
            };
            if (pending && fits) {
                claim();
                // A tuple that meets a closing bin (full, not yet reset by the wave that filled it: ~7 % of the tuples on
                // BASELINE config 2, where 24 waves share 256 bins) leaves as a single store into the back part of its
                // segment.  (Round 3 measured the alternative - let the tile's full bins go first, then claim again:
                // singles 7.4 % -> 1.5 %, kernel +1.5 %: the L2 combines the back parts' stores, the second flush pass costs
                // more than they do.)
                if (pending) {
                    const uint32_t ob = lds_add_rtn_u32(&part_cnt[part], 0x10000u) >> 16;
                    if (ob < a.capb) {
                        const size_t at = (size_t)part * a.region + (size_t)blockIdx.x * a.capq + (a.capq - 1u - ob);
                        if (FA_DBG(a, DBG_TUPLE_LOCAL)) {
                            reinterpret_cast<uint2*>(a.seg)[(size_t)blockIdx.x * 512u + (at & 511u)] = make_uint2(tv.x, tv.y);
                        } else if (!(FA_DBG(a, DBG_NO_TUPLE_STORE | DBG_NO_SINGLES))) {
                            if (T8) reinterpret_cast<uint2*>(a.seg)[at] = make_uint2(tv.x, tv.y);
                            else a.seg[at] = tv;
                        }
                        pending = false;
                    } else {
                        lds_add_u32(&part_cnt[part], 0xffff0000u);  // (not stored: -1 on the back count, which never wraps into stored tuples)
                    }
                }
            }
        } else if (!T8 && pending && fits) {  // workgroup-tile kernel: straight into the segment
            const uint32_t q = atomicAdd(&part_cnt[part], 1u);
            if (q < a.capq) {
                if (!(FA_DBG(a, DBG_NO_TUPLE_STORE))) a.seg[(size_t)part * a.region + (size_t)blockIdx.x * a.capq + q] = tv;
                pending = false;
            }
        }
        fill_out = fill_part;  // full bins leave in bins_flush(), which the wave-tile kernel runs right after this call
        // direct path (what is left): device-wide table, one atomic line transaction per record
        if (__builtin_amdgcn_ballot_w64(pending) != 0ull && !(FA_DBG(a, DBG_NO_GLOBAL))) {  // wave-uniform
            Slot* sp = nullptr;
            const KArgs ca = cold_args();  // (table, mask, spill buffer, counters: not kept in SGPRs for the tiles that never get here)
            if (pending) {
                tally.direct++;
                if (T8 && !lt_on) {
                    pack_key(tb, r.src_as, r.dst_as, r.etype, k0, k1);
                    h = key_hash(k0, k1);
                }
                sp = table_find_or_claim(ca, k0, k1, h);
                if (!sp) spill_park(ca, k0, k1, b, p, c);
            }
            quad_atomic_update(sp, b, p, c);
        }
    }
    if (KEYSETS & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS)) {
        if (!(FA_DBG(a, DBG_NO_CMS))) {
            if (cl && (CANDM >= 0 || a.cseg)) {  // scatter sink: no atomics (whole wave: the bin flushes need every lane)
                // (bin lists: the wave's tile buffer is dead by now - behind the 768 bytes the folds used)
                uint32_t* list = const_cast<uint32_t*>(tile) + 256;
                if (on_s) cms_scatter(a, *cl, list, 0u, vs, ws, sh1, sh2);
                if (on_d) cms_scatter(a, *cl, list, 1u, vd, wd, dh1, dh2);
            } else {
                if (vs) cms_add(a.cms_src, a.cms_depth, a.cms_wl2, a.cms_seed, r.src, ws, a.cms_nrep);
                if (vd) cms_add(a.cms_dst, a.cms_depth, a.cms_wl2, a.cms_seed, r.dst, wd, a.cms_nrep);
            }
        }
        if (keys_on && !cand) keyset_finish2(a, vs, ps, sh1, slo, shi, vd, pd, dh1, dlo, dhi);
        if (keys_on && cand) {
            // the few addresses whose estimate stood above the threshold at the last boundary join the candidates.  What this costs
            // is the test of EVERY address instance, not the set (same-box ablation, measurement build, 16.67 M records per launch:
            // no test at all 0.940 ms; the row-0 word of every instance +0.117; all four rows' words loaded ahead of the sink
            // +0.216 - slower than the exact mode's 1.147; the rows behind the first and the look into the set for what passes
            // row 0: +0.04).  Where the word comes from does not matter either: a 4 KiB fold of the row-0 bits asked first, out of
            // the CU's L1, measured 1.1645 against the exact mode's 1.1544 ms - an instance pays the test's instructions and the
            // values it keeps alive across the sink in a kernel that already lives on 128 VGPRs.  So: one word early, the rest late.
            const bool is = vs && cand_pass(a.cand_src, a.cms_depth, a.cms_wl2, cms_key(sh1, sh2, a.cms_wl2), cw0s);
            const bool id = vd && cand_pass(a.cand_dst, a.cms_depth, a.cms_wl2, cms_key(dh1, dh2, a.cms_wl2), cw0d);
            if (FA_ANY(is || id) && !FA_DBG(a, DBG_CAND_NO_SET)) {
                if (is) keyset_insert_h(a, a.ks_src, slo, shi, sh1);
                if (id) keyset_insert_h(a, a.ks_dst, dlo, dhi, dh1);
            }
        }
    }
    if (KEYSETS & FA_KEYS_WIDE) wide_sink_wave<KEYSETS>(a, lm, r, sure, tb, tb_base, wpart_cnt);
}

// End-of-kernel counters: one global atomic per WORKGROUP.  All waves of the grid finish at about the same
// time, and same-address atomics serialize at the memory side: one atomic per wave (8192 of them) was a
// ~30 us tail on a 0.4 ms launch.
__device__ __forceinline__ void block_counters_add(uint32_t* lds4, Counters* ctr, const LaneTally& t) {
    if (threadIdx.x < 5) lds4[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t ok = (uint32_t)wave_sum_u64(t.ok), direct = (uint32_t)wave_sum_u64(t.direct);
    const uint32_t second = (uint32_t)wave_sum_u64(t.second), mis = (uint32_t)wave_sum_u64(t.misfit8);
    const uint32_t late = __builtin_amdgcn_ballot_w64(t.late != 0u) != 0ull ? (uint32_t)wave_sum_u64(t.late) : 0u;
    if (__lane_id() == 0) {
        if (ok) atomicAdd(&lds4[0], ok);
        if (direct) atomicAdd(&lds4[1], direct);
        if (second) atomicAdd(&lds4[2], second);
        if (mis) atomicAdd(&lds4[3], mis);
        if (late) atomicAdd(&lds4[4], late);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (lds4[0]) atomicAdd(&ctr->ok, (unsigned long long)lds4[0]);
        if (lds4[1]) atomicAdd(&ctr->direct, (unsigned long long)lds4[1]);
        if (lds4[2]) atomicAdd(&ctr->retried, (unsigned long long)lds4[2]);
        if (lds4[3]) atomicAdd(&ctr->misfit8, (unsigned long long)lds4[3]);
        if (lds4[4]) atomicAdd(&ctr->late, (unsigned long long)lds4[4]);
    }
}

// ---- probe: where in time does this batch sit? ---------------------------------------------
// 64 evenly spaced records are decoded by the 64 lanes of a wave; tb_base = (smallest time bucket seen) - 2, so
// that the 4-bit relative bucket of the tuple path covers the batch (Kafka partitions are close to time-ordered;
// records outside [tb_base, tb_base+16) take the direct path).  Wave-uniform result, the same for every wave that
// asks: the wave-tile kernel evaluates it in its prologue; the workgroup-tile kernel gets it from probe_kernel.
__device__ __forceinline__ uint32_t probe_tb_base(const KArgs& a) {
    const uint32_t ln = __lane_id();
    const uint32_t idx = a.n <= 64 ? ln : (uint32_t)(((uint64_t)ln * (a.n - 1)) / 63u);
    uint32_t lo = 0xffffffffu;
    if (idx < a.n) {
        uint32_t pos = a.off[idx], end = a.off[idx + 1];
        // tb_base is only a hint (it decides which records may use the tuple path, never a result), so
        // the order-free fast parser is enough: samples it is not sure about are skipped
        GlobalSrc src{reinterpret_cast<const uint32_t*>(a.buf)};
        bool ok = end >= pos;
        if (ok && a.framed) {
            uint32_t pl = 0;
            ok = frame_fast(window64(src, pos), end - pos, pl);
            pos += pl;
        }
        if (ok) {
            // what proto.Marshal emits (mocker.go:97): [Type 08 xx] then TimeReceived 10 <varint> - one or two
            // cache-resident windows instead of a walk over the whole record; anything else: the general parser
            uint64_t w = window64(src, pos);
            if ((w & 0x80ffu) == 0x0008u) {
                pos += 2;
                w = window64(src, pos);
            }
            uint32_t vl;
            uint64_t val;
            if ((w & 0xffu) == 0x10u && varint6(w >> 8, vl, val) && pos + 1 + vl <= end) {
                lo = time_bucket(a, (uint32_t)val);
            } else {
                Rec r;
                rec_clear(r);
                if (parse_fast<COL_TIME_RECEIVED>(src, pos, end, r)) lo = time_bucket(a, (uint32_t)r.time_received);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
    lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    return lo == 0xffffffffu ? 0u : (lo > 2u ? lo - 2u : 0u);
}
__global__ __launch_bounds__(64) void probe_kernel(KArgs a) {
    const uint32_t tb = probe_tb_base(a);
    if (threadIdx.x == 0) a.ctr->tb_base = tb;
}

// ---- the tile kernel ----------------------------------------------------------
// Persistent workgroups; tile = 256 consecutive records (one per lane).  The wire
// bytes of tile t+1 stream into the second LDS buffer (async DMA) while tile t is
// parsed and aggregated, so the HBM latency hides behind the integer work.
// The descriptor is two loads (lo, hi) whose values must NOT be looked at before the tile's turn comes: any
// arithmetic on them right after the loads makes the compiler wait for them - and, vmcnt being in-order, for
// the DMA issued just before - and only then issue the per-lane offset loads: two serialized memory round
// trips per tile (this cost 25 % of the staging bandwidth, tools/read_bench2.hip).  fits() is evaluated
// when the tile is current.
struct TileDesc {
    uint32_t r0, nrec, lo, hi;  // records [r0,r0+nrec), wire bytes [lo,hi)
};
__device__ __forceinline__ TileDesc tile_desc(const KArgs& a, uint32_t t, uint32_t ntiles) {
    TileDesc d{0, 0, 0, 0};
    if (t < ntiles) {
        d.r0 = t * a.tile_recs;
        d.nrec = min(a.tile_recs, a.n - d.r0);
        // the indices are laundered through VGPRs: for provably uniform addresses the compiler moves the loaded
        // values to SGPRs at once (v_readfirstlane right behind the loads = the same premature wait)
        uint32_t i0 = d.r0, i1 = d.r0 + d.nrec;
        asm volatile("" : "+v"(i0), "+v"(i1));
        d.lo = a.off[i0];
        d.hi = a.off[i1];
    }
    return d;
}
// the bounds of the tile whose turn has come, as wave-uniform scalars
__device__ __forceinline__ TileDesc tile_current(const TileDesc& d) {
    return TileDesc{d.r0, d.nrec, (uint32_t)__builtin_amdgcn_readfirstlane((int)d.lo), (uint32_t)__builtin_amdgcn_readfirstlane((int)d.hi)};
}
template <int BYTES>
__device__ __forceinline__ bool tile_fits(const TileDesc& d) {  // whole tile fits one LDS buffer (the normal case)
    return d.nrec != 0 && d.hi >= d.lo && (d.hi - (d.lo & ~15u)) <= (uint32_t)BYTES;
}

// Persistent workgroups; tile = up to 256 consecutive records (one per lane) staged
// in ONE LDS buffer.  LDS bounds the number of records a CU can hold, and the parse
// is a long dependent chain per record, so the LDS goes to as many co-resident
// workgroups as possible (6 per CU = 6 waves/SIMD): while one workgroup waits for
// its DMA, the others parse.  (A double-buffered variant with 3 workgroups/CU
// staged at 4.0 TB/s but left the parse latency-bound at 3 waves/SIMD.)
template <int MODE, uint32_t KEYSETS>
__global__ __launch_bounds__(BLOCK) void tile_kernel(KArgs a) {
    constexpr uint32_t COLS = MODE == MODE_DECODE ? (uint32_t)COL_ALL : cols_for_keysets<KEYSETS>();
    __shared__ __attribute__((aligned(16))) uint32_t tile[TILE_STRIDE / 4];
    __shared__ LdsTable<LDS_SLOTS> lt;
    __shared__ uint32_t part_cnt[NPART_MAX];  // tuples this workgroup appended per key partition
    __shared__ LdsMinutes lm;                 // per-minute series pre-aggregation (KS_ALL variant only)

    const uint32_t tid = threadIdx.x;
    if (MODE == MODE_INGEST && (KEYSETS & FA_KEYS_AS_PAIR)) {
        lds_table_clear(lt);
        for (int i = tid; i < NPART_MAX; i += BLOCK) part_cnt[i] = 0;
    }
    if (MODE == MODE_INGEST && (KEYSETS & FA_KEYS_MINUTE_SERIES)) lds_minutes_clear(lm);
    const uint32_t tb_base = MODE == MODE_INGEST ? a.ctr->tb_base : 0u;

    LaneTally tally;
    uint32_t lt_seen = 0, lt_hits = 0, no_fill = 0, pmode = 0;
    const uint32_t ntiles = (a.n + a.tile_recs - 1) / a.tile_recs;
    const uint32_t stride = gridDim.x;
    uint32_t t = blockIdx.x;
    TileDesc cur = tile_current(tile_desc(a, t, ntiles));
    uint32_t o0 = 0, o1 = 0;  // this lane's record of the current tile
    if (t < ntiles && tid < cur.nrec) {
        o0 = a.off[cur.r0 + tid];
        o1 = a.off[cur.r0 + tid + 1];
    }
    __syncthreads();  // LDS table cleared

    const bool timing = (FA_DBG(a, DBG_TIMING)) != 0 && tid == 0;
    uint32_t tm_wait = 0, tm_work = 0, tm_tiles = 0;
    const uint32_t tm_start = timing ? (uint32_t)clock64() : 0u;
    for (; t < ntiles; t += stride) {
        const bool cur_sane = cur.hi >= cur.lo && cur.hi <= a.len;  // (bounds come from the caller: nothing beyond the buffer is ever read)
        const bool cur_fits = cur_sane && tile_fits<TILE_BYTES>(cur);
        const uint32_t tm0 = timing ? (uint32_t)clock64() : 0u;
        // (1) stream this tile's wire bytes into LDS (async DMA) ...
        if (cur_fits) {
            // nt: the wire bytes are read exactly once; keeping them out of the way of the L2's open tuple lines
            // is worth 11 % of the launch (MI355X, tools/knobs.sh FA_DEBUG_FLAGS=512)
            if (FA_DBG(a, DBG_DMA_NO_NT)) dma_to_lds<0>(a.buf + (cur.lo & ~15u), cur.hi - (cur.lo & ~15u), tile);
            else dma_to_lds<2>(a.buf + (cur.lo & ~15u), cur.hi - (cur.lo & ~15u), tile);
        }
        // ... and meanwhile fetch the next tile's descriptor and offsets
        const TileDesc nxt = tile_desc(a, t + stride, ntiles);
        uint32_t n0 = 0, n1 = 0;
        if (tid < nxt.nrec) {
            n0 = a.off[nxt.r0 + tid];
            n1 = a.off[nxt.r0 + tid + 1];
        }
        dma_wait_all();
        __syncthreads();
        const uint32_t tm1 = timing ? (uint32_t)clock64() : 0u;

        // (2) parse + aggregate out of LDS
        if (cur_fits) {
            const uint32_t cbase = cur.lo & ~15u;
            const bool mine = tid < cur.nrec && o1 >= o0 && o0 >= cur.lo && o1 <= cur.hi;
            if (tid < cur.nrec && !mine) {  // broken offsets: let the generic path judge it
                unsigned int j = atomicAdd(&a.ctr->exotic_count[a.par], 1u);
                a.exotic_idx[j] = cur.r0 + tid;
            }
            lane_work<MODE, KEYSETS, COLS>(a, lt, lm, part_cnt, tile, mine, o0 - cbase, o1 - cbase, cur.r0 + tid, tb_base, tally, pmode, lt_seen, lt_hits, nullptr, nullptr, no_fill);
        } else if (!cur_sane) {  // broken bounds: every record of the tile to the generic path (which checks them one by one)
            if (tid < cur.nrec) {
                unsigned int j = atomicAdd(&a.ctr->exotic_count[a.par], 1u);
                a.exotic_idx[j] = cur.r0 + tid;
            }
        } else {
            // rare: the tile's bytes exceed the LDS buffer (big records): stage it in passes
            uint32_t done = 0;
            while (done < cur.nrec) {
                const uint32_t first = a.off[cur.r0 + done];
                const uint32_t cbase = first & ~15u;
                const uint32_t climit = cbase + TILE_BYTES;
                const uint32_t stage_end = min(cur.hi, climit);
                if (stage_end > cbase) dma_to_lds(a.buf + cbase, stage_end - cbase, tile);
                const uint32_t k = done + tid;
                uint32_t p0 = 0, p1 = 0;
                bool mine = false;
                if (k < cur.nrec) {
                    p0 = a.off[cur.r0 + k];
                    p1 = a.off[cur.r0 + k + 1];
                    mine = p1 <= climit && p1 >= p0 && p0 >= cbase && p1 <= cur.hi;
                }
                dma_wait_all();
                const int nfit = __syncthreads_count(mine);  // offsets are monotone: a prefix fits
                if (nfit == 0) {
                    // one record larger than the LDS buffer (or broken offsets): generic path
                    if (tid == 0) {
                        unsigned int j = atomicAdd(&a.ctr->exotic_count[a.par], 1u);
                        a.exotic_idx[j] = cur.r0 + done;
                    }
                    done += 1;
                } else {
                    lane_work<MODE, KEYSETS, COLS>(a, lt, lm, part_cnt, tile, mine, p0 - cbase, p1 - cbase, cur.r0 + k, tb_base, tally, pmode, lt_seen, lt_hits, nullptr, nullptr, no_fill);
                    done += nfit;
                }
                __syncthreads();  // the buffer is restaged by the next pass
            }
        }
        __syncthreads();  // everyone is done reading the tile
        if (timing) {
            const uint32_t tm2 = (uint32_t)clock64();
            tm_wait += tm1 - tm0;
            tm_work += tm2 - tm1;
            tm_tiles++;
        }
        cur = tile_current(nxt);
        o0 = n0;
        o1 = n1;
    }
    if (timing) {
        atomicAdd(&a.ctr->t_wait, (unsigned long long)tm_wait);
        atomicAdd(&a.ctr->t_work, (unsigned long long)tm_work);
        atomicAdd(&a.ctr->t_tiles, (unsigned long long)tm_tiles);
        atomicAdd(&a.ctr->t_total, (unsigned long long)((uint32_t)clock64() - tm_start));
    }
    if (MODE == MODE_INGEST) {
        if (KEYSETS & FA_KEYS_MINUTE_SERIES) {
            __syncthreads();
            if (tid < LDS_MINUTES && lm.key[tid] != 0 && lm.c[tid] != 0) {
                WKey k;
                wkey_pack(WK_MINUTE, 0, 0, 0, lm.key[tid] - 1u, 0, k);
                wagg_global(wargs(a), k, lm.w[tid], 0, lm.c[tid]);
            }
        }
        if (KEYSETS & FA_KEYS_AS_PAIR) {
            __syncthreads();
            // hot-key table -> device-wide table, one atomic line transaction per group (uniform trip count:
            // the quad rounds need the whole wave)
            for (int i0 = 0; i0 < LDS_SLOTS; i0 += BLOCK) {
                const int i = i0 + tid;
                Slot* sp = nullptr;
                unsigned long long b = 0, p = 0, c = 0;
                if (i < LDS_SLOTS) {
                    const unsigned long long k0 = lt.k0[i], k1 = lt.k1[i];
                    b = lt.bytes[i];
                    p = lt.packets[i];
                    c = lt.count[i];
                    if (k0 != 0 && k1 != 0 && c != 0) {
                        sp = table_find_or_claim(a, k0, k1, key_hash(k0, k1));
                        if (!sp) spill_park(a, k0, k1, b, p, c);
                    }
                }
                quad_atomic_update(sp, b, p, c);
            }
        }
        if ((KEYSETS & FA_KEYS_AS_PAIR) && a.seg) {
            for (int i = tid; i < (1 << a.plog2); i += BLOCK)
{
                a.seg_counts[(size_t)i * a.nwg + blockIdx.x] = min(part_cnt[i], a.capq);
                a.seg_counts[((size_t)NPART_MAX + i) * a.nwg + blockIdx.x] = 0;
            }
        }
        block_counters_add(part_cnt, a.ctr, tally);  // (part_cnt has been written out: reused as scratch)
    }
}

// ---- the wave-tile kernel ---------------------------------------------------------------------------
// The production ingest kernel of the scatter sink.  Same per-record work as tile_kernel, different
// residency: 2 workgroups of 12 waves per CU (sketch variants: one of 16); every WAVE stages its own tile of <= 64 records into a private
// LDS buffer (its next DMA is issued the moment the tile is consumed) and parses it - there is no workgroup
// barrier anywhere in the steady state.  The LDS this frees (the 256-thread kernel spends all of it on
// co-resident tiles) holds the tuple bins: a tuple waits in the 8-slot bin of its key partition, and a full bin
// leaves as ONE aligned 128-byte line (lane_work), instead of as eight 16-byte stores whose cache line is
// evicted from the L2 long before its neighbours arrive (DESIGN.md "Measurements").  A segment therefore has a
// front part of whole lines and a back part for the odd tuples (bin leftovers at the end of the launch, tuples
// that met a bin on its way out).  The workgroup's segments are 3x longer than tile_kernel's, which also
// suits agg_kernel's 64-lane loads.
// LDS of the Count-Min scatter sink: only the kernel variants that serve a sketch carry it
template <bool ON>
struct CmsLdsOpt {
    CmsLds v;
    __device__ __forceinline__ CmsLds* get() { return &v; }
};
template <>
struct CmsLdsOpt<false> {
    __device__ __forceinline__ CmsLds* get() { return nullptr; }
};
// workgroup size of a variant: the sketch variants need 32 KiB of LDS more per workgroup (CmsLds) than two
// workgroups per CU leave - they run ONE workgroup of 16 waves per CU (the same 16 waves per CU)
template <uint32_t KEYSETS>
constexpr int wtile_block() { return wt_lean(KEYSETS) ? WBLOCK : WBLOCK_CMS; }
// ... and their tile buffers are 256 bytes shorter (62 instead of 64 mocker-sized records; the LDS goes to the sketch
// bins and the hot-address cache - the whole 160 KiB of the CU are spoken for)
template <uint32_t KEYSETS>
constexpr int wtile_stride() { return wt_stride(KEYSETS); }
template <bool ON>
struct HotAddrsOpt {
    HotAddrs v;
    __device__ __forceinline__ HotAddrs* get() { return &v; }
};
template <>
struct HotAddrsOpt<false> {
    __device__ __forceinline__ HotAddrs* get() { return nullptr; }
};

// SEQ: the variant with the learnt-field-order tier (lane_work tier 4) - a kernel of its own, launched while the counters say that
// most records need the order-free parser (maintain_host.inc, format_feedback): in the common kernel the tier's code cost the
// streams that never use it (GoFlow shape: +1.3 % per launch, same box, through the register allocation alone).
template <uint32_t KEYSETS, bool T8, bool SEQ = false, int CANDM = -1>
__global__ __launch_bounds__(wtile_block<KEYSETS>(), wt_lean(KEYSETS) ? 6 : 4) void wtile_kernel(KArgs a) {
    constexpr uint32_t BL = bin_line(KEYSETS);
    constexpr uint32_t TB = bin_cap<T8, BL>();
    constexpr uint32_t COLS = cols_for_keysets<KEYSETS>();
    constexpr int WBLOCK = wtile_block<KEYSETS>();  // (shadows the namespace constant inside this kernel)
    constexpr bool HAS_CMS = (KEYSETS & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS)) != 0;
    constexpr int WAVES = WBLOCK / 64;
    constexpr int WT_STRIDE = wtile_stride<KEYSETS>();  // (shadows the namespace constant inside this kernel)
    __shared__ CmsLdsOpt<HAS_CMS> cms_lds;
    __shared__ HotAddrsOpt<HAS_CMS> hot_lds;
    __shared__ uint32_t cms_scratch_all[HAS_CMS ? WAVES * 16 : 1];
    __shared__ __attribute__((aligned(16))) uint32_t tiles[WAVES * WT_STRIDE / 4];
    __shared__ __attribute__((aligned(16))) uint4 bins[NPART_MAX * BL];  // 256 x one store unit (sinks.cuh, bin_line)
    __shared__ uint32_t bin_cnt[NPART_MAX];
    __shared__ uint32_t part_cnt[NPART_MAX];
    __shared__ uint32_t flush_scratch[WAVES * 16];
    constexpr bool HAS_APP = (KEYSETS & FA_KEYS_ADDR_PORT_PROTO) != 0;
    __shared__ uint32_t wpart_cnt[HAS_APP ? (1u << WIDE_PLOG2_MAX) + 2u : 1u];  // tuples per region of the wide table (scatter sink, wagg.cuh) + their bucket range (min, ~max)
    __shared__ LdsTable<LDS_SLOTS> lt;
    __shared__ LdsMinutes lm;
    // the field order a wave has learnt (lane_work tier 4): [0] steps, [1] times learnt, then the steps.  The flows_5m variant has the
    // LDS (two workgroups per CU leave it 1.7 KiB) and the registers for it; config 5's pair has the LDS but pays for the tier with
    // spills in its tile loop (scratch 28 -> 84 bytes per lane), the sketch variants have neither.
    constexpr bool HAS_SEQ = SEQ && KEYSETS == FA_KEYS_AS_PAIR;
    static_assert(!SEQ || KEYSETS == FA_KEYS_AS_PAIR, "the learnt-order variant exists for the flows_5m rollup alone");
    __shared__ uint32_t seq_all[HAS_SEQ ? WAVES * (SEQ_MAX + 2) : 1];

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (HAS_SEQ && lane < 2) seq_all[wave * (SEQ_MAX + 2) + lane] = 0;
    if (KEYSETS & FA_KEYS_AS_PAIR) {
        lds_table_clear(lt);
        for (int i = tid; i < NPART_MAX; i += WBLOCK) {
            part_cnt[i] = 0;
            bin_cnt[i] = 0;
        }
    }
    if (KEYSETS & FA_KEYS_MINUTE_SERIES) lds_minutes_clear(lm);
    if (HAS_APP)
        for (int i = tid; i < (1 << WIDE_PLOG2_MAX) + 2; i += WBLOCK) wpart_cnt[i] = i < (1 << WIDE_PLOG2_MAX) ? 0u : 0xffffffffu;
    // (the per-contract variants are launched with the sketch scatter sink only: its segments' presence is a constant there, the
    // atomic fallback of odd sketch geometries compiles out of their tile loop)
    CmsLds* const cl = (HAS_CMS && (CANDM >= 0 || a.cseg)) ? cms_lds.get() : nullptr;
    HotAddrs* const hot = (HAS_CMS && (KEYSETS != KS_ALL || (a.key_sets & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS)))) ? hot_lds.get() : nullptr;
    if (HAS_CMS && cl)
        for (int i = tid; i < (int)(CMS_SETS * CMS_NPART); i += WBLOCK) {
            cl->bin_cnt[i] = 0;
            cl->part_cnt[i] = 0;
        }
    if (HAS_CMS && hot)  // the entries this workgroup's cache held at the end of the previous launch (sinks.cuh, HotAddrs)
        for (int i = tid; i < (int)(CMS_SETS * HOT_SLOTS); i += WBLOCK) {
            const int set = i / HOT_SLOTS, sl = i % HOT_SLOTS;
            const size_t at = (size_t)blockIdx.x * (CMS_SETS * HOT_SLOTS) + i;
            const uint32_t tg = a.hot_seed_tag ? a.hot_seed_tag[at] : 0u;
            if (tg >= 2u) {
                const HotSeed sd = a.hot_seed[at];
                hot->lo[set][sl] = sd.lo;
                hot->hi[set][sl] = sd.hi;
                hot->w[set][sl] = 0;
            }
            hot->touched[set][sl] = 0;
            *hot->tag(set, sl) = tg >= 2u ? tg : 0u;
        }
    uint32_t* tile = tiles + wave * (WT_STRIDE / 4);

    LaneTally tally;
    uint32_t lt_seen = 0, lt_hits = 0, pmode = 0;
    uint32_t tb_base = 0;  // (set in the prologue below, wave-uniform)
    const uint32_t ntiles = (a.n + a.tile_recs - 1) / a.tile_recs;
    // Dynamic tile assignment inside the workgroup: workgroup b owns the tiles k * gridDim.x + b and its waves draw k
    // from an LDS counter two rounds ahead (a tile's offsets are prefetched a round before its DMA), so a wave that runs
    // slower (crowded SIMD, younger wave slot) simply takes fewer tiles.  Neighbouring tiles share a 128-byte line:
    // XCD-aware numbering keeps them on one XCD (workgroups go to the XCDs round-robin), so the line is fetched into one L2.
    __shared__ uint32_t next_k;
    if (tid == 0) next_k = 2u * WAVES;
    const uint32_t wg_pos = (gridDim.x & 7u) == 0u ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    auto tile_after_next = [&]() {
        uint32_t k = 0;
        if (lane == 0) k = lds_add_rtn_u32(&next_k, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)k) * gridDim.x + wg_pos;
    };
    // A tile = tile_recs consecutive records, one per lane.  Every lane loads its own record's bounds with ONE 8-byte
    // load (off[r], off[r+1]); the tile's byte range is lane 0's start and the last lane's end, read out of those
    // registers when the tile's turn comes (no separate descriptor loads, no lane shuffle).  The loaded values must NOT
    // be looked at before that: any arithmetic on them right after the load makes the compiler wait for it - and, vmcnt
    // being in-order, for the DMA issued just before.
    struct __attribute__((packed, aligned(4))) OffPair {
        uint32_t lo, hi;
    };
    struct Tile {
        uint32_t r0, nrec;  // records [r0, r0 + nrec)   (wave-uniform)
        uint32_t q0, q1;    // this lane's record: wire bytes [q0, q1)
    };
    auto tile_load = [&](uint32_t t) {
        Tile d{0, 0, 0, 0};
        if (t < ntiles) {
            d.r0 = t * a.tile_recs;
            d.nrec = min(a.tile_recs, a.n - d.r0);
            if (lane < d.nrec) {
                const OffPair v = *reinterpret_cast<const OffPair*>(a.off + d.r0 + lane);
                d.q0 = v.lo;
                d.q1 = v.hi;
            }
        }
        return d;
    };
    constexpr uint32_t CAP = (uint32_t)WT_STRIDE - 16u;  // bytes of a tile buffer the DMA may fill (the part's start is rounded down to 16)
    // stage wire bytes [lo & ~15, min(hi, (lo & ~15) + CAP)) into the wave's buffer
    auto issue_dma = [&](uint32_t lo, uint32_t hi) {
        const uint32_t cbase = lo & ~15u;
        const uint32_t nbytes = hi > cbase ? min(hi - cbase, CAP) : 0u;
        for (uint32_t o = lane * 16u; o < nbytes; o += 1024u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.buf + cbase + o),
                                             (__attribute__((address_space(3))) void*)(tile + (o - lane * 16u) / 4u), 16, 0, 2);
    };
    // the current tile (part): live = lanes whose record is still to be parsed
    Tile cur = tile_load(wave * gridDim.x + wg_pos);
    uint32_t cur_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.q0);
    uint32_t cur_hi = cur.nrec ? min((uint32_t)__builtin_amdgcn_readlane((int)cur.q1, (int)(cur.nrec - 1u)), a.len) : 0u;  // (bounds come from the caller: nothing beyond the buffer is ever read)
    bool live = lane < cur.nrec;
    __syncthreads();  // LDS state cleared
    // pipeline: at the top of a round the wave's tile is already on its way (issued right after the previous tile was
    // consumed) and the offsets of the tile after it are in flight
    issue_dma(cur_lo, cur_hi);
    Tile nxt = tile_load((wave + WAVES) * gridDim.x + wg_pos);
    // where in time does this batch sit?  Every wave decodes the same 64 samples while its first tile is in flight
    // (no probe dispatch, no cross-workgroup hand-over: the answer is identical all over the grid)
    tb_base = probe_tb_base(a);
    if (blockIdx.x == 0 && tid == 0) a.ctr->tb_base = tb_base;  // (for agg_kernel)
    uint32_t t2 = 0;       // the tile after the next one (wave-uniform), claimed when a tile's first part starts
    bool t2_have = false;
    while (cur.nrec != 0) {
        dma_wait_all();  // the wave's own DMA has landed: no workgroup barrier on this path
        // The claim of the tile after the next one is an LDS atomic, and an LDS instruction behind a path that may have an LDS
        // DMA in flight makes the compiler wait for EVERYTHING in flight (vmcnt is one in-order counter): at the end of the
        // round, where it used to stand, that was a wait for the write acknowledgements of the tile's tuple stores - every
        // tile, in front of the next tile's DMA.  Here nothing is in flight.
        if (!t2_have) {
            t2 = tile_after_next();
            t2_have = true;
        }
        const uint32_t cbase = cur_lo & ~15u;
        // sane bounds (the offsets come from the caller): inside the tile's byte range, end >= start
        const bool valid = live && cur.q1 >= cur.q0 && cur.q0 >= cur_lo && cur.q1 <= cur_hi;
        const bool staged = valid && cur.q1 - cbase <= CAP;  // the whole record sits in the buffer
        // Tiles are sized by records (64 of them when the mean record allows) with about two sigma of byte headroom, so
        // now and then a tile's last records do not fit the buffer: they stay live and the wave takes the tile's rest as
        // one more part (re-staged from the first record left) before it moves on.  A record that does not fit an EMPTY
        // buffer, and records with broken bounds, go to the generic path (deferred_kernel).
        bool rest = valid && !staged;
        const bool hopeless = (live && !valid) || (rest && cur.q0 == cur_lo);
        if (FA_ANY(hopeless)) {
            if (hopeless) {
                unsigned int j = atomicAdd(&a.ctr->exotic_count[a.par], 1u);
                a.exotic_idx[j] = cur.r0 + lane;
            }
            rest = rest && !hopeless;
        }
        uint32_t fill = 0xffffffffu;  // the bin this lane fills in this round
        lane_work<MODE_INGEST, KEYSETS, COLS, T8, CANDM>(a, lt, lm, part_cnt, tile, staged, cur.q0 - cbase, cur.q1 - cbase, cur.r0 + lane, tb_base, tally,
                                                  pmode, lt_seen, lt_hits, bins, bin_cnt, fill, NoHook(), cl, cms_scratch_all + (HAS_CMS ? wave * 16 : 0), hot,
                                                  (HAS_APP && a.wseg) ? wpart_cnt : nullptr, HAS_SEQ ? seq_all + wave * (SEQ_MAX + 2) : nullptr);
        // (sketch variants, round 3: starting the next tile's DMA right behind the parse - the sink is long there and does
        // not look at the tile's bytes - measured +1.8 %, like the following for the lean variants)
        // full bins leave BEFORE the next DMA is issued: behind it their stores would sit in the in-order vmcnt
        // queue and the wave would wait for the write acknowledgements on top of its tile (measured: +15 %)
        if ((KEYSETS & FA_KEYS_AS_PAIR) && a.seg) bins_flush<T8, BL>(a, bins, bin_cnt, part_cnt, flush_scratch + wave * 16, fill, tb_base, tally.direct);
        const unsigned long long restm = __builtin_amdgcn_ballot_w64(rest);
        if (restm != 0ull) {  // (rare) the rest of this tile: same records, same lanes, staged from the first one left
            live = rest;
            cur_lo = (uint32_t)__builtin_amdgcn_readlane((int)cur.q0, (int)__builtin_ctzll(restm));
            issue_dma(cur_lo, cur_hi);
            continue;
        }
        t2_have = false;
        cur = nxt;
        cur_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.q0);
        cur_hi = cur.nrec ? min((uint32_t)__builtin_amdgcn_readlane((int)cur.q1, (int)(cur.nrec - 1u)), a.len) : 0u;
        live = lane < cur.nrec;
        issue_dma(cur_lo, cur_hi);  // next tile (the buffer is free: every read of the old tile has returned)
        nxt = tile_load(t2);
    }
    // what is left in the bins (fewer than a line each) goes to the back part of the segments
    if ((KEYSETS & FA_KEYS_AS_PAIR) && a.seg) {
        __syncthreads();
        for (uint32_t idx = tid; idx < (uint32_t)NPART_MAX * TB; idx += WBLOCK) {
            const uint32_t p = idx / TB, sl = idx % TB;
            const uint32_t cnt = min(bin_cnt[p] & 0xffffu, TB);
            if (sl < cnt) {
                const uint32_t ob = (part_cnt[p] >> 16) + sl;
                uint4 tv = make_uint4(0, 0, 0, 0);
                uint2 tc = make_uint2(0, 0);
                if (T8) tc = reinterpret_cast<const uint2*>(bins)[idx];
                else tv = bins[idx];
                if (ob < a.capb) {
                    const size_t at = (size_t)p * a.region + (size_t)blockIdx.x * a.capq + (a.capq - 1u - ob);
                    if (!(FA_DBG(a, DBG_NO_TUPLE_STORE))) {
                        if (T8) reinterpret_cast<uint2*>(a.seg)[at] = tc;
                        else a.seg[at] = tv;
                    }
                } else {  // back part full (skewed batch): straight to the device-wide table
                    TupleVals v;
                    if (T8) t8_unpack(tc, p, tb_base, v);
                    else tup16_unpack(tv, v);
                    uint64_t k0, k1;
                    pack_key(tb_base + v.tbr, v.src_as, v.dst_as, v.etype, k0, k1);
                    agg_global(cold_args(), k0, k1, key_hash(k0, k1), v.bytes, v.packets, 1);
                    tally.direct++;
                }
            }
        }
        __syncthreads();
        if (tid < NPART_MAX) {
            part_cnt[tid] += min(bin_cnt[tid] & 0xffffu, TB) << 16;
            bin_cnt[tid] = 0;
        }
        __syncthreads();
    }
    if (HAS_CMS && hot) {  // the hot addresses of this workgroup: one sketch update per row and one distinct-set insert each
        __syncthreads();
        for (int i = tid; i < (int)(CMS_SETS * HOT_SLOTS); i += WBLOCK) {
            const int set = i / HOT_SLOTS, sl = i % HOT_SLOTS;
            const uint32_t tg = *hot->tag(set, sl);
            const bool hit = tg >= 2u && hot->touched[set][sl] != 0;
            const unsigned long long lo = hot->lo[set][sl], hi = hot->hi[set][sl];
            if (hit) {
                const uint32_t key[4] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
                cms_add(set ? a.cms_dst : a.cms_src, a.cms_depth, a.cms_wl2, a.cms_seed, key, hot->w[set][sl], a.cms_nrep);
                if (!(FA_DBG(a, DBG_NO_KEYSET))) keyset_offer(a, (uint32_t)set, key);
            }
            if (a.hot_seed_tag) {  // the next launch's entries: the ones that were hit, minus the lightest of a full set (every other launch)
                const size_t at = (size_t)blockIdx.x * (CMS_SETS * HOT_SLOTS) + i;
                // (the four ways of a set are four neighbouring lanes: HOT_SLOTS and WBLOCK are multiples of 4)
                const unsigned long long wv = hit ? hot->w[set][sl] : ~0ull;
                unsigned long long wmin = wv;
                uint32_t nhit = hit ? 1u : 0u, amin = (uint32_t)sl;
#pragma unroll
                for (int o = 1; o <= 2; o <<= 1) {
                    const unsigned long long ow = (unsigned long long)__shfl_xor((long long)wmin, o);
                    const uint32_t oa = (uint32_t)__shfl_xor((int)amin, o);
                    nhit += (uint32_t)__shfl_xor((int)nhit, o);
                    if (ow < wmin || (ow == wmin && oa < amin)) {
                        wmin = ow;
                        amin = oa;
                    }
                }
                const bool keep = hit && !(nhit == (uint32_t)HOT_WAYS && amin == (uint32_t)sl && (((uint32_t)sl / HOT_WAYS + a.hot_epoch) & 1u) == 0u);
                a.hot_seed_tag[at] = keep ? tg : 0u;
                if (keep) a.hot_seed[at] = HotSeed{lo, hi};
            }
        }
    }
    if (HAS_CMS && cl) {  // what is left in the sketch bins (fewer than a chunk each) goes to the back part of the segments
        __syncthreads();
        for (uint32_t idx = tid; idx < CMS_SETS * CMS_NPART * CMS_BIN; idx += WBLOCK) {
            const uint32_t p = idx / CMS_BIN, sl = idx % CMS_BIN;
            const uint32_t cnt = min(cl->bin_cnt[p] & 0xffffu, CMS_BIN);
            if (sl < cnt) {
                const uint32_t ob = (cl->part_cnt[p] >> 16) + sl;
                const uint4 t = cl->bins[idx];
                if (ob < a.ccapb) a.cseg[(size_t)p * a.cregion + (size_t)blockIdx.x * a.ccapq + (a.ccapq - 1u - ob)] = t;
                else cms_atomic_tuple(a, p, t);
            }
        }
        __syncthreads();
        for (uint32_t p = tid; p < CMS_SETS * CMS_NPART; p += WBLOCK) {
            const uint32_t w = cl->part_cnt[p] + (min(cl->bin_cnt[p] & 0xffffu, CMS_BIN) << 16);
            a.cseg_counts[(size_t)p * a.nwg + blockIdx.x] = min((w & 0xffffu) * CMS_BIN, a.ccapf);
            a.cseg_counts[((size_t)CMS_SETS * CMS_NPART + p) * a.nwg + blockIdx.x] = min(w >> 16, a.ccapb);
        }
    }
    if (KEYSETS & FA_KEYS_MINUTE_SERIES) {
        __syncthreads();
        if (tid < LDS_MINUTES && lm.key[tid] != 0 && lm.c[tid] != 0) {
            WKey k;
            wkey_pack(WK_MINUTE, 0, 0, 0, lm.key[tid] - 1u, 0, k);
            wagg_global(wargs(a), k, lm.w[tid], 0, lm.c[tid]);
        }
    }
    if (HAS_APP && a.wseg) {  // how many wide tuples this workgroup left in each region's segment
        __syncthreads();
        for (int i = tid; i < (1 << a.wplog2); i += WBLOCK) a.wseg_counts[(size_t)i * a.nwg + blockIdx.x] = min(wpart_cnt[i], a.wcapq);
        if (tid == 0 && wpart_cnt[1u << WIDE_PLOG2_MAX] != 0xffffffffu) {  // (sinks.cuh, wide_sink_wave)
            atomicMin(&a.ctr->wtb_min, wpart_cnt[1u << WIDE_PLOG2_MAX]);
            atomicMin(&a.ctr->wtb_nmax, wpart_cnt[(1u << WIDE_PLOG2_MAX) + 1u]);
        }
    }
    if (KEYSETS & FA_KEYS_AS_PAIR) {
        __syncthreads();
        for (int i0 = 0; i0 < LDS_SLOTS; i0 += WBLOCK) {  // hot-key table -> device-wide table (full waves: quad rounds)
            const int i = i0 + tid;
            Slot* sp = nullptr;
            unsigned long long b = 0, p = 0, c = 0;
            if (i < LDS_SLOTS) {
                const unsigned long long k0 = lt.k0[i], k1 = lt.k1[i];
                b = lt.bytes[i];
                p = lt.packets[i];
                c = lt.count[i];
                if (k0 != 0 && k1 != 0 && c != 0) {
                    sp = table_find_or_claim(a, k0, k1, key_hash(k0, k1));
                    if (!sp) spill_park(a, k0, k1, b, p, c);
                }
            }
            quad_atomic_update(sp, b, p, c);
        }
        if (a.seg)
            for (int i = tid; i < (1 << a.plog2); i += WBLOCK) {
                const uint32_t w = part_cnt[i];
                a.seg_counts[(size_t)i * a.nwg + blockIdx.x] = min((w & 0xffffu) * TB, a.capf);
                a.seg_counts[((size_t)NPART_MAX + i) * a.nwg + blockIdx.x] = min(w >> 16, a.capb);
            }
    }
    block_counters_add(bin_cnt, a.ctr, tally);  // (the bins are empty by now: reused as scratch)
}

// Records the tile kernel could not stage (broken offsets, tiles larger than the LDS buffer): complete
// semantics, one record per lane straight from HBM.
template <int MODE, uint32_t KEYSETS>
__device__ __forceinline__ void exotic_pass(const KArgs& a) {
    const uint32_t cnt = a.ctr->exotic_count[a.par];
    for (uint32_t j = blockIdx.x * BLOCK + threadIdx.x; j < cnt; j += gridDim.x * BLOCK) {
        uint32_t idx = a.exotic_idx[j];
        const uint32_t x0 = a.off[idx], x1 = a.off[idx + 1];
        const uint8_t* p = a.buf + min(x0, a.len);
        const uint8_t* end = a.buf + min(x1, a.len);
        bool ok = x1 >= x0 && x1 <= a.len;  // (broken bounds: a bad record, never dereferenced)
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
        const uint32_t tb = (uint32_t)r.time_received / a.gran;
        if (tb < a.late_below) atomicAdd(&a.ctr->late, 1ull);
        if (ks_on<KEYSETS>(a, FA_KEYS_AS_PAIR)) {
            uint64_t k0, k1;
            pack_key(tb, r.src_as, r.dst_as, r.etype, k0, k1);
            agg_global(a, k0, k1, key_hash(k0, k1), r.bytes, r.packets, 1);
        }
        uint64_t w = r.bytes * r.sampling_rate;
        if (ks_on<KEYSETS>(a, FA_KEYS_SRCADDR_CMS)) {
            cms_add(a.cms_src, a.cms_depth, a.cms_wl2, a.cms_seed, r.src, w, a.cms_nrep);
            keyset_offer(a, 0u, r.src);
        }
        if (ks_on<KEYSETS>(a, FA_KEYS_DSTADDR_CMS)) {
            cms_add(a.cms_dst, a.cms_depth, a.cms_wl2, a.cms_seed, r.dst, w, a.cms_nrep);
            keyset_offer(a, 1u, r.dst);
        }
        if (KEYSETS & FA_KEYS_WIDE) wide_sink_slow<KEYSETS>(a, r, tb);
    }
}

// ---- second chance: records parse_canon deferred ---------------------------------------------
// One record per lane straight from HBM/L2 with the order-free fast parser; what it is not sure
// about is decided in place by the complete parser.  Updates go to the device-wide table after a wave-level combine
// (this tier is about staying exact and tolerable on producers that do not emit canonical order).
template <int MODE, uint32_t KEYSETS>
__global__ __launch_bounds__(BLOCK) void deferred_kernel(KArgs a) {
    constexpr uint32_t COLS = MODE == MODE_DECODE ? (uint32_t)COL_ALL : cols_for_keysets<KEYSETS>();
    exotic_pass<MODE, KEYSETS>(a);
    const uint32_t cnt = a.ctr->retry_count[a.par];
    const uint32_t rounds = (cnt + gridDim.x * BLOCK - 1) / (gridDim.x * BLOCK);
    uint32_t n_ok = 0;
    for (uint32_t it = 0; it < rounds; it++) {  // whole waves stay together (wave_combine below)
        const uint32_t j = (it * gridDim.x + blockIdx.x) * BLOCK + threadIdx.x;
        bool sure = false;
        Rec r;
        rec_clear(r);
        uint32_t idx = 0;
        if (j < cnt) {
            idx = a.retry_idx[j];
            uint32_t pos = a.off[idx], end = a.off[idx + 1];
            GlobalSrc src{reinterpret_cast<const uint32_t*>(a.buf)};
            sure = end >= pos && end <= a.len;
            if (sure && a.framed) {
                uint32_t pl = 0;
                sure = frame_fast(window64(src, pos), end - pos, pl);
                pos += pl;
            }
            if (sure) sure = parse_fast<COLS>(src, pos, end, r);
            if (!sure) {  // third tier, in place: the complete parser decides
                const uint32_t x0 = a.off[idx], x1 = a.off[idx + 1];
                const uint8_t* p = a.buf + min(x0, a.len);
                const uint8_t* pe = a.buf + min(x1, a.len);
                bool ok = x1 >= x0 && x1 <= a.len;
                if (ok && a.framed) ok = frame_generic(p, pe);
                if (ok) ok = parse_generic(p, pe, r);
                if (!ok) rec_clear(r);
                atomicAdd(&a.ctr->slow, 1ull);
                if (MODE == MODE_DECODE) store_columns(a.cols, idx, r, ok ? 0 : 1);
                else if (!ok) atomicAdd(&a.ctr->bad, 1ull);
                sure = ok && MODE != MODE_DECODE;
            }
        }
        if (MODE == MODE_DECODE) {
            if (sure) store_columns(a.cols, idx, r, 0);
            continue;
        }
        n_ok += sure ? 1 : 0;
        const uint32_t tb = time_bucket(a, (uint32_t)r.time_received);
        if (sure && tb < a.late_below) atomicAdd(&a.ctr->late, 1ull);  // (rare path, rare event)
        if (ks_on<KEYSETS>(a, FA_KEYS_AS_PAIR)) {
            uint64_t k0, k1;
            pack_key(tb, r.src_as, r.dst_as, r.etype, k0, k1);
            uint64_t b = r.bytes, p = r.packets, c = 1;
            bool valid = sure;
            wave_combine<16, 2>(valid, k0, k1, b, p, c);
            if (valid) agg_global(a, k0, k1, key_hash(k0, k1), b, p, c);
        }
        if (sure && (KEYSETS & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS))) {
            const uint64_t w = r.bytes * r.sampling_rate;
            if (ks_on<KEYSETS>(a, FA_KEYS_SRCADDR_CMS)) {
                cms_add(a.cms_src, a.cms_depth, a.cms_wl2, a.cms_seed, r.src, w, a.cms_nrep);
                keyset_offer(a, 0u, r.src);
            }
            if (ks_on<KEYSETS>(a, FA_KEYS_DSTADDR_CMS)) {
                cms_add(a.cms_dst, a.cms_depth, a.cms_wl2, a.cms_seed, r.dst, w, a.cms_nrep);
                keyset_offer(a, 1u, r.dst);
            }
        }
        if (sure && (KEYSETS & FA_KEYS_WIDE)) wide_sink_slow<KEYSETS>(a, r, tb);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the next batch's deferral lists start empty (see Counters)
        a.ctr->exotic_count[a.par ^ 1u] = 0;
        a.ctr->retry_count[a.par ^ 1u] = 0;
    }
    if (MODE == MODE_INGEST) {
        uint64_t tot = wave_sum_u64(n_ok);
        if (__lane_id() == 0 && tot) {
            atomicAdd(&a.ctr->ok, (unsigned long long)tot);
            atomicAdd(&a.ctr->retried, (unsigned long long)tot);
        }
    }
}

}  // namespace fa
