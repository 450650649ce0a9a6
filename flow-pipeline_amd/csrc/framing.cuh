// framing.cuh - device-side framing: a chain of varint(len)-framed records (mocker/mocker.go:98-101, -proto.fixedlen) is cut
// into records ON THE GPU when the caller hands over bytes without offsets (fa_ingest_device / fa_ingest with offsets == NULL).
//
// A frame walk is serial - record i + 1 starts where record i ends - so the stream is cut into blocks of FS_BLOCK bytes and
// the one thing a block has to know, the offset of the first frame that STARTS in it, is first guessed and then proven:
//   fs_guess_kernel    a wave per block tries the block's first FS_CAND byte positions as frame starts: a candidate stands when
//                      FS_PLAUSIBLE consecutive frames from it are non-empty protobuf messages that end exactly where their
//                      length prefix says, with field numbers in ascending order (what proto.Marshal and every canonical
//                      encoder emits; flow.pb.go:57-147).  The smallest standing candidate is the guess.  (Round 4 first
//                      let 64 unvalidated walks vote on where they leave the block: on real records - every record the same
//                      field layout - walks from wrong positions fall into OTHER self-consistent chains, not into the true
//                      one, and 35 of 39 guesses were wrong.  With the plausibility test 96-99 % are right, measured on the
//                      generator's four producers.)  The block's first FS_STAGE bytes are staged in LDS and a candidate has
//                      to pass a test of a few bytes (two sane length prefixes, two sane first tags) before it is parsed in
//                      full: 64 lanes parsing 64 different candidates ran every loop of the parser to the longest trip count
//                      among them, every trip a dependent load - 1.2 ms per 2.4 GB in rounds 4-5, whether the bytes came from
//                      L1 or from LDS.
//   fs_walk_kernel     a lane per block walks from start[b] to the block's end: exit[b], cnt[b], and where the first frame of each
//                      of the block's 64 sub-blocks of 256 bytes starts (ent / present: what lets the emit pass use 64 lanes per
//                      block).
//   fs_round_kernel    compares exit[b] with start[b + 1] everywhere; where they differ and block b's own start agreed with ITS
//                      predecessor's exit in the round before, fs_apply_kernel replaces start[b + 1] and puts block b + 1 on the
//                      list of blocks to walk again (a block walked from a start that has just been refuted must not overrule
//                      its successor's guess: the error would travel down the stream one block per round).  Run until nothing
//                      differs: at that fixed point start[0] = 0 and every block starts where its predecessor's walk ended -
//                      the true chain, by induction, whatever the guesses were.  They only decide the number of rounds: three
//                      for ordinary streams (verify, repair the isolated wrong guesses, confirm).
//   fs_rewalk_kernel   the listed blocks only (1-4 % of an ordinary stream), and of those only the frames up to the point where the
//                      walk from the new start meets the walk from the old one (rounds 4-5 walked EVERY block again in every
//                      round: 3 x 0.55 ms per 2.4 GB).
//   fs_emit_kernel     exclusive scan of cnt, then a wave per block, the block staged in LDS: lane j counts the frames that start
//                      in sub-block j (from ent), a prefix sum over the wave places them, a second walk writes their offsets -
//                      neighbouring lanes write neighbouring pieces of the offsets array (rounds 4-5: a lane per block, 240
//                      stores 1 KB apart from its neighbour's, 1.12 ms per 2.4 GB).
// Exact or refused: a stream that is not a chain of frames ending at `len` is FA_ERR_FRAMING, like the host split; a stream
// whose guesses do not settle in FS_MAX_ROUNDS rounds (a producer that does not marshal in field order, records longer than
// FS_CAND bytes or than several blocks, adversarial bytes) is split on the host instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

constexpr uint32_t FS_BLOCK = 16384;
constexpr uint32_t FS_ERR = 0xFFFFFFFFu;
constexpr int FS_MAX_ROUNDS = 12;
constexpr uint32_t FS_CAND = 256;   // candidate positions per block (covers records up to that long)
constexpr int FS_PLAUSIBLE = 2;     // consecutive plausible frames a candidate needs ...
constexpr int FS_PREFILTER = 2;     // ... and the frames from it whose first bytes have to look like frames before any is parsed (>= FS_PLAUSIBLE;
                                    // 3, 4, 6: the same guesses at the same cost, profiles/r06_framing_micro.txt)
constexpr bool FS_TIERS = true;      // candidates whose two frames begin with the same tag byte are parsed first
constexpr uint32_t FS_SUB = 256;    // sub-block: the emit pass walks one per lane
constexpr uint32_t FS_NSUB = FS_BLOCK / FS_SUB;  // 64 = a wave
constexpr uint32_t FS_STAGE = 1024; // bytes of a block the guess stages in LDS (the candidates + two frames behind the last one)
constexpr uint32_t FS_SLACK = 16;   // bytes staged behind a block: the prefix of a frame that starts on its last byte
constexpr uint32_t FS_REWALK_WGS = 64;   // fs_rewalk_kernel's grid: its lanes share the list
static_assert(FS_NSUB == 64, "fs_emit_kernel: one lane per sub-block");

// bytes [lo, lo + n) of the stream, staged in LDS (what lies behind is read as the last staged byte: the callers keep away)
struct FsLds {
    const uint8_t* lds;
    uint32_t lo, n;
    __device__ __forceinline__ uint8_t operator[](uint32_t p) const { return lds[min(p - lo, n - 1u)]; }
};
// a wave stages bytes [lo, lo + N) of buf[0, len) into its own piece of LDS: 16 bytes per lane and trip (buf is 16-byte aligned
// and lo a multiple of 16: a piece whose first byte lies inside the stream is read whole), zeros behind the stream; then the
// lanes may read what the others wrote.  Every load of the wave is in flight before the first LDS write: a loop of load -
// wait - write trips staged a 16 KB block in 17 memory latencies (3.6 TB/s over all waves; 6.5 TB/s this way,
// profiles/r06_framing_micro.txt).
template <uint32_t N>
__device__ __forceinline__ void fs_stage(const uint8_t* buf, uint32_t len, uint32_t lo, uint4* dst, uint32_t lane) {
    constexpr uint32_t TRIPS = (N + 1023u) / 1024u;
    uint4 r[TRIPS];
#pragma unroll
    for (uint32_t t = 0; t < TRIPS; t++) {
        const uint32_t i = lane * 16u + t * 1024u;
        r[t] = (i < N && lo + i < len) ? *reinterpret_cast<const uint4*>(buf + lo + i) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t t = 0; t < TRIPS; t++) {
        const uint32_t i = lane * 16u + t * 1024u;
        if (i < N) dst[i >> 4] = r[t];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the frame that starts at byte p of buf[0, len): the offset of the next frame, or FS_ERR (prefix longer than 10 bytes, prefix
// or payload beyond len) - the rules of the host split (flowagg.hip, frame_split_host).  *payload: where its payload begins.
template <class Bytes>
__device__ __forceinline__ uint32_t fs_next(const Bytes& buf, uint32_t p, uint32_t len, uint32_t* payload = nullptr) {
    unsigned long long v = 0;
    uint32_t q = p;
    for (int i = 0;; i++) {
        if (i >= 10 || q >= len) return FS_ERR;
        const uint32_t b = buf[q++];
        if (i < 9) v |= (unsigned long long)(b & 0x7fu) << (7 * i);
        else v |= (unsigned long long)(b & 1u) << 63;
        if (!(b & 0x80u)) break;
    }
    if (v > (unsigned long long)(len - q)) return FS_ERR;
    if (payload) *payload = q;
    return q + (uint32_t)v;
}
// buf[q, e): a non-empty sequence of protobuf fields with ascending field numbers that ends exactly at e?
template <class Bytes>
__device__ __forceinline__ bool fs_plausible_payload(const Bytes& buf, uint32_t q, uint32_t e, uint32_t last = 0) {  // last: the field number before q
    if (q >= e) return false;
    while (q < e) {
        uint32_t t = 0;
        int i = 0;
        for (;; i++) {
            if (i >= 5 || q >= e) return false;
            const uint32_t b = buf[q++];
            t |= (b & 0x7fu) << (7 * i);
            if (!(b & 0x80u)) break;
        }
        const uint32_t wt = t & 7u, fn = t >> 3;
        if (fn == 0 || fn < last) return false;
        last = fn;
        if (wt == 0u) {
            for (i = 0;; i++) {
                if (i >= 10 || q >= e) return false;
                if (!(buf[q++] & 0x80u)) break;
            }
        } else if (wt == 1u) {
            q += 8;
        } else if (wt == 5u) {
            q += 4;
        } else if (wt == 2u) {
            uint32_t l = 0;
            for (i = 0;; i++) {
                if (i >= 5 || q >= e) return false;
                const uint32_t b = buf[q++];
                l |= (b & 0x7fu) << (7 * i);
                if (!(b & 0x80u)) break;
            }
            if (l > e - q) return false;
            q += l;
        } else {
            return false;
        }
    }
    return q == e;
}

// fs_plausible_payload over staged bytes, a field per trip instead of a byte: eight bytes from the field's first (three LDS
// words, v_alignbyte) hold its tag and its value's varint or length.  64 lanes parsing 64 different candidates byte by byte were
// bound by instruction issue, not by loads: 40 instructions per byte x 142 bytes, every branch any lane takes
// (profiles/r06_framing_micro.txt).  A trip handles a tag of one or two bytes (fields below 2048) with a varint of up to
// 8 - tag bytes behind it, selects instead of branches; the first field that is anything else hands the rest of the payload to
// the byte-wise parser.  The stage is followed by FS_STAGE_PAD bytes: a window may begin on the stage's last byte.
constexpr uint32_t FS_STAGE_PAD = 16;
__device__ __forceinline__ uint32_t fs_varint28(uint32_t z, uint32_t bytes) {  // the value of a varint of 1..4 bytes in z
    const uint32_t v = (z & 0x7fu) | ((z >> 1) & 0x3f80u) | ((z >> 2) & 0x1fc000u) | ((z >> 3) & 0xfe00000u);
    return v & ((1u << (7u * bytes)) - 1u);
}
__device__ __forceinline__ bool fs_plausible_staged(const FsLds& buf, uint32_t q, uint32_t e) {
    if (q >= e) return false;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(buf.lds);
    uint32_t last = 0;
    while (q < e) {
        const uint32_t r = q - buf.lo, i = r >> 2;
        const uint32_t a = w[i], b = w[i + 1u], c = w[i + 2u];  // (q < e <= lo + FS_STAGE: inside the stage and its pad)
        const uint32_t x0 = __builtin_amdgcn_alignbyte(b, a, r & 3u), x1 = __builtin_amdgcn_alignbyte(c, b, r & 3u);  // bytes 0-3, 4-7 of the field
        if ((x0 & 0x8080u) == 0x8080u) break;  // a tag of three bytes or more
        const uint32_t tl = 1u + ((x0 >> 7) & 1u);
        const uint32_t t = fs_varint28(x0, tl);
        const uint32_t wt = t & 7u, fn = t >> 3;
        const uint32_t rem = e - q;  // >= 1
        if (fn == 0 || fn < last || tl > rem) return false;
        const uint32_t z0 = __builtin_amdgcn_alignbyte(x1, x0, tl);  // bytes tl .. tl + 3
        const uint32_t z1 = x1 >> (8u * tl);                        // bytes tl + 4 .. 7, zeros behind
        const uint32_t s0 = ~z0 & 0x80808080u, s1 = ~z1 & (0x80808080u >> (8u * tl));
        if (s0 == 0u && (wt == 2u || (wt == 0u && s1 == 0u))) break;  // a length of five bytes, a varint that leaves the window
        const uint32_t vl = s0 ? ((uint32_t)__builtin_ctz(s0) >> 3) + 1u : 5u + ((uint32_t)__builtin_ctz(s1 | 0x80000000u) >> 3);
        const uint32_t l = fs_varint28(z0, min(vl, 4u));
        const uint32_t adv = wt == 0u ? vl : wt == 2u ? vl + l : wt == 1u ? 8u : wt == 5u ? 4u : 0xFFFFFFFFu;  // (3, 4, 6, 7: no such wire type)
        if (adv > rem - tl) return false;  // (wire types 1 and 5 past the end: the byte-wise parser finds q != e)
        last = fn;
        q += tl + adv;
    }
    return q == e || (q < e && fs_plausible_payload(buf, q, e, last));
}

// a few bytes decide whether position p is worth a parse: two frames in a row with a length prefix of one to three bytes, a
// payload that is not empty and ends inside [.., lim], and a first tag of field >= 1 with a wire type that exists - what
// fs_next + fs_plausible_payload would find out first, without their loops (a first tag of several bytes passes)
// Returns 0: no; 1: worth a parse; 2: worth a parse, and the first two frames begin with the same tag byte - what the records of one
// producer nearly always do, and what a position inside a record (13 of 64 positions are field boundaries: they pass for a
// prefix and a tag) nearly never does.
template <class Bytes, int FRAMES>
__device__ __forceinline__ uint32_t fs_prefilter(const Bytes& buf, uint32_t p, uint32_t lim) {
    uint32_t first_tag = 0x100u, same = 0u;
#pragma unroll
    for (int k = 0; k < FRAMES; k++) {
        if (p >= lim) return k > 0 ? 1u : 0u;  // (the first frame ended where the stream - or the stage - does)
        const uint32_t b0 = buf[p], b1 = buf[p + 1], b2 = buf[p + 2];
        uint32_t v = b0 & 0x7fu, q = p + 1;
        if (b0 & 0x80u) {
            v |= (b1 & 0x7fu) << 7;
            q++;
            if (b1 & 0x80u) {
                if (b2 & 0x80u) return 0u;
                v |= b2 << 14;
                q++;
            }
        }
        if (v == 0) return 0u;
        if (q >= lim || v > lim - q) return k >= FS_PLAUSIBLE ? 1u + same : 0u;  // (leaves the stage: the frames that are parsed must not, the others cannot tell)
        const uint32_t t = buf[q];
        if (!(t & 0x80u)) {
            const uint32_t wt = t & 7u;
            if ((t >> 3) == 0 || !(wt == 0u || wt == 1u || wt == 2u || wt == 5u)) return 0u;
        }
        if (k == 0) first_tag = t;
        if (k == 1) same = t == first_tag ? 1u : 0u;
        p = q + v;
    }
    return 1u + same;
}

// start[b] for every block b >= 1 (start[0] is 0 by definition and written by the host): the smallest plausible candidate,
// the block's begin when none stands (the rounds below - or the host - sort that out).  The guess sees the stream end at the
// end of the stage: a candidate whose two frames leave it does not stand (FS_STAGE - FS_CAND bytes hold two frames of 384).
template <int PRE, int FULL, bool TIERS>
__global__ __launch_bounds__(256) void fs_guess_kernel_t(const uint8_t* buf, uint32_t len, uint32_t nblocks, uint32_t* start) {
    __shared__ uint4 stage[4][(FS_STAGE + FS_STAGE_PAD) / 16];
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * (blockDim.x >> 6) + wave;
    if (b == 0 || b >= nblocks) return;  // (wave-uniform; no workgroup barrier below: every wave reads what it staged itself)
    const uint32_t begin = b * FS_BLOCK;
    fs_stage<FS_STAGE + FS_STAGE_PAD>(buf, len, begin, stage[wave], lane);
    const FsLds bytes{reinterpret_cast<const uint8_t*>(stage[wave]), begin, FS_STAGE};
    const uint32_t lim = min(len, begin + FS_STAGE);
    uint32_t guess = begin;
    bool found = false;
    for (uint32_t r = 0; r < FS_CAND && !found; r += 64) {
        const uint32_t p0 = begin + r + lane;
        const uint32_t pre = p0 < lim ? fs_prefilter<FsLds, PRE>(bytes, p0, lim) : 0u;
        if (__builtin_amdgcn_ballot_w64(pre != 0u) == 0ull) continue;
        // the candidates whose two frames begin alike first: usually the true start alone - the others would each run the parser's
        // loops to their own trip counts (the wave pays for the longest of every loop); then, if none of them stands, the rest
#pragma unroll 1
        for (uint32_t tier = TIERS ? 2u : 1u; tier >= 1u && !found; tier--) {
            bool ok = TIERS ? pre == tier : pre != 0u;
            if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
            uint32_t p = p0;
            for (int k = 0; k < FULL && ok && p < lim; k++) {
                uint32_t payload = 0;
                const uint32_t q = fs_next(bytes, p, lim, &payload);
                ok = q != FS_ERR && fs_plausible_staged(bytes, payload, q);
                p = q;
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
            if (m != 0ull) {
                guess = begin + r + (uint32_t)__builtin_ctzll(m);
                found = true;
            }
        }
    }
    if (lane == 0) start[b] = guess;
}
// The walk of block b from `from`: frames that start in [from, end), where it leaves the block, whether it met a malformed frame;
// ent8[j] = where the first frame of sub-block j starts (offset inside the sub-block; only where bit j of *present is set),
// stored four sub-blocks at a time.
template <class Bytes>
__device__ __forceinline__ void fs_walk_block(const Bytes& bytes, uint32_t len, uint32_t begin, uint32_t end, uint32_t from, uint32_t* cnt, uint8_t* err, uint32_t* exit_out,
                                              uint8_t* ent8, unsigned long long* present) {
    uint32_t* e32 = reinterpret_cast<uint32_t*>(ent8);
    uint32_t p = from, c = 0, sub = 0;  // sub: sub-blocks below it have been passed
    uint32_t acc = 0, w = FS_NSUB;      // the word of four entries being filled (FS_NSUB: none yet)
    unsigned long long mask = 0ull;
    bool bad = false;
    while (p < end) {
        const uint32_t j = (p - begin) / FS_SUB;  // (p >= begin: a start is its predecessor's exit, or a guess inside the block)
        if (j >= sub) {
            if ((j >> 2) != w) {
                if (w != FS_NSUB) e32[w] = acc;
                w = j >> 2;
                acc = 0;
            }
            acc |= ((p - begin) % FS_SUB) << (8u * (j & 3u));
            mask |= 1ull << j;
            sub = j + 1;
        }
        const uint32_t q = fs_next(bytes, p, len);
        if (q == FS_ERR) {
            bad = true;
            break;
        }
        p = q;
        c++;
    }
    if (w != FS_NSUB) e32[w] = acc;
    *present = mask;
    *cnt = c;
    *err = bad ? 1 : 0;
    *exit_out = bad ? end : p;
}
// the first round: every block, a lane each, from its guess
__global__ __launch_bounds__(256) void fs_walk_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, const uint32_t* start, uint32_t* cnt, uint8_t* err,
                                                      uint32_t* exit_out, uint8_t* ent8, unsigned long long* present) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t begin = b * FS_BLOCK, end = b + 1 == nblocks ? len : (b + 1) * FS_BLOCK;
    fs_walk_block(buf, len, begin, end, start[b], cnt + b, err + b, exit_out + b, ent8 + (size_t)b * FS_NSUB, present + b);
}
// the later rounds: the blocks fs_apply_kernel listed, a lane each.  A guess that was wrong is a position whose chain falls
// into the true one after a frame or two (that is how it passed for plausible), so the walk from the new start runs beside the
// walk from the old one only until the two meet: behind that point everything the first walk found stands.  (Walking the
// whole block again took 0.1-0.2 ms per round however few blocks were listed - 240 dependent loads, in memory or in LDS.)
__global__ __launch_bounds__(256) void fs_rewalk_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, const uint32_t* start, const uint32_t* list,
                                                        const unsigned int* nlist, uint32_t* cnt, uint8_t* err, uint32_t* exit_out, uint8_t* ent8,
                                                        unsigned long long* present) {
    const uint32_t n = *nlist;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t b = list[2u * i], old_from = list[2u * i + 1u];
        const uint32_t begin = b * FS_BLOCK, end = b + 1 == nblocks ? len : (b + 1) * FS_BLOCK;
        uint8_t* e = ent8 + (size_t)b * FS_NSUB;
        uint32_t pn = start[b], po = err[b] ? FS_ERR : old_from;  // (a first walk that met a malformed frame is not followed)
        uint32_t cn = 0, co = 0, sub = 0;  // frames of either walk so far; sub-blocks below `sub` have been passed by the new one
        unsigned long long mask = 0ull;    // sub-blocks in which the new walk found its first frame
        bool bad = false;
        while (pn != po && (pn < end || po < end)) {
            if (pn < po) {  // (pn < end: the smaller of two different positions of which one lies inside the block)
                const uint32_t j = (pn - begin) / FS_SUB;
                if (j >= sub) {
                    e[j] = (uint8_t)((pn - begin) % FS_SUB);
                    mask |= 1ull << j;
                    sub = j + 1;
                }
                const uint32_t q = fs_next(buf, pn, len);
                if (q == FS_ERR) {
                    bad = true;
                    break;
                }
                pn = q;
                cn++;
            } else {
                po = fs_next(buf, po, len);  // (FS_ERR cannot happen: the first walk passed here without)
                co++;
            }
        }
        if (!bad && pn == po && pn < end) {  // met inside the block: the first walk's frames from here on, its exit, its entries behind this sub-block
            const uint32_t j = (pn - begin) / FS_SUB;
            if (j >= sub) {
                e[j] = (uint8_t)((pn - begin) % FS_SUB);
                mask |= 1ull << j;
            }
            const unsigned long long below = (2ull << j) - 1ull;  // sub-blocks up to j: the new walk's entries
            present[b] = (present[b] & ~below) | mask;
            cnt[b] = cnt[b] - co + cn;
        } else {  // the new walk covered the block on its own
            present[b] = mask;
            cnt[b] = cn;
            err[b] = bad ? 1 : 0;
            exit_out[b] = bad ? end : pn;
        }
    }
}
// one round: *differ += blocks whose exit is not their successor's start; trust_out[b + 1]: block b + 1's start agrees with
// block b's exit now, or will be replaced by it (block b is trusted: trust_in[b] - its own start agreed with its predecessor's
// exit in the round before; nullptr in the first round: nobody acts)
__global__ __launch_bounds__(256) void fs_round_kernel(uint32_t nblocks, const uint32_t* start, const uint32_t* exit_in, const uint8_t* trust_in, uint8_t* trust_out,
                                                       unsigned int* differ) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    if (b == 0) trust_out[0] = 1;
    if (b + 1 < nblocks) {
        const bool same = exit_in[b] == start[b + 1];
        if (!same) atomicAdd(differ, 1u);
        trust_out[b + 1] = (same || (trust_in && trust_in[b])) ? 1 : 0;
    }
}
// behind a round: the trusted blocks' exits become their successors' starts, and those successors are walked again (a separate
// pass: a block reads its own start while its predecessor would be replacing it)
__global__ __launch_bounds__(256) void fs_apply_kernel(uint32_t nblocks, uint32_t* start, const uint32_t* exit_in, const uint8_t* trust_in, uint32_t* list,
                                                       unsigned int* nlist) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b + 1 >= nblocks) return;
    if (trust_in[b] && exit_in[b] != start[b + 1]) {
        const unsigned int at = atomicAdd(nlist, 1u);
        list[2u * at] = b + 1;
        list[2u * at + 1u] = start[b + 1];  // (where it was walked from: fs_rewalk_kernel follows both walks until they meet)
        start[b + 1] = exit_in[b];
    }
}

// off[base[b] + i] = start of the block's i-th frame; the last block also writes off[n] = len.  A wave per block, lane j = the
// frames that start in sub-block j (the chain was proven: fs_next never fails here, and the frames of a sub-block end where
// the next entry - or the block's exit - begins).
__global__ __launch_bounds__(256) void fs_emit_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, const uint8_t* ent8, const unsigned long long* present,
                                                      const uint32_t* base, uint32_t* off, uint32_t n) {
    __shared__ uint4 stage[4][(FS_BLOCK + FS_SLACK) / 16];
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * 4u + wave;
    if (b >= nblocks) return;  // (wave-uniform; no workgroup barrier below)
    const uint32_t begin = b * FS_BLOCK, end = b + 1 == nblocks ? len : (b + 1) * FS_BLOCK;
    const unsigned long long mask = present[b];
    if (mask != 0ull) {
        fs_stage<FS_BLOCK + FS_SLACK>(buf, len, begin, stage[wave], lane);
        const FsLds bytes{reinterpret_cast<const uint8_t*>(stage[wave]), begin, FS_BLOCK + FS_SLACK};
        const bool mine = (mask >> lane) & 1ull;
        const uint32_t sub_end = min(begin + (lane + 1) * FS_SUB, end);
        const uint32_t p0 = begin + lane * FS_SUB + ent8[(size_t)b * FS_NSUB + lane];
        uint32_t c = 0;
        if (mine)
            for (uint32_t p = p0; p < sub_end; p = fs_next(bytes, p, len)) c++;
        uint32_t at = c;  // inclusive prefix sum over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(at, d, 64);
            if ((int)lane >= d) at += up;
        }
        uint32_t i = base[b] + at - c;
        if (mine)
            for (uint32_t p = p0; p < sub_end; p = fs_next(bytes, p, len)) off[i++] = p;
    }
    if (b + 1 == nblocks && lane == 0) off[n] = len;
}
// any malformed frame on the proven chain?
__global__ void fs_err_kernel(const uint8_t* err, uint32_t nblocks, unsigned int* bad) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += gridDim.x * blockDim.x)
        if (err[b]) atomicAdd(bad, 1u);
}

}  // namespace fa
