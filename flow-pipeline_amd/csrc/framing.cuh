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
//                      generator's four producers.)
//   fs_walk_kernel     a lane per block walks from start[b] to the block's end: exit[b], cnt[b].  A round compares exit[b] with
//                      start[b + 1] everywhere; where they differ and block b's own start agreed with ITS predecessor's exit in
//                      the round before, start[b + 1] is replaced (a block walked from a start that has just been refuted must
//                      not overrule its successor's guess: the error would travel down the stream one block per round).
//                      Run until nothing differs: at that fixed point start[0] = 0 and every block starts where its
//                      predecessor's walk ended - the true chain, by induction, whatever the guesses were.  They only decide
//                      the number of rounds: three for ordinary streams (verify, repair the isolated wrong guesses, confirm).
//   fs_emit_kernel     exclusive scan of cnt, then a lane per block writes its frames' offsets.
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
constexpr int FS_PLAUSIBLE = 2;     // consecutive plausible frames a candidate needs

// the frame that starts at byte p of buf[0, len): the offset of the next frame, or FS_ERR (prefix longer than 10 bytes, prefix
// or payload beyond len) - the rules of the host split (flowagg.hip, frame_split_host).  *payload: where its payload begins.
__device__ __forceinline__ uint32_t fs_next(const uint8_t* buf, uint32_t p, uint32_t len, uint32_t* payload = nullptr) {
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
__device__ __forceinline__ bool fs_plausible_payload(const uint8_t* buf, uint32_t q, uint32_t e) {
    if (q >= e) return false;
    uint32_t last = 0;
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

// start[b] for every block b >= 1 (start[0] is 0 by definition and written by the host): the smallest plausible candidate,
// the block's begin when none stands (the rounds below - or the host - sort that out)
__global__ __launch_bounds__(256) void fs_guess_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, uint32_t* start) {
    const uint32_t lane = __lane_id();
    const uint32_t b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b == 0 || b >= nblocks) return;  // (wave-uniform)
    const uint32_t begin = b * FS_BLOCK;
    uint32_t guess = begin;
    for (uint32_t r = 0; r < FS_CAND; r += 64) {
        uint32_t p = begin + r + lane;
        bool ok = p < len;
        for (int k = 0; k < FS_PLAUSIBLE && ok && p < len; k++) {
            uint32_t payload = 0;
            const uint32_t q = fs_next(buf, p, len, &payload);
            ok = q != FS_ERR && fs_plausible_payload(buf, payload, q);
            p = q;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        if (m != 0ull) {
            guess = begin + r + (uint32_t)__builtin_ctzll(m);
            break;
        }
    }
    if (lane == 0) start[b] = guess;
}

// one round: block b walked from start[b]: cnt[b] = frames, err[b] = malformed frame met, exit = where the walk leaves the block.
// *differ += blocks whose exit is not their successor's start; the successor's start is replaced when this block is trusted
// (trust_in[b]: its own start agreed with its predecessor's exit in the round before; nullptr in the first round: nobody acts).
// trust_out[b + 1]: block b + 1's start agrees with this block's exit now.
__global__ __launch_bounds__(256) void fs_walk_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, uint32_t* start, uint32_t* cnt, uint8_t* err,
                                                      const uint8_t* trust_in, uint8_t* trust_out, uint32_t* exit_out, unsigned int* differ) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t end = b + 1 == nblocks ? len : (b + 1) * FS_BLOCK;
    uint32_t p = start[b], c = 0;
    bool bad = false;
    while (p < end) {
        const uint32_t q = fs_next(buf, p, len);
        if (q == FS_ERR) {
            bad = true;
            break;
        }
        p = q;
        c++;
    }
    cnt[b] = c;
    err[b] = bad ? 1 : 0;
    exit_out[b] = bad ? end : p;
    if (b == 0) trust_out[0] = 1;
    if (b + 1 < nblocks) {
        const uint32_t out = bad ? end : p;
        const bool same = out == start[b + 1];
        if (!same) atomicAdd(differ, 1u);
        trust_out[b + 1] = (same || (trust_in && trust_in[b])) ? 1 : 0;
    }
}
// behind a round: the trusted blocks' exits become their successors' starts (a separate pass: a block reads its own start
// while its predecessor would be replacing it)
__global__ __launch_bounds__(256) void fs_apply_kernel(uint32_t nblocks, uint32_t* start, const uint32_t* exit_in, const uint8_t* trust_in) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b + 1 >= nblocks) return;
    if (trust_in[b] && exit_in[b] != start[b + 1]) start[b + 1] = exit_in[b];
}

// off[base[b] + i] = start of the block's i-th frame; the last block also writes off[n] = len
__global__ __launch_bounds__(256) void fs_emit_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, const uint32_t* start, const uint32_t* base, uint32_t* off,
                                                      uint32_t n) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t end = b + 1 == nblocks ? len : (b + 1) * FS_BLOCK;
    uint32_t p = start[b], i = base[b];
    while (p < end) {
        off[i++] = p;
        p = fs_next(buf, p, len);  // (the chain was proven: never FS_ERR here)
    }
    if (b + 1 == nblocks) off[n] = len;
}
// any malformed frame on the proven chain?
__global__ void fs_err_kernel(const uint8_t* err, uint32_t nblocks, unsigned int* bad) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += gridDim.x * blockDim.x)
        if (err[b]) atomicAdd(bad, 1u);
}

}  // namespace fa
