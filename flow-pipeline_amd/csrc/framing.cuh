// framing.cuh - device-side framing: a chain of varint(len)-framed records (mocker/mocker.go:98-101, -proto.fixedlen) is cut
// into records ON THE GPU when the caller hands over bytes without offsets (fa_ingest_device / fa_ingest with offsets == NULL).
//
// A frame walk is serial - record i + 1 starts where record i ends - so the stream is cut into blocks of FS_BLOCK bytes and
// the one thing a block has to know, the offset of the first frame that STARTS in it, is first guessed and then proven:
//   fs_guess_kernel    a wave per block: its 64 lanes walk from the block's first 64 byte positions to the block's end.  Walks
//                      from wrong positions fall into step with the true chain sooner or later (a misread length lands on a
//                      true frame start with probability ~1 / mean record length per jump, and stays on the chain from there),
//                      so the landing position most lanes agree on is the guess for the NEXT block's first frame.
//   fs_walk_kernel     a lane per block walks from start[b] to the block's end: next[b + 1] = where it leaves, cnt[b] = frames it
//                      passed.  Run until next == start everywhere: at that fixed point start[0] = 0 and every block starts
//                      where its predecessor's walk ended - the true chain, by induction, whatever the guesses were (they only
//                      decide the number of rounds: one for ordinary streams).
//   fs_emit_kernel     exclusive scan of cnt, then a lane per block writes its frames' offsets.
// Exact or refused: a stream that is not a chain of frames ending at `len` is FA_ERR_FRAMING, like the host split; a stream
// whose guesses do not settle in FS_MAX_ROUNDS rounds (records longer than several blocks, adversarial bytes) is split on the
// host instead.  Three passes over the bytes (every frame's length sits in another cache line): the offsets-free path runs at
// about a third of the offsets path's rate - and three orders of magnitude above the host walk it replaces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

constexpr uint32_t FS_BLOCK = 16384;
constexpr uint32_t FS_ERR = 0xFFFFFFFFu;
constexpr int FS_MAX_ROUNDS = 8;

// the frame that starts at byte p of buf[0, len): the offset of the next frame, or FS_ERR (prefix longer than 10 bytes, prefix
// or payload beyond len) - the rules of the host split (flowagg.hip, frame_split_host)
__device__ __forceinline__ uint32_t fs_next(const uint8_t* buf, uint32_t p, uint32_t len) {
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
    return q + (uint32_t)v;
}

// guess[b + 1] for every block b >= 0 (guess[0] is 0 by definition and written by the host)
__global__ __launch_bounds__(256) void fs_guess_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, uint32_t* guess) {
    const uint32_t lane = __lane_id();
    const uint32_t b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b + 1 >= nblocks) return;  // (wave-uniform; the last block has no successor)
    const uint32_t begin = b * FS_BLOCK, end = begin + FS_BLOCK;
    uint32_t p = begin + lane;
    if (b == 0) p = 0;  // (the first block's chain is known: every lane walks the true one)
    while (p != FS_ERR && p < end) p = fs_next(buf, p, len);
    // the landing most lanes agree on (ties: the smallest; FS_ERR never wins against a real landing)
    uint32_t votes = 0;
    for (int i = 0; i < 64; i++) votes += p == (uint32_t)__builtin_amdgcn_readlane((int)p, i) ? 1u : 0u;
    unsigned long long best = p == FS_ERR ? 0ull : ((unsigned long long)votes << 32) | (0xFFFFFFFFu - p);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = (unsigned long long)__shfl_xor((long long)best, o);
        best = other > best ? other : best;
    }
    if (lane == 0) guess[b + 1] = best ? 0xFFFFFFFFu - (uint32_t)best : end;
}

// one round: block b walked from start[b]; next[b + 1] = where the walk leaves the block, cnt[b] = frames, flags: *changed +=
// blocks whose successor's start moved, err[b] = the walk met a malformed frame
__global__ __launch_bounds__(256) void fs_walk_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, const uint32_t* start, uint32_t* next, uint32_t* cnt,
                                                      uint8_t* err, unsigned int* changed) {
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
    if (b + 1 < nblocks) {
        const uint32_t out = bad ? end : p;
        next[b + 1] = out;
        if (out != start[b + 1]) atomicAdd(changed, 1u);
    } else if (!bad && p != len) {
        err[b] = 1;  // (cannot happen: fs_next never passes len)
    }
    if (b == 0) next[0] = 0;
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
