// framing_bench: the kernels of csrc/framing.cuh one by one on a framed stream read from a file (tools/micro/framing_bench.py writes one
// with the library's generator), with hipEvents - and experimental variants of the pieces that did not behave as estimated
// (LDS staging, the guess's prefilter), to decide what goes into framing.cuh.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o framing_bench framing_bench.hip && ./framing_bench stream.bin
#include "../../flow-pipeline_amd/csrc/framing.cuh"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
using namespace fa;
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

// ---- for the record: staging as a loop of load - wait - write trips (what framing.cuh did first)
__device__ __forceinline__ void stage_loop(const uint8_t* buf, uint32_t len, uint32_t lo, uint32_t n, uint4* dst, uint32_t lane) {
    for (uint32_t i = lane * 16u; i < n; i += 64u * 16u) dst[i >> 4] = lo + i < len ? *reinterpret_cast<const uint4*>(buf + lo + i) : make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// staging only: what a wave-per-block kernel pays before it does anything (sum keeps the reads alive)
template <bool PIPE>
__global__ __launch_bounds__(256) void stage_only_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, uint32_t* sink) {
    __shared__ uint4 stage[4][(FS_BLOCK + FS_SLACK) / 16];
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * 4u + wave;
    if (b >= nblocks) return;
    if (PIPE) fs_stage<FS_BLOCK + FS_SLACK>(buf, len, b * FS_BLOCK, stage[wave], lane);
    else stage_loop(buf, len, b * FS_BLOCK, FS_BLOCK + FS_SLACK, stage[wave], lane);
    const uint8_t* s = reinterpret_cast<const uint8_t*>(stage[wave]);
    if (s[lane * 257u] == 0x7b && s[lane * 131u + 7u] == 0x11) sink[b] = 1;
}
// emit without staging: the lanes read their prefixes from memory, positions of the first walk kept in registers (up to 8)
__global__ __launch_bounds__(256) void emit_regs_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, const uint8_t* ent8, const unsigned long long* present,
                                                        const uint32_t* base, uint32_t* off, uint32_t n) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * 4u + wave;
    if (b >= nblocks) return;
    const uint32_t begin = b * FS_BLOCK, end = b + 1 == nblocks ? len : (b + 1) * FS_BLOCK;
    const unsigned long long mask = present[b];
    const bool mine = (mask >> lane) & 1ull;
    const uint32_t sub_end = min(begin + (lane + 1) * FS_SUB, end);
    const uint32_t p0 = begin + lane * FS_SUB + ent8[(size_t)b * FS_NSUB + lane];
    uint32_t c = 0, pos[8];
    if (mine) {
        uint32_t p = p0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            pos[k] = p;
            if (p < sub_end) {
                c++;
                p = fs_next(buf, p, len);
            }
        }
        for (; p < sub_end; p = fs_next(buf, p, len)) c++;
    }
    uint32_t at = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(at, d, 64);
        if ((int)lane >= d) at += up;
    }
    const uint32_t i0 = base[b] + at - c;
    if (mine) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((uint32_t)k < c) off[i0 + k] = pos[k];
        if (c > 8) {
            uint32_t p = pos[7], i = i0 + 7;
            for (; p < sub_end; p = fs_next(buf, p, len)) off[i++] = p;
        }
    }
    if (b + 1 == nblocks && lane == 0) off[n] = len;
}

// ---- experimental guesses.  NF = frames the prefilter looks at; FULL: survivors are parsed in full (the product's rule) or the
// smallest survivor is the guess.  surv[b] = survivors of the round that decided (statistics).
template <int NF, int FULL, bool PIPE>
__global__ __launch_bounds__(256) void guess_x_kernel(const uint8_t* buf, uint32_t len, uint32_t nblocks, uint32_t* start, uint32_t* surv) {
    __shared__ uint4 stage[4][FS_STAGE / 16];
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * (blockDim.x >> 6) + wave;
    if (b == 0 || b >= nblocks) return;
    const uint32_t begin = b * FS_BLOCK;
    fs_stage<FS_STAGE>(buf, len, begin, stage[wave], lane);
    const FsLds bytes{reinterpret_cast<const uint8_t*>(stage[wave]), begin, FS_STAGE};
    const uint32_t lim = min(len, begin + FS_STAGE);
    uint32_t guess = begin, ns = 0;
    for (uint32_t r = 0; r < FS_CAND; r += 64) {
        uint32_t p = begin + r + lane;
        bool ok = p < lim && fs_prefilter<FsLds, NF>(bytes, p, lim) != 0u;
        const unsigned long long s = __builtin_amdgcn_ballot_w64(ok);
        if (s == 0ull) continue;
        ns = (uint32_t)__builtin_popcountll(s);
        if (FULL == 1) {
            for (int k = 0; k < FS_PLAUSIBLE && ok && p < lim; k++) {
                uint32_t payload = 0;
                const uint32_t q = fs_next(bytes, p, lim, &payload);
                ok = q != FS_ERR && fs_plausible_payload(bytes, payload, q);
                p = q;
            }
        } else if (FULL == 2) {  // survivors one after the other, smallest first: the whole wave follows ONE parse (uniform control flow)
            unsigned long long rest = s;
            ok = false;
            while (rest) {
                const uint32_t l = (uint32_t)__builtin_ctzll(rest);
                rest &= rest - 1ull;
                uint32_t pp = begin + r + l;
                bool good = true;
                for (int k = 0; k < FS_PLAUSIBLE && good && pp < lim; k++) {
                    uint32_t payload = 0;
                    const uint32_t q = fs_next(bytes, pp, lim, &payload);
                    good = q != FS_ERR && fs_plausible_payload(bytes, payload, q);
                    pp = q;
                }
                if (good) {
                    ok = lane == l;
                    break;
                }
            }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        if (m != 0ull) {
            guess = begin + r + (uint32_t)__builtin_ctzll(m);
            break;
        }
    }
    if (lane == 0) {
        start[b] = guess;
        surv[b] = ns;
    }
}

static float time_it(const char* name, int reps, const std::function<void()>& f, double bytes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        sum += ms;
    }
    CK(hipGetLastError());
    printf("%-44s best %8.3f ms  avg %8.3f ms  %7.1f GB/s of stream\n", name, best, sum / reps, bytes / best / 1e6);
    return best;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const size_t len = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> h(len);
    if (fread(h.data(), 1, len, f) != len) return 2;
    fclose(f);
    const int copies = argc > 2 ? atoi(argv[2]) : 1;  // the stream repeated (a chain of frames stays one)
    const size_t total = len * copies;
    if (total >= 0xFFFFFFFFull) return 2;
    const uint32_t nb = (uint32_t)((total + FS_BLOCK - 1) / FS_BLOCK);
    uint8_t* d;
    CK(hipMalloc(&d, total + 64));
    for (int c = 0; c < copies; c++) CK(hipMemcpy(d + (size_t)c * len, h.data(), len, hipMemcpyHostToDevice));
    // the truth: where the first frame of every block starts (host walk over one copy, repeated)
    std::vector<uint32_t> truth(nb, 0), tcnt(nb, 0);
    size_t nframes = 0;
    {
        size_t p = 0;
        uint32_t b = 0;
        truth[0] = 0;
        while (p < total) {
            while ((size_t)(b + 1) * FS_BLOCK <= p && b + 1 < nb) truth[++b] = (uint32_t)p;  // (the first frame at or behind the block's begin)
            size_t q = p % len;
            uint64_t v = 0;
            int s = 0;
            for (;;) {
                const uint8_t x = h[q++];
                v |= (uint64_t)(x & 0x7f) << s;
                s += 7;
                if (!(x & 0x80)) break;
            }
            const size_t next = p + (q - p % len) + v;
            tcnt[p / FS_BLOCK]++;
            nframes++;
            p = next;
        }
    }
    printf("stream: %zu bytes x %d = %zu, %u blocks, %zu frames (%.1f B per frame)\n", len, copies, total, nb, nframes, (double)total / nframes);
    uint32_t *start, *surv, *cnt, *exits, *base, *off, *sink;
    uint8_t *err, *ent8;
    unsigned long long* present;
    CK(hipMalloc(&start, nb * 4));
    CK(hipMalloc(&surv, nb * 4));
    CK(hipMalloc(&cnt, nb * 4));
    CK(hipMalloc(&exits, nb * 4));
    CK(hipMalloc(&base, nb * 4));
    CK(hipMalloc(&sink, nb * 4));
    CK(hipMalloc(&err, nb));
    CK(hipMalloc(&ent8, (size_t)nb * 64));
    CK(hipMalloc(&present, (size_t)nb * 8));
    CK(hipMalloc(&off, (nframes + 1) * 4));
    CK(hipMemset(start, 0, nb * 4));
    CK(hipMemset(surv, 0, nb * 4));
    const dim3 gw((nb + 3) / 4), gl((nb + 255) / 256), bl(256);
    auto accuracy = [&](const char* what) {
        std::vector<uint32_t> g(nb), s(nb);
        CK(hipMemcpy(g.data(), start, nb * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s.data(), surv, nb * 4, hipMemcpyDeviceToHost));
        size_t right = 0, sum = 0, mx = 0;
        for (uint32_t b = 1; b < nb; b++) {
            right += g[b] == truth[b];
            sum += s[b];
            mx = std::max<size_t>(mx, s[b]);
        }
        printf("    %s: %zu of %u guesses right (%.2f %%), survivors of the deciding round: mean %.1f, max %zu\n", what, right, nb - 1, 100.0 * right / (nb - 1), (double)sum / (nb - 1), mx);
    };
    const int reps = 5;
    time_it("stage only (load - wait - write trips)", reps, [&] { hipLaunchKernelGGL(stage_only_kernel<false>, gw, bl, 0, 0, d, (uint32_t)total, nb, sink); }, total);
    time_it("stage only (product: loads in flight)", reps, [&] { hipLaunchKernelGGL(stage_only_kernel<true>, gw, bl, 0, 0, d, (uint32_t)total, nb, sink); }, total);
#define GP(PRE, FULL, TIERS, label)                                                                                                                      \
    time_it(label, reps, [&] { hipLaunchKernelGGL((fs_guess_kernel_t<PRE, FULL, TIERS>), gw, bl, 0, 0, d, (uint32_t)total, nb, start); }, total); \
    accuracy(label);
#define GX(NF, FULL, PIPE, label)                                                                                                                  \
    time_it(label, reps, [&] { hipLaunchKernelGGL((guess_x_kernel<NF, FULL, PIPE>), gw, bl, 0, 0, d, (uint32_t)total, nb, start, surv); }, total); \
    accuracy(label);
    GX(2, 1, false, "guess: prefilter 2, byte-wise parse of 2")
    GX(4, 0, false, "guess: prefilter 4 only (smallest survivor)")
    CK(hipMemset(surv, 0, nb * 4));
    GP(2, 2, false, "guess: prefilter 2, windowed parse of 2")
    GP(4, 2, false, "guess: prefilter 4, windowed parse of 2")
    GP(4, 1, false, "guess: prefilter 4, windowed parse of 1")
    GP(6, 2, false, "guess: prefilter 6, windowed parse of 2")
    GP(2, 2, true, "guess: prefilter 2, parse of 2, like-tagged first")
    GP(4, 2, true, "guess: prefilter 4, parse of 2, like-tagged first")
    GP(2, 1, true, "guess: prefilter 2, parse of 1, like-tagged first")
    GP(3, 1, true, "guess: prefilter 3, parse of 1, like-tagged first")
    GP(4, 1, true, "guess: prefilter 4, parse of 1, like-tagged first")
    GP(FS_PREFILTER, FS_PLAUSIBLE, FS_TIERS, "guess (product)")
    // leave the product's guess in place for the walk
    CK(hipMemcpy(start, truth.data(), nb * 4, hipMemcpyHostToDevice));  // (the proven chain: what the emit pass sees)
    time_it("walk (product, a lane per block)", reps, [&] { hipLaunchKernelGGL(fs_walk_kernel, gl, bl, 0, 0, d, (uint32_t)total, nb, (const uint32_t*)start, cnt, err, exits, ent8, present); }, total);
    {
        std::vector<uint32_t> c(nb), bs(nb);
        CK(hipMemcpy(c.data(), cnt, nb * 4, hipMemcpyDeviceToHost));
        size_t acc = 0, wrong = 0;
        for (uint32_t b = 0; b < nb; b++) {
            bs[b] = (uint32_t)acc;
            acc += c[b];
            wrong += c[b] != tcnt[b];
        }
        printf("    walk: %zu frames (truth %zu), %zu blocks with a wrong count\n", acc, nframes, wrong);
        CK(hipMemcpy(base, bs.data(), nb * 4, hipMemcpyHostToDevice));
    }
    auto check_off = [&](const char* what) {
        std::vector<uint32_t> o(nframes + 1);
        CK(hipMemcpy(o.data(), off, (nframes + 1) * 4, hipMemcpyDeviceToHost));
        size_t p = 0, i = 0, bad = 0;
        while (p < total) {
            bad += o[i++] != p;
            size_t q = p % len;
            uint64_t v = 0;
            int s = 0;
            for (;;) {
                const uint8_t x = h[q++];
                v |= (uint64_t)(x & 0x7f) << s;
                s += 7;
                if (!(x & 0x80)) break;
            }
            p += (q - p % len) + v;
        }
        bad += o[nframes] != total;
        printf("    %s: %zu wrong offsets\n", what, bad);
        CK(hipMemset(off, 0xee, (nframes + 1) * 4));
    };
    CK(hipMemset(off, 0xee, (nframes + 1) * 4));
    time_it("emit (product)", reps, [&] { hipLaunchKernelGGL(fs_emit_kernel, gw, bl, 0, 0, d, (uint32_t)total, nb, (const uint8_t*)ent8, (const unsigned long long*)present, (const uint32_t*)base, off, (uint32_t)nframes); }, total);
    check_off("emit (product)");
    time_it("emit (no staging, positions in registers)", reps, [&] { hipLaunchKernelGGL(emit_regs_kernel, gw, bl, 0, 0, d, (uint32_t)total, nb, (const uint8_t*)ent8, (const unsigned long long*)present, (const uint32_t*)base, off, (uint32_t)nframes); }, total);
    check_off("emit (registers)");
    return 0;
}
