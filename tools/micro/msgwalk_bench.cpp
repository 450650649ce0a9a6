// msgwalk_bench: what the consumer's bulk take loop (host/inserter_gpu.cpp, ConsumerGroupClaim::next_run) costs per message on a host:
// the walk over a mapped partition log is a dependent chain (a length byte tells where the next message starts).
//   g++ -O2 -o msgwalk_bench msgwalk_bench.cpp && ./msgwalk_bench p0.log <MAP_POPULATE 0|1> <touch the offsets first 0|1> <prefetch distance in bytes, 0 = none>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    int fd = open(argv[1], O_RDONLY);
    struct stat st; fstat(fd, &st);
    size_t size = st.st_size;
    int populate = atoi(argv[2]); const size_t pf = (size_t)atoi(argv[4]);
    double t0 = now();
    const uint8_t* base = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE | (populate ? MAP_POPULATE : 0), fd, 0);
    double t1 = now();
    size_t cap = size / 48 + 16;
    uint64_t* ends = (uint64_t*)malloc(cap * 8);
    if (atoi(argv[3])) memset(ends, 0, cap * 8);
    double t2 = now();
    for (int rep = 0; rep < 3; rep++) {
        size_t pos = 0, n = 0, bytes = 0;
        const size_t safe_end = size - 256;
        double ta = now();
        while (pos < safe_end) {
            const uint8_t b = base[pos];
            if (b & 0x80) break;
            if (pf) __builtin_prefetch(base + pos + pf);
            pos += 1u + (size_t)b;
            bytes += 1u + (size_t)b;
            ends[n++] = bytes;
        }
        double tb = now();
        printf("rep %d: %zu msgs, %.2f ns per message (map %.3f s, ends %.3f s)\n", rep, n, (tb - ta) / n * 1e9, t1 - t0, t2 - t1);
    }
}
