"""D2H copy rates of one 663 MB row set (a config-5 window) into page-locked memory: one hipMemcpyAsync, the same copy cut into
2 / 4 / 8 pieces on as many streams (several SDMA engines), and a copy KERNEL that stores into the mapped host buffer."""
import json
import time

import torch

n = 663_000_000
dev = torch.device("cuda", 0)
src = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
dst = torch.empty(n, dtype=torch.uint8, pin_memory=True)
out = {"bytes": n}


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


t = timed(lambda: dst.copy_(src, non_blocking=True))
out["one_copy_GBps"] = n / t / 1e9
for k in (2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(k)]
    step = (n + k - 1) // k

    def f():
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                dst[i * step:(i + 1) * step].copy_(src[i * step:(i + 1) * step], non_blocking=True)
    t = timed(f)
    out["%d_streams_GBps" % k] = n / t / 1e9
print(json.dumps(out))
