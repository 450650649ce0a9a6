"""RCCL on the GPU box (1 GPU => world size 1): the window-close exchange of dist.py through the `nccl` backend on
library-owned device buffers - out-of-place all-reduce of the sketches into the merged view, uneven all-gather of
every row kind's device buffer (fa_rows_device) followed by fa_rows_merge_device - must reproduce the single-rank
results bit for bit.  (The same code with two ranks runs over the gloo transport on the shared GPU in
test_ingest_sinks_gpu.py, the device merge of two contexts' rows in test_window_close_gpu.py; 8-GPU runs belong to the driver.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import _pkg
fa = _pkg.load(); po = _pkg.load_oracle()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 60000
gp = po.gen_params(mode=2, framed=1, seed=70, n_total=n, zipf_log2_universe=12)
buf, off = po.gen_records(gp, 0, n)
rows, status = po.decode_batch(buf, off, 1)
ks = 63
with fa.FlowAgg(framed=True, key_sets=ks, cms_width_log2=12, topk_capacity_log2=14) as agg:
    agg.ingest(buf, off)
    before = [agg.cms_read(k).copy() for k in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)]
    ports = [agg.top_ports(d) for d in (0, 1)]
    fa.dist.allreduce_sketches(agg)                      # RCCL all-reduce into the merged view, world 1: identity
    after = [agg.cms_read(k) for k in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)]
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
    assert all(fa.dist.top_ports_merged(agg, d, device=dev).tobytes() == ports[d].tobytes() for d in (0, 1))
    top = fa.dist.topk_merged(agg, fa.FA_KEYS_SRCADDR_CMS, 50, device=dev)
    assert top.tobytes() == agg.topk(fa.FA_KEYS_SRCADDR_CMS, 50).tobytes()
    assert fa.dist.minute_series_merged(agg, device=dev).tobytes() == po.minute_series(rows, status).tobytes()
    ref = po.Rollup(300); ref.ingest(buf, off, 1)
    app = fa.dist.close_window_app_merged(agg, fa.ALL_TIMESLOTS, device=dev)
    assert app.tobytes() == po.rollup_app(rows, status, 300).astype(fa.dist.ROW_APP_DTYPE).tobytes()
    merged = fa.dist.close_window_merged(agg, fa.ALL_TIMESLOTS, device=dev)
    assert merged.tobytes() == ref.rows().tobytes()
dist.destroy_process_group()
print("NCCL_OK")
'''


def test_window_close_exchange_over_rccl_world1(gpu_lib):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, r.stderr[-2000:]
