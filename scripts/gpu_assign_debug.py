#!/usr/bin/env python3
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
from vector_indexer_py import _native as N
dev = torch.device("cuda", 0); L = N.lib()
def assign(X, Cn, env=None):
    n, d = X.shape; k = Cn.shape[0]
    lab = torch.empty(n, dtype=torch.int32, device=dev); st = N.AssignStats(); torch.cuda.synchronize()
    for key, val in (env or {}).items(): os.environ[key] = val
    N.check(L.vi_assign_device(0, X.data_ptr(), n, d, Cn.data_ptr(), k, 42, 1, lab.data_ptr(), C.byref(st)))
    for key in (env or {}): os.environ.pop(key, None)
    return lab, st
for n in (100_000, 300_000, 1_000_000):
    g = torch.Generator(device=dev); g.manual_seed(42)
    d, k = 128, 4096
    X = torch.randn(n, d, generator=g, device=dev)
    Cn = X[torch.randperm(n, generator=g, device=dev)[:k]].contiguous()
    a, sa = assign(X, Cn)
    b, sb = assign(X, Cn, {"VI_NO_MFMA": "1"})
    bad = torch.nonzero(a != b).flatten()
    print(n, "mismatch", bad.numel(), "amb", sa.ambiguous_rows, flush=True)
    if bad.numel():
        idx = bad[:8]
        x = X[idx].double()
        da = ((x - Cn[a[idx].long()].double()) ** 2).sum(1)
        db = ((x - Cn[b[idx].long()].double()) ** 2).sum(1)
        full = torch.cdist(x, Cn.double()) ** 2
        best = full.argmin(1)
        print(" rows", idx.tolist()); print(" mfma label", a[idx].tolist(), da.tolist()); print(" scan label", b[idx].tolist(), db.tolist()); print(" f64 argmin", best.tolist(), full.min(1).values.tolist())
        print(" bad row range", bad.min().item(), bad.max().item(), "first few", bad[:20].tolist())
