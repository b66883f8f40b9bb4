#!/usr/bin/env python3
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import oracle_lib as O
import vector_indexer_py as vip
dev = torch.device("cuda", 0)
n, d = 20000, 64
rng = np.random.default_rng(n + d)
X = rng.standard_normal((n, d)).astype(np.float32)
tmp = tempfile.mkdtemp()
orc = O.OracleIndex.build(X, tmp + "/index", tmp + "/shards")
gpu = vip.load(tmp + "/index", tmp + "/shards", d)
Q = np.concatenate([X[:100], (X[100:400] + 0.01 * rng.standard_normal((300, d))).astype(np.float32),
                    rng.standard_normal((100, d)).astype(np.float32) * float(np.abs(X).mean() + 1)])
xq = torch.from_numpy(Q).to(dev)
nq, k, P = Q.shape[0], 10, 8
def run(flt):
    os.environ["VI_FILTER"] = flt
    D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    T = torch.empty((nq, k), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    gpu.search_device(xq.data_ptr(), nq, k, P, D.data_ptr(), I.data_ptr(), T.data_ptr())
    return D.cpu().numpy(), I.cpu().numpy(), T.cpu().numpy().view(np.uint64)
D0, I0, T0 = run("0")
D1, I1, T1 = run("1")
bad = np.nonzero((I0 != I1).any(1))[0]
print("bad", bad, gpu.last_stats())
for q in bad[:4]:
    print("q", q)
    print(" valu I", I0[q].tolist()); print(" filt I", I1[q].tolist())
    print(" valu tie", [(int(t >> 32), int(t & 0xffffffff)) for t in T0[q]])
    print(" filt tie", [(int(t >> 32), int(t & 0xffffffff)) for t in T1[q]])
