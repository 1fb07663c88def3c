import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import make_corpus
import kektordb_amd as hip
from oracle import oracle as O
O.build()
for (n, dim, keep, ef, m) in [(12000, 16, 300, 64, 16), (12000, 16, 300, 300, 16), (20000, 8, 100, 200, 16), (20000, 32, 50, 400, 32)]:
    X = make_corpus(n, dim, "normal", seed=77)
    rng = np.random.default_rng(78)
    deleted = rng.choice(np.arange(1, n + 1), n - keep, replace=False)
    idx = hip.HipIndex(dim, 0, 0, m, 40, capacity=n + 8)
    idx.upload_rows(X, 1)
    idx.build(n, batch=512, ef_construction=40, seed=3)
    c, e, ml, levels, offs, nbrs = idx.download_graph()
    lvl0 = np.nonzero(levels[1:n + 1] == 0)[0] + 1            # nodes that exist on the bottom layer only: the descent stays intact
    deleted = rng.choice(lvl0, lvl0.size - keep, replace=False)
    idx.Delete(deleted.tolist())
    Q = make_corpus(32, dim, "normal", seed=79)
    ids, dist, cnt, (nd, nh) = idx.search_batch(Q, 10, ef, trace=True)
    print((n, dim, keep, ef, m), "dropped", idx.counters()["n_dropped"], "n_dist mean", nd.mean(), "hops", nh.mean(), "cnt", cnt.mean())
