"""A second, independent restatement of the reference's sequential insert -- `Index.Add` (pkg/core/hnsw/hnsw_index.go:472-809),
`searchLayerUnlocked` (:2351-2611), `selectNeighbors` (:2629-2701), the two heaps (hnsw_heap.go:33-82,108-151) -- written in
plain Python straight from the Go text, and a trace test that puts it next to the C oracle (oracle/kdb_oracle.c orc_index_add)
insert by insert: after EVERY insert every adjacency list of every node at every level must be equal, entry point and maximum
level included.  Coordinates are multiples of 1/64 in six columns, so every squared distance is exact in float32 whatever the
accumulation order: the comparison is about the ALGORITHM (zoom-in with ef = 1, the candidates returned, the neighbours
selected, every reverse-link rewrite -- fast path below maxM, re-selection over the UNSORTED list of the neighbour's current
links plus the new node with distances neighbour <-> link, :748-771), equal distances included (they are frequent here and are
ordered by the heaps' history in both).

Why: the reference's own gate for this path, clients/python/stress_test_recall.py:60-87, asserts mean recall@10 >= 0.95 on
10 000 x 64 uniform rows (M 16, efConstruction 200, queries = stored vectors, ef_search 0 -> ef = k = 10, hnsw_index.go:2377-2380;
the client's docstring says "0 = the server's efConstruction" but no code path does that: ops.go:1006 passes the 0 through).  The
C restatement gives 0.39 there.  Two readings of the Go text that agree list for list make a slip of the restatement unlikely:
the number is the reference algorithm's (test_recall_protocol_of_the_reference_script records it)."""
import math

import numpy as np
import pytest


# ---- hnsw_heap.go: value heaps with strict comparisons, parent (j-1)/2, Pop moves the last element to the root ----------------
class Heap:
    def __init__(self, is_max):
        self.a = []
        self.is_max = is_max

    def before(self, x, y):  # h[j].Distance < h[i].Distance (min) / > (max)
        return x > y if self.is_max else x < y

    def push(self, item):  # item = (id, dist)
        self.a.append(item)
        j = len(self.a) - 1
        while True:
            i = (j - 1) // 2 if j > 0 else 0   # Go: (j-1)/2 truncates toward zero: (0-1)/2 == 0
            if i == j or not self.before(self.a[j][1], self.a[i][1]):
                break
            self.a[i], self.a[j] = self.a[j], self.a[i]
            j = i

    def pop(self):
        a = self.a
        n = len(a)
        x = a[0]
        a[0] = a[n - 1]
        a.pop()
        n -= 1
        if n > 0:
            i = 0
            while True:
                j1 = 2 * i + 1
                if j1 >= n:
                    break
                j = j1
                j2 = j1 + 1
                if j2 < n and self.before(a[j2][1], a[j1][1]):
                    j = j2
                if not self.before(a[j][1], a[i][1]):
                    break
                a[i], a[j] = a[j], a[i]
                i = j
        return x


class PyIndex:
    """hnsw.Index reduced to what Add touches: float32 rows, squared L2, no deletes, no allow list."""

    def __init__(self, dim, m, efc):
        self.dim, self.m, self.mmax0, self.efc = dim, m, 2 * m, efc   # New(): mMax0 = m * 2
        self.rows = {}
        self.conn = {}            # id -> list of lists (Connections[level])
        self.entry, self.max_level, self.counter = 0, -1, 0

    def dist(self, a, b):         # distance.SquaredEuclidean on exact inputs
        d = self.rows[a] - self.rows[b]
        return float(np.dot(d, d))

    def qdist(self, q, b):
        d = q - self.rows[b]
        return float(np.dot(d, d))

    def search_layer(self, q, ep, k, level, ef_search):   # :2351-2611
        ef = max(ef_search, k)                              # :2377-2380
        visited = set()
        cands, results = Heap(False), Heap(True)
        e = (ep, self.qdist(q, ep))
        cands.push(e)
        visited.add(ep)
        results.push(e)                                     # (no allow list, nothing deleted)
        while cands.a:
            cur = cands.pop()
            if len(results.a) >= ef and cur[1] > results.a[0][1]:   # :2501-2506, strict
                break
            cl = self.conn.get(cur[0])
            if cl is None or level >= len(cl):              # :2524-2527
                continue
            for nb in list(cl[level]):                      # the copied slice, stored order
                if nb in visited:
                    continue
                visited.add(nb)
                if nb not in self.conn:
                    continue
                d = self.qdist(q, nb)
                worst = results.a[0][1] if results.a else 1.7976931348623157e308
                if len(results.a) < ef or d < worst:        # :2577
                    c = (nb, d)
                    cands.push(c)
                    results.push(c)
                    if len(results.a) > ef:
                        results.pop()
        out = [None] * len(results.a)
        for i in range(len(results.a) - 1, -1, -1):         # :2596-2604: drained from the back
            out[i] = results.pop()
        return out[:k]

    def select_neighbors(self, cands, m, base_dist=None):   # :2629-2701, candidates in the GIVEN order
        if len(cands) <= m:
            return list(cands)
        results, discarded = [], []
        for e in cands:
            if len(results) >= m:
                break
            if not results:
                results.append(e)
                continue
            good = True
            for r in results:
                if self.dist(e[0], r[0]) < e[1]:            # d(e, r) < e.Distance (both metrics)
                    good = False
                    break
            (results if good else discarded).append(e)
        if len(results) < m:                                # fill from the discarded, in order
            results += discarded[:m - len(results)]
        return results

    def add(self, vec, level):                              # :472-809
        self.counter += 1
        nid = self.counter
        self.rows[nid] = np.asarray(vec, dtype=np.float64)
        if level > self.max_level + 1:                      # randomLevel's cap (:2620-2623)
            level = self.max_level + 1
        self.conn[nid] = [[] for _ in range(level + 1)]
        if self.max_level == -1:                            # first node (:656-670)
            self.entry, self.max_level = nid, level
            return nid
        cur_max, ep, q = self.max_level, self.entry, self.rows[nid]
        for l in range(cur_max, level, -1):                 # zoom in, k = 1, efSearch = 1 (:685-690)
            near = self.search_layer(q, ep, 1, l, 1)
            if near:
                ep = near[0][0]
        for l in range(min(level, cur_max), -1, -1):        # :698-789
            cands = self.search_layer(q, ep, self.efc, l, self.efc)
            maxm = self.mmax0 if l == 0 else self.m
            sel = self.select_neighbors(cands, maxm)
            self.conn[nid][l] = [c[0] for c in sel]         # forward links
            for nb, _ in sel:                               # reverse links (:725-783)
                cl = self.conn[nb]
                cur = list(cl[l]) if l < len(cl) else []
                if len(cur) < maxm:
                    final = cur + [nid]
                else:
                    allc = [(x, self.dist(nb, x)) for x in cur if x in self.conn]   # neighbour <-> its links, stored order
                    allc.append((nid, self.dist(nb, nid)))
                    final = [c[0] for c in self.select_neighbors(allc, maxm)]
                while len(cl) <= l:
                    cl.append([])
                cl[l] = final
            if cands:
                ep = cands[0][0]                            # :786-788
        if level > cur_max:                                 # :793-801
            self.max_level, self.entry = level, nid
        return nid


def _lists_of(graph, node, levels):
    out = []
    for l in range(int(levels[node]) + 1):
        off = graph.offsets[l]
        out.append(graph.neighbors[l][int(off[node]):int(off[node + 1])].tolist())
    return out


@pytest.mark.parametrize("m,efc,n,seed", [(4, 12, 260, 1), (3, 8, 200, 2), (8, 20, 200, 3)])
def test_add_trace_two_restatements_agree_insert_by_insert(oracle, m, efc, n, seed):
    O = oracle
    dim = 6
    rng = np.random.default_rng(seed)
    X = (rng.integers(0, 64, size=(n, dim)) / 64.0).astype(np.float32)
    X[rng.choice(n, 12, replace=False)] = X[3]          # duplicates: distance 0 ties, heap history decides
    ml = 1.0 / math.log(m)
    levels = [int(math.floor(-math.log(max(rng.random(), 1e-12)) * ml)) for _ in range(n)]
    py = PyIndex(dim, m, efc)
    orc = O.OracleIndex(dim, O.L2, O.F32, m, efc, seed=7)
    orc.set_arith(O.ARITH_GO)
    rewrites = 0
    for i in range(n):
        full_before = {k for k, v in py.conn.items() if len(v[0]) >= 2 * m}
        a = py.add(X[i], levels[i])
        b = orc.add(X[i], levels[i])
        assert a == b == i + 1
        assert (py.entry, py.max_level) == (orc.entry, orc.max_level), i
        rewrites += sum(1 for x in py.conn[a][0] if x in full_before)
        if i % 10 == 9 or i == n - 1 or i < 40:            # every list of every node (all inserts early on, then every tenth)
            g = orc.export_graph()
            for node in range(1, i + 2):
                assert int(g.levels[node]) + 1 == len(py.conn[node]), (i, node)
                assert _lists_of(g, node, g.levels) == py.conn[node], (i, node, _lists_of(g, node, g.levels), py.conn[node])
    assert rewrites > n // 2    # the re-selection path (:748-771) was exercised, not only the fast path


def test_recall_protocol_of_the_reference_script(oracle):
    """clients/python/stress_test_recall.py:11-87 as written: 10 000 x 64 uniform rows, one Add per row, M 16, efConstruction 200,
    queries = STORED vectors, ef_search 0 (ef = k = 10).  The script asserts >= 0.95; it needs a running Go server and is not in
    the reference's CI.  The restated Add -- two independent restatements agree list for list, above -- gives ~0.39 (0.68 at
    ef 100): a full neighbour re-selects over its links in STORED order and the fill-up keeps the first discarded ones, so the
    candidate dropped is usually the newest node (:748-771, :2688-2698) and late inserts are poorly reachable.  Recorded, not
    asserted against the script's bar; what is asserted is that this number is stable (so a change of the restatement shows)."""
    O = oracle
    n, dim, k = 10000, 64, 10
    rng = np.random.default_rng(42)
    X = rng.random((n, dim), dtype=np.float32)
    idx = O.OracleIndex(dim, O.L2, O.F32, 16, 200, seed=42)
    idx.set_arith(O.ARITH_RUST)
    idx.add_many(X)
    rows = idx.rows()
    pick = rng.integers(0, n, 40)
    rec = {}
    for ef in (0, 100):
        hit = 0
        for i in pick:
            bi, _ = O.bruteforce_l2_f64(rows, X[i], k)
            ids, _ = idx.search(X[i], k, ef=ef)
            hit += len(set(ids.tolist()) & set(bi.tolist()))
        rec[ef] = hit / (k * len(pick))
    print(f"stress_test_recall.py protocol on the restated Add: recall@10 {rec[0]:.3f} at ef_search 0, {rec[100]:.3f} at ef_search 100")
    assert 0.2 <= rec[0] <= 0.7 and rec[0] < rec[100] <= 0.9, rec   # the script's 0.95 is not reproduced by its own algorithm
