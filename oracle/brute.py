"""
oracle/brute.py -- independent brute-force checkers (numpy / pure Python).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package.

Two pieces, both independent of oracle/tda_oracle.c and of the HIP kernels:

* ``rips_brute``  -- textbook persistent homology: build the explicit Vietoris-Rips
  2-skeleton, sort simplices by (diameter, dimension, index), reduce the triangle
  boundary matrix over Z/2 column by column.  O(T * E) with Python big-int columns,
  meant for n <= ~50.  Conventions = the ones SURVEY.md section 8(c) records for
  ripser (float32 values, d <= thresh, zero-persistence pairs dropped, essential
  classes as +inf).  PARITY UNPINNED against the real ``ripser`` wheel (absent
  from /root/reference and from this image).

* ``wasserstein_persim`` -- line-by-line restatement of the published
  ``persim.wasserstein`` (persim >= 0.3, unpinned in requirements.txt:6) on top of
  the two solvers persim itself calls and which ARE installed:
  ``sklearn.metrics.pairwise_distances`` and ``scipy.optimize.linear_sum_assignment``.
  Call site in the reference: scripts/utils.py:189 (via safe_wasserstein, :180-191).
"""
import numpy as np


def eeg_prepare(dist, symmetrise=True):
    """scripts/utils.py:137-139 followed by ripser's float32 cast."""
    dm = np.asarray(dist, dtype=np.float64)
    if symmetrise:
        dm = (dm + dm.T) / 2
        np.fill_diagonal(dm, 0)
        dm = np.maximum(dm, 0)
    else:
        dm = np.triu(dm, 1)
        dm = dm + dm.T
    return dm.astype(np.float32)


def rips_brute(dm_f32, thresh=2.0):
    """Return (H0, H1) as float64 arrays of (birth, death) rows, sorted lexicographically."""
    d = np.asarray(dm_f32, dtype=np.float32)
    n = d.shape[0]
    thresh = np.float32(thresh)
    edges = [(float(d[i, j]), i, j) for i in range(n) for j in range(i) if d[i, j] <= thresh]
    edges.sort()
    epos = {(i, j): p for p, (_, i, j) in enumerate(edges)}
    # H0: union-find
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    h0 = []
    negative = set()
    ncomp = n
    for p, (w, i, j) in enumerate(edges):
        a, b = find(i), find(j)
        if a != b:
            parent[a] = b
            ncomp -= 1
            negative.add(p)
            if w != 0.0:
                h0.append((0.0, w))
    h0 += [(0.0, np.inf)] * ncomp
    # triangles
    tris = []
    for i in range(n):
        for j in range(i):
            if (i, j) not in epos:
                continue
            for k in range(j):
                if (i, k) in epos and (j, k) in epos:
                    dia = max(edges[epos[(i, j)]][0], edges[epos[(i, k)]][0], edges[epos[(j, k)]][0])
                    col = (1 << epos[(i, j)]) | (1 << epos[(i, k)]) | (1 << epos[(j, k)])
                    tris.append((dia, i, j, k, col))
    tris.sort(key=lambda t: t[:4])
    low2col = {}
    paired_edge = {}
    for dia, _, _, _, col in tris:
        while col:
            low = col.bit_length() - 1
            other = low2col.get(low)
            if other is None:
                low2col[low] = col
                paired_edge[low] = dia
                break
            col ^= other
    h1 = []
    for p, (w, i, j) in enumerate(edges):
        if p in negative:
            continue
        if p in paired_edge:
            if paired_edge[p] > w:
                h1.append((w, paired_edge[p]))
        else:
            h1.append((w, np.inf))
    h0 = np.array(sorted(h0), dtype=np.float64).reshape(-1, 2)
    h1 = np.array(sorted(h1), dtype=np.float64).reshape(-1, 2)
    return h0, h1


def sort_rows(dgm):
    dgm = np.asarray(dgm, dtype=np.float64).reshape(-1, 2)
    if len(dgm) == 0:
        return dgm
    order = np.lexsort((dgm[:, 1], dgm[:, 0]))
    return dgm[order]


def clean(d):
    """scripts/utils.py:182-187."""
    d = np.asarray(d)
    if d.ndim != 2 or d.shape[0] == 0:
        return np.array([[0, 0]])
    m = np.isfinite(d).all(axis=1)
    d = d[m]
    return d if len(d) > 0 else np.array([[0, 0]])


def wasserstein_persim(dgm1, dgm2):
    """persim.wasserstein(dgm1, dgm2, matching=False) restated."""
    from scipy import optimize
    from sklearn import metrics

    S = np.array(dgm1)
    M = min(S.shape[0], S.size)
    if S.size > 0:
        S = S[np.isfinite(S[:, 1]), :]
        M = S.shape[0]
    T = np.array(dgm2)
    N = min(T.shape[0], T.size)
    if T.size > 0:
        T = T[np.isfinite(T[:, 1]), :]
        N = T.shape[0]
    if M == 0:
        S = np.array([[0, 0]])
        M = 1
    if N == 0:
        T = np.array([[0, 0]])
        N = 1
    DUL = metrics.pairwise.pairwise_distances(S, T)
    cp = np.cos(np.pi / 4)
    sp = np.sin(np.pi / 4)
    R = np.array([[cp, -sp], [sp, cp]])
    S = S[:, 0:2].dot(R)
    T = T[:, 0:2].dot(R)
    D = np.zeros((M + N, M + N))
    D[0:M, 0:N] = DUL
    UR = np.inf * np.ones((M, M))
    np.fill_diagonal(UR, S[:, 1])
    D[0:M, N:N + M] = UR
    UL = np.inf * np.ones((N, N))
    np.fill_diagonal(UL, T[:, 1])
    D[M:N + M, 0:N] = UL
    matchi, matchj = optimize.linear_sum_assignment(D)
    return float(np.sum(D[matchi, matchj]))


def safe_wasserstein_oracle(dgm1, dgm2):
    """scripts/utils.py:180-191 on top of the persim restatement."""
    try:
        return wasserstein_persim(clean(np.asarray(dgm1)), clean(np.asarray(dgm2)))
    except Exception:
        return np.nan


def wasserstein_bruteforce(A, B):
    """Exhaustive optimum over all partial matchings (M+N <= 9), for known-answer tests."""
    import itertools
    A = np.asarray(A, float).reshape(-1, 2)
    B = np.asarray(B, float).reshape(-1, 2)
    M, N = len(A), len(B)
    s = (A[:, 1] - A[:, 0]) / np.sqrt(2.0)
    t = (B[:, 1] - B[:, 0]) / np.sqrt(2.0)
    C = np.sqrt(((A[:, None, :] - B[None, :, :]) ** 2).sum(-1))
    best = np.inf
    # assign every A point either to a distinct B point or to the diagonal (-1)
    for choice in itertools.product(range(-1, N), repeat=M):
        used = [c for c in choice if c >= 0]
        if len(set(used)) != len(used):
            continue
        tot = sum(C[i, c] if c >= 0 else s[i] for i, c in enumerate(choice))
        tot += sum(t[j] for j in range(N) if j not in used)
        best = min(best, tot)
    return float(best)
