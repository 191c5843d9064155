"""2-D cyclic layout helpers and the benchmark's synthetic matrices (host logic, numpy only).

Layout rules of the reference (src/eigen_libs0.F:526-570 grid shape, :1825-2258 index maps):
global (i, j), 0-based, lives on grid coordinate (i % Px, j % Py) at local (i // Px, j // Py);
ranks are numbered column-major over the grid (rank = px + py*Px) unless order == 'R'.
Matrix generators follow benchmark/mat_set.f (Frank :117-132, analytic spectrum :638-647) with a
counter-based random generator keyed on global indices so every layout sees the same matrix
(SURVEY.md 8d).
"""
import numpy as np


def grid_shape(nranks):
    """Px = largest divisor of nranks that is <= sqrt(nranks); Py = nranks // Px."""
    px = 1
    x = 1
    while x * x <= nranks:
        if nranks % x == 0:
            px = x
        x += 1
    return px, nranks // px


def rank_coords(rank, nranks, order="C"):
    px, py = grid_shape(nranks)
    if str(order)[:1].upper() == "R":
        return rank // py, rank % py
    return rank % px, rank // px


def local_count(n, p, nprocs):
    """number of global indices g < n with g % nprocs == p"""
    return (n - p + nprocs - 1) // nprocs if n > p else 0


def scatter_cyclic(a_global, nranks, rank, order="C", nx=None, ny=None):
    """local block (Fortran order, shape (nx, ny)) of a global matrix for `rank`."""
    n0, n1 = a_global.shape
    Px, Py = grid_shape(nranks)
    px, py = rank_coords(rank, nranks, order)
    loc = a_global[px::Px, py::Py]
    nx = nx or loc.shape[0]
    ny = ny or loc.shape[1]
    out = np.zeros((nx, ny), dtype=a_global.dtype, order="F")
    out[: loc.shape[0], : loc.shape[1]] = loc
    return out


def gather_cyclic(blocks, n0, n1, order="C", dims=None):
    """inverse of scatter_cyclic: blocks[rank] -> global (n0, n1) matrix; dims = explicit (Px, Py) grid"""
    nranks = len(blocks)
    Px, Py = dims or grid_shape(nranks)
    out = np.zeros((n0, n1), dtype=blocks[0].dtype)
    for rank, b in enumerate(blocks):
        if dims:
            px, py = (rank // Py, rank % Py) if order in ("R", "r") else (rank % Px, rank // Px)
        else:
            px, py = rank_coords(rank, nranks, order)
        r = local_count(n0, px, Px)
        c = local_count(n1, py, Py)
        out[px::Px, py::Py] = b[:r, :c]
    return out


def numroc(n, nb, p, nprocs):
    """ScaLAPACK NUMROC (source process 0): local extent of n indices dealt in blocks of nb to process p"""
    nblocks = n // nb
    cnt = (nblocks // nprocs) * nb
    extra = nblocks % nprocs
    if p < extra:
        cnt += nb
    elif p == extra:
        cnt += n % nb
    return cnt


def block_cyclic_indices(n, nb, p, nprocs):
    """global indices (ascending) owned by process p of a 1-D block-cyclic distribution; nb = 1 is cyclic"""
    g = np.arange(n)
    return g[(g // nb) % nprocs == p]


def scatter_block_cyclic(a_global, nb, nranks, rank, order="C"):
    """local block (Fortran order) of `rank` for the 2-D block-cyclic (nb x nb) distribution of a global matrix"""
    Px, Py = grid_shape(nranks)
    px, py = rank_coords(rank, nranks, order)
    rows = block_cyclic_indices(a_global.shape[0], nb, px, Px)
    cols = block_cyclic_indices(a_global.shape[1], nb, py, Py)
    return np.asfortranarray(a_global[np.ix_(rows, cols)])


def gather_block_cyclic(blocks, n0, n1, nb, order="C", dims=None):
    """inverse of scatter_block_cyclic: blocks[rank] -> global (n0, n1) matrix; dims = explicit (Px, Py) grid"""
    nranks = len(blocks)
    Px, Py = dims or grid_shape(nranks)
    out = np.zeros((n0, n1), dtype=blocks[0].dtype)
    for rank, b in enumerate(blocks):
        if dims:
            px, py = (rank // Py, rank % Py) if order in ("R", "r") else (rank % Px, rank // Px)
        else:
            px, py = rank_coords(rank, nranks, order)
        rows = block_cyclic_indices(n0, nb, px, Px)
        cols = block_cyclic_indices(n1, nb, py, Py)
        out[np.ix_(rows, cols)] = b[: len(rows), : len(cols)]
    return out


# ---- synthetic matrices -------------------------------------------------------------------------

def frank(n, rows=None, cols=None):
    """Frank matrix a_ij = min(i, j), 1-based (benchmark/mat_set.f:117-132); optional index subsets."""
    i = (np.arange(n) if rows is None else np.asarray(rows)) + 1
    j = (np.arange(n) if cols is None else np.asarray(cols)) + 1
    return np.minimum.outer(i, j).astype(np.float64)


def frank_eigenvalues(n):
    """analytic spectrum 1/(2(1-cos((2k-1)pi/(2n+1)))), ascending (benchmark/mat_set.f:638-647)."""
    k = np.arange(1, n + 1)
    return np.sort(1.0 / (2.0 * (1.0 - np.cos((2 * k - 1) * np.pi / (2 * n + 1)))))


def toeplitz(n):
    """Toeplitz matrix: -7.2 on the diagonal, -3/(i-j)^2 elsewhere (benchmark/mat_set.f:134-152)."""
    i = np.arange(n, dtype=np.float64)
    d = i[:, None] - i[None, :]
    with np.errstate(divide="ignore"):
        A = -3.0 / (d * d)
    A[np.diag_indices(n)] = -7.2
    return A


def frank2(n):
    """'Frank matrix 2': a_ij = n + 1 - max(i, j), 1-based (benchmark/mat_set.f:188-202); same spectrum as
    the Frank matrix (benchmark/mat_set.f:638-647 lists both under one formula)."""
    i = np.arange(1, n + 1)
    return (n + 1 - np.maximum.outer(i, i)).astype(np.float64)


def helmert(n):
    """orthogonal Helmert matrix H (rows): row 1 = 1/sqrt(n); row i >= 2 has i-1 entries 1/sqrt(i(i-1)),
    then -(i-1)/sqrt(i(i-1)), then zeros (benchmark/mat_set.f:395-423)."""
    H = np.zeros((n, n))
    H[0, :] = 1.0 / np.sqrt(n)
    for i in range(2, n + 1):
        c = 1.0 / np.sqrt(float(i) * (i - 1))
        H[i - 1, : i - 1] = c
        H[i - 1, i - 1] = -(i - 1) * c
    return H


def spectrum(n, mtype, seed=7):
    """prescribed spectra of the reference's matrix types 4..9 (benchmark/mat_set.f:651-718):
    4: 0..n-1;  5: sin(5 pi i/(n-1) + eps^(1/4))^3;  6: mod(i,5)+mod(i,2) (heavily degenerate);
    7: the Frank spectrum;  8: uniform [0,1);  9: normal(0,1);  10: the file W.dat.  Types 8/9 use the compiler RNG in the
    reference (not reproducible elsewhere); here a seeded numpy generator."""
    i = np.arange(1, n + 1, dtype=np.float64)
    if mtype == 4:
        return i - 1.0
    if mtype == 5:
        return np.sin(np.pi * 5.0 * i / max(n - 1, 1) + np.finfo(np.float64).eps ** 0.25) ** 3
    if mtype == 6:
        return np.mod(i, 5) + np.mod(i, 2)
    if mtype == 7:
        return frank_eigenvalues(n)
    if mtype == 10:
        # benchmark/W.dat (matrix type 10 of the driver, benchmark/mat_set.f:205-216, :714-729; also the spectrum file of the
        # KMATH_EIGEN_GEV driver, benchmark/KMATH_EIGEN_GEV_main.f:57-58).  A 'W.dat' in the working directory is read as
        # the reference reads it (free format, the first n numbers); without one the file's content is regenerated: its
        # k-th entry is 10 + sin(k-1) printed with six significant digits (checked against all 100000 entries)
        import os

        if os.path.exists("W.dat"):
            vals = np.array(open("W.dat").read().split()[:n], dtype=np.float64)
            if len(vals) < n:
                raise ValueError(f"W.dat holds {len(vals)} numbers, matrix type 10 needs {n}")
            return vals
        v = 10.0 + np.sin(i - 1.0)
        return np.where(v >= 10.0, np.round(v, 4), np.round(v, 5))
    rng = np.random.default_rng(seed)
    if mtype == 8:
        return rng.random(n)
    if mtype == 9:
        return rng.standard_normal(n)
    raise ValueError(f"no prescribed spectrum for matrix type {mtype}")


def helmert_spectrum_matrix(n, mtype, seed=7):
    """A = H^T diag(w) H with the spectrum of `mtype` in a seeded random order, w scaled by 1/max(1, max|w|)
    exactly as helmert_trans does (benchmark/mat_set.f:336-456).  Returns (A, ascending eigenvalues)."""
    w = spectrum(n, mtype, seed)
    w = w / max(1.0, np.abs(w).max())
    wp = np.random.default_rng(seed + 1).permutation(w)
    H = helmert(n)
    A = (H.T * wp[None, :]) @ H
    return 0.5 * (A + A.T), np.sort(w)


def matrix_market_dim(path):
    """order of the matrix in a Matrix-Market coordinate file (mat_dim_get, benchmark/mat_set.f:461-533): the first line
    that does not start with '%' holds "rows cols entries"; returns n or raises ValueError on a non-square size"""
    with open(path) as f:
        for line in f:
            if not line.startswith("%"):
                n1, n2, _ = (int(v) for v in line.split()[:3])
                if n1 != n2:
                    raise ValueError("Matrix size inconsistency has been found.")
                return n1
    raise ValueError(f"no size line in {path}")


def read_matrix_market(path, n):
    """symmetric matrix from a Matrix-Market coordinate file, as the reference driver reads it for matrix types -1
    ('A.mtx') and -2 ('B.mtx') (benchmark/mat_set.f:218-330): "i j value" triples, 1-based, every entry set at (i, j)
    and (j, i); entries that are not listed are zero"""
    A = np.zeros((n, n))
    with open(path) as f:
        for line in f:
            if not line.startswith("%"):
                n1, n2, ne = (int(v) for v in line.split()[:3])
                break
        else:
            raise ValueError(f"no size line in {path}")
        if n1 != n or n2 != n:
            raise ValueError("Matrix size inconsistency has been found.")
        k = 0
        for line in f:
            t = line.split()
            if len(t) < 3:
                continue
            i, j, v = int(t[0]) - 1, int(t[1]) - 1, float(t[2].replace("D", "E").replace("d", "e"))
            A[i, j] = v
            A[j, i] = v
            k += 1
            if k == ne:
                break
    return A


def reference_matrix(n, mtype, seed=7):
    """the reference benchmark's matrix families by its type number (benchmark/mat_set.f:566-595);
    returns (A, known ascending eigenvalues or None).  Types -1 / -2 read 'A.mtx' / 'B.mtx' from the working directory."""
    if mtype in (-1, -2):
        return read_matrix_market("A.mtx" if mtype == -1 else "B.mtx", n), None
    if mtype == 0:
        return frank(n), frank_eigenvalues(n)
    if mtype == 1:
        return toeplitz(n), None
    if mtype == 2:
        return random_symmetric(n, seed=seed), None
    if mtype == 3:
        return frank2(n), frank_eigenvalues(n)
    return helmert_spectrum_matrix(n, mtype, seed)


def _mix64(x):
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def random_symmetric(n, seed=20240807, rows=None, cols=None):
    """A = R + R^T with r_ij uniform [0,1) from a counter-based generator keyed (seed, i, j)."""
    i = (np.arange(n) if rows is None else np.asarray(rows)).astype(np.uint64)
    j = (np.arange(n) if cols is None else np.asarray(cols)).astype(np.uint64)
    with np.errstate(over="ignore"):
        def r(ii, jj):
            key = (ii[:, None] * np.uint64(n) + jj[None, :]) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
            return (_mix64(key) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return r(i, j) + r(j, i).T


def random_hermitian(n, seed=20240807, rows=None, cols=None):
    """the matrix family of the reference's eigen_h driver (benchmark_h/mat_set_h.f:36-64): S with real and imaginary parts
    uniform in [-1/2, 1/2) (real on the diagonal), A = S + S^H.  The reference draws S from the compiler's RNG seeded with
    the rank id; here from the counter-based generator of random_symmetric() at the global indices (layout-independent)."""
    i = (np.arange(n) if rows is None else np.asarray(rows)).astype(np.uint64)
    j = (np.arange(n) if cols is None else np.asarray(cols)).astype(np.uint64)
    with np.errstate(over="ignore"):
        def u(ii, jj, plane):
            key = (ii[:, None] * np.uint64(n) + jj[None, :]) + np.uint64(seed + plane) * np.uint64(0x9E3779B97F4A7C15)
            return (_mix64(key) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) - 0.5

        def s(ii, jj):
            im = np.where(ii[:, None] == jj[None, :], 0.0, u(ii, jj, 1))
            return u(ii, jj, 0) + 1j * im
        return s(i, j) + s(j, i).conj().T


def accuracy_metrics(A, w, Z):
    """the reference's three gates: residual ||AZ-ZW||_F/(N eps ||A||_F) (< 768), orthogonality
    ||Z^T Z - I||_F/(N eps) (< 8)  (benchmark/ev_test.f:181-204)."""
    n = A.shape[0]
    eps = np.finfo(np.float64).eps
    res = np.linalg.norm(A @ Z - Z * w[None, : Z.shape[1]]) / (n * eps * np.linalg.norm(A))
    orth = np.linalg.norm(Z.T @ Z - np.eye(Z.shape[1])) / (n * eps)
    return res, orth


def random_symmetric_torch(n, device, seed=20240807, rows=None, cols=None, chunk=2048):
    """same matrix as random_symmetric(), generated on `device` with torch int64 arithmetic (wrap-around
    multiplication, logical shifts emulated), in column chunks; returns the (len(rows) x len(cols)) block."""
    import torch

    M64 = (1 << 64) - 1

    def to_i64(u):  # python unsigned 64-bit constant -> signed value with the same bits
        u &= M64
        return u - (1 << 64) if u >= (1 << 63) else u

    C1, C2 = to_i64(0xBF58476D1CE4E5B9), to_i64(0x94D049BB133111EB)
    base = to_i64(seed * 0x9E3779B97F4A7C15)

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def mix(x):
        x = (x ^ lsr(x, 30)) * C1
        x = (x ^ lsr(x, 27)) * C2
        return x ^ lsr(x, 31)

    def r(ii, jj):
        key = ii[:, None] * n + jj[None, :] + base
        return lsr(mix(key), 11).to(torch.float64) * (1.0 / 9007199254740992.0)

    ri = torch.arange(n, device=device) if rows is None else torch.as_tensor(rows, device=device)
    ci = torch.arange(n, device=device) if cols is None else torch.as_tensor(cols, device=device)
    ri = ri.to(torch.int64)
    ci = ci.to(torch.int64)
    out = torch.empty(len(ri), len(ci), dtype=torch.float64, device=device)
    for c0 in range(0, len(ci), chunk):
        cj = ci[c0:c0 + chunk]
        out[:, c0:c0 + chunk] = r(ri, cj) + r(cj, ri).T
    return out
