"""ctypes wrapper of the CPU oracle (oracle/eigx_oracle.c).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from eigenexa_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liborc.so")
_lib = None
_dp = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = C.CDLL(_SO)
        lib.orc_eigen_sx.argtypes = lib.orc_eigen_s.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, C.c_int,
                                                                C.c_char, _dp]
        lib.orc_band_reduce.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int]
        lib.orc_band_dc.argtypes = [C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]
        lib.orc_band_bisect.argtypes = [C.c_int, _dp, _dp, C.c_int, C.c_int, _dp]
        lib.orc_gev.argtypes = [C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, _dp, C.c_int]
        lib.orc_eigen_h.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, _dp, C.c_void_p, C.c_int, C.c_char]
        lib.orc_scaling_sigma_reference.argtypes = [C.c_double]
        lib.orc_scaling_sigma_reference.restype = C.c_double
        lib.orc_trbak.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int]
        _lib = lib
    return _lib


def scaling_sigma_reference(anrm):
    """SIGMA of the reference's eigen_scaling for max |a_ij| = anrm (src/eigen_scaling.F:76-81, :127-135)"""
    return float(load().orc_scaling_sigma_reference(float(anrm)))


def threads():
    """OpenMP threads the oracle's parallel loops run on"""
    return int(load().orc_threads())


def host_cores():
    """CPU cores this process may actually use: the scheduler affinity mask cut by the cgroup CPU quota (a GPU box shows
    all of the host's logical CPUs but grants a share of them; more threads than that only add barrier time)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "quota period" or "max period"
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
                if q > 0:
                    n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def set_threads(n):
    load().orc_set_threads(int(n))
    return threads()


def _p(a):
    return a.ctypes.data_as(_dp)


def eigen(A, route="sx", mode="A"):
    """full solve of the symmetric matrix A (numpy, any order): returns (w, Z, stats, stage_seconds)."""
    lib = load()
    n = A.shape[0]
    a = np.asfortranarray(A, dtype=np.float64).copy(order="F")
    w = np.zeros(n)
    z = np.zeros((n, n), order="F")
    t = np.zeros(3)
    fn = lib.orc_eigen_sx if route == "sx" else lib.orc_eigen_s
    rc = fn(n, n, _p(a), n, _p(w), _p(z), n, mode.encode()[:1], _p(t))
    if rc not in (0, 1):
        raise RuntimeError(f"oracle failed rc={rc}")
    return w, z, a[: min(3, n), 0].copy(), t


def band_reduce(A, band):
    """returns d, e(band, n) with e[b-1, i] = T(i-b, i), and the matrix holding the reflectors."""
    lib = load()
    n = A.shape[0]
    a = np.asfortranarray(A, dtype=np.float64).copy(order="F")
    d = np.zeros(n)
    e = np.zeros((band, n))
    rc = lib.orc_band_reduce(n, _p(a), n, _p(d), _p(e), n, band)
    if rc != 0:
        raise RuntimeError(f"oracle band_reduce rc={rc}")
    return d, e, a


def band_dc(d, e, band):
    lib = load()
    n = len(d)
    w = np.zeros(n)
    z = np.zeros((n, n), order="F")
    fl = np.zeros(1)
    ee = np.ascontiguousarray(e, dtype=np.float64)
    rc = lib.orc_band_dc(n, _p(np.ascontiguousarray(d)), _p(ee), ee.shape[1], band, _p(w), _p(z), n, _p(fl))
    if rc != 0:
        raise RuntimeError(f"oracle band_dc rc={rc}")
    return w, z


def band_bisect(d, e, band):
    """eigenvalues only of the band matrix (d, e) by Sturm counts (eigen_bisect / eigen_bisect2)."""
    lib = load()
    n = len(d)
    w = np.zeros(n)
    ee = np.ascontiguousarray(e, dtype=np.float64)
    rc = lib.orc_band_bisect(n, _p(np.ascontiguousarray(d, dtype=np.float64)), _p(ee), ee.shape[1], band, _p(w))
    if rc != 0:
        raise RuntimeError(f"oracle band_bisect rc={rc}")
    return w


def gev(A, B):
    """KMATH_EIGEN_GEV restatement: returns (w, Z) of A x = lambda B x (Z is B-orthonormal); raises if B is not SPD."""
    lib = load()
    n = A.shape[0]
    a = np.asfortranarray(A, dtype=np.float64).copy(order="F")
    b = np.asfortranarray(B, dtype=np.float64).copy(order="F")
    w = np.zeros(n)
    z = np.zeros((n, n), order="F")
    rc = lib.orc_gev(n, _p(a), n, _p(b), n, _p(w), _p(z), n)
    if rc == 2:
        raise ValueError("Matrix B is not positive definite!")
    if rc != 0:
        raise RuntimeError(f"oracle gev rc={rc}")
    return w, z


def eigen_h(A, mode="A"):
    """eigen_h restatement (PARITY UNPINNED, see eigx_oracle.c): Hermitian A (complex128) -> (w, Z complex)."""
    lib = load()
    n = A.shape[0]
    a = np.asfortranarray(A, dtype=np.complex128).copy(order="F")
    w = np.zeros(n)
    z = np.zeros((n, n), dtype=np.complex128, order="F")
    rc = lib.orc_eigen_h(n, n, a.ctypes.data, n, _p(w), z.ctypes.data, n, mode.encode()[:1])
    if rc != 0:
        raise RuntimeError(f"oracle eigen_h rc={rc}")
    return w, z
