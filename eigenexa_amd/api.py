"""Python mirror of the reference's public API (module eigen_libs_mod, src/eigen_libs.F:14-218).

Names, argument order, defaults and error behaviour follow the Fortran interface:

    call eigen_init([comm],[order])                      src/eigen_libs.F:70-104
    call eigen_get_matdims(n, nx, ny[, m_f, m_b, mode])  src/eigen_libs.F:106-148
    call eigen_sx(n, nvec, a, lda, w, z, ldz[, m_forward, m_backward, mode])   src/eigen_sx.F:30-308
    call eigen_s (n, nvec, a, lda, w, z, ldz[, m_forward, m_backward, mode])   src/eigen_libs.F:150-202
    call eigen_free()                                    src/eigen_libs.F:204-216

``a``, ``w``, ``z`` are either numpy arrays (host, Fortran order, as in the reference) or torch CUDA
tensors laid out column-major (device-resident: no PCIe traffic).  Like the reference the solvers have
no status argument: precondition failures print a warning and return; NaN/Inf input sets ``w`` to NaN
(src/eigen_sx.F:82-131, :151-155).  The last status code is kept in ``last_status`` for tests.
"""
import ctypes as C
import sys

import numpy as np

from . import _lib

# defaults of the reference (src/eigen_libs0.F:49-51)
eigen_NB_f = 48
eigen_NB_b = 128

_state = {"initialized": False, "comm": None, "last_status": 0}


def last_status():
    return _state["last_status"]


def _char(c, default):
    if c is None:
        c = default
    if isinstance(c, bytes):
        return c[:1]
    return str(c)[:1].encode()


def eigen_init(comm=None, order="C", device=None, dims=None):
    """eigen_init(comm, order): ``comm`` is None (single GPU) or an initialised ``torch.distributed``
    process group / True for the default group (one process per GPU; RCCL communicators are built from
    a unique id broadcast over it).  ``order`` 'R' or 'C' as in the reference (src/eigen_libs.F:88-97).
    ``dims`` = (Px, Py): explicit process grid, the counterpart of passing a 2-D cartesian communicator
    (src/eigen_libs0.F:579-715)."""
    lib = _lib.load()
    if dims is not None:
        _lib.check(lib.eigx_set_grid_dims(int(dims[0]), int(dims[1])), "eigx_set_grid_dims")
    rank, nranks = 0, 1
    dist = None
    if comm is not None and comm is not False:
        import torch.distributed as dist_mod

        dist = dist_mod
        group = None if comm is True else comm
        rank = dist.get_rank(group)
        nranks = dist.get_world_size(group)
    if device is None:
        import os

        device = int(os.environ.get("LOCAL_RANK", "0")) if nranks > 1 else 0
    if nranks == 1:
        rc = lib.eigx_init(int(device))
    else:
        import torch

        # the 128-byte session id (an ncclUniqueId when RCCL is installed) is made on rank 0 and broadcast over the
        # caller's process group; everything else -- the shared-memory board, the hipIpc window exchange, the RCCL
        # world / X / Y communicators -- happens inside eigx_init_multi
        group = None if comm is True else comm
        be = dist.get_backend(group)
        uid = (C.c_char * 128)()
        if rank == 0:
            _lib.check(lib.eigx_get_rccl_unique_id(uid), "eigx_get_rccl_unique_id")
        t = torch.tensor(list(bytes(uid)), dtype=torch.uint8)
        if be == "nccl":
            t = t.cuda(int(device))
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        raw = bytes(t.cpu().tolist())
        buf = C.create_string_buffer(raw, 128)
        rc = lib.eigx_init_multi(int(device), rank, nranks, buf, _char(order, "C"))
    _lib.check(rc, "eigen_init")
    _state["initialized"] = True
    _state["comm"] = comm
    return None


def eigen_comm_info():
    """transports chosen at eigen_init (per-step exchange, its wait, bulk collectives) and the init-time self-test's
    counts, as a dict (eigx_comm_info); {"ranks": 1} on one GPU"""
    import json

    lib = _lib.load()
    buf = C.create_string_buffer(2048)
    _lib.check(lib.eigx_comm_info(buf, 2048), "eigx_comm_info")
    return json.loads(buf.value.decode())


def eigen_free():
    lib = _lib.load()
    lib.eigx_free()
    _state["initialized"] = False


def eigen_get_matdims(n, m_forward=None, m_backward=None, mode="O"):
    """returns (nx, ny): extents of the local arrays a(nx,ny), z(nx,ny); (-1,-1) if too large."""
    lib = _lib.load()
    nx, ny = C.c_int(-1), C.c_int(-1)
    lib.eigx_get_matdims(int(n), C.byref(nx), C.byref(ny), int(m_forward or eigen_NB_f),
                         int(m_backward or eigen_NB_b), _char(mode, "O"))
    return nx.value, ny.value


def eigen_get_procs():
    lib = _lib.load()
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    _lib.check(lib.eigx_get_procs(C.byref(a), C.byref(b), C.byref(c)), "eigen_get_procs")
    return a.value, b.value, c.value


def eigen_get_id():
    lib = _lib.load()
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    _lib.check(lib.eigx_get_id(C.byref(a), C.byref(b), C.byref(c)), "eigen_get_id")
    return a.value, b.value, c.value


def eigen_get_version():
    lib = _lib.load()
    v = C.c_int()
    d = C.create_string_buffer(32)
    vc = C.create_string_buffer(32)
    lib.eigx_get_version(C.byref(v), d, vc)
    return v.value, d.value.decode(), vc.value.decode()


def eigen_get_errinfo():
    lib = _lib.load()
    v = C.c_int64()
    lib.eigx_get_errinfo(C.byref(v))
    return v.value


def eigen_memory_internal(n, lda, ldz, m1=None, m0=None):
    lib = _lib.load()
    return lib.eigx_memory_internal(int(n), int(lda), int(ldz), int(m1 or eigen_NB_f), int(m0 or eigen_NB_b))


def _grid_dim(grid):
    procs, xp, yp = eigen_get_procs()
    idn, xi, yi = eigen_get_id()
    g = str(grid)[:1].upper()
    if g == "X":
        return xp, xi
    if g == "Y":
        return yp, yi
    return procs, idn


# index helpers: (value, 'X'|'Y') like the reference (src/eigen_libs0.F:1744-2356); 1-based
def eigen_loop_start(istart, grid):
    nnod, inod = _grid_dim(grid)
    return _lib.load().eigx_loop_start(int(istart), nnod, inod)


def eigen_loop_end(iend, grid):
    nnod, inod = _grid_dim(grid)
    return _lib.load().eigx_loop_end(int(iend), nnod, inod)


def eigen_translate_l2g(ictr, grid):
    nnod, inod = _grid_dim(grid)
    return _lib.load().eigx_translate_l2g(int(ictr), nnod, inod)


def eigen_translate_g2l(ictr, grid):
    nnod, inod = _grid_dim(grid)
    return _lib.load().eigx_translate_g2l(int(ictr), nnod, inod)


def eigen_owner_node(ictr, grid):
    nnod, inod = _grid_dim(grid)
    return _lib.load().eigx_owner_node(int(ictr), nnod, inod)


def eigen_owner_index(ictr, grid):
    nnod, inod = _grid_dim(grid)
    return _lib.load().eigx_owner_index(int(ictr), nnod, inod)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x, name, want_device):
    if x is None:
        return None
    if _is_torch(x):
        if not x.is_cuda:
            raise ValueError(f"{name}: torch tensors must live on the GPU (use numpy for host arrays)")
        import torch

        if x.dtype != torch.float64:
            raise ValueError(f"{name}: float64 required")
        if not want_device:
            raise ValueError("a, w, z must all be host arrays or all be device tensors")
        return x.data_ptr()
    if want_device:
        raise ValueError("a, w, z must all be host arrays or all be device tensors")
    if x.dtype != np.float64:
        raise ValueError(f"{name}: float64 required")
    if x.ndim == 2 and not x.flags.f_contiguous:
        raise ValueError(f"{name}: Fortran (column-major) order required, as in the reference")
    return x.ctypes.data


def _solve(which, n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode, nb=None):
    lib = _lib.load()
    if not _state["initialized"]:
        # reference: silent return when eigen_init has not been called (src/eigen_sx.F:82-86)
        _state["last_status"] = -1
        return
    dev = _is_torch(a)
    if dev:
        import torch

        torch.cuda.current_stream().synchronize()  # inputs written on torch's stream must be visible
    pa, pw, pz = _ptr(a, "a", dev), _ptr(w, "w", dev), _ptr(z, "z", dev)
    mf = eigen_NB_f if m_forward is None else int(m_forward)
    mb = eigen_NB_b if m_backward is None else int(m_backward)
    if nb is None:
        fn = getattr(lib, ("eigx_sx" if which == "sx" else "eigx_s") + ("_dev" if dev else ""))
        rc = fn(int(n), int(nvec), pa, int(lda), pw, pz, int(ldz), mf, mb, _char(mode, "A"))
    else:
        fn = lib.eigx_solve_bc_dev if dev else lib.eigx_solve_bc
        rc = fn(2 if which == "sx" else 1, int(n), int(nvec), pa, int(lda), pw, pz, int(ldz), int(nb), mf, mb,
                _char(mode, "A"))
    _state["last_status"] = rc
    if rc not in (0, -5):
        print(f"Warning: eigen_{which} returned without computing (status {rc})", file=sys.stderr)


def eigen_sx(n, nvec, a, lda, w, z, ldz, m_forward=None, m_backward=None, mode="A"):
    """Pentadiagonal route (eigen_prd -> eigen_dcx -> trbakwy, src/eigen_sx.F:30-308)."""
    _solve("sx", n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)


def eigen_s(n, nvec, a, lda, w, z, ldz, m_forward=None, m_backward=None, mode="A"):
    """Tridiagonal route (eigen_trd -> dc2 -> trbakwy, src/eigen_libs.F:150-202)."""
    _solve("s", n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)


def eigen_h(n, nvec, a, lda, w, z, ldz, m_forward=None, m_backward=None, mode="A"):
    """Complex Hermitian solver (src/eigen_h.F:30-322: eigen_hrd -> dc2 -> eigen_hrbakwyx).  ``a``, ``z``: complex128,
    column-major (numpy, Fortran order) or GPU tensors holding the column-major image (``a[j, i] = A(i, j)``); upper
    triangle of ``a`` significant, ``a`` destroyed; ``w`` float64 ascending.  modes 'A', 'N', 'X'.  One GPU."""
    lib = _lib.load()
    if not _state["initialized"]:
        _state["last_status"] = -1
        return
    dev = _is_torch(a)

    def cptr(x, name, real=False):
        if x is None:
            return None
        if dev:
            import torch

            if not (_is_torch(x) and x.is_cuda):
                raise ValueError("a, w, z must all be host arrays or all be device tensors")
            if x.dtype != (torch.float64 if real else torch.complex128):
                raise ValueError(f"{name}: {'float64' if real else 'complex128'} required")
            return x.data_ptr()
        if _is_torch(x):
            raise ValueError("a, w, z must all be host arrays or all be device tensors")
        if x.dtype != (np.float64 if real else np.complex128):
            raise ValueError(f"{name}: {'float64' if real else 'complex128'} required")
        if x.ndim == 2 and not x.flags.f_contiguous:
            raise ValueError(f"{name}: Fortran (column-major) order required, as in the reference")
        return x.ctypes.data

    if dev:
        import torch

        torch.cuda.current_stream().synchronize()
    mf = eigen_NB_f if m_forward is None else int(m_forward)
    mb = eigen_NB_b if m_backward is None else int(m_backward)
    fn = lib.eigx_h_dev if dev else lib.eigx_h
    rc = fn(int(n), int(nvec), cptr(a, "a"), int(lda), cptr(w, "w", real=True), cptr(z, "z"), int(ldz), mf, mb,
            _char(mode, "A"))
    _state["last_status"] = rc
    if rc not in (0, -5):
        print(f"Warning: eigen_h returned without computing (status {rc})", file=sys.stderr)


def eigen_sx_bc(n, nvec, a, lda, w, z, ldz, nb, m_forward=None, m_backward=None, mode="A"):
    """eigen_sx on the local blocks of a 2-D block-cyclic (ScaLAPACK, MB = NB = nb) distribution over the process grid:
    no pdgemr2d redistribution into the cyclic layout is needed (manual 3.4).  ``z`` returns in the same distribution."""
    _solve("sx", n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode, nb=nb)


def eigen_s_bc(n, nvec, a, lda, w, z, ldz, nb, m_forward=None, m_backward=None, mode="A"):
    """eigen_s on block-cyclic local blocks (see eigen_sx_bc)."""
    _solve("s", n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode, nb=nb)


def numroc(n, nb, iproc, nprocs):
    """ScaLAPACK NUMROC with source process 0: local extent of n indices in blocks of nb on process iproc of nprocs"""
    return _lib.load().eigx_numroc(int(n), int(nb), int(iproc), int(nprocs))


def KMATH_EIGEN_GEV(n, a, lda, b, ldb, w, z, ldz):
    """Generalised symmetric-definite problem A x = lambda B x (src/KMATH_EIGEN_GEV.F:1-64): two eigen_s solves and
    three GEMMs.  Upper triangles of ``a``, ``b`` significant; ``w`` ascending, ``z`` B-orthonormal; ``a`` and ``b``
    are destroyed.  If B is not positive definite a message is printed and the call returns (status -7)."""
    lib = _lib.load()
    if not _state["initialized"]:
        _state["last_status"] = -1
        return
    dev = _is_torch(a)
    if dev:
        import torch

        torch.cuda.current_stream().synchronize()
    pa, pb, pw, pz = _ptr(a, "a", dev), _ptr(b, "b", dev), _ptr(w, "w", dev), _ptr(z, "z", dev)
    fn = lib.eigx_gev_dev if dev else lib.eigx_gev
    rc = fn(int(n), pa, int(lda), pb, int(ldb), pw, pz, int(ldz))
    _state["last_status"] = rc
    if rc not in (0, -7):
        print(f"Warning: KMATH_EIGEN_GEV returned without computing (status {rc})", file=sys.stderr)
