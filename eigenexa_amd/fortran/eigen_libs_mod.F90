!> eigen_libs_mod -- thin ISO_C_BINDING module over the C-ABI of libeigenexa_amd.so
!! (include/eigenexa_amd.h).  It keeps the reference's Fortran API surface
!! (module eigen_libs_mod, src/eigen_libs.F:14-218) so that benchmark/main2.f-style callers
!! compile unchanged: same names, same argument order, same optional arguments and defaults
!! (m_forward = 48, m_backward = 128, mode = 'A', order = 'C'; src/eigen_libs0.F:49-51).
!!
!! Build:  flang -cpp -c eigen_libs_mod.F90            (single rank, no MPI)
!!         flang -cpp -DEIGX_WITH_MPI -I<mpi include> -c eigen_libs_mod.F90
!!                                                      (one MPI rank per GPU: the RCCL unique id
!!                                                       is broadcast with MPI_Bcast)
!! Link :  -L<repo>/eigenexa_amd/lib -leigenexa_amd
module eigen_libs_mod
  use, intrinsic :: iso_c_binding
  implicit none
  private

  integer, parameter, public :: eigen_NB_f = 48, eigen_NB_b = 128

  public :: eigen_init, eigen_free, eigen_get_matdims, eigen_get_procs, eigen_get_id
  public :: eigen_get_version, eigen_get_errinfo, eigen_memory_internal
  public :: eigen_loop_start, eigen_loop_end, eigen_translate_l2g, eigen_translate_g2l
  public :: eigen_owner_node, eigen_owner_index
  public :: eigen_sx, eigen_s
  public :: eigen_sx_bc, eigen_s_bc   ! ScaLAPACK block-cyclic local blocks in and out (no pdgemr2d step)
  public :: eigen_h                   ! complex Hermitian solver (src/eigen_h.F)

  interface
    integer(c_int) function eigx_init(device) bind(C, name="eigx_init")
      import :: c_int
      integer(c_int), value :: device
    end function
    integer(c_int) function eigx_init_multi(device, rank, nranks, uid, order) bind(C, name="eigx_init_multi")
      import :: c_int, c_char
      integer(c_int), value :: device, rank, nranks
      character(kind=c_char), intent(in) :: uid(128)
      character(kind=c_char), value :: order
    end function
    integer(c_int) function eigx_get_rccl_unique_id(uid) bind(C, name="eigx_get_rccl_unique_id")
      import :: c_int, c_char
      character(kind=c_char), intent(out) :: uid(128)
    end function
    integer(c_int) function eigx_free() bind(C, name="eigx_free")
      import :: c_int
    end function
    integer(c_int) function eigx_get_matdims(n, nx, ny, mf, mb, mode) bind(C, name="eigx_get_matdims")
      import :: c_int, c_char
      integer(c_int), value :: n, mf, mb
      integer(c_int), intent(out) :: nx, ny
      character(kind=c_char), value :: mode
    end function
    integer(c_int) function eigx_get_procs(p, xp, yp) bind(C, name="eigx_get_procs")
      import :: c_int
      integer(c_int), intent(out) :: p, xp, yp
    end function
    integer(c_int) function eigx_get_id(p, xp, yp) bind(C, name="eigx_get_id")
      import :: c_int
      integer(c_int), intent(out) :: p, xp, yp
    end function
    integer(c_int) function eigx_get_version(v, d, vc) bind(C, name="eigx_get_version")
      import :: c_int, c_char
      integer(c_int), intent(out) :: v
      character(kind=c_char), intent(out) :: d(32), vc(32)
    end function
    integer(c_int) function eigx_get_errinfo(info) bind(C, name="eigx_get_errinfo")
      import :: c_int, c_int64_t
      integer(c_int64_t), intent(out) :: info
    end function
    integer(c_int64_t) function eigx_memory_internal(n, lda, ldz, m1, m0) bind(C, name="eigx_memory_internal")
      import :: c_int, c_int64_t
      integer(c_int), value :: n, lda, ldz, m1, m0
    end function
    integer(c_int) function eigx_loop_start(i, nnod, inod) bind(C, name="eigx_loop_start")
      import :: c_int
      integer(c_int), value :: i, nnod, inod
    end function
    integer(c_int) function eigx_loop_end(i, nnod, inod) bind(C, name="eigx_loop_end")
      import :: c_int
      integer(c_int), value :: i, nnod, inod
    end function
    integer(c_int) function eigx_translate_l2g(i, nnod, inod) bind(C, name="eigx_translate_l2g")
      import :: c_int
      integer(c_int), value :: i, nnod, inod
    end function
    integer(c_int) function eigx_translate_g2l(i, nnod, inod) bind(C, name="eigx_translate_g2l")
      import :: c_int
      integer(c_int), value :: i, nnod, inod
    end function
    integer(c_int) function eigx_owner_node(i, nnod, inod) bind(C, name="eigx_owner_node")
      import :: c_int
      integer(c_int), value :: i, nnod, inod
    end function
    integer(c_int) function eigx_owner_index(i, nnod, inod) bind(C, name="eigx_owner_index")
      import :: c_int
      integer(c_int), value :: i, nnod, inod
    end function
    integer(c_int) function eigx_sx(n, nvec, a, lda, w, z, ldz, mf, mb, mode) bind(C, name="eigx_sx")
      import :: c_int, c_double, c_char
      integer(c_int), value :: n, nvec, lda, ldz, mf, mb
      real(c_double), intent(inout) :: a(lda, *)
      real(c_double), intent(out) :: w(*), z(ldz, *)
      character(kind=c_char), value :: mode
    end function
    integer(c_int) function eigx_s(n, nvec, a, lda, w, z, ldz, mf, mb, mode) bind(C, name="eigx_s")
      import :: c_int, c_double, c_char
      integer(c_int), value :: n, nvec, lda, ldz, mf, mb
      real(c_double), intent(inout) :: a(lda, *)
      real(c_double), intent(out) :: w(*), z(ldz, *)
      character(kind=c_char), value :: mode
    end function
    integer(c_int) function eigx_h(n, nvec, a, lda, w, z, ldz, mf, mb, mode) bind(C, name="eigx_h")
      import :: c_int, c_double, c_double_complex, c_char
      integer(c_int), value :: n, nvec, lda, ldz, mf, mb
      complex(c_double_complex) :: a(*), z(*)
      real(c_double) :: w(*)
      character(kind=c_char), value :: mode
    end function
    integer(c_int) function eigx_set_grid_dims(px, py) bind(C, name="eigx_set_grid_dims")
      import :: c_int
      integer(c_int), value :: px, py
    end function
    integer(c_int) function eigx_solve_bc(route, n, nvec, a, lda, w, z, ldz, nb, mf, mb, mode) &
        bind(C, name="eigx_solve_bc")
      import :: c_int, c_double, c_char
      integer(c_int), value :: route, n, nvec, lda, ldz, nb, mf, mb
      real(c_double) :: a(*), w(*), z(*)
      character(kind=c_char), value :: mode
    end function
  end interface

contains

  !> eigen_init(comm, order)   (src/eigen_libs.F:70-104)
  subroutine eigen_init(comm, order)
#ifdef EIGX_WITH_MPI
    use mpi
#endif
    integer, intent(in), optional :: comm
    character(*), intent(in), optional :: order
    character(kind=c_char) :: ord, uid(128)
    integer :: rc, rank, nranks, ierr, dev, lcomm, topo, ndims, dims(2), coords(2)
    logical :: periods(2)
    ord = 'C'
    if (present(order)) ord = order(1:1)
    rank = 0; nranks = 1
#ifdef EIGX_WITH_MPI
    lcomm = MPI_COMM_WORLD
    if (present(comm)) lcomm = comm
    if (lcomm == MPI_COMM_NULL) return        ! non-participant (src/eigen_libs0.F:405-415)
    call MPI_Comm_rank(lcomm, rank, ierr)
    call MPI_Comm_size(lcomm, nranks, ierr)
#endif
    dev = rank      ! one rank per GPU on one node; a launcher may remap with ROCR_VISIBLE_DEVICES
    if (nranks == 1) then
      rc = eigx_init(0)
    else
#ifdef EIGX_WITH_MPI
      ! a 2-D cartesian communicator fixes the process grid (eigen_init_cartesian_check, src/eigen_libs0.F:579-715);
      ! MPI numbers cartesian ranks row-major
      call MPI_Topo_test(lcomm, topo, ierr)
      if (topo == MPI_CART) then
        call MPI_Cartdim_get(lcomm, ndims, ierr)
        if (ndims == 2) then
          call MPI_Cart_get(lcomm, 2, dims, periods, coords, ierr)
          rc = eigx_set_grid_dims(dims(1), dims(2))
          ord = 'R'
        end if
      end if
      if (rank == 0) rc = eigx_get_rccl_unique_id(uid)
      call MPI_Bcast(uid, 128, MPI_CHARACTER, 0, lcomm, ierr)
      rc = eigx_init_multi(dev, rank, nranks, uid, ord)
#else
      rc = -1
#endif
    end if
    if (rc /= 0) print *, "eigen_init: libeigenexa_amd returned ", rc
  end subroutine eigen_init

  subroutine eigen_free(flag)
    integer, intent(in), optional :: flag
    integer :: rc
    rc = eigx_free()
  end subroutine eigen_free

  !> eigen_get_matdims(n, nx, ny, m_forward, m_backward, mode)   (src/eigen_libs.F:106-148)
  subroutine eigen_get_matdims(n, nx, ny, m_forward, m_backward, mode)
    integer, intent(in) :: n
    integer, intent(out) :: nx, ny
    integer, intent(in), optional :: m_forward, m_backward
    character(*), intent(in), optional :: mode
    integer :: mf, mb, rc
    character(kind=c_char) :: md
    mf = eigen_NB_f; mb = eigen_NB_b; md = 'O'
    if (present(m_forward)) mf = m_forward
    if (present(m_backward)) mb = m_backward
    if (present(mode)) md = mode(1:1)
    rc = eigx_get_matdims(n, nx, ny, mf, mb, md)
  end subroutine eigen_get_matdims

  subroutine eigen_get_procs(procs, x_procs, y_procs)
    integer, intent(out) :: procs, x_procs, y_procs
    integer :: rc
    rc = eigx_get_procs(procs, x_procs, y_procs)
  end subroutine

  subroutine eigen_get_id(id, x_id, y_id)
    integer, intent(out) :: id, x_id, y_id
    integer :: rc
    rc = eigx_get_id(id, x_id, y_id)
  end subroutine

  subroutine eigen_get_version(version, date, vcode)
    integer, intent(out) :: version
    character(*), intent(out), optional :: date, vcode
    character(kind=c_char) :: d(32), v(32)
    integer :: rc, i
    rc = eigx_get_version(version, d, v)
    if (present(date)) then
      date = ' '
      do i = 1, min(len(date), 32)
        if (d(i) == c_null_char) exit
        date(i:i) = d(i)
      end do
    end if
    if (present(vcode)) then
      vcode = ' '
      do i = 1, min(len(vcode), 32)
        if (v(i) == c_null_char) exit
        vcode(i:i) = v(i)
      end do
    end if
  end subroutine

  integer(8) function eigen_get_errinfo() result(info)
    integer :: rc
    integer(c_int64_t) :: v
    rc = eigx_get_errinfo(v)
    info = v
  end function

  integer(8) function eigen_memory_internal(n, lda, ldz, m1_opt, m0_opt) result(bytes)
    integer, intent(in) :: n, lda, ldz
    integer, intent(in), optional :: m1_opt, m0_opt
    integer :: m1, m0
    m1 = eigen_NB_f; m0 = eigen_NB_b
    if (present(m1_opt)) m1 = m1_opt
    if (present(m0_opt)) m0 = m0_opt
    bytes = eigx_memory_internal(n, lda, ldz, m1, m0)
  end function

  subroutine grid_of(grid, nnod, inod)
    character(*), intent(in) :: grid
    integer, intent(out) :: nnod, inod
    integer :: p, xp, yp, id, xi, yi, rc
    rc = eigx_get_procs(p, xp, yp)
    rc = eigx_get_id(id, xi, yi)
    select case (grid(1:1))
    case ('X', 'x'); nnod = xp; inod = xi
    case ('Y', 'y'); nnod = yp; inod = yi
    case default;    nnod = p;  inod = id
    end select
  end subroutine

  !> index helpers (src/eigen_libs0.F:1744-2356); 1-based, grid = 'X' or 'Y'
  integer function eigen_loop_start(istart, grid) result(r)
    integer, intent(in) :: istart
    character(*), intent(in) :: grid
    integer :: nnod, inod
    call grid_of(grid, nnod, inod); r = eigx_loop_start(istart, nnod, inod)
  end function
  integer function eigen_loop_end(iend, grid) result(r)
    integer, intent(in) :: iend
    character(*), intent(in) :: grid
    integer :: nnod, inod
    call grid_of(grid, nnod, inod); r = eigx_loop_end(iend, nnod, inod)
  end function
  integer function eigen_translate_l2g(ictr, grid) result(r)
    integer, intent(in) :: ictr
    character(*), intent(in) :: grid
    integer :: nnod, inod
    call grid_of(grid, nnod, inod); r = eigx_translate_l2g(ictr, nnod, inod)
  end function
  integer function eigen_translate_g2l(ictr, grid) result(r)
    integer, intent(in) :: ictr
    character(*), intent(in) :: grid
    integer :: nnod, inod
    call grid_of(grid, nnod, inod); r = eigx_translate_g2l(ictr, nnod, inod)
  end function
  integer function eigen_owner_node(ictr, grid) result(r)
    integer, intent(in) :: ictr
    character(*), intent(in) :: grid
    integer :: nnod, inod
    call grid_of(grid, nnod, inod); r = eigx_owner_node(ictr, nnod, inod)
  end function
  integer function eigen_owner_index(ictr, grid) result(r)
    integer, intent(in) :: ictr
    character(*), intent(in) :: grid
    integer :: nnod, inod
    call grid_of(grid, nnod, inod); r = eigx_owner_index(ictr, nnod, inod)
  end function

  !> eigen_sx(n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)   (src/eigen_sx.F:30-308)
  subroutine eigen_sx(n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)
    integer, intent(in) :: n, nvec, lda, ldz
    real(8), intent(inout) :: a(lda, *)
    real(8), intent(out) :: w(*), z(ldz, *)
    integer, intent(in), optional :: m_forward, m_backward
    character(*), intent(in), optional :: mode
    integer :: mf, mb, rc
    character(kind=c_char) :: md
    mf = eigen_NB_f; mb = eigen_NB_b; md = 'A'
    if (present(m_forward)) mf = m_forward
    if (present(m_backward)) mb = m_backward
    if (present(mode)) md = mode(1:1)
    rc = eigx_sx(n, nvec, a, lda, w, z, ldz, mf, mb, md)   ! no status argument in the reference
  end subroutine eigen_sx

  !> eigen_s(n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)   (src/eigen_libs.F:150-202)
  subroutine eigen_s(n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)
    integer, intent(in) :: n, nvec, lda, ldz
    real(8), intent(inout) :: a(lda, *)
    real(8), intent(out) :: w(*), z(ldz, *)
    integer, intent(in), optional :: m_forward, m_backward
    character(*), intent(in), optional :: mode
    integer :: mf, mb, rc
    character(kind=c_char) :: md
    mf = eigen_NB_f; mb = eigen_NB_b; md = 'A'
    if (present(m_forward)) mf = m_forward
    if (present(m_backward)) mb = m_backward
    if (present(mode)) md = mode(1:1)
    rc = eigx_s(n, nvec, a, lda, w, z, ldz, mf, mb, md)
  end subroutine eigen_s

  !> eigen_sx on the local blocks of a ScaLAPACK descriptor with MB = NB = nb, RSRC = CSRC = 0 on the eigen process
  !> grid: a is numroc(n,nb,x_id-1,0,x_procs) x numroc(n,nb,y_id-1,0,y_procs); z returns in the same distribution.
  !> Replaces the pdgemr2d round trip of the reference manual 3.4 (the layout is an index map at the solver's entry).
  subroutine eigen_sx_bc(n, nvec, a, lda, w, z, ldz, nb, m_forward, m_backward, mode)
    integer, intent(in) :: n, nvec, lda, ldz, nb
    real(8), intent(inout) :: a(lda, *)
    real(8), intent(out) :: w(*), z(ldz, *)
    integer, intent(in), optional :: m_forward, m_backward
    character(*), intent(in), optional :: mode
    integer :: mf, mb, rc
    character(kind=c_char) :: md
    mf = eigen_NB_f; mb = eigen_NB_b; md = 'A'
    if (present(m_forward)) mf = m_forward
    if (present(m_backward)) mb = m_backward
    if (present(mode)) md = mode(1:1)
    rc = eigx_solve_bc(2, n, nvec, a, lda, w, z, ldz, nb, mf, mb, md)
  end subroutine eigen_sx_bc

  !> eigen_s on block-cyclic local blocks (see eigen_sx_bc)
  subroutine eigen_s_bc(n, nvec, a, lda, w, z, ldz, nb, m_forward, m_backward, mode)
    integer, intent(in) :: n, nvec, lda, ldz, nb
    real(8), intent(inout) :: a(lda, *)
    real(8), intent(out) :: w(*), z(ldz, *)
    integer, intent(in), optional :: m_forward, m_backward
    character(*), intent(in), optional :: mode
    integer :: mf, mb, rc
    character(kind=c_char) :: md
    mf = eigen_NB_f; mb = eigen_NB_b; md = 'A'
    if (present(m_forward)) mf = m_forward
    if (present(m_backward)) mb = m_backward
    if (present(mode)) md = mode(1:1)
    rc = eigx_solve_bc(1, n, nvec, a, lda, w, z, ldz, nb, mf, mb, md)
  end subroutine eigen_s_bc

  !> eigen_h(n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)   (src/eigen_h.F:30-322): complex Hermitian
  !> matrix, upper triangle of a significant; w real ascending; z unitary.  complex(8) arrays are passed as they are
  !> (interleaved re/im = the C-ABI's layout).  One GPU in this version.
  subroutine eigen_h(n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)
    integer, intent(in) :: n, nvec, lda, ldz
    complex(8), intent(inout) :: a(lda, *)
    real(8), intent(out) :: w(*)
    complex(8), intent(out) :: z(ldz, *)
    integer, intent(in), optional :: m_forward, m_backward
    character(*), intent(in), optional :: mode
    integer :: mf, mb, rc
    character(kind=c_char) :: md
    mf = eigen_NB_f; mb = eigen_NB_b; md = 'A'
    if (present(m_forward)) mf = m_forward
    if (present(m_backward)) mb = m_backward
    if (present(mode)) md = mode(1:1)
    rc = eigx_h(n, nvec, a, lda, w, z, ldz, mf, mb, md)
  end subroutine eigen_h

end module eigen_libs_mod


! KMATH_EIGEN_GEV is an external subroutine in the reference (src/KMATH_EIGEN_GEV.F:1-64, not a module procedure):
! generalised symmetric-definite problem A x = lambda B x.  Same argument list; one GPU.
subroutine KMATH_EIGEN_GEV(n, a, lda, b, ldb, w, z, ldz)
  use, intrinsic :: iso_c_binding
  implicit none
  integer, intent(inout) :: n, lda, ldb, ldz
  real(8), intent(inout) :: a(lda, *), b(ldb, *)
  real(8), intent(inout) :: w(*), z(ldz, *)
  interface
    integer(c_int) function eigx_gev(n, a, lda, b, ldb, w, z, ldz) bind(C, name="eigx_gev")
      import :: c_int, c_double
      integer(c_int), value :: n, lda, ldb, ldz
      real(c_double), intent(inout) :: a(lda, *), b(ldb, *), w(*), z(ldz, *)
    end function
  end interface
  integer(c_int) :: rc
  rc = eigx_gev(int(n, c_int), a, int(lda, c_int), b, int(ldb, c_int), w, z, int(ldz, c_int))
end subroutine KMATH_EIGEN_GEV
