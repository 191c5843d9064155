!> eigen_libs_mod -- thin ISO_C_BINDING module over the C-ABI of libeigenexa_amd.so
!! (include/eigenexa_amd.h).  It keeps the reference's Fortran API surface
!! (module eigen_libs_mod, src/eigen_libs.F:14-218) so that benchmark/main2.f-style callers
!! compile unchanged: same names, same argument order, same optional arguments and defaults
!! (m_forward = 48, m_backward = 128, mode = 'A', order = 'C'; src/eigen_libs0.F:49-51).
!!
!! Build:  flang -cpp -c eigen_libs_mod.F90            (single rank, no MPI)
!!         flang -cpp -DEIGX_WITH_MPI -I<mpi include> -c eigen_libs_mod.F90
!!                                                      (one MPI rank per GPU: the 128-byte session id
!!                                                       is broadcast with MPI_Bcast; `include 'mpif.h'`,
!!                                                       so any MPI's Fortran-77 binding serves, whatever
!!                                                       compiler built its mpi.mod)
!!         add -DEIGX_WITH_BLACS for module eigen_blacs_mod (eigen_get_blacs_context, src/eigen_blacs.F:167-174;
!!                                                       the BLACS is the caller's, as in the reference)
!! Link :  -L<repo>/eigenexa_amd/lib -leigenexa_amd
module eigen_libs_mod
  use, intrinsic :: iso_c_binding
  implicit none
  private

  ! src/eigen_libs0.F:49-51
  integer, parameter, public :: eigen_NB = 64, eigen_NB_f = 48, eigen_NB_b = 128

  public :: eigen_init, eigen_free, eigen_get_matdims, eigen_get_procs, eigen_get_id
  public :: eigen_get_version, eigen_show_version, eigen_initialized, eigen_get_comm
  public :: eigen_get_errinfo, eigen_memory_internal
  public :: eigen_loop_start, eigen_loop_end, eigen_loop_info, eigen_translate_l2g, eigen_translate_g2l
  public :: eigen_owner_node, eigen_owner_index, eigen_convert_ID_xy2w, eigen_convert_ID_w2xy
  public :: eigen_diag_loop_info
  public :: get_constant_eps, get_constant_nan, get_constant_pai, get_constant_2pai, get_constant_pai_2
  public :: eigen_sx, eigen_s, eigen_s0
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
    integer(c_int) function eigx_get_timers(out16) bind(C, name="eigx_get_timers")
      import :: c_int, c_double
      real(c_double), intent(out) :: out16(16)
    end function eigx_get_timers
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
    integer(c_int) function eigx_get_device_count() bind(C, name="eigx_get_device_count")
      import :: c_int
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

  ! the reference's index helpers are generic: (index, 'X'|'Y' [, inod]) and (index, nnod, inod)
  ! (src/eigen_libs0.F:1816-2258)
  interface eigen_loop_start
    module procedure loop_start_xy, loop_start_nn
  end interface
  interface eigen_loop_end
    module procedure loop_end_xy, loop_end_nn
  end interface
  interface eigen_loop_info
    module procedure loop_info_nn, loop_info_xy
  end interface
  interface eigen_translate_l2g
    module procedure l2g_xy, l2g_nn
  end interface
  interface eigen_translate_g2l
    module procedure g2l_xy, g2l_nn
  end interface
  interface eigen_owner_node
    module procedure owner_node_xy, owner_node_nn
  end interface
  interface eigen_owner_index
    module procedure owner_index_xy, owner_index_nn
  end interface

  logical, save :: initialized_ = .false.
  character, save :: grid_major_ = 'C'
  ! communicators handed back by eigen_get_comm (MPI build; src/eigen_libs0.F:1655-1669)
  integer, save :: comm_world_ = 0, comm_x_ = 0, comm_y_ = 0

contains

  !> eigen_init(comm, order)   (src/eigen_libs.F:70-104)
  subroutine eigen_init(comm, order)
#ifdef EIGX_WITH_MPI
    include 'mpif.h'
#endif
    integer, intent(in), optional :: comm
    character(*), intent(in), optional :: order
    character(kind=c_char) :: ord, uid(128)
    integer :: rc, rank, nranks, ierr, dev, lcomm, topo, ndims, dims(2), coords(2), ndev, lrank, ncomm, nlocal
    integer :: p, xp, yp, id, xi, yi
    logical :: periods(2)
    ord = 'C'
    if (present(order)) ord = order(1:1)
    if (ord == 'r') ord = 'R'
    rank = 0; nranks = 1
#ifdef EIGX_WITH_MPI
    lcomm = MPI_COMM_WORLD
    if (present(comm)) lcomm = comm
    if (lcomm == MPI_COMM_NULL) return        ! non-participant (src/eigen_libs0.F:405-415)
    call MPI_Comm_rank(lcomm, rank, ierr)
    call MPI_Comm_size(lcomm, nranks, ierr)
#endif
    if (nranks == 1) then
      rc = eigx_init(0)
    else
#ifdef EIGX_WITH_MPI
      ! GPU of this rank: its rank among the ranks of the node (a sub-communicator or a multi-node world does not
      ! number the GPUs), modulo the visible devices (ranks may share a card); ROCR_VISIBLE_DEVICES remaps further
      call MPI_Comm_split_type(lcomm, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, ncomm, ierr)
      call MPI_Comm_rank(ncomm, lrank, ierr)
      call MPI_Comm_size(ncomm, nlocal, ierr)
      call MPI_Comm_free(ncomm, ierr)
      ! The library serves the ranks of ONE node (at most 8, one xGMI domain: its ranks meet on a shared-memory board and
      ! map each other's windows).  A communicator that spans nodes -- which the reference supports -- is refused at
      ! once with the reference's fatal-error policy (MPI_Abort, src/eigen_libs0.F:392-404), not after a bootstrap time-out.
      if (nlocal /= nranks .or. nranks > 8) then
        if (rank == 0) print *, "eigen_init: this build serves up to 8 ranks of one node; the communicator has ", nranks, &
                                " ranks, ", nlocal, " of them on this node"
        call MPI_Abort(lcomm, 1, ierr)
      end if
      ndev = eigx_get_device_count()
      dev = 0
      if (ndev > 0) dev = mod(lrank, ndev)
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
    if (rc /= 0) then
      print *, "eigen_init: libeigenexa_amd returned ", rc
      return
    end if
    initialized_ = .true.
    grid_major_ = ord
#ifdef EIGX_WITH_MPI
    ! row / column communicators for the CALLER (the library itself talks over peer windows and RCCL):
    ! x = ranks sharing my y_id, ordered by x_id; y = ranks sharing my x_id (src/eigen_libs0.F:579-585)
    rc = eigx_get_procs(p, xp, yp)
    rc = eigx_get_id(id, xi, yi)
    call MPI_Comm_dup(lcomm, comm_world_, ierr)
    call MPI_Comm_split(comm_world_, yi - 1, xi - 1, comm_x_, ierr)
    call MPI_Comm_split(comm_world_, xi - 1, yi - 1, comm_y_, ierr)
#endif
  end subroutine eigen_init

  subroutine eigen_free(flag)
#ifdef EIGX_WITH_MPI
    include 'mpif.h'
    integer :: ierr
#endif
    integer, intent(in), optional :: flag
    integer :: rc
    rc = eigx_free()
#ifdef EIGX_WITH_MPI
    if (initialized_ .and. comm_world_ /= 0) then
      call MPI_Comm_free(comm_x_, ierr)
      call MPI_Comm_free(comm_y_, ierr)
      call MPI_Comm_free(comm_world_, ierr)
      comm_world_ = 0; comm_x_ = 0; comm_y_ = 0
    end if
#endif
    initialized_ = .false.
  end subroutine eigen_free

  !> eigen_initialized(flag)   (src/eigen_libs0.F:256-265)
  subroutine eigen_initialized(flag)
    logical, intent(out) :: flag
    flag = initialized_
  end subroutine

  !> eigen_get_comm(comm, x_comm, y_comm)   (src/eigen_libs0.F:1655-1669): duplicates of the communicator given to
  !> eigen_init and its row / column splits (MPI build); zeros in the single-rank build
  subroutine eigen_get_comm(comm, x_comm, y_comm)
    integer, intent(out) :: comm, x_comm, y_comm
    comm = comm_world_; x_comm = comm_x_; y_comm = comm_y_
  end subroutine

  !> eigen_show_version()   (src/eigen_libs0.F:207-236)
  subroutine eigen_show_version()
    integer :: version, id, xi, yi, rc
    character(32) :: date, vcode
    call eigen_get_version(version, date, vcode)
    rc = eigx_get_id(id, xi, yi)
    if (rc /= 0 .or. id == 1) then
      print '(A,I0,A,I0,A,A,A,A,A)', " ## EigenExa-AMD version (", version / 100, ".", mod(version, 100), &
            ") / (", trim(date), ") / (", trim(vcode), ")"
    end if
  end subroutine

  !> machine constants by bit pattern, as the reference returns them (src/eigen_libs0.F:2446-2540)
  real(8) function get_constant_eps() result(r)
    r = transfer(int(z'3CB0000000000000', 8), 1.0d0)
  end function
  real(8) function get_constant_nan() result(r)
    r = transfer(int(z'7FFFFFFFFFFFFFFF', 8), 1.0d0)
  end function
  real(8) function get_constant_pai() result(r)
    r = transfer(int(z'400921FB54442D18', 8), 1.0d0)
  end function
  real(8) function get_constant_2pai() result(r)
    r = transfer(int(z'401921FB54442D18', 8), 1.0d0)
  end function
  real(8) function get_constant_pai_2() result(r)
    r = transfer(int(z'3FF921FB54442D18', 8), 1.0d0)
  end function

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

  subroutine grid_of(grid, nnod, inod, inod_opt)
    character(*), intent(in) :: grid
    integer, intent(out) :: nnod, inod
    integer, intent(in), optional :: inod_opt
    integer :: p, xp, yp, id, xi, yi, rc
    rc = eigx_get_procs(p, xp, yp)
    rc = eigx_get_id(id, xi, yi)
    select case (grid(1:1))
    case ('X', 'x'); nnod = xp; inod = xi
    case ('Y', 'y'); nnod = yp; inod = yi
    case default;    nnod = p;  inod = id
    end select
    if (present(inod_opt)) inod = inod_opt
  end subroutine

  !> index helpers (src/eigen_libs0.F:1744-2356); 1-based; grid = 'X' or 'Y' (optionally for another process id),
  !> or explicit (nnod, inod)
  integer function loop_start_xy(istart, grid, inod) result(r)
    integer, intent(in) :: istart
    character(*), intent(in) :: grid
    integer, intent(in), optional :: inod
    integer :: nn, id
    call grid_of(grid, nn, id, inod); r = eigx_loop_start(istart, nn, id)
  end function
  integer function loop_start_nn(istart, nnod, inod) result(r)
    integer, intent(in) :: istart, nnod, inod
    r = eigx_loop_start(istart, nnod, inod)
  end function
  integer function loop_end_xy(iend, grid, inod) result(r)
    integer, intent(in) :: iend
    character(*), intent(in) :: grid
    integer, intent(in), optional :: inod
    integer :: nn, id
    call grid_of(grid, nn, id, inod); r = eigx_loop_end(iend, nn, id)
  end function
  integer function loop_end_nn(iend, nnod, inod) result(r)
    integer, intent(in) :: iend, nnod, inod
    r = eigx_loop_end(iend, nnod, inod)
  end function
  subroutine loop_info_nn(istart, iend, lstart, lend, nnod, inod)
    integer, intent(in) :: istart, iend, nnod, inod
    integer, intent(out) :: lstart, lend
    lstart = eigx_loop_start(istart, nnod, inod); lend = eigx_loop_end(iend, nnod, inod)
  end subroutine
  subroutine loop_info_xy(istart, iend, lstart, lend, pdir, inod)
    integer, intent(in) :: istart, iend
    integer, intent(out) :: lstart, lend
    character(*), intent(in) :: pdir
    integer, intent(in), optional :: inod
    lstart = loop_start_xy(istart, pdir, inod); lend = loop_end_xy(iend, pdir, inod)
  end subroutine
  integer function l2g_xy(ictr, grid, inod) result(r)
    integer, intent(in) :: ictr
    character(*), intent(in) :: grid
    integer, intent(in), optional :: inod
    integer :: nn, id
    call grid_of(grid, nn, id, inod); r = eigx_translate_l2g(ictr, nn, id)
  end function
  integer function l2g_nn(ictr, nnod, inod) result(r)
    integer, intent(in) :: ictr, nnod, inod
    r = eigx_translate_l2g(ictr, nnod, inod)
  end function
  integer function g2l_xy(ictr, grid, inod) result(r)
    integer, intent(in) :: ictr
    character(*), intent(in) :: grid
    integer, intent(in), optional :: inod
    integer :: nn, id
    call grid_of(grid, nn, id, inod); r = eigx_translate_g2l(ictr, nn, id)
  end function
  integer function g2l_nn(ictr, nnod, inod) result(r)
    integer, intent(in) :: ictr, nnod, inod
    r = eigx_translate_g2l(ictr, nnod, inod)
  end function
  integer function owner_node_xy(ictr, grid, inod) result(r)
    integer, intent(in) :: ictr
    character(*), intent(in) :: grid
    integer, intent(in), optional :: inod
    integer :: nn, id
    call grid_of(grid, nn, id, inod); r = eigx_owner_node(ictr, nn, id)
  end function
  integer function owner_node_nn(ictr, nnod, inod) result(r)
    integer, intent(in) :: ictr, nnod, inod
    r = eigx_owner_node(ictr, nnod, inod)
  end function
  integer function owner_index_xy(ictr, grid, inod) result(r)
    integer, intent(in) :: ictr
    character(*), intent(in) :: grid
    integer, intent(in), optional :: inod
    integer :: nn, id
    call grid_of(grid, nn, id, inod); r = eigx_owner_index(ictr, nn, id)
  end function
  integer function owner_index_nn(ictr, nnod, inod) result(r)
    integer, intent(in) :: ictr, nnod, inod
    r = eigx_owner_index(ictr, nnod, inod)
  end function

  !> world id <-> (x_id, y_id), all 1-based, in the grid order given to eigen_init (src/eigen_libs0.F:2316-2356;
  !> xy2w is written as the exact inverse of w2xy)
  integer function eigen_convert_ID_xy2w(xinod, yinod) result(ret)
    integer, intent(in) :: xinod, yinod
    integer :: p, xp, yp, rc
    rc = eigx_get_procs(p, xp, yp)
    if (grid_major_ == 'R') then
      ret = (xinod - 1) * yp + yinod
    else
      ret = (yinod - 1) * xp + xinod
    end if
  end function
  subroutine eigen_convert_ID_w2xy(inod, xinod, yinod)
    integer, intent(in) :: inod
    integer, intent(out) :: xinod, yinod
    integer :: p, xp, yp, rc
    rc = eigx_get_procs(p, xp, yp)
    if (grid_major_ == 'R') then
      xinod = (inod - 1) / yp + 1
      yinod = mod(inod - 1, yp) + 1
    else
      xinod = mod(inod - 1, xp) + 1
      yinod = (inod - 1) / xp + 1
    end if
  end subroutine

  !> eigen_diag_loop_info (src/eigen_libs0.F:2596-2636): local loop over the diagonal elements this process owns in
  !> [lstart, lend]: local row istart + k*istep, local column jstart + k*jstep, k = kstart .. kend.  n_common =
  !> gcd(x_procs, y_procs); (diag_0, diag_1) = first local (row, column) pair on the diagonal (src/eigen_libs0.F:586-683)
  subroutine eigen_diag_loop_info(lstart, lend, kstart, kend, istart, istep, jstart, jstep)
    integer, intent(in) :: lstart, lend
    integer, intent(out) :: kstart, kend, istart, istep, jstart, jstep
    integer :: p, xp, yp, id, xi, yi, rc, nc, n1, n2, n3, i, j, k, diag_0, diag_1, iend, jend
    rc = eigx_get_procs(p, xp, yp)
    rc = eigx_get_id(id, xi, yi)
    n1 = max(xp, yp); n2 = min(xp, yp)
    do while (n1 /= n2)
      n3 = n1 - n2; n1 = max(n2, n3); n2 = min(n2, n3)
    end do
    nc = n1
    diag_0 = 0; diag_1 = 0
    if (xp /= yp) then
      do i = 1, yp / nc
        j = (i - 1) * yp + yi
        k = mod(j - 1, xp) + 1
        if (k == xi) then
          diag_0 = i; diag_1 = (j - 1) / xp + 1
          exit
        end if
      end do
    else if (yi == xi) then
      diag_0 = 1; diag_1 = 1
    end if
    istart = 0; istep = 1; jstart = 0; jstep = 1; kstart = 0; kend = -1
    if (diag_0 <= 0 .or. lstart > lend) return
    istart = max(diag_0, loop_start_xy(lstart, 'Y'))
    iend = loop_end_xy(lend, 'Y')
    jstart = max(diag_1, loop_start_xy(lstart, 'X'))
    jend = loop_end_xy(lend, 'X')
    if (istart > iend .or. jstart > jend) then
      istart = 0; jstart = 0
      return
    end if
    istep = xp / nc
    jstep = yp / nc
    kend = (iend - istart) / istep
  end subroutine

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
    call timer_print_lines(n, nvec, md, 'TRD-BLK ')
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
    call timer_print_lines(n, nvec, md, 'TRD-BLK ')
  end subroutine eigen_s

  !> The reference's per-stage report of a TIMER_PRINT=1 build (src/eigen_sx.F:167-174, :225-232, :252-258, format 10000 at
  !> :304; src/eigen_FS.F likewise): one line per stage on rank 1 -- name, n, seconds, flops of the reference's model, GFLOPS.
  !> Here a run-time switch: EIGX_TIMER_PRINT=1 in the environment.
  subroutine timer_print_lines(n, nvec, md, first)
    integer, intent(in) :: n, nvec
    character(kind=c_char), intent(in) :: md
    character(len=8), intent(in) :: first
    character(len=8) :: val
    integer :: st, rc, procs, xp, yp, id, xi, yi
    real(c_double) :: tm(16), r1, r2, r3
    call get_environment_variable('EIGX_TIMER_PRINT', val, status=st)
    if (st /= 0) return
    if (len_trim(val) == 0 .or. val(1:1) == '0') return
    call eigen_get_id(id, xi, yi)
    if (id /= 1) return
    rc = eigx_get_timers(tm)
    r1 = dble(n)**3 * 4 / 3
    r2 = tm(12)
    r3 = 2 * dble(abs(nvec)) * dble(n)**2
    if (tm(2) > 0d0) print 10000, first, n, tm(2), r1, 1d-9 * r1 / tm(2), "GFLOPS"
    if (tm(3) > 0d0) print 10000, "D&C     ", n, tm(3), r2, 1d-9 * r2 / tm(3), "GFLOPS"
    if (tm(4) > 0d0 .and. md /= 'N' .and. md /= 'n') print 10000, "TRDBAK  ", n, tm(4), r3, 1d-9 * r3 / tm(4), "GFLOPS"
    flush(6)
10000 format (X, A8, I8, 3E25.16e2, X, A)
  end subroutine timer_print_lines

  !> eigen_s0: the reference's classic tridiagonal driver (src/eigen_s.F:30-307); eigen_s dispatches to it or to
  !> eigen_FS by process count -- one implementation serves both here
  subroutine eigen_s0(n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)
    integer, intent(in) :: n, nvec, lda, ldz
    real(8), intent(inout) :: a(lda, *)
    real(8), intent(out) :: w(*), z(ldz, *)
    integer, intent(in), optional :: m_forward, m_backward
    character(*), intent(in), optional :: mode
    call eigen_s(n, nvec, a, lda, w, z, ldz, m_forward, m_backward, mode)
  end subroutine eigen_s0

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

#if defined(EIGX_WITH_BLACS) && defined(EIGX_WITH_MPI)
!> eigen_blacs_mod (src/eigen_blacs.F:14-176): a BLACS context with the eigen process grid, for callers that check
!> or post-process with ScaLAPACK (benchmark/ev_test.f:67, benchmark/mat_set.f:160).  The BLACS is the caller's.
module eigen_blacs_mod
  use eigen_libs_mod
  implicit none
  private
  public :: eigen_get_blacs_context, eigen_blacs_exit
  integer, save :: ctxt_ = -1
  logical, save :: have_ctxt_ = .false.
contains
  integer function eigen_get_blacs_context() result(ctxt)
    include 'mpif.h'
    integer :: comm, xc, yc, nnod, xp, yp, i, j, k, ierr, g0, g1
    integer, allocatable :: tmpgrid(:, :), kk0(:), kk1(:)
    external :: BLACS_GET, BLACS_GRIDMAP
    if (.not. have_ctxt_) then
      call eigen_get_comm(comm, xc, yc)
      call eigen_get_procs(nnod, xp, yp)
      call BLACS_GET(0, 0, ctxt_)
      allocate(tmpgrid(xp, yp), kk0(xp), kk1(xp))
      call MPI_Comm_group(MPI_COMM_WORLD, g0, ierr)
      call MPI_Comm_group(comm, g1, ierr)
      do j = 1, yp
        do i = 1, xp
          kk1(i) = eigen_convert_ID_xy2w(i, j) - 1
        end do
        k = xp
        call MPI_Group_translate_ranks(g1, k, kk1, g0, kk0, ierr)
        tmpgrid(:, j) = kk0(:)
      end do
      call BLACS_GRIDMAP(ctxt_, tmpgrid, xp, xp, yp)
      call MPI_Group_free(g0, ierr)
      call MPI_Group_free(g1, ierr)
      deallocate(tmpgrid, kk0, kk1)
      have_ctxt_ = .true.
    end if
    ctxt = ctxt_
  end function
  subroutine eigen_blacs_exit()
    external :: BLACS_GRIDEXIT
    if (have_ctxt_) call BLACS_GRIDEXIT(ctxt_)
    have_ctxt_ = .false.
  end subroutine
end module eigen_blacs_mod
#endif


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
