!> A caller written against the REFERENCE's interface: it uses exactly the eigen_libs_mod symbol set of the
!! reference's own benchmark sources -- benchmark/main2.f:173-216,:218,:240-246,:377-420,:438,:558 (eigen_init with
!! order / communicator, eigen_get_version(version, date=), eigen_get_procs / eigen_get_id, get_constant_pai / _eps,
!! eigen_get_matdims(mode='O'), eigen_memory_internal, eigen_sx / eigen_s with keyword arguments, eigen_show_version,
!! eigen_free) and benchmark/mat_set.f:103-130,:296-320 (eigen_get_comm, eigen_loop_start / _end, eigen_translate_l2g /
!! _g2l, eigen_owner_node, eigen_NB) -- minus BLACS / PBLAS, which stay the caller's.  Frank matrix, both routes,
!! analytic spectrum check (benchmark/mat_set.f:638-647).  Built by tests against the module with -DEIGX_WITH_MPI.
program ref_caller
  use eigen_libs_mod
  implicit none
  include 'mpif.h'
  integer :: n, nm, ny, ierr, nnod, x_nnod, y_nnod, inod, x_inod, y_inod, version
  integer :: COMM, x_COMM, y_COMM, i, j, i_1, j_1, iloop_sta, iloop_end, jloop_sta, jloop_end, m, mb, nvec, msolver
  integer :: n1, n2, xs, ys
  integer(8) :: imem
  logical :: flag
  character(32) :: date
  character(1) :: mode
  real(8) :: PAI, EPS, lam, err, errmax, flops
  real(8), allocatable :: a(:, :), z(:, :), w(:)

  call MPI_Init(ierr)
  call eigen_init(order='c')
  call eigen_initialized(flag)
  call eigen_get_version(version, date=date)
  call eigen_get_procs(nnod, x_nnod, y_nnod)
  call eigen_get_id(inod, x_inod, y_inod)
  call eigen_get_comm(COMM, x_COMM, y_COMM)
  call MPI_Comm_size(x_COMM, xs, ierr)
  call MPI_Comm_size(y_COMM, ys, ierr)
  if (.not. flag .or. xs /= x_nnod .or. ys /= y_nnod) then
    print *, "FAILED: communicators", flag, xs, x_nnod, ys, y_nnod
    call MPI_Abort(MPI_COMM_WORLD, 1, ierr)
  end if
  PAI = get_constant_pai()
  EPS = get_constant_eps()
  n = 600; m = 48; mb = 128; nvec = n; mode = 'A'
  if (eigen_NB /= 64) stop 2
  errmax = 0d0
  do msolver = 0, 1
    call eigen_get_matdims(n, nm, ny, mode='O')
    imem = eigen_memory_internal(n, nm, nm, m, 128)
    if (nm <= 0 .or. ny <= 0 .or. imem <= 0) stop 3
    allocate(a(nm, ny), z(nm, ny), w(n))
    a = 0d0
    jloop_sta = eigen_loop_start(1, 'X')
    jloop_end = eigen_loop_end  (n, 'X')
    iloop_sta = eigen_loop_start(1, 'Y')
    iloop_end = eigen_loop_end  (n, 'Y')
    do i_1 = iloop_sta, iloop_end
      i = eigen_translate_l2g(i_1, 'Y')
      do j_1 = jloop_sta, jloop_end
        j = eigen_translate_l2g(j_1, 'X')
        a(j_1, i_1) = dble(min(i, j))
      end do
    end do
    ! element-wise ownership query as in the Matrix-Market reader of mat_set.f
    n1 = 7; n2 = 11
    i = eigen_owner_node(n1, 'Y')
    j = eigen_owner_node(n2, 'X')
    if (i == y_inod .and. j == x_inod) then
      i_1 = eigen_translate_g2l(n1, 'Y')
      j_1 = eigen_translate_g2l(n2, 'X')
      if (a(j_1, i_1) /= dble(min(n1, n2))) stop 4
    end if
    if (msolver == 0) then
      call eigen_sx(n, nvec, a, nm, w, z, nm, m_forward=m, m_backward=mb, mode=mode)
    else
      call eigen_s (n, nvec, a, nm, w, z, nm, m_forward=m, m_backward=mb, mode=mode)
    end if
    flops = a(1, 1)
    err = 0d0
    do i = 1, n
      lam = 1d0 / (2d0 * (1d0 - cos((2 * (n - i + 1) - 1) * PAI / (2 * n + 1))))
      err = max(err, abs(w(i) - lam) / lam)
    end do
    errmax = max(errmax, err)
    deallocate(a, z, w)
  end do
  if (inod == 1) then
    call eigen_show_version()
    print '(A,I0,A,I0,A,I0,A,ES10.2)', " ref_caller: ranks ", nnod, " grid ", x_nnod, "x", y_nnod, &
          " max rel eigenvalue error ", errmax
    if (errmax < sqrt(EPS)) then
      print *, "REF_CALLER PASSED"
    else
      print *, "REF_CALLER FAILED"
    end if
  end if
  call eigen_free()
  call MPI_Finalize(ierr)
end program ref_caller
