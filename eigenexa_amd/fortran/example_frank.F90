!> Fortran caller in the style of the reference benchmark (benchmark/main2.f:173-216, :377-420, :558):
!! Frank matrix, eigen_sx, eigenvalue check against the analytic spectrum (benchmark/mat_set.f:638-647).
program example_frank
  use eigen_libs_mod
  implicit none
  integer, parameter :: n = 1000
  integer :: nx, ny, i, j
  real(8), allocatable :: a(:, :), z(:, :), w(:)
  real(8) :: lam, err, pi
  call eigen_init()
  call eigen_get_matdims(n, nx, ny)
  allocate(a(nx, ny), z(nx, ny), w(n))
  a = 0d0
  do j = 1, n
    do i = 1, n
      a(i, j) = dble(min(i, j))
    end do
  end do
  call eigen_sx(n, n, a, nx, w, z, nx, mode='A')
  pi = 4d0 * atan(1d0)
  err = 0d0
  do i = 1, n
    lam = 1d0 / (2d0 * (1d0 - cos((2 * (n - i + 1) - 1) * pi / (2 * n + 1))))
    err = max(err, abs(w(i) - lam) / lam)
  end do
  print *, "eigen_sx N=", n, " max rel eigenvalue error =", err, " flops=", a(1, 1), " seconds=", a(2, 1)
  ! tridiagonal route on the same matrix
  do j = 1, n
    do i = 1, n
      a(i, j) = dble(min(i, j))
    end do
  end do
  call eigen_s(n, n, a, nx, w, z, nx)
  err = 0d0
  do i = 1, n
    lam = 1d0 / (2d0 * (1d0 - cos((2 * (n - i + 1) - 1) * pi / (2 * n + 1))))
    err = max(err, abs(w(i) - lam) / lam)
  end do
  print *, "eigen_s  N=", n, " max rel eigenvalue error =", err
  ! complex Hermitian route: D F D^H with a unitary diagonal D has the Frank spectrum
  block
    complex(8), allocatable :: ah(:, :), zh(:, :)
    complex(8) :: pi_i, pj
    allocate(ah(n, n), zh(n, n))
    do j = 1, n
      pj = exp(cmplx(0d0, 0.37d0 * j, kind=8))
      do i = 1, n
        pi_i = exp(cmplx(0d0, 0.37d0 * i, kind=8))
        ah(i, j) = pi_i * dble(min(i, j)) * conjg(pj)
      end do
    end do
    call eigen_h(n, n, ah, n, w, zh, n)
    err = 0d0
    do i = 1, n
      lam = 1d0 / (2d0 * (1d0 - cos((2 * (n - i + 1) - 1) * pi / (2 * n + 1))))
      err = max(err, abs(w(i) - lam) / lam)
    end do
    print *, "eigen_h  N=", n, " max rel eigenvalue error =", err
  end block
  call eigen_free()
end program example_frank
