! Fortran smoke driver for the ISO_C_BINDING shim: a homogeneous 3_10 box like config 1
! (examples/pprts -Nx 4 -Ny 4 -Nz 20 -dtau_cld 0), coefficients = a simple energy-conserving block,
! solved through hip_ediff; prints residual history and checks A x = b with hip_diff_apply.
program test_shim
  use iso_c_binding
  use m_pprts_hip
  implicit none
  integer, parameter :: D = 10, Nz = 20, Nx = 4, Ny = 4
  type(t_tsx_grid) :: grid
  type(c_ptr) :: h
  integer(c_int) :: ierr, niter, reason
  real(c_double), allocatable, target :: c(:, :, :, :), a11(:, :, :), a12(:, :, :), alb(:, :), b(:, :, :, :), x(:, :, :, :), y(:, :, :, :)
  integer(c_int8_t), allocatable, target :: l1d(:)
  real(c_double) :: hist(100), err
  integer :: src, dst

  block   ! the bind(C) types of the shim must be the library's structs, field for field
    integer(c_int32_t) :: sz(3)
    type(t_tsx_ksp_opts) :: o
    type(t_tsx_ksp_result) :: rr
    ierr = tsx_abi_sizes(sz)
    if (ierr .ne. 0 .or. sz(1) .ne. c_sizeof(grid) .or. sz(2) .ne. c_sizeof(o) .or. sz(3) .ne. c_sizeof(rr)) then
      print *, 'ABI mismatch: library', sz, ' shim', c_sizeof(grid), c_sizeof(o), c_sizeof(rr)
      stop 6
    end if
  end block
  grid = t_tsx_grid(310, Nz, Nx, Ny, 0, 0, Nx, Ny, 0, 1, 0, 0, 0, 0, -1, 0)
  call hip_diff_create(grid, h, ierr)
  if (ierr .ne. 0) then
    print *, 'create failed: ', hip_last_error()
    stop 2
  end if
  allocate (c(D * D, Nz, Nx, Ny), a11(Nz, Nx, Ny), a12(Nz, Nx, Ny), alb(Nx, Ny), l1d(Nz))
  allocate (b(D, Nz + 1, Nx, Ny), x(D, Nz + 1, Nx, Ny), y(D, Nz + 1, Nx, Ny))
  ! every source keeps 50% in its own stream and spreads 40% evenly: column sums 0.9 (10% absorbed)
  do dst = 1, D
    do src = 1, D
      c((dst - 1) * D + src, :, :, :) = 0.04_c_double
      if (src .eq. dst) c((dst - 1) * D + src, :, :, :) = 0.5_c_double + 0.04_c_double
    end do
  end do
  a11 = 0.5_c_double; a12 = 0.2_c_double; alb = 0.1_c_double; l1d = 0_c_int8_t
  l1d(1:2) = 1_c_int8_t
  call hip_diff_set_coeffs(h, c, l1d, a11, a12, alb, ierr)
  if (ierr .ne. 0) stop 3
  b = 0.01_c_double; b(2, 1, :, :) = 1._c_double   ! diffuse light entering at TOA (Edn at level 0) + weak sources
  b(3:, Nz + 1, :, :) = 0; b(3:, 1:2, :, :) = 0     ! identity rows (bottom side dummies, 1-D layers) carry no source
  x = 0; hist = -1
  call hip_ediff(h, b, x, 1e-10_c_double, 1e-30_c_double, 1000_c_int, TSX_PC_REDBLACK, niter, hist, reason, ierr)
  if (ierr .ne. 0) then
    print *, 'solve failed ierr=', ierr, ' reason=', reason, ' ', hip_last_error()
    stop 4
  end if
  call hip_diff_apply(h, x, y, ierr)
  err = maxval(abs(y - b))
  print '(a,i0,a,i0,a,es10.3,a,es10.3)', 'shim ok: iterations ', niter, ' reason ', reason, ' r0 ', hist(1), ' max|Ax-b| ', err
  if (err .gt. 1e-9_c_double .or. reason .ne. 2) stop 5
  call hip_diff_destroy(h, ierr)
end program
