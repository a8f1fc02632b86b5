! Fortran smoke driver for the ISO_C_BINDING shim: a homogeneous 3_10 box like config 1
! (examples/pprts -Nx 4 -Ny 4 -Nz 20 -dtau_cld 0), coefficients = a simple energy-conserving block,
! solved through hip_ediff; prints residual history and checks A x = b with hip_diff_apply.
module test_shim_callbacks
  ! tsx_comm_set_callbacks from Fortran: the exchange an MPI host without GPU-aware transport would do with MPI_Sendrecv
  ! (exchange_diffuse_boundary's pattern, src/pprts_explicit.F90:769-843) -- here one rank whose four neighbours are itself:
  ! recv(W) <- send(E), recv(E) <- send(W), recv(S) <- send(N), recv(N) <- send(S)
  use iso_c_binding
  implicit none
  integer :: n_exchanges = 0, n_allreduces = 0
contains
  function shim_exchange(ctx, send, recv, cnt, peer) bind(C) result(ierr)
    type(c_ptr), value :: ctx
    type(c_ptr), intent(in) :: send(4), recv(4)
    integer(c_size_t), intent(in) :: cnt(4)
    integer(c_int), intent(in) :: peer(4)
    integer(c_int) :: ierr
    integer, parameter :: from(4) = [2, 1, 4, 3]
    real(c_double), pointer :: a(:), b(:)
    integer :: q
    ierr = 0
    do q = 1, 4
      if (cnt(q) .eq. 0) cycle
      if (peer(q) .ne. 0 .or. cnt(from(q)) .ne. cnt(q)) then
        ierr = 1
        return
      end if
      call c_f_pointer(recv(q), a, [cnt(q)])
      call c_f_pointer(send(from(q)), b, [cnt(q)])
      a = b
    end do
    n_exchanges = n_exchanges + 1
  end function
  function shim_allreduce(ctx, inout, n) bind(C) result(ierr)   ! one rank: the sum over the ranks is the value itself
    type(c_ptr), value :: ctx
    real(c_double), intent(inout) :: inout(*)
    integer(c_int), value :: n
    integer(c_int) :: ierr
    ierr = 0
    n_allreduces = n_allreduces + 1
  end function
end module

program test_shim
  use iso_c_binding
  use m_pprts_hip
  use test_shim_callbacks
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
  call check_real32()
  call check_all_real32()
  call check_direct_seam()
  call check_thermal_source()
  call check_comm_bindings()
  call check_pipeline()
  call hip_diff_destroy(h, ierr)
  print '(a)', 'shim all ok'

contains

  subroutine must(cond, code, what)
    logical, intent(in) :: cond
    integer, intent(in) :: code
    character(len=*), intent(in) :: what
    if (.not. cond) then
      print *, 'FAILED: ', what, ' -- ', hip_last_error()
      stop code
    end if
  end subroutine

  !> ireals = real32: the generic hip_ediff / hip_diff_apply resolve to the real32 specifics (tsx_diff_solve_r, vec_kind 4)
  subroutine check_real32()
    real(c_float), allocatable, target :: b4(:, :, :, :), x4(:, :, :, :), y4(:, :, :, :)
    real(c_double) :: h4(100)
    integer(c_int) :: it4, rs4, ie
    allocate (b4(D, Nz + 1, Nx, Ny), x4(D, Nz + 1, Nx, Ny), y4(D, Nz + 1, Nx, Ny))
    b4 = real(b, c_float); x4 = 0; h4 = -1
    call hip_ediff(h, b4, x4, 1e-10_c_double, 1e-30_c_double, 1000_c_int, TSX_PC_REDBLACK, it4, h4, rs4, ie)
    call must(ie .eq. 0 .and. rs4 .eq. 2, 10, 'real32 hip_ediff')
    call must(maxval(abs(real(x4, c_double) - x)) .le. 1e-6_c_double * maxval(abs(x)), 11, 'real32 solution equals the real64 one')
    call hip_diff_apply(h, x4, y4, ie)
    call must(ie .eq. 0 .and. maxval(abs(y4 - b4)) .le. 1e-5, 12, 'real32 hip_diff_apply')
    print '(a,i0,a,es10.3)', 'shim real32 ok: iterations ', it4, ' max|x32-x64| ', maxval(abs(real(x4, c_double) - x))
  end subroutine

  !> ireals = real32 THROUGHOUT (round 6, the `_k4` specifics): coefficients, 1-D layer data, albedo, tolerances, history and vectors
  !> all real(c_float), as a TenStream built with real32 ireals holds them (src/pprts_base.F90:112-119, 252) -- a second solver
  !> gets the same blocks as real32 (lossless: 0.04, 0.54 ... are rounded the same way on both sides once they are real32) and must
  !> reproduce the real32-vector solve of the first one
  subroutine check_all_real32()
    type(c_ptr) :: h2
    real(c_float), allocatable, target :: c4(:, :, :, :), a114(:, :, :), a124(:, :, :), alb4(:, :), b4(:, :, :, :), x4(:, :, :, :), xr(:, :, :, :)
    real(c_double), allocatable, target :: c8(:, :, :, :), a118(:, :, :), a128(:, :, :), alb8(:, :)
    real(c_float) :: h4(100)
    real(c_double) :: h8(100)
    integer(c_int) :: it4, rs4, it8, rs8, ie
    allocate (c4(D * D, Nz, Nx, Ny), a114(Nz, Nx, Ny), a124(Nz, Nx, Ny), alb4(Nx, Ny))
    allocate (b4(D, Nz + 1, Nx, Ny), x4(D, Nz + 1, Nx, Ny), xr(D, Nz + 1, Nx, Ny))
    c4 = real(c, c_float); a114 = real(a11, c_float); a124 = real(a12, c_float); alb4 = real(alb, c_float)
    b4 = real(b, c_float)
    call hip_diff_create(grid, h2, ie)
    call must(ie .eq. 0, 60, 'second solver')
    ! reference: the same real32-representable numbers handed over as real64, real32 vectors
    c8 = real(c4, c_double); a118 = real(a114, c_double); a128 = real(a124, c_double); alb8 = real(alb4, c_double)
    call hip_diff_set_coeffs(h2, c8, l1d, a118, a128, alb8, ie)
    call must(ie .eq. 0, 61, 'real64 coefficients')
    xr = 0; h8 = -1
    call hip_ediff(h2, b4, xr, 1e-6_c_double, 1e-30_c_double, 1000_c_int, TSX_PC_REDBLACK, it8, h8, rs8, ie)
    call must(ie .eq. 0 .and. rs8 .eq. 2, 62, 'reference solve')
    ! everything real32
    call hip_diff_set_coeffs(h2, c4, l1d, a114, a124, alb4, ie)
    call must(ie .eq. 0, 63, 'all-real32 hip_diff_set_coeffs')
    x4 = 0; h4 = -1
    call hip_ediff(h2, b4, x4, 1e-6_c_float, 1e-30_c_float, 1000_c_int, TSX_PC_REDBLACK, it4, h4, rs4, ie)
    call must(ie .eq. 0 .and. rs4 .eq. 2 .and. it4 .eq. it8, 64, 'all-real32 hip_ediff')
    call must(all(x4 .eq. xr), 65, 'all-real32 solve equals the real64-argument solve bit for bit')
    call must(abs(h4(1) - real(h8(1), c_float)) .le. 1e-6 * abs(h4(1)), 66, 'all-real32 residual history')
    call hip_diff_destroy(h2, ie)
    print '(a,i0)', 'shim all-real32 ok: iterations ', it4
  end subroutine

  !> the direct seam (src/pprts.F90:2698-2755): sun overhead, a homogeneous box whose cells pass 7/8 of the top stream on and
  !> scatter 1/16 of it into Edn -- the beam is Beer-Lambert, b follows from it (set_solar_source, :4684-4846)
  subroutine check_direct_seam()
    integer, parameter :: S = 3
    real(c_double), allocatable, target :: t(:, :, :, :), sd(:, :, :, :), e(:, :, :, :), bs(:, :, :, :), a33(:, :, :), a13(:, :, :), a23(:, :, :)
    real(c_float), allocatable, target :: e4(:, :, :, :)
    real(c_double), parameter :: dx = 100, dy = 50, s0 = 2
    real(c_double) :: res, want
    integer(c_int) :: it, ie
    integer(c_int8_t), allocatable, target :: no1d(:)
    logical :: conv
    integer :: k
    allocate (t(S * S, Nz, Nx, Ny), sd(S * D, Nz, Nx, Ny), e(S, Nz + 1, Nx, Ny), bs(D, Nz + 1, Nx, Ny), e4(S, Nz + 1, Nx, Ny))
    allocate (a33(Nz, Nx, Ny), a13(Nz, Nx, Ny), a23(Nz, Nx, Ny), no1d(Nz))
    t = 0; sd = 0; a33 = 0; a13 = 0; a23 = 0; no1d = 0_c_int8_t
    t(1, :, :, :) = 0.875_c_double              ! c(src 0 -> dst 0): flat index dst * S + src + 1
    sd(1 * S + 0 + 1, :, :, :) = 0.0625_c_double  ! c(src 0 -> diffuse dst 1 = Edn)
    ie = tsx_pprts_set_angles(h, 180._c_double, 0._c_double)
    call must(ie .eq. 0, 20, 'tsx_pprts_set_angles')
    call hip_dir_set_coeffs(h, t, no1d, dx, dy, ie, dir2diff=sd, a33=a33, a13=a13, a23=a23)
    call must(ie .eq. 0, 21, 'hip_dir_set_coeffs')
    e = 0
    call hip_edir(h, s0, e, 1e-12_c_double, 1e-30_c_double, 100_c_int, it, res, conv, ie)
    call must(ie .eq. 0 .and. conv, 22, 'hip_edir')
    do k = 0, Nz
      want = s0 * dx * dy * 0.875_c_double**k
      call must(maxval(abs(e(1, k + 1, :, :) - want)) .le. 1e-12_c_double * want, 23, 'beam is Beer-Lambert')
    end do
    call must(maxval(abs(e(2:, :, :, :))) .eq. 0, 24, 'no side streams with the sun overhead')
    call hip_setup_b_solar(h, alb, bs, ie)
    call must(ie .eq. 0, 25, 'hip_setup_b_solar')
    do k = 1, Nz   ! Edn (dof 1, inward) of level k <- the beam entering cell k-1
      call must(maxval(abs(bs(2, k + 1, :, :) - 0.0625_c_double * e(1, k, :, :))) .le. 1e-13_c_double * s0 * dx * dy, 26, 'b from the beam')
    end do
    call must(maxval(abs(bs(1, Nz + 1, :, :) - 0.1_c_double * e(1, Nz + 1, :, :))) .le. 1e-13_c_double * s0 * dx * dy, 27, 'surface reflection in b')
    e4 = 0
    call hip_edir(h, s0, e4, 1e-12_c_double, 1e-30_c_double, 100_c_int, it, res, conv, ie)   ! the real32 specific
    call must(ie .eq. 0 .and. conv .and. all(e4 .eq. real(e, c_float)), 28, 'real32 hip_edir')
    print '(a,i0,a,es10.3)', 'shim direct seam ok: sweeps ', it, ' residual ', res
  end subroutine

  !> setup_b's thermal branch with atm%Bsrfc (src/pprts.F90:4848-4987): the upward stream at the ground carries the skin's emission
  subroutine check_thermal_source()
    real(c_double), allocatable, target :: planck(:, :, :), skin(:, :), kabs(:, :, :), dz(:, :, :), bt(:, :, :, :), bt2(:, :, :, :)
    real(c_double), parameter :: dx = 100, dy = 50, pi = 3.14159265358979323846_c_double
    integer(c_int) :: ie
    allocate (planck(Nz + 1, Nx, Ny), skin(Nx, Ny), kabs(Nz, Nx, Ny), dz(Nz, Nx, Ny), bt(D, Nz + 1, Nx, Ny), bt2(D, Nz + 1, Nx, Ny))
    planck = 3; skin = 5; kabs = 1e-4_c_double; dz = 40
    call hip_setup_b_thermal(h, planck, kabs, dz, dx, dy, bt, ie)
    call must(ie .eq. 0, 30, 'hip_setup_b_thermal')
    call hip_setup_b_thermal(h, planck, kabs, dz, dx, dy, bt2, ie, planck_srfc=skin)
    call must(ie .eq. 0, 31, 'hip_setup_b_thermal with planck_srfc')
    call must(maxval(abs(bt2(1, Nz + 1, :, :) - bt(1, Nz + 1, :, :) - (5 - 3) * dx * dy * 0.9_c_double * pi)) .le. 1e-9_c_double, 32, &
      & 'Bsrfc term of the surface emission')
    bt2(1, Nz + 1, :, :) = bt(1, Nz + 1, :, :)
    call must(all(bt2 .eq. bt) .and. minval(bt) .ge. 0, 33, 'nothing else changes')
    print '(a,es12.5)', 'shim thermal source ok: surface emission ', bt(1, Nz + 1, 1, 1)
  end subroutine

  !> tsx_comm_set_callbacks and tsx_comm_unique_id / tsx_comm_init from Fortran: one rank whose faces go through the exchange
  !> (force_halo) solves the same system -- host-staged Fortran callbacks, then a 1-rank RCCL communicator
  subroutine check_comm_bindings()
    type(t_tsx_grid) :: gh
    type(c_ptr) :: hh
    real(c_double), allocatable, target :: xh(:, :, :, :)
    integer(c_int8_t), target :: id(128)
    real(c_double) :: hist2(100), rt, at
    integer(c_int32_t) :: mx
    integer(c_int) :: ie, it2, rs2
    integer :: pass
    allocate (xh(D, Nz + 1, Nx, Ny))
    do pass = 1, 2
      gh = grid
      gh%force_halo = 1
      call hip_diff_create(gh, hh, ie)
      call must(ie .eq. 0, 40, 'create (force_halo)')
      if (pass .eq. 1) then
        ie = tsx_comm_set_callbacks(hh, c_funloc(shim_exchange), c_funloc(shim_allreduce), c_null_ptr)
        call must(ie .eq. 0, 41, 'tsx_comm_set_callbacks')
      else
        ie = tsx_comm_unique_id(c_loc(id))
        call must(ie .eq. 0, 42, 'tsx_comm_unique_id')
        ie = tsx_comm_init(hh, c_loc(id))
        call must(ie .eq. 0, 43, 'tsx_comm_init')
      end if
      call hip_diff_set_coeffs(hh, c, l1d, a11, a12, alb, ie)
      call must(ie .eq. 0, 44, 'set_coeffs (force_halo)')
      xh = 0; hist2 = -1
      call hip_ediff(hh, b, xh, 1e-10_c_double, 1e-30_c_double, 1000_c_int, TSX_PC_REDBLACK, it2, hist2, rs2, ie)
      call must(ie .eq. 0 .and. rs2 .eq. 2, 45, 'solve through the exchange')
      call must(maxval(abs(xh - x)) .le. 1e-9_c_double * maxval(abs(x)), 46, 'same solution as the periodic rank')
      ie = tsx_determine_ksp_tolerances(hh, 1._c_double, rt, at, mx)
      call must(ie .eq. 0 .and. rt .eq. 1e-5_c_double .and. mx .eq. 1000, 47, 'tsx_determine_ksp_tolerances')
      call hip_diff_destroy(hh, ie)
    end do
    call must(n_exchanges .gt. 0, 48, 'the Fortran exchange callback ran')
    print '(a,i0,a)', 'shim comm bindings ok: ', n_exchanges, ' exchanges through the Fortran callback, then a 1-rank RCCL communicator'
  end subroutine

  !> tsx_lut_* and tsx_pprts_* from Fortran: a whole solar g-point and a thermal one with planck_srfc on constant tables
  !> (every LUT entry the same block), checked by the energy balance of the horizontally periodic box
  subroutine check_pipeline()
    integer, parameter :: S = 3, ne4 = 16, ne6 = 64
    type(c_ptr) :: hp
    real(c_float), allocatable, target :: tab(:, :), td(:, :), ts(:, :), axes4(:), axes6(:)
    integer(c_int32_t), target :: n4(4), n6(6)
    real(c_double), allocatable, target :: kabs(:, :, :), ksca(:, :, :), gg(:, :, :), dz(:, :, :), albp(:, :), planck(:, :, :), skin(:, :)
    real(c_double), allocatable, target :: edn(:, :, :), eup(:, :, :), abso(:, :, :), edir(:, :, :)
    type(t_tsx_ksp_result), target :: rr
    real(c_double), parameter :: dx = 100, dy = 100, s0 = 1000, theta = 40
    real(c_double) :: net_top, net_bot, atm, mu0
    integer(c_int) :: ie
    integer :: dst, src
    call hip_diff_create(grid, hp, ie)
    call must(ie .eq. 0, 50, 'create (pipeline)')
    allocate (tab(D * D, ne4), td(S * S, ne6), ts(S * D, ne6), axes4(8), axes6(12))
    do dst = 1, D
      do src = 1, D
        tab((dst - 1) * D + src, :) = 0.04
        if (src .eq. dst) tab((dst - 1) * D + src, :) = 0.54
      end do
    end do
    td = 0; ts = 0
    do src = 1, S
      td((src - 1) * S + src, :) = 0.75       ! every direct stream passes 3/4 on ...
      ts(src:S * D:S, :) = 0.02               ! ... and scatters 2 % into each of the ten diffuse streams (5 % absorbed)
    end do
    n4 = 2; n6 = 2
    axes4 = [1e-10, 100., 0., 0.99999, 0.02, 7.451, 0., 0.85]
    axes6 = [1e-10, 100., 0., 0.99999, 0.02, 7.451, 0., 0.85, 0., 90., 0., 90.]
    ie = tsx_lut_set_diffuse(hp, c_loc(tab), int(D * D, c_int32_t), int(ne4, c_int64_t), 4_c_int32_t, c_loc(n4), c_loc(axes4), TSX_HOST)
    call must(ie .eq. 0, 51, 'tsx_lut_set_diffuse')
    ie = tsx_lut_set_direct(hp, c_loc(td), c_loc(ts), int(ne6, c_int64_t), 6_c_int32_t, c_loc(n6), c_loc(axes6), TSX_HOST)
    call must(ie .eq. 0, 52, 'tsx_lut_set_direct')
    ie = tsx_pprts_set_angles(hp, 200._c_double, theta)
    call must(ie .eq. 0, 53, 'tsx_pprts_set_angles')
    allocate (kabs(Nz, Nx, Ny), ksca(Nz, Nx, Ny), gg(Nz, Nx, Ny), dz(Nz, Nx, Ny), albp(Nx, Ny), planck(Nz + 1, Nx, Ny), skin(Nx, Ny))
    allocate (edn(Nz + 1, Nx, Ny), eup(Nz + 1, Nx, Ny), abso(Nz, Nx, Ny), edir(Nz + 1, Nx, Ny))
    kabs = 1e-4_c_double; ksca = 1e-3_c_double; gg = 0.5_c_double; dz = 50; albp = 0.2_c_double
    ie = tsx_pprts_set_optical_properties(hp, c_loc(albp), c_loc(kabs), c_loc(ksca), c_loc(gg), c_loc(dz), c_null_ptr, c_null_ptr, &
      & dx, dy, 1_c_int, TSX_HOST)
    call must(ie .eq. 0, 54, 'tsx_pprts_set_optical_properties (solar)')
    ie = tsx_pprts_select_solution(hp, 1_c_int32_t)
    call must(ie .eq. 0, 55, 'tsx_pprts_select_solution')
    ie = tsx_pprts_solve(hp, s0, 1_c_int, c_null_ptr, c_loc(rr))
    call must(ie .eq. 0 .and. rr%reason .gt. 0, 56, 'tsx_pprts_solve (solar)')
    ie = tsx_pprts_get_result(hp, c_loc(edn), c_loc(eup), c_loc(abso), c_loc(edir), TSX_HOST)
    call must(ie .eq. 0, 57, 'tsx_pprts_get_result')
    mu0 = cos(theta * 3.14159265358979323846_c_double / 180)
    call must(maxval(abs(edir(1, :, :) - s0 * mu0)) .le. 1e-9_c_double * s0, 58, 'direct flux at TOA is S0 mu0')
    net_top = sum(edir(1, :, :) + edn(1, :, :) - eup(1, :, :))
    net_bot = sum(edir(Nz + 1, :, :) + edn(Nz + 1, :, :) - eup(Nz + 1, :, :))
    atm = sum(abso * dz)
    call must(abs(atm - (net_top - net_bot)) .le. 2e-4_c_double * net_top .and. minval(abso) .ge. 0, 59, 'solar energy balance')
    print '(a,i0,a,es10.3)', 'shim pipeline ok (solar): iterations ', rr%niter, ' balance ', abs(atm - (net_top - net_bot)) / net_top
    ! thermal, with the surface's own emission as the rrtmg driver passes it (rrtmg/rrtmg/pprts_rrtmg.F90:681)
    planck = 30; skin = 40
    ie = tsx_pprts_set_optical_properties(hp, c_loc(albp), c_loc(kabs), c_loc(ksca), c_loc(gg), c_loc(dz), c_loc(planck), c_loc(skin), &
      & dx, dy, 1_c_int, TSX_HOST)
    call must(ie .eq. 0, 60, 'tsx_pprts_set_optical_properties (thermal, planck_srfc)')
    ie = tsx_pprts_select_solution(hp, 501_c_int32_t)
    ie = tsx_pprts_solve(hp, 0._c_double, 0_c_int, c_null_ptr, c_loc(rr))
    call must(ie .eq. 0 .and. rr%reason .gt. 0, 61, 'tsx_pprts_solve (thermal)')
    ie = tsx_pprts_get_result(hp, c_loc(edn), c_loc(eup), c_loc(abso), c_null_ptr, TSX_HOST)
    call must(ie .eq. 0, 62, 'tsx_pprts_get_result (thermal)')
    ! the ground emits (1 - albedo) pi Bsrfc and reflects albedo Edn
    call must(maxval(abs(eup(Nz + 1, :, :) - (0.8_c_double * 40 * 3.14159265358979323846_c_double + 0.2_c_double * edn(Nz + 1, :, :)))) &
      & .le. 1e-4_c_double * maxval(eup), 63, 'surface emission from planck_srfc in the upward flux')
    print '(a,i0,a,f9.4)', 'shim pipeline ok (thermal): iterations ', rr%niter, ' Eup at the ground ', eup(Nz + 1, 1, 1)
    call hip_diff_destroy(hp, ie)
  end subroutine
end program
