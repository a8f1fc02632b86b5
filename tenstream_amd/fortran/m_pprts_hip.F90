!-------------------------------------------------------------------------
! m_pprts_hip -- thin ISO_C_BINDING shim between TenStream's pprts driver and libtsx (include/tsx.h).
!
! This module is what a patched TenStream links: it sits next to src/pprts_explicit.F90 and is called from
! the diffuse-solve branch of `pprts()` (src/pprts.F90:2794-2813) when `-<prefix>hip` is set, with exactly
! the data that branch already has in hand (see INTEGRATION.md for the patch):
!     solver%C_diff, solver%diff2diff, atm%l1d/a11/a12/albedo, solver%b, solution%ediff, tolerances.
! It contains no arithmetic: arrays are passed by address in the reference's own (dof, z, x, y) layout and
! real(ireals)=real64 kind; the library owns device mirrors.  Errors follow the reference's convention: a
! non-zero return or a negative KSP reason is handed back as ierr for CHKERR (src/helper_functions.fypp:888-904).
!-------------------------------------------------------------------------
module m_pprts_hip
  use iso_c_binding, only: c_int32_t, c_int64_t, c_int, c_double, c_float, c_ptr, c_null_ptr, c_loc, c_char, c_associated, &
    & c_int8_t, c_f_pointer, c_size_t, c_funptr
  implicit none
  private
  public :: t_tsx_grid, t_tsx_ksp_opts, t_tsx_ksp_result, &
    & hip_diff_create, hip_diff_destroy, hip_diff_set_coeffs, hip_ediff, hip_diff_apply, hip_last_error, &
    & hip_dir_set_coeffs, hip_edir, hip_setup_b_solar, hip_setup_b_thermal, &
    & tsx_abi_sizes, tsx_comm_peer_export, tsx_comm_peer_attach, tsx_comm_peer_selftest, tsx_comm_peer_disable, &
    & tsx_comm_peer_set_fences, tsx_comm_peer_reset, &
    & tsx_comm_unique_id, tsx_comm_init, tsx_comm_set_callbacks, tsx_determine_ksp_tolerances, tsx_default_ksp_opts, &
    & tsx_lut_set_diffuse, tsx_lut_load_diffuse_mmap4, tsx_lut_set_direct, tsx_lut_load_direct_mmap4, &
    & tsx_diff_set_optprop, tsx_diff_get_coeffs, &
    & tsx_pprts_set_angles, tsx_pprts_set_direct_tolerances, tsx_pprts_set_optical_properties, tsx_pprts_set_optprop, &
    & tsx_pprts_solve, tsx_pprts_zero_guess, tsx_pprts_select_solution, tsx_pprts_get_result, tsx_pprts_get_field, &
    & TSX_HOST, TSX_DEVICE, TSX_PC_NONE, TSX_PC_COLUMN, TSX_PC_ZEBRA, TSX_PC_REDBLACK

  !> the seam's VECTORS in either real kind: TenStream's ireals is real32 or real64 by build (src/data_parameters.F90), so
  !> `call hip_ediff(h, solver%b, solution%ediff, ...)` resolves to the matching specific whatever the build chose.
  !> Three specifics per entry: everything real64 (`_r64`); real32 vectors with real64 scalars and fields (`_r32`, rounds 4-5: a
  !> caller that converts its small arrays itself); and, round 6, EVERYTHING real32 (`_k4`): a TenStream built with ireals = real32
  !> declares edirTOA, rtol / atol, albedo, planck, planck_srfc, kabs, dz, a11 / a12 / a13 / a23 / a33 and the coefficient arrays
  !> dir2dir / dir2diff / diff2diff real(ireals) too (src/pprts_base.F90:112-119, 252) and calls with them as they are -- the
  !> coefficient arrays cross as real32 (coeff_kind = 4, lossless: the LUT delivers real32), the small fields and scalars are
  !> widened in here.
  interface hip_ediff
    module procedure hip_ediff_r64, hip_ediff_r32, hip_ediff_k4
  end interface
  interface hip_diff_apply
    module procedure hip_diff_apply_r64, hip_diff_apply_r32
  end interface
  interface hip_edir
    module procedure hip_edir_r64, hip_edir_r32, hip_edir_k4
  end interface
  interface hip_setup_b_solar
    module procedure hip_setup_b_solar_r64, hip_setup_b_solar_r32, hip_setup_b_solar_k4
  end interface
  interface hip_setup_b_thermal
    module procedure hip_setup_b_thermal_r64, hip_setup_b_thermal_r32, hip_setup_b_thermal_k4
  end interface
  interface hip_diff_set_coeffs
    module procedure hip_diff_set_coeffs_r64, hip_diff_set_coeffs_k4
  end interface
  interface hip_dir_set_coeffs
    module procedure hip_dir_set_coeffs_r64, hip_dir_set_coeffs_k4
  end interface

  integer(c_int), parameter :: TSX_HOST = 0, TSX_DEVICE = 1
  integer(c_int), parameter :: TSX_PC_NONE = 0, TSX_PC_COLUMN = 1, TSX_PC_ZEBRA = 2, TSX_PC_REDBLACK = 3

  ! mirrors tsx_grid (include/tsx.h) == the fields of t_coord the back-end needs (src/pprts_base.F90:92-109)
  type, bind(C) :: t_tsx_grid
    integer(c_int32_t) :: solver_id            ! c_wrapper/f2c_solver_ids.h: 310 (3_10), 816 (8_16)
    integer(c_int32_t) :: Nz                   ! C_diff%zm - 1
    integer(c_int32_t) :: xm, ym               ! C_diff%xm, %ym
    integer(c_int32_t) :: xs, ys               ! C_diff%xs, %ys
    integer(c_int32_t) :: glob_xm, glob_ym
    integer(c_int32_t) :: rank, nranks         ! solver%myid, size(solver%comm)
    integer(c_int32_t) :: neigh_w, neigh_e     ! C_diff%neighbors(10), (16)
    integer(c_int32_t) :: neigh_s, neigh_n     ! C_diff%neighbors(4), (22)
    integer(c_int32_t) :: device
    integer(c_int32_t) :: force_halo
  end type

  type, bind(C) :: t_tsx_ksp_opts
    real(c_double) :: rtol, atol, dtol
    integer(c_int32_t) :: maxit
    integer(c_int32_t) :: pc
    integer(c_int32_t) :: pc_sweeps
    integer(c_int32_t) :: check_every
    integer(c_int32_t) :: fp32_directions
    integer(c_int32_t) :: pc_coeff_fp16
    integer(c_int32_t) :: skip_complete_initial_run   ! 0 (default): -ksp_complete_initial_run semantics, src/pprts.F90:4245-4256
    integer(c_int32_t) :: explicit_solver             ! 1: explicit_ediff's stationary iteration instead of the Krylov solve
    integer(c_int32_t) :: accept_incomplete_solve     ! 1: -accept_incomplete_solve, no retry from zero (src/pprts.F90:4271-4273)
    integer(c_int32_t) :: initial_guess_zero          ! 1: x is not read, the solve starts from r = b (tsx_diff_solve)
  end type

  type, bind(C) :: t_tsx_ksp_result
    integer(c_int32_t) :: reason
    integer(c_int32_t) :: niter
    real(c_double) :: rnorm0, rnorm
    real(c_double) :: res_hist(100)
    integer(c_int32_t) :: nhist
    real(c_float) :: solve_ms
    real(c_float) :: import_ms, export_ms
  end type

  interface
    function tsx_create(grid, handle) bind(C, name='tsx_create') result(ierr)
      import :: t_tsx_grid, c_ptr, c_int
      type(t_tsx_grid), intent(in) :: grid
      type(c_ptr), intent(out) :: handle
      integer(c_int) :: ierr
    end function
    function tsx_destroy(handle) bind(C, name='tsx_destroy') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
      integer(c_int) :: ierr
    end function
    subroutine tsx_default_ksp_opts(opts) bind(C, name='tsx_default_ksp_opts')
      import :: t_tsx_ksp_opts
      type(t_tsx_ksp_opts), intent(out) :: opts
    end subroutine
    function tsx_diff_set_coeffs(handle, diff2diff, coeff_kind, l1d, a11, a12, albedo, where) &
        & bind(C, name='tsx_diff_set_coeffs') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, diff2diff, l1d, a11, a12, albedo
      integer(c_int), value :: coeff_kind, where
      integer(c_int) :: ierr
    end function
    function tsx_diff_apply(handle, x, y, where) bind(C, name='tsx_diff_apply') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, x, y
      integer(c_int), value :: where
      integer(c_int) :: ierr
    end function
    function tsx_diff_solve(handle, b, x, where, opts, res) bind(C, name='tsx_diff_solve') result(ierr)
      import :: c_ptr, c_int, t_tsx_ksp_opts, t_tsx_ksp_result
      type(c_ptr), value :: handle, b, x
      integer(c_int), value :: where
      type(t_tsx_ksp_opts), intent(in) :: opts
      type(t_tsx_ksp_result), intent(out) :: res
      integer(c_int) :: ierr
    end function
    ! device-resident peer transport (node-local): export this rank's 192-byte blob, MPI_Allgather them in rank order, attach
    function tsx_comm_peer_export(handle, blob) bind(C, name='tsx_comm_peer_export') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, blob
      integer(c_int) :: ierr
    end function
    function tsx_comm_peer_attach(handle, blobs) bind(C, name='tsx_comm_peer_attach') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, blobs
      integer(c_int) :: ierr
    end function
    ! collective self test of the attached transport; failed = 0 if this rank saw nothing wrong (reduce it over the ranks)
    function tsx_comm_peer_selftest(handle, rounds, failed) bind(C, name='tsx_comm_peer_selftest') result(ierr)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: handle
      integer(c_int), value :: rounds
      real(c_double), intent(out) :: failed
      integer(c_int) :: ierr
    end function
    function tsx_comm_peer_disable(handle) bind(C, name='tsx_comm_peer_disable') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
      integer(c_int) :: ierr
    end function
    ! heavy = 1: full system-scope fences around every flag (after a failed self test); reset: back to the state after attach
    function tsx_comm_peer_set_fences(handle, heavy) bind(C, name='tsx_comm_peer_set_fences') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
      integer(c_int), value :: heavy
      integer(c_int) :: ierr
    end function
    function tsx_comm_peer_reset(handle) bind(C, name='tsx_comm_peer_reset') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
      integer(c_int) :: ierr
    end function
    ! ---- the seam in the caller's real kind (vec_kind 4 | 8), the direct seam, setup_b (include/tsx.h)
    function tsx_diff_apply_r(handle, x, y, vec_kind, where) bind(C, name='tsx_diff_apply_r') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, x, y
      integer(c_int), value :: vec_kind, where
      integer(c_int) :: ierr
    end function
    function tsx_diff_solve_r(handle, b, x, vec_kind, where, opts, res) bind(C, name='tsx_diff_solve_r') result(ierr)
      import :: c_ptr, c_int, t_tsx_ksp_opts, t_tsx_ksp_result
      type(c_ptr), value :: handle, b, x
      integer(c_int), value :: vec_kind, where
      type(t_tsx_ksp_opts), intent(in) :: opts
      type(t_tsx_ksp_result), intent(out) :: res
      integer(c_int) :: ierr
    end function
    function tsx_dir_set_coeffs(handle, dir2dir, dir2diff, coeff_kind, l1d, a33, a13, a23, dx, dy, where) &
        & bind(C, name='tsx_dir_set_coeffs') result(ierr)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: handle, dir2dir, dir2diff, l1d, a33, a13, a23
      integer(c_int), value :: coeff_kind, where
      real(c_double), value :: dx, dy
      integer(c_int) :: ierr
    end function
    function tsx_dir_solve(handle, edirTOA, edir, vec_kind, where, rtol, atol, maxit, niter, residual, converged) &
        & bind(C, name='tsx_dir_solve') result(ierr)
      import :: c_ptr, c_int, c_double, c_int32_t
      type(c_ptr), value :: handle, edir
      real(c_double), value :: edirTOA, rtol, atol
      integer(c_int), value :: vec_kind, where
      integer(c_int32_t), value :: maxit
      integer(c_int32_t), intent(out) :: niter, converged
      real(c_double), intent(out) :: residual
      integer(c_int) :: ierr
    end function
    function tsx_setup_b_solar(handle, edir, albedo, b, vec_kind, where) bind(C, name='tsx_setup_b_solar') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, edir, albedo, b
      integer(c_int), value :: vec_kind, where
      integer(c_int) :: ierr
    end function
    function tsx_setup_b_thermal(handle, planck, planck_srfc, kabs, dz, dx, dy, b, vec_kind, where) &
        & bind(C, name='tsx_setup_b_thermal') result(ierr)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: handle, planck, planck_srfc, kabs, dz, b
      real(c_double), value :: dx, dy
      integer(c_int), value :: vec_kind, where
      integer(c_int) :: ierr
    end function
    ! ---- communicator: RCCL (rank 0 makes the 128-byte id, MPI_Bcast it, every rank inits) or host-staged callbacks for an
    !      MPI host without GPU-aware transport (exchange / allreduce: bind(C) functions, see include/tsx.h)
    function tsx_comm_unique_id(id128) bind(C, name='tsx_comm_unique_id') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: id128
      integer(c_int) :: ierr
    end function
    function tsx_comm_init(handle, id128) bind(C, name='tsx_comm_init') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, id128
      integer(c_int) :: ierr
    end function
    function tsx_comm_set_callbacks(handle, exchange, allreduce, ctx) bind(C, name='tsx_comm_set_callbacks') result(ierr)
      import :: c_ptr, c_funptr, c_int
      type(c_ptr), value :: handle, ctx
      type(c_funptr), value :: exchange, allreduce
      integer(c_int) :: ierr
    end function
    function tsx_determine_ksp_tolerances(handle, unconstrained_fraction, rtol, atol, maxit) &
        & bind(C, name='tsx_determine_ksp_tolerances') result(ierr)
      import :: c_ptr, c_int, c_double, c_int32_t
      type(c_ptr), value :: handle
      real(c_double), value :: unconstrained_fraction
      real(c_double), intent(out) :: rtol, atol
      integer(c_int32_t), intent(out) :: maxit
      integer(c_int) :: ierr
    end function
    ! ---- coefficient tables on the device (optprop_LUT payloads, src/optprop_base.F90:438-442; `.mmap4`, src/mmap.F90:63-203)
    function tsx_lut_set_diffuse(handle, table, nvec, nentries, ndim, n, axes_concat, where) &
        & bind(C, name='tsx_lut_set_diffuse') result(ierr)
      import :: c_ptr, c_int, c_int32_t, c_int64_t
      type(c_ptr), value :: handle, table, n, axes_concat
      integer(c_int32_t), value :: nvec, ndim
      integer(c_int64_t), value :: nentries
      integer(c_int), value :: where
      integer(c_int) :: ierr
    end function
    function tsx_lut_load_diffuse_mmap4(handle, path) bind(C, name='tsx_lut_load_diffuse_mmap4') result(ierr)
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: handle
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int) :: ierr
    end function
    function tsx_lut_set_direct(handle, Tdir, Sdir, nentries, ndim, n, axes_concat, where) &
        & bind(C, name='tsx_lut_set_direct') result(ierr)
      import :: c_ptr, c_int, c_int32_t, c_int64_t
      type(c_ptr), value :: handle, Tdir, Sdir, n, axes_concat
      integer(c_int64_t), value :: nentries
      integer(c_int32_t), value :: ndim
      integer(c_int), value :: where
      integer(c_int) :: ierr
    end function
    function tsx_lut_load_direct_mmap4(handle, tdir_path, sdir_path) bind(C, name='tsx_lut_load_direct_mmap4') result(ierr)
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: handle
      character(kind=c_char), intent(in) :: tdir_path(*), sdir_path(*)
      integer(c_int) :: ierr
    end function
    function tsx_diff_set_optprop(handle, kabs, ksca, g, dz, dx, l1d, a11, a12, albedo, where) &
        & bind(C, name='tsx_diff_set_optprop') result(ierr)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: handle, kabs, ksca, g, dz, l1d, a11, a12, albedo
      real(c_double), value :: dx
      integer(c_int), value :: where
      integer(c_int) :: ierr
    end function
    function tsx_diff_get_coeffs(handle, diff2diff, where) bind(C, name='tsx_diff_get_coeffs') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, diff2diff
      integer(c_int), value :: where
      integer(c_int) :: ierr
    end function
    ! ---- a whole g-point on the device: set_angles / set_optical_properties / solve_pprts / pprts_get_result
    !      (src/pprts.F90:1100, 1764, 2487, 5799); NULL (c_null_ptr) where tsx.h allows it (planck, planck_srfc, opts, res, edir)
    function tsx_pprts_set_angles(handle, phi0, theta0) bind(C, name='tsx_pprts_set_angles') result(ierr)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: handle
      real(c_double), value :: phi0, theta0
      integer(c_int) :: ierr
    end function
    function tsx_pprts_set_direct_tolerances(handle, rtol, atol, maxit) bind(C, name='tsx_pprts_set_direct_tolerances') result(ierr)
      import :: c_ptr, c_int, c_double, c_int32_t
      type(c_ptr), value :: handle
      real(c_double), value :: rtol, atol
      integer(c_int32_t), value :: maxit
      integer(c_int) :: ierr
    end function
    function tsx_pprts_set_optical_properties(handle, albedo, kabs, ksca, g, dz, planck, planck_srfc, dx, dy, ldelta_scaling, &
        & where) bind(C, name='tsx_pprts_set_optical_properties') result(ierr)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: handle, albedo, kabs, ksca, g, dz, planck, planck_srfc
      real(c_double), value :: dx, dy
      integer(c_int), value :: ldelta_scaling, where
      integer(c_int) :: ierr
    end function
    function tsx_pprts_set_optprop(handle, kabs, ksca, g, dz, dx, dy, albedo, l1d, a11, a12, a13, a23, a33, planck, planck_srfc, &
        & where) bind(C, name='tsx_pprts_set_optprop') result(ierr)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: handle, kabs, ksca, g, dz, albedo, l1d, a11, a12, a13, a23, a33, planck, planck_srfc
      real(c_double), value :: dx, dy
      integer(c_int), value :: where
      integer(c_int) :: ierr
    end function
    function tsx_pprts_solve(handle, edirTOA, lsolar, opts, res) bind(C, name='tsx_pprts_solve') result(ierr)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: handle, opts, res   ! c_loc of a t_tsx_ksp_opts / t_tsx_ksp_result, or c_null_ptr
      real(c_double), value :: edirTOA
      integer(c_int), value :: lsolar
      integer(c_int) :: ierr
    end function
    function tsx_pprts_zero_guess(handle) bind(C, name='tsx_pprts_zero_guess') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
      integer(c_int) :: ierr
    end function
    function tsx_pprts_select_solution(handle, uid) bind(C, name='tsx_pprts_select_solution') result(ierr)
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: uid
      integer(c_int) :: ierr
    end function
    function tsx_pprts_get_result(handle, edn, eup, abso, edir, where) bind(C, name='tsx_pprts_get_result') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, edn, eup, abso, edir
      integer(c_int), value :: where
      integer(c_int) :: ierr
    end function
    function tsx_pprts_get_field(handle, which, out, where) bind(C, name='tsx_pprts_get_field') result(ierr)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, out
      integer(c_int), value :: which, where
      integer(c_int) :: ierr
    end function
    function tsx_abi_sizes(sizes3) bind(C, name='tsx_abi_sizes') result(ierr)
      import :: c_int32_t, c_int
      integer(c_int32_t), intent(out) :: sizes3(3)
      integer(c_int) :: ierr
    end function
    function tsx_last_error() bind(C, name='tsx_last_error') result(msg)
      import :: c_ptr
      type(c_ptr) :: msg
    end function
  end interface

contains

  !> create the device-side mirror of C_diff; call once per solver (like init_Matrix, src/pprts.F90:1240-1289)
  subroutine hip_diff_create(grid, handle, ierr)
    type(t_tsx_grid), intent(in) :: grid
    type(c_ptr), intent(out) :: handle
    integer(c_int), intent(out) :: ierr
    ierr = tsx_create(grid, handle)
  end subroutine

  subroutine hip_diff_destroy(handle, ierr)
    type(c_ptr), intent(inout) :: handle
    integer(c_int), intent(out) :: ierr
    ierr = 0
    if (c_associated(handle)) ierr = tsx_destroy(handle)
    handle = c_null_ptr
  end subroutine

  !> replaces set_diff_coeff (src/pprts.F90:5511-5796): hand over solver%diff2diff and the 1-D layer data.
  !> l1d is passed as 0/1 bytes (atm%l1d(atmk(atm,k)) for k = zs..ze-1).
  subroutine hip_diff_set_coeffs_r64(handle, diff2diff, l1d, a11, a12, albedo, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_double), target, contiguous, intent(in) :: diff2diff(:, :, :, :) ! (D*D, zs:ze-1, xs:xe, ys:ye)
    integer(c_int8_t), target, contiguous, intent(in) :: l1d(:)              ! (zs:ze-1)
    real(c_double), target, contiguous, intent(in) :: a11(:, :, :), a12(:, :, :) ! (zs:ze-1, xs:xe, ys:ye)
    real(c_double), target, contiguous, intent(in) :: albedo(:, :)           ! (xs:xe, ys:ye)
    integer(c_int), intent(out) :: ierr
    ierr = tsx_diff_set_coeffs(handle, c_loc(diff2diff), 8_c_int, c_loc(l1d), c_loc(a11), c_loc(a12), c_loc(albedo), &
      & TSX_HOST)
  end subroutine

  !> y = (I - T) x, the MatShell MatMult (op_mat_mult_ediff, src/pprts_shell.F90:366-541)
  subroutine hip_diff_apply_r64(handle, x, y, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_double), target, contiguous, intent(in) :: x(:, :, :, :)  ! (0:D-1, zs:ze, xs:xe, ys:ye)
    real(c_double), target, contiguous, intent(inout) :: y(:, :, :, :)
    integer(c_int), intent(out) :: ierr
    ierr = tsx_diff_apply(handle, c_loc(x), c_loc(y), TSX_HOST)
  end subroutine
  subroutine hip_diff_apply_r32(handle, x, y, ierr)   ! ireals = real32: the vectors cross as they are
    type(c_ptr), intent(in) :: handle
    real(c_float), target, contiguous, intent(in) :: x(:, :, :, :)
    real(c_float), target, contiguous, intent(inout) :: y(:, :, :, :)
    integer(c_int), intent(out) :: ierr
    ierr = tsx_diff_apply_r(handle, c_loc(x), c_loc(y), 4_c_int, TSX_HOST)
  end subroutine

  !> same contract as explicit_ediff(solver, prefix, vb, vediff, solution, ierr) (src/pprts_explicit.F90:461):
  !> vediff holds the initial guess on entry and the solution on exit; niter / residual history go to
  !> solution%Niter_diff / solution%diff_ksp_residual_history (src/pprts_base.F90:163-166).
  !> ierr = library error code, or -reason when the Krylov solver stopped with a negative KSP reason.
  subroutine hip_ediff_r64(handle, vb, vediff, rtol, atol, maxit, pc, niter, res_hist, reason, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_double), target, contiguous, intent(in) :: vb(:, :, :, :)
    real(c_double), target, contiguous, intent(inout) :: vediff(:, :, :, :)
    real(c_double), intent(in) :: rtol, atol
    integer(c_int), intent(in) :: maxit, pc
    integer(c_int), intent(out) :: niter
    real(c_double), intent(inout) :: res_hist(:)
    integer(c_int), intent(out) :: reason
    integer(c_int), intent(out) :: ierr
    type(t_tsx_ksp_opts) :: opts
    type(t_tsx_ksp_result) :: res
    integer :: n

    call tsx_default_ksp_opts(opts)
    opts%rtol = rtol
    opts%atol = atol
    opts%maxit = maxit
    opts%pc = pc
    if (pc .eq. TSX_PC_ZEBRA) opts%pc_sweeps = 5
    if (pc .eq. TSX_PC_REDBLACK) opts%pc_sweeps = 0  ! automatic (tsx_default_ksp_opts)
    ierr = tsx_diff_solve(handle, c_loc(vb), c_loc(vediff), TSX_HOST, opts, res)
    niter = res%niter
    reason = res%reason
    n = min(size(res_hist), int(res%nhist))
    if (n .gt. 0) res_hist(1:n) = res%res_hist(1:n)
    if (ierr .eq. 0 .and. reason .le. 0) ierr = -reason
  end subroutine

  !> hip_ediff for ireals = real32 (tsx_diff_solve_r, vec_kind 4); tolerances and the residual history stay real64
  subroutine hip_ediff_r32(handle, vb, vediff, rtol, atol, maxit, pc, niter, res_hist, reason, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_float), target, contiguous, intent(in) :: vb(:, :, :, :)
    real(c_float), target, contiguous, intent(inout) :: vediff(:, :, :, :)
    real(c_double), intent(in) :: rtol, atol
    integer(c_int), intent(in) :: maxit, pc
    integer(c_int), intent(out) :: niter
    real(c_double), intent(inout) :: res_hist(:)
    integer(c_int), intent(out) :: reason
    integer(c_int), intent(out) :: ierr
    type(t_tsx_ksp_opts) :: opts
    type(t_tsx_ksp_result) :: res
    integer :: n

    call tsx_default_ksp_opts(opts)
    opts%rtol = rtol
    opts%atol = atol
    opts%maxit = maxit
    opts%pc = pc
    if (pc .eq. TSX_PC_ZEBRA) opts%pc_sweeps = 5
    if (pc .eq. TSX_PC_REDBLACK) opts%pc_sweeps = 0
    ierr = tsx_diff_solve_r(handle, c_loc(vb), c_loc(vediff), 4_c_int, TSX_HOST, opts, res)
    niter = res%niter
    reason = res%reason
    n = min(size(res_hist), int(res%nhist))
    if (n .gt. 0) res_hist(1:n) = res%res_hist(1:n)
    if (ierr .eq. 0 .and. reason .le. 0) ierr = -reason
  end subroutine

  !> replaces set_dir_coeff (src/pprts.F90:4493-4630): hand over solver%dir2dir / solver%dir2diff (c_null_ptr-able through the
  !> optional) and the 1-D layer data atm%a33 / a13 / a23.  set_angles (tsx_pprts_set_angles) comes first.
  subroutine hip_dir_set_coeffs_r64(handle, dir2dir, l1d, dx, dy, ierr, dir2diff, a33, a13, a23)
    type(c_ptr), intent(in) :: handle
    real(c_double), target, contiguous, intent(in) :: dir2dir(:, :, :, :)             ! (S*S, zs:ze-1, xs:xe, ys:ye)
    integer(c_int8_t), target, contiguous, intent(in) :: l1d(:)
    real(c_double), intent(in) :: dx, dy
    integer(c_int), intent(out) :: ierr
    real(c_double), target, contiguous, intent(in), optional :: dir2diff(:, :, :, :)  ! (S*D, zs:ze-1, xs:xe, ys:ye)
    real(c_double), target, contiguous, intent(in), optional :: a33(:, :, :), a13(:, :, :), a23(:, :, :)
    type(c_ptr) :: p_sd, p33, p13, p23
    p_sd = c_null_ptr; p33 = c_null_ptr; p13 = c_null_ptr; p23 = c_null_ptr
    if (present(dir2diff)) p_sd = c_loc(dir2diff)
    if (present(a33)) p33 = c_loc(a33)
    if (present(a13)) p13 = c_loc(a13)
    if (present(a23)) p23 = c_loc(a23)
    ierr = tsx_dir_set_coeffs(handle, c_loc(dir2dir), p_sd, 8_c_int, c_loc(l1d), p33, p13, p23, dx, dy, TSX_HOST)
  end subroutine

  !> same contract as explicit_edir(solver, prefix, edirTOA, vedir, lb, v0, solution, ierr) (src/pprts_explicit.F90:60): vedir
  !> (0:S-1, zs:ze, xs:xe, ys:ye) holds the initial iterate on entry (the caller's v0 is a ghosted copy of it, src/pprts.F90:2746)
  !> and the beam on exit; the incoming solar radiation lb = setup_incSolar(edirTOA) is formed on the device.  rtol / atol /
  !> maxit as explicit_edir derives them from -solar_dir_ksp_* (:94-121); niter and residual go to solution%Niter_dir /
  !> solution%dir_ksp_residual_history.  lconverged = .false. when maxit sweeps did not suffice.
  subroutine hip_edir_r64(handle, edirTOA, vedir, rtol, atol, maxit, niter, residual, lconverged, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_double), intent(in) :: edirTOA
    real(c_double), target, contiguous, intent(inout) :: vedir(:, :, :, :)
    real(c_double), intent(in) :: rtol, atol
    integer(c_int), intent(in) :: maxit
    integer(c_int), intent(out) :: niter
    real(c_double), intent(out) :: residual
    logical, intent(out) :: lconverged
    integer(c_int), intent(out) :: ierr
    integer(c_int32_t) :: it, cv
    ierr = tsx_dir_solve(handle, edirTOA, c_loc(vedir), 8_c_int, TSX_HOST, rtol, atol, int(maxit, c_int32_t), it, residual, cv)
    niter = it
    lconverged = cv .ne. 0
  end subroutine
  subroutine hip_edir_r32(handle, edirTOA, vedir, rtol, atol, maxit, niter, residual, lconverged, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_double), intent(in) :: edirTOA
    real(c_float), target, contiguous, intent(inout) :: vedir(:, :, :, :)
    real(c_double), intent(in) :: rtol, atol
    integer(c_int), intent(in) :: maxit
    integer(c_int), intent(out) :: niter
    real(c_double), intent(out) :: residual
    logical, intent(out) :: lconverged
    integer(c_int), intent(out) :: ierr
    integer(c_int32_t) :: it, cv
    ierr = tsx_dir_solve(handle, edirTOA, c_loc(vedir), 4_c_int, TSX_HOST, rtol, atol, int(maxit, c_int32_t), it, residual, cv)
    niter = it
    lconverged = cv .ne. 0
  end subroutine

  !> setup_b's solar branch (set_solar_source, src/pprts.F90:4684-4846) into solver%b from the beam hip_edir left on the device
  subroutine hip_setup_b_solar_r64(handle, albedo, b, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_double), target, contiguous, intent(in) :: albedo(:, :)
    real(c_double), target, contiguous, intent(inout) :: b(:, :, :, :)
    integer(c_int), intent(out) :: ierr
    ierr = tsx_setup_b_solar(handle, c_null_ptr, c_loc(albedo), c_loc(b), 8_c_int, TSX_HOST)
  end subroutine
  subroutine hip_setup_b_solar_r32(handle, albedo, b, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_double), target, contiguous, intent(in) :: albedo(:, :)
    real(c_float), target, contiguous, intent(inout) :: b(:, :, :, :)
    integer(c_int), intent(out) :: ierr
    ierr = tsx_setup_b_solar(handle, c_null_ptr, c_loc(albedo), c_loc(b), 4_c_int, TSX_HOST)
  end subroutine

  !> setup_b's thermal branch (set_thermal_source, src/pprts.F90:4848-4987); planck_srfc = atm%Bsrfc where allocated
  subroutine hip_setup_b_thermal_r64(handle, planck, kabs, dz, dx, dy, b, ierr, planck_srfc)
    type(c_ptr), intent(in) :: handle
    real(c_double), target, contiguous, intent(in) :: planck(:, :, :), kabs(:, :, :), dz(:, :, :)
    real(c_double), intent(in) :: dx, dy
    real(c_double), target, contiguous, intent(inout) :: b(:, :, :, :)
    integer(c_int), intent(out) :: ierr
    real(c_double), target, contiguous, intent(in), optional :: planck_srfc(:, :)
    type(c_ptr) :: ps
    ps = c_null_ptr
    if (present(planck_srfc)) ps = c_loc(planck_srfc)
    ierr = tsx_setup_b_thermal(handle, c_loc(planck), ps, c_loc(kabs), c_loc(dz), dx, dy, c_loc(b), 8_c_int, TSX_HOST)
  end subroutine
  subroutine hip_setup_b_thermal_r32(handle, planck, kabs, dz, dx, dy, b, ierr, planck_srfc)
    type(c_ptr), intent(in) :: handle
    real(c_double), target, contiguous, intent(in) :: planck(:, :, :), kabs(:, :, :), dz(:, :, :)
    real(c_double), intent(in) :: dx, dy
    real(c_float), target, contiguous, intent(inout) :: b(:, :, :, :)
    integer(c_int), intent(out) :: ierr
    real(c_double), target, contiguous, intent(in), optional :: planck_srfc(:, :)
    type(c_ptr) :: ps
    ps = c_null_ptr
    if (present(planck_srfc)) ps = c_loc(planck_srfc)
    ierr = tsx_setup_b_thermal(handle, c_loc(planck), ps, c_loc(kabs), c_loc(dz), dx, dy, c_loc(b), 4_c_int, TSX_HOST)
  end subroutine

  ! ---- ireals = real32 throughout (round 6): the same entries with every real argument real(c_float)
  subroutine hip_diff_set_coeffs_k4(handle, diff2diff, l1d, a11, a12, albedo, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_float), target, contiguous, intent(in) :: diff2diff(:, :, :, :)
    integer(c_int8_t), target, contiguous, intent(in) :: l1d(:)
    real(c_float), contiguous, intent(in) :: a11(:, :, :), a12(:, :, :)
    real(c_float), contiguous, intent(in) :: albedo(:, :)
    integer(c_int), intent(out) :: ierr
    real(c_double), allocatable, target :: d11(:, :, :), d12(:, :, :), dalb(:, :)
    allocate (d11, source=real(a11, c_double))
    allocate (d12, source=real(a12, c_double))
    allocate (dalb, source=real(albedo, c_double))
    ierr = tsx_diff_set_coeffs(handle, c_loc(diff2diff), 4_c_int, c_loc(l1d), c_loc(d11), c_loc(d12), c_loc(dalb), TSX_HOST)
  end subroutine
  subroutine hip_dir_set_coeffs_k4(handle, dir2dir, l1d, dx, dy, ierr, dir2diff, a33, a13, a23)
    type(c_ptr), intent(in) :: handle
    real(c_float), target, contiguous, intent(in) :: dir2dir(:, :, :, :)
    integer(c_int8_t), target, contiguous, intent(in) :: l1d(:)
    real(c_float), intent(in) :: dx, dy
    integer(c_int), intent(out) :: ierr
    real(c_float), target, contiguous, intent(in), optional :: dir2diff(:, :, :, :)
    real(c_float), contiguous, intent(in), optional :: a33(:, :, :), a13(:, :, :), a23(:, :, :)
    real(c_double), allocatable, target :: d33(:, :, :), d13(:, :, :), d23(:, :, :)
    type(c_ptr) :: p_sd, p33, p13, p23
    p_sd = c_null_ptr; p33 = c_null_ptr; p13 = c_null_ptr; p23 = c_null_ptr
    if (present(dir2diff)) p_sd = c_loc(dir2diff)
    if (present(a33)) then
      allocate (d33, source=real(a33, c_double))
      p33 = c_loc(d33)
    end if
    if (present(a13)) then
      allocate (d13, source=real(a13, c_double))
      p13 = c_loc(d13)
    end if
    if (present(a23)) then
      allocate (d23, source=real(a23, c_double))
      p23 = c_loc(d23)
    end if
    ierr = tsx_dir_set_coeffs(handle, c_loc(dir2dir), p_sd, 4_c_int, c_loc(l1d), p33, p13, p23, real(dx, c_double), real(dy, c_double), &
      & TSX_HOST)
  end subroutine
  subroutine hip_ediff_k4(handle, vb, vediff, rtol, atol, maxit, pc, niter, res_hist, reason, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_float), target, contiguous, intent(in) :: vb(:, :, :, :)
    real(c_float), target, contiguous, intent(inout) :: vediff(:, :, :, :)
    real(c_float), intent(in) :: rtol, atol
    integer(c_int), intent(in) :: maxit, pc
    integer(c_int), intent(out) :: niter
    real(c_float), intent(inout) :: res_hist(:)
    integer(c_int), intent(out) :: reason
    integer(c_int), intent(out) :: ierr
    real(c_double) :: h(size(res_hist))
    h = real(res_hist, c_double)
    call hip_ediff_r32(handle, vb, vediff, real(rtol, c_double), real(atol, c_double), maxit, pc, niter, h, reason, ierr)
    res_hist = real(h, c_float)
  end subroutine
  subroutine hip_edir_k4(handle, edirTOA, vedir, rtol, atol, maxit, niter, residual, lconverged, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_float), intent(in) :: edirTOA
    real(c_float), target, contiguous, intent(inout) :: vedir(:, :, :, :)
    real(c_float), intent(in) :: rtol, atol
    integer(c_int), intent(in) :: maxit
    integer(c_int), intent(out) :: niter
    real(c_float), intent(out) :: residual
    logical, intent(out) :: lconverged
    integer(c_int), intent(out) :: ierr
    real(c_double) :: r8
    call hip_edir_r32(handle, real(edirTOA, c_double), vedir, real(rtol, c_double), real(atol, c_double), maxit, niter, r8, lconverged, ierr)
    residual = real(r8, c_float)
  end subroutine
  subroutine hip_setup_b_solar_k4(handle, albedo, b, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_float), contiguous, intent(in) :: albedo(:, :)
    real(c_float), target, contiguous, intent(inout) :: b(:, :, :, :)
    integer(c_int), intent(out) :: ierr
    real(c_double), allocatable, target :: dalb(:, :)
    allocate (dalb, source=real(albedo, c_double))
    call hip_setup_b_solar_r32(handle, dalb, b, ierr)
  end subroutine
  subroutine hip_setup_b_thermal_k4(handle, planck, kabs, dz, dx, dy, b, ierr, planck_srfc)
    type(c_ptr), intent(in) :: handle
    real(c_float), contiguous, intent(in) :: planck(:, :, :), kabs(:, :, :), dz(:, :, :)
    real(c_float), intent(in) :: dx, dy
    real(c_float), target, contiguous, intent(inout) :: b(:, :, :, :)
    integer(c_int), intent(out) :: ierr
    real(c_float), contiguous, intent(in), optional :: planck_srfc(:, :)
    real(c_double), allocatable, target :: dpl(:, :, :), dka(:, :, :), ddz(:, :, :), dps(:, :)
    allocate (dpl, source=real(planck, c_double))
    allocate (dka, source=real(kabs, c_double))
    allocate (ddz, source=real(dz, c_double))
    if (present(planck_srfc)) then
      allocate (dps, source=real(planck_srfc, c_double))
      call hip_setup_b_thermal_r32(handle, dpl, dka, ddz, real(dx, c_double), real(dy, c_double), b, ierr, dps)
    else
      call hip_setup_b_thermal_r32(handle, dpl, dka, ddz, real(dx, c_double), real(dy, c_double), b, ierr)
    end if
  end subroutine

  !> text of the last library error on this thread
  function hip_last_error() result(msg)
    character(len=:), allocatable :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: n
    p = tsx_last_error()
    msg = ''
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, [1024])
    n = 0
    do while (n .lt. 1024)
      if (s(n + 1) .eq. achar(0)) exit
      n = n + 1
    end do
    allocate (character(len=n) :: msg)
    msg = transfer(s(1:n), msg)
  end function
end module
