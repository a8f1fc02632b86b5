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
  use iso_c_binding, only: c_int32_t, c_int, c_double, c_float, c_ptr, c_null_ptr, c_loc, c_char, c_associated, &
    & c_int8_t, c_f_pointer, c_size_t
  implicit none
  private
  public :: t_tsx_grid, t_tsx_ksp_opts, t_tsx_ksp_result, &
    & hip_diff_create, hip_diff_destroy, hip_diff_set_coeffs, hip_ediff, hip_diff_apply, hip_last_error, &
    & tsx_abi_sizes, tsx_comm_peer_export, tsx_comm_peer_attach, tsx_comm_peer_selftest, tsx_comm_peer_disable, &
    & tsx_comm_peer_set_fences, tsx_comm_peer_reset, &
    & TSX_HOST, TSX_DEVICE, TSX_PC_NONE, TSX_PC_COLUMN, TSX_PC_ZEBRA, TSX_PC_REDBLACK

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
  subroutine hip_diff_set_coeffs(handle, diff2diff, l1d, a11, a12, albedo, ierr)
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
  subroutine hip_diff_apply(handle, x, y, ierr)
    type(c_ptr), intent(in) :: handle
    real(c_double), target, contiguous, intent(in) :: x(:, :, :, :)  ! (0:D-1, zs:ze, xs:xe, ys:ye)
    real(c_double), target, contiguous, intent(inout) :: y(:, :, :, :)
    integer(c_int), intent(out) :: ierr
    ierr = tsx_diff_apply(handle, c_loc(x), c_loc(y), TSX_HOST)
  end subroutine

  !> same contract as explicit_ediff(solver, prefix, vb, vediff, solution, ierr) (src/pprts_explicit.F90:461):
  !> vediff holds the initial guess on entry and the solution on exit; niter / residual history go to
  !> solution%Niter_diff / solution%diff_ksp_residual_history (src/pprts_base.F90:163-166).
  !> ierr = library error code, or -reason when the Krylov solver stopped with a negative KSP reason.
  subroutine hip_ediff(handle, vb, vediff, rtol, atol, maxit, pc, niter, res_hist, reason, ierr)
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
