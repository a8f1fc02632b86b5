/* mpiexec -n N ./f2c_mpi_selftest nxp nyp -- the two collectives libtsx_f2c_mpi.so runs over `fcomm` = MPI_Comm_c2f(comm)
 * (c_wrapper/f2c_pprts.F90:130-230 takes the communicator that way), on host buffers: no GPU needed.  Exit code 0 = every face
 * of every rank holds what its neighbour sent through the opposite face and the sums over the ranks are right. */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>

int tsx_f2c_mpi_selftest(int fcomm, int nxp, int nyp, int count);

int main(int argc, char **argv) {
  MPI_Init(&argc, &argv);
  int rank, size;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  const int nxp = argc > 1 ? atoi(argv[1]) : size, nyp = argc > 2 ? atoi(argv[2]) : 1;
  const int bad = tsx_f2c_mpi_selftest((int)MPI_Comm_c2f(MPI_COMM_WORLD), nxp, nyp, 1000);
  if (rank == 0) printf("f2c_mpi_selftest: %d ranks (%d x %d): %d wrong values\n", size, nxp, nyp, bad);
  MPI_Finalize();
  return bad == 0 ? 0 : 1;
}
