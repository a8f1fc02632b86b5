// Self-test of the guard zones of poison_malloc.c: a deliberate overrun of 100 bytes must be reported at hipFree.
//   hipcc -o guard_selftest guard_selftest.cpp && LD_PRELOAD=./libpoison_malloc.so TSX_GUARD=1 ./guard_selftest
#include <hip/hip_runtime.h>
#include <cstdio>
int main() {
  char *p = nullptr;
  if (hipMalloc((void **)&p, 1000) != hipSuccess) return 1;
  (void)hipMemset(p, 0, 1100);
  (void)hipDeviceSynchronize();
  (void)hipFree(p);
  (void)hipMemset(p + 10, 1, 4);  // use after free (TSX_QUARANTINE=1 reports it at the next hipFree)
  char *q = nullptr;
  if (hipMalloc((void **)&q, 64) == hipSuccess) (void)hipFree(q);
  std::puts("guard_selftest: done (a TSX_GUARD line above means the shim works)");
  return 0;
}
