// Stand-alone probe 2 (round 6; no libtsx): does freshly allocated device memory keep what a kernel has written into it while other
// processes on the device start and exit?  Each process: a run of hipMallocs of mixed sizes, every block filled with a pattern at
// once by a kernel on a non-default stream; then for `watch_ms` milliseconds a kernel re-checks every block again and again and the
// first damage is reported with its age; before exit the process dirties `exit_mb` MB that its teardown hands back to the driver
// (released memory is wiped by the platform -- the load the next processes' fresh allocations meet).
//   hipcc --offload-arch=gfx950 -O2 -o fresh_alloc_probe2 fresh_alloc_probe2.hip ; run by scripts/fresh_alloc_loop2.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <chrono>
#include <vector>
__global__ void k_fill(unsigned *p, size_t n, unsigned tag) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ (unsigned)(i * 2654435761u);
}
__global__ void k_check(const unsigned *p, size_t n, unsigned tag, unsigned long long *out) {
  unsigned long long bad = 0, first = ~0ull, zero = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned v = __builtin_nontemporal_load(p + i);
    if (v != (tag ^ (unsigned)(i * 2654435761u))) {
      bad++;
      zero += v == 0;
      if (i < first) first = i;
    }
  }
  if (bad) {
    atomicAdd(&out[0], bad);
    atomicMin(&out[1], first);
    atomicAdd(&out[2], zero);
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)
int main(int argc, char **argv) {
  const int watch_ms = argc > 1 ? atoi(argv[1]) : 20;
  const int exit_mb = argc > 2 ? atoi(argv[2]) : 2048;
  hipStream_t st;
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long *flag;
  CK(hipHostMalloc((void **)&flag, 3 * sizeof(unsigned long long), hipHostMallocDefault));
  const size_t sizes[] = {64, 4096, 65536, 1 << 20, (2 << 20) + 4096, 8 << 20, 64 << 20, 256 << 20};  // bytes
  std::vector<unsigned *> ptr;
  std::vector<size_t> len;
  std::vector<std::chrono::steady_clock::time_point> born;
  for (int r = 0; r < 3; ++r)
    for (size_t sz : sizes) {
      unsigned *p = nullptr;
      CK(hipMalloc((void **)&p, sz));
      const size_t n = sz / 4;
      hipLaunchKernelGGL(k_fill, dim3(n > 65536 ? 1024 : 1), dim3(256), 0, st, p, n, 0x9e3779b9u + (unsigned)ptr.size());
      ptr.push_back(p);
      len.push_back(n);
      born.push_back(std::chrono::steady_clock::now());
    }
  CK(hipStreamSynchronize(st));
  int bad = 0;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<char> dead(ptr.size(), 0);
  while (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() < watch_ms) {
    for (size_t b = 0; b < ptr.size(); ++b) {
      if (dead[b]) continue;
      flag[0] = 0; flag[1] = ~0ull; flag[2] = 0;
      hipLaunchKernelGGL(k_check, dim3(len[b] > 65536 ? 1024 : 1), dim3(256), 0, st, ptr[b], len[b], 0x9e3779b9u + (unsigned)b, flag);
      CK(hipStreamSynchronize(st));
      if (flag[0]) {
        const long long age = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - born[b]).count();
        printf("LOST pid %d block %zu (%zu bytes at %p): %llu words damaged (%llu of them zero), first at word %llu, seen %lld us after its hipMalloc\n",
               (int)getpid(), b, len[b] * 4, (void *)ptr[b], flag[0], flag[2], flag[1], age);
        dead[b] = 1;
        bad++;
      }
    }
  }
  if (exit_mb > 0) {  // dirty memory for the teardown to release
    unsigned *big = nullptr;
    if (hipMalloc((void **)&big, (size_t)exit_mb << 20) == hipSuccess) {
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, big, ((size_t)exit_mb << 20) / 4, 0x12345678u);
      CK(hipStreamSynchronize(st));
    }
  }
  fflush(stdout);
  _exit(bad ? 1 : 0);  // no hipFree, no runtime teardown of ours: the driver releases everything
}
