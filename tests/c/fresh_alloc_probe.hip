// Stand-alone probe (round 6; no libtsx): are stores into memory that was hipMalloc'ed microseconds earlier ever lost?
// Mirrors what tsx_dedup_from_coords did in rounds 4-5: a run of small hipMallocs, each followed at once by a kernel on a
// non-default stream that stores a pattern into the fresh block, one synchronisation at the end, everything read back.
// Run as 4 concurrent fresh processes in a loop (scripts/fresh_alloc_loop.sh); exit code 1 and a line on stdout on any lost store.
//   hipcc --offload-arch=gfx950 -O2 -o fresh_alloc_probe fresh_alloc_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <vector>
__global__ void k_fill(int *p, int n, int tag) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = tag * 1000003 + i + 1;
}
// the conditional store of tsx_k_dd_index: a few lanes of a wave write, keyed on loaded values
__global__ void k_cond(const int *rep, const int *pos, int n, int *idx, int *ent) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    const int r = rep[c], id = pos[r];
    idx[c] = id;
    if (r == c) ent[id] = c + 1;
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)
int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 8;
  hipStream_t st;
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int sizes[] = {4, 80, 480, 800, 1920, 3200, 65536, 262144, 1 << 20};
  std::vector<int *> ptr;
  std::vector<int> len;
  int bad = 0;
  for (int r = 0; r < rounds; ++r)
    for (int sz : sizes) {
      int *p = nullptr;
      CK(hipMalloc((void **)&p, sizeof(int) * sz));
      hipLaunchKernelGGL(k_fill, dim3(sz > 4096 ? 64 : 1), dim3(256), 0, st, p, sz, (int)ptr.size());
      ptr.push_back(p);
      len.push_back(sz);
    }
  // the index pattern: 240 cells, 4 groups
  const int n = 240;
  std::vector<int> rep(n), pos(n, 0);
  const int heads[4] = {0, 30, 84, 114};
  for (int c = 0; c < n; ++c) rep[c] = c < 30 ? 0 : c < 84 ? 30 : c < 114 ? 84 : 114;
  for (int q = 0; q < 4; ++q) pos[heads[q]] = q;
  int *drep, *dpos, *didx, *dent;
  CK(hipMalloc((void **)&drep, sizeof(int) * n));
  CK(hipMalloc((void **)&dpos, sizeof(int) * n));
  CK(hipMemcpyAsync(drep, rep.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
  CK(hipMemcpyAsync(dpos, pos.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
  CK(hipStreamSynchronize(st));
  CK(hipMalloc((void **)&didx, sizeof(int) * n));
  CK(hipMalloc((void **)&dent, sizeof(int) * 4));
  hipLaunchKernelGGL(k_cond, dim3(1), dim3(256), 0, st, drep, dpos, n, didx, dent);
  CK(hipStreamSynchronize(st));
  int ent[4] = {-1, -1, -1, -1};
  CK(hipMemcpy(ent, dent, sizeof(ent), hipMemcpyDeviceToHost));
  for (int q = 0; q < 4; ++q)
    if (ent[q] != heads[q] + 1) {
      printf("LOST conditional store: ent[%d] = %d (want %d) pid %d\n", q, ent[q], heads[q] + 1, (int)getpid());
      bad++;
    }
  std::vector<int> host;
  for (size_t b = 0; b < ptr.size(); ++b) {
    host.assign(len[b], 0);
    CK(hipMemcpy(host.data(), ptr[b], sizeof(int) * len[b], hipMemcpyDeviceToHost));
    int first = -1, cnt = 0;
    for (int i = 0; i < len[b]; ++i)
      if (host[i] != (int)b * 1000003 + i + 1) {
        if (first < 0) first = i;
        cnt++;
      }
    if (cnt) {
      printf("LOST block %zu (%d ints at %p): %d wrong from %d, holds %d pid %d\n", b, len[b], (void *)ptr[b], cnt, first, host[first], (int)getpid());
      bad++;
    }
  }
  return bad ? 1 : 0;
}
