// calib_fetch.hip -- calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths libtsx
// uses (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 for 16 B/lane streams; other widths uncalibrated).
// Each kernel streams a 2 GiB buffer exactly once (>> 256 MiB Infinity Cache), so bytes are known.
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch scripts/calib_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T> __global__ __launch_bounds__(256) void k_read(const T* __restrict__ a, size_t n, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    T v = a[i];
    const float* f = reinterpret_cast<const float*>(&v);
    for (int q = 0; q < (int)(sizeof(T) / 4); ++q) s += f[q];
  }
  if (s == 123.456) out[0] = s;  // keep the loads alive
}
template <typename T> __global__ __launch_bounds__(256) void k_copy(const T* __restrict__ a, T* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// 100 fp32 planes read 8 B/lane + 10 fp64 planes read 16 B/lane, like tsx_k_spmv_w<...,CPT=2>
__global__ __launch_bounds__(256) void k_planes(const float2* __restrict__ c, const double2* __restrict__ x, size_t ncell2, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < ncell2; i += (size_t)gridDim.x * 256) {
#pragma unroll 10
    for (int p = 0; p < 100; ++p) { float2 v = c[(size_t)p * ncell2 + i]; s += v.x + v.y; }
#pragma unroll
    for (int p = 0; p < 10; ++p) { double2 v = x[(size_t)p * ncell2 + i]; s += v.x + v.y; }
  }
  if (s == 123.456) out[0] = s;
}
int main() {
  const size_t bytes = (size_t)2 << 30;
  void *a, *b; double* out;
  CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMalloc(&out, 8));
  CHK(hipMemset(a, 0, bytes)); CHK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float ms;
#define RUN(name, launch, nbytes) CHK(hipMemset(b, 1, bytes)); CHK(hipDeviceSynchronize()); CHK(hipEventRecord(e0)); launch; CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize()); \
  CHK(hipEventElapsedTime(&ms, e0, e1)); printf("%-14s %8.3f ms  %8.1f GB/s (bytes %.0f)\n", name, ms, (double)(nbytes) / ms / 1e6, (double)(nbytes));
  for (int grid : {2048, 4096, 16384}) {
    printf("grid %d\n", grid);
    RUN("read4",  (k_read<float><<<grid, 256>>>((const float*)a, bytes / 4, out)), bytes);
    RUN("read8",  (k_read<float2><<<grid, 256>>>((const float2*)a, bytes / 8, out)), bytes);
    RUN("read16", (k_read<float4><<<grid, 256>>>((const float4*)a, bytes / 16, out)), bytes);
    RUN("copy8",  (k_copy<float2><<<grid, 256>>>((const float2*)a, (float2*)b, bytes / 8)), 2 * bytes);
    RUN("copy16", (k_copy<float4><<<grid, 256>>>((const float4*)a, (float4*)b, bytes / 16)), 2 * bytes);
  }
  // planes: 100*8 + 10*16 = 960 B per cell pair -> ncell2 = bytes/960 (a holds both c and x regions)
  size_t ncell2 = bytes / 960;
  RUN("planes", (k_planes<<<4096, 256>>>((const float2*)a, (const double2*)((char*)a + ncell2 * 800), ncell2, out)), ncell2 * 960);
  return 0;
}
