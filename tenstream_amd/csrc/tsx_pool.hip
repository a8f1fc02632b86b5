// tsx_pool.hip -- libtsx's device memory: a per-device pool of driver allocations ("slabs") that are taken from the driver ONCE,
// held in quarantine until their contents are proven stable, and then sub-allocated for the lifetime of the process.
//
// Why (round 6; DESIGN.md section 10, profiles/r06/README.md).  The rare wrong result of rounds 4-5 -- 1-2 % of the fresh four-process
// runs of the sharded pipeline test -- was data LOST FROM FRESHLY ALLOCATED DEVICE MEMORY: kernels of tsx_dedup_from_coords wrote
// into blocks that hipMalloc had returned microseconds earlier (the build's scratch, the representatives `dd_ent_cell`), a later
// kernel on the same stream -- or a hipMemcpy -- read zeros.  scripts/fresh_loop.py caught it with the inputs of the reading kernel
// read back (`rep`, `pos` of the 11 KB scratch all zero 0.3 ms after a read-back of its last word had been right), every byte of the
// library's device code verified against the file (scripts/code_verify.py: intact), and a second launch of the same kernel giving the
// same result: memory that a process has just been given can be cleared by the platform AFTER the process has started to use it --
// seen only while other processes on the device start and exit (their released memory is wiped; new allocations are cleared).
// Nothing a kernel does can defend against that; what the library can do is never to compute in memory that is younger than the
// window in which this happens:
//   * a slab comes from hipMalloc, is filled with a pattern, and is verified (a kernel that counts words that are not the pattern)
//     again and again until it has stayed intact for TSX_POOL_GUARD_US (default 3000) microseconds; a verify that finds the pattern
//     damaged is counted (tsx_pool_stats) and restarts the clock; then the slab is zeroed;
//   * everything the solvers allocate (tsx_dev_malloc / tsx_dev_free, used throughout the library instead of hipMalloc / hipFree)
//     is a first-fit piece of a slab; freed pieces go back to the pool (coalesced), never to the driver: after the first coefficient
//     set no driver allocation lies on a solver's path at all (the reference allocates its coefficient arrays once per solver
//     too, alloc_coeff_diff2diff, src/pprts.F90:3396-3490) -- which also takes hipMalloc / hipFree off config 4's per-g-point path.
// tsx_dev_free keeps hipFree's contract: it synchronises the device first (callers free scratch that queued kernels still read).
#include <stdio.h>
#include <unistd.h>

#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "tsx_host.hpp"
#include "tsx_pool_map.hpp"

namespace {
constexpr unsigned kPattern = 0xA5C3F00Du;
constexpr size_t kAlign = TsxPieceMap::kAlign;

__global__ __launch_bounds__(256) void tsx_k_pool_fill(unsigned *p, size_t nwords, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) p[i] = v;
}
// how many words differ from v, and the offset of the lowest one
__global__ __launch_bounds__(256) void tsx_k_pool_verify(const unsigned *p, size_t nwords, unsigned v, unsigned long long *out) {
  unsigned long long bad = 0, first = ~0ull;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256)
    if (__builtin_nontemporal_load(p + i) != v) {
      bad++;
      if (i < first) first = i;
    }
  if (bad) {
    atomicAdd(&out[0], bad);
    atomicMin(&out[1], first);
  }
}

struct Pool {
  std::mutex mu;
  TsxPieceMap m;  // every byte of every slab belongs to exactly one piece (tsx_pool_map.hpp)
  unsigned long long *flag = nullptr;  // [2] device words of the verify kernel
  hipStream_t st = nullptr;            // the NULL stream, synchronised as a stream (never the device): see pool_tools
  long long wipes = 0, wiped_words = 0, first_wipe_us = -1, driver_allocs = 0, guard_us_spent = 0;
};
std::mutex g_mu;
std::map<int, Pool *> g_pools;

Pool *pool_of(int dev) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_pools.find(dev);
  if (it != g_pools.end()) return it->second;
  Pool *p = new Pool();
  g_pools[dev] = p;
  return p;
}
long long env_ll(const char *name, long long dflt) {
  const char *e = getenv(name);
  return e && *e ? atoll(e) : dflt;
}
int grid_of(size_t nwords) {
  size_t b = (nwords + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// The pool's kernels run on the NULL stream and only that stream is synchronised.
//  * It must NOT synchronise the device: a rank process of a multi-rank run allocates lazily while kernels of its solver are in flight
//    that wait for a neighbour rank's message (peer transport: in-kernel waits), and that neighbour may in turn wait for a kernel this
//    rank's host has yet to enqueue -- a hipDeviceSynchronize here closes the cycle (caught by the pytest loop of round 6: one
//    peer-transport all-reduce timed out in 231 runs of the four-process pipeline case,
//    profiles/r06/flaky_loop_pytest_pool_both_transports.txt; hipMalloc itself never waited).  The solvers' streams are non-blocking:
//    the null stream neither waits for them nor holds them up.
//  * It must NOT own a stream either: a stream is a hardware queue, eight rank processes sharing one device (the config-3 test) hold
//    four each already, and a fifth per process oversubscribed the queue slots -- the scheduler then multiplexes the queues with a
//    coarse quantum and the peer transport's in-kernel waits ran into their bound (tests/test_gpu_config3.py failed 3 of 3 with the
//    stream, passed with the same library before it and passes without it).
hipError_t pool_tools(Pool *P) {
  hipError_t e = hipSuccess;
  // the verify kernel's two result words: pinned host memory (nothing the platform clears behind our back)
  if (!P->flag && (e = hipHostMalloc((void **)&P->flag, 2 * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess) return e;
  return e;
}

// pattern, watch, zero: `p` (bytes) has just come from the driver
hipError_t quarantine(Pool *P, void *p, size_t bytes) {
  hipError_t e;
  const long long guard_us = env_ll("TSX_POOL_GUARD_US", 3000);
  const size_t nwords = bytes / 4;
  const auto t_alloc = std::chrono::steady_clock::now();
  auto us_since = [&](std::chrono::steady_clock::time_point t) {
    return (long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count();
  };
  if (guard_us > 0) {
    hipLaunchKernelGGL(tsx_k_pool_fill, dim3(grid_of(nwords)), dim3(256), 0, P->st, (unsigned *)p, nwords, kPattern);
    if ((e = hipStreamSynchronize(P->st)) != hipSuccess) return e;
    auto t_clean = std::chrono::steady_clock::now();  // since when the pattern is known to be whole
    for (int round = 0; round < 100000; ++round) {
      std::this_thread::sleep_for(std::chrono::microseconds(guard_us / 8 > 50 ? guard_us / 8 : 50));
      P->flag[0] = 0;
      P->flag[1] = ~0ull;
      hipLaunchKernelGGL(tsx_k_pool_verify, dim3(grid_of(nwords)), dim3(256), 0, P->st, (const unsigned *)p, nwords, kPattern, P->flag);
      if ((e = hipStreamSynchronize(P->st)) != hipSuccess) return e;
      if (P->flag[0]) {  // the platform wrote into memory it had already handed out: count it, write the pattern again, start over
        P->wipes++;
        P->wiped_words += (long long)P->flag[0];
        if (P->first_wipe_us < 0) P->first_wipe_us = us_since(t_alloc);
        if (getenv("TSX_POOL_VERBOSE"))
          fprintf(stderr, "[tsx_pool] pid %d: %llu words of a fresh %zu-byte allocation lost their contents %lld us after hipMalloc (first at word %llu)\n",
                  (int)getpid(), P->flag[0], bytes, us_since(t_alloc), P->flag[1]);
        hipLaunchKernelGGL(tsx_k_pool_fill, dim3(grid_of(nwords)), dim3(256), 0, P->st, (unsigned *)p, nwords, kPattern);
        if ((e = hipStreamSynchronize(P->st)) != hipSuccess) return e;
        t_clean = std::chrono::steady_clock::now();
        continue;
      }
      if (us_since(t_clean) >= guard_us) break;
    }
  }
  hipLaunchKernelGGL(tsx_k_pool_fill, dim3(grid_of(nwords)), dim3(256), 0, P->st, (unsigned *)p, nwords, 0u);
  if (bytes & 3) {  // (a tail shorter than a word)
    if ((e = hipMemsetAsync((char *)p + (bytes & ~(size_t)3), 0, bytes & 3, P->st)) != hipSuccess) return e;
  }
  if ((e = hipStreamSynchronize(P->st)) != hipSuccess) return e;
  P->guard_us_spent += us_since(t_alloc);
  return hipSuccess;
}

// one driver allocation, proven stable.  Not a hot path; synchronises the null stream only (never the device).
hipError_t new_slab(Pool *P, size_t bytes, char **out) {
  char *base = nullptr;
  hipError_t e = pool_tools(P);
  if (e != hipSuccess) return e;
  if ((e = hipMalloc((void **)&base, bytes)) != hipSuccess) return e;
  P->driver_allocs++;
  if ((e = quarantine(P, base, bytes)) != hipSuccess) return e;
  P->m.add_slab(base, bytes);
  *out = base;
  return hipSuccess;
}
}  // namespace

hipError_t tsx_dev_malloc_bytes(void **out, size_t bytes) {
  *out = nullptr;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (getenv("TSX_POOL") && atoi(getenv("TSX_POOL")) == 0) return hipMalloc(out, bytes);  // A/B: straight to the driver
  Pool *P = pool_of(dev);
  std::lock_guard<std::mutex> lk(P->mu);
  const size_t need = TsxPieceMap::rounded(bytes);
  if ((*out = P->m.take(need))) return hipSuccess;
  // a new slab: small requests share slabs of TSX_POOL_SLAB_MB (default 64), a large one gets a slab of its own size
  const size_t slab_min = (size_t)env_ll("TSX_POOL_SLAB_MB", 64) << 20;
  const size_t sz = need > slab_min ? need : slab_min;
  char *base = nullptr;
  e = new_slab(P, sz, &base);
  if (e != hipSuccess && sz > need) e = new_slab(P, need, &base);  // (a nearly full device: exactly what was asked for)
  if (e != hipSuccess) return e;
  *out = P->m.take(need);
  return *out ? hipSuccess : hipErrorOutOfMemory;
}

hipError_t tsx_dev_free(void *p) {
  if (!p) return hipSuccess;
  // hipFree's contract: everything queued on the device has finished before the memory can be handed out again
  hipError_t e = hipDeviceSynchronize();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
  Pool *P = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    // the piece may belong to another device's pool than the current device: look it up everywhere
    for (auto &kv : g_pools) {
      std::lock_guard<std::mutex> lk2(kv.second->mu);
      if (kv.second->m.owns((const char *)p)) {
        P = kv.second;
        break;
      }
    }
  }
  if (!P) return hipFree(p);  // not ours (TSX_POOL=0 allocations)
  std::lock_guard<std::mutex> lk(P->mu);
  if (!P->m.give((char *)p)) return hipErrorInvalidValue;
  return e;
}

// Make sure the pool holds a free piece of at least `bytes` (one slab, one quarantine) before a solver takes its dozens of large
// buffers one by one: without it every vector of a first solver is a slab of its own with its own 3 ms in quarantine (round 6: the
// first solve of a process 33 -> 64 ms wall, config 4's cold call 75 -> 65 g-points/s).  A hint: failure to get that much is not an
// error, the allocations then come as they did.
void tsx_dev_reserve(size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  if (getenv("TSX_POOL") && atoi(getenv("TSX_POOL")) == 0) return;
  Pool *P = pool_of(dev);
  std::lock_guard<std::mutex> lk(P->mu);
  const size_t need = (bytes + kAlign - 1) & ~(kAlign - 1);
  const size_t free_total = P->m.free_total();
  if (free_total >= need) return;  // (pieces of earlier solvers will serve; fragmentation costs a slab later, not correctness)
  char *base = nullptr;
  if (new_slab(P, need - free_total > ((size_t)64 << 20) ? need - free_total : ((size_t)64 << 20), &base) != hipSuccess) (void)hipGetLastError();
}

// A driver allocation that cannot come from the pool (the peer mailbox: uncached / fine-grained memory): the same quarantine in
// place.  `p` holds `bytes` bytes fresh from the driver; on return it is zeroed and has stayed intact for the guard time.
hipError_t tsx_dev_quarantine(void *p, size_t bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  Pool *P = pool_of(dev);
  std::lock_guard<std::mutex> lk(P->mu);
  if ((e = pool_tools(P)) != hipSuccess) return e;
  if (getenv("TSX_POOL") && atoi(getenv("TSX_POOL")) == 0) {
    if ((e = hipMemsetAsync(p, 0, bytes, P->st)) != hipSuccess) return e;
    return hipStreamSynchronize(P->st);
  }
  return quarantine(P, p, bytes);
}

// out8 = {slabs taken from the driver, their bytes, bytes handed out now, pieces, verifies that found a fresh slab damaged, words
// damaged, microseconds after hipMalloc of the first damaged verify (-1: none), microseconds spent in quarantine} for `device`
extern "C" int tsx_pool_stats(int device, int64_t *out8) {
  ARGCHK(out8, "tsx_pool_stats: null");
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return TSX_ERR_NO_DEVICE;
  Pool *P = pool_of(device);
  std::lock_guard<std::mutex> lk(P->mu);
  out8[0] = (int64_t)P->m.slabs.size();
  out8[1] = (int64_t)P->m.bytes;
  out8[2] = (int64_t)P->m.live;
  out8[3] = (int64_t)P->m.pieces.size();
  out8[4] = P->wipes;
  out8[5] = P->wiped_words;
  out8[6] = P->first_wipe_us;
  out8[7] = P->guard_us_spent;
  return TSX_OK;
}
