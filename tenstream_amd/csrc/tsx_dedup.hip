// tsx_dedup.hip -- shared storage of bit-identical transport blocks.
//
// The operator's 100 (3_10) / 256 (8_16) coefficients per cell are 400 of the 563 B a cell costs an operator apply and
// 128 of the 200 B it costs a preconditioner pass.  Cells with *bit-identical* blocks -- every clear-sky cell of a
// horizontally homogeneous background has the same (tau, w0, aspect, g) and therefore the same interpolated block; the
// reference's own examples build their atmospheres that way (examples/pprts/pprts_ex1.F90:85-102,
// examples/pprts/pprts_rrtm_lw_sw.F90) -- can share one stored copy behind a per-cell index.  Lossless: the operator
// computes with the same numbers in the same order.  Built from the dense planes (whatever produced them), so it serves
// tsx_diff_set_coeffs and the LUT path alike:
//   1. hash every cell's block (64 bit);  2. open-addressing table keyed by the hash, owner = smallest cell index;
//   3. every cell compares its block with its owner's bit for bit -- a hash collision just keeps the cell unique;
//   4. exclusive scan over "I am a representative" -> entry ids in cell order (neighbouring unique cells get neighbouring
//      entries: their loads stay coalesced);  5. compact planes Cd[q * Nent + id].
// Cells of 1-D layers (their planes are never read: a11 / a12 are) all map to one entry.
// Used only when it pays (Nent <= Nc / 2); the dense planes stay (setup_b thermal, flux divergence, export read them).
#include <stdio.h>
#include <unistd.h>

#include "tsx_host.hpp"
#include "tsx_lut_dev.hpp"


__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_hash(TsxGeo g, int DD, const float *__restrict__ C,
                                                           const uint8_t *__restrict__ l1d, unsigned long long *__restrict__ h) {
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int k = (int)(c / g.ncol);
    unsigned long long v = TSX_DD_SEED;
    if (l1d[k]) {
      v = TSX_DD_H1D;
    } else {
      for (int q = 0; q < DD; ++q) v = tsx_dd_hash_step(v, q, C[(size_t)q * Nc + c]);
      v = tsx_dd_hash_final(v);
    }
    h[c] = v;
  }
}

// tsx_k_dd_insert elects one lane per distinct hash with 64-bit ballots and width-64 shuffles: gfx950 runs wave64 (the only
// target this library is built for); on a device that reports another wave size the shared storage is simply not used
static bool tsx_dd_wave64(const tsx_solver *s) {
  int ws = 0;
  if (hipDeviceGetAttribute(&ws, hipDeviceAttributeWarpSize, s->device) != hipSuccess) return false;
  return ws == 64;
}

__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_insert(long long Nc, unsigned long long mask, const unsigned long long *__restrict__ h,
                                                             unsigned long long *__restrict__ keys, int *__restrict__ owner) {
  // Most cells of a homogeneous background carry the same hash: millions of atomics on one slot would serialise (measured:
  // 40 ms).  A cell therefore looks first (relaxed loads): once the key is in place and the slot's owner is already a
  // smaller cell index, it has nothing to add.  Cells are visited in ascending order within a wave and roughly so across
  // the grid, so after the first arrivals almost nobody issues an atomic.
  // ... and of a run of neighbouring lanes with the same hash only the first one (the smallest cell index) goes to the table.  The
  // heads of all runs of a wave probe at the same time: electing one lane per distinct hash in a loop (rounds 1-4) put a wave of 64
  // different cloudy cells through 64 rounds of dependent loads / atomics, one after the other -- 194 us for 4.2 M cells, a third
  // of what a g-point's sharing pipeline cost.  Two runs of one wave with the same hash both go: the atomicMin sorts them out.
  for (long long c0 = (long long)blockIdx.x * TSX_BLOCK; c0 < Nc; c0 += (long long)gridDim.x * TSX_BLOCK) {
    const long long c = c0 + threadIdx.x;
    const bool live = c < Nc;
    const unsigned long long hv = live ? h[c] : TSX_DD_EMPTY;
    const unsigned plo = (unsigned)__shfl_up((int)(unsigned)(hv & 0xffffffffull), 1, 64);
    const unsigned phi = (unsigned)__shfl_up((int)(unsigned)(hv >> 32), 1, 64);
    const bool head = live && ((threadIdx.x & 63) == 0 || ((((unsigned long long)phi << 32) | plo) != hv));
    if (head) {
      unsigned long long slot = hv & mask;
      for (;;) {
        unsigned long long cur = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == TSX_DD_EMPTY) cur = atomicCAS(&keys[slot], TSX_DD_EMPTY, hv);
        if (cur == TSX_DD_EMPTY || cur == hv) {
          if (__hip_atomic_load(&owner[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (int)c) atomicMin(&owner[slot], (int)c);
          break;
        }
        slot = (slot + 1) & mask;
      }
    }
  }
}

// rep[c] = the representative cell of c's block; flag[c] = 1 where c represents itself
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_resolve(TsxGeo g, int DD, const float *__restrict__ C,
                                                              const uint8_t *__restrict__ l1d, unsigned long long mask,
                                                              const unsigned long long *__restrict__ h,
                                                              const unsigned long long *__restrict__ keys,
                                                              const int *__restrict__ owner, int *__restrict__ rep,
                                                              int *__restrict__ flag, const float4 *__restrict__ samp) {
  // samp (nullable): the LUT coordinates the blocks were interpolated from (tsx_k_cell_samples).  The interpolation is a
  // deterministic function of them, so two cells with bit-identical coordinates have bit-identical blocks and the 2 x 400-byte
  // comparison is spared (the whole clear-sky background); different coordinates are compared block by block as ever
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const unsigned long long hv = h[c];
    unsigned long long slot = hv & mask;
    while (keys[slot] != hv) slot = (slot + 1) & mask;
    const int o = owner[slot];
    bool same = true;
    if (o != (int)c && !(l1d[(int)(c / g.ncol)] && l1d[o / g.ncol])) {
      bool coords = false;
      if (samp) {
        const float4 a = samp[c], b = samp[o];
        coords = __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
                 __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
      }
      if (!coords)
        for (int q = 0; q < DD; ++q)
          same &= __float_as_uint(C[(size_t)q * Nc + c]) == __float_as_uint(C[(size_t)q * Nc + o]);
    }
    const int r = same ? o : (int)c;
    rep[c] = r;
    flag[c] = r == (int)c;
  }
}

// ---- exclusive scan of int flags (three launches: per-block sums, scan of the sums by one block, write-out)
constexpr int TSX_SCAN_CHUNK = 2048;  // elements per block
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_scan_sums(long long n, const int *__restrict__ f, int *__restrict__ sums) {
  __shared__ int sm[TSX_BLOCK];
  const long long b0 = (long long)blockIdx.x * TSX_SCAN_CHUNK;
  int t = 0;
  for (int q = threadIdx.x; q < TSX_SCAN_CHUNK; q += TSX_BLOCK)
    if (b0 + q < n) t += f[b0 + q];
  sm[threadIdx.x] = t;
  __syncthreads();
  for (int s2 = TSX_BLOCK / 2; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) sm[threadIdx.x] += sm[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(1024) void tsx_k_scan_top(int nb, int *__restrict__ sums, int *__restrict__ total) {
  // serial over chunks of 1024 block sums, Hillis-Steele inside a chunk; nb <= a few 10^4
  __shared__ int sm[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int q = b0 + threadIdx.x;
    const int v = q < nb ? sums[q] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int a = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
      __syncthreads();
      sm[threadIdx.x] += a;
      __syncthreads();
    }
    if (q < nb) sums[q] = carry + sm[threadIdx.x] - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == 1023) carry += sm[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_scan_write(long long n, const int *__restrict__ f, const int *__restrict__ sums,
                                                              int *__restrict__ pos) {
  // one wave-serial pass per block: chunk of 2048, thread t owns 8 consecutive elements
  __shared__ int sm[TSX_BLOCK];
  const long long b0 = (long long)blockIdx.x * TSX_SCAN_CHUNK;
  constexpr int PER = TSX_SCAN_CHUNK / TSX_BLOCK;
  int loc[PER], t = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const long long e = b0 + (long long)threadIdx.x * PER + q;
    loc[q] = e < n ? f[e] : 0;
    t += loc[q];
  }
  sm[threadIdx.x] = t;
  __syncthreads();
  for (int off = 1; off < TSX_BLOCK; off <<= 1) {
    const int a = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] += a;
    __syncthreads();
  }
  int run = sums[blockIdx.x] + sm[threadIdx.x] - t;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const long long e = b0 + (long long)threadIdx.x * PER + q;
    if (e < n) pos[e] = run;
    run += loc[q];
  }
}

// cidx[c] = entry of c's block (natural and colour-split order); ent_cell[id] = the representative cell
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_index(TsxGeo g, const int *__restrict__ rep, const int *__restrict__ pos,
                                                            int *__restrict__ cidx, int *__restrict__ cidx_split,
                                                            int *__restrict__ ent_cell) {
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int r = rep[c], id = pos[r];
    if (cidx) cidx[c] = id;
    if (cidx_split) {
      const int i = (int)(c % g.xm);
      const long long t = c / g.xm;
      const int j = (int)(t % g.ym), k = (int)(t / g.ym);
      cidx_split[(size_t)k * g.ncol + tsx_split_col(i, j, g.xm)] = id;
    }
    if (r == (int)c) ent_cell[id] = (int)c;
  }
}
// Cd: plane-major [D*D][nent] (what the preconditioner's pack kernels read); Ce: entry-major [nent][D*D] (the operator).
// A workgroup moves a tile of 32 entries: plane by plane the lanes read 32 consecutive entries (their representative cells
// ascend, so clouds read nearly contiguously) and write Cd; the tile goes through LDS and leaves as one contiguous run of Ce
// (writing Ce straight from the plane-major loop scattered 4-byte stores 4 * D*D bytes apart: 0.57 ms instead of 0.07).
template <int DD>
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_compact(long long Nc, long long nent, const float *__restrict__ C,
                                                              const int *__restrict__ ent_cell, float *__restrict__ Cd,
                                                              float *__restrict__ Ce) {
  constexpr int T = 32;
  __shared__ float sm[T][DD + 1];
  const int lane = threadIdx.x % T, grp = threadIdx.x / T;  // 8 groups of 32 lanes walk the planes
  const long long ntile = (nent + T - 1) / T;
  for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long long id = tile * T + lane;
    const bool live = id < nent;
    const long long cell = live ? (long long)ent_cell[id] : 0;
    for (int pl = grp; pl < DD; pl += TSX_BLOCK / T) {
      const float v = live ? C[(size_t)pl * Nc + cell] : 0.0f;
      if (live) Cd[(size_t)pl * nent + id] = v;
      sm[lane][pl] = v;
    }
    __syncthreads();
    const long long n = (nent - tile * T < T ? nent - tile * T : T) * DD;
    float *out = Ce + (size_t)tile * T * DD;
    for (long long q = threadIdx.x; q < n; q += TSX_BLOCK) out[q] = sm[q / DD][q % DD];
    __syncthreads();
  }
}

// colsum[s][e] = sum over dst d of c(src s -> dst d) of distinct block e, accumulated in double in the order d = 0 .. D-1 exactly
// like the per-cell loops of tsx_k_setup_b_thermal / tsx_k_flx_div it replaces (1 - colsum = what a stream loses to absorption)
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_colsum(int D, long long nent, const float *__restrict__ Cd,
                                                             double *__restrict__ out) {
  for (long long e = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; e < nent; e += (long long)gridDim.x * TSX_BLOCK)
    for (int sI = 0; sI < D; ++sI) {
      double sum = 0.0;
      for (int d = 0; d < D; ++d) sum += (double)Cd[(size_t)(d * D + sI) * nent + e];
      out[(size_t)sI * nent + e] = sum;
    }
}

// capacity for `nent` entries: a quarter more (sharing is only used while nent <= Nc / 2, so Nc / 2 is the ceiling)
static inline int dd_grow(long long nent, long long Nc) {
  long long cap = nent + nent / 4 + 16;
  const long long top = Nc / 2 + 16;
  if (cap > top) cap = top > nent ? top : nent;
  return (int)cap;
}

struct TsxDdSlice {  // a piece of the solver's scratch allocation
  void *p;
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

// scratch kept with the solver (one allocation; hipMalloc / hipFree per g-point cost more than the kernels): hashes, table
// keys, table owners, representative, flag, position, per-chunk sums, total
struct TsxDdScratch {
  TsxDdSlice th, tk, to, trep, tflag, tpos, tsum, ttot;
  unsigned long long tsz;
  int nsb;
};
static int dd_scratch(tsx_solver *s, long long Nc, TsxDdScratch *w) {
  unsigned long long tsz = 1;
  while (tsz < (unsigned long long)(2 * Nc)) tsz <<= 1;
  const int nsb = (int)((Nc + TSX_SCAN_CHUNK - 1) / TSX_SCAN_CHUNK);
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t sz[8] = {al(sizeof(unsigned long long) * (size_t)Nc), al(sizeof(unsigned long long) * (size_t)tsz),
                        al(sizeof(int) * (size_t)tsz), al(sizeof(int) * (size_t)Nc), al(sizeof(int) * (size_t)Nc),
                        al(sizeof(int) * (size_t)Nc), al(sizeof(int) * (size_t)nsb), 256};
  size_t tot = 0;
  for (size_t v : sz) tot += v;
  if (s->dd_scratch_bytes < tot) {
    if (s->dd_scratch) HIPCHK(tsx_dev_free(s->dd_scratch));
    s->dd_scratch = nullptr;
    HIPCHK(tsx_dev_malloc(&s->dd_scratch, tot));
    s->dd_scratch_bytes = tot;
  }
  TsxDdSlice *d[8] = {&w->th, &w->tk, &w->to, &w->trep, &w->tflag, &w->tpos, &w->tsum, &w->ttot};
  size_t off = 0;
  for (int q = 0; q < 8; ++q) {
    d[q]->p = (char *)s->dd_scratch + off;
    off += sz[q];
  }
  w->tsz = tsz;
  w->nsb = nsb;
  return TSX_OK;
}

static bool dedup_enabled();
// where a kernel that produces the blocks can leave their hashes (same function as tsx_k_dd_hash): null if sharing is off
int tsx_dedup_hash_buffer(tsx_solver *s, unsigned long long **h) {
  *h = nullptr;
  s->dd_hash_ready = false;
  if (!dedup_enabled() || s->geo.Nc >= (1ll << 31)) return TSX_OK;
  TsxDdScratch w;
  int rc = dd_scratch(s, s->geo.Nc, &w);
  if (rc) return rc;
  *h = w.th.as<unsigned long long>();
  return TSX_OK;
}

static bool dedup_enabled() {
  const char *e = getenv("TSX_DEDUP");  // TSX_DEDUP=0: always the dense planes (A/B knob)
  return e ? atoi(e) != 0 : true;
}

// ---- NEAR-identical blocks, for the preconditioner only (round 3).  Where every cell has a block of its own (an LES humidity
// field: tau and w0 differ from cell to cell) nothing is bit-identical, but the blocks still come from a smooth 4-parameter
// family (tau, w0, aspect, g), and the preconditioner is an approximation anyway: its couplings are stored in fp8 (2^-4) and
// fp16.  So cells whose blocks agree to about 1 % share one stored copy of the PRECONDITIONER's per-block records (the column
// recurrences, record 0, stay exact and per cell; the operator keeps every cell's exact block): a pass then reads 68 B per cell
// instead of 196 B.  Key: five characteristic coefficients (vertical transmission and reflection, top -> side, side straight
// through, side x -> y) in logarithmic bins of 0.7 %; owner = the smallest cell of a bin; a cell joins its owner only if ALL
// D*D coefficients agree to 1.2 % + 3e-4 (else it keeps an entry of its own) -- the 6 mantissa bits the side -> top couplings
// need (tests/studies/quant_study.py) are kept.
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_hash_near(TsxGeo g, int D, const float *__restrict__ C,
                                                                const uint8_t *__restrict__ l1d, unsigned long long *__restrict__ h) {
  const long long Nc = g.Nc;
  const int ns = g.ntop;  // first side dof
  // (dst, src) of the five features
  const int fd[5] = {0, 0, ns, ns + 1, ns + 5}, fs[5] = {0, 1, 0, ns + 1, ns + 1};
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int k = (int)(c / g.ncol);
    unsigned long long v = TSX_DD_SEED;
    if (l1d[k]) {
      v = TSX_DD_H1D;
    } else {
      for (int q = 0; q < 5; ++q) {
        const float x = C[(size_t)(fd[q] * D + fs[q]) * Nc + c];
        const int bin = x > 1e-7f ? (int)floorf(__logf(x) * (1.0f / 0.00698f)) : -100000;  // ln(1.007)
        v = tsx_dd_hash_step(v, q, __int_as_float(bin));
      }
      v = tsx_dd_hash_final(v);
    }
    h[c] = v;
  }
}
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_resolve_near(TsxGeo g, int DD, const float *__restrict__ C,
                                                                   const uint8_t *__restrict__ l1d, unsigned long long mask,
                                                                   const unsigned long long *__restrict__ h,
                                                                   const unsigned long long *__restrict__ keys,
                                                                   const int *__restrict__ owner, int *__restrict__ rep,
                                                                   int *__restrict__ flag) {
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const unsigned long long hv = h[c];
    unsigned long long slot = hv & mask;
    while (keys[slot] != hv) slot = (slot + 1) & mask;
    const int o = owner[slot];
    bool same = true;
    if (o != (int)c && !(l1d[(int)(c / g.ncol)] && l1d[o / g.ncol])) {
      for (int q = 0; q < DD; ++q) {
        const float a = C[(size_t)q * Nc + c], b = C[(size_t)q * Nc + o];
        same &= fabsf(a - b) <= 0.012f * fmaxf(fabsf(a), fabsf(b)) + 3e-4f;
      }
    }
    const int r = same ? o : (int)c;
    rep[c] = r;
    flag[c] = r == (int)c;
  }
}
static bool dedup_near_enabled() {
  const char *e = getenv("TSX_DEDUP_NEAR");  // TSX_DEDUP_NEAR=0: only bit-identical blocks are shared
  return e ? atoi(e) != 0 : true;
}

// one build: near = the approximate grouping (preconditioner only), else bit-identical blocks.  *pays: at most half of the
// cells need an entry of their own -- only then are the index and the compact copies made
static int dd_build(tsx_solver *s, bool near, bool *pays) {
  *pays = false;
  const TsxGeo &g = s->geo;
  const int DD = g.D * g.D;
  const long long Nc = g.Nc;
  TsxDdScratch w;
  {
    int rc = dd_scratch(s, Nc, &w);
    if (rc) return rc;
  }
  const unsigned long long tsz = w.tsz;
  const int nsb = w.nsb;
  TsxDdSlice &th = w.th, &tk = w.tk, &to = w.to, &trep = w.trep, &tflag = w.tflag, &tpos = w.tpos, &tsum = w.tsum, &ttot = w.ttot;
  HIPCHK(hipMemsetAsync(tk.p, 0, sizeof(unsigned long long) * (size_t)tsz, s->stream));
  HIPCHK(hipMemsetAsync(to.p, 0x7f, sizeof(int) * (size_t)tsz, s->stream));
  const float *C = (const float *)s->coef;
  const int nb = grid_for(Nc, 8192);
  if (near)
    hipLaunchKernelGGL(tsx_k_dd_hash_near, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g, g.D, C, s->l1d, th.as<unsigned long long>());
  else if (!s->dd_hash_ready)  // else tsx_k_lut_diff2diff has left the hashes of the blocks it produced (tsx_dedup_hash_buffer)
    hipLaunchKernelGGL(tsx_k_dd_hash, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g, DD, C, s->l1d, th.as<unsigned long long>());
  // the blocks came from tsx_k_lut_diff2diff a moment ago: the cells' LUT coordinates are still there (tsx_cell_samples)
  const float4 *samp = (!near && s->dd_hash_ready) ? (const float4 *)s->cell_samp : (const float4 *)nullptr;
  s->dd_hash_ready = false;
  hipLaunchKernelGGL(tsx_k_dd_insert, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, Nc, tsz - 1, th.as<unsigned long long>(),
                     tk.as<unsigned long long>(), to.as<int>());
  if (near)
    hipLaunchKernelGGL(tsx_k_dd_resolve_near, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g, DD, C, s->l1d, tsz - 1,
                       th.as<unsigned long long>(), tk.as<unsigned long long>(), to.as<int>(), trep.as<int>(), tflag.as<int>());
  else
    hipLaunchKernelGGL(tsx_k_dd_resolve, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g, DD, C, s->l1d, tsz - 1,
                       th.as<unsigned long long>(), tk.as<unsigned long long>(), to.as<int>(), trep.as<int>(), tflag.as<int>(), samp);
  hipLaunchKernelGGL(tsx_k_scan_sums, dim3(nsb), dim3(TSX_BLOCK), 0, s->stream, Nc, tflag.as<int>(), tsum.as<int>());
  hipLaunchKernelGGL(tsx_k_scan_top, dim3(1), dim3(1024), 0, s->stream, nsb, tsum.as<int>(), ttot.as<int>());
  hipLaunchKernelGGL(tsx_k_scan_write, dim3(nsb), dim3(TSX_BLOCK), 0, s->stream, Nc, tflag.as<int>(), tsum.as<int>(), tpos.as<int>());
  HIPCHK(hipGetLastError());
  int nent = 0;
  HIPCHK(hipMemcpyAsync(&nent, ttot.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  if (near) s->dd_nent_near = nent;
  else s->dd_nent = nent;
  if ((long long)nent * 2 > Nc) return TSX_OK;  // mostly unique blocks: the dense planes are the better layout
  // where the result goes: the operator's + preconditioner's arrays (bit-identical), or the preconditioner's own (near)
  int **cidx_split = near ? &s->pcn_cidx_split : &s->dd_cidx_split, **ent_cell = near ? &s->pcn_ent_cell : &s->dd_ent_cell;
  float **coef = near ? &s->pcn_coef : &s->dd_coef;
  int *cap = near ? &s->pc_cap : &s->dd_cap;
  if (!near && !s->dd_cidx) HIPCHK(tsx_dev_malloc((void **)&s->dd_cidx, sizeof(int) * (size_t)Nc));
  if (!*cidx_split) HIPCHK(tsx_dev_malloc((void **)cidx_split, sizeof(int) * (size_t)Nc));
  if (*cap < nent) {
    if (*coef) HIPCHK(tsx_dev_free(*coef));
    if (*ent_cell) HIPCHK(tsx_dev_free(*ent_cell));
    *coef = nullptr;
    *ent_cell = nullptr;
    const int ncap = dd_grow(nent, Nc);
    HIPCHK(tsx_dev_malloc((void **)coef, sizeof(float) * (size_t)DD * ncap * 2));  // plane-major, then entry-major
    HIPCHK(tsx_dev_malloc((void **)ent_cell, sizeof(int) * (size_t)ncap));
    *cap = ncap;
  }
  const bool split = g.xm % 2 == 0;
  hipLaunchKernelGGL(tsx_k_dd_index, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g, trep.as<int>(), tpos.as<int>(),
                     near ? (int *)nullptr : s->dd_cidx, split ? *cidx_split : (int *)nullptr, *ent_cell);
  {
    const int nbt = (int)(((long long)nent + 31) / 32 < 65536 ? ((long long)nent + 31) / 32 : 65536);
    float *Cd = *coef, *Ce = *coef + (size_t)DD * *cap;
    if (DD == 100)
      hipLaunchKernelGGL(tsx_k_dd_compact<100>, dim3(nbt > 0 ? nbt : 1), dim3(TSX_BLOCK), 0, s->stream, Nc, (long long)nent, C,
                         *ent_cell, Cd, Ce);
    else
      hipLaunchKernelGGL(tsx_k_dd_compact<256>, dim3(nbt > 0 ? nbt : 1), dim3(TSX_BLOCK), 0, s->stream, Nc, (long long)nent, C,
                         *ent_cell, Cd, Ce);
  }
  if (!near) {
    s->dd_coef_e = s->dd_coef + (size_t)DD * s->dd_cap;
    if (s->dd_colsum_cap < (long long)g.D * nent) {
      if (s->dd_colsum) HIPCHK(tsx_dev_free(s->dd_colsum));
      s->dd_colsum = nullptr;
      HIPCHK(tsx_dev_malloc((void **)&s->dd_colsum, sizeof(double) * (size_t)g.D * dd_grow(nent, Nc)));
      s->dd_colsum_cap = (long long)g.D * dd_grow(nent, Nc);
    }
    hipLaunchKernelGGL(tsx_k_dd_colsum, dim3(grid_for(nent)), dim3(TSX_BLOCK), 0, s->stream, g.D, (long long)nent, s->dd_coef, s->dd_colsum);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s->stream));
  *pays = true;
  return TSX_OK;
}

// TSX_DEBUG_CHECKS (tests only, tsx_pipeline_api.inc): directly behind tsx_k_dd_index the representatives are read back.  The rare wrong
// result of rounds 4-5 had ent_cell = {0, 0, 0, 0} on rank 0 while both indices written by the SAME launch were right: this says whether
// the stores never arrived (zero right here) or were wiped later, and a second launch of the same kernel into the same buffer says
// whether the loss is transient or a property of the process (e.g. of its copy of the kernel's code).
template <typename Relaunch>
static void dd_debug_after_index(tsx_solver *s, const char *where, int nent, const int *ent_cell, const int *rep, const int *pos,
                                 Relaunch relaunch) {
  const char *pre = getenv("TSX_DEBUG_CHECKS");
  if (!pre || !*pre || nent <= 0) return;
  char fn[512];
  snprintf(fn, sizeof(fn), "%s.%d", pre, (int)getpid());
  FILE *fh = fopen(fn, "a");
  if (!fh) return;
  const int n = nent < 16 ? nent : 16;
  int ec[16] = {0}, again[16] = {0};
  bool lost = false;
  if (hipStreamSynchronize(s->stream) == hipSuccess && hipMemcpy(ec, ent_cell, sizeof(int) * n, hipMemcpyDeviceToHost) == hipSuccess) {
    lost = nent > 1;
    for (int q = 1; q < n; ++q) lost = lost && ec[q] == 0;
    fprintf(fh, "rank %d index_check %s nent %d ptr %p%s first", s->grid.rank, where, nent, (const void *)ent_cell, lost ? " LOST" : "");
    for (int q = 0; q < n; ++q) fprintf(fh, " %d", ec[q]);
    fprintf(fh, "\n");
    if (lost) {
      relaunch();
      if (hipStreamSynchronize(s->stream) == hipSuccess && hipMemcpy(again, ent_cell, sizeof(int) * n, hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(fh, "rank %d index_check %s RELAUNCHED first", s->grid.rank, where);
        for (int q = 0; q < n; ++q) fprintf(fh, " %d", again[q]);
        // what the kernel read: rep / pos of the first cells
        int r8[8] = {0}, p8[8] = {0};
        (void)hipMemcpy(r8, rep, sizeof(r8), hipMemcpyDeviceToHost);
        (void)hipMemcpy(p8, pos, sizeof(p8), hipMemcpyDeviceToHost);
        fprintf(fh, " rep");
        for (int q = 0; q < 8; ++q) fprintf(fh, " %d", r8[q]);
        fprintf(fh, " pos");
        for (int q = 0; q < 8; ++q) fprintf(fh, " %d", p8[q]);
        fprintf(fh, "\n");
      }
    }
  }
  fclose(fh);
}

// ---- coordinates first (round 4).  On the LUT path a block is a deterministic function of its cell's four clamped float32
// coordinates, so cells can be grouped BEFORE anything is interpolated: hash the 16 bytes, same table / owner / compare (16
// bytes instead of 400) / scan / index pipeline, then only the nent distinct tuples are interpolated, straight into the shared
// storage (tsx_k_lut_diff2diff_ent).  The dense per-cell planes are not written at all (s->coef_dense_valid = false; whoever
// needs them -- the exact fp64 preconditioner, the one-lane kernels, tsx_diff_get_coeffs -- has them expanded from the entries,
// tsx_coef_ensure_dense).  Cells whose coordinates differ but whose blocks happen to coincide keep separate entries here
// (lossless either way).  3_10 and fp32 blocks only; TSX_DEDUP_COORDS=0 switches it off.
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_hash_coords(TsxGeo g, TsxLutDev L, const float4 *__restrict__ samp,
                                                                  const uint8_t *__restrict__ l1d, unsigned long long *__restrict__ h) {
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    unsigned long long v = TSX_DD_SEED;
    if (l1d[(int)(c / g.ncol)]) {
      v = TSX_DD_H1D;
    } else {
      const float4 a = tsx_lut_diff_clamp(L, samp[c]);
      v = tsx_dd_hash_step(v, 0, a.x);
      v = tsx_dd_hash_step(v, 1, a.y);
      v = tsx_dd_hash_step(v, 2, a.z);
      v = tsx_dd_hash_step(v, 3, a.w);
      v = tsx_dd_hash_final(v);
    }
    h[c] = v;
  }
}
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_resolve_coords(TsxGeo g, TsxLutDev L, const float4 *__restrict__ samp,
                                                                     const uint8_t *__restrict__ l1d, unsigned long long mask,
                                                                     const unsigned long long *__restrict__ h,
                                                                     const unsigned long long *__restrict__ keys,
                                                                     const int *__restrict__ owner, int *__restrict__ rep,
                                                                     int *__restrict__ flag) {
  const long long Nc = g.Nc;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const unsigned long long hv = h[c];
    unsigned long long slot = hv & mask;
    while (keys[slot] != hv) slot = (slot + 1) & mask;
    const int o = owner[slot];
    bool same = true;
    if (o != (int)c && !(l1d[(int)(c / g.ncol)] && l1d[o / g.ncol])) {
      const float4 a = tsx_lut_diff_clamp(L, samp[c]), b = tsx_lut_diff_clamp(L, samp[o]);
      same = __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
             __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w) &&
             !l1d[(int)(c / g.ncol)] && !l1d[o / g.ncol];
    }
    const int r = same ? o : (int)c;
    rep[c] = r;
    flag[c] = r == (int)c;
  }
}
// Does the grouping of the previous coefficient set still hold for these coordinates?  A spectral loop hands over one set of
// optical properties per g-point: the values change, but which cells share a tuple mostly does not (every clear-sky cell of a level
// still equals its neighbours; gas optics are functions of the level).  Every cell compares its clamped tuple with the one of its
// entry's representative cell, bit for bit; one mismatch anywhere (or a changed set of 1-D layers) and the grouping is rebuilt.  Two
// entries may have become equal to each other: they stay two entries (lossless, a little less shared).
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_validate_coords(TsxGeo g, TsxLutDev L, const float4 *__restrict__ samp,
                                                                      const uint8_t *__restrict__ l1d, const int *__restrict__ cidx,
                                                                      const int *__restrict__ ent_cell, int *__restrict__ bad) {
  const long long Nc = g.Nc;
  int mine = 0;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int o = ent_cell[cidx[c]];
    if (o == (int)c) continue;
    const bool c1 = l1d[(int)(c / g.ncol)] != 0, o1 = l1d[o / g.ncol] != 0;
    if (c1 || o1) {
      mine |= (c1 != o1);
      continue;
    }
    const float4 a = tsx_lut_diff_clamp(L, samp[c]), b = tsx_lut_diff_clamp(L, samp[o]);
    mine |= !(__float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
              __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w));
  }
  if (__any(mine) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}
// dense planes from the entries: C[q * Nc + c] = Cd[q * nent + cidx[c]]
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_dd_expand(long long Nc, int DD, long long nent, const float *__restrict__ Cd,
                                                             const int *__restrict__ cidx, float *__restrict__ C) {
  const long long n = Nc * DD;
  for (long long e = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; e < n; e += (long long)gridDim.x * TSX_BLOCK) {
    const long long q = e / Nc, c = e - q * Nc;
    C[e] = Cd[(size_t)q * nent + cidx[c]];
  }
}
int tsx_coef_ensure_dense(tsx_solver *s) {
  if (s->coef_dense_valid) return TSX_OK;
  const TsxGeo &g = s->geo;
  hipLaunchKernelGGL(tsx_k_dd_expand, dim3(grid_for(g.Nc * g.D * g.D, 8192)), dim3(TSX_BLOCK), 0, s->stream, g.Nc, g.D * g.D,
                     (long long)s->dd_nent, (const float *)s->dd_coef, (const int *)s->dd_cidx, (float *)s->coef);
  HIPCHK(hipGetLastError());
  s->coef_dense_valid = true;
  return TSX_OK;
}

// called by the LUT path right after tsx_cell_samples, BEFORE any interpolation.  *built = true: the shared storage holds the
// coefficients (dd_on, Cd / Ce / cidx / colsum), nothing dense was written; false: sharing would not pay (or is off) -- the caller
// interpolates every cell into the dense planes as before
int tsx_dedup_from_coords(tsx_solver *s, const TsxLutDev &L, bool *built) {
  *built = false;
  const TsxGeo &g = s->geo;
  const char *e = getenv("TSX_DEDUP_COORDS");
  if ((e && atoi(e) == 0) || !dedup_enabled() || g.D != 10 || s->coef_bytes != 4 || !s->cell_samp) return TSX_OK;
  if (g.Nc >= 0x7f7f7f7fll || !tsx_dd_wave64(s)) return TSX_OK;
  const int DD = g.D * g.D;
  const long long Nc = g.Nc;
  TsxDdScratch w;
  {
    int rc = dd_scratch(s, Nc, &w);
    if (rc) return rc;
  }
  const float4 *samp = (const float4 *)s->cell_samp;
  const int nb = grid_for(Nc, 8192);
  // the previous set's grouping, if it was made from coordinates too and still holds (TSX_DEDUP_REUSE=0: always rebuild)
  bool reuse = false;
  {
    const char *er = getenv("TSX_DEDUP_REUSE");
    if (s->dd_from_coords && s->dd_cidx && s->dd_ent_cell && s->dd_nent > 0 && !(er && atoi(er) == 0)) {
      HIPCHK(hipMemsetAsync(w.ttot.p, 0, sizeof(int), s->stream));
      hipLaunchKernelGGL(tsx_k_dd_validate_coords, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g, L, samp, s->l1d, (const int *)s->dd_cidx,
                         (const int *)s->dd_ent_cell, w.ttot.as<int>());
      HIPCHK(hipGetLastError());
      int bad = 1;
      HIPCHK(hipMemcpyAsync(&bad, w.ttot.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      reuse = bad == 0;
    }
  }
  s->dd_from_coords = false;
  int nent = 0;
  if (reuse) {
    nent = s->dd_nent;
  } else {
  HIPCHK(hipMemsetAsync(w.tk.p, 0, sizeof(unsigned long long) * (size_t)w.tsz, s->stream));
  HIPCHK(hipMemsetAsync(w.to.p, 0x7f, sizeof(int) * (size_t)w.tsz, s->stream));
  hipLaunchKernelGGL(tsx_k_dd_hash_coords, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g, L, samp, s->l1d, w.th.as<unsigned long long>());
  hipLaunchKernelGGL(tsx_k_dd_insert, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, Nc, w.tsz - 1, w.th.as<unsigned long long>(),
                     w.tk.as<unsigned long long>(), w.to.as<int>());
  hipLaunchKernelGGL(tsx_k_dd_resolve_coords, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g, L, samp, s->l1d, w.tsz - 1,
                     w.th.as<unsigned long long>(), w.tk.as<unsigned long long>(), w.to.as<int>(), w.trep.as<int>(), w.tflag.as<int>());
  hipLaunchKernelGGL(tsx_k_scan_sums, dim3(w.nsb), dim3(TSX_BLOCK), 0, s->stream, Nc, w.tflag.as<int>(), w.tsum.as<int>());
  hipLaunchKernelGGL(tsx_k_scan_top, dim3(1), dim3(1024), 0, s->stream, w.nsb, w.tsum.as<int>(), w.ttot.as<int>());
  hipLaunchKernelGGL(tsx_k_scan_write, dim3(w.nsb), dim3(TSX_BLOCK), 0, s->stream, Nc, w.tflag.as<int>(), w.tsum.as<int>(), w.tpos.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(&nent, w.ttot.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  }
  if ((long long)nent * 2 > Nc) return TSX_OK;  // mostly distinct coordinates: dense planes, then the block-based build decides
  s->dd_nent = nent;
  if (!s->dd_cidx) HIPCHK(tsx_dev_malloc((void **)&s->dd_cidx, sizeof(int) * (size_t)Nc));
  if (!s->dd_cidx_split) HIPCHK(tsx_dev_malloc((void **)&s->dd_cidx_split, sizeof(int) * (size_t)Nc));
  if (s->dd_cap < nent) {
    if (s->dd_coef) HIPCHK(tsx_dev_free(s->dd_coef));
    if (s->dd_ent_cell) HIPCHK(tsx_dev_free(s->dd_ent_cell));
    s->dd_coef = nullptr;
    s->dd_ent_cell = nullptr;
    // grow-only, with slack: a spectral loop's entry count wanders from g-point to g-point, the storage should not
    const int cap = dd_grow(nent, Nc);
    HIPCHK(tsx_dev_malloc((void **)&s->dd_coef, sizeof(float) * (size_t)DD * cap * 2));  // plane-major, then entry-major
    HIPCHK(tsx_dev_malloc((void **)&s->dd_ent_cell, sizeof(int) * (size_t)cap));
    s->dd_cap = cap;
  }
  const bool split = g.xm % 2 == 0;
  if (!reuse) {
    auto launch_index = [&]() {
      hipLaunchKernelGGL(tsx_k_dd_index, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, g, w.trep.as<int>(), w.tpos.as<int>(), s->dd_cidx,
                         split ? s->dd_cidx_split : (int *)nullptr, s->dd_ent_cell);
    };
    launch_index();
    dd_debug_after_index(s, "coords", nent, s->dd_ent_cell, w.trep.as<int>(), w.tpos.as<int>(), launch_index);
  }
  s->dd_coef_e = s->dd_coef + (size_t)DD * s->dd_cap;
  hipLaunchKernelGGL((tsx_k_lut_diff2diff_ent<100>), dim3(grid_for(nent, 8192)), dim3(TSX_BLOCK), 0, s->stream, g, L, s->l1d,
                     (long long)nent, (const int *)s->dd_ent_cell, samp, s->dd_coef, s->dd_coef_e);
  if (s->dd_colsum_cap < (long long)g.D * nent) {
    if (s->dd_colsum) HIPCHK(tsx_dev_free(s->dd_colsum));
    s->dd_colsum = nullptr;
    HIPCHK(tsx_dev_malloc((void **)&s->dd_colsum, sizeof(double) * (size_t)g.D * dd_grow(nent, Nc)));
    s->dd_colsum_cap = (long long)g.D * dd_grow(nent, Nc);
  }
  hipLaunchKernelGGL(tsx_k_dd_colsum, dim3(grid_for(nent)), dim3(TSX_BLOCK), 0, s->stream, g.D, (long long)nent, s->dd_coef, s->dd_colsum);
  HIPCHK(hipGetLastError());
  s->dd_valid = true;
  s->dd_on = true;
  s->dd_pc = false;
  s->dd_nent_near = 0;
  s->dd_hash_ready = false;
  s->pc_coef = s->dd_coef;
  s->pc_cidx_split = s->dd_cidx_split;
  s->pc_ent_cell = s->dd_ent_cell;
  s->pc_nent = s->dd_nent;
  s->coef_dense_valid = false;
  s->dd_from_coords = true;
  s->dd_reused = reuse;
  *built = true;
  return TSX_OK;
}

// (re)build the shared-block storage for the current coefficients; leaves s->dd_on = false where it does not pay, and then tries
// the preconditioner-only grouping of near-identical blocks (s->dd_pc)
int tsx_dedup_ensure(tsx_solver *s) {
  if (s->dd_valid) return TSX_OK;
  s->dd_valid = true;
  s->dd_on = false;
  s->dd_pc = false;
  s->dd_nent_near = 0;
  if (!dedup_enabled() || s->coef_bytes != 4 || !s->have_coeffs) return TSX_OK;
  const TsxGeo &g = s->geo;
  // cell indices are ints and the owner table's "no owner yet" value is 0x7f7f7f7f (byte-wise memset): stay below it
  if (g.Nc >= 0x7f7f7f7fll || !tsx_dd_wave64(s)) return TSX_OK;
  bool pays = false;
  s->dd_from_coords = false;  // the index is rewritten from the blocks
  int rc = dd_build(s, false, &pays);
  if (rc) return rc;
  s->dd_on = pays;
  if (pays) {  // the preconditioner reads the same arrays unless the near grouping below takes over
    s->pc_coef = s->dd_coef;
    s->pc_cidx_split = s->dd_cidx_split;
    s->pc_ent_cell = s->dd_ent_cell;
    s->pc_nent = s->dd_nent;
  }
  // the preconditioner's grouping of near-identical blocks: where nothing (or too little) is bit-identical.  On top of the
  // bit-identical sharing it was measured too (TSX_DEDUP_NEAR=2; the benchmark field: 291 533 distinct blocks -> 1 915 groups):
  // pass 33.7 -> 31.8 us, solve 15.33 -> 15.07 ms -- the table was served by L2 / MALL already; not worth a second build per
  // coefficient set
  const char *ne = getenv("TSX_DEDUP_NEAR");
  const bool on_top = ne && atoi(ne) == 2;
  // 3_10 only: the 8_16 pass gathers 20 records per level through the index -- scattered over the groups they cost more than the
  // same records streamed per cell (every block distinct, 256 x 256 x 64: pass 246 -> 408 us, 48.7 -> 34.2 M cells/s)
  if (!dedup_near_enabled() || (s->dd_on && !on_top) || (g.ntop != 2 && !on_top)) return TSX_OK;
  bool npays = false;
  if ((rc = dd_build(s, true, &npays))) return rc;
  if (npays && (!s->dd_on || (long long)s->dd_nent_near * 2 <= (long long)s->dd_nent)) {
    s->dd_pc = true;
    s->pc_coef = s->pcn_coef;
    s->pc_cidx_split = s->pcn_cidx_split;
    s->pc_ent_cell = s->pcn_ent_cell;
    s->pc_nent = s->dd_nent_near;
  }
  return TSX_OK;
}


// =====================================================================================================================
// Shared storage of identical *packed records* (the preconditioner's per-cell recurrence records, tsx_kernels_pcs.hpp).
// The column recurrences depend on everything between a cell and the surface; below the lowest cloud of a column -- and in
// every clear column -- they are the same at a given level for all columns (83 % of the cells of the benchmark field).  Same
// pipeline as for the blocks: hash the R records of a cell, table with the smallest cell as owner, exact comparison, scan,
// compact.  Cells are taken in the order the records are stored in (colour-split).  Lossless.
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_rec_hash(long long Nc, int R, const uint4 *__restrict__ P,
                                                            unsigned long long *__restrict__ h) {
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    unsigned long long v = 0x13198a2e03707344ull;
    for (int r = 0; r < R; ++r) {
      const uint4 q = P[(size_t)r * Nc + c];
      v = tsx_mix64(v, ((unsigned long long)q.x << 32) | q.y);
      v = tsx_mix64(v, ((unsigned long long)q.z << 32) | q.w);
    }
    if (v == TSX_DD_EMPTY) v = 0x5555555555555555ull;
    h[c] = v;
  }
}
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_rec_resolve(long long Nc, int R, const uint4 *__restrict__ P, unsigned long long mask,
                                                               const unsigned long long *__restrict__ h,
                                                               const unsigned long long *__restrict__ keys,
                                                               const int *__restrict__ owner, int *__restrict__ rep,
                                                               int *__restrict__ flag) {
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const unsigned long long hv = h[c];
    unsigned long long slot = hv & mask;
    while (keys[slot] != hv) slot = (slot + 1) & mask;
    const int o = owner[slot];
    bool same = true;
    if (o != (int)c) {
      for (int r = 0; r < R; ++r) {
        const uint4 a = P[(size_t)r * Nc + c], b = P[(size_t)r * Nc + o];
        same &= a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
      }
    }
    const int rr = same ? o : (int)c;
    rep[c] = rr;
    flag[c] = rr == (int)c;
  }
}
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_rec_index(long long Nc, const int *__restrict__ rep, const int *__restrict__ pos,
                                                             int *__restrict__ pidx, int *__restrict__ ent_cell) {
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int r = rep[c], id = pos[r];
    pidx[c] = id;
    if (r == (int)c) ent_cell[id] = (int)c;
  }
}
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_rec_compact(long long Nc, int R, long long n, const uint4 *__restrict__ P,
                                                               const int *__restrict__ ent_cell, uint4 *__restrict__ PT) {
  for (long long q = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; q < n * R; q += (long long)gridDim.x * TSX_BLOCK) {
    const long long r = q / n, id = q - r * n;
    PT[q] = P[(size_t)r * Nc + ent_cell[id]];
  }
}

// Does the previous coefficient set's grouping still hold for these records?  Every cell compares its R records with those of its
// group's representative cell, bit for bit (the dd grouping's tsx_k_dd_validate_coords, for the packed recurrence records): a
// spectral loop changes the values from g-point to g-point, rarely which cells -- every clear column below its lowest cloud --
// have equal ones.  Two groups may have become equal to each other: they stay two (lossless).
__global__ __launch_bounds__(TSX_BLOCK) void tsx_k_rec_validate(long long Nc, int R, const uint4 *__restrict__ P, const int *__restrict__ pidx,
                                                                const int *__restrict__ ent_cell, int *__restrict__ bad) {
  int mine = 0;
  for (long long c = (long long)blockIdx.x * TSX_BLOCK + threadIdx.x; c < Nc; c += (long long)gridDim.x * TSX_BLOCK) {
    const int o = ent_cell[pidx[c]];
    if (o == (int)c) continue;
    for (int r = 0; r < R; ++r) {
      const uint4 a = P[(size_t)r * Nc + c], b = P[(size_t)r * Nc + o];
      mine |= !(a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w);
    }
  }
  if (__any(mine) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

// P: R planes of Nc records.  On success with sharing worth it (2 n <= Nc): s->pcr_idx[c], s->pcr_tab[r * n + id], s->pcr_n = n
// and returns with s->pcr_on = true; else s->pcr_on = false.  TSX_PC_RECSHARE=0 switches it off.
int tsx_records_share(tsx_solver *s, int R, const uint4 *P) {
  const long long n_have = s->pcr_n;  // entries of the grouping the arrays still hold (pcr_have_R > 0)
  s->pcr_on = false;
  s->pcr_n = 0;
  const bool enabled = !(getenv("TSX_PC_RECSHARE") && atoi(getenv("TSX_PC_RECSHARE")) == 0);  // read per call: tests switch it
  const long long Nc = s->geo.Nc;
  if (!enabled || Nc >= 0x7f7f7f7fll || !tsx_dd_wave64(s)) return TSX_OK;
  TsxDdScratch w;
  {
    int rc = dd_scratch(s, Nc, &w);
    if (rc) return rc;
  }
  const int nb = grid_for(Nc, 8192);
  s->pcr_reused = false;
  {
    // round 6: the previous set's grouping, if one validation kernel finds it still exact (TSX_DEDUP_REUSE=0: always rebuild -- the
    // switch of the block grouping's reuse, so that the test that compares against a solver rebuilding everything covers both)
    const char *er = getenv("TSX_DEDUP_REUSE");
    // (on the LUT path only where the block grouping itself was taken over: a rebuilt one renumbers the block indices inside the
    // records, and the validation -- 0.12 ms on 4.2 M cells -- would be spent to learn that; config 4, profiles/r06)
    const bool plausible = !s->dd_from_coords || s->dd_reused;
    if (plausible && s->pcr_have_R == R && s->pcr_have_P == (const void *)P && s->pcr_idx && s->pcr_ent && s->pcr_tab && n_have > 0 &&
        n_have * 2 <= Nc && !(er && atoi(er) == 0)) {
      HIPCHK(hipMemsetAsync(w.ttot.p, 0, sizeof(int), s->stream));
      hipLaunchKernelGGL(tsx_k_rec_validate, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, Nc, R, P, (const int *)s->pcr_idx, (const int *)s->pcr_ent,
                         w.ttot.as<int>());
      HIPCHK(hipGetLastError());
      int bad = 1;
      HIPCHK(hipMemcpyAsync(&bad, w.ttot.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      if (!bad) {
        hipLaunchKernelGGL(tsx_k_rec_compact, dim3(grid_for(n_have * R, 8192)), dim3(TSX_BLOCK), 0, s->stream, Nc, R, n_have, P, s->pcr_ent,
                           (uint4 *)s->pcr_tab);
        HIPCHK(hipGetLastError());
        s->pcr_n = n_have;
        s->pcr_on = true;
        s->pcr_reused = true;
        return TSX_OK;
      }
    }
  }
  s->pcr_have_R = 0;
  HIPCHK(hipMemsetAsync(w.tk.p, 0, sizeof(unsigned long long) * (size_t)w.tsz, s->stream));
  HIPCHK(hipMemsetAsync(w.to.p, 0x7f, sizeof(int) * (size_t)w.tsz, s->stream));
  hipLaunchKernelGGL(tsx_k_rec_hash, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, Nc, R, P, w.th.as<unsigned long long>());
  hipLaunchKernelGGL(tsx_k_dd_insert, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, Nc, w.tsz - 1, w.th.as<unsigned long long>(),
                     w.tk.as<unsigned long long>(), w.to.as<int>());
  hipLaunchKernelGGL(tsx_k_rec_resolve, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, Nc, R, P, w.tsz - 1, w.th.as<unsigned long long>(),
                     w.tk.as<unsigned long long>(), w.to.as<int>(), w.trep.as<int>(), w.tflag.as<int>());
  hipLaunchKernelGGL(tsx_k_scan_sums, dim3(w.nsb), dim3(TSX_BLOCK), 0, s->stream, Nc, w.tflag.as<int>(), w.tsum.as<int>());
  hipLaunchKernelGGL(tsx_k_scan_top, dim3(1), dim3(1024), 0, s->stream, w.nsb, w.tsum.as<int>(), w.ttot.as<int>());
  hipLaunchKernelGGL(tsx_k_scan_write, dim3(w.nsb), dim3(TSX_BLOCK), 0, s->stream, Nc, w.tflag.as<int>(), w.tsum.as<int>(),
                     w.tpos.as<int>());
  HIPCHK(hipGetLastError());
  int n = 0;
  HIPCHK(hipMemcpyAsync(&n, w.ttot.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  s->pcr_n = n;
  if ((long long)n * 2 > Nc) return TSX_OK;
  if (!s->pcr_idx) HIPCHK(tsx_dev_malloc((void **)&s->pcr_idx, sizeof(int) * (size_t)Nc));
  if (s->pcr_cap < (long long)n * R) {
    if (s->pcr_tab) HIPCHK(tsx_dev_free(s->pcr_tab));
    if (s->pcr_ent) HIPCHK(tsx_dev_free(s->pcr_ent));
    s->pcr_tab = nullptr;
    s->pcr_ent = nullptr;
    const long long ncap = dd_grow(n, Nc);
    HIPCHK(tsx_dev_malloc((void **)&s->pcr_tab, sizeof(uint4) * (size_t)ncap * R));
    HIPCHK(tsx_dev_malloc((void **)&s->pcr_ent, sizeof(int) * (size_t)ncap));
    s->pcr_cap = ncap * R;
  }
  hipLaunchKernelGGL(tsx_k_rec_index, dim3(nb), dim3(TSX_BLOCK), 0, s->stream, Nc, w.trep.as<int>(), w.tpos.as<int>(), s->pcr_idx,
                     s->pcr_ent);
  hipLaunchKernelGGL(tsx_k_rec_compact, dim3(grid_for((long long)n * R, 8192)), dim3(TSX_BLOCK), 0, s->stream, Nc, R, (long long)n, P,
                     s->pcr_ent, (uint4 *)s->pcr_tab);
  HIPCHK(hipGetLastError());
  s->pcr_on = true;
  s->pcr_have_R = R;
  s->pcr_have_P = (const void *)P;
  return TSX_OK;
}

TSX_CODE_PROBE(dedup)  // tsx_host.hpp: this unit's code object as it sits in device memory (diagnostics)
