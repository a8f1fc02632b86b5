// tsx_peer.hip -- device-resident peer transport for the face halos and the dot-product all-reduces.
//
// What it replaces: `exchange_diffuse_boundary` / `exchange_direct_boundary` (MPI_Isend / Irecv of the entering side streams,
// src/pprts_explicit.F90:715-848, 1076-1140) and the MPI_Allreduce of the Krylov dots inside PETSc's KSP.  RCCL's grouped
// ncclSend / ncclRecv costs a launch of its own per exchange (15-30 us); an iteration of the default solver issues 40 exchanges
// of 64 KB - 0.5 MB and 3 all-reduces of 3 doubles -- a latency problem, not a bandwidth problem.
//
// Here every rank owns one *mailbox*: a fine-grained (uncached) device allocation that the other ranks of the node map into
// their address space with hipIpcGetMemHandle / hipIpcOpenMemHandle (over xGMI on a node; two rank processes sharing one
// device work the same way, which is how the 1-GPU test pool executes it).  A message is written by the SENDER's kernel
// straight into the receiver's mailbox with plain stores, followed by the store of a sequence number once those stores have been
// acknowledged; the receiver's kernel polls that number, copies the payload out and stores an acknowledgement into the sender's
// mailbox (ordering for uncached memory without cache maintenance: tsx_peer_dev.hpp).  Nothing but kernels on the caller's
// stream: no host round trip, no library call, no second stream needed.  Kernels that produce or consume halo data can take over
// either half themselves (tsx_peer_prepare_send / tsx_peer_expect below): the red-black passes store their boundary records
// into the neighbours' slots and read their neighbours' in place (tsx_kernels_pcs.hpp), acknowledging with their next send.
//
//   mailbox of rank r:
//     seq[q]   q = W, E, S, N: number of messages delivered THROUGH MY FACE q (written by the neighbour behind that face)
//     ack[q]   number of MY messages sent through my face q that the neighbour has consumed (written by that neighbour)
//     ar[2][R] all-reduce contributions: {v[TSX_NSLOTS], seq} per rank and parity (written by every rank, also by r itself)
//     data[q][2][cap]  payload, double-buffered by message parity
//
//   send n through face q (tsx_k_peer_send): wait ack[q] >= n - 2 (the slot's previous message was consumed), store the payload
//     into the neighbour's data[q ^ 1][n & 1], fence, the last workgroup stores the neighbour's seq[q ^ 1] = n.
//   recv n through face q (tsx_k_peer_recv): wait seq[q] >= n, copy data[q][n & 1] to the caller's buffer, the last workgroup
//     stores the neighbour's ack[q ^ 1] = n.
//   all-reduce n (tsx_k_peer_allreduce, one workgroup): lane r stores my partial sums + n into rank r's ar[n & 1][me]; lane r
//     then waits for ar[n & 1][r] of my own mailbox; lane 0 adds them in rank order -- every rank gets bit-identical sums.
//     No acknowledgement needed: a rank writes n only after it finished n - 1, which needed every rank's contribution n - 1,
//     which that rank sent after it had finished reading n - 2.
//
// Messages between two ranks are matched by count, per face, like MPI's non-overtaking rule: every rank must issue the same
// sequence of exchanges (the solver does: the exchanges are part of the collective solve).  With two ranks along a periodic
// axis the W and E neighbour are the same rank, with distinct faces and therefore distinct counters.
//
// Every wait is bounded (TSX_PEER_TIMEOUT_S, default 20 s of the 100 MHz wall clock): on expiry the kernel records an error in
// the mailbox header and gives up, the host reports TSX_ERR_COMM at the next synchronisation -- a lost rank never hangs the GPU;
// once an error is recorded every further wait of this rank gives up at once.
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "tsx_host.hpp"
#include "tsx_peer.hpp"
#include "tsx_peer_dev.hpp"

namespace {

constexpr int kMaxRanks = TSX_PEER_MAX_RANKS;
constexpr size_t kHdrBytes = TSX_PEER_HDR_BYTES;

using PeerArSlot = TsxPeerArSlot;
static_assert(TSX_PEER_MAX_RANKS == TSX_PEER_MAX_RANKS_DEV, "rank limit");

using PeerHdr = TsxPeerHdr;

struct PeerBlob {  // what tsx_comm_peer_export hands to the host's all-gather (TSX_PEER_BLOB_BYTES)
  hipIpcMemHandle_t handle;
  unsigned long long bytes, cap;
  void *ptr;  // valid in the exporting process only (ranks living in one process use it directly)
  int pid, device, rank, nranks;
  char host[32];
  int pci[3];  // PCI domain, bus, device of `device`: rank processes that each see ONE GPU (HIP_VISIBLE_DEVICES) all call it device 0
};
static_assert(sizeof(PeerBlob) <= TSX_PEER_BLOB_BYTES, "blob size");

using PeerXArgs = TsxPeerXArgs;
#define ld_acquire_sys tsx_peer_ld_acquire
#define st_release_sys tsx_peer_st_release
#define wait_ge tsx_peer_wait_ge
#define peer_data tsx_peer_data

__device__ __forceinline__ void copy16(char *__restrict__ d, const char *__restrict__ s, unsigned long long bytes, int nblk) {
  // bytes is a multiple of 8 (doubles); 16-byte pieces, the odd double at the end by lane 0
  const unsigned long long n16 = bytes >> 4;
  const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
  uint4 *d4 = reinterpret_cast<uint4 *>(d);
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)nblk * blockDim.x)
    d4[i] = s4[i];
  if ((bytes & 8) && blockIdx.x == 0 && threadIdx.x == 0)
    *reinterpret_cast<unsigned long long *>(d + (n16 << 4)) = *reinterpret_cast<const unsigned long long *>(s + (n16 << 4));
}
#define peer_fail tsx_peer_fail

// grid (nblk, 4): blockIdx.y = face
__global__ __launch_bounds__(256) void tsx_k_peer_send(PeerXArgs a) {
  const int q = blockIdx.y;
  if (a.bytes[q] == 0) return;
  if (a.done && *a.done) return;
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const PeerHdr *h = reinterpret_cast<const PeerHdr *>(a.mine);
    unsigned long long have = 0;
    ok = 1;
    // acknowledge what kernels before this one consumed in place (tsx_peer_expect): first, so that two ranks never wait for
    // each other's acknowledgement
    if (blockIdx.x == 0 && a.ackn[q]) tsx_peer_post(&reinterpret_cast<PeerHdr *>(a.remote[q])->ack[q ^ 1], a.ackn[q], a.heavy);
    if (a.n[q] > 2 && !wait_ge(&h->ack[q], a.n[q] - 2, a.ticks, &have, a.heavy, a.mine)) {
      ok = 0;
      peer_fail(a.mine, 1, q, a.n[q] - 2, have);
    }
  }
  __syncthreads();
  if (!ok) return;
  copy16(peer_data(a.remote[q], a.data_off, a.cap, q ^ 1, (int)(a.n[q] & 1)), a.src[q], a.bytes[q], gridDim.x);
  tsx_peer_stores_done(a.heavy);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = atomicAdd(&a.blkctr[q], 1u);
    if (prev + 1 == gridDim.x) {  // every workgroup's stores have reached memory: publish
      a.blkctr[q] = 0;
      if (a.heavy) __threadfence_system();
      tsx_peer_post(&reinterpret_cast<PeerHdr *>(a.remote[q])->seq[q ^ 1], a.n[q], a.heavy);
    }
  }
}

__global__ __launch_bounds__(256) void tsx_k_peer_recv(PeerXArgs a) {
  const int q = blockIdx.y;
  if (a.bytes[q] == 0) return;
  if (a.done && *a.done) return;
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const PeerHdr *h = reinterpret_cast<const PeerHdr *>(a.mine);
    unsigned long long have = 0;
    ok = 1;
    if (!wait_ge(&h->seq[q], a.n[q], a.ticks, &have, a.heavy, a.mine)) {
      ok = 0;
      peer_fail(a.mine, 2, q, a.n[q], have);
    }
  }
  __syncthreads();
  if (!ok) return;
  copy16(a.dst[q], peer_data(a.mine, a.data_off, a.cap, q, (int)(a.n[q] & 1)), a.bytes[q], gridDim.x);
  __syncthreads();
  if (threadIdx.x == 0) {  // every lane's loads from the slot have returned (their stores were issued before the barrier)
    if (a.heavy) __threadfence();
    const unsigned int prev = atomicAdd(&a.blkctr[q], 1u);
    if (prev + 1 == gridDim.x) {
      a.blkctr[q] = 0;
      tsx_peer_post(&reinterpret_cast<PeerHdr *>(a.remote[q])->ack[q ^ 1], a.n[q], a.heavy);
    }
  }
}

using PeerArArgs = TsxPeerArArgs;

// one workgroup of 64 lanes; v: nvals (<= TSX_NSLOTS + 1) doubles in device memory, summed over the ranks in place
__global__ __launch_bounds__(64) void tsx_k_peer_allreduce(PeerArArgs a, double *__restrict__ v) { tsx_peer_allreduce_wg(a, v); }

// ---- self test: patterns through the mailboxes (tsx_comm_peer_selftest)
__device__ __forceinline__ double peer_pattern(int rank, int face, int round, long long i) {
  return (double)rank * 1048576.0 + (double)face * 65536.0 + (double)(round & 255) * 256.0 + (double)(i % 251) + 0.5;
}
__global__ void tsx_k_peer_fill(double *__restrict__ b0, double *__restrict__ b1, double *__restrict__ b2, double *__restrict__ b3,
                                long long nx, long long ny, int rank, int round) {
  double *b[4] = {b0, b1, b2, b3};
  for (int q = 0; q < 4; ++q) {
    const long long n = q < 2 ? nx : ny;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      b[q][i] = peer_pattern(rank, q, round, i);
  }
}
// recv[q] must hold what the neighbour behind face q sent through ITS face q ^ 1
__global__ void tsx_k_peer_verify(const double *__restrict__ b0, const double *__restrict__ b1, const double *__restrict__ b2,
                                  const double *__restrict__ b3, long long nx, long long ny, int nw, int ne, int ns, int nn, int round,
                                  unsigned long long *__restrict__ bad) {
  const double *b[4] = {b0, b1, b2, b3};
  const int nb[4] = {nw, ne, ns, nn};
  unsigned long long mine = 0;
  for (int q = 0; q < 4; ++q) {
    const long long n = q < 2 ? nx : ny;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      mine += b[q][i] != peer_pattern(nb[q], q ^ 1, round, i);
  }
  if (mine) atomicAdd(bad, mine);
}

}  // namespace

struct TsxPeer {
  char *mine = nullptr;
  size_t bytes = 0, cap = 0, data_off = 0, ar_off = 0, tag_off = 0, tag_edge = 0;
  char *box[kMaxRanks] = {nullptr};      // every rank's mailbox in this process's address space (box[rank] == mine)
  bool opened[kMaxRanks] = {false};      // mapped with hipIpcOpenMemHandle (to be closed)
  unsigned long long sent[4] = {0, 0, 0, 0}, rcvd[4] = {0, 0, 0, 0}, ar_n = 0;
  unsigned int *blkctr = nullptr;        // [32]: send 0..3, recv 4..7, producer kernels 8, the flow kernel's faces 16..23
  unsigned long long ticks = 0;
  int heavy = 0;                         // TSX_PEER_FENCES / tsx_comm_peer_set_fences (tsx_peer_dev.hpp)
  bool attached = false;
  int colocated = 1;  // ranks of the job on this rank's device (tests and one-GPU boxes: all of them)
};

static size_t peer_capacity(const tsx_solver *s) {
  // the largest face message any exchange of this solver sends: the diffuse halo (nside / 2 streams x Nz x edge doubles), the
  // direct beam's (dirside streams x Nz x edge), the preconditioner's records (Nz x edge words)
  // ... sized by the GLOBAL extents so that every rank of an uneven split (xs = (xi * Nx) / nxp) arrives at the same slot size
  const TsxGeo &g = s->geo;
  const int gx = s->grid.glob_xm > g.xm ? s->grid.glob_xm : g.xm, gy = s->grid.glob_ym > g.ym ? s->grid.glob_ym : g.ym;
  const size_t edge = (size_t)(gx > gy ? gx : gy);
  const size_t streams = 4;  // >= nside / 2 (2) and dirside (1 / 2)
  size_t b = streams * (size_t)(g.Nz + 1) * edge * sizeof(double);
  return (b + 255) & ~(size_t)255;
}

extern "C" int tsx_comm_peer_export(tsx_solver *s, void *blob) {
  ARGCHK(s && blob, "tsx_comm_peer_export: null");
  ARGCHK(s->grid.nranks >= 1 && s->grid.nranks <= kMaxRanks, "tsx_comm_peer_export: more ranks than TSX_PEER_MAX_RANKS");
  HIPCHK(hipSetDevice(s->device));
  if (!s->peer) {
    TsxPeer *p = new TsxPeer();
    p->cap = peer_capacity(s);
    p->ar_off = kHdrBytes;
    p->data_off = p->ar_off + sizeof(PeerArSlot) * 2 * kMaxRanks;
    p->data_off = (p->data_off + 255) & ~(size_t)255;
    // behind the payload slots: the flow kernel's tags (tsx_k_pcs_flow FPEER), one word per row / column of a face and parity
    p->tag_off = p->data_off + (size_t)4 * 2 * p->cap;
    {
      const int gx = s->grid.glob_xm > s->geo.xm ? s->grid.glob_xm : s->geo.xm, gy = s->grid.glob_ym > s->geo.ym ? s->grid.glob_ym : s->geo.ym;
      p->tag_edge = (size_t)(gx > gy ? gx : gy);
    }
    p->bytes = (p->tag_off + (size_t)4 * 2 * p->tag_edge * sizeof(unsigned) + 255) & ~(size_t)255;
    void *m = nullptr;
    // uncached: the counters and payloads are written by other agents while kernels of this rank are running
    hipError_t e = hipExtMallocWithFlags(&m, p->bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) e = hipExtMallocWithFlags(&m, p->bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
      delete p;
      tsx_set_error(std::string("tsx_comm_peer_export: fine-grained allocation failed: ") + hipGetErrorString(e));
      return TSX_ERR_HIP;
    }
    p->mine = (char *)m;
    HIPCHK(tsx_dev_quarantine(p->mine, p->bytes));  // zeroed, and proven to keep its contents (tsx_pool.hip: fresh driver memory may not)
    HIPCHK(tsx_dev_malloc((void **)&p->blkctr, sizeof(unsigned int) * 32));
    HIPCHK(hipMemset(p->blkctr, 0, sizeof(unsigned int) * 32));
    const char *to = getenv("TSX_PEER_TIMEOUT_S");
    const double sec = to ? atof(to) : 20.0;
    p->ticks = (unsigned long long)((sec > 0 ? sec : 20.0) * 1e8);  // wall_clock64: 100 MHz
    p->heavy = getenv("TSX_PEER_FENCES") && atoi(getenv("TSX_PEER_FENCES")) != 0;
    HIPCHK(hipDeviceSynchronize());
    s->peer = p;
  }
  TsxPeer *p = s->peer;
  PeerBlob b;
  memset(&b, 0, sizeof(b));
  HIPCHK(hipIpcGetMemHandle(&b.handle, p->mine));
  b.bytes = p->bytes;
  b.cap = p->cap;
  b.ptr = p->mine;
  b.pid = (int)getpid();
  b.device = s->device;
  if (hipDeviceGetAttribute(&b.pci[0], hipDeviceAttributePciDomainID, s->device) != hipSuccess ||
      hipDeviceGetAttribute(&b.pci[1], hipDeviceAttributePciBusId, s->device) != hipSuccess ||
      hipDeviceGetAttribute(&b.pci[2], hipDeviceAttributePciDeviceId, s->device) != hipSuccess) {
    b.pci[0] = b.pci[1] = b.pci[2] = -1;
    (void)hipGetLastError();
  }
  b.rank = s->grid.rank;
  b.nranks = s->grid.nranks;
  gethostname(b.host, sizeof(b.host) - 1);
  memset(blob, 0, TSX_PEER_BLOB_BYTES);
  memcpy(blob, &b, sizeof(b));
  return TSX_OK;
}

extern "C" int tsx_comm_peer_attach(tsx_solver *s, const void *blobs) {
  ARGCHK(s && blobs, "tsx_comm_peer_attach: null");
  if (!s->peer) {
    tsx_set_error("tsx_comm_peer_attach: call tsx_comm_peer_export first");
    return TSX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  TsxPeer *p = s->peer;
  const int R = s->grid.nranks, me = s->grid.rank;
  char myhost[32] = {0};
  gethostname(myhost, sizeof(myhost) - 1);
  for (int r = 0; r < R; ++r) {
    PeerBlob b;
    memcpy(&b, (const char *)blobs + (size_t)r * TSX_PEER_BLOB_BYTES, sizeof(b));
    if (b.rank != r || b.nranks != R || b.bytes != p->bytes || b.cap != p->cap) {
      tsx_set_error("tsx_comm_peer_attach: blob " + std::to_string(r) + " does not describe rank " + std::to_string(r) +
                    " of this job (all ranks need the same Nz and local extents' maximum)");
      return TSX_ERR_ARG;
    }
    if (strncmp(b.host, myhost, sizeof(myhost)) != 0) {
      tsx_set_error("tsx_comm_peer_attach: rank " + std::to_string(r) + " runs on another host: the peer transport is node-local");
      return TSX_ERR_COMM;
    }
    if (r == me) {
      p->box[r] = p->mine;
    } else if (b.pid == (int)getpid()) {
      p->box[r] = (char *)b.ptr;  // several ranks in one process: the pointer is valid as it is
      if (b.device != s->device) {
        int can = 0;
        HIPCHK(hipDeviceCanAccessPeer(&can, s->device, b.device));
        if (!can) {
          tsx_set_error("tsx_comm_peer_attach: no peer access between devices");
          return TSX_ERR_COMM;
        }
        hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIPCHK(e);
        (void)hipGetLastError();
      }
    } else {
      void *m = nullptr;
      hipError_t e = hipIpcOpenMemHandle(&m, b.handle, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        tsx_set_error(std::string("tsx_comm_peer_attach: hipIpcOpenMemHandle(rank ") + std::to_string(r) + "): " + hipGetErrorString(e));
        (void)hipGetLastError();
        return TSX_ERR_COMM;
      }
      p->box[r] = (char *)m;
      p->opened[r] = true;
    }
  }
  // Ordering around the mailbox flags.  The light ordering (tsx_peer_dev.hpp: uncached mailboxes, s_waitcnt + relaxed stores) has
  // only ever run between processes sharing ONE device; the first contact across devices (xGMI, another GPU's HBM) starts with the
  // textbook full system-scope fences.  TSX_PEER_FENCES=0 relaxes, =1 forces; tsx_comm_peer_set_fences overrides either.
  {
    bool cross = false;
    int here = 0;
    PeerBlob mine;
    memcpy(&mine, (const char *)blobs + (size_t)me * TSX_PEER_BLOB_BYTES, sizeof(mine));
    for (int r = 0; r < R; ++r) {
      PeerBlob b;
      memcpy(&b, (const char *)blobs + (size_t)r * TSX_PEER_BLOB_BYTES, sizeof(b));
      // the same physical device: by PCI address where both sides could read it (the ordinals of processes that each see one GPU
      // coincide), else by ordinal
      const bool pci_ok = mine.pci[1] >= 0 && b.pci[1] >= 0;
      const bool same = pci_ok ? (b.pci[0] == mine.pci[0] && b.pci[1] == mine.pci[1] && b.pci[2] == mine.pci[2]) : b.device == s->device;
      cross = cross || !same;
      here += same ? 1 : 0;
    }
    if (!getenv("TSX_PEER_FENCES")) p->heavy = cross;
    p->colocated = here > 0 ? here : 1;
  }
  p->attached = true;
  s->pcg_key = -1;  // decisions agreed over the previous transport are agreed again (tsx_pc_global_agree)
  return TSX_OK;
}

bool tsx_peer_ready(const tsx_solver *s) { return s->peer && s->peer->attached; }
int tsx_peer_colocated(const tsx_solver *s) { return s->peer && s->peer->attached ? s->peer->colocated : 1; }

void tsx_peer_destroy(tsx_solver *s) {
  TsxPeer *p = s->peer;
  if (!p) return;
  for (int r = 0; r < kMaxRanks; ++r)
    if (p->opened[r] && p->box[r]) (void)hipIpcCloseMemHandle(p->box[r]);
  if (p->mine) (void)tsx_dev_free(p->mine);
  if (p->blkctr) (void)tsx_dev_free(p->blkctr);
  delete p;
  s->peer = nullptr;
}

// the error a kernel recorded (bounded waits), turned into a host error; call after a stream synchronisation
int tsx_peer_check(tsx_solver *s) {
  TsxPeer *p = s->peer;
  if (!p || !p->attached) return TSX_OK;
  PeerHdr h;
  HIPCHK(hipMemcpy(&h, p->mine, sizeof(h), hipMemcpyDeviceToHost));
  if (h.error == 0) return TSX_OK;
  static const char *what[] = {"", "send: no acknowledgement from the neighbour", "recv: no message from the neighbour",
                               "all-reduce: no contribution from a rank"};
  tsx_set_error(std::string("peer transport timed out -- ") + what[h.error & 3] + " (face / rank " + std::to_string(h.error_face) +
                ", waiting for " + std::to_string(h.error_want) + ", have " + std::to_string(h.error_have) + ")");
  return TSX_ERR_COMM;
}

int tsx_peer_exchange(tsx_solver *s, hipStream_t st, double *const send[4], double *const recv[4], size_t cx, size_t cy,
                      const int *done) {
  TsxPeer *p = s->peer;
  const TsxGeo &g = s->geo;
  const tsx_grid &gr = s->grid;
  const size_t cnt[4] = {g.wrap_x ? 0 : cx, g.wrap_x ? 0 : cx, g.wrap_y ? 0 : cy, g.wrap_y ? 0 : cy};
  const int nb[4] = {gr.neigh_w, gr.neigh_e, gr.neigh_s, gr.neigh_n};
  PeerXArgs a;
  memset(&a, 0, sizeof(a));
  a.mine = p->mine;
  a.cap = p->cap;
  a.data_off = p->data_off;
  a.ticks = p->ticks;
  a.heavy = p->heavy;
  a.done = done;
  size_t maxb = 0;
  for (int q = 0; q < 4; ++q) {
    a.bytes[q] = cnt[q] * sizeof(double);
    if (a.bytes[q] > p->cap) {
      tsx_set_error("peer exchange: message larger than the mailbox slots");
      return TSX_ERR_ARG;
    }
    if (a.bytes[q] == 0) continue;
    if (nb[q] < 0 || nb[q] >= gr.nranks || !p->box[nb[q]]) {
      tsx_set_error("peer exchange: neighbour rank out of range");
      return TSX_ERR_ARG;
    }
    a.remote[q] = p->box[nb[q]];
    a.src[q] = (const char *)send[q];
    a.dst[q] = (char *)recv[q];
    maxb = a.bytes[q] > maxb ? a.bytes[q] : maxb;
  }
  if (maxb == 0) return TSX_OK;
  // `done` skips an exchange on every rank alike, so the counters advance only when ... they must advance identically on both
  // sides: an exchange skipped by `done` is skipped by both kernels of both ranks, and the host counts it nowhere -- but the
  // host cannot see `done`.  The counters therefore count ISSUED exchanges and the kernels of a skipped exchange still publish
  // their numbers: see below (done is honoured only for the payload copy).
  int nblk = (int)((maxb + 32767) / 32768);
  nblk = nblk < 1 ? 1 : (nblk > 8 ? 8 : nblk);
  a.done = nullptr;  // see the comment above: sequence numbers must stay in step whatever `done` says
  for (int q = 0; q < 4; ++q)
    if (a.bytes[q]) {
      a.ackn[q] = p->rcvd[q];
      a.n[q] = ++p->sent[q];
    }
  a.blkctr = p->blkctr;
  hipLaunchKernelGGL(tsx_k_peer_send, dim3(nblk, 4), dim3(256), 0, st, a);
  for (int q = 0; q < 4; ++q)
    if (a.bytes[q]) a.n[q] = ++p->rcvd[q];
  a.blkctr = p->blkctr + 4;
  hipLaunchKernelGGL(tsx_k_peer_recv, dim3(nblk, 4), dim3(256), 0, st, a);
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

// ---- producers / consumers that move their messages themselves (tsx_peer_dev.hpp) -----------------------------------------
// A producer kernel stores its face messages straight into the neighbours' mailboxes (tsx_peer_send_begin / _end in its body):
// count them as sent and fill the kernel's view.  bytes[q]: payload through my face q (W, E, S, N; 0 = none).
int tsx_peer_prepare_send(tsx_solver *s, const size_t bytes[4], TsxPeerXArgs *a) {
  TsxPeer *p = s->peer;
  const tsx_grid &gr = s->grid;
  const int nb[4] = {gr.neigh_w, gr.neigh_e, gr.neigh_s, gr.neigh_n};
  memset(a, 0, sizeof(*a));
  a->mine = p->mine;
  a->cap = p->cap;
  a->data_off = p->data_off;
  a->ticks = p->ticks;
  a->heavy = p->heavy;
  a->blkctr = p->blkctr + 8;
  for (int q = 0; q < 4; ++q) {
    if (!bytes[q]) continue;
    if (bytes[q] > p->cap || nb[q] < 0 || nb[q] >= gr.nranks || !p->box[nb[q]]) {
      tsx_set_error("peer send: message larger than the mailbox slots, or neighbour rank out of range");
      return TSX_ERR_ARG;
    }
    a->bytes[q] = bytes[q];
    a->remote[q] = p->box[nb[q]];
    a->ackn[q] = p->rcvd[q];
    a->n[q] = ++p->sent[q];
  }
  return TSX_OK;
}
// A consumer kernel reads its face messages in place (tsx_peer_wait_faces in its body): count them as received.  slot[q]: where
// the message through my face q lands (null: none); w: what the kernel waits for.  The acknowledgement travels with this
// rank's next send (ackn), which by stream order follows the consumer.
int tsx_peer_expect(tsx_solver *s, const size_t bytes[4], TsxPeerWait *w, const void *slot[4]) {
  TsxPeer *p = s->peer;
  memset(w, 0, sizeof(*w));
  w->mine = p->mine;
  w->ticks = p->ticks;
  w->heavy = p->heavy;
  for (int q = 0; q < 4; ++q) {
    slot[q] = nullptr;
    if (!bytes[q]) continue;
    if (bytes[q] > p->cap) {
      tsx_set_error("peer receive: message larger than the mailbox slots");
      return TSX_ERR_ARG;
    }
    w->want[q] = ++p->rcvd[q];
    slot[q] = tsx_peer_data(p->mine, p->data_off, p->cap, q, (int)(w->want[q] & 1));
  }
  return TSX_OK;
}

int tsx_peer_flow_view(tsx_solver *s, const size_t bytes[4], int npass, TsxFlowPeer *v) {
  TsxPeer *p = s->peer;
  const tsx_grid &gr = s->grid;
  const int nb[4] = {gr.neigh_w, gr.neigh_e, gr.neigh_s, gr.neigh_n};
  memset((void *)v, 0, sizeof(*v));
  v->mine = p->mine;
  v->cap = p->cap;
  v->data_off = p->data_off;
  v->ticks = p->ticks;
  v->heavy = p->heavy;
  v->fctr = p->blkctr + 16;
  v->tag_off = p->tag_off;
  v->tag_edge = p->tag_edge;
  for (int q = 0; q < 4; ++q) {
    if (!bytes[q]) continue;
    if (bytes[q] > p->cap || nb[q] < 0 || nb[q] >= gr.nranks || !p->box[nb[q]]) {
      tsx_set_error("peer flow: message larger than the mailbox slots, or neighbour rank out of range");
      return TSX_ERR_ARG;
    }
    v->remote[q] = p->box[nb[q]];
    v->R0[q] = p->rcvd[q];
    v->S0[q] = p->sent[q];
    p->sent[q] += (unsigned long long)npass;
    p->rcvd[q] += (unsigned long long)(npass - 1);
  }
  return TSX_OK;
}

// v: nvals doubles on the device, summed over the ranks in place, on stream st
// the next all-reduce of this rank as a kernel sees it (tsx_peer_allreduce_wg in its body: tsx_k_scalar does the Krylov dots' sum
// between its reduction of the partial sums and the scalar algebra)
int tsx_peer_ar_args(tsx_solver *s, int nvals, TsxPeerArArgs *a) {
  TsxPeer *p = s->peer;
  ARGCHK(nvals >= 1 && nvals <= TSX_NSLOTS + 1, "peer all-reduce: too many values");
  memset(a, 0, sizeof(*a));
  a->mine = p->mine;
  for (int r = 0; r < s->grid.nranks; ++r) a->box[r] = p->box[r];
  a->rank = s->grid.rank;
  a->nranks = s->grid.nranks;
  a->nvals = nvals;
  a->n = ++p->ar_n;
  a->ar_off = p->ar_off;
  a->ticks = p->ticks;
  a->heavy = p->heavy;
  return TSX_OK;
}
int tsx_peer_allreduce(tsx_solver *s, hipStream_t st, double *v, int nvals, const int *done) {
  PeerArArgs a;
  int rc = tsx_peer_ar_args(s, nvals, &a);
  if (rc) return rc;
  (void)done;  // as for the exchange: the sequence stays in step on every rank
  hipLaunchKernelGGL(tsx_k_peer_allreduce, dim3(1), dim3(64), 0, st, a, v);
  HIPCHK(hipGetLastError());
  return TSX_OK;
}

// Collective self test of the attached transport: `rounds` face exchanges of varying length with a pattern that names sender,
// face, round and position, verified on the receiving side, and an all-reduce per round.  Meant to run right after attach
// (TSX_PEER_TIMEOUT_S short): a node where the mailboxes cannot be reached -- or are reached with stale data -- is found out here,
// and the caller falls back to RCCL (bench.py does).  *failed: 0 ok, else mismatching payload words + failed sums + 1e9 per
// expired wait on this rank; the caller reduces it over the ranks (the transport cannot be trusted to) and disables it everywhere.
extern "C" int tsx_comm_peer_selftest(tsx_solver *s, int rounds, double *failed) {
  ARGCHK(s && failed && rounds >= 1, "tsx_comm_peer_selftest: bad argument");
  if (!tsx_peer_ready(s)) {
    tsx_set_error("tsx_comm_peer_selftest: no peer transport attached");
    return TSX_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  TsxPeer *p = s->peer;
  const TsxGeo &g = s->geo;
  const tsx_grid &gr = s->grid;
  const size_t capd = p->cap / sizeof(double);
  double *buf[8] = {nullptr};
  TsxDevTmp guard[9];
  for (int q = 0; q < 8; ++q) {
    HIPCHK(guard[q].alloc(p->cap));
    buf[q] = guard[q].as<double>();
  }
  HIPCHK(guard[8].alloc(sizeof(unsigned long long) + 4 * sizeof(double)));
  unsigned long long *bad = guard[8].as<unsigned long long>();
  double *red = reinterpret_cast<double *>(bad + 1);
  HIPCHK(hipMemsetAsync(bad, 0, sizeof(unsigned long long), s->stream));
  double wrong_sums = 0.0;
  const unsigned long long keep_ticks = p->ticks;
  if (p->ticks > 300000000ull) p->ticks = 300000000ull;  // 3 s per wait is plenty for a test message
  struct Restore {
    TsxPeer *p;
    unsigned long long t;
    ~Restore() { p->ticks = t; }
  } restore{p, keep_ticks};
  for (int n = 0; n < rounds; ++n) {
    // lengths from a few doubles to the full slot; the same on every rank (the peers must agree on them like on any exchange)
    const size_t cx = 1 + (capd - 1) * (size_t)((n * 37) % 101) / 100, cy = 1 + (capd - 1) * (size_t)((n * 53 + 11) % 101) / 100;
    hipLaunchKernelGGL(tsx_k_peer_fill, dim3(64), dim3(256), 0, s->stream, buf[0], buf[1], buf[2], buf[3], (long long)cx, (long long)cy,
                       gr.rank, n);
    double *const send[4] = {buf[0], buf[1], buf[2], buf[3]};
    double *const recv[4] = {buf[4], buf[5], buf[6], buf[7]};
    int rc = tsx_peer_exchange(s, s->stream, send, recv, cx, cy, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(tsx_k_peer_verify, dim3(64), dim3(256), 0, s->stream, buf[4], buf[5], buf[6], buf[7],
                       g.wrap_x ? 0ll : (long long)cx, g.wrap_y ? 0ll : (long long)cy, gr.neigh_w, gr.neigh_e, gr.neigh_s, gr.neigh_n, n, bad);
    const double mine[3] = {(double)(gr.rank + 1), 1.0, (double)n};
    HIPCHK(hipMemcpyAsync(red, mine, sizeof(mine), hipMemcpyHostToDevice, s->stream));
    if ((rc = tsx_peer_allreduce(s, s->stream, red, 3, nullptr))) return rc;
    double got[3];
    HIPCHK(hipMemcpyAsync(got, red, sizeof(got), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    const double R = (double)gr.nranks;
    if (gr.nranks > 1 && (got[0] != R * (R + 1.0) / 2.0 || got[1] != R || got[2] != R * (double)n)) wrong_sums += 1.0;
    if (tsx_peer_check(s) != TSX_OK) {  // an expired wait: stop here, the counters of the peers are out of step now
      *failed = 1e9;
      return TSX_OK;
    }
  }
  unsigned long long hbad = 0;
  HIPCHK(hipMemcpy(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost));
  *failed = (double)hbad + wrong_sums;
  return TSX_OK;
}

// Full system-scope fences around the flags (1) or the light ordering that uncached mailboxes allow (0, tsx_peer_dev.hpp).
// Collective in effect: every rank must use the same setting only in so far as each wants correct data -- the two are
// compatible on the wire.
extern "C" int tsx_comm_peer_set_fences(tsx_solver *s, int heavy) {
  ARGCHK(s && s->peer, "tsx_comm_peer_set_fences: no peer transport");
  s->peer->heavy = heavy != 0;
  return TSX_OK;
}
// Back to the state after attach: counters, flags and the recorded error cleared, on this rank's mailbox and host side.  For a
// second self test after a failed one (whose expired waits leave the ranks' counters out of step).  The caller must put a host
// barrier over all ranks BEFORE the call (nobody still sends) and AFTER it (nobody sends into a mailbox that is being cleared).
extern "C" int tsx_comm_peer_reset(tsx_solver *s) {
  ARGCHK(s && s->peer, "tsx_comm_peer_reset: no peer transport");
  TsxPeer *p = s->peer;
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipDeviceSynchronize());
  // the whole mailbox: header, payload slots AND the flow kernel's tag area behind them (round 5 added it at tag_off; tags that kept the
  // previous run's large message numbers would satisfy `(int)(v - n) >= 0` at once after the counters restart -- ADVICE r5), and all
  // 32 workgroup counters (16..23 are the flow kernel's faces)
  HIPCHK(hipMemset(p->mine, 0, p->bytes));
  HIPCHK(hipMemset(p->blkctr, 0, sizeof(unsigned int) * 32));
  HIPCHK(hipDeviceSynchronize());
  for (int q = 0; q < 4; ++q) p->sent[q] = p->rcvd[q] = 0;
  p->ar_n = 0;
  if (s->flow_pr_shadow) memset((void *)s->flow_pr_shadow, 0xff, sizeof(TsxFlowPeer));  // the flow kernel's view is sent again
  s->pch_inplace = false;
  s->pcg_key = -1;
  return TSX_OK;
}

// give the transport up (after a failed self test on any rank): the solver falls back to RCCL or the callbacks
extern "C" int tsx_comm_peer_disable(tsx_solver *s) {
  ARGCHK(s, "tsx_comm_peer_disable: null");
  if (s->peer) s->peer->attached = false;
  // a rank whose agreement (tsx_pc_global_agree) went through before the transport failed must not keep it while a rank whose
  // all-reduce expired asks again over the next transport: everybody asks again
  s->pcg_key = -1;
  s->pch_inplace = false;
  return TSX_OK;
}

TSX_CODE_PROBE(peer)  // tsx_host.hpp: this unit's code object as it sits in device memory (diagnostics)
