// tsx_peer_dev.hpp -- device side of the peer transport (tsx_peer.hip): mailbox layout, bounded waits, the send / receive
// protocol as inline functions, so that a producer kernel can write its boundary records straight into the neighbours'
// mailboxes and a consumer kernel can read them in place (tsx_k_pcs_halo_send / tsx_k_pcs_rb, tsx_kernels_pcs.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "tsx_internal.hpp"  // TsxPeerWait, TsxPeerXArgs

#define TSX_PEER_HDR_BYTES 4096

struct TsxPeerHdr {
  unsigned long long seq[4];
  unsigned long long ack[4];
  int error;  // 0 ok; 1 send timed out waiting for an acknowledgement, 2 recv timed out, 3 all-reduce timed out
  int error_face;
  unsigned long long error_want, error_have;
};

// ---- ordering.  The mailboxes are UNCACHED device memory (hipDeviceMallocUncached): a store to them goes to memory (over xGMI
// for a neighbour's), a load comes from memory; no line of them is ever dirty or stale in an L2.  What the protocol needs is
// therefore only (i) a lane's payload stores have been acknowledged by memory before the sequence number is stored -- s_waitcnt
// vmcnt(0) -- and (ii) the payload loads are issued after the sequence number has been seen -- program order behind the poll.
// The language-level way to say this, a system-scope release fence + release store and an acquire load, also writes back and
// invalidates the whole L2 (buffer_wbl2 / buffer_inv sc0 sc1) every time: measured, a 128 x 64-column rank with itself as its
// neighbours solved in 7.7 ms with them against 5.8 ms without (fused send kernel + in-place reader, scripts/shard_study.py).  heavy = 1 (TSX_PEER_FENCES=1, or hostcomm.attach_peer_checked after a failed self test) keeps
// the full fences, for a platform where a peer's mapping of the mailbox turns out to be cached.
__device__ __forceinline__ unsigned long long tsx_peer_ld_acquire(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void tsx_peer_st_release(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long tsx_peer_peek(const unsigned long long *p, int heavy) {
  if (heavy) return tsx_peer_ld_acquire(p);
  const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("" ::: "memory");  // later loads stay behind the poll
  return v;
}
__device__ __forceinline__ void tsx_peer_post(unsigned long long *p, unsigned long long v, int heavy) {
  if (heavy) tsx_peer_st_release(p, v);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// this lane's stores so far have reached memory
__device__ __forceinline__ void tsx_peer_stores_done(int heavy) {
  if (heavy) __threadfence_system();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// spin until *p >= want; false after `ticks` of the constant-rate wall clock -- or at once when a wait of this rank has expired
// before (mine: my mailbox; its header records the first failure): after one expired wait the kernels still queued give up
// immediately instead of `ticks` each, so a lost rank costs every rank one timeout, not one per exchange
__device__ __forceinline__ bool tsx_peer_wait_ge(const unsigned long long *p, unsigned long long want, unsigned long long ticks,
                                                 unsigned long long *have, int heavy = 1, const char *mine = nullptr) {
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    const unsigned long long v = tsx_peer_peek(p, heavy);
    if (v >= want) return true;
    const bool failed_before =
        mine && __hip_atomic_load(&reinterpret_cast<const TsxPeerHdr *>(mine)->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
    if (failed_before || wall_clock64() - t0 > ticks) {
      *have = v;
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}
__host__ __device__ __forceinline__ char *tsx_peer_data(char *box, unsigned long long data_off, unsigned long long cap, int face,
                                                        int parity) {
  return box + data_off + ((size_t)face * 2 + parity) * cap;
}
__device__ __forceinline__ void tsx_peer_fail(char *mine, int code, int face, unsigned long long want, unsigned long long have) {
  TsxPeerHdr *h = reinterpret_cast<TsxPeerHdr *>(mine);
  if (atomicCAS(&h->error, 0, code) == 0) {
    h->error_face = face;
    h->error_want = want;
    h->error_have = have;
  }
}

// ---- sender, start of the kernel (every workgroup; lanes 0..3 take one face each): acknowledge what this rank has consumed
// so far (a kernel that read its messages in place leaves that to the next sender), then wait until the slot of message n is
// free (message n - 2 acknowledged).  Returns false after a timeout (recorded in the mailbox).  Contains a barrier.
__device__ __forceinline__ bool tsx_peer_send_begin(const TsxPeerXArgs &a) {
  __shared__ int ok_;
  if (threadIdx.x == 0) ok_ = 1;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q)  // (all acknowledgements first, then the waits: see tsx_peer_begin_both)
    if (blockIdx.x == 0 && (int)threadIdx.x == q && a.bytes[q] && a.ackn[q])
      tsx_peer_post(&reinterpret_cast<TsxPeerHdr *>(a.remote[q])->ack[q ^ 1], a.ackn[q], a.heavy);
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // static indices into the kernel arguments: lane q takes face q
    if ((int)threadIdx.x != q || !a.bytes[q]) continue;
    const TsxPeerHdr *h = reinterpret_cast<const TsxPeerHdr *>(a.mine);
    unsigned long long have = 0;
    if (a.n[q] > 2 && !tsx_peer_wait_ge(&h->ack[q], a.n[q] - 2, a.ticks, &have, a.heavy, a.mine)) {
      ok_ = 0;
      tsx_peer_fail(a.mine, 1, q, a.n[q] - 2, have);
    }
  }
  __syncthreads();
  return ok_ != 0;
}
// ---- a kernel that consumes its neighbours' previous messages in place AND sends the next ones (a red-black pass): both waits
// side by side -- lanes 0..3 poll the sequence numbers, lanes 4..7 the acknowledgements -- behind one barrier pair.
// recv: w.mine != null; send: `sending`.  Every lane must call.
__device__ __forceinline__ void tsx_peer_begin_both(const TsxPeerWait &w, bool need_recv, bool sending, const TsxPeerXArgs &a,
                                                    bool need_send) {
  const bool rcv = w.mine != nullptr;
  const int any = __syncthreads_or(((rcv && need_recv) || (sending && need_send)) ? 1 : 0);
  if (!any && !(sending && blockIdx.x == 0)) return;
  // every acknowledgement is posted before any lane of the wave starts to wait: the lanes of a wave run their waits one after the
  // other, and the acknowledgement one of them waits for may be the one a later lane (of this rank itself with self neighbours,
  // of the neighbour's wave in the same position otherwise) has yet to post
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (sending && blockIdx.x == 0 && (int)threadIdx.x == 4 + q && a.bytes[q] && a.ackn[q])
      tsx_peer_post(&reinterpret_cast<TsxPeerHdr *>(a.remote[q])->ack[q ^ 1], a.ackn[q], a.heavy);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (rcv && any && (int)threadIdx.x == q && w.want[q]) {
      unsigned long long have = 0;
      if (!tsx_peer_wait_ge(&reinterpret_cast<const TsxPeerHdr *>(w.mine)->seq[q], w.want[q], w.ticks, &have, w.heavy, w.mine))
        tsx_peer_fail(w.mine, 2, q, w.want[q], have);
    }
    if (sending && (int)threadIdx.x == 4 + q && a.bytes[q]) {
      unsigned long long have = 0;
      if (any && a.n[q] > 2 &&
          !tsx_peer_wait_ge(&reinterpret_cast<const TsxPeerHdr *>(a.mine)->ack[q], a.n[q] - 2, a.ticks, &have, a.heavy, a.mine))
        tsx_peer_fail(a.mine, 1, q, a.n[q] - 2, have);
    }
  }
  __syncthreads();
}
// ---- sender, end of the kernel (every workgroup, after its last payload store): the workgroup that finishes last publishes the
// sequence numbers.  ctr: one counter for the kernel (reset here).  Contains a barrier.
__device__ __forceinline__ void tsx_peer_send_end(const TsxPeerXArgs &a, unsigned int *ctr, unsigned int nblocks) {
  tsx_peer_stores_done(a.heavy);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = atomicAdd(ctr, 1u);
    if (prev + 1 == nblocks) {  // every workgroup's stores have reached memory: publish
      *ctr = 0;
      if (a.heavy) __threadfence_system();
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (a.bytes[q]) tsx_peer_post(&reinterpret_cast<TsxPeerHdr *>(a.remote[q])->seq[q ^ 1], a.n[q], a.heavy);
    }
  }
}
// ---- consumer in place: lanes 0..3 wait for the faces this workgroup reads (need[q]); every lane must call (barrier inside).
// After a timeout the kernel goes on with whatever the slot holds; the host reports TSX_ERR_COMM at its next check.
__device__ __forceinline__ void tsx_peer_wait_faces(const TsxPeerWait &w, bool needW, bool needE, bool needS, bool needN) {
  const int any = __syncthreads_or((needW ? 1 : 0) | (needE ? 2 : 0) | (needS ? 4 : 0) | (needN ? 8 : 0));
  // __syncthreads_or returns non-zero if any lane's predicate is non-zero, not the OR of the values: poll all active faces
  if (!any) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if ((int)threadIdx.x != q || !w.want[q]) continue;
    unsigned long long have = 0;
    if (!tsx_peer_wait_ge(&reinterpret_cast<const TsxPeerHdr *>(w.mine)->seq[q], w.want[q], w.ticks, &have, w.heavy, w.mine))
      tsx_peer_fail(w.mine, 2, q, w.want[q], have);
  }
  __syncthreads();
}

// ---- all-reduce through the mailboxes (see tsx_peer.hip): contribution slots of 64 bytes, [parity][rank]
struct TsxPeerArSlot {  // one line per contribution
  double v[TSX_NSLOTS + 1];
  unsigned long long seq;
  unsigned long long pad[8 - (TSX_NSLOTS + 1) - 1];
};
static_assert(sizeof(TsxPeerArSlot) == 64, "all-reduce slot is one 64-byte line");
#define TSX_PEER_MAX_RANKS_DEV 16
struct TsxPeerArArgs {
  char *mine;
  char *box[TSX_PEER_MAX_RANKS_DEV];
  int rank, nranks, nvals, heavy;  // nranks <= 1: no all-reduce
  unsigned long long n, ar_off, ticks;
};
// v: a.nvals doubles in device memory, summed over the ranks in place, in rank order.  Called by every thread of ONE workgroup of
// at least 64 threads (barriers inside); lane r < nranks talks to rank r.
__device__ __forceinline__ void tsx_peer_allreduce_wg(const TsxPeerArArgs &a, double *__restrict__ v) {
  const int r = threadIdx.x;
  const int par = (int)(a.n & 1);
  __shared__ int bad_;
  if (r == 0) bad_ = 0;
  __syncthreads();
  if (r < a.nranks) {
    TsxPeerArSlot *slot = reinterpret_cast<TsxPeerArSlot *>(a.box[r] + a.ar_off) + (size_t)par * TSX_PEER_MAX_RANKS_DEV + a.rank;
    for (int k = 0; k < a.nvals; ++k) slot->v[k] = v[k];
    tsx_peer_stores_done(a.heavy);
    tsx_peer_post(&slot->seq, a.n, a.heavy);
    const TsxPeerArSlot *in = reinterpret_cast<const TsxPeerArSlot *>(a.mine + a.ar_off) + (size_t)par * TSX_PEER_MAX_RANKS_DEV + r;
    unsigned long long have = 0;
    if (!tsx_peer_wait_ge(&in->seq, a.n, a.ticks, &have, a.heavy, a.mine)) {
      bad_ = 1;
      tsx_peer_fail(a.mine, 3, r, a.n, have);
    }
  }
  __syncthreads();
  if (r == 0 && !bad_) {
    const TsxPeerArSlot *in = reinterpret_cast<const TsxPeerArSlot *>(a.mine + a.ar_off) + (size_t)par * TSX_PEER_MAX_RANKS_DEV;
    for (int k = 0; k < a.nvals; ++k) {
      double sum = 0.0;
      for (int q = 0; q < a.nranks; ++q) sum += __hip_atomic_load(&in[q].v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v[k] = sum;
    }
  }
  __syncthreads();
}
