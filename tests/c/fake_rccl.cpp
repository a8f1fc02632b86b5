// fake_rccl.cpp -- a TEST DOUBLE for librccl: the ten entry points libtsx binds (tsx_api.hip: rccl_load) over POSIX shared memory,
// so that the RCCL transport's host code -- ncclCommInitRank, the second communicator of ncclCommSplit, the grouped
// ncclSend / ncclRecv of the face exchanges on comm_stream, the 3-double ncclAllReduce on the solver stream, and above all
// the ORDER the receives are posted in against the sends -- runs with REAL ranks on a one-GPU box, where librccl itself refuses
// two ranks on one device (tests/test_gpu_multirank.py::test_rccl_transport_with_two_ranks_on_one_device).
//
// Test infrastructure only.  Nothing in tenstream_amd/ links or names this file; libtsx reaches it only when the environment
// variable TSX_RCCL_LIB holds its path (tsx_api.hip), which only tests/test_gpu_multirank.py sets.
//
// Semantics kept from NCCL (what the product's code may rely on, no more):
//  * point-to-point messages between two ranks of one communicator are matched in ISSUE ORDER per (sender, receiver) pair
//    -- no tags; two ranks along a periodic axis are each other's W and E neighbour, so the receive order matters
//    (tsx_face_exchange_bufs posts E, W, N, S against sends W, E, S, N);
//  * operations between ncclGroupStart and ncclGroupEnd are issued together: no send of the group blocks a receive of it;
//  * every operation is ordered after the work queued before it on its stream, and the work queued after it sees its result
//    (here: the stream is synchronised, the copy is synchronous -- stronger than NCCL's stream order, never weaker);
//  * a communicator made by ncclCommSplit has message queues of its own.
// Not modelled: everything libtsx does not use (other datatypes than float64 for the all-reduce, other ops than sum, ...).
//
// build: hipcc -O1 -fPIC -shared -o libfake_rccl.so fake_rccl.cpp -lrt   (tests/test_gpu_multirank.py does it in tmp_path)
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

namespace {

constexpr int kMaxRanks = 8, kMaxComms = 4, kSlots = 4;
constexpr size_t kSlotBytes = 512 * 1024;
constexpr int kArMax = 16;

struct Queue {  // one direction of one pair of one communicator: a ring of messages
  std::atomic<uint64_t> head, tail;  // written / consumed
  uint64_t bytes[kSlots];
};
struct ArSlot {
  std::atomic<uint64_t> gen;
  double v[kArMax];
};
struct CommArea {
  Queue q[kMaxRanks][kMaxRanks];  // [src][dst]
  ArSlot ar[2][kMaxRanks];        // [generation parity][rank]
};
struct Shm {
  std::atomic<int> arrived;       // ranks that have attached (ncclCommInitRank's barrier)
  std::atomic<int> next_comm;     // communicators handed out by ncclCommSplit (every rank makes the same calls in the same order)
  int nranks;
  CommArea comm[kMaxComms];
  // payload behind it: [comm][src][dst][slot][kSlotBytes]
};
struct Comm {
  Shm *shm;
  char *payload;
  int id, rank, nranks;
  int nsplit;        // splits made from this communicator so far (names the child)
  uint64_t ar_gen;
};
struct Op {
  bool send;
  void *buf;
  size_t bytes;
  int peer;
  Comm *c;
  hipStream_t st;
};
thread_local int g_group = 0;
thread_local std::vector<Op> g_ops;
char g_err[256] = "fake rccl: ok";

size_t shm_bytes() { return sizeof(Shm) + (size_t)kMaxComms * kMaxRanks * kMaxRanks * kSlots * kSlotBytes; }
char *slot_ptr(Comm *c, int src, int dst, uint64_t n) {
  return c->payload + ((((size_t)c->id * kMaxRanks + src) * kMaxRanks + dst) * kSlots + (size_t)(n % kSlots)) * kSlotBytes;
}
bool wait_until(const std::atomic<uint64_t> &a, uint64_t want_ge, double seconds, const char *what) {
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (long spins = 0;; ++spins) {
    if (a.load(std::memory_order_acquire) >= want_ge) return true;
    if ((spins & 1023) == 1023) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > seconds) {
        snprintf(g_err, sizeof(g_err), "fake rccl: timed out waiting for %s", what);
        return false;
      }
      usleep(50);
    }
  }
}
int do_send(const Op &o) {
  Comm *c = o.c;
  if (o.bytes > kSlotBytes || o.peer < 0 || o.peer >= c->nranks) {
    snprintf(g_err, sizeof(g_err), "fake rccl: message of %zu bytes to rank %d (slots hold %zu)", o.bytes, o.peer, kSlotBytes);
    return 4;
  }
  Queue &q = c->shm->comm[c->id].q[c->rank][o.peer];
  const uint64_t n = q.head.load(std::memory_order_relaxed);
  if (n >= (uint64_t)kSlots && !wait_until(q.tail, n - kSlots + 1, 60.0, "a free slot (the receiver is more than four messages behind)")) return 1;
  if (hipMemcpy(slot_ptr(c, c->rank, o.peer, n), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  q.bytes[n % kSlots] = o.bytes;
  q.head.store(n + 1, std::memory_order_release);
  return 0;
}
int do_recv(const Op &o) {
  Comm *c = o.c;
  if (o.peer < 0 || o.peer >= c->nranks) return 4;
  Queue &q = c->shm->comm[c->id].q[o.peer][c->rank];
  const uint64_t n = q.tail.load(std::memory_order_relaxed);
  if (!wait_until(q.head, n + 1, 60.0, "a message (send / receive order mismatch between two ranks?)")) return 1;
  if (q.bytes[n % kSlots] != o.bytes) {  // NCCL would hang or corrupt: the double says what happened
    snprintf(g_err, sizeof(g_err), "fake rccl: rank %d expects %zu bytes from rank %d, the next message in issue order has %llu",
             c->rank, o.bytes, o.peer, (unsigned long long)q.bytes[n % kSlots]);
    return 5;
  }
  if (hipMemcpy(o.buf, slot_ptr(c, o.peer, c->rank, n), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
  q.tail.store(n + 1, std::memory_order_release);
  return 0;
}
int run_ops(std::vector<Op> &ops) {
  for (const Op &o : ops)
    if (hipStreamSynchronize(o.st) != hipSuccess) return 1;  // everything queued before the operation on its stream
  for (const Op &o : ops)
    if (o.send) {
      int rc = do_send(o);
      if (rc) return rc;
    }
  for (const Op &o : ops)
    if (!o.send) {
      int rc = do_recv(o);
      if (rc) return rc;
    }
  return 0;
}
int queue_or_run(const Op &o) {
  if (g_group > 0) {
    g_ops.push_back(o);
    return 0;
  }
  std::vector<Op> one(1, o);
  return run_ops(one);
}

}  // namespace

struct FakeUniqueId {
  char name[64];
  char pad[64];
};
static_assert(sizeof(FakeUniqueId) == 128, "ncclUniqueId is 128 bytes");

extern "C" {

const char *ncclGetErrorString(int) { return g_err; }

int ncclGetUniqueId(FakeUniqueId *id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->name, sizeof(id->name), "/tsx_fake_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
  int fd = shm_open(id->name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)shm_bytes()) != 0) {
    snprintf(g_err, sizeof(g_err), "fake rccl: cannot create the shared segment %s", id->name);
    return 2;
  }
  close(fd);  // (zero-filled by ftruncate: counters start at 0)
  return 0;
}

int ncclCommInitRank(void **comm, int nranks, FakeUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) {
    snprintf(g_err, sizeof(g_err), "fake rccl: %d ranks (at most %d)", nranks, kMaxRanks);
    return 4;
  }
  int fd = -1;
  for (int tries = 0; tries < 2000 && fd < 0; ++tries) {
    fd = shm_open(id.name, O_RDWR, 0600);
    if (fd < 0) usleep(1000);
  }
  if (fd < 0) {
    snprintf(g_err, sizeof(g_err), "fake rccl: cannot open the shared segment %s", id.name);
    return 2;
  }
  void *m = mmap(nullptr, shm_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return 2;
  Shm *shm = (Shm *)m;
  shm->nranks = nranks;
  Comm *c = new Comm();
  c->shm = shm;
  c->payload = (char *)m + sizeof(Shm);
  c->id = 0;
  c->rank = rank;
  c->nranks = nranks;
  c->nsplit = 0;
  c->ar_gen = 0;
  shm->arrived.fetch_add(1);
  for (long spins = 0; shm->arrived.load() < nranks; ++spins) {  // like ncclCommInitRank: collective
    usleep(200);
    if (spins > 300000) {
      snprintf(g_err, sizeof(g_err), "fake rccl: only %d of %d ranks arrived", shm->arrived.load(), nranks);
      return 1;
    }
  }
  if (rank == 0) shm_unlink(id.name);  // every rank has mapped it: the name can go, the memory stays until the last unmap
  *comm = c;
  return 0;
}

int ncclCommSplit(void *comm, int color, int key, void **newcomm, void *) {
  Comm *p = (Comm *)comm;
  (void)color;  // libtsx splits with one colour: every rank lands in the child, ranks ordered by key = rank
  if (key != p->rank) {
    snprintf(g_err, sizeof(g_err), "fake rccl: ncclCommSplit with key != rank is not modelled");
    return 4;
  }
  const int id = p->id * 2 + 1 + p->nsplit++;  // the same on every rank: they make the same calls in the same order
  if (id >= kMaxComms) {
    snprintf(g_err, sizeof(g_err), "fake rccl: more than %d communicators", kMaxComms);
    return 4;
  }
  Comm *c = new Comm(*p);
  c->id = id;
  c->nsplit = 0;
  c->ar_gen = 0;
  *newcomm = c;
  return 0;
}

int ncclCommDestroy(void *comm) {
  delete (Comm *)comm;  // (the mapping stays for the other communicators of the process; the test processes exit)
  return 0;
}

int ncclGroupStart() {
  ++g_group;
  return 0;
}
int ncclGroupEnd() {
  if (--g_group > 0) return 0;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) {
  if (dtype != 8) return 4;  // ncclFloat64: all libtsx sends
  Op o = {true, const_cast<void *>(buf), count * sizeof(double), peer, (Comm *)comm, st};
  return queue_or_run(o);
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) {
  if (dtype != 8) return 4;
  Op o = {false, buf, count * sizeof(double), peer, (Comm *)comm, st};
  return queue_or_run(o);
}

int ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, int dtype, int op, void *comm, hipStream_t st) {
  Comm *c = (Comm *)comm;
  if (dtype != 8 || op != 0 || count > (size_t)kArMax) {
    snprintf(g_err, sizeof(g_err), "fake rccl: all-reduce of %zu values of type %d, op %d is not modelled", count, dtype, op);
    return 4;
  }
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  const uint64_t g = ++c->ar_gen;
  ArSlot *mine = &c->shm->comm[c->id].ar[g & 1][c->rank];
  if (hipMemcpy(mine->v, sendbuf, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  mine->gen.store(g, std::memory_order_release);
  double sum[kArMax] = {0};
  for (int r = 0; r < c->nranks; ++r) {  // rank order: every rank gets the same bits.  No second barrier: a rank writes
    ArSlot *s = &c->shm->comm[c->id].ar[g & 1][r];  // generation g + 2 only after g + 1 completed, which needed everybody's g + 1,
    if (!wait_until(s->gen, g, 60.0, "an all-reduce contribution")) return 1;  // which they sent after reading g
    for (size_t k = 0; k < count; ++k) sum[k] += s->v[k];
  }
  if (hipMemcpy(recvbuf, sum, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}

}  // extern "C"
