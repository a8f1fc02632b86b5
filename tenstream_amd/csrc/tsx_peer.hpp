// tsx_peer.hpp -- device-resident peer transport (tsx_peer.hip): mailboxes in IPC-shared fine-grained device memory
#pragma once
#include <hip/hip_runtime.h>

#include "tsx_internal.hpp"

#define TSX_PEER_MAX_RANKS 16    // node-local: one rank per GPU of a node (8 on MI355X nodes)

bool tsx_peer_ready(const tsx_solver *s);
void tsx_peer_destroy(tsx_solver *s);
int tsx_peer_check(tsx_solver *s);  // after a synchronisation: did a bounded wait expire?
// the four face buffers W, E, S, N (cx / cy doubles per x / y face), entirely on stream st
int tsx_peer_exchange(tsx_solver *s, hipStream_t st, double *const send[4], double *const recv[4], size_t cx, size_t cy,
                      const int *done);
// nvals (<= TSX_NSLOTS + 1) doubles in device memory, summed over the ranks in place in rank order, on stream st
int tsx_peer_allreduce(tsx_solver *s, hipStream_t st, double *v, int nvals, const int *done);

// kernels that move their messages themselves (tsx_peer_dev.hpp): host-side bookkeeping of one message per face
struct TsxPeerXArgs;
struct TsxPeerWait;
int tsx_peer_prepare_send(tsx_solver *s, const size_t bytes[4], TsxPeerXArgs *a);
int tsx_peer_expect(tsx_solver *s, const size_t bytes[4], TsxPeerWait *w, const void *slot[4]);
struct TsxPeerArArgs;
int tsx_peer_ar_args(tsx_solver *s, int nvals, TsxPeerArArgs *a);
// the flow kernel (tsx_k_pcs_flow FPEER) sends and consumes `npass` messages per face by itself: its view of the transport, the
// counters advanced by what it will have sent (npass) and consumed beyond the message already expected (npass - 1)
int tsx_peer_flow_view(tsx_solver *s, const size_t bytes[4], int npass, TsxFlowPeer *v);
// rank processes of this job whose mailbox lives on this rank's device, this rank included (1: a device of its own)
int tsx_peer_colocated(const tsx_solver *s);
