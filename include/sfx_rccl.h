/*
 * sfx_rccl.h -- C ABI of libsfx_rccl: the per-substep halo exchange of slab worlds over RCCL (xGMI),
 * for hosts that have no torch (a Rust or C game loop).  BASELINE.json:5: "Large scenes are split across
 * the 8 GPUs of one node by slab domain decomposition, exchanging boundary bodies each substep with RCCL
 * over xGMI."  The reference has no counterpart (single process, single thread: SURVEY.md section 2).
 *
 * One process per GPU.  Each process owns one sfx_world (include/sfx.h) holding its slab's bodies plus
 * ghost copies of the neighbours' boundary bodies; sfx_halo_set has told the world which bodies it sends
 * (left block, then right block) and which ghosts it receives (from the left, then from the right).
 * This library only moves the world's two halo buffers: plain rccl.h ncclGroupStart / ncclSend /
 * ncclRecv / ncclGroupEnd on the world's stream, between sfx_halo_pack and sfx_halo_unpack.  There is no
 * collective on the data path: at most two peers, 32 bytes per boundary body.
 */
#ifndef SFX_RCCL_H
#define SFX_RCCL_H

#include <stddef.h>
#include <stdint.h>

#include "sfx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SFX_RCCL_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

typedef struct sfx_rccl_comm sfx_rccl_comm;

/* Rank 0 makes the id and hands it to the other ranks by any means the host has (a file, a pipe, MPI,
 * torch.distributed ...); then every rank calls sfx_rccl_comm_create (collective: it returns when all
 * `nranks` ranks have called it).  `device` is the HIP device of this rank's world. */
int sfx_rccl_unique_id(unsigned char id[SFX_RCCL_ID_BYTES]);
int sfx_rccl_comm_create(const unsigned char id[SFX_RCCL_ID_BYTES], int rank, int nranks, int device, sfx_rccl_comm** out);
void sfx_rccl_comm_destroy(sfx_rccl_comm* c); /* NULL is a no-op */
/* For a rank that has FAILED (a capacity overflow, a stale halo, a device error) while its peers may still be inside
 * a halo exchange: ends the communicator without waiting for them (ncclCommAbort) and frees it.  The peers' pending
 * operations then fail instead of waiting for ever; the host ends the run (tools/sfx_slab_run.cpp kills the other
 * ranks on the first bad exit). */
void sfx_rccl_comm_abort(sfx_rccl_comm* c);

/* One substep's exchange, enqueued on the world's stream (nothing synchronises, nothing is allocated):
 *   sfx_halo_pack(w); group { send left block -> `left`, send right block -> `right`, recv ghosts from `left`,
 *   recv ghosts from `right` }; sfx_halo_unpack(w).
 * left / right: neighbour ranks, -1 for none.  Counts are bodies (8 floats each) and must add up to what
 * sfx_halo_set was given: n_send_left + n_send_right = n_send, n_recv_left + n_recv_right = n_recv.  A rank may
 * name itself as a neighbour (send to self: used by the single-GPU tests); ANOTHER rank as both neighbours (a ring of
 * two slabs) is refused with SFX_ERR_UNSUPPORTED: its two blocks would pair up in issue order, the wrong way round.
 * The world must run on the communicator's device (SFX_ERR_INVALID otherwise).  A failing ncclSend / ncclRecv never
 * leaves the group open: the first error is returned after ncclGroupEnd. */
int sfx_rccl_halo_exchange(sfx_rccl_comm* c, sfx_world* w, int left, int right, uint32_t n_send_left,
                           uint32_t n_send_right, uint32_t n_recv_left, uint32_t n_recv_right);

/* The building block of the exchange, for drivers that lay their buffers out differently (several worlds per
 * process, the single-GPU self test): ONE group of an ncclSend of n_send floats to rank `to` and an ncclRecv of
 * n_recv floats from rank `from` (either may be -1 / 0 floats: left out), device pointers, enqueued on hip_stream. */
int sfx_rccl_sendrecv(sfx_rccl_comm* c, void* hip_stream, const float* send_dev, size_t n_send, int to, float* recv_dev,
                      size_t n_recv, int from);

/* Reporting figures of the whole scene: sums[0..n_sum) are added up and maxs[0..n_max)
 * maximised over all ranks, in place, through a small device staging buffer of the communicator; synchronises. */
int sfx_rccl_allreduce_stats(sfx_rccl_comm* c, double* sums, uint32_t n_sum, double* maxs, uint32_t n_max);

const char* sfx_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SFX_RCCL_H */
