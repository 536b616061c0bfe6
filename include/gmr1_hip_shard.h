/*
 * gmr1_hip_shard.h -- the receive loop sharded over the GPUs of one node, from C: one process per GPU, RCCL over xGMI.
 *
 * What it replaces: nothing in the reference is distributed; its unit of work is one `gmr1_rx` process per carrier
 * file (src/gmr1_rx.c:897-975).  Carriers never interact (chains are independent copies of the channel state,
 * gmr1_rx.c:732-741), so N GPUs take N disjoint sets of carriers and the only exchanges are the two SURVEY.md 8(e) names:
 *
 *   scatter  the rank that holds the channelised capture (`root`) sends carrier a to rank a mod N, every transfer of
 *            the exchange posted in ONE ncclGroupStart / ncclGroupEnd: point-to-point ncclSend / ncclRecv, one peer per
 *            xGMI link, no ring;
 *   gather   per-carrier counts by one small ncclAllGather, then each rank's fixed 40-byte frame records to `root`
 *            by ncclSend / ncclRecv; `root` hands them back ordered by carrier, chain and order of emission -- the order
 *            gmr1_hip_rx_run over all carriers on one GPU returns.
 *
 * RCCL is looked up at run time (the copy already loaded into the process -- PyTorch's, say -- else librccl.so): the
 * library itself does not link it, and every entry point here returns -ENOSYS where there is none.
 */
#ifndef GMR1_HIP_SHARD_H
#define GMR1_HIP_SHARD_H

#include <stdint.h>

#include <gmr1_hip.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMR1_HIP_SHARD_ID_BYTES 128     /* an ncclUniqueId */

struct gmr1_hip_shard;                  /* this process's end of the node's communicator */

/* One rank (any) makes the id (ncclGetUniqueId); the host program hands the 128 bytes to the other ranks by its own
 * means (MPI_Bcast, a file, a socket). */
int gmr1_hip_shard_unique_id(void *id);

/* Collective over the `world` ranks: ncclCommInitRank on the calling thread's current device. */
int gmr1_hip_shard_create(struct gmr1_hip_shard **out, const void *id, int rank, int world);

/* Or adopt a communicator the program already has (an ncclComm_t, passed as void *); not destroyed with the shard. */
int gmr1_hip_shard_adopt(struct gmr1_hip_shard **out, void *nccl_comm, int rank, int world);

void gmr1_hip_shard_destroy(struct gmr1_hip_shard *sh);

/* gmr1_hip_rx_run_dev over the ranks of `sh`.  Collective: every rank calls it with the same n_arfcn, sps, offset[],
 * length[], arfcn[] (host arrays; samples; arfcn optional labels) and root.  `iq` is read on `root` only (device
 * memory holding every carrier: carrier a = iq[offset[a] .. offset[a] + length[a])); elsewhere it may be NULL.
 * `stream` (may be NULL) orders the exchanges and the kernels.  On `root`: out / n_records / status / n_chains as
 * gmr1_hip_rx_run_dev returns them for all carriers; on the other ranks they are not touched and may be NULL.
 * timing_ms (optional, 3 floats): wall time of scatter, receive loop, gather on this rank.
 * Returns 0 or -errno on the calling rank (-EIO: an RCCL or HIP call failed, gmr1_hip_last_error() has its text). */
int gmr1_hip_rx_run_sharded(struct gmr1_hip_shard *sh, void *stream, int root, int n_arfcn, int sps, const float *iq,
                            const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                            struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                            int32_t *status, int32_t *n_chains, float *timing_ms);

/* The same with the samples already where they are processed: every rank passes ITS OWN device memory as `iq`, holding
 * the carriers it owns (a % world == rank) at offset[a] .. offset[a] + length[a]; nothing is scattered (a recorder hands
 * every GPU its carriers directly: the scatter of a minute of 8 carriers costs about as much as the loop that follows).
 * offset[] / length[] / arfcn[] still have n_arfcn entries on every rank; a rank reads its own carriers' offsets only. */
int gmr1_hip_rx_run_sharded_resident(struct gmr1_hip_shard *sh, void *stream, int root, int n_arfcn, int sps, const float *iq,
                                     const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                                     struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                                     int32_t *status, int32_t *n_chains, float *timing_ms);

/* Failure behaviour of both (world > 1): a rank that fails locally -- bad pointer, allocation, its receive loop -- does
 * not leave the others inside a collective: the failure travels in a status word at three agreement points (before the
 * scatter, with the counts, before the records) and EVERY rank returns an error from the same point; an RCCL group that
 * was opened is always closed. */

#ifdef __cplusplus
}
#endif

#endif
