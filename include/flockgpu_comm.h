/* flockgpu_comm.h -- multi-GPU entry of libflockgpu: the key-partitioned exchange of the reference's distributed plans
 *   filter / Partial aggregate  ->  RepartitionExec Hash([key], n)  ->  join / FinalPartitioned aggregate
 * (flock/src/distributed_plan/planner.rs:152-171, playground/src/distributed_plan/nexmark/q{3,5,8}.dag), which the
 * reference moves between Lambda functions as payloads, partition j to ring member j
 * (flock-function/src/aws/actor.rs:425-543; there is no collective library in the reference).  Here the ranks are GPUs of
 * one node and the repartition is ONE variable-size all-to-all per column buffer over xGMI, issued inside the library
 * (RCCL: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the ctx stream), so a Rust host binds it like any
 * other entry point.
 *
 * One rank = one flockgpu_ctx (one device, one stream, one thread at a time).  Every rank calls the same entry points
 * in the same order.  Every window of the schedule is striped across the ranks: rank r passes ITS rows of each window.
 *   partition rows by  dest = (fmix32(key) * n_ranks) >> 32  (count -> scan -> emit, shuffle.hip)
 *   -> take into send order  -> counts exchange (rows per (destination, window), bytes per Utf8 column)
 *   -> one all-to-all per column buffer  -> regroup (source, window) runs into windows  -> the single-GPU operator.
 * The result stays sharded by key: rank r returns the rows whose key it owns.  Which rank owns a key is unobservable
 * in the union (the reference hashes with ahash).
 */
#ifndef FLOCKGPU_COMM_H
#define FLOCKGPU_COMM_H

#include "flockgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct flockgpu_comm flockgpu_comm;

#define FLOCKGPU_COMM_ID_BYTES 128

/* One process per GPU (how bench.py and torchrun launch): rank 0 creates an id, the host ships its 128 bytes to the other
 * ranks through its own channel (a torch.distributed store, MPI, a socket ...), then EVERY rank calls init_rank --
 * collectively, as ncclCommInitRank requires.  The communicator runs on the ctx's device and stream. */
int flockgpu_comm_unique_id(uint8_t out_id[FLOCKGPU_COMM_ID_BYTES]);
int flockgpu_comm_init_rank(flockgpu_ctx *ctx, const uint8_t id[FLOCKGPU_COMM_ID_BYTES], int n_ranks, int rank,
                            flockgpu_comm **out);
/* One process, n_ranks ranks driven by n_ranks host threads (SURVEY.md section 8(b) "single-process, multi-device"):
 * out[r] is rank r's handle, used with rank r's ctx.  Ranks may sit on different devices or share one (how the exchange
 * is tested on a one-GPU box); buffers move with device-to-device copies, the ranks meet at host barriers. */
int flockgpu_comm_init_local(int n_ranks, flockgpu_comm **out /* n_ranks entries */);
/* n_ranks PROCESSES of one node, each with a ctx of its own -- on different devices or all on ONE (RCCL refuses several ranks per
 * device): the same call sequence as flockgpu_comm_init_rank (rank 0 draws the id, the host ships its 128 bytes, every rank calls
 * this: collective), but the bytes move through hipIpc mappings of the peers' send buffers and the counts / reductions / barriers
 * through a POSIX shared-memory segment named after the id.  Up to 16 ranks; the processes need HSA_ENABLE_IPC_MODE_LEGACY=0 where
 * the driver only offers dmabuf handles.  What it is for: the exchange protocol end to end across processes where no second GPU is
 * at hand (tests, a one-GPU box). */
int flockgpu_comm_init_ipc(flockgpu_ctx *ctx, const uint8_t id[FLOCKGPU_COMM_ID_BYTES], int n_ranks, int rank, flockgpu_comm **out);
void flockgpu_comm_destroy(flockgpu_comm *comm);
int flockgpu_comm_rank(const flockgpu_comm *comm);
int flockgpu_comm_size(const flockgpu_comm *comm);
/* "rccl" | "local" | "ipc" */
const char *flockgpu_comm_transport(const flockgpu_comm *comm);

/* ---- q5 as q5.dag runs it: HashAggregateExec(Partial) COUNT on this rank's rows, per 8192-row tile of every pane (a tile stands for one
 * input partition of the Partial stage: the same auction may leave a rank in several pairs, as it does in the reference) -> the (auction, count)
 * GROUPS are repartitioned on `auction` (a group moves once although its pane is in two hopping windows; the bids
 * themselves never cross the fabric) -> FinalPartitioned COUNT, MAX and the num = maxn join over the owned auctions ->
 * all-reduce(MAX) of the per-window maxima.  out: this rank's winners; win_max is the GLOBAL maximum per window, rows of
 * windows whose local maximum is below it are dropped; win_groups counts the groups this rank owns. */
int flockgpu_q5_hot_items_exchange(flockgpu_ctx *ctx, flockgpu_comm *comm, const flockgpu_bid_cols *bid,
                                   const flockgpu_windows *win, flockgpu_q5_result *out);

/* ---- q3 as planner.rs:152-171 stages it: stage 0 = FilterExec category = lit on this rank's auctions and state = a OR b OR ... on its
 * persons, then RepartitionExec Hash([seller]) / Hash([p_id]) of the rows the filters keep (only those travel); stage 1 = the join on the
 * rank that owns the key. */
int flockgpu_q3_join_exchange(flockgpu_ctx *ctx, flockgpu_comm *comm, const flockgpu_auction_cols *auction,
                              const flockgpu_windows *auction_win, const flockgpu_person_cols *person,
                              const flockgpu_windows *person_win, int64_t category_lit, const char *const *state_lits,
                              int n_state_lits, flockgpu_q3_result *out);

/* ---- q8 with the join shuffle of q8.dag: persons (p_id, name) repartitioned on p_id, auction sellers on seller, then
 * the local DISTINCT + DISTINCT + join. */
int flockgpu_q8_join_exchange(flockgpu_ctx *ctx, flockgpu_comm *comm, const flockgpu_person_cols *person,
                              const flockgpu_windows *person_win, const flockgpu_auction_cols *auction,
                              const flockgpu_windows *auction_win, flockgpu_q8_result *out);

/* Host barrier + stream synchronisation across the ranks (benchmark bracketing). */
int flockgpu_comm_barrier(flockgpu_ctx *ctx, flockgpu_comm *comm);

/* Failure semantics.  Every rank calls the same entry points in the same order with valid arguments (argument errors are
 * returned at once, before any collective).  A failure at RUN time on one rank -- a device error, a capacity limit, a
 * data-dependent refusal such as "rank 3 would receive more than 2^31 rows" -- is carried in the exchange's own messages
 * (the counts exchange before any data moves, the closing all-reduce), so EVERY rank returns: the failing rank its own status,
 * the others FLOCKGPU_ERR_PEER.  Nobody is left waiting for a rank that has gone.  A failure of the transport itself (RCCL /
 * device lost) marks the communicator dead on that rank: its later calls fail at once, a local group wakes its waiting peers at
 * once, RCCL peers come back with FLOCKGPU_ERR_PEER from their next wait -- as soon as RCCL reports the error, at the latest after
 * flockgpu_comm_set_timeout's deadline (reference: a failed function fails the whole query, flock-function/src/aws/actor.rs:425-543). */

/* Test hook for the failure semantics above: the next exchange call of this rank fails in its preparation (where = 1: a run-time
 * failure BEFORE the agreement -- every rank returns, the communicator stays usable) or in its data movement (where = 2: a
 * transport failure AFTER it -- this rank's communicator dies and wakes its local peers).  0 clears. */
int flockgpu_comm_inject_failure(flockgpu_comm *comm, int where);

/* Every host wait behind RCCL work (counts exchange, all-to-all, all-reduce) polls the stream together with the communicator's
 * asynchronous error state instead of blocking blindly: when RCCL reports an error, or ONE such wait lasts longer than `seconds`
 * (default 600; the clock starts when the wait begins and covers everything queued ahead of the collective on the ctx stream,
 * so a host that shares a busy stream with the ctx must size it for that), this rank aborts its communicator -- its own queued
 * sends / receives are cancelled, the stream drains -- and the call returns FLOCKGPU_ERR_PEER.  So a rank that died AFTER the
 * counts agreement (process killed, device lost) costs its peers at most the time-out, never a hang. */
int flockgpu_comm_set_timeout(flockgpu_comm *comm, double seconds);
/* The largest single transfer per (source, destination) pair; a longer run crosses in several rounds, each side walking the same
 * pieces.  Default (and maximum) 1 GiB.  Every rank of a communicator must set the same value; lowering it to a few KiB is how the
 * tests drive the multi-round path on small inputs. */
int flockgpu_comm_set_max_piece_bytes(flockgpu_comm *comm, int64_t bytes);

/* Per-phase timeline of the exchange calls on this rank's stream (HIP events at the phase boundaries: partial / stage-0
 * filters, partition + take, counts, all-to-all + regroup, final / join, all-reduce).  Off by default; totals accumulate over
 * the calls since the last reset and are read back like the kernel statistics (name, calls, total ms). */
int flockgpu_comm_phase_enable(flockgpu_comm *comm, int on);
int flockgpu_comm_phase_reset(flockgpu_comm *comm);
int flockgpu_comm_phase_read(flockgpu_comm *comm, flockgpu_kernel_stat *out, int cap, int *n);

#ifdef __cplusplus
}
#endif
#endif /* FLOCKGPU_COMM_H */
