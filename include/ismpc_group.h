/*
 * ismpc_group.h -- multi-GPU layer of the C ABI: a batch of independent gait instances sharded over the GPUs of one node,
 * with the path's ONE collective -- an all-gather of the 80-byte output records, RCCL over xGMI -- behind the same plain-C
 * boundary as include/ismpc.h and include/ismpc_a.h (BASELINE north_star: "sharded across the 8 GPUs of one
 * node with a single RCCL all-gather over xGMI to collect trajectories", "host code stays C++ calling HIP through a thin
 * C-ABI").
 *
 * Reference interface: the reference constructs ONE MPCSolver and calls it from one thread (Controller.cpp:105-106,
 * :346-348); a parameter study re-runs that program per instance.  A group is that construction done once per GPU
 * (every device holds its own copy of the read-only tables: ismpc_create / ismpc_a_create per device) plus one RCCL
 * communicator.  Two ways to build one, same entry points afterwards:
 *
 *   ismpc_group_create        one process drives n devices ("1 process / 8 streams"): ncclCommInitAll
 *   ismpc_group_create_rank   one process per GPU (the launcher's ranks): rank 0 calls ismpc_group_unique_id, the caller
 *                             distributes the 128 bytes any way it likes (MPI, a file, torch.distributed), every rank
 *                             calls ismpc_group_create_rank: ncclCommInitRank
 *
 * Sharding: instance i of a batch belongs to the rank whose contiguous range [first, first + count) holds it
 * (ismpc_shard_range: the first batch % world ranks get one more).  No instance ever needs another one's data, so the
 * only exchange is the all-gather of the results: equal shards take ONE ncclAllGather, in place in the gathered buffer
 * (each device's kernel writes its shard straight to its final position); ragged shards take one fused group of per-rank
 * ncclBroadcast calls (an all-gather-v).  The collective runs on a side stream behind an event, so with the two-buffer
 * device entry point step k's collective overlaps step k+1's kernel.
 *
 * RCCL is bound at run time (dlopen of the copy already mapped into the process, else librccl.so.1): the library loads and
 * every single-GPU entry point works on a machine without RCCL; the group entry points then fail with ISMPC_E_NO_DEVICE.
 * Conventions as ismpc.h (plain C, int status, caller-owned buffers, no CPU fallback).
 */
#ifndef ISMPC_GROUP_H
#define ISMPC_GROUP_H

#include "ismpc.h"
#include "ismpc_a.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ISMPC_UNIQUE_ID_BYTES 128

typedef struct ismpc_group ismpc_group;       /* Formulation B: n x (ismpc_handle + streams) + communicator   */
typedef struct ismpc_a_group ismpc_a_group;   /* Formulation A: n x (ismpc_a_handle + streams) + communicator */

/* Contiguous shard of `rank` among `world` ranks (host arithmetic, no GPU): the first (batch % world) ranks hold one
 * instance more.  Returns ISMPC_E_INVALID for batch < 0, world < 1 or rank outside [0, world).                       */
int ismpc_shard_range(int batch, int rank, int world, int* first, int* count);

/* 128 opaque bytes naming a new communicator (ncclGetUniqueId): rank 0 makes them, every rank passes the same bytes. */
int ismpc_group_unique_id(void* id128);

/* ---- Formulation B (MPCSolver::solve, include/ismpc.h) ---------------------------------------------------------- */
/* One process, n devices: devices[k] becomes rank k.  Arguments as ismpc_create (MPCSolver::MPCSolver,
 * MPCSolver.cpp:5-200), once per device.                                                                            */
int ismpc_group_create(const ismpc_params* params, const double* ftsp, int rows,
                       const int* devices, int n, ismpc_group** out);
/* One process per GPU: this process is `rank` of `world` and drives `device`.                                       */
int ismpc_group_create_rank(const ismpc_params* params, const double* ftsp, int rows, int device,
                            const void* id128, int rank, int world, ismpc_group** out);
void ismpc_group_destroy(ismpc_group* g);

int ismpc_group_world(const ismpc_group* g);            /* ranks of the communicator AS RCCL REPORTS THEM (ncclCommCount) */
int ismpc_group_local(const ismpc_group* g);            /* devices this process drives (n, or 1 in rank mode)             */
int ismpc_group_rank(const ismpc_group* g, int local);  /* rank of local device `local`                                   */
ismpc_handle* ismpc_group_handle(ismpc_group* g, int local);   /* its plain handle (owned by the group)                  */

/* One MPCSolver::solve per instance for `batch` instances given as host records: every local device copies in ITS
 * shard, runs it, the all-gather collects all shards on every device, and out_host receives all `batch` records (in rank
 * mode every rank passes the same in_host and gets the complete out_host).  Returns when done.  Records are
 * byte-identical to ismpc_solve_batch_device of the same shard on a plain handle.                                    */
int ismpc_group_solve_batch(ismpc_group* g, int batch, const ismpc_tick_in* in_host, ismpc_tick_out* out_host);

/* The same on device memory, asynchronous and double-buffered: in_dev[l] points at local device l's SHARD (count_l
 * records, on that device); the gathered records of the step land in the group's buffer `buf` (0 or 1) on every device.
 * Before the kernel of a step overwrites buffer `buf`, the collective that last read it has completed (stream-side
 * wait); nothing else is waited for, so step k's all-gather (side stream) overlaps step k+1's kernel (launch stream).
 * ismpc_group_result_device: the gathered buffer (batch records) of local device l; valid after ismpc_group_sync, or
 * on a caller stream that waits for it with ismpc_group_wait_on (a hipStream_t).                                     */
int ismpc_group_step_device(ismpc_group* g, int batch, const ismpc_tick_in* const* in_dev, int buf);
int ismpc_group_result_device(ismpc_group* g, int local, int buf, ismpc_tick_out** out_dev);
int ismpc_group_wait_on(ismpc_group* g, int local, int buf, void* stream);
int ismpc_group_sync(ismpc_group* g);                   /* drains the launch and side streams of every local device      */
/* Inputs produced on a caller stream (a hipStream_t; NULL = the default stream): the group's launch stream of local device
 * `local` starts nothing later before the work enqueued so far on `stream` has completed (event + stream-side wait).   */
int ismpc_group_order_after(ismpc_group* g, int local, void* stream);
/* Sizes the device buffers for batches up to max_batch now (otherwise they grow, with a synchronisation, inside a call). */
int ismpc_group_reserve(ismpc_group* g, int max_batch);

/* ---- Formulation A (the MATLAB generators' tick, include/ismpc_a.h) --------------------------------------------- */
int ismpc_a_group_create(const ismpc_a_params* p, const double* center, const int* devices, int n, ismpc_a_group** out);
int ismpc_a_group_create_rank(const ismpc_a_params* p, const double* center, int device,
                              const void* id128, int rank, int world, ismpc_a_group** out);
void ismpc_a_group_destroy(ismpc_a_group* g);
int ismpc_a_group_world(const ismpc_a_group* g);
int ismpc_a_group_local(const ismpc_a_group* g);
int ismpc_a_group_rank(const ismpc_a_group* g, int local);
ismpc_a_handle* ismpc_a_group_handle(ismpc_a_group* g, int local);
int ismpc_a_group_add_plan(ismpc_a_group* g, const double* center);      /* ismpc_a_add_plan on every device   */
int ismpc_a_group_set_precision(ismpc_a_group* g, int fp32);              /* ismpc_a_set_precision on every one  */
/* One tick for `batch` instances given as host records.  state_host (batch records) is read and, for the shards of
 * THIS process, updated in place (all of it in a one-process group); inst_host is NULL or batch per-instance records
 * (ismpc_a_tick_batch_inst_device), push_host NULL or batch x 2 doubles; out_host receives all batch output records.  */
int ismpc_a_group_tick_batch(ismpc_a_group* g, int batch, ismpc_a_state* state_host, const ismpc_a_inst* inst_host,
                             const double* push_host, ismpc_a_out* out_host);
/* Device memory, asynchronous, double-buffered (as ismpc_group_step_device): state_dev[l] / inst_dev[l] / push_dev[l]
 * are local device l's shard (inst_dev and push_dev may be NULL, or hold NULL entries).                               */
int ismpc_a_group_step_device(ismpc_a_group* g, int batch, ismpc_a_state* const* state_dev,
                              const ismpc_a_inst* const* inst_dev, const double* const* push_dev, int buf);
int ismpc_a_group_result_device(ismpc_a_group* g, int local, int buf, ismpc_a_out** out_dev);
int ismpc_a_group_wait_on(ismpc_a_group* g, int local, int buf, void* stream);
int ismpc_a_group_sync(ismpc_a_group* g);
int ismpc_a_group_order_after(ismpc_a_group* g, int local, void* stream);
int ismpc_a_group_reserve(ismpc_a_group* g, int max_batch);

const char* ismpc_group_last_error(void);     /* thread-local, never NULL: errors of the entry points of this header */
/* RCCL version the process bound (ncclGetVersion), 0 when RCCL could not be loaded.                                  */
int ismpc_group_rccl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ISMPC_GROUP_H */
