/* vp8hip_multi.h -- more than one context at a time (part of the C ABI of libvp8hip.so; included by vp8hip.h): one frame's reference searches on several devices (vp8hip_shard_*), the process group of a GOP-sharded run (vp8hip_group_*, 8e(ii); RCCL inside the library), and batched contexts (vp8hip_batch_*: one launch per stage for up to eight GOP chunks). */
#ifndef VP8HIP_MULTI_H
#define VP8HIP_MULTI_H

#include "vp8hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- one frame's reference searches on different devices (reference: the three searches of a frame run
 * on three command queues and share only the current frame, inter_part.h:122-135, 201-236) --------------------------------
 * Every device holds a context with the same frames.  Per inter frame each calls vp8hip_inter_search with the references
 * it is to search (search_mask: bit 0 LAST, bit 1 GOLDEN, bit 2 ALTREF; the use_* flags are the frame's, as for
 * vp8hip_inter_transform), hands its vectors and costs over -- vp8hip_export_search / vp8hip_import_search copy one
 * reference's quarter-pel vector net (short2 per 8x8 block) and cost net (int per 8x8 block) to / from DEVICE memory of the
 * caller, e.g. the buffers of an RCCL all_gather -- and the device that has all of them calls vp8hip_inter_finish
 * (select_reference ... SSIM, filter mask: what vp8hip_inter_transform does after its searches), then the loop filter;
 * vp8hip_export_last copies the filtered reconstruction (tight planes) to device memory for the broadcast that makes it
 * the other devices' LAST (vp8hip_set_last_device there).  All asynchronous on the context's stream (vp8hip_stream). */
int vp8hip_inter_search(vp8hip_ctx *ctx, int prev_is_golden, int prev_is_altref, int use_golden, int use_altref, int search_mask);
int vp8hip_inter_finish(vp8hip_ctx *ctx, int use_golden, int use_altref);
int vp8hip_export_search(vp8hip_ctx *ctx, int ref, void *d_vectors, void *d_costs);
int vp8hip_import_search(vp8hip_ctx *ctx, int ref, const void *d_vectors, const void *d_costs);
int vp8hip_export_last(vp8hip_ctx *ctx, void *d_y, void *d_u, void *d_v);
/* The other end of vp8hip_export_last: tight planes in this device's memory become LAST exactly the way a receiving rank of
 * vp8hip_shard_share_last gets it (the same code: a free surface of the pool, adopted without touching GOLDEN / ALTREF; edges and
 * pyramid when the next frame begins).  vp8enc.cpp:395-401.  VP8HIP_ERR_STATE if no surface is free. */
int vp8hip_import_last(vp8hip_ctx *ctx, const void *d_y, const void *d_u, const void *d_v);

/* The same exchanges made by the library itself: RCCL (ncclBroadcast groups) on the context's stream, event-ordered with the
 * kernels around them, NO host synchronisation per frame -- what stands where the reference's three queues meet
 * (inter_part.h:122-135, 201-236, 263-266).  One context per process and device takes part:
 *   vp8hip_shard_unique_id   on ONE rank: 128 opaque bytes (ncclGetUniqueId) that the host hands to the other ranks by its own means
 *                            (a TCP store, MPI, a file);
 *   vp8hip_shard_init        every rank, the same id: the context's communicator (ncclCommInitRank; world <= 3, collective);
 *   vp8hip_shard_share_search  after vp8hip_inter_search: the vector and cost nets of every reference in used_mask (bit r) from
 *                            the rank that searched it (reference r belongs to rank r mod world) to all ranks, in place in the
 *                            nets vp8hip_inter_finish reads;
 *   vp8hip_shard_share_last  after rank root's vp8hip_loop_filter: root's filtered reconstruction becomes every rank's LAST;
 *   vp8hip_shard_max         barrier + maximum of one double over the ranks (a wall time); blocks.
 * The frame-type state machine runs identically on every rank (RefShardDriver in vp8oclenc_amd/ref_shard.py), so the GOLDEN /
 * ALTREF rotation needs no message.  VP8HIP_ERR_STATE before vp8hip_shard_init or for a member of a batch. */
#define VP8HIP_SHARD_ID_BYTES 128
int vp8hip_shard_unique_id(uint8_t id[VP8HIP_SHARD_ID_BYTES]);
int vp8hip_shard_init(vp8hip_ctx *ctx, const uint8_t id[VP8HIP_SHARD_ID_BYTES], int rank, int world);
int vp8hip_shard_rank(const vp8hip_ctx *ctx);    /* -1 before vp8hip_shard_init */
int vp8hip_shard_world(const vp8hip_ctx *ctx);   /*  0 before vp8hip_shard_init */
int vp8hip_shard_share_search(vp8hip_ctx *ctx, int used_mask);
int vp8hip_shard_share_last(vp8hip_ctx *ctx, int root);
int vp8hip_shard_max(vp8hip_ctx *ctx, double *value);

/* ---- the process group of a GOP-sharded run: one process per GPU, no data-path collective -------------------
 * GOP chunks are independent (intra_part.h:1091-1098).  What the ranks of a node still need from each other -- starting together,
 * the slowest rank's time, the finished frames in the hands of the one writer (the reference's single output file, encIO.h:1-30,
 * vp8enc.cpp:476-481) -- is here, over RCCL on a stream of the group's own, so that a host needs no GPU framework of its own for it:
 *   vp8hip_group_rendezvous   the 128 id bytes from rank 0 to the other ranks of ONE node through a file
 *                             <dir>/vp8hip-rdzv-<uid>-<key>, <dir> = $VP8HIP_RENDEZVOUS_DIR (taken as named), else $XDG_RUNTIME_DIR, else
 *                             /tmp/vp8hip-<uid> -- the last two only if they belong to this user and are closed to everybody else
 *                             (0700; VP8HIP_ERR_STATE otherwise): rank 0 removes a leftover of the same name, makes the id
 *                             (vp8hip_shard_unique_id) and writes the file atomically (O_EXCL | O_NOFOLLOW, rename); the others poll for
 *                             it up to timeout_s (VP8HIP_ERR_TIMEOUT) and accept only a regular file of this user no older than timeout_s
 *                             before their own start.  `key` names the run: the same string on every rank, different for runs alive on
 *                             the node at the same time.  A host with a store of its own (MPI, TCP) hands the id over itself and skips
 *                             this call;
 *   vp8hip_group_create       every rank, the same id (ncclCommInitRank: collective); key (may be NULL): rank 0 removes the rendezvous file.
 *                             Waits for the other ranks at most $VP8HIP_GROUP_TIMEOUT_S seconds (default 300, 0 = for ever), then
 *                             VP8HIP_ERR_TIMEOUT: a rank that died or read a wrong id must not hang the others for good (a process that
 *                             gets this error should exit; the abandoned init cannot be taken back);
 *   vp8hip_group_count        the ranks RCCL counts in the communicator (ncclCommCount);
 *   vp8hip_group_barrier / _max (maximum of one double: a wall time) / _all_gather (<= 4 KB per rank, in rank order, on every rank) /
 *   _broadcast (host memory of rank root to every rank);
 *   vp8hip_group_gather_bytes every rank's `bytes` of host memory (counts[] = every rank's size, from _all_gather, the same on all
 *                             ranks) end to end in rank order into dst on root.  One code path at every world size: every rank, root
 *                             included, sends (ncclSend), root receives from every rank, itself included (ncclRecv).
 * All calls are collective (every rank, same order) and block.  RCCL is loaded (dlopen) by the first call that needs it: the librccl.so.1
 * beside the HIP runtime the process runs on, $ROCM_PATH/lib, /opt/rocm/lib, the process's search path.
 * TEST HOOK, not configuration: $VP8HIP_RCCL_LIBRARY, if set, names the file that is dlopen'ed INSTEAD (tests/standin_rccl: several ranks
 * on one GPU over shared memory, which RCCL refuses).  It makes the library load and run an arbitrary shared object with the
 * process's rights: never set it in a production environment, and a set-uid / privileged host should clear it before the first call. */
typedef struct vp8hip_group vp8hip_group;
int vp8hip_group_rendezvous(const char *key, int rank, double timeout_s, uint8_t id[VP8HIP_SHARD_ID_BYTES]);
int vp8hip_group_create(vp8hip_group **out, int device_ordinal, const uint8_t id[VP8HIP_SHARD_ID_BYTES], int rank, int world, const char *key);
void vp8hip_group_destroy(vp8hip_group *g);
int vp8hip_group_rank(const vp8hip_group *g);
int vp8hip_group_world(const vp8hip_group *g);
int vp8hip_group_count(const vp8hip_group *g);
int vp8hip_group_barrier(vp8hip_group *g);
int vp8hip_group_max(vp8hip_group *g, double *value);
int vp8hip_group_all_gather(vp8hip_group *g, const void *mine, size_t bytes, void *all);
int vp8hip_group_broadcast(vp8hip_group *g, int root, void *buf, size_t bytes);
int vp8hip_group_gather_bytes(vp8hip_group *g, int root, const void *src, size_t bytes, void *dst, const uint64_t *counts);
int vp8hip_group_last_hip_error(const vp8hip_group *g);

/* ---- batched contexts: one launch per stage for up to four GOP chunks ---------------------------------------------------
 * The MI355X runs four to five kernels at once however many streams offer work, so sixteen contexts
 * that each launch their own kernels leave most of the part idle.  A batch groups up to VP8HIP_MAX_BATCH contexts of one
 * geometry, SSIM target and device; its stage calls do for every member what the per-context call of the same name does,
 * in ONE kernel launch per stage (same kernels, blockIdx.z = member), on one stream that the members share from then on --
 * so a member's own calls (vp8hip_intra_transform for a chunk's key frame, vp8hip_encode_frame, downloads) stay ordered
 * with the batched stages.  Arrays are indexed by member; `active` (may be NULL = all) leaves members out of a stage.
 * Everything of a batch runs on that one stream by default.  VP8HIP_BATCH_PREP in the environment (read once per process) gives
 * the head of a frame -- vp8hip_batch_set_current_device, vp8hip_batch_auto_segments and the new frame's pyramid, none of which
 * depends on the previous frame's reconstruction -- a second, low-priority stream beside the PREVIOUS frame's chain, which
 * waits for it where it starts: 1 = a stream per batch, 2 = one stream for all batches.  Off (0) by default: it measured 2-4 %
 * slower with the part full; vp8hip_batch_prep_mode() reports the mode in force.
 * vp8hip_batch_auto_segments launches nothing by itself: the scan rides in the quarter-pel search launch that
 * vp8hip_batch_inter_transform makes for the same frame (a launch of its own cost 4 % with the part full), and any
 * entry point that needs the segment data earlier launches it on its own first; VP8HIP_BATCH_SCAN_LAUNCH=1 = always on its own.
 * No reference counterpart: the reference codes one video on one in-order queue set. */
#define VP8HIP_MAX_BATCH 8
typedef struct vp8hip_batch vp8hip_batch;
int vp8hip_batch_create(vp8hip_batch **out, vp8hip_ctx *const *ctxs, int n);
void vp8hip_batch_destroy(vp8hip_batch *b);      /* the contexts stay, each back on its own stream; destroy a batch before its members */
int vp8hip_batch_set_current_device(vp8hip_batch *b, const int *active, const void *const *y, const void *const *u, const void *const *v);
/* The same from HOST memory -- the reference's own hand-over (clEnqueueWriteBuffer of the frame it has read, vp8enc.cpp:386-388) for a
 * batch: tight planes of the source size, copied on a stream of the batch's own into staging buffers (two per member, made on the first
 * call) and packed from there.  With page-locked planes (vp8hip_host_alloc) the copies are asynchronous and run beside what the batch
 * still has on the device (the previous frame's loop filter); pageable planes work as well (the runtime stages them).  Either way the
 * planes must stay unchanged until the NEXT vp8hip_batch_upload_current of this batch has returned, or its contexts are synchronised. */
int vp8hip_batch_upload_current(vp8hip_batch *b, const int *active, const uint8_t *const *y, const uint8_t *const *u, const uint8_t *const *v);
/* The NEXT frame's planes started on their way early (y[i] NULL: nothing for member i): the following vp8hip_batch_upload_current, given the
 * same planes, finds them in its staging buffers and copies nothing -- the copies had a whole frame's time instead of standing in front of
 * the frame's first launch.  The planes stay unchanged until that vp8hip_batch_upload_current has returned. */
int vp8hip_batch_prefetch_current(vp8hip_batch *b, const uint8_t *const *y, const uint8_t *const *u, const uint8_t *const *v);
int vp8hip_batch_auto_segments(vp8hip_batch *b, const int *active, const int *is_key_frame, const int32_t (*refqi)[4], int qi_min);
int vp8hip_batch_inter_transform(vp8hip_batch *b, const int *active, const int *prev_is_golden, const int *prev_is_altref,
                                 const int *use_golden, const int *use_altref);
/* vp8hip_intra_transform + vp8hip_prepare_filter_mask for the members whose frame is a KEY frame (active[i] != 0), the intra wavefronts
 * of all of them in one launch; vp8hip_batch_loop_filter for the same members follows.  (The members' segment data: vp8hip_batch_auto_segments
 * with is_key_frame set, or vp8hip_set_segments per member.) */
int vp8hip_batch_intra_transform(vp8hip_batch *b, const int *active);
int vp8hip_batch_loop_filter(vp8hip_batch *b, const int *active);
/* vp8hip_check_ssim_async for the active members (one launch; the verdicts ride in the following vp8hip_batch_loop_filter);
 * vp8hip_check_ssim_result per member afterwards */
int vp8hip_batch_check_ssim_async(vp8hip_batch *b, const int *active, const int32_t (*refqi)[4], int qi_min);
/* vp8hip_encode_frame_begin for the active members in the same nine launches (params[i] = member i's header parameters;
 * every member is then between _begin and _end: take each frame with vp8hip_encode_frame_end) */
int vp8hip_batch_encode_frame_begin(vp8hip_batch *b, const int *active, int num_partitions, const vp8hip_header_params *params);

#ifdef __cplusplus
}
#endif
#endif
