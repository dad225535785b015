/*
 * fql_amd.h -- C ABI of the MI355X-native FQL step engine (libfql_amd.so).
 *
 * This is the drop-in boundary for ONE path of zhouzypaul/fql: the FQL gradient step
 * (FQLAgent.update / total_loss / sample_actions / compute_flow_actions, agents/fql.py:94-171)
 * plus the batch source feeding it (utils/datasets.py:64-100).  Every entry point cites the
 * reference interface it replaces (paths relative to the reference repo).  The reference is pure
 * Python, so "what the reference's FFI would bind" is a ctypes stub: see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a negative FQL_E_* code on failure; the message is
 *     available from fql_last_error(handle) (or fql_last_error(NULL) for create failures).
 *     The Python mirror turns FQL_E_INVALID into ValueError and everything else into
 *     RuntimeError (the reference surfaces shape errors as Python exceptions at jit-trace time).
 *   - all tensors are fp32, contiguous, row-major.  Batch / observation / action / noise
 *     pointers may be DEVICE pointers (e.g. torch-ROCm tensor.data_ptr()) or HOST pointers;
 *     the engine detects which (hipPointerGetAttributes) and stages host memory itself.
 *     Pointers are borrowed for the duration of the call only (device pointers: until the work
 *     enqueued by the call has run on `stream`).
 *   - `stream` is a hipStream_t passed as void*.  NULL = the engine's own (non-blocking) stream: nothing
 *     orders it against work the caller has in flight elsewhere, so device pointers passed with a NULL
 *     stream must already be complete, and calls that write a DEVICE output on the NULL stream finish it
 *     before returning.  FQL_STREAM_LEGACY ((void*)1, the value of hipStreamLegacy) = the legacy default
 *     stream - what torch's *default* stream is (its handle reads 0 and cannot be told from NULL): pass it
 *     when the batch was produced on that stream.  Calls on one handle must be serialised by the caller;
 *     the engine never synchronises the host except where documented (host output pointers, get_* calls).
 *   - the engine owns params, Adam state, the target network and all workspaces in HBM.
 *     There is NO CPU fallback: without a HIP device fql_create fails with FQL_E_NODEVICE.
 */
#ifndef FQL_AMD_H
#define FQL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQL_ABI_VERSION 1

#define FQL_OK 0
#define FQL_E_INVALID (-1)  /* bad argument / shape mismatch against the config captured at create */
#define FQL_E_NODEVICE (-2) /* no HIP device / wrong architecture */
#define FQL_E_HIP (-3)      /* a HIP runtime call failed */
#define FQL_E_STATE (-4)    /* call order violated (e.g. update_end without update_begin) */
#define FQL_E_NOTFOUND (-5) /* unknown leaf name */

#define FQL_STREAM_LEGACY ((void*)1) /* == hipStreamLegacy: the legacy default (NULL) stream, e.g. torch's default stream */

#define FQL_MAX_HIDDEN 8
#define FQL_NUM_INFO 13

typedef struct fql_engine* fql_handle;

/* agents/fql.py:249-270 get_config() -- same keys, same defaults (see fql_default_config). */
typedef struct fql_config {
    int32_t obs_dim;                         /* config['ob_dims'][0]; state-based path only        */
    int32_t act_dim;                         /* config['action_dim']                               */
    int32_t num_actor_hidden;                /* len(actor_hidden_dims), <= FQL_MAX_HIDDEN          */
    int32_t actor_hidden[FQL_MAX_HIDDEN];    /* actor_hidden_dims  (512,512,512,512)               */
    int32_t num_value_hidden;                /* len(value_hidden_dims)                             */
    int32_t value_hidden[FQL_MAX_HIDDEN];    /* value_hidden_dims  (512,512,512,512)               */
    int32_t layer_norm;                      /* critic LayerNorm (True)                            */
    int32_t actor_layer_norm;                /* actor LayerNorm (False)                            */
    float lr;                                /* 3e-4                                               */
    float discount;                          /* 0.99                                               */
    float tau;                               /* 0.005                                              */
    float alpha;                             /* 300.0                                              */
    int32_t q_agg;                           /* 0 = 'mean', 1 = 'min'                              */
    int32_t flow_steps;                      /* 10                                                 */
    int32_t normalize_q_loss;                /* False                                              */
    int32_t batch_size;                      /* 256: rows per update on this device (multiple of 16) */
    int32_t precision;                       /* 0 = fp32 MFMA (exact fp32 fma chains; default).  2 = "bf16x3": every fp32 operand of the dense
                                                contractions is split into hi + lo bf16 and a b = a_hi b_hi + a_hi b_lo + a_lo b_hi runs on the
                                                bf16 matrix cores with fp32 accumulation (~2^-17 relative per product; parameters, activations,
                                                gradients and the optimizer stay fp32).  1 (plain bf16) is rejected: it cannot hold the 1e-4 loss bound */
    int32_t encoder;                         /* 0 = state observations; 1 = impala_small, 2 = impala (utils/encoders.py:104,106; stacks 16/32/32, 1 or 2 blocks): obs_dim is
                                                ignored, observations are uint8 [B, img_h, img_w, img_c] (agents/fql.py:196-202) */
    int32_t img_h, img_w, img_c;             /* image shape after frame stacking (e.g. 64, 64, 9); img_h, img_w multiples of 8 */
    int32_t reserved[3];
} fql_config;

/* The five random tensors one update draws (agents/fql.py:24,49-54,62-63,82,143-150; derivation in
 * SURVEY.md section 8a).  Any pointer may be NULL: that tensor then comes from the engine's own
 * counter-based RNG (Philox4x32-10 keyed by the create seed and the step counter). */
typedef struct fql_noise {
    const float* eps1; /* [B, act]  critic_loss: sample_actions(next_observations) noise */
    const float* x0;   /* [B, act]  actor_loss: x_0                                      */
    const float* t;    /* [B]       actor_loss: t ~ U[0,1)                               */
    const float* z;    /* [B, act]  actor_loss: distillation noises                      */
    const float* eps2; /* [B, act]  actor_loss: sample_actions(observations) noise (mse)  */
} fql_noise;

/* info13 order == the reference's info dict (agents/fql.py:39-44,85-92,103-108;
 * utils/flax_utils.py:151-157):
 *  0 critic/critic_loss 1 critic/q_mean 2 critic/q_max 3 critic/q_min 4 actor/actor_loss
 *  5 actor/bc_flow_loss 6 actor/distill_loss 7 actor/q_loss 8 actor/q 9 actor/mse
 *  10 grad/max 11 grad/min 12 grad/norm */
const char* fql_info_name(int i);

int fql_abi_version(void);
void fql_default_config(fql_config* cfg);
const char* fql_last_error(fql_handle h);

/* FQLAgent.create(seed, ex_observations, ex_actions, config)  agents/fql.py:173-246.
 * Builds the 4 networks (critic x2 members, target critic x2, actor_bc_flow, actor_onestep_flow),
 * Glorot-uniform kernels / zero biases / LN scale 1 bias 0 (utils/networks.py:9-11), target := critic
 * (agents/fql.py:241-242), Adam state zero, step = 1 (utils/flax_utils.py:81). */
int fql_create(const fql_config* cfg, uint64_t seed, fql_handle* out);
int fql_destroy(fql_handle h);
/* Re-size the per-step workspace for another batch size (params/optimizer state are kept). */
int fql_set_batch_size(fql_handle h, int batch_size);

/* Parameter / optimizer-state access by the reference's leaf path (utils/flax_utils.py:16-50 ModuleDict
 * naming), e.g. "modules_critic/value_net/Dense_0/kernel" (shape [2,in,out]),
 * "modules_actor_bc_flow/mlp/Dense_4/bias".  Host pointers, reference shapes, synchronous. */
int fql_num_leaves(fql_handle h);
int fql_leaf_info(fql_handle h, int index, char* name, int name_cap, int* ndim, int64_t shape[4]);
int fql_get_param(fql_handle h, const char* leaf, float* host_out, size_t n);
int fql_set_param(fql_handle h, const char* leaf, const float* host_in, size_t n);
/* which: 0 = Adam mu, 1 = Adam nu (optax ScaleByAdamState; utils/flax_utils.py:120-130) */
int fql_get_opt_state(fql_handle h, int which, const char* leaf, float* host_out, size_t n);
int fql_set_opt_state(fql_handle h, int which, const char* leaf, const float* host_in, size_t n);
/* optax count (0 at create) and TrainState.step (1 at create, utils/flax_utils.py:81) */
int fql_get_step(fql_handle h, int64_t* adam_count, int64_t* train_step);
int fql_set_step(fql_handle h, int64_t adam_count, int64_t train_step);

/* FQLAgent.update(batch)  agents/fql.py:122-133 (+ utils/flax_utils.py:120-159 grad stats, Adam;
 * agents/fql.py:113-120 Polyak from the PRE-step critic).  batch keys observations/actions/rewards/
 * masks/next_observations of utils/datasets.py:68-100 ('terminals' is unused by FQL).
 * `info13` may be NULL (async, read later with fql_read_info); a HOST info13 pointer makes the call
 * synchronous. */
int fql_update(fql_handle h, const float* observations, const float* actions, const float* rewards,
               const float* masks, const float* next_observations, int batch_size,
               const fql_noise* noise, float* info13, void* stream);
/* The two halves of fql_update for data-parallel runs: begin = forward + backward (gradients of the
 * LOCAL batch mean land in the buffer fql_grad_buffer returns), the caller all-reduces that buffer
 * (RCCL via torch.distributed), end = grad stats + Adam + Polyak.  fql_set_grad_scale(1/world) turns
 * the all-reduced SUM into the mean inside the optimizer kernel. */
int fql_update_begin(fql_handle h, const float* observations, const float* actions, const float* rewards,
                     const float* masks, const float* next_observations, int batch_size,
                     const fql_noise* noise, void* stream);
int fql_update_end(fql_handle h, float* info13, void* stream);
/* Overlapped variant of fql_update_begin for data-parallel runs: the same update enqueued as single-lane graphs on TWO
 * streams.  When the call returns, everything gradient bucket 0 (fql_grad_buckets: both critic members + the BC flow
 * actor) depends on is on `stream1`, everything bucket 1 (the one-step actor) depends on is on `stream0`: all-reduce
 * bucket 0 on stream1 while stream0 still runs the Euler chain, bucket 1 on stream0, make stream0 wait for stream1,
 * then fql_update_end(stream0).  Returns FQL_E_STATE if the engine has no two-lane program (then use fql_update_begin). */
int fql_update_begin_split(fql_handle h, const float* observations, const float* actions, const float* rewards,
                           const float* masks, const float* next_observations, int batch_size,
                           const fql_noise* noise, void* stream0, void* stream1);
int fql_update_from_dataset_begin_split(fql_handle h, const int64_t* idx, int batch_size, int64_t lo, int64_t hi,
                                        const fql_noise* noise, void* stream0, void* stream1);
/* The optimizer half on the two streams of the split begin: Adam (+ Polyak) of the critic and the BC flow - gradient bucket 0 - and the
 * refresh of the Euler chain's weight copies on stream1 (behind bucket 0's all-reduce, which the caller enqueued there), Adam of the
 * one-step actor - bucket 1 - and the step bookkeeping on stream0 (behind bucket 1's all-reduce); stream0 ends behind both.  Same
 * result as fql_update_end. */
int fql_update_end_split(fql_handle h, float* info13, void* stream0, void* stream1);
/* offsets / lengths (in floats) of the two gradient buckets inside the buffer fql_grad_buffer returns */
int fql_grad_buckets(fql_handle h, size_t offsets[2], size_t lengths[2]);
int fql_grad_buffer(fql_handle h, void** device_ptr, size_t* num_floats);
int fql_set_grad_scale(fql_handle h, float scale);
/* Data-parallel replicas are created with the SAME seed (identical initial parameters); stream_id (the rank) is mixed into the
 * key of the device RNG so that every rank draws different indices inside its shard and different noise (agents/fql.py:24,49-54,
 * 62-63 draw per-sample noise: ranks must not repeat each other's).  0 = the single-process stream. */
int fql_set_rng_stream(fql_handle h, uint64_t stream_id);

/* FQLAgent.total_loss(batch, grad_params=None)  agents/fql.py:94-111 -- the validation probe of
 * main.py:284: pure forward, no state change.  loss = critic_loss + actor_loss; info10 = the first 10
 * info entries.  Synchronous. */
int fql_total_loss(fql_handle h, const float* observations, const float* actions, const float* rewards,
                   const float* masks, const float* next_observations, int batch_size,
                   const fql_noise* noise, float* loss, float* info10, void* stream);

/* FQLAgent.sample_actions(observations, seed)  agents/fql.py:135-153: clip(onestep(obs, noise)).
 * `noise` [n, act] or NULL (engine RNG keyed by `seed`); `out` [n, act] device or host. */
int fql_sample_actions(fql_handle h, const float* observations, int n, const float* noise, uint64_t seed,
                       float* out, void* stream);
/* FQLAgent.compute_flow_actions(observations, noises)  agents/fql.py:155-171 (Euler, then clip). */
int fql_flow_actions(fql_handle h, const float* observations, const float* noises, int n, float* out,
                     void* stream);

/* Dataset.create / ReplayBuffer (utils/datasets.py:36-62,435-495): the transition arrays live in HBM.
 * `capacity` >= n rows are allocated (main.py:112-114 uses max(2M, n+1) for online runs). */
int fql_dataset_upload(fql_handle h, int64_t n, int64_t capacity, const float* observations,
                       const float* actions, const float* rewards, const float* masks,
                       const float* next_observations);
/* ReplayBuffer.add_transition (utils/datasets.py:483-491): ring insert of one row; returns new size via
 * fql_dataset_size. */
int fql_dataset_add(fql_handle h, const float* observation, const float* action, float reward, float mask,
                    const float* next_observation);
int fql_dataset_size(fql_handle h, int64_t* size, int64_t* pointer);
/* Dataset.sample(batch_size, idxs) + update (main.py:201,216) without leaving the device:
 * idx [B] int64 (device or host) or NULL = uniform indices from the engine RNG
 * (utils/datasets.py:64-66).  [lo, hi) restricts sampling to a shard of the rows (data-parallel:
 * rank r owns [r*n/W, (r+1)*n/W)); pass 0, 0 for the whole dataset. */
int fql_update_from_dataset(fql_handle h, const int64_t* idx, int batch_size, int64_t lo, int64_t hi,
                            const fql_noise* noise, float* info13, void* stream);
int fql_update_from_dataset_begin(fql_handle h, const int64_t* idx, int batch_size, int64_t lo, int64_t hi,
                                  const fql_noise* noise, void* stream);

/* Visual datasets (encoder != 0; utils/datasets.py:73-112): the uint8 frames [n, img_h, img_w, img_c / frame_stack] live in HBM;
 * a batch row is built on the device as Dataset.sample does: `frame_stack` frames ending at the sampled index, clamped to the first
 * index of its episode (initial_locs from `terminals`, utils/datasets.py:58-62), concatenated on the channel axis; next_observations
 * shifts by one and ends with next_frames[idx]; with probability p_aug per BATCH both are edge-padded by 3 and cropped at a
 * per-sample offset in [0, 6]^2 (Dataset.augment).  Device or host pointers. */
int fql_dataset_upload_frames(fql_handle h, int64_t n, const uint8_t* frames, const uint8_t* next_frames,
                              const float* actions, const float* rewards, const float* masks,
                              const float* terminals, int frame_stack, float p_aug);
/* fql_update_from_dataset for frames.  crop_froms: int32 [B, 2] (y, x) offsets as Dataset.augment draws them (3, 3 = identity),
 * or NULL = the engine's RNG (coin with p_aug, then offsets).  fql_update_from_dataset on a frames dataset == this with NULL. */
int fql_update_from_frames(fql_handle h, const int64_t* idx, const int32_t* crop_froms, int batch_size, int64_t lo,
                           int64_t hi, const fql_noise* noise, float* info13, void* stream);

/* JAX-compatible noise generated on the device (agents/fql.py:24,49-54,62-63,82,125,143-150): keys10 = the five uint32[2] keys the
 * reference's split chain derives for (eps1, x0, t, z, eps2) of one update (the chain itself is a dozen 2-word hashes: host side,
 * fql_amd/jax_prng.py fql_update_keys); partitionable = the jax_threefry_partitionable layout (0: original counter layout).
 * The tensors are written on `stream` into engine-owned device buffers and *out is filled with pointers to them: pass it as the
 * `noise` argument of the next update call on the same stream (valid until the next call of this function or the next update
 * given host noise).  No host tensor crosses PCIe. */
int fql_noise_from_jax_keys(fql_handle h, const uint32_t* keys10, int partitionable, int batch_size, fql_noise* out, void* stream);

/* ---- online fine-tuning hooks (main.py:217-272) ------------------------------------------------------------------------------
 * ReplayBuffer.create_from_initial_dataset(dataset, size) (utils/datasets.py:457-473, main.py:111-115): grow the uploaded dataset
 * (state or frames) to a ring of `capacity` >= current rows, new rows zero; size and pointer stay at the row count.  For frames
 * the episode starts are NOT recomputed on insert, as in the reference (Dataset.__init__ computes initial_locs once): rows
 * inserted later clamp their frame stack at the last episode start of the initial data. */
int fql_dataset_reserve(fql_handle h, int64_t capacity);
/* ReplayBuffer.add_transition (utils/datasets.py:483-491) for a frames dataset: one uint8 frame pair [img_h, img_w, img_c / frame_stack]
 * + action, reward, mask into the ring.  Device or host pointers. */
int fql_dataset_add_frames(fql_handle h, const uint8_t* frame, const uint8_t* next_frame, const float* action, float reward,
                           float mask);
/* ReplayBuffer.create(example_transition, size) (utils/datasets.py:441-455; main.py:106-109, --balanced_sampling): a second,
 * initially empty ring of `capacity` rows in HBM beside the training dataset; state rows or uint8 frames as the agent has them
 * (frames: after fql_dataset_upload_frames, whose frame_stack and p_aug it shares, main.py:117-120). */
int fql_replay_create(fql_handle h, int64_t capacity);
int fql_replay_add(fql_handle h, const float* observation, const float* action, float reward, float mask,
                   const float* next_observation);
int fql_replay_add_frames(fql_handle h, const uint8_t* frame, const uint8_t* next_frame, const float* action, float reward,
                          float mask);
int fql_replay_size(fql_handle h, int64_t* size, int64_t* pointer);
/* main.py:255-259 + :216: batch = concat(train_dataset.sample(B // 2), replay_buffer.sample(B // 2)); agent.update(batch), without
 * leaving the device.  idx_dataset / idx_replay: int64 [B / 2] each (device or host), or both NULL = uniform draws over the
 * dataset rows / the replay ring's `size` rows from the engine RNG.  crop_froms (frames only): int32 [B, 2] or NULL = one
 * augmentation coin per HALF (two sample() calls in the reference), offsets per row.  batch_size must be even and equal to the
 * workspace batch; FQL_E_INVALID while the replay ring is empty. */
int fql_update_balanced(fql_handle h, const int64_t* idx_dataset, const int64_t* idx_replay, const int32_t* crop_froms,
                        int batch_size, const fql_noise* noise, float* info13, void* stream);

/* Blocking read of the info of the last update (the reference reads lazily at log time, main.py:276). */
int fql_read_info(fql_handle h, float* info13_host);
/* Host wait for everything the engine has enqueued: its HIP stream AND the hardware queues of its own that updates with stream = NULL
 * run on (a hipStreamSynchronize / hipDeviceSynchronize of the caller does not see those queues; every other entry point of this header
 * waits for them by itself before it touches the engine's buffers).  `mode` (out, may be NULL): 1 if the last fql_update* call went to
 * the engine's own queues, 0 if it ran as a captured graph on a stream.  (jax.block_until_ready on the agent, main.py:216.) */
int fql_synchronize(fql_handle h, int* mode);
/* Lazy infos, as the reference's `agent, info = agent.update(batch)` returns them (device scalars nobody waits for until they are
 * logged, main.py:216,276): fql_info_enqueue, called right after an update on the same stream, snapshots that update's 13 scalars
 * asynchronously (no host synchronisation) and returns a ticket; fql_info_wait blocks until the snapshot has landed and copies it
 * out.  A ticket expires (FQL_E_STATE) once 64 later tickets have been taken. */
int fql_info_enqueue(fql_handle h, void* stream, uint64_t* ticket);
int fql_info_wait(fql_handle h, uint64_t ticket, float* info13_host);

/* Introspection for tests / bench: per-update kernel-launch count, algorithmic MAC count per update
 * (SURVEY.md section 8d figure), and the engine's HIP stream. */
int fql_stats(fql_handle h, int64_t* launches_per_update, int64_t* macs_per_update, int64_t* param_count);
void* fql_stream(fql_handle h);

#ifdef __cplusplus
}
#endif
#endif /* FQL_AMD_H */
