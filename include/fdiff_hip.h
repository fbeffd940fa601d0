/*
 * fdiff_hip.h -- C ABI of libfdiff_hip.so, the MI355X (gfx950) engine for the
 * score-matching hot path of JonathanCrabbe/FourierDiffusion.
 *
 * The reference has no FFI of its own for this path: it sits behind plain Python
 * classes.  Each entry point below therefore names the reference
 * *Python* interface it replaces (file:line relative to /root/reference); the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and lives in
 * fourierdiffusion_amd/_C.py.
 *
 * Conventions
 *   - extern "C"; plain pointers and sizes only (no torch types).
 *   - every function returns 0 on success, <0 on error (FD_ERR_*); the message is
 *     available from fd_last_error(ctx).
 *   - all data pointers are CALLER-OWNED DEVICE pointers (float32, row-major,
 *     contiguous (B,T,C)) unless marked "host".  bf16 exists only inside the engine.
 *   - asynchronous on the passed hipStream_t (void* to keep the header toolchain-free);
 *     no host synchronisation inside any call unless stated.
 *   - one fd_ctx per (process, device); a ctx and the models created from it are
 *     not thread-safe.
 */
#ifndef FDIFF_HIP_H
#define FDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_OK 0
#define FD_ERR_ARG (-1)      /* bad argument (shape, null pointer, unsupported size) */
#define FD_ERR_HIP (-2)      /* a HIP runtime call failed */
#define FD_ERR_STATE (-3)    /* call order violated (e.g. forward before prepare) */
#define FD_ERR_COMM (-4)     /* RCCL failure */
#define FD_ERR_UNSUPPORTED (-5)

typedef struct fd_ctx fd_ctx;
typedef struct fd_score fd_score;

/* ---------------------------------------------------------------- context */
int fd_version(void);
int fd_ctx_create(int device, fd_ctx** out);
int fd_ctx_destroy(fd_ctx* ctx);
const char* fd_last_error(fd_ctx* ctx);   /* host string, valid until the next call on ctx */
/* number of bytes currently held by the ctx workspace (activations, scratch) */
size_t fd_ctx_workspace_bytes(fd_ctx* ctx);
/* Asynchronous device-side errors recorded since the last report: FD_OK, or FD_ERR_STATE when a kernel of an earlier training
 * call gave up a bounded inter-workgroup wait (the F-split hand-over of the training FFN kernels, or a cluster exchange of the
 * persistent training forward: fd_last_error names the token block / layer and series; that step's gradients are invalid).  Never
 * synchronises -- call it behind a stream synchronisation to cover everything enqueued so far.  Every training / optimizer entry
 * point performs the same check on entry, i.e. DETECTION LAGS BY ONE CALL unless the stream is synchronised; the optimizer
 * kernel itself tests a device-resident copy of the error word in stream order and skips the update of such a step.  (The reference has no
 * counterpart: torch autograd, src/fdiff/models/score_models.py:96-108, has no inter-workgroup protocol to fail.) */
int fd_ctx_check(fd_ctx* ctx);
/* After a reported cluster timeout of the persistent training forward the context trains on the per-layer kernels (whatever kept a
 * cluster from becoming resident -- a co-tenant process, a CU mask -- may still be there).  This re-arms the persistent form once the
 * caller knows the cause is gone.  No reference counterpart (see fd_ctx_check). */
int fd_ctx_rearm(fd_ctx* ctx);

/* ---------------------------------------------------------- measurement hooks
 * bench.py's roofline leg: between fd_prof_begin and fd_prof_end the engine brackets every launch of its
 * dominant kernel (the persistent score-network/sampler kernel; the fused FFN kernel on the step-by-step
 * fallback path) with HIP events on the launch stream.  fd_prof_end synchronises those events and returns
 * the kernel name, the average launch duration, the launch count and the ALGORITHMIC flops of one launch
 * (SURVEY.md 8d formula x series x diffusion steps in the launch; padding flops are not counted). */
int fd_prof_begin(fd_ctx* ctx);
/* after fd_prof_begin: bracket only every `every`-th launch of each kernel (an event pair costs the stream ~5 us; the
 * training step launches its bracketed kernels 20 times) */
int fd_prof_stride(fd_ctx* ctx, int every);
int fd_prof_end(fd_ctx* ctx, char* name_out /* >= 128 bytes */, double* avg_us, int* launches,
                double* flops_per_launch);
/* after fd_prof_end: the shader clock (MHz) the window's last persistent-kernel launch ran at -- workgroup 0's shader-clock counter
 * against the 100 MHz wall clock between its entry and the end of its last step; 0 when the window held no such launch.  The chip is
 * power-limited on this kernel and boxes differ by +-2.5 %: the bench line records the clock beside the time. */
int fd_prof_shader_clock_mhz(fd_ctx* ctx, double* mhz);

/* --------------------------------------------- a1/a2 spectral representation
 * replaces fdiff.utils.fourier.dft   (src/fdiff/utils/fourier.py:8-45)
 *      and fdiff.utils.fourier.idft  (src/fdiff/utils/fourier.py:48-87)
 * y[b, 0:T/2+1, c] = Re X_k ; y[b, T/2+1:T, c] = Im X_k (k=1..), ortho norm.
 * In place (x == y) is NOT allowed. */
int fd_rfft_pack(fd_ctx* ctx, const float* x, float* y, int B, int T, int C, void* stream);
int fd_irfft_unpack(fd_ctx* ctx, const float* x, float* y, int B, int T, int C, void* stream);
/* fused dataset front-end (src/fdiff/dataloaders/datamodules.py:61-62, cmd/sample.py:76-82):
 *   fd_rfft_pack_standardize : y = (dft(x) - mean) / std      mean,std (T,C)
 *   fd_destandardize_irfft   : y = idft(x * std + mean)                              */
int fd_rfft_pack_standardize(fd_ctx* ctx, const float* x, const float* mean, const float* std,
                             float* y, int B, int T, int C, void* stream);
int fd_destandardize_irfft(fd_ctx* ctx, const float* x, const float* mean, const float* std,
                           float* y, int B, int T, int C, void* stream);
/* spectral utilities on the packed representation xt = dft(x) (dataset front-end); the Python surface
 * (fdiff.utils.fourier.spectral_density / localization_metrics / smooth_frequency) composes them with fd_rfft_pack /
 * fd_irfft_unpack exactly as the reference composes its own dft / idft:
 *   fd_spectral_density     replaces spectral_density(x, apply_dft=False)  (src/fdiff/utils/fourier.py:90-124)
 *                           dens (B, T/2+1, C) = Re X_k^2 + Im X_k^2
 *   fd_localization_metrics replaces localization_metrics(X)               (src/fdiff/utils/fourier.py:127-175)
 *                           x (B,T,C) and xt = dft(x); loc, spec_loc (B,) = min_s sum_t e_t min(|t-s|, T-|t-s|)^2 with e the
 *                           normalised energy per time step / per bin of the two-sided spectrum
 *   fd_frequency_smooth     replaces the Gaussian mixing of smooth_frequency(X, sigma)  (src/fdiff/utils/fourier.py:189-203)
 *                           out[b,s,c] = sum_t xt[b,t,c] G[t,s]; gauss_scratch = T*T floats (caller-owned, receives G);
 *                           T must be odd (the reference's frequency vector has T-1 entries for even T and its einsum fails) */
int fd_spectral_density(fd_ctx* ctx, const float* xt, float* dens, int B, int T, int C, void* stream);
int fd_localization_metrics(fd_ctx* ctx, const float* x, const float* xt, float* loc, float* spec_loc,
                            int B, int T, int C, void* stream);
int fd_frequency_smooth(fd_ctx* ctx, const float* xt, float sigma, float* gauss_scratch, float* out,
                        int B, int T, int C, void* stream);

/* ------------------------------------------------------------ a3..a8 SDE
 * kind 0 = VP  (p0 = beta_min,  p1 = beta_max)   fdiff.schedulers.sde.VPScheduler (sde.py:168-246)
 * kind 1 = VE  (p0 = sigma_min, p1 = sigma_max)  fdiff.schedulers.sde.VEScheduler (sde.py:90-165) */
typedef struct fd_sde_params {
    int kind;
    float p0, p1;
} fd_sde_params;

/* Standard normals from the engine's Philox4x32-10 stream: element i of the call uses
 * counter (offset + i/4), key = seed.  Used for the prior, the per-step noise and tests. */
int fd_randn(fd_ctx* ctx, float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);
/* The generator underneath, for audits and tests: out[4 i .. 4 i + 3] = Philox4x32-10(counter = (offset + i, 0), key = seed)
 * (Salmon et al., SC'11; the stream torch.randn / nn.Dropout draw from on the reference's CUDA path is the same generator
 * with another counter layout, so the draws are equal in distribution, not bit for bit: SURVEY 7.2). */
int fd_philox_words(fd_ctx* ctx, uint32_t* out, size_t n_counters, uint64_t seed, uint64_t offset, void* stream);
/* The dropout decisions of the bf16 training path (replaces nn.Dropout's masks inside nn.TransformerEncoderLayer,
 * score_models.py:41-49): out[i] bit e = KEEP decision e of counter offset + i, e = 0..15.  Decision e compares the 16-bit
 * window at byte offset e of the 128-bit Philox output (little endian, wrapping) with thr16 = round(p * 65536): kept with
 * probability 1 - thr16 / 65536 exactly; windows overlap in one byte, so neighbours depend on each other only through ties
 * of the high byte (1 in 256). */
int fd_dropout_decisions(fd_ctx* ctx, uint16_t* out, size_t n_counters, float p, uint64_t seed, uint64_t offset,
                         void* stream);

/* replaces SDE.prior_sampling (sde.py:79-87) + VE override (sde.py:125-127):
 *   out = G[t] * z (VE: * sigma_max).  z == NULL -> z drawn on device (seed, offset). */
int fd_prior_sample(fd_ctx* ctx, const fd_sde_params* sde, const float* G, const float* z,
                    uint64_t seed, uint64_t offset, float* out, int B, int T, int C, void* stream);

/* replaces VPScheduler.step (sde.py:215-246) / VEScheduler.step (sde.py:129-165):
 *   VP: out = x + (0.5*beta*x + beta*G^2*score)*dt + sqrt(dt*beta)*G*z
 *   VE: out = x + g^2*G^2*score*dt + sqrt(dt)*g*G*z
 * One fused pass (reads x, score [, z]; writes out; out may alias x).
 * z == NULL -> on-device Philox noise (seed, offset). t is the Python float the reference passes. */
int fd_sde_step(fd_ctx* ctx, const fd_sde_params* sde, const float* G, const float* x,
                const float* score, const float* z, uint64_t seed, uint64_t offset, double t,
                float dt, float* out, int B, int T, int C, void* stream);

/* replaces the perturbation half of loss_fn (src/fdiff/utils/losses.py:66-85) with
 * marginal_prob (sde.py:108-123,187-210) and add_noise (sde.py:66-77) fused:
 *   std[b,k] = s(t_b)*G[k];  x_noisy = m(t_b)*x + std*z;  target = z/std
 * z == NULL -> Philox noise.  std_out (B,T) and target (B,T,C) may be NULL. */
int fd_perturb(fd_ctx* ctx, const fd_sde_params* sde, const float* G, const float* x,
               const float* t, const float* z, uint64_t seed, uint64_t offset, float* x_noisy,
               float* target, float* std_out, int B, int T, int C, void* stream);

/* replaces the reduction half of loss_fn (losses.py:92-124, reduce_mean=True):
 *   likelihood_weighting == 0: mean_b mean_{t,c} w_b (score+target)^2, w_b = 1/sum_k std^-2
 *   likelihood_weighting == 1: mean_b mean_{t,c} (std (score+target))^2
 * loss_out: device float[1].  dscore (nullable): d loss / d score, (B,T,C). */
int fd_dsm_loss(fd_ctx* ctx, const float* score, const float* target, const float* std,
                int likelihood_weighting, float* loss_out, float* dscore, int B, int T, int C,
                void* stream);

/* ------------------------------------------------------- a9 score network
 * replaces fdiff.models.score_models.ScoreModule (score_models.py:22-166): transformer
 * encoder (post-LN, relu, dim_ff) between Linear embed/unembed, learned positional table
 * with max_norm, Gaussian-Fourier time embedding.                                       */
typedef struct fd_model_dims {
    int n_channels;   /* C */
    int max_len;      /* T */
    int d_model;      /* D */
    int n_head;       /* H, D % H == 0 */
    int num_layers;   /* L */
    int dim_ff;       /* F (torch default 2048) */
} fd_model_dims;

/* Flat fp32 parameter buffer layout.  Order = the reference's state_dict order
 * (SURVEY.md A.4); every tensor starts on a 16-byte boundary.  name uses the reference's
 * state_dict key.  Call with entries == NULL to get the count. */
typedef struct fd_param_entry {
    char name[96];
    int64_t offset;   /* in floats */
    int64_t numel;
    int32_t rows, cols;   /* cols == 0 for vectors */
    int32_t trainable;    /* 0 for time_encoder.W (requires_grad=False, transformer.py:72-74) */
} fd_param_entry;
/* Stand-alone encoders of src/fdiff/models/transformer.py (the fused score network does not call these; they
 * back the PositionalEncoding / GaussianFourierProjection classes of the fdiff surface):
 *   fd_positional_add: renorm rows of table (T,D) IN PLACE to max_norm (nn.Embedding(max_norm), :13-15), then
 *                      out[b,t,:] = x[b,t,:] + table[t,:]                                   (:17-29)
 *   fd_time_embed_add: e = cat[sin,cos](2*pi*t*W)[:D]; p = e Wd^T + bd; out = x + p (broadcast over the time
 *                      axis when T > 0; T == 0 means x is (B,D))                            (:77-91) */
int fd_positional_add(fd_ctx* ctx, const float* x, float* table, float* out, int B, int T, int D, float max_norm,
                      void* stream);
int fd_time_embed_add(fd_ctx* ctx, const float* x, const float* t, const float* W, const float* Wd,
                      const float* bd, float* out, int B, int T, int D, void* stream);

int64_t fd_score_param_count(const fd_model_dims* dims);
int fd_score_layout(const fd_model_dims* dims, fd_param_entry* entries, int* n_entries);

int fd_score_create(fd_ctx* ctx, const fd_model_dims* dims, fd_score** out);

/* The reference's other two score backbones, behind the same fd_score handle and entry points
 * (forward, training pair, sampler loop, optimiser):
 *   FD_BACKBONE_MLP   replaces fdiff.models.score_models.MLPScoreModule  (score_models.py:169-246): the series is flattened
 *                     to (B, T*C), Linear embed, + time embedding, num_layers x { h += Linear(relu-dropout(Linear(h))) with
 *                     torchvision.ops.MLP(hidden=[d_mlp, d_model], dropout 0.1) }, Linear unembed; n_head / dim_ff unused.
 *   FD_BACKBONE_LSTM  replaces fdiff.models.score_models.LSTMScoreModule (score_models.py:249-317): Linear embed, + time
 *                     embedding, num_layers x { h += nn.LSTM(d_model, d_model, batch_first)(h) }, Linear unembed.
 * Both run exact-f32 kernels in either mode (no positional table: pos_encoder is None in the reference).
 * State-dict names: backbone.{i}.0.weight|bias, backbone.{i}.3.weight|bias (MLP); backbone.{i}.weight_ih_l0,
 * weight_hh_l0, bias_ih_l0, bias_hh_l0 (LSTM, gate order i|f|g|o). */
#define FD_BACKBONE_TRANSFORMER 0
#define FD_BACKBONE_MLP 1
#define FD_BACKBONE_LSTM 2
int64_t fd_score_param_count_ex(const fd_model_dims* dims, int backbone, int d_mlp);
int fd_score_layout_ex(const fd_model_dims* dims, int backbone, int d_mlp, fd_param_entry* entries, int* n_entries);
int fd_score_create_ex(fd_ctx* ctx, const fd_model_dims* dims, int backbone, int d_mlp, fd_score** out);
int fd_score_destroy(fd_score* m);

/* Derive the engine-side weight images from the flat fp32 parameters (device pointer):
 * max_norm-renormed positional table (the reference renorms in place inside forward,
 * transformer.py:13-15,27), bf16 MFMA-fragment-ordered matrices, folded biases.
 * Must be called after every change of params (load, optimizer step) before forward.
 * The engine keeps the pointer `params` (no copy of the fp32 masters). */
int fd_score_prepare(fd_score* m, const float* params, void* stream);

#define FD_MODE_F32 0    /* fp32 parity path (exact-f32 arithmetic) */
#define FD_MODE_BF16 1   /* bf16 MFMA operands, fp32 accumulate / residual / LN / softmax */

/* ScoreModule.forward (score_models.py:67-94), eval mode: x (B,T,C), t (B) -> out (B,T,C) */
int fd_score_forward(fd_score* m, const float* x, const float* t, float* out, int B, int mode,
                     void* stream);

/* Introspection (no launch): writes a description of the kernel path that fd_score_forward / fd_sampler_run take for a
 * batch of B series in `mode` -- for the persistent kernel the template instantiation, series per workgroup S, grid and
 * LDS bytes.  The parity tests assert through it that the instantiation they target really runs (the ecg bench workload
 * is `ShapeStatic<100,72,12,12,2,...>` with S = 2 on a 256-CU device).  No reference counterpart. */
int fd_score_plan(fd_score* m, int B, int mode, char* out /* >= 192 bytes */, int* series_per_workgroup /* nullable */);

/* Run-time specialisation of the persistent kernel (csrc/fd_mega_rtc.hip).  The library carries static-shape instantiations of
 * k_mega for the BASELINE shapes only; any other (model class, series shape, workgroup plan) gets its own through hiprtc on first use
 * -- FDIFF_MEGA_JIT: unset = sampler loops of >= 100 diffusion steps, 1 = every launch, 0 = never -- cached on disk under
 * $FDIFF_CACHE_DIR (default ~/.cache/fdiff_hip).  This entry compiles ONE instantiation into that cache without loading it (no GPU
 * needed): key14 = {KS1, DT, KSO, MT, T, D, C, H, S, NPG, rot, L, F, FFN32} as fd_score_plan prints them; msg (nullable, n bytes)
 * receives the instantiation and where its code object came from.  FD_ERR_UNSUPPORTED when hiprtc is missing or the compilation
 * fails (the engine then runs its run-time-shape instantiation).  Replaces nothing in the reference: torch specialises nothing per
 * dataset shape (src/fdiff/models/score_models.py:57-62 builds one nn.TransformerEncoder for every (max_len, n_channels)). */
int fd_mega_jit_compile(const int* key14, char* msg, int n);

/* Arithmetic of the training pair below: FD_MODE_F32 (default; exact-f32 kernels, the parity anchor, any model) or
 * FD_MODE_BF16 (bf16 MFMA operands, fp32 accumulate/LayerNorm/softmax; five fused kernels per encoder layer, weight
 * gradients reduced in a fixed order: bit-reproducible).  FD_ERR_UNSUPPORTED when the bf16 kernels are not instantiated
 * for the model's dims (the mode then stays unchanged).  Replaces nothing in the reference (torch autograd is fp32). */
int fd_score_set_train_mode(fd_score* m, int mode);
/* The training launch plan of a batch of B series, nothing launched: out (>= 192 bytes) names the arithmetic and, on the bf16
 * path, the token splits of the weight-gradient kernel (also in *token_splits, nullable; 0 on the exact-f32 path). */
int fd_score_train_plan(fd_score* m, int B, char* out, int* token_splits);

/* Training forward: keeps activations in the ctx workspace for fd_score_backward.
 * dropout_p > 0 applies the four dropout sites of nn.TransformerEncoderLayer with masks
 * from Philox(seed, offset) (regenerated in backward). */
int fd_score_forward_train(fd_score* m, const float* x, const float* t, float* out, int B,
                           float dropout_p, uint64_t seed, uint64_t offset, void* stream);
/* dout (B,T,C) -> grads (flat, same layout as params; ACCUMULATED into if accumulate != 0) */
int fd_score_backward(fd_score* m, const float* dout, float* grads, int accumulate, void* stream);

/* One optimisation step's device work in one call: fd_score_forward_train -> fd_dsm_loss -> fd_score_backward
 * (the body of get_sde_loss_fn's loss_fn + loss.backward(), src/fdiff/utils/losses.py:39-125 and
 * score_models.py:96-120) with the unembedder, the loss and the unembedder's backward fused into one kernel.
 * x: the perturbed batch (B,T,C); target, std: fd_perturb's outputs; loss_out: device float[1];
 * grad_weight scales the gradient (not the loss); grads as in fd_score_backward.
 * FD_ERR_UNSUPPORTED when the model does not train on the bf16 transformer path: run the three calls. */
int fd_score_train_dsm(fd_score* m, const float* x, const float* t, const float* target, const float* std,
                       int likelihood_weighting, float grad_weight, int B, float dropout_p, uint64_t seed,
                       uint64_t offset, float* loss_out, float* grads, int accumulate, void* stream);
/* 1 when fd_score_train_dsm has a fused step for this (model, train mode, B), else 0 (it would return FD_ERR_UNSUPPORTED).
 * No side effects: the host asks before it draws the step's Philox key. */
int fd_score_train_dsm_supported(fd_score* m, int B);

/* ----------------------------------------------------------- a12 sampler
 * replaces the inner loop of DiffusionSampler.sample (src/fdiff/sampling/sampler.py:83-104):
 * for i in range(n_steps): score = model(x, t_i); x = sde.step(score, t_i, x).
 * timesteps: HOST float[n_steps] (the scheduler's linspace grid, sde.py:62-64).
 * x: device (B,T,C), in/out.  z_steps: device (n_steps,B,T,C) injected noise or NULL for
 * on-device Philox (seed; step i, element e uses offset + i*ceil(BTC/4) + e/4).
 * The whole loop is enqueued on `stream` without any host synchronisation. */
int fd_sampler_run(fd_score* m, const fd_sde_params* sde, const float* G, const float* timesteps,
                   int n_steps, float dt, float* x, const float* z_steps, uint64_t seed,
                   uint64_t offset, int B, int mode, void* stream);

/* Predictor-corrector extension (NOT in the reference, whose sampler is predictor-only; default off in the Python surface;
 * parity unpinned -- follows Song et al. 2021, Alg. 4/5, in the coordinates whitened by G):
 *   fd_langevin_step : per series eps = 2 alpha (snr |z| / |G score|)^2 ; out = x + eps G^2 score + sqrt(2 eps) G z
 *                      (z == NULL: on-device Philox noise, needs T*C % 4 == 0; out may alias x)
 *   fd_sampler_run_pc: fd_sampler_run with n_corr corrector steps (a score evaluation each) before every predictor step;
 *                      alpha = 1 - beta(t) dt (VP) or 1 (VE); zc_steps (n_steps, n_corr, B, T, C) injected corrector noise or NULL */
int fd_langevin_step(fd_ctx* ctx, const float* G, const float* x, const float* score, const float* z, uint64_t seed,
                     uint64_t offset, float snr, float alpha, float* out, int B, int T, int C, void* stream);
int fd_sampler_run_pc(fd_score* m, const fd_sde_params* sde, const float* G, const float* timesteps, int n_steps, float dt,
                      float* x, const float* z_steps, const float* zc_steps, int n_corr, float snr, uint64_t seed,
                      uint64_t offset, int B, int mode, void* stream);

/* ------------------------------------------------------------ a11 optimiser
 * torch.optim.AdamW defaults + diffusers cosine-warmup + Lightning global-norm clip
 * (score_models.py:122-130, cmd/conf/trainer/default.yaml:4), fused over the flat buffer.
 * fd_grad_sqnorm: norm2_out[0] = sum(grads^2) (device float[1], fp32 accumulated in fp64 blocks).
 * fd_adamw_step : clip_coef is read from device: coef = min(1, max_norm/(sqrt(*sqnorm)+1e-6))
 *                 when sqnorm != NULL, else 1.  frozen [frozen_begin, frozen_end) is skipped.
 *                 The whole update is skipped ON THE DEVICE (parameters and moments unchanged) when a bounded wait of the
 *                 training step whose gradients these are timed out (see fd_ctx_check): the host only reports it, one call later. */
int fd_grad_sqnorm(fd_ctx* ctx, const float* grads, int64_t n, float* sqnorm_out, void* stream);
int fd_adamw_step(fd_ctx* ctx, float* params, const float* grads, float* exp_avg,
                  float* exp_avg_sq, int64_t n, int step, float lr, float beta1, float beta2,
                  float eps, float weight_decay, const float* sqnorm, float max_norm,
                  float grad_scale, int64_t frozen_begin, int64_t frozen_end, void* stream);

/* ----------------------------------------------------- (e) multi-GPU exchange
 * Data-parallel gradient all-reduce over RCCL/xGMI on ONE flat fp32 buffer.
 * unique_id: host bytes from fd_comm_unique_id on rank 0, broadcast by the launcher. */
#define FD_COMM_ID_BYTES 128
int fd_comm_unique_id(void* id_out /* host, FD_COMM_ID_BYTES */);
int fd_comm_init(fd_ctx* ctx, int rank, int nranks, const void* unique_id);
int fd_comm_destroy(fd_ctx* ctx);
/* path of the shared object the bound ncclAllReduce lives in (one RCCL image per process: an already mapped librccl --
 * torch's bundled one under the Python host -- is reused, never a second copy).  buf: host, n bytes. */
int fd_comm_rccl_path(char* buf, int n);
/* buf = sum over ranks (buf) * scale, in place */
int fd_allreduce_grads(fd_ctx* ctx, float* buf, int64_t n, float scale, void* stream);

/* ------------------------------------------- (f)3 evaluation metrics on the GPU
 * Sliced / marginal Wasserstein-2 between two sample sets (fdiff.sampling.metrics.SlicedWasserstein / MarginalWasserstein,
 * src/fdiff/sampling/metrics.py:100-217, over fdiff.utils.wasserstein.WassersteinDistances, src/fdiff/utils/wasserstein.py:95-199).
 * The host draws the directions exactly as the reference does (numpy Generator) and composes:
 *   fd_project_rows    out (K, n) = dirs (K, d) . x (n, d)^T      replaces WassersteinDistances._project  (wasserstein.py:150-153)
 *   fd_transpose_rows  out (d, n) = x (n, d)^T                    the marginal "directions" (standard basis, wasserstein.py:77-89)
 *   fd_sort_rows       every row of (K, n) sorted ascending; temp = fd_sort_rows_temp_bytes(K, n) caller-owned device bytes
 *   fd_w2_sorted_rows  out[k] = W2 between sorted row k of a (K, n) and of b (K, m), uniform weights, any n, m: the exact 1-D
 *                      transport POT's emd2_1d solves, then sqrt               (wasserstein.py:112-113, 139-141)            */
int fd_project_rows(fd_ctx* ctx, const float* x, const float* dirs, float* out, int n, int d, int K, void* stream);
int fd_transpose_rows(fd_ctx* ctx, const float* x, float* out, int n, int d, void* stream);
int fd_sort_rows_temp_bytes(fd_ctx* ctx, int K, int n, size_t* bytes);
int fd_sort_rows(fd_ctx* ctx, const float* in, float* out, int K, int n, void* temp, size_t temp_bytes, void* stream);
int fd_w2_sorted_rows(fd_ctx* ctx, const float* a, const float* b, float* out, int K, int n, int m, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDIFF_HIP_H */
