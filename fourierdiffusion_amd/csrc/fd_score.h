// fd_score.h -- internal definition of the score-network object (fd_score) shared by the fp32
// parity path, the bf16 MFMA path, the backward pass and the sampler.
#pragma once
#include <vector>

#include "fd_common.h"

struct fd_layer_off {
    int64_t in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b;
};

struct fd_bf16_images;   // fd_score_bf16.hip

// parameter offsets of one layer of the MLP / LSTM backbones (fd_backbones.hip):
//   MLP : a = linear1.weight (d_mlp, D), b = its bias, c = linear2.weight (D, d_mlp), d = its bias
//   LSTM: a = weight_ih_l0 (4D, D), b = weight_hh_l0 (4D, D), c = bias_ih_l0 (4D), d = bias_hh_l0 (4D)
struct fd_bb_off {
    int64_t a, b, c, d;
};

struct fd_score {
    fd_ctx* ctx = nullptr;
    fd_model_dims d{};
    int64_t nparams = 0;
    int64_t pos = 0, tW = 0, td_w = 0, td_b = 0, emb_w = 0, emb_b = 0, un_w = 0, un_b = 0;
    std::vector<fd_layer_off> layers;
    int backbone = 0;               // FD_BACKBONE_TRANSFORMER / _MLP / _LSTM
    int d_mlp = 0;                  // hidden width of the MLP backbone
    std::vector<fd_bb_off> bb;      // per-layer offsets of the MLP / LSTM backbones
    float* params = nullptr;        // caller-owned flat fp32 masters (set by fd_score_prepare)
    bool prepared = false;
    bool bf16_stale = true;         // bf16 weight images older than the fp32 masters (rebuilt lazily: training never reads them)
    hipEvent_t prep_event = nullptr;   // completed by fd_score_prepare's last kernel (its launch's stop event): what a rebuild of the
    void* prep_stream = nullptr;       // weight images on another stream waits for, without an event packet on the caller's stream
    bool prep_event_bound = false;
    hipEvent_t img_event = nullptr;    // completed by a rebuild of the bf16 weight images that runs on a side stream (fd_train_bf16.hip)
    fd_bf16_images* bf16 = nullptr; // engine-owned bf16 weight images (built by prepare)
    // ---- training state (valid between forward_train and backward)
    bool have_saved = false;
    int saved_B = 0;
    float saved_p = 0.f;
    uint64_t saved_seed = 0, saved_offset = 0;
    const float* saved_x = nullptr;
    const float* saved_t = nullptr;
    int saved_mask_set = 0;         // which of the two sets of dropout-decision buffers the saved forward used (fd_train_bf16.hip)
    bool saved_bf16 = false;        // the training forward ran the bf16 MFMA kernels (fd_train_bf16.hip)
    int train_mode = 0;             // FD_MODE_F32 (exact-f32 kernels) or FD_MODE_BF16 for fd_score_forward_train
    uint64_t saved_ws_gen = 0;      // ctx->ws_gen right after the training forward carved the arena
    void* saved_ws = nullptr;       // arena base then (a regrow moves it)
};

// activations kept by the training forward, carved from the ctx workspace
struct fd_saved_layer {
    float* x0;     // (M,D)  layer input
    float* qkv;    // (M,3D)
    float* lse;    // (B,H,T) log-sum-exp of scaled scores
    float* att;    // (M,D)  concatenated head outputs
    float* s1;     // (M,D)  x0 + drop(out_proj)
    float* mr1;    // (M,2)  mean, rstd of LN1
    float* x1;     // (M,D)  LN1 output
    float* hact;   // (M,F)  drop(relu(linear1))
    float* s2;     // (M,D)  x1 + drop(linear2)
    float* mr2;    // (M,2)
};

struct fd_saved {
    float* emb;    // (B,D)  Gaussian-Fourier features (input of time_encoder.dense)
    float* temb;   // (B,D)
    float* hL;     // (M,D)  last layer output
    std::vector<fd_saved_layer> layers;
};

// fd_score_f32.hip
int fd_score_forward_f32(fd_score* m, const float* x, const float* t, float* out, int B, hipStream_t s,
                         bool train, float dropout_p, uint64_t seed, uint64_t offset);
size_t fd_score_f32_workspace(const fd_score* m, int B, bool train);
void fd_score_carve_saved(const fd_score* m, int B, fd_ws& ws, fd_saved& sv);
size_t fd_score_bwd_workspace(const fd_score* m, int B);   // fd_score_bwd.hip

// fd_score_bf16.hip
int fd_bf16_create(fd_score* m);
void fd_bf16_destroy(fd_score* m);
int fd_bf16_prepare(fd_score* m, hipStream_t s, bool training_only = false, bool ffn32_only = false);
int fd_bf16_refresh(fd_score* m, hipStream_t s, bool training_only = false);   // (training_only: without the sampler-only images)
// rebuilds the images if the masters changed since the last build
int fd_score_forward_bf16(fd_score* m, const float* x, const float* t, float* out, int B, hipStream_t s);
// fd_backbones.hip: the reference's other two score backbones (MLPScoreModule / LSTMScoreModule), exact-f32
int fd_bb_forward(fd_score* m, const float* x, const float* t, float* out, int B, hipStream_t s, bool train, float p,
                  uint64_t seed, uint64_t offset);
int fd_bb_backward(fd_score* m, const float* dout, float* grads, int accumulate, hipStream_t s);
size_t fd_bb_workspace(const fd_score* m, int B, bool train);
// any backbone, eval mode (sampler loop, fd_score_forward)
int fd_score_forward_any(fd_score* m, const float* x, const float* t, float* out, int B, int mode, hipStream_t s);
// fd_train_bf16.hip: bf16 MFMA training path (forward with dropout, backward)
bool fd_train_bf16_supported(const fd_score* m);
bool fd_score_train_dsm_bf16_supported(const fd_score* m, int B);
int fd_train_bf16_token_splits(const fd_score* m, int B, int* nblk);
void fd_train_bf16_forward_plan(const fd_score* m, int B, char* out, size_t n);   // "k_tr_fwd_layers NT=.. x .." or "2 kernels per layer"
int fd_score_forward_train_bf16(fd_score* m, const float* x, const float* t, float* out, int B, float p, uint64_t seed,
                                uint64_t offset, hipStream_t s);
int fd_score_backward_bf16(fd_score* m, const float* dout, float* grads, int accumulate, hipStream_t s);
int fd_score_train_dsm_bf16(fd_score* m, const float* x, const float* t, const float* target, const float* stdv, int lw,
                            float grad_weight, int B, float p, uint64_t seed, uint64_t offset, float* loss_out, float* grads,
                            int accumulate, hipStream_t s);
int fd_embed_backward(fd_score* m, const float* dh, const float* emb, float* dtemb, float* grads, int B, float* skp,
                      size_t skp_floats, hipStream_t s);   // fd_score_bwd.hip
// fd_attn_bf16.hip
// fd_linear_bf16.hip: bf16 MFMA projections of the step-by-step path (weight images [row tile][k-step], bias in k-slot K)
int fd_linear_bf16(fd_ctx* ctx, const float* x, const char* img, float* out, int M, int N, int K, int ks, hipStream_t s);
int fd_linear_res_ln_bf16(fd_ctx* ctx, const float* x, const char* img, const float* res, const float* gamma, const float* beta,
                          float* out, int M, int D, int ks, int dt, hipStream_t s);
// head_dim 8 .. 32, one head per contraction (fd_attn_wide.hip); qkv = packed projections (B*T, 3D)
int fd_attention_bf16_wide(fd_ctx* ctx, const float* qkv, float* out, int B, int T, int H, int hd, hipStream_t s);
int fd_attention_bf16(fd_ctx* ctx, const float* in, float* out, int B, int T, int H, int hd, hipStream_t s,
                      const char* wk = nullptr, const char* wv = nullptr, const char* wq = nullptr, int ks1 = 0, int out_bf16 = 0,
                      const void* in_rows = nullptr);
