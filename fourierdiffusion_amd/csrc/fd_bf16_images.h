// fd_bf16_images.h -- engine-owned bf16 weight images of one score network (built by fd_bf16_prepare from the flat fp32
// parameters): MFMA-fragment-ordered 1 KiB blocks (64 lanes x 8 bf16) for the inference kernels (fd_score_bf16.hip,
// fd_mega.hip, fd_attn_bf16.hip) and, on first use, for the bf16 training kernels (fd_train_bf16.hip).
#pragma once
#include "fd_mega.h"

struct fd_bf16_images {
    bool supported = false;     // fused FFN kernel available (hybrid path)
    bool mega = false;          // persistent series-resident kernel available
    int ks1 = 0, dt = 0;        // k-steps of GEMM1 (incl. bias slot), 16-row tiles of d_model
    int kso = 0, kse = 0, ct = 0, np = 0;
    size_t ffn_layer_bytes = 0;
    char* ffn = nullptr;        // [L][fh 2][chunk F/64][NB blocks][64 lanes][8 bf16]
    // persistent-kernel images: emb | unemb | per layer {wk, wv, wq, wo} (FFN image shared with `ffn`)
    char* mimg = nullptr;
    size_t off_emb = 0, off_unemb = 0, off_layers = 0, layer_stride = 0;
    size_t off_wk = 0, off_wv = 0, off_wq = 0, off_wo = 0, off_ffn = 0;
    size_t off_lpar = 0;        // fp32 [6][D] out_b, l2_b, n1_w, n1_b, n2_w, n2_b of the layer, zero-padded to nlp KiB
    int nlp = 0;
    bool ffn32_stale = false;                      // skipped by the last (training-step) rebuild
    size_t off_ffn32 = 0, ffn32_layer_bytes = 0;   // pair-form FFN image of the layer (32x32x16 H; 0 bytes: not built)
    // ---- training (fd_train_bf16.hip): transposed-weight images of the backward pass, built lazily with the others
    bool train = false;         // bf16 training kernels instantiated for this model
    char* bimg = nullptr;       // per layer: FFN backward blocks (chunk-major, same block count as the forward image),
                                // W_o^T, and the per-pair W_q|W_k|W_v^T blocks of the attention backward
    size_t b_layer_stride = 0, boff_ffn = 0, boff_wot = 0, boff_win = 0;
    // ---- widths outside the persistent kernel's family: projection images of the step-by-step path (fd_linear_bf16.hip),
    // per layer [W_in: nrt_in row tiles x ks1][W_o: dt row tiles x ks1] 1 KiB blocks (rows of W, bias in k-slot d_model)
    char* pimg = nullptr;
    size_t p_layer_stride = 0, poff_wo = 0;
    int nrt_in = 0;
    long long* layer_off_tab = nullptr;   // device [L][12]: fd_layer_off of every layer (single-launch image build)
};
