// fd_mega_params.h -- parameter block of the persistent series-resident kernel (fd_mega_kernel.h).  Plain data only: this header is
// also handed to hiprtc (fd_mega_rtc.hip), which has no host headers.
#pragma once
#ifndef __HIPCC_RTC__
#include <cstddef>
#endif

#ifndef FD_W1_SWAP34
#define FD_W1_SWAP34 1      // pair-form W1 image rows stored with index bits 3 <-> 4 swapped (LDS bank slots; 0 = natural order, A/B builds)
#endif
#define FD_MEGA_FORWARD 0   // one score-network forward: x, tvec -> score_out
#define FD_MEGA_SAMPLE 1    // nsteps x {forward, reverse-SDE step}, x updated in place

struct fd_sde_step_coef {
    float a_x, g, dt, sqrt_dt, t;   // SdeCoef of fd_sde.h + the timestep itself (time embedding)
};

struct fd_mega_params {
    // shapes
    int B, T, KT /* ceil(T/16) */, C, D, H, hd, L, F;
    int S;        // series per workgroup
    int NPG;      // head pairs per attention group (K/V buffers hold one group)
    int KSE;      // k-steps of the embed GEMM  (ceil((C+1)/32))
    int CT;       // 16-row tiles of the unembed GEMM (ceil(C/16))
    int rot;      // rotation of the second wave set (SIMD load balance)
    int num_cu;   // CUs of the device (co-resident 4-wave workgroups alternate their tile split)
    int mode, nsteps;
    int lds_temb; // byte offset of the time-embedding scratch in LDS
    int lds_afr;  // byte offset of the attention-output fragments in LDS
    int dbg;      // debugging aid (FDIFF_MEGA_DBG): bit0 zero the attention output, bit1 skip the FFN
    unsigned long long* prof;   // profiling aid (FDIFF_MEGA_PROF): (phase, s_memtime) pairs of WG 0 / wave 0, steps 0-3
    unsigned long long* clk_out;   // measurement aid (fd_prof_begin .. fd_prof_end): workgroup 0 stores {shader-clock counter, 100 MHz wall
                                   // clock} at entry ([0], [1]) and behind its last step ([2], [3]): the shader clock the launch ran at
    unsigned* dbg_out;   // debugging aid: LDS image of workgroup 0 after layer 0's attention
    int dbg_bytes;
    // tensors
    float* x;
    float* score_out;
    const float* tvec;
    const float* params;
    long long pos, tW, td_w, td_b;
    // bf16 fragment images
    const char* img_emb;
    const char* img_unemb;
    const char* img_layers;
    size_t layer_stride;
    size_t off_wk, off_wv, off_wq, off_wo, off_ffn;
    size_t off_ffn32;                    // pair-form FFN image (32x32x16 H) of the layer, 0 when the model has none
    size_t off_lpar;                     // fp32 block [6][D] (bo, b2, g1, b1, g2, b2) of the layer, nlp KiB: fetched by DMA
    int nlp;
    // sampler
    const float* G;
    const fd_sde_step_coef* steps;       // device array [nsteps]
    const float* z_steps;                // injected noise (nsteps, B, T, C) or null
    const float* temb_table;             // (nsteps, D) time embedding of every step's t (sampler mode: t is shared by all
                                         // series, fd_mega_temb_table fills it before the launch) or null
    unsigned long long seed, offset, ctr_per_step, n_elem;
};
