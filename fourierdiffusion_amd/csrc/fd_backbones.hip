// fd_backbones.hip -- the reference's other two score backbones behind the fd_score handle:
//   MLPScoreModule  (src/fdiff/models/score_models.py:169-246)  flatten -> Linear -> + time embedding ->
//                   L x { h += Linear(dropout(relu(Linear(h)))) , dropout }  (torchvision.ops.MLP(hidden=[d_mlp, d_model], dropout 0.1))
//                   -> Linear -> unflatten
//   LSTMScoreModule (src/fdiff/models/score_models.py:249-317)  Linear embed -> + time embedding -> L x { h += LSTM(h) } -> Linear
// Exact-f32 arithmetic: the dense pieces are the fp32-MFMA GEMMs of fd_gemm_f32.h (deterministic split-K), the LSTM
// recurrence is one workgroup per series with the W_hh row of each gate unit in registers (forward) / W_hh in LDS
// (backward through time); bias / time-embedding reductions are the fixed-order column sums of fd_score_bwd.hip.
// No float atomics: gradients are bit-reproducible.
#include <algorithm>

#include "fd_gemm_f32.h"
#include "fd_philox.h"
#include "fd_score.h"

void fd_dropout_inplace(fd_ctx* ctx, float* x, size_t n, float p, uint64_t seed, uint64_t offset, hipStream_t s);
uint64_t fd_dropout_site_offset(uint64_t base, int layer, int site);
int fd_time_embed_train(const float* t, const float* W, const float* Wd, const float* bd, float* emb, float* temb, int B, int D,
                        hipStream_t s);
namespace fdf32 {
void embed(const float* x, const float* We, const float* be, const float* pe, const float* temb, float* h, int M, int T, int C,
           int D, hipStream_t s);
}

namespace {

inline unsigned ew_grid(fd_ctx* ctx, size_t n) {
    size_t b = (n + 255) / 256;
    const size_t cap = (size_t)ctx->num_cu * 8;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}
inline size_t fl(size_t n) { return fd_ws::padded(n * sizeof(float)); }

// a[r, :] += v[r / rows_per_v, :]      (time embedding broadcast; rows_per_v = 1 for the MLP, T for the LSTM)
__global__ __launch_bounds__(256) void k_add_rows(float* __restrict__ a, const float* __restrict__ v, size_t n, int D, int rows_per_v) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / D;
        a[i] += v[(r / rows_per_v) * D + (i - r * D)];
    }
}
__global__ __launch_bounds__(256) void k_add3(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void k_add_inplace(float* __restrict__ a, const float* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] += b[i];
}
// out = dropout-backward(in): same Philox stream as fd_dropout_inplace (4 elements per counter)
__global__ __launch_bounds__(256) void k_drop_copy(const float* __restrict__ in, float* __restrict__ out, size_t n, float p,
                                                    uint64_t seed, uint64_t offset) {
    const size_t ng = (n + 3) / 4;
    const float sc = (p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
    for (size_t g = blockIdx.x * (size_t)256 + threadIdx.x; g < ng; g += (size_t)gridDim.x * 256) {
        uint32_t rv[4] = {~0u, ~0u, ~0u, ~0u};
        if (p > 0.f) {
            const fd_u4 r = fd_philox4x32_10(offset + g, seed);
            rv[0] = r.x; rv[1] = r.y; rv[2] = r.z; rv[3] = r.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t e = g * 4 + i;
            if (e < n) out[e] = (p <= 0.f || fd_u01(rv[i]) >= p) ? in[e] * sc : 0.f;
        }
    }
}
// d(relu + inverted dropout): g *= (act != 0) / (1 - p)   (act = drop(relu(.)): nonzero <=> kept and positive)
__global__ __launch_bounds__(256) void k_relu_drop_bwd(float* __restrict__ g, const float* __restrict__ act, size_t n, float inv_keep) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        g[i] = (act[i] != 0.f) ? g[i] * inv_keep : 0.f;
}
// dtemb[b, :] = sum_t dh[b, t, :]     (four strands, fixed order)
__global__ __launch_bounds__(256) void k_sum_time(const float* __restrict__ dh, float* __restrict__ dtemb, int B, int T, int D) {
    const size_t id = blockIdx.x * (size_t)256 + threadIdx.x;
    if (id >= (size_t)B * D) return;
    const int b = (int)(id / D), d = (int)(id % D);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int t = 0;
    for (; t + 3 < T; t += 4) {
        a0 += dh[((size_t)b * T + t + 0) * D + d];
        a1 += dh[((size_t)b * T + t + 1) * D + d];
        a2 += dh[((size_t)b * T + t + 2) * D + d];
        a3 += dh[((size_t)b * T + t + 3) * D + d];
    }
    for (; t < T; ++t) a0 += dh[((size_t)b * T + t) * D + d];
    dtemb[id] = (a0 + a1) + (a2 + a3);
}
// hprev[b, t, :] = h[b, t-1, :] (zeros at t = 0): the W_hh gradient contracts d gates[t] with h[t-1]
__global__ __launch_bounds__(256) void k_shift_time(const float* __restrict__ h, float* __restrict__ hprev, int B, int T, int D) {
    const size_t n = (size_t)B * T * D;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t row = i / D;
        const int t = (int)(row % T);
        hprev[i] = (t > 0) ? h[i - D] : 0.f;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// LSTM recurrence of one series per workgroup (torch.nn.LSTM, gate order i | f | g | o):
//   pre[j] = xg[t, j] + b_hh[j] + sum_k W_hh[j, k] h[t-1, k]       (xg = x W_ih^T + b_ih from the GEMM)
//   c = f c + i g ;  h = o tanh(c)
// thread j < 4D owns gate unit j with its W_hh row in registers; h[t-1] is broadcast from LDS.
template <int DMAX, bool TRAIN>
__global__ __launch_bounds__(4 * DMAX) void k_lstm_rec(const float* __restrict__ xg, const float* __restrict__ Whh,
                                                        const float* __restrict__ bhh, float* __restrict__ hseq,
                                                        float* __restrict__ gates_out, float* __restrict__ cseq, int T, int D) {
    __shared__ float hs[DMAX];
    __shared__ float act[4 * DMAX];
    const int b = blockIdx.x, j = threadIdx.x, G = 4 * D;
    const bool on = j < G;
    float w[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) w[k] = (on && k < D) ? Whh[(size_t)j * D + k] : 0.f;
    const float bj = on ? bhh[j] : 0.f;
    const int gate = on ? j / D : 0;
    float c = 0.f;
    if (j < DMAX) hs[j] = 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)b * T + t;
        float pre = on ? xg[row * G + j] + bj : 0.f;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) pre = fmaf(w[k], hs[k], pre);
        const float a = (gate == 2) ? tanhf(pre) : sigmoidf_(pre);
        if (on) {
            act[j] = a;
            if (TRAIN) gates_out[row * G + j] = a;
        }
        __syncthreads();
        if (j < D) {
            const float ig = act[j], fg = act[D + j], gg = act[2 * D + j], og = act[3 * D + j];
            c = fg * c + ig * gg;
            const float h = og * tanhf(c);
            hs[j] = h;
            hseq[row * D + j] = h;
            if (TRAIN) cseq[row * D + j] = c;
        }
        __syncthreads();
    }
}

// Backward through time of one series per workgroup: from dy[t] = d loss / d h[t] (the layer's output branch) to the
// pre-activation gate gradients dgates[t] (the weight / input gradients are GEMMs afterwards).
// thread (q = tid / D, k = tid % D): the W_hh^T matvec is split into the 4 gate blocks q, added in a fixed order.
template <int DMAX>
__global__ __launch_bounds__(4 * DMAX) void k_lstm_bwd_rec(const float* __restrict__ dy, const float* __restrict__ gates,
                                                            const float* __restrict__ cseq, const float* __restrict__ Whh,
                                                            float* __restrict__ dgates, int T, int D) {
    extern __shared__ float sh[];
    float* const W = sh;                  // [4D][D]
    float* const dg = W + 4 * D * D;      // [4D]
    float* const part = dg + 4 * D;       // [4][D]
    const int b = blockIdx.x, tid = threadIdx.x, G = 4 * D;
    for (int i = tid; i < G * D; i += blockDim.x) W[i] = Whh[i];
    const int q = tid / D, k = tid - q * D;
    const bool on = tid < G;
    float dh_next = 0.f, dc_next = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)b * T + t;
        if (tid < D) {
            const float dh = dy[row * D + tid] + dh_next;
            const float ig = gates[row * G + tid], fg = gates[row * G + D + tid], gg = gates[row * G + 2 * D + tid],
                        og = gates[row * G + 3 * D + tid];
            const float ct = cseq[row * D + tid], cp = (t > 0) ? cseq[(row - 1) * D + tid] : 0.f;
            const float tc = tanhf(ct);
            const float dc = dh * og * (1.0f - tc * tc) + dc_next;
            const float gi = dc * gg * ig * (1.0f - ig);
            const float gf = dc * cp * fg * (1.0f - fg);
            const float ggp = dc * ig * (1.0f - gg * gg);
            const float go = dh * tc * og * (1.0f - og);
            dc_next = dc * fg;
            dg[tid] = gi; dg[D + tid] = gf; dg[2 * D + tid] = ggp; dg[3 * D + tid] = go;
            dgates[row * G + tid] = gi; dgates[row * G + D + tid] = gf; dgates[row * G + 2 * D + tid] = ggp;
            dgates[row * G + 3 * D + tid] = go;
        }
        __syncthreads();
        if (on) {
            float a = 0.f;
            for (int jj = 0; jj < D; ++jj) a = fmaf(W[(size_t)(q * D + jj) * D + k], dg[q * D + jj], a);
            part[q * D + k] = a;
        }
        __syncthreads();
        if (tid < D) dh_next = (part[tid] + part[D + tid]) + (part[2 * D + tid] + part[3 * D + tid]);
        // (part is rewritten only after the next barrier pair)
    }
}

// ------------------------------------------------------------------------------------------------ workspace
struct BbBufs {
    float *emb, *temb, *hL, *tmp;
    std::vector<float*> h;       // layer inputs (rows x D)
    std::vector<float*> act;     // MLP: drop(relu(linear1)) (rows x d_mlp); LSTM: post-activation gates (M x 4D)
    std::vector<float*> aux;     // LSTM: cell states (M x D)
    std::vector<float*> hs;      // LSTM: hidden sequence of the layer (M x D)
    float *xg;                   // LSTM: x W_ih^T + b_ih (M x 4D), reused per layer
    // backward
    float *dh, *dtmp, *dact, *dtemb, *skp, *hprev;
};
constexpr size_t kSkp = (size_t)1 << 20;

size_t bb_carve(const fd_score* m, int B, bool train, char* base, BbBufs* out) {
    const size_t T = m->d.max_len, D = m->d.d_model, L = m->d.num_layers;
    const bool mlp = m->backbone == FD_BACKBONE_MLP;
    const size_t R = mlp ? (size_t)B : (size_t)B * T;          // rows of the hidden stream
    const size_t W = mlp ? (size_t)m->d_mlp : 4 * D;
    size_t off = 0;
    auto take = [&](size_t nfl) { float* p = base ? (float*)(base + off) : nullptr; off += fl(nfl); return p; };
    BbBufs b;
    b.emb = take((size_t)B * D);
    b.temb = take((size_t)B * D);
    b.hL = take(R * D);
    b.tmp = take(R * std::max(D, W));
    const size_t nl = train ? L : 1;
    b.h.resize(L); b.act.resize(L); b.aux.resize(L); b.hs.resize(L);
    std::vector<float*> hh(nl), aa(nl), xx(nl), ss(nl);
    for (size_t i = 0; i < nl; ++i) {
        hh[i] = take(R * D);
        aa[i] = take(R * W);
        xx[i] = mlp ? nullptr : take(R * D);
        ss[i] = mlp ? nullptr : take(R * D);
    }
    float* h_alt = train ? nullptr : take(R * D);             // eval: ping-pong layer input / output
    for (size_t i = 0; i < L; ++i) {
        b.h[i] = train ? hh[i] : ((i & 1) ? h_alt : hh[0]);
        b.act[i] = aa[train ? i : 0];
        b.aux[i] = xx[train ? i : 0];
        b.hs[i] = ss[train ? i : 0];
    }
    b.xg = mlp ? nullptr : take(R * 4 * D);
    if (train) {
        b.dh = take(R * D);
        b.dtmp = take(R * std::max(D, W));
        b.dact = take(R * W);
        b.dtemb = take((size_t)B * D);
        b.skp = take(kSkp);
        b.hprev = mlp ? nullptr : take(R * D);
    } else {
        b.dh = b.dtmp = b.dact = b.dtemb = b.skp = b.hprev = nullptr;
    }
    if (out) *out = b;
    return off + 4096;
}

template <bool TRAIN>
int launch_lstm_rec(fd_ctx* ctx, const float* xg, const float* Whh, const float* bhh, float* hseq, float* gates, float* cseq, int B,
                    int T, int D, hipStream_t s) {
    if (D <= 72) hipLaunchKernelGGL((k_lstm_rec<72, TRAIN>), dim3(B), dim3(4 * 72), 0, s, xg, Whh, bhh, hseq, gates, cseq, T, D);
    else if (D <= 128) hipLaunchKernelGGL((k_lstm_rec<128, TRAIN>), dim3(B), dim3(4 * 128), 0, s, xg, Whh, bhh, hseq, gates, cseq, T, D);
    else return fd_fail(ctx, FD_ERR_UNSUPPORTED, "LSTM backbone: d_model %d > 128", D);
    return FD_OK;
}

}  // namespace

size_t fd_bb_workspace(const fd_score* m, int B, bool train) { return bb_carve(m, B, train, nullptr, nullptr); }

int fd_bb_forward(fd_score* m, const float* x, const float* t, float* out, int B, hipStream_t s, bool train, float p, uint64_t seed,
                  uint64_t offset) {
    fd_ctx* ctx = m->ctx;
    const int T = m->d.max_len, C = m->d.n_channels, D = m->d.d_model, L = m->d.num_layers;
    const bool mlp = m->backbone == FD_BACKBONE_MLP;
    const float* P = m->params;
    if (int rc = fd_ws_reserve(ctx, fd_bb_workspace(m, B, train))) return rc;
    fd_ws ws(ctx);
    BbBufs b;
    bb_carve(m, B, train, (char*)ctx->ws, &b);
    size_t gsk_n = 0;
    float* gsk = fd_gemm_scratch(ctx, &gsk_n);
    if (fd_time_embed_train(t, P + m->tW, P + m->td_w, P + m->td_b, train ? b.emb : nullptr, b.temb, B, D, s)) return FD_ERR_HIP;
    const size_t R = mlp ? (size_t)B : (size_t)B * T;
    float* h0 = L > 0 ? b.h[0] : b.hL;
    if (mlp) {
        fdgemm::linear_fwd(x, P + m->emb_w, P + m->emb_b, h0, B, D, T * C, false, s, gsk, gsk_n);
        hipLaunchKernelGGL(k_add_rows, dim3(ew_grid(ctx, R * D)), dim3(256), 0, s, h0, b.temb, R * D, D, 1);
    } else {
        fdf32::embed(x, P + m->emb_w, P + m->emb_b, nullptr, b.temb, h0, B * T, T, C, D, s);
    }
    const float pp = train ? p : 0.f;
    for (int i = 0; i < L; ++i) {
        const fd_bb_off& o = m->bb[i];
        float* hin = b.h[i];
        float* hout = (i + 1 < L) ? b.h[i + 1] : b.hL;
        if (mlp) {
            const int F = m->d_mlp;
            // torchvision.ops.MLP: Linear -> ReLU -> Dropout -> Linear -> Dropout, then the residual
            fdgemm::linear_fwd(hin, P + o.a, P + o.b, b.act[i], B, F, D, true, s);
            if (pp > 0.f) fd_dropout_inplace(ctx, b.act[i], (size_t)B * F, pp, seed, fd_dropout_site_offset(offset, i, 0), s);
            fdgemm::linear_fwd(b.act[i], P + o.c, P + o.d, b.tmp, B, D, F, false, s, gsk, gsk_n);
            if (pp > 0.f) fd_dropout_inplace(ctx, b.tmp, (size_t)B * D, pp, seed, fd_dropout_site_offset(offset, i, 1), s);
            hipLaunchKernelGGL(k_add3, dim3(ew_grid(ctx, R * D)), dim3(256), 0, s, hin, b.tmp, hout, R * D);
        } else {
            fdgemm::linear_fwd(hin, P + o.a, P + o.c, b.xg, B * T, 4 * D, D, false, s);
            int rc = train ? launch_lstm_rec<true>(ctx, b.xg, P + o.b, P + o.d, b.hs[i], b.act[i], b.aux[i], B, T, D, s)
                           : launch_lstm_rec<false>(ctx, b.xg, P + o.b, P + o.d, b.hs[i], nullptr, nullptr, B, T, D, s);
            if (rc) return rc;
            hipLaunchKernelGGL(k_add3, dim3(ew_grid(ctx, R * D)), dim3(256), 0, s, hin, b.hs[i], hout, R * D);
        }
    }
    if (mlp) fdgemm::linear_fwd(b.hL, P + m->un_w, P + m->un_b, out, B, T * C, D, false, s);
    else fdgemm::linear_fwd(b.hL, P + m->un_w, P + m->un_b, out, B * T, C, D, false, s);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

int fd_bb_backward(fd_score* m, const float* dout, float* grads, int accumulate, hipStream_t s) {
    fd_ctx* ctx = m->ctx;
    const int B = m->saved_B;
    const int T = m->d.max_len, C = m->d.n_channels, D = m->d.d_model, L = m->d.num_layers;
    const bool mlp = m->backbone == FD_BACKBONE_MLP;
    const float* P = m->params;
    const float p = m->saved_p;
    if (ctx->ws_bytes < fd_bb_workspace(m, B, true))
        return fd_fail(ctx, FD_ERR_STATE, "fd_score_backward: workspace was resized since the training forward");
    BbBufs b;
    bb_carve(m, B, true, (char*)ctx->ws, &b);
    size_t gsk_n = 0;
    float* gsk = fd_gemm_scratch(ctx, &gsk_n);
    if (!accumulate) FD_HIP(ctx, hipMemsetAsync(grads, 0, sizeof(float) * (size_t)m->nparams, s));
    const size_t R = mlp ? (size_t)B : (size_t)B * T;
    const int Ri = (int)R;
    const int Cout = mlp ? T * C : C;
    const float inv_keep = (p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
    // ---- unembedder
    fdgemm::linear_bwd_weight(dout, b.hL, grads + m->un_w, Ri, Cout, D, true, s, b.skp, kSkp);
    if (int rc = fd_colsum_det(ctx, dout, grads + m->un_b, Ri, Cout, s)) return rc;
    fdgemm::linear_bwd_input(dout, P + m->un_w, b.dh, Ri, Cout, D, false, s, gsk, gsk_n);
    for (int i = L - 1; i >= 0; --i) {
        const fd_bb_off& o = m->bb[i];
        if (mlp) {
            const int F = m->d_mlp;
            // h_out = h + drop1(f), f = act W2^T + b2, act = drop0(relu(h W1^T + b1))
            hipLaunchKernelGGL(k_drop_copy, dim3(ew_grid(ctx, (R * D + 3) / 4)), dim3(256), 0, s, b.dh, b.dtmp, R * D, p, m->saved_seed,
                               fd_dropout_site_offset(m->saved_offset, i, 1));
            if (int rc = fd_colsum_det(ctx, b.dtmp, grads + o.d, Ri, D, s)) return rc;
            fdgemm::linear_bwd_weight(b.dtmp, b.act[i], grads + o.c, Ri, D, F, true, s, b.skp, kSkp);
            fdgemm::linear_bwd_input(b.dtmp, P + o.c, b.dact, Ri, D, F, false, s);
            hipLaunchKernelGGL(k_relu_drop_bwd, dim3(ew_grid(ctx, R * F)), dim3(256), 0, s, b.dact, b.act[i], R * F, inv_keep);
            if (int rc = fd_colsum_det(ctx, b.dact, grads + o.b, Ri, F, s)) return rc;
            fdgemm::linear_bwd_weight(b.dact, b.h[i], grads + o.a, Ri, F, D, true, s, b.skp, kSkp);
            fdgemm::linear_bwd_input(b.dact, P + o.a, b.dh, Ri, F, D, true, s, gsk, gsk_n);       // + residual path already in dh
        } else {
            const int G = 4 * D;
            const size_t lds = ((size_t)G * D + G + 4 * D) * sizeof(float);
            static unsigned long long attr[2] = {};
            if (D <= 72) {
                if (fd_first_on_device(attr[0], ctx->device))
                    FD_HIP(ctx, hipFuncSetAttribute((const void*)k_lstm_bwd_rec<72>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipLaunchKernelGGL((k_lstm_bwd_rec<72>), dim3(B), dim3(4 * 72), lds, s, b.dh, b.act[i], b.aux[i], P + o.b, b.dact, T, D);
            } else {
                if (lds > 160 * 1024) return fd_fail(ctx, FD_ERR_UNSUPPORTED, "LSTM backward: W_hh of d_model %d exceeds the LDS", D);
                if (fd_first_on_device(attr[1], ctx->device))
                    FD_HIP(ctx, hipFuncSetAttribute((const void*)k_lstm_bwd_rec<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipLaunchKernelGGL((k_lstm_bwd_rec<128>), dim3(B), dim3(4 * 128), lds, s, b.dh, b.act[i], b.aux[i], P + o.b, b.dact, T, D);
            }
            // b_ih and b_hh enter the pre-activations identically
            if (int rc = fd_colsum_det(ctx, b.dact, grads + o.c, Ri, G, s)) return rc;
            if (int rc = fd_colsum_det(ctx, b.dact, grads + o.d, Ri, G, s)) return rc;
            fdgemm::linear_bwd_weight(b.dact, b.h[i], grads + o.a, Ri, G, D, true, s, b.skp, kSkp);
            hipLaunchKernelGGL(k_shift_time, dim3(ew_grid(ctx, R * D)), dim3(256), 0, s, b.hs[i], b.hprev, B, T, D);
            fdgemm::linear_bwd_weight(b.dact, b.hprev, grads + o.b, Ri, G, D, true, s, b.skp, kSkp);
            fdgemm::linear_bwd_input(b.dact, P + o.a, b.dh, Ri, G, D, true, s);                    // + residual path already in dh
        }
    }
    // ---- embedder + time embedding (no positional table in these backbones)
    const int Cin = mlp ? T * C : C;
    fdgemm::linear_bwd_weight(b.dh, m->saved_x, grads + m->emb_w, Ri, D, Cin, true, s, b.skp, kSkp);
    if (int rc = fd_colsum_det(ctx, b.dh, grads + m->emb_b, Ri, D, s)) return rc;
    const float* dtemb = b.dh;
    if (!mlp) {
        const size_t n = (size_t)B * D;
        hipLaunchKernelGGL(k_sum_time, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, b.dh, b.dtemb, B, T, D);
        dtemb = b.dtemb;
    }
    fdgemm::linear_bwd_weight(dtemb, b.emb, grads + m->td_w, B, D, D, true, s, b.skp, kSkp);
    if (int rc = fd_colsum_det(ctx, dtemb, grads + m->td_b, B, D, s)) return rc;
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}
