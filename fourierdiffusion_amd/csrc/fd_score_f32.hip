// fd_score_f32.hip -- score network object + the exact-f32 forward (parity / training path).
//
// Reference: ScoreModule.forward (src/fdiff/models/score_models.py:67-94), PositionalEncoding
// (src/fdiff/models/transformer.py:8-29), GaussianFourierProjection (transformer.py:61-91) and
// torch's nn.TransformerEncoderLayer(batch_first, post-LN, relu, dim_ff) built at
// score_models.py:57-62.  SURVEY.md A.3 is the op-by-op specification this file follows.
#include <algorithm>

#include "fd_gemm_f32.h"
#include "fd_philox.h"
#include <hip/hip_ext.h>

#include "fd_score.h"

// ------------------------------------------------------------------ layout
namespace {

struct LayoutBuilder {
    int64_t off = 0;
    std::vector<fd_param_entry>* out;
    int64_t add(const std::string& name, int rows, int cols, int trainable = 1) {
        const int64_t numel = (int64_t)rows * (cols ? cols : 1);
        const int64_t at = off;
        if (out) {
            fd_param_entry e;
            memset(&e, 0, sizeof e);
            snprintf(e.name, sizeof e.name, "%s", name.c_str());
            e.offset = at;
            e.numel = numel;
            e.rows = rows;
            e.cols = cols;
            e.trainable = trainable;
            out->push_back(e);
        }
        off += (numel + 3) & ~int64_t(3);   // every tensor starts on a 16-byte boundary
        return at;
    }
};

int64_t build_layout(const fd_model_dims& d, fd_score* m, std::vector<fd_param_entry>* entries, int backbone = 0, int d_mlp = 0) {
    LayoutBuilder lb;
    lb.out = entries;
    const int D = d.d_model, C = d.n_channels, T = d.max_len, F = d.dim_ff;
    // MLP: the series is flattened, embed / unembed work on T*C features; MLP and LSTM have no positional table
    const int Cin = (backbone == FD_BACKBONE_MLP) ? T * C : C;
    int64_t pos = (backbone == FD_BACKBONE_TRANSFORMER) ? lb.add("pos_encoder.embedding.weight", T, D) : 0;
    int64_t tW = lb.add("time_encoder.W", (D + 1) / 2, 0, 0);
    int64_t td_w = lb.add("time_encoder.dense.weight", D, D);
    int64_t td_b = lb.add("time_encoder.dense.bias", D, 0);
    int64_t emb_w = lb.add("embedder.weight", D, Cin);
    int64_t emb_b = lb.add("embedder.bias", D, 0);
    int64_t un_w = lb.add("unembedder.weight", Cin, D);
    int64_t un_b = lb.add("unembedder.bias", Cin, 0);
    if (m) {
        m->pos = pos; m->tW = tW; m->td_w = td_w; m->td_b = td_b;
        m->emb_w = emb_w; m->emb_b = emb_b; m->un_w = un_w; m->un_b = un_b;
        m->layers.clear();
        m->bb.clear();
    }
    for (int i = 0; i < d.num_layers; ++i) {
        const std::string p = "backbone.layers." + std::to_string(i) + ".";
        const std::string q = "backbone." + std::to_string(i) + ".";
        if (backbone == FD_BACKBONE_MLP) {
            // torchvision.ops.MLP = Sequential(Linear, ReLU, Dropout, Linear, Dropout): parameters at indices 0 and 3
            fd_bb_off o;
            o.a = lb.add(q + "0.weight", d_mlp, D);
            o.b = lb.add(q + "0.bias", d_mlp, 0);
            o.c = lb.add(q + "3.weight", D, d_mlp);
            o.d = lb.add(q + "3.bias", D, 0);
            if (m) m->bb.push_back(o);
            continue;
        }
        if (backbone == FD_BACKBONE_LSTM) {
            fd_bb_off o;
            o.a = lb.add(q + "weight_ih_l0", 4 * D, D);
            o.b = lb.add(q + "weight_hh_l0", 4 * D, D);
            o.c = lb.add(q + "bias_ih_l0", 4 * D, 0);
            o.d = lb.add(q + "bias_hh_l0", 4 * D, 0);
            if (m) m->bb.push_back(o);
            continue;
        }
        fd_layer_off lo;
        lo.in_w = lb.add(p + "self_attn.in_proj_weight", 3 * D, D);
        lo.in_b = lb.add(p + "self_attn.in_proj_bias", 3 * D, 0);
        lo.out_w = lb.add(p + "self_attn.out_proj.weight", D, D);
        lo.out_b = lb.add(p + "self_attn.out_proj.bias", D, 0);
        lo.l1_w = lb.add(p + "linear1.weight", F, D);
        lo.l1_b = lb.add(p + "linear1.bias", F, 0);
        lo.l2_w = lb.add(p + "linear2.weight", D, F);
        lo.l2_b = lb.add(p + "linear2.bias", D, 0);
        lo.n1_w = lb.add(p + "norm1.weight", D, 0);
        lo.n1_b = lb.add(p + "norm1.bias", D, 0);
        lo.n2_w = lb.add(p + "norm2.weight", D, 0);
        lo.n2_b = lb.add(p + "norm2.bias", D, 0);
        if (m) m->layers.push_back(lo);
    }
    return lb.off;
}

bool dims_ok(const fd_model_dims* d) {
    return d && d->n_channels > 0 && d->max_len > 0 && d->d_model > 0 && d->n_head > 0 && d->num_layers >= 0 &&
           d->dim_ff > 0 && d->d_model % d->n_head == 0;
}

}  // namespace

extern "C" int64_t fd_score_param_count(const fd_model_dims* dims) {
    if (!dims_ok(dims)) return FD_ERR_ARG;
    return build_layout(*dims, nullptr, nullptr);
}

extern "C" int fd_score_layout(const fd_model_dims* dims, fd_param_entry* entries, int* n_entries) {
    if (!dims_ok(dims) || !n_entries) return FD_ERR_ARG;
    std::vector<fd_param_entry> v;
    build_layout(*dims, nullptr, &v);
    if (entries) {
        if (*n_entries < (int)v.size()) return FD_ERR_ARG;
        memcpy(entries, v.data(), v.size() * sizeof(fd_param_entry));
    }
    *n_entries = (int)v.size();
    return FD_OK;
}

bool bb_ok(const fd_model_dims* d, int backbone, int d_mlp) {
    if (backbone == FD_BACKBONE_TRANSFORMER) return true;
    if (backbone == FD_BACKBONE_MLP) return d_mlp > 0 && d->d_model <= 1024;
    // one thread per gate row, W_hh row in registers; the backward through time keeps W_hh (4 D^2 floats) + 8 D floats in the
    // 160 KiB LDS, which holds up to d_model = 100 -- wider models would run forward and fail in their first training step
    if (backbone == FD_BACKBONE_LSTM) return d->d_model <= 100;
    return false;
}

extern "C" int64_t fd_score_param_count_ex(const fd_model_dims* dims, int backbone, int d_mlp) {
    if (!dims_ok(dims) || !bb_ok(dims, backbone, d_mlp)) return FD_ERR_ARG;
    return build_layout(*dims, nullptr, nullptr, backbone, d_mlp);
}

extern "C" int fd_score_layout_ex(const fd_model_dims* dims, int backbone, int d_mlp, fd_param_entry* entries, int* n_entries) {
    if (!dims_ok(dims) || !n_entries || !bb_ok(dims, backbone, d_mlp)) return FD_ERR_ARG;
    std::vector<fd_param_entry> v;
    build_layout(*dims, nullptr, &v, backbone, d_mlp);
    if (entries) {
        if (*n_entries < (int)v.size()) return FD_ERR_ARG;
        memcpy(entries, v.data(), v.size() * sizeof(fd_param_entry));
    }
    *n_entries = (int)v.size();
    return FD_OK;
}

extern "C" int fd_score_create_ex(fd_ctx* ctx, const fd_model_dims* dims, int backbone, int d_mlp, fd_score** out) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, out != nullptr, "fd_score_create: null out");
    FD_REQUIRE(ctx, dims_ok(dims), "fd_score_create: bad dims (need positive sizes and d_model %% n_head == 0)");
    FD_REQUIRE(ctx, bb_ok(dims, backbone, d_mlp), "fd_score_create_ex: backbone %d unsupported at d_model=%d d_mlp=%d (MLP: d_mlp > 0; "
               "LSTM: d_model <= 128)", backbone, dims->d_model, d_mlp);
    FD_REQUIRE(ctx, backbone != FD_BACKBONE_TRANSFORMER || dims->d_model / dims->n_head <= 64,
               "fd_score_create: head_dim %d > 64 unsupported", dims->d_model / dims->n_head);
    FD_REQUIRE(ctx, dims->d_model <= 1024, "fd_score_create: d_model %d > 1024 unsupported", dims->d_model);
    fd_score* m = new fd_score();
    m->ctx = ctx;
    m->d = *dims;
    m->backbone = backbone;
    m->d_mlp = d_mlp;
    m->nparams = build_layout(*dims, m, nullptr, backbone, d_mlp);
    if (backbone == FD_BACKBONE_TRANSFORMER) {
        int rc = fd_bf16_create(m);
        if (rc != FD_OK) {
            delete m;
            return rc;
        }
    }
    *out = m;
    return FD_OK;
}

extern "C" int fd_score_create(fd_ctx* ctx, const fd_model_dims* dims, fd_score** out) {
    return fd_score_create_ex(ctx, dims, FD_BACKBONE_TRANSFORMER, 0, out);
}

extern "C" int fd_score_destroy(fd_score* m) {
    if (!m) return FD_ERR_ARG;
    fd_bf16_destroy(m);
    if (m->prep_event) (void)hipEventDestroy(m->prep_event);
    if (m->img_event) (void)hipEventDestroy(m->img_event);
    delete m;
    return FD_OK;
}

// ------------------------------------------------------------------ kernels
namespace {

// torch embedding_renorm_ (nn.Embedding(max_norm=sqrt(D)), transformer.py:13-15): rows with
// ||row|| > max_norm are scaled IN PLACE by max_norm/(norm+1e-7).  One wave per row.
__global__ __launch_bounds__(64) void k_renorm_rows(float* __restrict__ P, int T, int D, float max_norm) {
    const int row = blockIdx.x;
    float* r = P + (size_t)row * D;
    float ss = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) ss += r[d] * r[d];
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float nrm = sqrtf(ss);
    if (nrm > max_norm) {
        const float sc = max_norm / (nrm + 1e-7f);
        for (int d = threadIdx.x; d < D; d += 64) r[d] *= sc;
    }
}

// Gaussian-Fourier features + dense (transformer.py:80-89).  The phase is formed in float32 in the
// reference's op order ((t*W)*2)*pi; sinf/cosf are the full-range-reduction OCML versions.
__global__ __launch_bounds__(128) void k_time_embed(const float* __restrict__ t, const float* __restrict__ W,
                                                     const float* __restrict__ Wd, const float* __restrict__ bd,
                                                     float* __restrict__ emb_out, float* __restrict__ temb, int D) {
    extern __shared__ float emb[];
    const int b = blockIdx.x;
    const int half = (D + 1) / 2;
    const float tb = t[b];
    for (int j = threadIdx.x; j < D; j += blockDim.x) {
        const int jj = (j < half) ? j : j - half;
        const float ph = ((tb * W[jj]) * 2.0f) * 3.14159274101257324f;   // float32(np.pi)
        const float v = (j < half) ? sinf(ph) : cosf(ph);
        emb[j] = v;
        if (emb_out) emb_out[(size_t)b * D + j] = v;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float acc = bd[d];
        const float* w = Wd + (size_t)d * D;
        for (int j = 0; j < D; ++j) acc = fmaf(w[j], emb[j], acc);
        temb[(size_t)b * D + d] = acc;
    }
}

// h[b,t,:] = x[b,t,:] . We^T + be + pe[t,:] + temb[b,:]     (score_models.py:78-84)
// One block = kEmbRows consecutive (b, t) rows: We (D, C) is transposed into LDS as [c][d] ONCE per block (consecutive
// threads = consecutive d read consecutive words) and the block's x rows are staged next to it; one thread per output
// element, fma order over c unchanged.  (One block per 256 outputs re-staged We 18 432 times at T = 1024, B = 64: 36 us.)
constexpr int kEmbRows = 64;
__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ x, const float* __restrict__ We,
                                                const float* __restrict__ be, const float* __restrict__ pe,
                                                const float* __restrict__ temb, float* __restrict__ h, int M, int T,
                                                int C, int D, int use_lds) {
    extern __shared__ __attribute__((aligned(16))) float wsh[];
    if (use_lds) {
        float* const xs = wsh + D * C;
        const int row0 = blockIdx.x * kEmbRows, nrows = min(kEmbRows, M - row0);
        for (int i = threadIdx.x; i < D * C; i += 256) {
            const int d = i / C, c = i - d * C;
            wsh[c * D + d] = We[i];
        }
        for (int i = threadIdx.x; i < nrows * C; i += 256) xs[i] = x[(size_t)row0 * C + i];
        __syncthreads();
        if (use_lds == 2) {
            // four consecutive features per thread: one (row, b, t) decomposition, 16-byte LDS / global accesses (one output
            // per thread spent most of its ~150 instructions on the two integer divisions: 23 us at 65 536 tokens)
            const int D4 = D >> 2;
            for (int i = threadIdx.x; i < nrows * D4; i += 256) {
                const int r = i / D4, d = (i - r * D4) * 4, m = row0 + r;
                const int b = m / T, tt = m - b * T;
                const float* xr = xs + r * C;
                float4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int c = 0; c < C; ++c) {
                    const float xv = xr[c];
                    const float4 w = *reinterpret_cast<const float4*>(wsh + c * D + d);
                    acc.x = fmaf(xv, w.x, acc.x); acc.y = fmaf(xv, w.y, acc.y);
                    acc.z = fmaf(xv, w.z, acc.z); acc.w = fmaf(xv, w.w, acc.w);
                }
                const float4 bv = *reinterpret_cast<const float4*>(be + d);
                const float4 pv = pe ? *reinterpret_cast<const float4*>(pe + (size_t)tt * D + d) : float4{0.f, 0.f, 0.f, 0.f};
                const float4 tv = *reinterpret_cast<const float4*>(temb + (size_t)b * D + d);
                float4 o;
                o.x = ((acc.x + bv.x) + pv.x) + tv.x; o.y = ((acc.y + bv.y) + pv.y) + tv.y;
                o.z = ((acc.z + bv.z) + pv.z) + tv.z; o.w = ((acc.w + bv.w) + pv.w) + tv.w;
                *reinterpret_cast<float4*>(h + (size_t)m * D + d) = o;
            }
            return;
        }
        for (int i = threadIdx.x; i < nrows * D; i += 256) {
            const int r = i / D, d = i - r * D, m = row0 + r;
            const int b = m / T, tt = m - b * T;
            const float* xr = xs + r * C;
            float acc = 0.f;
            for (int c = 0; c < C; ++c) acc = fmaf(xr[c], wsh[c * D + d], acc);
            h[(size_t)m * D + d] = ((acc + be[d]) + (pe ? pe[(size_t)tt * D + d] : 0.f)) + temb[(size_t)b * D + d];
        }
        return;
    }
    const size_t id = blockIdx.x * (size_t)256 + threadIdx.x;
    if (id >= (size_t)M * D) return;
    const int m = (int)(id / D), d = (int)(id % D);
    const int b = m / T, tt = m % T;
    const float* xr = x + (size_t)m * C;
    const float* w = We + (size_t)d * C;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(xr[c], w[c], acc);
    h[id] = ((acc + be[d]) + (pe ? pe[(size_t)tt * D + d] : 0.f)) + temb[(size_t)b * D + d];
}

// The same sum with the weights in registers: a thread owns four consecutive features (its 4 x C slice of We, read once from
// global memory as it lies: no transpose, no LDS) and walks rows; the 16-byte loads of a row's x are shared by the D / 4 threads of
// the row.  fma order over c as in k_embed.  (The LDS form waits for an LDS round trip per channel: 21 us at 65 536 tokens, C = 16;
// this one streams: C % 4 == 0, D % 4 == 0, 16-byte aligned operands.)
template <int CQ>
__global__ __launch_bounds__(256) void k_embed_rows(const float* __restrict__ x, const float* __restrict__ We,
                                                     const float* __restrict__ be, const float* __restrict__ pe,
                                                     const float* __restrict__ temb, float* __restrict__ h, int M, int T, int D) {
    constexpr int C = 4 * CQ;
    const int D4 = D >> 2, RP = 256 / D4;                  // rows per pass of the block
    const int fg = threadIdx.x % D4, rl = threadIdx.x / D4;
    if (rl >= RP) return;
    const int d = 4 * fg;
    float4 w[4][CQ];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < CQ; ++q) w[j][q] = *reinterpret_cast<const float4*>(We + (size_t)(d + j) * C + 4 * q);
    const float4 bv = *reinterpret_cast<const float4*>(be + d);
    for (int m = blockIdx.x * RP + rl; m < M; m += gridDim.x * RP) {
        const int b = m / T, tt = m - b * T;
        float4 xv[CQ];
#pragma unroll
        for (int q = 0; q < CQ; ++q) xv[q] = *reinterpret_cast<const float4*>(x + (size_t)m * C + 4 * q);
        const float4 pv = pe ? *reinterpret_cast<const float4*>(pe + (size_t)tt * D + d) : float4{0.f, 0.f, 0.f, 0.f};
        const float4 tv = *reinterpret_cast<const float4*>(temb + (size_t)b * D + d);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = fmaf(xv[q].x, w[j][q].x, acc[j]);
                acc[j] = fmaf(xv[q].y, w[j][q].y, acc[j]);
                acc[j] = fmaf(xv[q].z, w[j][q].z, acc[j]);
                acc[j] = fmaf(xv[q].w, w[j][q].w, acc[j]);
            }
        }
        float4 o;
        o.x = ((acc[0] + bv.x) + pv.x) + tv.x; o.y = ((acc[1] + bv.y) + pv.y) + tv.y;
        o.z = ((acc[2] + bv.z) + pv.z) + tv.z; o.w = ((acc[3] + bv.w) + pv.w) + tv.w;
        *reinterpret_cast<float4*>(h + (size_t)m * D + d) = o;       // (cached: layer 0 reads it next; nontemporal 14.9 vs 17.8 us here,
                                                                      //  but the same store in k_ffn_ln cost the next attention kernel 3 us)
    }
}

// Multi-head attention core for one (b, h, 64-query block): lane per query, online softmax over 32-key tiles in LDS.
// The key tiles are dealt round-robin to the NWV waves of the workgroup and the per-wave (max, sum, output) states are
// merged through LDS at the end: at the training batch (B=64, T=100) one wave per block left each SIMD with 1-2 waves
// walking 100 keys serially (56 us per layer); four waves per block quarter the chain and give the SIMDs 6 waves.
// qkv: (M, 3D) rows [q | k | v]; out: (M, D) heads concatenated (torch MHA layout).
template <int HDP, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_attention_f32(const float* __restrict__ qkv, float* __restrict__ out,
                                                              float* __restrict__ lse, int T, int H, int hd, float scale,
                                                              float drop_p, uint64_t seed, uint64_t offset) {
    constexpr int KT = 32;
    __shared__ float Ks[NWV][KT][HDP];
    __shared__ float Vs[NWV][KT][HDP];
    __shared__ float part[NWV][HDP + 2][64];
    const int D = H * hd;
    const int b = blockIdx.z, h = blockIdx.y;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 64 + lane;
    const bool active = q < T;
    const size_t row0 = (size_t)b * T;
    float qr[HDP];
#pragma unroll
    for (int d = 0; d < HDP; ++d) qr[d] = 0.f;
    if (active) {
        const float* qp = qkv + (row0 + q) * 3 * D + h * hd;
        for (int d = 0; d < hd; ++d) qr[d] = qp[d] * scale;
    }
    float mrun = -INFINITY, l = 0.f;
    float o[HDP];
#pragma unroll
    for (int d = 0; d < HDP; ++d) o[d] = 0.f;
    const float keep_scale = (drop_p > 0.f) ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int groups_per_row = (T + 3) / 4;
    // attention-probability dropout (nn.MultiheadAttention dropout=0.1): one Philox counter per 4 consecutive keys of a
    // (b,h,q) row, evaluated ONCE per group (a Philox4x32-10 is 40 quarter-rate integer multiplies, ~900 cycles)
    const uint64_t row_ctr = offset + ((uint64_t)(b * H + h) * T + (active ? q : 0)) * groups_per_row;

    for (int k0 = w * KT; k0 < T; k0 += NWV * KT) {           // this wave's tiles; its LDS slice is private: no barrier
        const int kn = min(KT, T - k0);
        for (int id = lane; id < KT * HDP; id += 64) {
            const int j = id / HDP, d = id % HDP;
            float kv = 0.f, vv = 0.f;
            if (j < kn && d < hd) {
                const float* base = qkv + (row0 + k0 + j) * 3 * D + h * hd + d;
                kv = base[D];
                vv = base[2 * D];
            }
            Ks[w][j][d] = kv;
            Vs[w][j][d] = vv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float s[KT];
        float mt = -INFINITY;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HDP; ++d) a = fmaf(qr[d], Ks[w][j][d], a);
            s[j] = (j < kn) ? a : -INFINITY;
            mt = fmaxf(mt, s[j]);
        }
        const float mnew = fmaxf(mrun, mt);
        const float alpha = __expf(mrun - mnew);   // exp(-inf) = 0 on the first tile
        l *= alpha;
#pragma unroll
        for (int d = 0; d < HDP; ++d) o[d] *= alpha;
#pragma unroll
        for (int j4 = 0; j4 < KT; j4 += 4) {
            uint32_t rv[4] = {0u, 0u, 0u, 0u};
            if (drop_p > 0.f) {
                const fd_u4 r = fd_philox4x32_10(row_ctr + (uint64_t)((k0 + j4) >> 2), seed);
                rv[0] = r.x; rv[1] = r.y; rv[2] = r.z; rv[3] = r.w;
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j4 + jj;
                const float p = expf(s[j] - mnew);
                l += p;
                float pk = p;
                if (drop_p > 0.f) pk = (fd_u01(rv[jj]) >= drop_p) ? p * keep_scale : 0.f;
#pragma unroll
                for (int d = 0; d < HDP; ++d) o[d] = fmaf(pk, Vs[w][j][d], o[d]);
            }
        }
        mrun = mnew;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // tile reads done before the next fill
    }
    if (NWV > 1) {                                             // merge the waves' softmax states (wave 0 always has a tile)
        part[w][0][lane] = mrun;
        part[w][1][lane] = l;
#pragma unroll
        for (int d = 0; d < HDP; ++d) part[w][2 + d][lane] = o[d];
        __syncthreads();
        if (w != 0) return;
        float mm = mrun;
#pragma unroll
        for (int v = 1; v < NWV; ++v) mm = fmaxf(mm, part[v][0][lane]);
        const float a0 = __expf(mrun - mm);
        l *= a0;
#pragma unroll
        for (int d = 0; d < HDP; ++d) o[d] *= a0;
#pragma unroll
        for (int v = 1; v < NWV; ++v) {
            const float mv = part[v][0][lane];
            const float av = (mv == -INFINITY) ? 0.f : __expf(mv - mm);
            l = fmaf(part[v][1][lane], av, l);
#pragma unroll
            for (int d = 0; d < HDP; ++d) o[d] = fmaf(part[v][2 + d][lane], av, o[d]);
        }
        mrun = mm;
    }
    if (active) {
        const float inv = 1.0f / l;
        float* op = out + (row0 + q) * D + h * hd;
        for (int d = 0; d < hd; ++d) op[d] = o[d] * inv;
        if (lse) lse[((size_t)b * H + h) * T + q] = mrun + logf(l);
    }
}

// y = LayerNorm(a + r) * gamma + beta (eps 1e-5, biased variance); optionally keep the pre-norm sum and
// (mean, rstd) for the backward pass.  One wave per token.
__global__ __launch_bounds__(256) void k_add_layernorm(const float* __restrict__ a, const float* __restrict__ r,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ sum_out,
                                                        float* __restrict__ mr_out, float* __restrict__ y, int M,
                                                        int D) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (m >= M) return;
    const float* ap = a + (size_t)m * D;
    const float* rp = r + (size_t)m * D;
    float v[16];   // D <= 1024
    float s = 0.f;
    int n = 0;
    for (int d = lane; d < D; d += 64, ++n) {
        v[n] = ap[d] + rp[d];
        s += v[n];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float q = 0.f;
    for (int i = 0; i < n; ++i) {
        const float c = v[i] - mean;
        q += c * c;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)D + 1e-5f);
    n = 0;
    for (int d = lane; d < D; d += 64, ++n) {
        if (sum_out) sum_out[(size_t)m * D + d] = v[n];
        y[(size_t)m * D + d] = (v[n] - mean) * rstd * gamma[d] + beta[d];
    }
    if (mr_out && lane == 0) {
        mr_out[(size_t)m * 2] = mean;
        mr_out[(size_t)m * 2 + 1] = rstd;
    }
}

}  // namespace

// In-place inverted dropout, 4 elements per Philox counter (exported for the backward pass).
__global__ __launch_bounds__(256) void fd_k_dropout(float* __restrict__ x, size_t n, float p, uint64_t seed,
                                                     uint64_t offset) {
    const size_t ng = (n + 3) / 4;
    const float sc = 1.0f / (1.0f - p);
    for (size_t g = blockIdx.x * (size_t)256 + threadIdx.x; g < ng; g += (size_t)gridDim.x * 256) {
        const fd_u4 r = fd_philox4x32_10(offset + g, seed);
        const uint32_t rv[4] = {r.x, r.y, r.z, r.w};
        for (int i = 0; i < 4; ++i) {
            const size_t e = g * 4 + i;
            if (e < n) x[e] = (fd_u01(rv[i]) >= p) ? x[e] * sc : 0.f;
        }
    }
}

void fd_dropout_inplace(fd_ctx* ctx, float* x, size_t n, float p, uint64_t seed, uint64_t offset, hipStream_t s) {
    if (p <= 0.f) return;
    size_t blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > (size_t)ctx->num_cu * 8) blocks = (size_t)ctx->num_cu * 8;
    hipLaunchKernelGGL(fd_k_dropout, dim3((unsigned)blocks), dim3(256), 0, s, x, n, p, seed, offset);
}

// Philox counter ranges of the dropout sites (disjoint per layer/site; each <= 2^40 counters)
uint64_t fd_dropout_site_offset(uint64_t base, int layer, int site) {
    return base + (((uint64_t)layer * 4 + (uint64_t)site) << 40);
}

// thin launch wrappers reused by the bf16 hybrid path (fd_score_bf16.hip)
namespace fdf32 {
void time_embed(const float* t, const float* W, const float* Wd, const float* bd, float* temb, int B, int D,
                hipStream_t s) {
    hipLaunchKernelGGL(k_time_embed, dim3(B), dim3(128), D * sizeof(float), s, t, W, Wd, bd, (float*)nullptr, temb,
                       D);
}
// The weights-in-registers form for a channel count that is not a multiple of four (the reference's datasets: 1, 5, 13 channels;
// BASELINE nasdaq: 6): scalar loads of the 4 x C weight slice and of the row's x, the same fma order over c -- bit-identical to
// k_embed, which took 25 us at 16 128 tokens, C = 6 (one LDS round trip per channel) against ~10 us here.
template <int C>
__global__ __launch_bounds__(256) void k_embed_rows_c(const float* __restrict__ x, const float* __restrict__ We,
                                                       const float* __restrict__ be, const float* __restrict__ pe,
                                                       const float* __restrict__ temb, float* __restrict__ h, int M, int T, int D) {
    const int D4 = D >> 2, RP = 256 / D4;
    const int fg = threadIdx.x % D4, rl = threadIdx.x / D4;
    if (rl >= RP) return;
    const int d = 4 * fg;
    float w[4][C];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) w[j][c] = We[(size_t)(d + j) * C + c];
    const float4 bv = *reinterpret_cast<const float4*>(be + d);
    for (int m = blockIdx.x * RP + rl; m < M; m += gridDim.x * RP) {
        const int b = m / T, tt = m - b * T;
        float xv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) xv[c] = x[(size_t)m * C + c];
        const float4 pv = pe ? *reinterpret_cast<const float4*>(pe + (size_t)tt * D + d) : float4{0.f, 0.f, 0.f, 0.f};
        const float4 tv = *reinterpret_cast<const float4*>(temb + (size_t)b * D + d);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv[c], w[j][c], acc[j]);
        float4 o;
        o.x = ((acc[0] + bv.x) + pv.x) + tv.x; o.y = ((acc[1] + bv.y) + pv.y) + tv.y;
        o.z = ((acc[2] + bv.z) + pv.z) + tv.z; o.w = ((acc[3] + bv.w) + pv.w) + tv.w;
        *reinterpret_cast<float4*>(h + (size_t)m * D + d) = o;
    }
}
void embed(const float* x, const float* We, const float* be, const float* pe, const float* temb, float* h, int M,
           int T, int C, int D, hipStream_t s) {
    const size_t n = (size_t)M * D;
    const size_t lds = ((size_t)D * C + (size_t)kEmbRows * C) * sizeof(float);
    int use_lds = (lds <= 48 * 1024) ? 1 : 0;
    // 16-byte form: D % 4 == 0 and every vector operand 16-byte aligned (the parameter offsets are not when d_model / 2 is odd)
    const uintptr_t al = (uintptr_t)be | (uintptr_t)pe | (uintptr_t)temb | (uintptr_t)h;
    if (use_lds && (D & 3) == 0 && (al & 15) == 0) use_lds = 2;
    // weights-in-registers form: C % 4 == 0 (<= 40), D % 4 == 0 (<= 256), every vector operand 16-byte aligned
    const uintptr_t al2 = al | (uintptr_t)x | (uintptr_t)We;
    if ((C & 3) == 0 && C <= 40 && (D & 3) == 0 && D <= 256 && (al2 & 15) == 0 && !getenv("FDIFF_EMBED_LDS")) {
        const int RP = 256 / (D >> 2);
        static int ncu = 0;                          // (one chip type per process)
        if (!ncu) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
            else ncu = 256;
        }
        const long long want = ((long long)M + RP - 1) / RP, cap = (long long)ncu * 4;   // four co-resident blocks per CU walk the rows
        const unsigned g = (unsigned)(want < cap ? want : cap);
#define FD_EMB(CQ_) case CQ_: hipLaunchKernelGGL(k_embed_rows<CQ_>, dim3(g), dim3(256), 0, s, x, We, be, pe, temb, h, M, T, D); return;
        switch (C >> 2) {
            FD_EMB(1) FD_EMB(2) FD_EMB(3) FD_EMB(4) FD_EMB(5) FD_EMB(6) FD_EMB(7) FD_EMB(8) FD_EMB(9) FD_EMB(10)
        }
#undef FD_EMB
    }
    if ((C & 3) != 0 && C <= 15 && (D & 3) == 0 && D <= 256 && (al & 15) == 0 && !getenv("FDIFF_EMBED_LDS")) {
        const int RP = 256 / (D >> 2);
        int dev = 0, ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const long long want = ((long long)M + RP - 1) / RP, cap = (long long)(ncu > 0 ? ncu : 256) * 4;
        const unsigned g = (unsigned)(want < cap ? want : cap);
#define FD_EMBC(C_) case C_: hipLaunchKernelGGL(k_embed_rows_c<C_>, dim3(g), dim3(256), 0, s, x, We, be, pe, temb, h, M, T, D); return;
        switch (C) {
            FD_EMBC(1) FD_EMBC(2) FD_EMBC(3) FD_EMBC(5) FD_EMBC(6) FD_EMBC(7) FD_EMBC(9) FD_EMBC(10) FD_EMBC(11) FD_EMBC(13) FD_EMBC(14) FD_EMBC(15)
        }
#undef FD_EMBC
    }
    const unsigned grid = use_lds ? (unsigned)((M + kEmbRows - 1) / kEmbRows) : (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_embed, dim3(grid), dim3(256), use_lds ? lds : 0, s, x, We, be, pe, temb, h, M, T, C, D, use_lds);
}
void add_layernorm(const float* a, const float* r, const float* gamma, const float* beta, float* y, int M, int D,
                   hipStream_t s) {
    hipLaunchKernelGGL(k_add_layernorm, dim3((M + 3) / 4), dim3(256), 0, s, a, r, gamma, beta, (float*)nullptr,
                       (float*)nullptr, y, M, D);
}
}  // namespace fdf32

// ------------------------------------------------------------------ workspace
namespace {
inline size_t fl(size_t n) { return fd_ws::padded(n * sizeof(float)); }
}

size_t fd_score_f32_workspace(const fd_score* m, int B, bool train) {
    const size_t M = (size_t)B * m->d.max_len, D = m->d.d_model, F = m->d.dim_ff, H = m->d.n_head;
    const size_t T = m->d.max_len;
    size_t per_layer = fl(M * D) /*x0*/ + fl(M * 3 * D) + fl((size_t)B * H * T) + fl(M * D) /*att*/ +
                       fl(M * D) /*s1*/ + fl(M * 2) + fl(M * D) /*x1*/ + fl(M * F) + fl(M * D) /*s2*/ + fl(M * 2);
    size_t fixed = 2 * fl((size_t)B * D) + fl(M * D) /*hL*/ + fl(M * D) /*tmp proj*/;
    return fixed + per_layer * (train ? (size_t)std::max(1, m->d.num_layers) : 1) + 4096;
}

void fd_score_carve_saved(const fd_score* m, int B, fd_ws& ws, fd_saved& sv) {
    const size_t M = (size_t)B * m->d.max_len, D = m->d.d_model, F = m->d.dim_ff, H = m->d.n_head;
    const size_t T = m->d.max_len;
    sv.emb = ws.take<float>((size_t)B * D);
    sv.temb = ws.take<float>((size_t)B * D);
    sv.hL = ws.take<float>(M * D);
    sv.layers.resize(m->d.num_layers);
    for (auto& L : sv.layers) {
        L.x0 = ws.take<float>(M * D);
        L.qkv = ws.take<float>(M * 3 * D);
        L.lse = ws.take<float>((size_t)B * H * T);
        L.att = ws.take<float>(M * D);
        L.s1 = ws.take<float>(M * D);
        L.mr1 = ws.take<float>(M * 2);
        L.x1 = ws.take<float>(M * D);
        L.hact = ws.take<float>(M * F);
        L.s2 = ws.take<float>(M * D);
        L.mr2 = ws.take<float>(M * 2);
    }
}

template <int HDP>
static void launch_attn(const float* qkv, float* out, float* lse, int B, int T, int H, int hd, float drop_p,
                        uint64_t seed, uint64_t offset, hipStream_t s) {
    constexpr int NWV = HDP <= 16 ? 4 : 1;                     // (LDS: K | V tiles + merge area per wave)
    dim3 grid((T + 63) / 64, H, B);
    hipLaunchKernelGGL((k_attention_f32<HDP, NWV>), grid, dim3(64 * NWV), 0, s, qkv, out, lse, T, H, hd,
                       1.0f / sqrtf((float)hd), drop_p, seed, offset);
}

void fd_attention_f32(const float* qkv, float* out, float* lse, int B, int T, int H, int hd, float drop_p,
                      uint64_t seed, uint64_t offset, hipStream_t s) {
    if (hd <= 8) launch_attn<8>(qkv, out, lse, B, T, H, hd, drop_p, seed, offset, s);
    else if (hd <= 16) launch_attn<16>(qkv, out, lse, B, T, H, hd, drop_p, seed, offset, s);
    else if (hd <= 32) launch_attn<32>(qkv, out, lse, B, T, H, hd, drop_p, seed, offset, s);
    else launch_attn<64>(qkv, out, lse, B, T, H, hd, drop_p, seed, offset, s);
}

// ------------------------------------------------------------------ forward
int fd_score_forward_f32(fd_score* m, const float* x, const float* t, float* out, int B, hipStream_t s, bool train,
                         float p, uint64_t seed, uint64_t offset) {
    fd_ctx* ctx = m->ctx;
    const int T = m->d.max_len, C = m->d.n_channels, D = m->d.d_model, H = m->d.n_head, F = m->d.dim_ff;
    const int L = m->d.num_layers, hd = D / H;
    const int M = B * T;
    const float* P = m->params;
    // a training forward also reserves the backward pass's scratch: growing the arena later would move it
    if (int rc = fd_ws_reserve(ctx, fd_score_f32_workspace(m, B, train) + (train ? fd_score_bwd_workspace(m, B) : 0)))
        return rc;
    fd_ws ws(ctx);
    size_t gsk_n = 0;
    float* gsk = fd_gemm_scratch(ctx, &gsk_n);
    fd_saved sv;
    fd_saved_layer scratch{};
    float* tmp;   // (M,D) projection scratch
    if (train) {
        fd_score_carve_saved(m, B, ws, sv);
        tmp = ws.take<float>((size_t)M * D);
    } else {
        sv.emb = nullptr;
        sv.temb = ws.take<float>((size_t)B * D);
        (void)ws.take<float>((size_t)B * D);
        sv.hL = ws.take<float>((size_t)M * D);
        scratch.x0 = ws.take<float>((size_t)M * D);
        scratch.qkv = ws.take<float>((size_t)M * 3 * D);
        scratch.lse = nullptr;
        (void)ws.take<float>((size_t)B * H * T);
        scratch.att = ws.take<float>((size_t)M * D);
        scratch.s1 = nullptr;
        (void)ws.take<float>((size_t)M * D);
        scratch.mr1 = nullptr;
        (void)ws.take<float>((size_t)M * 2);
        scratch.x1 = ws.take<float>((size_t)M * D);
        scratch.hact = ws.take<float>((size_t)M * F);
        scratch.s2 = nullptr;
        (void)ws.take<float>((size_t)M * D);
        scratch.mr2 = nullptr;
        (void)ws.take<float>((size_t)M * 2);
        tmp = ws.take<float>((size_t)M * D);
    }

    hipLaunchKernelGGL(k_time_embed, dim3(B), dim3(128), D * sizeof(float), s, t, P + m->tW, P + m->td_w,
                       P + m->td_b, sv.emb, sv.temb, D);
    float* h_in = (L > 0) ? (train ? sv.layers[0].x0 : scratch.x0) : sv.hL;
    fdf32::embed(x, P + m->emb_w, P + m->emb_b, P + m->pos, sv.temb, h_in, M, T, C, D, s);
    for (int i = 0; i < L; ++i) {
        const fd_layer_off& lo = m->layers[i];
        fd_saved_layer& A = train ? sv.layers[i] : scratch;
        float* x0 = A.x0;
        // next layer's input buffer (eval: ping-pong x0 <- hL is avoided by writing LN2 straight into x0)
        float* x_next = (i + 1 < L) ? (train ? sv.layers[i + 1].x0 : scratch.x0) : sv.hL;
        fdgemm::linear_fwd(x0, P + lo.in_w, P + lo.in_b, A.qkv, M, 3 * D, D, false, s);
        fd_attention_f32(A.qkv, A.att, A.lse, B, T, H, hd, p, seed, fd_dropout_site_offset(offset, i, 0), s);
        // the three dropout sites of the layer ride in the epilogue of the GEMM that produces their input (three launches
        // and a second pass over the (M, F) activations less per layer); same mask as the stand-alone kernel
        const bool dfd = train && p > 0.f && fdgemm::can_fuse_dropout(D), dff = train && p > 0.f && fdgemm::can_fuse_dropout(F);
        if (dfd) {
            fdgemm::linear_fwd_dropout(A.att, P + lo.out_w, P + lo.out_b, tmp, M, D, D, false, p, seed,
                                       fd_dropout_site_offset(offset, i, 1), s);
        } else {
            fdgemm::linear_fwd(A.att, P + lo.out_w, P + lo.out_b, tmp, M, D, D, false, s);
            if (train) fd_dropout_inplace(ctx, tmp, (size_t)M * D, p, seed, fd_dropout_site_offset(offset, i, 1), s);
        }
        hipLaunchKernelGGL(k_add_layernorm, dim3((M + 3) / 4), dim3(256), 0, s, x0, tmp, P + lo.n1_w, P + lo.n1_b,
                           A.s1, A.mr1, A.x1, M, D);
        if (dff) {
            fdgemm::linear_fwd_dropout(A.x1, P + lo.l1_w, P + lo.l1_b, A.hact, M, F, D, true, p, seed,
                                       fd_dropout_site_offset(offset, i, 2), s);
        } else {
            fdgemm::linear_fwd(A.x1, P + lo.l1_w, P + lo.l1_b, A.hact, M, F, D, true, s);
            if (train) fd_dropout_inplace(ctx, A.hact, (size_t)M * F, p, seed, fd_dropout_site_offset(offset, i, 2), s);
        }
        if (dfd) {
            fdgemm::linear_fwd_dropout(A.hact, P + lo.l2_w, P + lo.l2_b, tmp, M, D, F, false, p, seed,
                                       fd_dropout_site_offset(offset, i, 3), s, gsk, gsk_n);   // K = F: split
        } else {
            fdgemm::linear_fwd(A.hact, P + lo.l2_w, P + lo.l2_b, tmp, M, D, F, false, s, gsk, gsk_n);   // K = F: split
            if (train) fd_dropout_inplace(ctx, tmp, (size_t)M * D, p, seed, fd_dropout_site_offset(offset, i, 3), s);
        }
        hipLaunchKernelGGL(k_add_layernorm, dim3((M + 3) / 4), dim3(256), 0, s, A.x1, tmp, P + lo.n2_w, P + lo.n2_b,
                           A.s2, A.mr2, x_next, M, D);
    }
    fdgemm::linear_fwd(sv.hL, P + m->un_w, P + m->un_b, out, M, C, D, false, s);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

// ------------------------------------------------------------------ C ABI
extern "C" int fd_score_prepare(fd_score* m, const float* params, void* stream) {
    if (!m) return FD_ERR_ARG;
    fd_ctx* ctx = m->ctx;
    FD_REQUIRE(ctx, params != nullptr, "fd_score_prepare: null params");
    m->params = const_cast<float*>(params);
    // The reference renorms the looked-up rows of the positional table in place on every forward
    // (nn.Embedding(max_norm), transformer.py:13-15,27); all T rows are looked up, so do it here once.
    m->prep_event_bound = false;
    if (m->backbone == FD_BACKBONE_TRANSFORMER) {
        // (the stop event: the bf16 training forward rebuilds the weight images on a side stream once THIS kernel has run,
        // fd_train_bf16.hip -- an event recorded there would be a packet of its own in front of the step's first kernels)
        if (!m->prep_event && hipEventCreateWithFlags(&m->prep_event, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) m->prep_event = nullptr;
        hipExtLaunchKernelGGL(k_renorm_rows, dim3(m->d.max_len), dim3(64), 0, (hipStream_t)stream, nullptr, m->prep_event, 0,
                              m->params + m->pos, m->d.max_len, m->d.d_model, sqrtf((float)m->d.d_model));
        m->prep_event_bound = m->prep_event != nullptr;
        m->prep_stream = stream;
    }
    FD_LAUNCH_CHECK(ctx);
    m->bf16_stale = true;     // the bf16 MFMA entry points rebuild their images on first use (fd_bf16_refresh)
    m->prepared = true;
    return FD_OK;
}

extern "C" int fd_score_forward(fd_score* m, const float* x, const float* t, float* out, int B, int mode,
                                void* stream) {
    if (!m) return FD_ERR_ARG;
    fd_ctx* ctx = m->ctx;
    FD_REQUIRE(ctx, x && t && out, "fd_score_forward: null pointer");
    FD_REQUIRE(ctx, B > 0, "fd_score_forward: B=%d", B);
    if (!m->prepared) return fd_fail(ctx, FD_ERR_STATE, "fd_score_forward: call fd_score_prepare first");
    if (mode != FD_MODE_F32 && mode != FD_MODE_BF16) return fd_fail(ctx, FD_ERR_ARG, "fd_score_forward: unknown mode %d", mode);
    return fd_score_forward_any(m, x, t, out, B, mode, (hipStream_t)stream);
}

int fd_score_forward_any(fd_score* m, const float* x, const float* t, float* out, int B, int mode, hipStream_t s) {
    if (m->backbone != FD_BACKBONE_TRANSFORMER) return fd_bb_forward(m, x, t, out, B, s, false, 0.f, 0, 0);   // one arithmetic: exact f32
    if (mode == FD_MODE_F32) return fd_score_forward_f32(m, x, t, out, B, s, false, 0.f, 0, 0);
    return fd_score_forward_bf16(m, x, t, out, B, s);
}

extern "C" int fd_score_forward_train(fd_score* m, const float* x, const float* t, float* out, int B,
                                      float dropout_p, uint64_t seed, uint64_t offset, void* stream) {
    if (!m) return FD_ERR_ARG;
    fd_ctx* ctx = m->ctx;
    FD_REQUIRE(ctx, x && t && out, "fd_score_forward_train: null pointer");
    FD_REQUIRE(ctx, B > 0, "fd_score_forward_train: B=%d", B);
    FD_REQUIRE(ctx, dropout_p >= 0.f && dropout_p < 1.f, "fd_score_forward_train: dropout_p=%f", dropout_p);
    if (!m->prepared) return fd_fail(ctx, FD_ERR_STATE, "fd_score_forward_train: call fd_score_prepare first");
    const bool bf16 = m->train_mode == FD_MODE_BF16 && m->backbone == FD_BACKBONE_TRANSFORMER;
    if (bf16 && !fd_train_bf16_supported(m))
        return fd_fail(ctx, FD_ERR_UNSUPPORTED, "fd_score_forward_train: bf16 training kernels are not instantiated for this "
                       "model (needs bf16 weight images -- fd_score_plan says which widths have them --, dim_ff %% 1024 == 0, dim_ff <= 2048, "
                       "max_len <= 1024); select FD_MODE_F32");
    int rc = (m->backbone != FD_BACKBONE_TRANSFORMER)
                 ? fd_bb_forward(m, x, t, out, B, (hipStream_t)stream, true, dropout_p, seed, offset)
                 : (bf16 ? fd_score_forward_train_bf16(m, x, t, out, B, dropout_p, seed, offset, (hipStream_t)stream)
                         : fd_score_forward_f32(m, x, t, out, B, (hipStream_t)stream, true, dropout_p, seed, offset));
    if (rc == FD_OK) {
        m->saved_bf16 = bf16;
        m->have_saved = true;
        m->saved_B = B;
        m->saved_p = dropout_p;
        m->saved_seed = seed;
        m->saved_offset = offset;
        m->saved_x = x;
        m->saved_t = t;
        m->saved_ws_gen = ctx->ws_gen;
        m->saved_ws = ctx->ws;
    }
    return rc;
}

// fd_score_forward_train + fd_dsm_loss + fd_score_backward in one call (include/fdiff_hip.h); only the bf16 transformer path has
// the fused loss head, every other model answers FD_ERR_UNSUPPORTED and the caller runs the three calls.
extern "C" int fd_score_train_dsm(fd_score* m, const float* x, const float* t, const float* target, const float* std,
                                  int likelihood_weighting, float grad_weight, int B, float dropout_p, uint64_t seed, uint64_t offset,
                                  float* loss_out, float* grads, int accumulate, void* stream) {
    if (!m) return FD_ERR_ARG;
    fd_ctx* ctx = m->ctx;
    FD_REQUIRE(ctx, x && t && target && std && loss_out && grads, "fd_score_train_dsm: null pointer");
    FD_REQUIRE(ctx, B > 0, "fd_score_train_dsm: B=%d", B);
    FD_REQUIRE(ctx, dropout_p >= 0.f && dropout_p < 1.f, "fd_score_train_dsm: dropout_p=%f", dropout_p);
    if (!m->prepared) return fd_fail(ctx, FD_ERR_STATE, "fd_score_train_dsm: call fd_score_prepare first");
    if (m->train_mode != FD_MODE_BF16 || m->backbone != FD_BACKBONE_TRANSFORMER || !fd_train_bf16_supported(m) ||
        getenv("FDIFF_TRAIN_DSM_UNFUSED"))
        return fd_fail(ctx, FD_ERR_UNSUPPORTED, "fd_score_train_dsm: the fused step exists on the bf16 transformer training path only");
    return fd_score_train_dsm_bf16(m, x, t, target, std, likelihood_weighting, grad_weight, B, dropout_p, seed, offset, loss_out, grads,
                                   accumulate, (hipStream_t)stream);
}

// 1 when fd_score_train_dsm has a fused step for this (model, train mode, B), 0 when it would answer FD_ERR_UNSUPPORTED; no side
// effects (the host asks BEFORE drawing the step's Philox key, so an unsupported batch consumes no key).
extern "C" int fd_score_train_dsm_supported(fd_score* m, int B) {
    if (!m || B <= 0 || !m->prepared) return 0;
    if (m->train_mode != FD_MODE_BF16 || m->backbone != FD_BACKBONE_TRANSFORMER || getenv("FDIFF_TRAIN_DSM_UNFUSED")) return 0;
    return fd_score_train_dsm_bf16_supported(m, B) ? 1 : 0;
}

// The training launch plan of a batch of B series (no launch): which arithmetic runs and, on the bf16 path, the token splits of
// the weight-gradient kernel and whether the fused loss head exists -- the parity tests assert the plan they exercised.
extern "C" int fd_score_train_plan(fd_score* m, int B, char* out, int* token_splits) {
    if (!m) return FD_ERR_ARG;
    fd_ctx* ctx = m->ctx;
    FD_REQUIRE(ctx, out && B > 0, "fd_score_train_plan: null buffer or B=%d", B);
    if (token_splits) *token_splits = 0;
    if (m->train_mode != FD_MODE_BF16 || m->backbone != FD_BACKBONE_TRANSFORMER || !fd_train_bf16_supported(m)) {
        snprintf(out, 192, "exact-f32 training kernels (fd_score_bwd.hip)");
        return FD_OK;
    }
    int nblk = 0;
    const int ts = fd_train_bf16_token_splits(m, B, &nblk);
    if (token_splits) *token_splits = ts;
    // forward: every layer in one persistent launch (fd_train_persist.hip) where it applies, else two kernels per layer
    char fwd[64];
    fd_train_bf16_forward_plan(m, B, fwd, sizeof fwd);
    snprintf(out, 192, "bf16 training: forward %s; backward 3 kernels per layer, k_tr_wgrad token splits TS=%d over %d 32-token blocks, fused loss head %s",
             fwd, ts, nblk, (m->prepared && fd_score_train_dsm_bf16_supported(m, B) && !getenv("FDIFF_TRAIN_DSM_UNFUSED")) ? "yes" : "no");
    return FD_OK;
}

// Arithmetic of fd_score_forward_train / fd_score_backward: FD_MODE_F32 = exact-f32 kernels (parity anchor, any model),
// FD_MODE_BF16 = bf16 MFMA operands with fp32 accumulation (fd_train_bf16.hip).  Returns FD_ERR_UNSUPPORTED (and keeps the
// previous mode) when the bf16 kernels are not instantiated for the model.
extern "C" int fd_score_set_train_mode(fd_score* m, int mode) {
    if (!m) return FD_ERR_ARG;
    fd_ctx* ctx = m->ctx;
    FD_REQUIRE(ctx, mode == FD_MODE_F32 || mode == FD_MODE_BF16, "fd_score_set_train_mode: unknown mode %d", mode);
    if (mode == FD_MODE_BF16 && (m->backbone != FD_BACKBONE_TRANSFORMER || !fd_train_bf16_supported(m)))
        return fd_fail(ctx, FD_ERR_UNSUPPORTED, "fd_score_set_train_mode: bf16 training kernels not instantiated for this model");
    m->train_mode = mode;
    return FD_OK;
}

int fd_time_embed_train(const float* t, const float* W, const float* Wd, const float* bd, float* emb, float* temb, int B, int D,
                        hipStream_t s) {
    hipLaunchKernelGGL(k_time_embed, dim3(B), dim3(128), D * sizeof(float), s, t, W, Wd, bd, emb, temb, D);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------ stand-alone encoders (transformer.py)
namespace {
__global__ __launch_bounds__(256) void k_bcast_add(const float* __restrict__ x, const float* __restrict__ v,
                                                    float* __restrict__ out, size_t n, int T, int D, int mode) {
    // mode 0: v is (T,D) indexed by the time position; mode 1: v is (B,D) indexed by the batch element
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i % D);
        const size_t row = i / D;
        const size_t vr = (mode == 0) ? (row % (size_t)T) : (T > 0 ? row / (size_t)T : row);
        out[i] = x[i] + v[vr * D + d];
    }
}
}  // namespace

extern "C" int fd_positional_add(fd_ctx* ctx, const float* x, float* table, float* out, int B, int T, int D,
                                 float max_norm, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, x && table && out && B > 0 && T > 0 && D > 0, "fd_positional_add: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_renorm_rows, dim3(T), dim3(64), 0, s, table, T, D, max_norm);
    const size_t n = (size_t)B * T * D;
    size_t blocks = std::min((n + 255) / 256, (size_t)ctx->num_cu * 8);
    hipLaunchKernelGGL(k_bcast_add, dim3((unsigned)blocks), dim3(256), 0, s, x, table, out, n, T, D, 0);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_time_embed_add(fd_ctx* ctx, const float* x, const float* t, const float* W, const float* Wd,
                                 const float* bd, float* out, int B, int T, int D, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, x && t && W && Wd && bd && out && B > 0 && T >= 0 && D > 0, "fd_time_embed_add: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = fd_ws_reserve(ctx, fd_ws::padded((size_t)B * D * sizeof(float)))) return rc;
    float* temb = (float*)ctx->ws;
    hipLaunchKernelGGL(k_time_embed, dim3(B), dim3(128), D * sizeof(float), s, t, W, Wd, bd, (float*)nullptr, temb, D);
    const size_t n = (size_t)B * (T > 0 ? T : 1) * D;
    size_t blocks = std::min((n + 255) / 256, (size_t)ctx->num_cu * 8);
    hipLaunchKernelGGL(k_bcast_add, dim3((unsigned)blocks), dim3(256), 0, s, x, temb, out, n, T, D, 1);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}
