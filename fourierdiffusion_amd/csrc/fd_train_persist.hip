// fd_train_persist.hip -- the bf16 training FORWARD of every encoder layer as ONE persistent launch.
//
// Reference: the forward half of ScoreModule.training_step (src/fdiff/models/score_models.py:96-108) through
// nn.TransformerEncoder (post-LN layers, relu, dropout at the attention probabilities, behind the out-projection, behind relu
// and behind linear2; score_models.py:57-62), i.e. what k_tr_attn_fwd + k_tr_ffn_fwd of fd_train_bf16.hip compute with two
// launches per layer.
//
// Why: at the benched shapes (64 series per GPU) the per-layer chain is 20 latency-bound launches whose activation rows make
// a round trip through L2 between any two of them, and 20 of k_tr_ffn_fwd's 42 us are prologue + epilogue
// (profiles/r05_train_ffn_fwd_ablations.txt).  Here a CLUSTER of NQ workgroups owns a series for all layers: workgroup
// (series b, part q) owns the token tiles q NT .. q NT + NT - 1 (NT = 4 or 2 tiles of 16 tokens; 8 waves = NT tiles x 8 / NT
// parts of the hidden dimension), the residual stream of its tiles stays in registers from layer to layer, and the attention
// output reaches the out-projection through LDS.  What a layer needs from the other workgroups of its cluster is the layer
// input of the WHOLE series (K and V of every token): every workgroup publishes the bf16 rows of its tiles -- the rows the
// backward needs anyway (x0rb) -- raises one flag per token tile, waits for the flags of the series and projects K / V^T of
// all tokens and all head pairs itself (36 MFMAs per token tile: cheaper than exchanging K and V^T, which are twice the
// bytes of the rows).  grid = NQ x series <= CUs and 1 workgroup per CU (LDS), so every workgroup of a cluster is resident;
// the waits are bounded like the F-split's (struct FSplit in fd_train_dev.h): a partner that never arrives is reported through
// the context's error word instead of hanging the device.
//
// Exchange without cache-wide fences (same mechanism and the same caveat as the F-split hand-over: it rests on the gfx950
// memory pipeline, not on the HIP memory model): the published rows are agent-scope relaxed atomic stores (write-through),
// the flag is stored after `s_waitcnt vmcnt(0)`, the consumers read flags and rows with agent-scope relaxed atomic loads
// (which bypass the reader XCD's non-coherent L2 lines).
//
// Everything the backward reads is written exactly as the per-layer kernels write it (att, attT, lse2, s1, s2, the x1 rows of
// the stage records, x0rb / x0T of the next layer, the activity bytes), so k_tr_ffn_bwd / k_tr_attn_bwd / k_tr_wgrad run
// unchanged.  One buffer is gone: the activity WORDS of the weight-gradient kernel (a 16-token ballot per hidden unit, 16-token
// aligned in the flat token index) cannot be written by a workgroup whose tiles are aligned to its series; k_tr_wgrad now picks
// the bits out of the activity bytes itself, which also takes the ballots (4.8 us per layer,
// profiles/r05_train_ffn_fwd_ablations.txt) out of every forward chunk loop.
// With FDIFF_TR_ROT=0 (natural chunk order in both forms) the saved activations and therefore the gradients are bit-identical to
// the per-layer kernels' (tests/test_gpu_train_persist.py).
#include <hip/hip_ext.h>

#include "fd_train_dev.h"

namespace {

__device__ __forceinline__ u32x4 ld_coh16(const void* p) {      // 16 bytes, two agent-scope (L2-bypassing) 8-byte loads
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
}
__device__ __forceinline__ void st_coh8(void* p, u32x2 v) {     // 8 bytes, agent-scope write-through store
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v[0] | ((unsigned long long)v[1] << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef FD_TRP_ABL
#define FD_TRP_ABL 0        // timing ablations of the chunk loop (wrong results): 1 no weight DMA inside the loop, 2 no per-step barrier, 4 no relu /
#endif                      // dropout / activity block, 8 no fragment prefetch reads (the same registers every step)
#ifndef FD_TRP_UNROLL1
#define FD_TRP_UNROLL1 1    // unroll of the units' key loops (pass 1: row maxima; pass 2: exp + P V)
#endif
#ifndef FD_TRP_UNROLL2
#define FD_TRP_UNROLL2 1
#endif
#ifndef FD_TRP_NUMAX
#define FD_TRP_NUMAX 1      // attention units a wave interleaves at four tiles per workgroup (3 = all of its head pairs at d_model 72: spills)
#endif
#ifdef FD_TRP_PROF          // variant build: phase clocks of workgroup 0 (waves 0 and NT), summed over the layers of a launch
__device__ unsigned long long fd_trp_dbg[2 * 24];
#define TRP_STAMP(slot, t_prev)                                                                          \
    do {                                                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter();                                    \
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && tl == 0 && fq < 2) fd_trp_dbg[fq * 24 + (slot)] += now_ - (t_prev); \
        (t_prev) = now_;                                                                                 \
    } while (0)
// sub-phase mark: everything issued so far has landed (perturbs the kernel: only in this variant), time since the last mark of any kind
#define TRP_SUB(slot, t_prev)                                                                            \
    do {                                                                                                 \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
        TRP_STAMP(slot, t_prev);                                                                         \
    } while (0)
#else
#define TRP_STAMP(slot, t_prev) do { } while (0)
#define TRP_SUB(slot, t_prev) do { } while (0)
#endif

// grid (NQ, series of this launch), 512 threads.  Wave = (token tile tl = wave % NT, hidden part fq = wave / NT).
template <int KS1, int DT, int KSO, int NT>
__global__ __launch_bounds__(512, 2) void k_tr_fwd_layers(const TrDims d, const fd_trp_args a) {
    constexpr int NFQ = 8 / NT, CPS = NFQ / 2;               // hidden parts per tile; 64-unit image chunks per ring step
    constexpr int NB = 2 * KS1 + DT, WB = 2 * NB * 1024, SB = CPS * WB;
    constexpr int NBUF = (NT == 4) ? 4 : 3, PD = NBUF - 1;   // ring of NBUF steps, PD steps in flight
    constexpr int NBLK = SB / 1024, NDMA = (NBLK + 7) / 8;
    static_assert(NT == 4 || NT == 2, "tiles per workgroup");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane0 = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int q, bl;
    xcd_deal(d, q, bl);                                      // (the workgroups of a series share an L2 where the dispatcher allows)
    const int b = a.b0 + bl;
    const int T = d.T, KT = d.KT, NJ = d.NJ, NTOK = KT * 16, hd = d.hd, H = d.H, D = d.D, F = d.F, NP = d.NP;
    const int NS = F / 64, NSTEP = NS / CPS, NS2 = 2 * NS;
    unsigned long long tprev = __builtin_readcyclecounter();
    (void)tprev;
    // ---- LDS
    const size_t kv_bytes = (size_t)NP * ((size_t)NTOK * 32 + (size_t)NJ * 1024);
    const size_t szA = (((size_t)NBUF * SB > kv_bytes ? (size_t)NBUF * SB : kv_bytes) + 1023) & ~size_t(1023);
    char* const ring = smem;                                 // region A: K | V^T of every head pair, then the weight ring, then xch / tscr
    char* const kbf0 = smem;                                 // [pair][NTOK][4][8 B]
    char* const vbf0 = smem + (size_t)NP * NTOK * 32;        // [pair][NJ][4][16][16 B]
    char* const attf = smem + szA;                           // [NT][KSO] KiB: attention output of the own tiles as out-projection B fragments
    char* const xfr = attf + NT * KSO * 1024;                // [NT][KS1] KiB: LayerNorm1 output of the own tiles as B fragments
    unsigned char* const actB0 = reinterpret_cast<unsigned char*>(xfr + NT * KS1 * 1024);      // [NT][2][64][NS]
    unsigned* const klut = reinterpret_cast<unsigned*>(xfr + NT * KS1 * 1024 + (size_t)NT * 2 * 64 * NS);
    unsigned* const misc = klut + 32;                                // [0]: every workgroup of the cluster sits on this XCD (per layer, wave 0)
    float4* const lvec = reinterpret_cast<float4*>(klut + 64);      // [6 vectors][4 DT]: bo, gamma1, beta1, b2, gamma2, beta2
    // W_o fragment image of the layer (DT x KSO KiB, copied by global_load_lds at the start of a layer, read by the owners behind
    // barrier (3)): in the part of region A that neither K | V^T nor the first PD ring steps touch when there is one, else behind
    // the vectors (fd_trp_lds_bytes mirrors this)
    constexpr int WOB = DT * KSO * 1024;
    const size_t wo_in_a = (kv_bytes > (size_t)PD * SB ? kv_bytes : (size_t)PD * SB);
    char* const wol = (wo_in_a + WOB <= szA) ? smem + wo_in_a : reinterpret_cast<char*>(lvec + 6 * 4 * DT);
    f32x4* const xch = reinterpret_cast<f32x4*>(smem);       // (after the chunk loop) [NFQ - 1][NT][DT][64]
    if (threadIdx.x < 32) {
        const unsigned n = threadIdx.x >> 1, hi = threadIdx.x & 1;
        klut[threadIdx.x] = ((n >> (2 * hi)) & 1u ? 0x0000ffffu : 0u) | ((n >> (2 * hi + 1)) & 1u ? 0xffff0000u : 0u);
    }
    for (int i = threadIdx.x; i < NT * KSO * 64; i += 512) reinterpret_cast<u32x4*>(attf)[i] = u32x4{0u, 0u, 0u, 0u};
    const size_t pstride = (size_t)KS1 * 1024;
    const f32x4 allneg = {kNegBig, kNegBig, kNegBig, kNegBig};
    // The XCD this workgroup runs on (HW_REG_XCC_ID, bits 3:0).  The workgroups of a cluster publish it in the low bits of their tile
    // flags; when ALL of them share an XCD -- the dispatcher deals consecutive workgroup ids round-robin to the XCDs and xcd_deal puts a
    // series' workgroups on one, but nothing guarantees it -- their L2 is the coherence point: rows are plain stores (acknowledged by the
    // L2 = `s_waitcnt vmcnt(0)`, the write-through L1 keeps nothing back) and L1-bypassing (sc0) loads that HIT in the L2.  Otherwise,
    // and for a launch's first layer, the memory-side path: agent-scope (sc1) stores and loads.  The memory-side path moved the whole
    // series' rows of every layer through the fabric for every workgroup: 9.5 K cycles per layer until the rows had landed, 4.9 K until
    // the published rows were acknowledged (profiles/r06_train_fwd_layers_phase_clocks.txt).
    const unsigned my_xcc = (unsigned)__builtin_amdgcn_s_getreg(6164) & 15u;      // hwreg(HW_REG_XCC_ID = 20, 0, 4)
    bool same_xcd = false;                                // (known from the first wait of the launch on)
    const int rot = d.norot ? 0 : (int)(((unsigned)(b * gridDim.x + q) * 5u) % (unsigned)NSTEP);

    f32x4 v[DT];                                             // owners: the residual stream of the own tile
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) v[dt] = f4zero();
    for (int l = a.l0; l < a.l1; ++l) {
        // Everything that depends on the lane / wave index is re-derived per layer from an OPAQUE copy: hipcc otherwise hoists a
        // layer's ~250 address computations out of this loop and keeps them in scratch (888 bytes per lane, 455 KB per workgroup,
        // more than an XCD's L2 holds for its 32 workgroups: every reload was a trip to memory -- 190 us per layer)
        int lane = lane0, wave = wave0;
        asm volatile("" : "+v"(lane));
        asm volatile("" : "+s"(wave));
        const int tok = lane & 15, g = lane >> 4;
        const int tl = wave % NT, fq = wave / NT, sub = fq >> 1, fhw = fq & 1;
        const bool owner = fq == 0;
        const int kt_own = q * NT + tl;
        const bool tile_ok = kt_own < KT;
        const int t_own = kt_own * 16 + tok;
        const bool valid = tile_ok && t_own < T;
        const int m = b * T + (valid ? t_own : 0);               // (invalid lanes: a row that exists; they never store)
        unsigned char* const actB = actB0 + (size_t)(tl * 2 + fhw) * 64 * NS;
        char* const tscr = smem + (size_t)(NFQ - 1) * NT * DT * 1024 + tl * (32 * DT * 16);
        const bool lo_grp = (g >> 1) == 0;
        f32x4 cmask;
#pragma unroll
        for (int r = 0; r < 4; ++r) cmask[r] = ((KT - 1) * 16 + 4 * g + r >= T) ? kNegBig : 0.f;
        const size_t lo = (size_t)l * a.lstride;
        const __bf16* const x0rb = reinterpret_cast<const __bf16*>(reinterpret_cast<const char*>(a.x0rb) + lo);
        float* const att = reinterpret_cast<float*>(reinterpret_cast<char*>(a.att) + lo);
        __bf16* const attT = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(a.attT) + lo);
        float* const lse2 = reinterpret_cast<float*>(reinterpret_cast<char*>(a.lse2) + lo);
        const unsigned char* const pmask = a.pmask + lo;
        const char* const limg = a.limg + (size_t)l * a.limg_stride;
        const float* const Pl = a.P + (long long)l * a.pstride;
        const bool first = l == a.l0, last_of_launch = l + 1 == a.l1, has_next = l + 1 < a.L;
        // ---- (1) the layer's inputs: the rows of every token tile of the series (published by the cluster) and the layer's dropout decisions
        if (wave == 0) {
            const unsigned long long xt = a.epoch * 64ull + (unsigned long long)l;      // value of a tile flag once layer l - 1 is published
            const bool wx = !first && lane < KT, wm = a.mflag != nullptr && lane == 16;
            const unsigned long long* fp = wm ? a.mflag + l : a.xflag + (size_t)b * KT + (lane < KT ? lane : 0);
            const unsigned long long tgt = wm ? a.epoch : xt;
            // (tile flags: value >> 4 = epoch * 64 + layers published, low bits = the publisher's XCD; the decision flags carry the epoch)
            unsigned long long fv = (wx || wm) ? __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            bool ok = !(wx || wm) || (long long)((wm ? fv : fv >> 4) - tgt) >= 0;
            if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {
                const unsigned long long t0 = wall_clock64();
                unsigned spins = 0u;
                for (;;) {
                    __builtin_amdgcn_s_sleep(8);
                    if (!ok) {
                        fv = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (long long)((wm ? fv : fv >> 4) - tgt) >= 0;
                    }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if ((++spins & 63u) == 0u && wall_clock64() - t0 > a.timeout) break;
                }
                if (!ok) {          // gave up: record (series, lane) once; the step's results are wrong and the next training call says so
                    unsigned expected = 0u;
                    __hip_atomic_compare_exchange_strong(a.err, &expected, 0x40000000u | ((unsigned)l << 20) | ((unsigned)(b & 0xffff) << 4) | (unsigned)(lane & 15) | (wm ? 0x80000u : 0u),
                                                         __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(a.err_gpu, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (!first) {
                const bool all_here = __builtin_amdgcn_ballot_w64(wx && ((unsigned)fv & 15u) != my_xcc) == 0ull;
                if (lane == 0) misc[0] = all_here ? 1u : 0u;
                // this CU's L1 may still hold lines of the rows from the previous training step (same addresses): drop them before any wave
                // reads the new ones (the loads below also carry sc0, which misses the L1 by itself)
                asm volatile("buffer_inv sc0" ::: "memory");
            }
        }
        __syncthreads();
        if (!first) same_xcd = misc[0] != 0u;
        TRP_STAMP(0, tprev);          // wait for the cluster + barrier
        if (first && owner) load_ctile<DT>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x0) + lo), m, valid, D, g, v);
        // the six small vectors of the layer, through LDS (written in front of barrier (3))
        float4 lval = {0.f, 0.f, 0.f, 0.f};
        if (threadIdx.x < 6 * 4 * DT) {
            const int vq = threadIdx.x / (4 * DT), cq = threadIdx.x - vq * (4 * DT);
            const long long off = vq == 0 ? a.o_bo : vq == 1 ? a.o_g1 : vq == 2 ? a.o_be1 : vq == 3 ? a.o_b2 : vq == 4 ? a.o_g2 : a.o_be2;
            if (4 * cq < D) lval = *reinterpret_cast<const float4*>(Pl + off + 4 * cq);
        }
        // a token tile's B fragments from the published rows (clamped row, cleared afterwards)
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x0rb), 0, 0x7ffffff0, 0x00020000);
        auto xload = [&](int tile, u32x4 (&r)[KS1]) {
            const int t = tile * 16 + tok, tc = t < T ? t : T - 1;
            const int voff = ((b * T + tc) * d.RBW + 8 * g) * 2;
            if (same_xcd) {
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) r[ks] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff + 64 * ks, 0, 1);       // sc0: L1 bypass, L2 hit
            } else {
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) r[ks] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff + 64 * ks, 0, 16);      // sc1: agent scope
            }
        };
        auto xfrag_of = [&](int tile, const u32x4 (&r)[KS1], int ks) -> bf16x8 {
            const unsigned keep = (tile * 16 + tok < T) ? ~0u : 0u;
            return __builtin_bit_cast(bf16x8, u32x4{r[ks][0] & keep, r[ks][1] & keep, r[ks][2] & keep, r[ks][3] & keep});
        };
        // ---- (2) K and V^T of every token of the series, every head pair: wave w projects the tiles w and w + 8
        u32x4 xq[KS1];
        unsigned bits1p = 0u, bits3p = 0u;       // owners: dropout bits of the two residual sites (requested here, used behind the units),
                                                 // one nibble per C tile (two registers live across the units instead of 2 DT)
        {
            {   // W_o image -> LDS (asynchronous: nothing waits for it before barrier (3))
                const char* wo = limg + a.off_wo + (size_t)lane * 16;
                for (int bb = wave; bb < DT * KSO; bb += 8)
                    __builtin_amdgcn_global_load_lds(GLB_PTR(wo + (size_t)bb * 1024), LDS_PTR(wol + bb * 1024), 16, 0, 0);
            }
            const int kt0 = wave < KT ? wave : KT - 1, kt1 = wave + 8 < KT ? wave + 8 : KT - 1;
            const bool two = wave + 8 < KT;
            u32x4 xa[KS1], xb[KS1];
            xload(kt0, xa);
            xload(kt1, xb);
            xload(tile_ok ? kt_own : KT - 1, xq);             // (the own tile's rows for the Q projections of phase (3))
            if (owner) {
                unsigned bt1[DT], bt3[DT];
                row_drop_bits<DT>(d, a.rb1 + lo, m, valid, g, bt1);
                row_drop_bits<DT>(d, a.rb3 + lo, m, valid, g, bt3);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) { bits1p |= (bt1[dt] & 15u) << (4 * dt); bits3p |= (bt3[dt] & 15u) << (4 * dt); }
            }
            const char* wkb = limg + a.off_wk + (size_t)lane * 16;
            const char* wvb = limg + a.off_wv + (size_t)lane * 16;
            // weight fragments of a pair: three register sets, two pairs ahead (an L2 round trip is ~2 pairs of this wave's MFMAs)
            bf16x8 wA[2 * KS1], wB[2 * KS1], wC[2 * KS1];
            auto wload = [&](int p, bf16x8 (&w)[2 * KS1]) {
                const int pc = p < NP ? p : NP - 1;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    w[ks] = *reinterpret_cast<const bf16x8*>(wkb + pc * pstride + (size_t)ks * 1024);
                    w[KS1 + ks] = *reinterpret_cast<const bf16x8*>(wvb + pc * pstride + (size_t)ks * 1024);
                }
            };
            // both tiles of the wave in ONE basic block (four independent MFMA chains); a wave without a second tile computes its
            // first one twice and stores once
            auto project = [&](int p, const bf16x8 (&w)[2 * KS1]) {
                char* const kbf = kbf0 + (size_t)p * NTOK * 32;
                char* const vbf = vbf0 + (size_t)p * NJ * 1024;
                f32x4 ka0 = f4zero(), vc0 = f4zero(), ka1 = f4zero(), vc1 = f4zero();
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    const bf16x8 x0f = xfrag_of(kt0, xa, ks), x1f = xfrag_of(kt1, xb, ks);
                    ka0 = MFMA(w[ks], x0f, ka0);
                    vc0 = MFMA(x0f, w[KS1 + ks], vc0);
                    ka1 = MFMA(w[ks], x1f, ka1);
                    vc1 = MFMA(x1f, w[KS1 + ks], vc1);
                }
                if (wave < KT) {
                    const int kt = kt0;
                    *reinterpret_cast<u32x2*>(kbf + ((size_t)(kt * 16 + tok) * 4 + g) * 8) = u32x2{cvt_pk_bf16(ka0[0], ka0[1]), cvt_pk_bf16(ka0[2], ka0[3])};
                    char* dst = vbf + ((size_t)((kt >> 1) * 4 + g) * 16 + tok) * 16;
                    *reinterpret_cast<u32x2*>(dst + 8 * (kt & 1)) = u32x2{cvt_pk_bf16(vc0[0], vc0[1]), cvt_pk_bf16(vc0[2], vc0[3])};
                    if ((KT & 1) && kt == KT - 1) *reinterpret_cast<u32x2*>(dst + 8) = u32x2{0u, 0u};
                }
                if (two) {
                    const int kt = kt1;
                    *reinterpret_cast<u32x2*>(kbf + ((size_t)(kt * 16 + tok) * 4 + g) * 8) = u32x2{cvt_pk_bf16(ka1[0], ka1[1]), cvt_pk_bf16(ka1[2], ka1[3])};
                    char* dst = vbf + ((size_t)((kt >> 1) * 4 + g) * 16 + tok) * 16;
                    *reinterpret_cast<u32x2*>(dst + 8 * (kt & 1)) = u32x2{cvt_pk_bf16(vc1[0], vc1[1]), cvt_pk_bf16(vc1[2], vc1[3])};
                    if ((KT & 1) && kt == KT - 1) *reinterpret_cast<u32x2*>(dst + 8) = u32x2{0u, 0u};
                }
            };
            wload(0, wA);
            wload(1, wB);
            TRP_SUB(10, tprev);       // [variant] x rows, dropout bits and the first two pairs' weight fragments landed
            for (int p = 0; p < NP; p += 3) {
                wload(p + 2, wC);
                project(p, wA);
                if (p + 1 < NP) {
                    wload(p + 3, wA);
                    project(p + 1, wB);
                }
                if (p + 2 < NP) {
                    wload(p + 4, wB);
                    project(p + 2, wC);
                }
            }
        }
        if (threadIdx.x < 6 * 4 * DT) lvec[threadIdx.x] = lval;
        TRP_STAMP(1, tprev);          // K / V^T of the series
        __syncthreads();
        TRP_STAMP(2, tprev);
        // ---- (3) attention of the own tile: wave (tl, fq) runs the head pairs fq, fq + NFQ, ...  (k_tr_attn_fwd's unit, same arithmetic
        // per (query tile, head pair)).  The wave's units run INTERLEAVED -- one key loop for all of them: a unit is a chain of
        // LDS read -> score MFMAs -> exponentials -> pack -> P V MFMA per key block, and with two waves per SIMD nothing else hides
        // its latencies (one unit after the other: 15 K cycles per unit, 45 K of a layer's 134 K; profiles/r06_train_fwd_layers_phase_clocks.txt)
        auto units = [&](auto nu_c, int pair0) {
            constexpr int NU = decltype(nu_c)::value;
            const int qt = kt_own, t = t_own;
            const char* kbf[NU];
            const char* vbf[NU];
            s16x4 qb[NU][2];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int pair = pair0 + u * NFQ;
                kbf[u] = kbf0 + (size_t)pair * NTOK * 32 + ((size_t)tok * 4 + g) * 8;
                vbf[u] = vbf0 + (size_t)pair * NJ * 1024 + ((size_t)g * 16 + tok) * 16;
                f32x4 qa = f4zero();
                const char* wqb = limg + a.off_wq + pair * pstride + (size_t)lane * 16;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) qa = MFMA(*reinterpret_cast<const bf16x8*>(wqb + (size_t)ks * 1024), xfrag_of(qt, xq, ks), qa);
                const unsigned q01 = cvt_pk_bf16(qa[0], qa[1]), q23 = cvt_pk_bf16(qa[2], qa[3]);
                qb[u][0] = __builtin_bit_cast(s16x4, u32x2{lo_grp ? q01 : 0u, lo_grp ? q23 : 0u});
                qb[u][1] = __builtin_bit_cast(s16x4, u32x2{lo_grp ? 0u : q01, lo_grp ? 0u : q23});
            }
            // pass 1: exact row maxima (base-2 logits: log2(e) / sqrt(hd) is folded into W_q)
            float mx[NU][2];
#pragma unroll
            for (int u = 0; u < NU; ++u) mx[u][0] = mx[u][1] = kNegBig;
            auto maxtile = [&](int kt, const f32x4 c0) {          // (the last key tile alone carries the padding mask: peeled, no select per tile)
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const s16x4 kf = *reinterpret_cast<const s16x4*>(kbf[u] + (size_t)kt * 512);
#pragma unroll
                    for (int hs = 0; hs < 2; ++hs) {
                        const f32x4 sv = MFMA16(kf, qb[u][hs], c0);
                        mx[u][hs] = fmaxf(fmaxf(fmaxf(mx[u][hs], sv[0]), sv[1]), fmaxf(sv[2], sv[3]));
                    }
                }
            };
#pragma unroll FD_TRP_UNROLL1
            for (int kt = 0; kt < KT - 1; ++kt) maxtile(kt, f4zero());
            maxtile(KT - 1, cmask);
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) mx[u][hs] = group_max(mx[u][hs]);
            // pass 2: P = exp2(S - max), row sums of the undropped P, dropped P (unscaled) times V; keep bytes one key block ahead
            float ls[NU][2];
            f32x4 o2[NU][2];
            unsigned prow[NU][2];          // (byte offsets into pmask: one register each instead of a pointer pair)
            unsigned bnext[NU][2];
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    const int head = 2 * (pair0 + u * NFQ) + hs, hc = head < H ? head : H - 1;
                    prow[u][hs] = (unsigned)(((((size_t)b * H + hc) * T + (t < T ? t : 0)) * NJ) * 4 + g);
                    ls[u][hs] = 0.f;
                    o2[u][hs] = f4zero();
                    bnext[u][hs] = pmask[prow[u][hs]];      // (unconditional: the buffer exists without dropout too; selected below)
                }
            const bool nodrop = !(d.p > 0.f);
            // one key block (32 keys) of every unit of the wave.  Only the LAST block carries a mask in the C operand of its score MFMAs
            // (padded keys, a missing odd tile); everywhere else C is the plain -max splat: adding the zero mask cost eight VALU adds
            // per (block, head), 14 % of the loop's VALU cycles (x + 0.0f = x bit for bit, so the results do not change)
            auto keyblock = [&](int jb, auto last_c) {
                constexpr bool LAST = decltype(last_c)::value;
                const int ka = 2 * jb, kb = (2 * jb + 1 < KT) ? 2 * jb + 1 : ka;
                const f32x4 ma = (ka == KT - 1) ? cmask : f4zero();
                const f32x4 mb = (2 * jb + 1 >= KT) ? allneg : ((kb == KT - 1) ? cmask : f4zero());
                const int jn = jb + 1 < NJ ? jb + 1 : jb;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    // (K / V^T fragments requested one block ahead instead -- eight more registers -- measured slower: 42.0 -> 42.9 K cycles per
                    // layer for the units, 2.136 -> 2.153 ms per step; the units are bound by VALU issue, not by the LDS round trip)
                    const s16x4 kfa = *reinterpret_cast<const s16x4*>(kbf[u] + (size_t)ka * 512), kfb = *reinterpret_cast<const s16x4*>(kbf[u] + (size_t)kb * 512);
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vbf[u] + (size_t)jb * 1024);
#pragma unroll
                    for (int hs = 0; hs < 2; ++hs) {
                        // (no branch inside the key loop: a uniform `if (p > 0)` here split the loop body into one basic block per (unit,
                        // head), and the units' chains were not interleaved at all)
                        const unsigned bits = nodrop ? 0xffu : bnext[u][hs];
                        bnext[u][hs] = pmask[prow[u][hs] + (unsigned)jn * 4u];
                        const float nm = -mx[u][hs];
                        const f32x4 nmv = {nm, nm, nm, nm};
                        f32x4 pa = MFMA16(kfa, qb[u][hs], LAST ? (ma + nmv) : nmv);
                        f32x4 pb = MFMA16(kfb, qb[u][hs], LAST ? (mb + nmv) : nmv);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pa[r] = __builtin_amdgcn_exp2f(pa[r]);
                            pb[r] = __builtin_amdgcn_exp2f(pb[r]);
                        }
                        ls[u][hs] += (pa[0] + pa[1]) + (pa[2] + pa[3]) + (pb[0] + pb[1]) + (pb[2] + pb[3]);
                        const u32x2 ka2 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits & 15u)), kb2 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits >> 4));
                        const u32x4 pk = __builtin_bit_cast(u32x4, pack8(pa, pb));
                        o2[u][hs] = MFMA(vf, __builtin_bit_cast(bf16x8, u32x4{pk[0] & ka2[0], pk[1] & ka2[1], pk[2] & kb2[0], pk[3] & kb2[1]}), o2[u][hs]);
                    }
                }
            };
#pragma unroll FD_TRP_UNROLL2
            for (int jb = 0; jb < NJ - 1; ++jb) keyblock(jb, std::false_type{});
            keyblock(NJ - 1, std::true_type{});
            const int mq = b * T + (t < T ? t : 0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int pair = pair0 + u * NFQ;
                const int myhead = 2 * pair + (g >> 1);
                const float l0s = group_sum(ls[u][0]), l1s = group_sum(ls[u][1]);
                const float lmine = lo_grp ? l0s : l1s, mmine = lo_grp ? mx[u][0] : mx[u][1];
                const float inv = d.keep_scale * __builtin_amdgcn_rcpf(lmine);
                float ov[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (lo_grp ? o2[u][0][r] : o2[u][1][r]) * inv;
                const bool okh = t < T && myhead < H;
                if (okh) {
                    float* orow = att + (size_t)mq * D + myhead * hd + 4 * (g & 1);
                    const int nv = min(4, max(0, hd - 4 * (g & 1)));
                    if (nv == 4) *reinterpret_cast<f32x4_a4*>(orow) = f32x4_a4{ov[0], ov[1], ov[2], ov[3]};
                    else if (nv == 2) *reinterpret_cast<f32x2_a4*>(orow) = f32x2_a4{ov[0], ov[1]};
                    else
                        for (int r = 0; r < nv; ++r) orow[r] = ov[r];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int dd = 4 * (g & 1) + r;
                        if (dd < hd) attT[((size_t)(mq >> 5) * d.NFT + myhead * hd + dd) * 32 + (mq & 31)] = (__bf16)ov[r];
                    }
                    if ((g & 1) == 0) lse2[((size_t)b * H + myhead) * T + t] = mmine + __builtin_amdgcn_logf(lmine);
                }
                if (pair == 0 && t < T) {          // ones row (bias column of d W_o) and the zero rows of the T-block padding
                    for (int f = D + g; f < d.NFT; f += 4)
                        attT[((size_t)(mq >> 5) * d.NFT + f) * 32 + (mq & 31)] = (__bf16)((f == D) ? 1.0f : 0.f);
                }
                // the out-projection's B fragment: k-step head >> 2, lane group head & 3, the head's 8 dim slots (k_tr_ffn_fwd builds
                // the same bf16 values from the fp32 rows of `att`)
                u32x2 pk = {0u, 0u};
                if (okh) {
                    float e4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) e4[r] = (4 * (g & 1) + r < hd) ? ov[r] : 0.f;
                    pk = u32x2{cvt_pk_bf16(e4[0], e4[1]), cvt_pk_bf16(e4[2], e4[3])};
                }
                *reinterpret_cast<u32x2*>(attf + ((size_t)((tl * KSO + (myhead >> 2)) * 64 + (myhead & 3) * 16 + tok)) * 16 + 8 * (g & 1)) = pk;
            }
        };
        if (tile_ok) {
            constexpr int NUMAX = FD_TRP_NUMAX;
            int pair = fq;
            if constexpr (NUMAX >= 3) {
                for (; pair + 2 * NFQ < NP; pair += 3 * NFQ) units(std::integral_constant<int, 3>{}, pair);
            }
            if constexpr (NUMAX >= 2) {
                for (; pair + NFQ < NP; pair += 2 * NFQ) units(std::integral_constant<int, 2>{}, pair);
            }
            for (; pair < NP; pair += NFQ) units(std::integral_constant<int, 1>{}, pair);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (this wave's share of the W_o image has landed)
        TRP_STAMP(3, tprev);          // attention units
        __syncthreads();
        TRP_STAMP(4, tprev);
        // ---- (4) FFN weight ring (region A is free now), dropout bytes of the hidden units, out-projection + LN1 on the owner waves
        const char* const ffn_img = limg + a.off_ffn;
        auto issue = [&](int st) {
            int cs = st + rot;
            cs -= (cs >= NSTEP) ? NSTEP : 0;
            const char* src = ffn_img + (size_t)cs * SB + lane * 16;
            char* dst = ring + (st % NBUF) * SB;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                int bb = wave + i * 8;
                bb %= NBLK;                   // padding copies repeat a block (uniform vmcnt bookkeeping)
                __builtin_amdgcn_global_load_lds(GLB_PTR(src + bb * 1024), LDS_PTR(dst + bb * 1024), 16, 0, 0);
            }
        };
#pragma unroll
        for (int st = 0; st < PD; ++st)
            if (st < NSTEP) issue(st);
        {   // the CPS waves of a (tile, half) share one [64][NS] byte image; each stages its share of the columns
            const unsigned char* hk = a.hkeep + lo;
            unsigned char* dstl = actB + lane * NS;
            const int c0 = sub * (NS / CPS), c1 = c0 + NS / CPS;
            if (d.p > 0.f && valid) {
                const unsigned char* srcb = hk + ((size_t)m * 4 + g) * NS2 + fhw * NS;
                for (int c = c0; c < c1; c += 16) *reinterpret_cast<u32x4*>(dstl + c) = *reinterpret_cast<const u32x4*>(srcb + c);
            } else {
                const unsigned fill = valid ? ~0u : 0u;
                for (int c = c0; c < c1; c += 16) *reinterpret_cast<u32x4*>(dstl + c) = u32x4{fill, fill, fill, fill};
            }
        }
        f32x4 s1keep[DT];
        bf16x8 xf[KS1];
        TRP_SUB(11, tprev);           // [variant] ring DMA issued and landed, dropout bytes staged
        if (owner) {
            f32x4 o[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = f4zero();
#pragma unroll
            for (int ks = 0; ks < KSO; ++ks) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(attf + ((size_t)(tl * KSO + ks) * 64 + lane) * 16);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[dt] = MFMA(*reinterpret_cast<const bf16x8*>(wol + ((size_t)(dt * KSO + ks) * 64 + lane) * 16), af, o[dt]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d0 = 16 * dt + 4 * g;
                if (d0 < D) {
                    const float4 bo4 = lvec[0 * 4 * DT + 4 * dt + g];
                    const float bv[4] = {bo4.x, bo4.y, bo4.z, bo4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[dt][r] += ((bits1p >> (4 * dt + r)) & 1u) ? (o[dt][r] + bv[r]) * d.keep_scale : 0.f;
                }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) s1keep[dt] = v[dt];
            TRP_SUB(12, tprev);       // [variant] out-projection + residual
            {
                float mean, rstd;
                ln_stats<DT>(v, D, g, mean, rstd);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int d0 = 16 * dt + 4 * g;
                    if (d0 < D) {
                        const float4 gm = lvec[1 * 4 * DT + 4 * dt + g], bt = lvec[2 * 4 * DT + 4 * dt + g];
                        v[dt][0] = (v[dt][0] - mean) * rstd * gm.x + bt.x;
                        v[dt][1] = (v[dt][1] - mean) * rstd * gm.y + bt.y;
                        v[dt][2] = (v[dt][2] - mean) * rstd * gm.z + bt.z;
                        v[dt][3] = (v[dt][3] - mean) * rstd * gm.w + bt.w;
                    } else {
                        v[dt] = f4zero();
                    }
                }
            }
            TRP_SUB(13, tprev);       // [variant] LayerNorm1
            ctile_to_frags<DT, KS1>(xfr + tl * KS1 * 1024, lane, D, v, true, xf);
            TRP_SUB(14, tprev);       // [variant] fragments through LDS
        }
        f32x4 acc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = f4zero();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRP_STAMP(5, tprev);          // out-projection + LN1 (owners), dropout bytes staged, first weight steps landed
        __syncthreads();
        TRP_STAMP(6, tprev);
        if (!owner) {
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xfr + ((size_t)(tl * KS1 + ks) * 64 + lane) * 16);
        }
        // ---- (5) FFN over this wave's part of the hidden dimension (k_tr_ffn_fwd's chunk loop; no ballots)
        unsigned bits_cur = 0u;
        u32x2 kq0 = {0u, 0u}, kq1 = {0u, 0u};
        auto frags = [&](int c, bf16x8 (&w1)[2 * KS1], bf16x8 (&w2)[DT]) {
            const int cc = c < NSTEP ? c : NSTEP - 1;
            const char* wb = ring + (cc % NBUF) * SB + (sub * 2 + fhw) * NB * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < 2 * KS1; ++i) w1[i] = *reinterpret_cast<const bf16x8*>(wb + i * 1024);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) w2[dt] = *reinterpret_cast<const bf16x8*>(wb + (2 * KS1 + dt) * 1024);
        };
        // A step's fragments are read one step AHEAD only with three steps in flight (PD = 3: the wait at the end of step c - 1 leaves
        // just the newest batch, c + 2, outstanding, so batch c + 1 is visible when step c prefetches it).  With PD = 2 (two tiles per
        // workgroup: a ring of three 44-KiB steps is all the LDS holds) the newest outstanding batch at that point IS c + 1: the step
        // reads its own fragments at its start instead (the first version prefetched there too and read a buffer still in flight --
        // caught by the run-to-run comparison of tests/test_gpu_benched_shapes.py at the full grid).
        constexpr bool PREF = PD >= 3;
        auto step = [&](int c, bf16x8 (&w1)[2 * KS1], bf16x8 (&w2)[DT], bf16x8 (&n1)[2 * KS1], bf16x8 (&n2)[DT]) {
            if (!(FD_TRP_ABL & 1) && c + PD < NSTEP) issue(c + PD);
            if constexpr (PREF) { if (!(FD_TRP_ABL & 8)) frags(c + 1, n1, n2); }
            else frags(c, w1, w2);
            f32x4 h0 = f4zero(), h1 = f4zero();
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                h0 = MFMA(w1[ks], xf[ks], h0);
                h1 = MFMA(w1[KS1 + ks], xf[ks], h1);
            }
            int cs = c + rot;
            cs -= (cs >= NSTEP) ? NSTEP : 0;
            int cn = cs + 1;                                           // next step's chunk (clamped read after the last step)
            cn -= (cn >= NSTEP) ? NSTEP : 0;
            const int ce = cs * CPS + sub;
            // The keep masks of this step's eight hidden values (lane masks of packed bf16 pairs, by nibble of the keep byte) were read
            // from the table one step AHEAD, like the byte itself: between the H MFMAs and the W2 MFMAs that need the masked values
            // sits no LDS round trip any more (timing ablations of this loop, profiles/r06_train_fwd_layers_loop_ablations.txt: without
            // the relu / dropout / activity block 39.7 -> 25.1 K cycles per layer, without the barrier 27.6, without the weight DMA
            // 34.1, without the fragment reads 33.0, without all four 7.8)
            // (two steps deep: the byte of step c + 2 is requested while the table entries of step c + 1 are looked up with the byte
            // that arrived a step ago -- one LDS hop per step on either chain, none in front of this step's end-of-step wait)
            const u32x2 k0 = kq0, k1 = kq1;
            kq0 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits_cur & 15u));
            kq1 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits_cur >> 4));
            {
                int c2 = cn + 1;
                c2 -= (c2 >= NSTEP) ? NSTEP : 0;
                bits_cur = actB[lane * NS + c2 * CPS + sub];
            }
            u32x4 pk;
#if FD_TRP_ABL & 4
            pk = __builtin_bit_cast(u32x4, pack8(h0, h1));
            (void)k0; (void)k1; (void)ce;
#else
            {
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                pk = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, pack8(h0, h1)), z8));
                pk = u32x4{pk[0] & k0[0], pk[1] & k0[1], pk[2] & k1[0], pk[3] & k1[1]};
            }
            {
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 one8 = {1, 1, 1, 1, 1, 1, 1, 1};
                const u32x4 mm = __builtin_bit_cast(u32x4, __builtin_elementwise_min(__builtin_bit_cast(s16x8, pk), one8));
                const unsigned tt = mm[0] | (mm[1] << 2) | (mm[2] << 4) | (mm[3] << 6);
                actB[lane * NS + ce] = (unsigned char)((tt & 0x55u) | ((tt >> 15) & 0xAAu));
            }
#endif
            const bf16x8 hb = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = MFMA(w2[dt], hb, acc[dt]);
            if (!(FD_TRP_ABL & 1) && c + PD < NSTEP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the prefetch reads and this step's LDS writes are done
            if (!(FD_TRP_ABL & 2)) __builtin_amdgcn_s_barrier();
        };
        {
            bf16x8 wa1[2 * KS1], wa2[DT], wb1[2 * KS1], wb2[DT];
            if constexpr (PREF) frags(0, wa1, wa2);
            bits_cur = actB[lane * NS + rot * CPS + sub];                 // step 0's byte -> its table entries; then step 1's byte
            kq0 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits_cur & 15u));
            kq1 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits_cur >> 4));
            {
                int c1 = rot + 1;
                c1 -= (c1 >= NSTEP) ? NSTEP : 0;
                bits_cur = actB[lane * NS + c1 * CPS + sub];
            }
            for (int c = 0; c < NSTEP; c += 2) {         // (NSTEP is even: F % 1024 == 0)
                step(c, wa1, wa2, wb1, wb2);
                step(c + 1, wb1, wb2, wa1, wa2);
            }
        }
        TRP_STAMP(7, tprev);          // chunk loop
        if (owner) {
            store_ctile<DT>(reinterpret_cast<float*>(reinterpret_cast<char*>(a.s1) + lo), m, valid, D, g, s1keep);
            if (valid) stage_rows<DT, KS1>(a.stage + lo, StageL<KS1, DT>::off_xr, m, true, D, g, v, true);
        }
        // ---- (6) combine the hidden parts; activity bytes out; the owner finishes the tile
        __syncthreads();
        if (!owner) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) xch[(((fq - 1) * NT + tl) * DT + dt) * 64 + lane] = acc[dt];
        }
        if (valid) {
            unsigned char* dstb = a.active + lo + ((size_t)m * 4 + g) * NS2 + fhw * NS;
            const unsigned char* srcl = actB + lane * NS;
            const int c0 = sub * (NS / CPS), c1 = c0 + NS / CPS;
            for (int c = c0; c < c1; c += 16) *reinterpret_cast<u32x4*>(dstb + c) = *reinterpret_cast<const u32x4*>(srcl + c);
        }
        // (the rows of the stage records beyond the batch's last token: the weight-gradient kernel multiplies whole 32-token blocks)
        if (b == d.B - 1 && q == (int)gridDim.x - 1) {
            using SL = StageL<KS1, DT>;
            const int nrow = a.Mpad - d.M, per = 32 * KS1 / 4;
            for (int i = threadIdx.x; i < nrow * per; i += 512) {
                const int mm = d.M + i / per, c4 = i - (i / per) * per;
                *reinterpret_cast<u32x2*>(a.stage + lo + (size_t)(mm >> 5) * SL::bytes + SL::off_xr + ((size_t)(mm & 31) * SL::RBS + 4 * c4) * 2) = u32x2{0u, 0u};
            }
        }
        __syncthreads();
        TRP_STAMP(8, tprev);          // s1 / stage / activity bytes out, two barriers + exchange
        if (owner) {
#pragma unroll
            for (int pq = 0; pq < NFQ - 1; ++pq)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc[dt] += xch[((pq * NT + tl) * DT + dt) * 64 + lane];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] *= d.keep_scale;      // hidden-unit keep scale
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d0 = 16 * dt + 4 * g;
                if (d0 < D) {
                    const float4 bb = lvec[3 * 4 * DT + 4 * dt + g];
                    const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[dt][r] += ((bits3p >> (4 * dt + r)) & 1u) ? (acc[dt][r] + bv[r]) * d.keep_scale : 0.f;
                }
            }
            TRP_SUB(15, tprev);       // [variant] partial sums added, bias + dropout + residual
            store_ctile<DT>(reinterpret_cast<float*>(reinterpret_cast<char*>(a.s2) + lo), m, valid, D, g, v);
            {
                float mean, rstd;
                ln_stats<DT>(v, D, g, mean, rstd);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int d0 = 16 * dt + 4 * g;
                    if (d0 < D) {
                        const float4 gm = lvec[4 * 4 * DT + 4 * dt + g], bt = lvec[5 * 4 * DT + 4 * dt + g];
                        v[dt][0] = (v[dt][0] - mean) * rstd * gm.x + bt.x;
                        v[dt][1] = (v[dt][1] - mean) * rstd * gm.y + bt.y;
                        v[dt][2] = (v[dt][2] - mean) * rstd * gm.z + bt.z;
                        v[dt][3] = (v[dt][3] - mean) * rstd * gm.w + bt.w;
                    } else {
                        v[dt] = f4zero();
                    }
                }
            }
            TRP_SUB(16, tprev);       // [variant] s2 stored (acknowledged), LayerNorm2
            // publish the next layer's input rows of this tile, then the tile's flag (write-through stores, acknowledged before the flag)
            if (has_next) {
                if (valid) {
                    char* row = reinterpret_cast<char*>(const_cast<__bf16*>(x0rb)) + a.lstride + (size_t)m * d.RBW * 2;
#pragma unroll
                    for (int dt = 0; dt < 2 * KS1; ++dt) {
                        const int d0 = 16 * dt + 4 * g;
                        u32x2 pk = {0u, 0u};
                        if (dt < DT && d0 < D) {
                            pk[0] = cvt_pk_bf16(v[dt < DT ? dt : 0][0], v[dt < DT ? dt : 0][1]);
                            pk[1] = cvt_pk_bf16(v[dt < DT ? dt : 0][2], v[dt < DT ? dt : 0][3]);
                        } else if (d0 == D) {
                            pk[0] = 0x00003F80u;
                        }
                        if (same_xcd) *reinterpret_cast<u32x2*>(row + d0 * 2) = pk;
                        else st_coh8(row + d0 * 2, pk);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0 && tile_ok && !a.stall)
                    __hip_atomic_store(a.xflag + (size_t)b * KT + kt_own, ((a.epoch * 64ull + (unsigned long long)(l + 1)) << 4) | my_xcc, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            TRP_SUB(17, tprev);       // [variant] rows published (acknowledged) + flag
            if (last_of_launch)
                store_ctile<DT>(has_next ? reinterpret_cast<float*>(reinterpret_cast<char*>(const_cast<float*>(a.x0)) + lo + a.lstride) : a.hL, m, valid, D, g, v);
            if (has_next && tile_ok) {
                // T-block of the next layer's input (weight gradient of in_proj): [32-token block][feature row][32 tokens] in the FLAT
                // token index, while this tile starts at b T + 16 kt: 4-token runs through the wave's LDS transpose when T % 4 == 0
                // (a run never crosses a series or a block), single elements otherwise
                __bf16* const xT = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(a.x0T) + lo + a.lstride);
                const int m0 = b * T + kt_own * 16;
                if ((T & 3) == 0) {
                    __bf16* lt = reinterpret_cast<__bf16*>(tscr);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = 16 * dt + 4 * g + r;
                            lt[f * 16 + tok] = (__bf16)((f < D) ? v[dt][r] : (f == D ? 1.0f : 0.f));
                        }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < (16 * DT * 4 + 63) / 64; ++i) {
                        const int idx = lane + 64 * i;
                        if (idx < 16 * DT * 4) {
                            const int f = idx >> 2, qr = idx & 3, mr = m0 + 4 * qr;
                            if (kt_own * 16 + 4 * qr < T)
                                *reinterpret_cast<u32x2*>(xT + ((size_t)(mr >> 5) * (16 * DT) + f) * 32 + (mr & 31)) = *reinterpret_cast<const u32x2*>(lt + f * 16 + 4 * qr);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                } else if (valid) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = 16 * dt + 4 * g + r;
                            xT[((size_t)(m >> 5) * (16 * DT) + f) * 32 + (m & 31)] = (__bf16)((f < D) ? v[dt][r] : (f == D ? 1.0f : 0.f));
                        }
                }
            }
        }
        TRP_STAMP(9, tprev);          // owner epilogue (LN2, publish, stores)
    }
}

__global__ void k_tr_set_flag(unsigned long long* flag, unsigned long long value) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KS1, int DT, int KSO, int NT>
int trp_launch(fd_ctx* ctx, const TrDims& d, const fd_trp_args& a, int nq, int nseries, size_t lds, hipStream_t s, hipEvent_t stop) {
    static unsigned long long attr = 0;
    if (fd_first_on_device(attr, ctx->device))
        FD_HIP(ctx, hipFuncSetAttribute((const void*)k_tr_fwd_layers<KS1, DT, KSO, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipExtLaunchKernelGGL((k_tr_fwd_layers<KS1, DT, KSO, NT>), dim3(nq, nseries), dim3(512), lds, s, nullptr, stop, 0, d, a);
    return FD_OK;
}

}  // namespace

size_t fd_trp_lds_bytes(int ks1, int dt, int kso, int NT, int T, int F, int NP) {
    const size_t NFQ = 8 / NT, CPS = NFQ / 2, NB = 2 * (size_t)ks1 + dt, SB = CPS * 2 * NB * 1024, NBUF = NT == 4 ? 4 : 3;
    const size_t KT = ((size_t)T + 15) / 16, NJ = (KT + 1) / 2, NS = (size_t)F / 64;
    const size_t kv = (size_t)NP * (KT * 16 * 32 + NJ * 1024);
    const size_t szA = (std::max(NBUF * SB, kv) + 1023) & ~size_t(1023);
    const size_t PD = NBUF - 1, wob = (size_t)dt * kso * 1024, wo_in_a = std::max(kv, PD * SB);
    const size_t tail = (wo_in_a + wob <= szA) ? 0 : wob;      // (the W_o image: inside region A when it fits there, see the kernel)
    return szA + (size_t)NT * kso * 1024 + (size_t)NT * ks1 * 1024 + (size_t)NT * 2 * 64 * NS + 256 + (size_t)6 * 4 * dt * 16 + tail;
}

// Tiles per workgroup for a batch of B series: 2 when the doubled grid still fits the chip (twice the workgroups, half the
// chunk steps each), else 4; 0 = the persistent forward does not apply (series too long for one LDS image of K | V^T, more
// workgroups than CUs even at 4 tiles -- the caller then launches series ranges --, an F that does not divide into the steps).
int fd_trp_tiles(const fd_score* m, int B, int* nq_out, int* series_per_launch) {
    const fd_bf16_images* im = m->bf16;
    const int T = m->d.max_len, F = m->d.dim_ff, KT = (T + 15) / 16, cu = m->ctx->num_cu;
    if (KT > 16 || F % 1024 != 0 || m->d.num_layers > 60) return 0;
    static const char* kNames[] = {"FDIFF_TR_PERSIST_NT"};
    const char* e = getenv(kNames[0]);
    const int forced = e ? atoi(e) : 0;
    for (int NT : {2, 4}) {
        if (forced && forced != NT) continue;
        const int nq = (KT + NT - 1) / NT;
        if (fd_trp_lds_bytes(im->ks1, im->dt, im->kso, NT, T, F, im->np) > 160 * 1024) continue;
        if (NT == 2 && !forced && (long long)nq * B > cu) continue;       // (4 tiles when the doubled grid does not fit)
        const int spl = std::max(1, std::min(B, cu / nq));
        *nq_out = nq;
        *series_per_launch = spl;
        return NT;
    }
    return 0;
}

int fd_trp_forward(fd_score* m, const TrDims& d, fd_trp_args a, int NT, int nq, int series_per_launch, hipStream_t s, hipEvent_t done) {
    fd_ctx* ctx = m->ctx;
    const fd_bf16_images* im = m->bf16;
    const size_t lds = fd_trp_lds_bytes(im->ks1, im->dt, im->kso, NT, d.T, d.F, d.NP);
    for (int b0 = 0; b0 < d.B; b0 += series_per_launch) {
        a.b0 = b0;
        const int ns = std::min(series_per_launch, d.B - b0);
        hipEvent_t stop = (b0 + series_per_launch >= d.B) ? done : nullptr;      // (bound to the LAST launch: no packet of its own on `s`)
        int rc = FD_ERR_UNSUPPORTED;
#define TRP_CASE(K, T_, O)                                                                                     \
    if (im->ks1 == K && im->dt == T_ && im->kso == O)                                                          \
        rc = NT == 4 ? trp_launch<K, T_, O, 4>(ctx, d, a, nq, ns, lds, s, stop) : trp_launch<K, T_, O, 2>(ctx, d, a, nq, ns, lds, s, stop);
        TRP_CASE(3, 5, 3)
        TRP_CASE(3, 5, 2)
        TRP_CASE(2, 3, 1)
        TRP_CASE(2, 4, 3)
        TRP_CASE(1, 2, 1)
        TRP_CASE(1, 1, 1)
#undef TRP_CASE
        if (rc == FD_ERR_UNSUPPORTED) return fd_fail(ctx, FD_ERR_UNSUPPORTED, "persistent training forward not instantiated for this model");
        if (rc) return rc;
    }
    FD_LAUNCH_CHECK(ctx);
#ifdef FD_TRP_PROF
    {
        static int calls = 0;
        if (++calls == 30) {
            unsigned long long h[48];
            hipStreamSynchronize(s);
            hipMemcpyFromSymbol(h, HIP_SYMBOL(fd_trp_dbg), sizeof(h));
            static const char* nm[10] = {"cluster wait", "K/V staging", "barrier", "attention units", "barrier", "out-proj + LN1 + DMA wait", "barrier", "chunk loop",
                                         "stores + exchange", "owner epilogue"};
            fprintf(stderr, "[k_tr_fwd_layers phase clocks, workgroup (0,0), K cycles per LAYER, average of %d launches x %d layers]\n", calls, a.l1 - a.l0);
            for (int w = 0; w < 2; ++w) {
                fprintf(stderr, "  wave %d:", w * NT);
                for (int qq = 0; qq < 10; ++qq) fprintf(stderr, " %s %.1f |", nm[qq], (double)h[w * 24 + qq] / calls / (a.l1 - a.l0) / 1000.0);
                fprintf(stderr, "\n    sub-marks (included above: each is the time since the previous mark of any kind):");
                static const char* sn[8] = {"x rows + bits + first weights landed", "ring DMA + dropout bytes landed", "out-proj + residual", "LN1", "fragments via LDS",
                                            "partials + bias + dropout + residual", "s2 acked + LN2", "rows published + flag"};
                for (int qq = 0; qq < 8; ++qq) fprintf(stderr, " %s %.1f |", sn[qq], (double)h[w * 24 + 10 + qq] / calls / (a.l1 - a.l0) / 1000.0);
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    return FD_OK;
}

int fd_trp_set_flag(fd_ctx* ctx, unsigned long long* flag, unsigned long long value, hipStream_t s) {
    hipLaunchKernelGGL(k_tr_set_flag, dim3(1), dim3(64), 0, s, flag, value);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}
