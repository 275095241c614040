// MiniLM-class sentence encoder on the matrix cores (replaces the CPU torch forward that
// sentence-transformers runs behind SentenceTransformerEmbeddingFunction, scripts/chromadb_helper.py:58-71).
//
// No LayerNorm kernel runs.  A block's output is stored un-normalised (bf16 z) together with per-row sums of the
// STORED values; whoever reads z applies the LayerNorm: a projection by linearity in its epilogue,
//   LN(z) W^T + b = rstd (z (W gamma)^T) - rstd mean colsum(W gamma) + (W beta + b)
// (k_fold_ln prepares W gamma, its column sums and W beta + b when the weights are loaded; rows carry
// (rstd, -mean rstd)), a residual add by normalising the value it adds.
//
//   k_embed_sum          word + position + type embedding gather              -> z bf16 [M, H], row sums
//   per layer:
//   k_gemm_ws<QKV>       LN(z) @ Wqkv^T + b; Q (pre-scaled) and K in MFMA operand order, V in fragment order
//   k_attention          per (sequence, head): S = K Q^T in registers (S <= 256), masked softmax, O = P V
//   k_gemm_ws<LN>        LN(ctx @ Wo^T + b + LN(z))                            -> y bf16 (normalised: the FFN reads it thrice)
//   k_gemm_ws<GELU>      gelu(y @ Wi^T + b)                                    -> h bf16 [M, I], down projection's operand order
//   k_gemm_down          h @ Wd^T + b + y                                      -> z bf16, row sums
//   k_pool_norm          LN of the last z per token, masked mean, L2 normalise -> out fp32 [B, H]
// (FFN widths other than 1536 take the generic k_gemm below with k_row_stats behind it.)
//
// GEMM: C[M,N] = A[M,K] W[N,K]^T, bf16 in, fp32 accumulate (v_mfma_f32_32x32x16_bf16). Persistent
// workgroups of 8 waves walk 256x192 output tiles; K advances in 64-wide stages that arrive in LDS
// by LDS-DMA (double buffered, one barrier per stage, the next tile's first stage already in flight
// during the epilogue). Both operands are K-contiguous, so either can be the MFMA's first operand:
// token-major outputs use D = W x A^T (token on the lane: each lane stores 4 consecutive features),
// the V output uses D = A x W^T (feature on the lane: each lane stores 4 consecutive tokens).
// LDS rows are 128 B; 16-byte piece p of row r is stored at p ^ ((r >> 1) & 7) (permuted on the
// DMA's source side), which makes the ds_read_b128 fragment reads conflict-free.
#include <type_traits>

#include "hrs_internal.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define ENC_H 384
#define ENC_HEADS 12
#define ENC_DH 32
#define G_BM 256
#define G_BN 192
#define G_BK 64
#define G_STAGE_BYTES ((G_BM + G_BN) * G_BK * 2)  // 57344

enum { EPI_QKV = 0, EPI_RESID = 1, EPI_GELU = 2, EPI_PLAIN = 3, EPI_LN = 4 };

struct GemmArgs {
    const __bf16* A;     // [M, K]
    const __bf16* W;     // [N, K]
    const float* bias;   // [N]
    const __bf16* resid; // [M, N] (EPI_RESID)
    __bf16* C;           // [M, N] (EPI_RESID, EPI_GELU)
    __bf16* q;           // EPI_QKV: [B][heads][S][32], pre-scaled
    __bf16* k;           // EPI_QKV: [B][heads][S][32]
    __bf16* vf;          // EPI_QKV: [B][heads][S/32][2][64][8]  (MFMA B-fragment order)
    int M, N, K, S;
    float qscale;
};

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) with erf as an odd polynomial of degree 13 on |x| <= 3.677
// (minimax fit, scaled to reach exactly +-1 at the clamp, |erf error| <= 2.4e-4, |GELU error| <= 4.4e-4:
// below the bf16 rounding of the stored activation).  No transcendental, and two values per instruction
// (v_pk_fma_f32 / v_pk_mul_f32): the FFN up-projection is bound by this epilogue, not by its MFMAs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    f32x2 xc;
    xc.x = __builtin_amdgcn_fmed3f(x.x, -3.67695526f, 3.67695526f);
    xc.y = __builtin_amdgcn_fmed3f(x.y, -3.67695526f, 3.67695526f);
    const f32x2 x2 = xc * xc;
    f32x2 q = __builtin_elementwise_fma(x2, (f32x2)(8.601600188e-08f), (f32x2)(-5.069113089e-06f));
    q = __builtin_elementwise_fma(q, x2, (f32x2)(1.296143067e-04f));
    q = __builtin_elementwise_fma(q, x2, (f32x2)(-1.924680215e-03f));
    q = __builtin_elementwise_fma(q, x2, (f32x2)(1.885473079e-02f));
    q = __builtin_elementwise_fma(q, x2, (f32x2)(-1.317864380e-01f));
    q = __builtin_elementwise_fma(q, x2, (f32x2)(7.976052351e-01f));
    const f32x2 hh = __builtin_elementwise_fma(xc * q, (f32x2)(0.5f), (f32x2)(0.5f));
    return x * hh;
}
// Four values side by side with plain (one value per instruction) fp32 math: beside MFMAs the SIMD's vector issue
// port is the scarce resource and a packed v_pk_fma_f32 takes more of it than the two v_fma_f32 it replaces
// (measured: 8.97 -> 8.79 ms per 1024 x 256 batch; the file is built with -fno-slp-vectorize for the same reason).
__device__ __forceinline__ void gelu_erf2x2(f32x2& a, f32x2& b) {
    // one value per instruction, four independent chains side by side (the same operations as gelu_erf2)
    float x[4] = {a.x, a.y, b.x, b.y}, c[4], s[4], q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = __builtin_amdgcn_fmed3f(x[j], -3.67695526f, 3.67695526f);
        s[j] = c[j] * c[j];
    }
    const float k[7] = {8.601600188e-08f, -5.069113089e-06f, 1.296143067e-04f, -1.924680215e-03f, 1.885473079e-02f, -1.317864380e-01f, 7.976052351e-01f};
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = __builtin_fmaf(s[j], k[0], k[1]);
#pragma unroll
    for (int t = 2; t < 7; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = __builtin_fmaf(q[j], s[j], k[t]);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = x[j] * __builtin_fmaf(c[j] * q[j], 0.5f, 0.5f);
    a.x = x[0];
    a.y = x[1];
    b.x = x[2];
    b.y = x[3];
}
__device__ __forceinline__ float gelu_erf(float x) {
    f32x2 v;
    v.x = x;
    v.y = x;
    return gelu_erf2(v).x;
}

template <int EPI>
__global__ __launch_bounds__(512) void k_gemm(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & 3, wn = wv >> 2;  // 4 waves along M (64 rows), 2 along N (96 columns)
    const int h = lane >> 5, i = lane & 31;
    const int ksteps = a.K / G_BK;
    const int tiles_n = a.N / G_BN;
    const int tiles = (a.M / G_BM) * tiles_n;
    // blocks that share an XCD (blockIdx % 8) take neighbouring tiles: they reuse A rows / W columns in L2
    const int per_xcd = (int)gridDim.x >> 3;
    const int bid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int my_tiles = bid < tiles ? (tiles - bid + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int stages = my_tiles * ksteps;
    if (stages == 0) return;
    const uint32_t lds0 = lds_addr_of(smem);

    // DMA source offsets: this wave moves 7 of the 56 one-KiB pieces of a stage (32 for A, 24 for W)
    int srcoff[7];
    bool isA[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int piece = wv * 7 + j;
        const bool pa = piece < 32;
        const int r = (pa ? piece : piece - 32) * 8 + (lane >> 3);  // row inside the A / W tile
        const int p = lane & 7;
        isA[j] = pa;
        srcoff[j] = (r * a.K + ((p ^ ((r >> 1) & 7)) << 3)) * 2;
    }
    auto issue = [&](int s, int buf) {
        const int t = bid + (s / ksteps) * (int)gridDim.x;
        const int kk = s % ksteps;
        const int tm = t / tiles_n, tn = t % tiles_n;
        const char* baseA = uniform_ptr(a.A + ((size_t)tm * G_BM * a.K + kk * G_BK));
        const char* baseW = uniform_ptr(a.W + ((size_t)tn * G_BN * a.K + kk * G_BK));
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + buf * G_STAGE_BYTES + wv * 7 * 1024);
#pragma unroll
        for (int j = 0; j < 7; ++j) glds16_asm(isA[j] ? baseA : baseW, srcoff[j], dst + j * 1024);
    };

    // fragment read offsets: k16-step ks of tile row i wants piece 2ks+h, stored at (2ks+h) ^ ((i>>1)&7)
    int off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) off[ks] = i * 128 + (((2 * ks + h) ^ ((i >> 1) & 7)) << 4);
    typedef const __attribute__((address_space(3))) char* lds_cptr;

    f32x16 acc[2][3];
    issue(0, 0);
    bool prev_epilogue = false;
    for (int s = 0; s < stages; ++s) {
        const int buf = s & 1;
        // retire this stage's DMA (older than the stores of a just-finished epilogue, if any)
        if (prev_epilogue)
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 1 < stages) issue(s + 1, buf ^ 1);
        const int kk = s % ksteps;
        if (kk == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        }
        lds_cptr pa = (lds_cptr)(size_t)(lds0 + buf * G_STAGE_BYTES + wm * 64 * 128);
        lds_cptr pw = (lds_cptr)(size_t)(lds0 + buf * G_STAGE_BYTES + G_BM * 128 + wn * 96 * 128);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[2], fw[3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) fa[mt] = *(const __attribute__((address_space(3))) bf16x8*)(pa + mt * 4096 + off[ks]);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) fw[nt] = *(const __attribute__((address_space(3))) bf16x8*)(pw + nt * 4096 + off[ks]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    if (EPI == EPI_QKV) {
                        // orientation is decided per wave below; both products are needed only for V
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[mt][nt], 0, 0, 0);
                    } else {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[mt][nt], 0, 0, 0);
                    }
                }
        }
        prev_epilogue = false;
        if (kk == ksteps - 1) {
            prev_epilogue = true;
            const int t = bid + (s / ksteps) * (int)gridDim.x;
            const int tm = t / tiles_n, tn = t % tiles_n;
            // D = W x A^T: lane <-> token (column), registers <-> features (rows):
            //   feature = n0 + (r & 3) + 8 (r >> 2) + 4 h, token = m0 + i
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int token = tm * G_BM + wm * 64 + mt * 32 + i;
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    const int n0 = tn * G_BN + wn * 96 + nt * 32;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int f = n0 + 8 * g + 4 * h;  // 4 consecutive features f .. f+3
                        const float4 bv = *(const float4*)(a.bias + f);
                        float v0 = acc[mt][nt][4 * g + 0] + bv.x, v1 = acc[mt][nt][4 * g + 1] + bv.y;
                        float v2 = acc[mt][nt][4 * g + 2] + bv.z, v3 = acc[mt][nt][4 * g + 3] + bv.w;
                        if (EPI == EPI_RESID) {
                            const bf16x4 rv = *(const bf16x4*)(a.resid + (size_t)token * a.N + f);
                            v0 += (float)rv[0];
                            v1 += (float)rv[1];
                            v2 += (float)rv[2];
                            v3 += (float)rv[3];
                        }
                        if (EPI == EPI_GELU) {
                            v0 = gelu_erf(v0);
                            v1 = gelu_erf(v1);
                            v2 = gelu_erf(v2);
                            v3 = gelu_erf(v3);
                        }
                        bf16x4 o;
                        if (EPI == EPI_QKV) {
                            const int which = f / ENC_H;          // 0 q, 1 k, 2 v (uniform per wave)
                            const int hf = f % ENC_H;
                            const int head = hf >> 5, d = hf & 31;  // d = 8g + 4h (+0..3)
                            const int b = token / a.S, sp = token % a.S;
                            if (which == 0) {
                                o[0] = (__bf16)(v0 * a.qscale);
                                o[1] = (__bf16)(v1 * a.qscale);
                                o[2] = (__bf16)(v2 * a.qscale);
                                o[3] = (__bf16)(v3 * a.qscale);
                                *(bf16x4*)(a.q + (((size_t)b * ENC_HEADS + head) * a.S + sp) * ENC_DH + d) = o;
                            } else if (which == 1) {
                                o[0] = (__bf16)v0;
                                o[1] = (__bf16)v1;
                                o[2] = (__bf16)v2;
                                o[3] = (__bf16)v3;
                                *(bf16x4*)(a.k + (((size_t)b * ENC_HEADS + head) * a.S + sp) * ENC_DH + d) = o;
                            } else {
                                // V in the order the attention kernel's P.V product consumes it as B operand:
                                // key = 32 kt + 16 s2 + 8 (e >> 2) + 4 hh + (e & 3), column = d
                                //   -> vf[kt][s2][lane' = 32 hh + d][e]; this lane holds ONE key (sp) and 4 columns
                                const int kt = sp >> 5, s2 = (sp >> 4) & 1, e = (((sp >> 3) & 1) << 2) | (sp & 3), hh = (sp >> 2) & 1;
                                __bf16* vb = a.vf + (((size_t)b * ENC_HEADS + head) * (a.S >> 5) + kt) * (2 * 64 * 8) + s2 * 512 + hh * 256 + e;
                                vb[(d + 0) * 8] = (__bf16)v0;
                                vb[(d + 1) * 8] = (__bf16)v1;
                                vb[(d + 2) * 8] = (__bf16)v2;
                                vb[(d + 3) * 8] = (__bf16)v3;
                            }
                        } else {
                            o[0] = (__bf16)v0;
                            o[1] = (__bf16)v1;
                            o[2] = (__bf16)v2;
                            o[3] = (__bf16)v3;
                            *(bf16x4*)(a.C + (size_t)token * a.N + f) = o;
                        }
                    }
                }
            }
        }
    }
}


// ---- weights-stationary GEMM for K = 384 ----------------------------------------------------------
// For the three GEMMs whose depth is the hidden size (QKV, attention output, FFN up) the weights fit
// the register file: a workgroup of 12 waves holds 384 output features (each wave 32 features x 384
// depth = 96 VGPRs) and streams 64-token activation tiles through a 3-deep LDS ring (LDS-DMA, two
// tiles in flight, one barrier per tile) -- the scan kernel's structure with an output epilogue.
// No weight byte is ever re-read; activations are read once per 384-feature group.
#ifdef HRS_ENC_PROF
// in-kernel phase clocks of the FFN up projection (10 ns units, summed over waves): see tools/enc_phases.py
static __device__ unsigned long long g_enc_prof[8];
extern "C" int hrs_encoder_prof_dump() {
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_enc_prof), sizeof(h)) != hipSuccess) return -1;
    const double w = (double)h[7] > 0 ? (double)h[7] : 1.0;  // waves
    fprintf(stderr, "gemm_ws<GELU> per wave (us): total %.1f | wait+barrier %.1f | mfma phase %.1f | epilogue %.1f (of which the two stores %.1f) | tiles/wave %.1f\n",
            h[0] / w / 100.0, h[1] / w / 100.0, h[2] / w / 100.0, h[3] / w / 100.0, h[5] / w / 100.0, h[4] / w);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_enc_prof), z, sizeof(z));
    return 0;
}
#endif

struct GemmWsArgs {
    const __bf16* A;     // [M, 384]
    const __bf16* W;     // [N, 384]
    const float* bias;   // [N]
    __bf16* C;           // [M, N] (EPI_GELU); fragment-major when frag_out is set (see there)
    // EPI_GELU: the output leaves in the order in which the next GEMM's MFMAs take it as an operand -- per 32 tokens
    // and 16 features one 1 KB block of 64 lanes x 16 bytes (lane (token i, half h): features 8h .. 8h+7), blocks
    // [M / 32][N / 16].  A store instruction of a wave then writes 1 KB of consecutive bytes instead of 32 bytes into
    // each of 32 rows (a third of this kernel's time went into the 25 M 32-byte write requests of the row-major form),
    // the consumer's LDS-DMA copies consecutive bytes and its fragment reads are lane-linear.
    int frag_out;
    __bf16* q;   // EPI_QKV: Q (pre-scaled), K and V all in MFMA operand order: [B][heads][S/32][2][64 lanes][8]
    __bf16* k;
    __bf16* vf;
    int M, N, S;
    float qscale;
    // EPI_LN (N = 384): C = LayerNorm(A W^T + bias + resid) * ln_g + ln_b
    const __bf16* resid;  // [M, 384]
    const float* ln_g;
    const float* ln_b;
    float eps;
    // The block input x arrives WITHOUT its LayerNorm, as the rows z that k_gemm_down (or the embedding kernel) stored
    // plus three partial (sum, sum of squares) per row: x = (z - mean) rstd g + b.
    // EPI_QKV reads it as the GEMM operand, by linearity: x W^T = rstd (z (W g)^T - mean colsum(W g)) + W b, so W holds
    // W g (bf16), csum its row sums and bias the original bias plus W b.  EPI_LN reads it as the residual and applies
    // rin_g / rin_b element by element.
    const float2* stats_in;  // [3][M]; EPI_QKV and EPI_LN always take their input this way
    const float* csum;       // EPI_QKV with stats_in: [N]
    const float* rin_g;      // EPI_LN with stats_in: [384]
    const float* rin_b;
};

// LDS behind the three-tile ring of k_gemm_ws, in floats: bias | csum or ln_g | ln_b | rin_g | row partials of EPI_LN
// [2][12 waves][32] float2 | (mean, rstd) of the rows of two tiles [2][64] float2 | per wave 64 floats: the partial
// statistics of eight rows of the next tile as the wave's DMA piece delivers them [3 partials][sum, sum of squares][8]
#define WS_OFF_PART 1536
#define WS_OFF_MR (WS_OFF_PART + 2 * 12 * 32 * 2)
#define WS_OFF_RAW (WS_OFF_MR + 2 * 64 * 2)
#define WS_LDS_BYTES (3 * 64 * 768 + (WS_OFF_RAW + 12 * 64) * 4)

// A row's normalisation (z - mean) rstd as z * A + B: (A, B) = (rstd, -mean rstd) from its three partial (sum, sum of
// squares), added in a fixed order
__device__ __forceinline__ float2 row_mean_rstd(float2 p0, float2 p1, float2 p2, float eps) {
    const float s1 = (p0.x + p1.x) + p2.x, s2 = (p0.y + p1.y) + p2.y;
    const float mean = s1 * (1.0f / ENC_H);
    const float rstd = rsqrtf(fmaxf(s2 * (1.0f / ENC_H) - mean * mean, 0.f) + eps);
    return make_float2(rstd, -mean * rstd);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Workgroup -> (token slice, feature group).  The ngroups workgroups that cover the feature groups of one
// token slice read the same activation tiles at the same pace; consecutive workgroup ids go to the eight
// XCDs in turn, each with an L2 of its own, so the groups of a slice are given ids that are congruent
// modulo 8: the slice then comes from HBM once instead of ngroups times.  Workgroups beyond the largest
// multiple of 8 * ngroups (255 = 240 + 15 for three groups) keep the plain order.
__device__ __forceinline__ void slice_of_block(int b, int grid, int ngroups, int& fg, int& slice, int& nslices) {
    const int per_xcd = (grid >> 3) / ngroups;  // whole slices per XCD
    const int main_blocks = per_xcd * ngroups * 8;
    nslices = per_xcd * 8 + (grid - main_blocks) / ngroups;
    if (b < main_blocks) {
        const int j = b >> 3;
        fg = j % ngroups;
        slice = (j / ngroups) * 8 + (b & 7);
    } else {
        const int r = b - main_blocks;
        fg = r % ngroups;
        slice = per_xcd * 8 + r / ngroups;
    }
}

// The 32x32 accumulator leaves a lane (token i, half h) with features 8g + 4h .. 8g + 4h + 3 of every
// 8-feature piece g: 8 bytes of bf16 per piece.  One v_permlane32_swap per dword between the lanes i and
// i + 32 gives the lower lane the whole of piece ga and the upper lane the whole of piece gb, so a lane
// stores 16 bytes and the two lanes of a token 32 contiguous bytes (half as many store instructions).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 piece_exchange(bf16x4 pa, bf16x4 pb) {
    const u32x2 a = (const u32x2&)pa, b = (const u32x2&)pb;
    const auto r0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    u32x4 o;
    o.x = r0[0];
    o.y = r1[0];
    o.z = r0[1];
    o.w = r1[1];
    return o;
}

template <int EPI, bool VW>
__device__ __forceinline__ void gemm_ws_body(const GemmWsArgs& a) {
    constexpr int KS = 24, TR = 64, RB = 768, TILE_BYTES = TR * RB, IPW = 4, PPR = 48, NST = 4;  // NST: stores per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, i = lane & 31;
    const int ngroups = a.N / 384;
    int fg, slice, nslices;
    slice_of_block((int)blockIdx.x, (int)gridDim.x, ngroups, fg, slice, nslices);
    const int wfeat = fg * 384 + wv * 32;  // this wave's 32 output features
    constexpr bool vwave = VW;

    bf16x8 wf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[s] = *(const bf16x8*)(a.W + (size_t)(wfeat + i) * 384 + s * 16 + h * 8);
    // this workgroup's 384 biases live in LDS behind the tile ring (read back in the epilogue)
    float* bias_lds = (float*)(smem + 3 * TILE_BYTES);
    const bool fold = a.stats_in != nullptr;  // the input rows still need their LayerNorm (see GemmWsArgs)
    // (EPI_LN with fold: the residual x = (z - mean) rstd g + b enters as bias + b here and as g (z - mean) rstd there)
    if (threadIdx.x < 384) bias_lds[threadIdx.x] = a.bias[fg * 384 + threadIdx.x] + (EPI == EPI_LN && fold ? a.rin_b[threadIdx.x] : 0.f);
    if (EPI == EPI_QKV && fold && threadIdx.x < 384) bias_lds[384 + threadIdx.x] = a.csum[fg * 384 + threadIdx.x];
    if (EPI == EPI_LN && threadIdx.x < 384) {
        bias_lds[384 + threadIdx.x] = a.ln_g[threadIdx.x];
        bias_lds[768 + threadIdx.x] = a.ln_b[threadIdx.x];
        if (fold) bias_lds[1152 + threadIdx.x] = a.rin_g[threadIdx.x];
    }
    float2* part = (float2*)(bias_lds + WS_OFF_PART);  // EPI_LN: [2][12 waves][32 tokens] (sum, sum of squares)
    float2* mr = (float2*)(bias_lds + WS_OFF_MR);      // fold: [2][64 tokens] (mean, rstd) of the input rows of two tiles
    float* raw = bias_lds + WS_OFF_RAW;
    int pb = 0;

    const int tiles = a.M / TR;
    const int t_begin = (int)((int64_t)tiles * slice / nslices);
    const int t_end = (int)((int64_t)tiles * (slice + 1) / nslices);
    const int count = t_end - t_begin;
    // the first tile's row statistics the ordinary way (nothing of the ring is in flight yet); those of tile n + 1
    // are fetched by wave 0 during tile n (see the loop)
    if (fold && wv == 0 && count > 0) {
        const size_t tok = (size_t)t_begin * TR + lane;
        mr[lane] = row_mean_rstd(a.stats_in[tok], a.stats_in[(size_t)a.M + tok], a.stats_in[2 * (size_t)a.M + tok], a.eps);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wf[s]));
    if (count <= 0) return;

    int srcoff[IPW];
#pragma unroll
    for (int ii = 0; ii < IPW; ++ii) {
        const int g = (wv * IPW + ii) * 64 + lane;
        const int r = g / PPR, p = g % PPR;
        srcoff[ii] = r * RB + ((p ^ (r & 15)) << 4);
    }
    const uint32_t lds0 = lds_addr_of(smem);
    auto issue = [&](int n, int buf) {
        if (n >= count) n = count - 1;
        const char* base = uniform_ptr((const char*)a.A + (size_t)(t_begin + n) * TILE_BYTES);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + buf * TILE_BYTES + wv * (IPW * 1024));
        glds16_run_asm<IPW>(base, srcoff, dst);
    };
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    const int rowoff = (int)lds0 + i * RB;
    const int vx = h ^ (i & 15);

#ifdef HRS_ENC_PROF
    long long p_store = 0;
#endif
    // epilogue of one 32-token subtile for the outputs without LayerNorm: 4 stores per lane
    auto epilogue = [&](const f32x16& acc, const int tok0, const int mr_half, const int mr_sub) __attribute__((always_inline)) {
        if (vwave) {
            // lane <-> feature d = i (+ head), registers <-> tokens tok0 + (r&3) + 8(r>>2) + 4h
            const int head = (wfeat - 768) >> 5;
            const int b = tok0 / a.S, sp0 = tok0 % a.S;  // 32-token tiles never straddle sequences (S % 32 == 0)
            const int kt = sp0 >> 5;
            __bf16* vb = a.vf + (((size_t)b * ENC_HEADS + head) * (a.S >> 5) + kt) * 1024 + (h * 32 + i) * 8;
            const float bz = bias_lds[wv * 32 + i];
            const float cs = fold ? bias_lds[384 + wv * 32 + i] : 0.f;
            // fold: (A, B) of the tokens on the registers, four consecutive tokens = two 16-byte reads (the same address
            // in every lane of a half wave); value = A acc + (B cs + bias)
            const float4* mrt = (const float4*)(mr + mr_half * 64 + mr_sub * 32 + 4 * h);
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                bf16x8 o;
#pragma unroll
                for (int eh = 0; eh < 2; ++eh) {
                    float4 ab0 = make_float4(1.f, 0.f, 1.f, 0.f), ab1 = ab0;
                    if (fold) {
                        ab0 = mrt[4 * (2 * g2 + eh)];
                        ab1 = mrt[4 * (2 * g2 + eh) + 1];
                    }
                    const int r0 = 8 * g2 + 4 * eh;
                    o[4 * eh + 0] = (__bf16)fmaf(ab0.x, acc[r0 + 0], fmaf(ab0.y, cs, bz));
                    o[4 * eh + 1] = (__bf16)fmaf(ab0.z, acc[r0 + 1], fmaf(ab0.w, cs, bz));
                    o[4 * eh + 2] = (__bf16)fmaf(ab1.x, acc[r0 + 2], fmaf(ab1.y, cs, bz));
                    o[4 * eh + 3] = (__bf16)fmaf(ab1.z, acc[r0 + 3], fmaf(ab1.w, cs, bz));
                }
                *(bf16x8*)(vb + g2 * 512) = o;
            }
        } else {
            const int token = tok0 + i;
            bf16x4 o[4];
            float rs = 1.f, rb = 0.f;  // fold: value = A acc + (B colsum + bias) with this lane's token's (A, B)
            if (EPI == EPI_QKV && fold) {
                const float2 t = mr[mr_half * 64 + mr_sub * 32 + i];
                rs = t.x;
                rb = t.y;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(bias_lds + wv * 32 + 8 * g + 4 * h);
                float v0, v1, v2, v3;
                if (EPI == EPI_QKV && fold) {
                    const float4 cv = *(const float4*)(bias_lds + 384 + wv * 32 + 8 * g + 4 * h);
                    v0 = fmaf(rs, acc[4 * g + 0], fmaf(rb, cv.x, bv.x));
                    v1 = fmaf(rs, acc[4 * g + 1], fmaf(rb, cv.y, bv.y));
                    v2 = fmaf(rs, acc[4 * g + 2], fmaf(rb, cv.z, bv.z));
                    v3 = fmaf(rs, acc[4 * g + 3], fmaf(rb, cv.w, bv.w));
                } else {
                    v0 = acc[4 * g + 0] + bv.x, v1 = acc[4 * g + 1] + bv.y;
                    v2 = acc[4 * g + 2] + bv.z, v3 = acc[4 * g + 3] + bv.w;
                }
#ifdef HRS_ENC_NOGELU
                if (false) {
#else
                if (EPI == EPI_GELU) {
#endif
                    f32x2 g01, g23;
                    g01.x = v0;
                    g01.y = v1;
                    g23.x = v2;
                    g23.y = v3;
                    gelu_erf2x2(g01, g23);
                    v0 = g01.x;
                    v1 = g01.y;
                    v2 = g23.x;
                    v3 = g23.y;
                } else if (EPI == EPI_QKV) {
                    const float sc = fg == 0 ? a.qscale : 1.0f;
                    v0 *= sc;
                    v1 *= sc;
                    v2 *= sc;
                    v3 *= sc;
                }
                o[g][0] = (__bf16)v0;
                o[g][1] = (__bf16)v1;
                o[g][2] = (__bf16)v2;
                o[g][3] = (__bf16)v3;
            }
            // lower lane: pieces 0 and 2, upper lane: pieces 1 and 3 (features 8 * piece .. + 7 of this wave's 32)
            __bf16* dst = a.C + (size_t)token * a.N + wfeat;
#ifdef HRS_ENC_NOSTORE
            const u32x4 e0 = piece_exchange(o[0], o[1]), e1 = piece_exchange(o[2], o[3]);
            // (never taken, but every value stays live: an ablation must not let the compiler drop the arithmetic)
            if ((e0.x ^ e0.y ^ e0.z ^ e0.w ^ e1.x ^ e1.y ^ e1.z ^ e1.w) == 0x12345678u) *(u32x4*)(dst + 8 * h) = e0;
#else
            if (EPI == EPI_QKV) {
                // Q and K leave as the attention kernel's MFMA operands: this wave's 32 features are head wv, two 16-deep
                // k-steps = the two exchanged registers, so a store writes 1 KB of consecutive bytes
                // [sequence][head][32-token tile][k-step][64 lanes][8] (32-token tiles never straddle sequences)
                const int b = tok0 / a.S, qt = (tok0 % a.S) >> 5;
                __bf16* fb = (fg == 0 ? a.q : a.k) + ((((size_t)b * ENC_HEADS + wv) * (a.S >> 5) + qt) * 128 + lane) * 8;
                *(u32x4*)fb = piece_exchange(o[0], o[1]);
                *(u32x4*)(fb + 512) = piece_exchange(o[2], o[3]);
            } else if (EPI == EPI_GELU && a.frag_out) {
                // this wave's features are two 16-feature k-steps of the consumer: two consecutive 1 KB blocks
                __bf16* fb = a.C + (((size_t)(tok0 >> 5) * (a.N >> 4) + (wfeat >> 4)) * 64 + lane) * 8;
#ifdef HRS_ENC_PROF
                u32x4 e0 = piece_exchange(o[0], o[1]), e1 = piece_exchange(o[2], o[3]);
                asm volatile("" : "+v"(e0), "+v"(e1));
                const long long ps0 = wall_clock64();
                *(u32x4*)fb = e0;
                *(u32x4*)(fb + 512) = e1;
                asm volatile("" ::: "memory");
                p_store += wall_clock64() - ps0;
#else
                *(u32x4*)fb = piece_exchange(o[0], o[1]);
                *(u32x4*)(fb + 512) = piece_exchange(o[2], o[3]);
#endif
            } else {
                *(u32x4*)(dst + 8 * h) = piece_exchange(o[0], o[1]);
                *(u32x4*)(dst + 16 + 8 * h) = piece_exchange(o[2], o[3]);
            }
#endif
        }
    };

#ifdef HRS_ENC_PROF
    const long long pk0 = wall_clock64();
    long long p_wait = 0, p_mfma = 0, p_epi = 0;
    p_store = 0;
#endif
    issue(0, 0);
    issue(1, 1);
    int buf = 0;
    for (int n = 0; n < count; ++n) {
#ifdef HRS_ENC_PROF
        const long long pw0 = wall_clock64();
#endif
        // outstanding, oldest first: DMA(n), DMA(n+1), then the NST stores of tile n-1 (none before the first tile)
        // (EPI_LN: two subtiles of 4 stores and DMA(n+1) between them: its DMA is issued in the middle of the tile)
        if (n == 0)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW + NST) : "memory");
        __builtin_amdgcn_s_barrier();
#ifdef HRS_ENC_PROF
        p_wait += wall_clock64() - pw0;
#endif
        int nb = buf + 2;
        if (nb >= 3) nb -= 3;
        // fold: the partial row statistics of tile n + 1 come into LDS by one 4-byte DMA piece per wave: waves 0..7 fetch
        // the six values of eight rows each (lane l: row l & 7, value l >> 3; waves 8..11 repeat wave 7's piece into an
        // area of their own, so that every wave counts the same operations).  The piece is issued in front of this
        // iteration's tile DMA, so the wait that ends the iteration -- the very wait the next one begins with -- covers
        // it; the wave then turns its eight rows into (mean, rstd) in the other half of `mr`, which all waves read
        // behind the next barrier.  A wave's part of `raw` is touched by that wave alone.
        float* raw_w = raw + wv * 64;
        if (fold) {
            const int tn = n + 1 < count ? n + 1 : count - 1;
            const int sw = wv < 8 ? wv : 7;
            const char* sbase = uniform_ptr((const char*)a.stats_in + ((size_t)(t_begin + tn) * TR + sw * 8) * 8);
            int ll = lane;
            asm volatile("" : "+v"(ll));  // (the offset is recomputed here each time rather than held in a register across the loop)
            const int sp = ll < 48 ? ll >> 4 : 2;
            glds4_asm(sbase, (sp * a.M + (ll & 7)) * 8 + ((ll >> 3) & 1) * 4, lds_addr_of(raw_w));
        }
        auto stats_to_lds = [&]() __attribute__((always_inline)) {
            int ll = lane;
            asm volatile("" : "+v"(ll));  // (addresses recomputed here, not held across the loop)
            if (wv < 8 && ll < 8)
                mr[((n + 1) & 1) * 64 + wv * 8 + ll] = row_mean_rstd(make_float2(raw_w[ll], raw_w[8 + ll]), make_float2(raw_w[16 + ll], raw_w[24 + ll]),
                                                                     make_float2(raw_w[32 + ll], raw_w[40 + ll]), a.eps);
        };
        if (EPI != EPI_LN) issue(n + 2, nb);
        // EPI_LN: the residual rows come by asm loads so that their waits can be counted by hand next to
        // the DMA's (a load the compiler knows about would be waited for with everything older, DMA included)
        u32x2 rr[4];
        int roff = 0;  // this lane's residual pieces: + 16 bytes per piece
        if (EPI == EPI_LN) {
            int ll = lane;
            asm volatile("" : "+v"(ll));  // (recomputed per tile, see above)
            roff = ((ll & 31) * 384 + wv * 32 + 4 * (ll >> 5)) * 2;
        }
        const char* rbase = nullptr;
        auto load_resid = [&](const char* rb) __attribute__((always_inline)) {
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(rr[0]) : "v"(roff), "s"(rb) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "=&v"(rr[1]) : "v"(roff), "s"(rb) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, %2 offset:32" : "=&v"(rr[2]) : "v"(roff), "s"(rb) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, %2 offset:48" : "=&v"(rr[3]) : "v"(roff), "s"(rb) : "memory");
        };
        if (EPI == EPI_LN) {
            rbase = uniform_ptr((const char*)a.resid + (size_t)(t_begin + n) * TILE_BYTES);
            load_resid(rbase);
        }
        lds_cptr tb[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) tb[c] = (lds_cptr)(size_t)(uint32_t)(rowoff + buf * TILE_BYTES + (((2 * c) ^ vx) << 4));
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
#ifdef HRS_ENC_PROF
            const long long pm0 = wall_clock64();
#endif
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int PF = (EPI == EPI_LN ? 2 : 4);
            bf16x8 av[KS];
#pragma unroll
            for (int s = 0; s < PF; ++s)
                av[s] = *(const __attribute__((address_space(3))) bf16x8*)(tb[s & 7] + sb * 32 * RB + (s >> 3) * 256);
            if (vwave) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s + PF < KS)
                        av[s + PF] = *(const __attribute__((address_space(3))) bf16x8*)(tb[(s + PF) & 7] + sb * 32 * RB + ((s + PF) >> 3) * 256);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s], wf[s], acc, 0, 0, 0);  // D[token][feature]
                }
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s + PF < KS)
                        av[s + PF] = *(const __attribute__((address_space(3))) bf16x8*)(tb[(s + PF) & 7] + sb * 32 * RB + ((s + PF) >> 3) * 256);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], av[s], acc, 0, 0, 0);  // D[feature][token]
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
            for (int s = 0; s < KS - PF; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);

#ifdef HRS_ENC_PROF
            asm volatile("s_nop 0" : "+v"(acc));  // the clock is read when the last result exists
            const long long pm1 = wall_clock64();
            p_mfma += pm1 - pm0;
#endif
            const int tok0 = (t_begin + n) * TR + sb * 32;
            if (EPI == EPI_LN) {
                const int token = tok0 + i;
                f32x16& v = acc;  // the sums replace the products where they stand
                // residual of this subtile has landed; younger: nothing (subtile 0) or the DMA issued below (subtile 1)
                if (sb == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3])::"memory");
                    if (fold) stats_to_lds();  // (its DMA piece is covered by the wait above)
                } else {
                    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3]) : "n"(IPW) : "memory");
                }
                // the residual row still needs its LayerNorm: (A, B) of this lane's token
                float rs = 1.f, rb = 0.f;
                if (fold) {
                    const float2 t = mr[(n & 1) * 64 + sb * 32 + i];
                    rs = t.x;
                    rb = t.y;
                }
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *(const float4*)(bias_lds + wv * 32 + 8 * g + 4 * h);
                    float r[4] = {__uint_as_float(rr[g].x << 16), __uint_as_float(rr[g].x & 0xffff0000u),
                                  __uint_as_float(rr[g].y << 16), __uint_as_float(rr[g].y & 0xffff0000u)};
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                    if (fold) {
                        const float4 ig = *(const float4*)(bias_lds + 1152 + wv * 32 + 8 * g + 4 * h);
                        const float gg[4] = {ig.x, ig.y, ig.z, ig.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * g + e] = fmaf(fmaf(r[e], rs, rb), gg[e], v[4 * g + e] + bb[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * g + e] = v[4 * g + e] + bb[e] + r[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1 += v[4 * g + e];
                        s2 = fmaf(v[4 * g + e], v[4 * g + e], s2);
                    }
                    asm volatile("" : "+v"(s1), "+v"(s2));  // one piece at a time: keeps the LDS operands of the next from piling up
                }
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (h == 0) part[(pb * 12 + wv) * 32 + i] = make_float2(s1, s2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < 12; ++w) {
                    const float2 pv = part[(pb * 12 + w) * 32 + i];
                    t1 += pv.x;
                    t2 += pv.y;
                }
                pb ^= 1;
                const float mean = t1 * (1.0f / 384.0f);
                const float rstd = rsqrtf(fmaxf(t2 * (1.0f / 384.0f) - mean * mean, 0.f) + a.eps);
                bf16x4 o[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = wv * 32 + 8 * g + 4 * h;
                    const float4 gv = *(const float4*)(bias_lds + 384 + f);
                    const float4 be = *(const float4*)(bias_lds + 768 + f);
                    o[g][0] = (__bf16)((v[4 * g + 0] - mean) * rstd * gv.x + be.x);
                    o[g][1] = (__bf16)((v[4 * g + 1] - mean) * rstd * gv.y + be.y);
                    o[g][2] = (__bf16)((v[4 * g + 2] - mean) * rstd * gv.z + be.z);
                    o[g][3] = (__bf16)((v[4 * g + 3] - mean) * rstd * gv.w + be.w);
                }
                __bf16* dst = a.C + (size_t)token * 384 + wv * 32;
                *(u32x4*)(dst + 8 * h) = piece_exchange(o[0], o[1]);
                *(u32x4*)(dst + 16 + 8 * h) = piece_exchange(o[2], o[3]);
                if (sb == 0) {
                    // residual of the second subtile, then the tile two ahead (its buffer was freed by the barrier at the top)
                    load_resid(rbase + 32 * RB);
                    issue(n + 2, nb);
                }
            } else {
                epilogue(acc, tok0, n & 1, sb);
            }
#ifdef HRS_ENC_PROF
            p_epi += wall_clock64() - pm1;
#endif
        }
        if (EPI != EPI_LN && fold) {
            // younger than the statistics piece: this iteration's tile DMA and its stores
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW + NST) : "memory");
            stats_to_lds();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // in LDS before this wave joins the next barrier
        }
        buf = buf + 1 == 3 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef HRS_ENC_PROF
    if (EPI == EPI_GELU && lane == 0) {
        atomicAdd(&g_enc_prof[0], (unsigned long long)(wall_clock64() - pk0));
        atomicAdd(&g_enc_prof[1], (unsigned long long)p_wait);
        atomicAdd(&g_enc_prof[2], (unsigned long long)p_mfma);
        atomicAdd(&g_enc_prof[3], (unsigned long long)p_epi);
        atomicAdd(&g_enc_prof[4], (unsigned long long)count);
        atomicAdd(&g_enc_prof[5], (unsigned long long)p_store);
        atomicAdd(&g_enc_prof[7], 1ull);
    }
#endif
}

template <int EPI>
__global__ __launch_bounds__(768) void k_gemm_ws(GemmWsArgs a) {
    // the V third of the QKV projection is produced with the operands swapped (feature on the lane)
    int fg, slice, nslices;
    slice_of_block((int)blockIdx.x, (int)gridDim.x, a.N / 384, fg, slice, nslices);
    if (EPI == EPI_QKV && fg == 2)
        gemm_ws_body<EPI, true>(a);
    else
        gemm_ws_body<EPI, false>(a);
}


// ---- weights-stationary GEMM for the FFN down projection (K = 1536, N = 384) -----------------------
// The 1.18 MB weight matrix does not fit one CU's registers, a third of it does: a workgroup of four
// waves (one per SIMD, the whole 512-register file each) holds 128 output features x 1536 depth
// (96 fragments = 384 registers per lane) and three workgroups cover a row.  Activations stream as
// 32-token x 512-deep stages (32 KB) through a 4-deep LDS ring by LDS-DMA, three stages in flight;
// a token tile is finished every third stage.  Output: bias added, bf16, token-major.
struct GemmDownArgs {
    const __bf16* A;      // fragment-major [M / 32][96][64 lanes][8]: GemmWsArgs::frag_out
    const __bf16* W;      // [384, 1536]
    const float* bias;    // [384]
    const __bf16* resid;  // [M, 384]: the block's input, added before the statistics are taken
    __bf16* C;            // [M, 384]: A W^T + bias + resid, NOT normalised (its readers apply the LayerNorm)
    float2* stats;        // [3][M]: per feature group, (sum, sum of squares) of a row's 128 stored (rounded) values
    int M;
};

// MFMA with the accumulator and (optionally) the weight fragment in the accumulator half of the
// register file.  hipcc never picks AGPR source operands by itself, so these are asm; the leading
// s_nop covers a preceding VALU write of an operand, the caller pads the result's read (mfma_fence).
__device__ __forceinline__ void mfma_wa(f32x16& acc, const bf16x8& w, const bf16x8& x) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(x));
}
__device__ __forceinline__ void mfma_wv(f32x16& acc, const bf16x8& w, const bf16x8& x) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
}
__device__ __forceinline__ void mfma_fence(f32x16& acc) {  // 8-pass MFMA result -> any other reader
    asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc));
}

// The block's LayerNorm is not applied here: the row sums it needs span the three workgroups that share a token
// slice.  Each writes the (sum, sum of squares) of its 128 features of every row -- taken from the bf16 values it
// stores, so that the readers normalise exactly what they read -- and the readers (the next QKV projection, the
// residual input of the next output projection, the pooling) finish the LayerNorm on the fly.
__global__ __launch_bounds__(256, 1) void k_gemm_down(GemmDownArgs a) {
    constexpr int K = 1536, KC = 512, SPT = 3, KSS = 32, ROWB = KC * 2, STAGE = 32 * ROWB, IPW = 8, NBUF = 4;
    constexpr int NA = 56;  // weight fragments kept in AGPRs (232 of 256; 16 hold the accumulator)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, i = lane & 31;
    int fg, slice, nslices;
    slice_of_block((int)blockIdx.x, (int)gridDim.x, 3, fg, slice, nslices);
    const int wfeat = fg * 128 + wv * 32;

    bf16x8 wf[SPT * KSS];
#pragma unroll
    for (int s = 0; s < SPT * KSS; ++s) wf[s] = *(const bf16x8*)(a.W + (size_t)(wfeat + i) * K + s * 16 + h * 8);
    // behind the ring: this workgroup's 128 biases, the per-wave row sums of a finished tile [2][4 waves][32 rows], and
    // per wave the 32 rows x 32 features of the residual that it adds (2 KB, by LDS-DMA: a load into registers would
    // have to stay untouched across a stage, and the compiler is free to copy a register before its data has arrived)
    float* bias_lds = (float*)(smem + NBUF * STAGE);
    float2* part = (float2*)(bias_lds + 128);
    char* resid_lds = (char*)(part + 2 * 4 * 32) + wv * 2048;
    if (threadIdx.x < 128) bias_lds[threadIdx.x] = a.bias[fg * 128 + threadIdx.x];
#pragma unroll
    for (int s = 0; s < SPT * KSS; ++s) {
        if (s < NA)
            asm volatile("" : "+a"(wf[s]));
        else
            asm volatile("" : "+v"(wf[s]));
    }
    __syncthreads();

    const int tiles = a.M / 32;
    const int t_begin = (int)((int64_t)tiles * slice / nslices);
    const int t_end = (int)((int64_t)tiles * (slice + 1) / nslices);
    const int stages = (t_end - t_begin) * SPT;
    if (stages > 0) {
        int srcoff[IPW];
#pragma unroll
        for (int ii = 0; ii < IPW; ++ii) {
            srcoff[ii] = ((wv * IPW + ii) * 64 + lane) * 16;  // a stage is 32 consecutive 1 KB operand fragments
        }
        const uint32_t lds0 = lds_addr_of(smem);
        auto issue = [&](int g) {
            if (g >= stages) g = stages - 1;
            const int t = t_begin + g / SPT, kc = g % SPT;
            const char* base = uniform_ptr((const char*)a.A + ((size_t)t * (K / 16) + kc * KSS) * 1024);
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(g % NBUF) * STAGE + wv * (IPW * 1024));
            glds16_run_asm<IPW>(base, srcoff, dst);
        };
        // the row sums of tile t (complete in `part` behind a barrier) -> memory: every wave one store, 8 rows each
        auto flush_stats = [&](int t) {
            if (lane < 8) {
                const int r = wv * 8 + lane;
                const float2* pp = part + (t & 1) * 128 + r;
                const float2 p0 = pp[0], p1 = pp[32], p2 = pp[64], p3 = pp[96];
                a.stats[(size_t)fg * a.M + (size_t)(t_begin + t) * 32 + r] = make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
            }
        };
        typedef const __attribute__((address_space(3))) char* lds_cptr;
        typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
        const int fragoff = (int)lds0 + lane * 16;  // fragment s of a stage: + s * 1024, lane-linear
        // residual DMA, two pieces per wave: lane l of piece j fetches the 16 bytes (8 features) l >> 4 of row 16 j + (l & 15)
        // of this wave's 32 columns; a row's four pieces then sit 256 bytes apart and the epilogue's reads of 16
        // consecutive rows fall on 16 different bank groups
        const int roff = ((lane & 15) * ENC_H + wfeat + (lane >> 4) * 8) * 2;
        const uint32_t resid_dst = lds_addr_of(resid_lds);

        issue(0);
        issue(1);
        issue(2);
        // One iteration per token tile with its three stages written out: the accumulator lives inside the iteration
        // (carried around the loop, the compiler moved it between two register ranges at every stage, and such a copy
        // right behind the last MFMA of a stage reads results that are not interlocked for it).
        const int ntiles = stages / SPT;
        for (int tile = 0; tile < ntiles; ++tile) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int PF = 6;
            auto stage = [&](auto kc_tag) __attribute__((always_inline)) {
                constexpr int KCI = decltype(kc_tag)::value;
                const int g = tile * SPT + KCI;
                // Vector-memory operations of a wave per tile, in order: stage 0: the statistics store of the tile before
                // (1), DMA (8); stage 1: residual DMA (2), DMA (8); stage 2: DMA (8), output stores (2).  Each stage waits
                // until its own DMA -- issued three stages earlier -- has landed and leaves everything younger in flight.
                if (tile == 0)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KCI == 2 ? 2 * IPW + 2 : 2 * IPW) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KCI == 0 ? 2 * IPW + 2 + 2 : (KCI == 1 ? 2 * IPW + 2 + 1 : 2 * IPW + 2 + 1 + 2)) : "memory");
                __builtin_amdgcn_s_barrier();
                if (KCI == 0 && tile > 0) flush_stats(tile - 1);
                if (KCI == 1) {
                    const char* rbase = uniform_ptr((const char*)a.resid + (size_t)(t_begin + tile) * 32 * ENC_H * 2);
                    glds16_asm(rbase, roff, resid_dst);
                    glds16_asm(rbase + 16 * ENC_H * 2, roff, resid_dst + 1024);
                }
                issue(g + 3);
                const lds_cptr tb = (lds_cptr)(size_t)(uint32_t)(fragoff + (g % NBUF) * STAGE);
                bf16x8 av[KSS];
#pragma unroll
                for (int s = 0; s < PF; ++s) av[s] = *(lds_frag)(tb + s * 1024);
#pragma unroll
                for (int s = 0; s < KSS; ++s) {
                    if (s + PF < KSS) av[s + PF] = *(lds_frag)(tb + (s + PF) * 1024);
                    if (KCI * KSS + s < NA)
                        mfma_wa(acc, wf[KCI * KSS + s], av[s]);  // D[feature][token]
                    else
                        mfma_wv(acc, wf[KCI * KSS + s], av[s]);
                }
                mfma_fence(acc);  // (every stage: nothing the compiler places behind it may meet an unfinished MFMA)
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 2>{});
            // the residual rows have landed; younger than them: the DMA of this stage and of the one before
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
            const int token = (t_begin + tile) * 32 + i;
            const char* rrow = resid_lds + (i >> 4) * 1024 + (i & 15) * 16 + h * 8;  // + 256 per 8-feature piece
            bf16x4 o[4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 bv = *(const float4*)(bias_lds + wv * 32 + 8 * gq + 4 * h);
                const u32x2 rr = *(const u32x2*)(rrow + gq * 256);
                o[gq][0] = (__bf16)(acc[4 * gq + 0] + bv.x + __uint_as_float(rr.x << 16));
                o[gq][1] = (__bf16)(acc[4 * gq + 1] + bv.y + __uint_as_float(rr.x & 0xffff0000u));
                o[gq][2] = (__bf16)(acc[4 * gq + 2] + bv.z + __uint_as_float(rr.y << 16));
                o[gq][3] = (__bf16)(acc[4 * gq + 3] + bv.w + __uint_as_float(rr.y & 0xffff0000u));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = (float)o[gq][e];
                    s1 += r;
                    s2 = fmaf(r, r, s2);
                }
            }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (h == 0) part[((tile & 1) * 4 + wv) * 32 + i] = make_float2(s1, s2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // in LDS before this wave joins the next barrier
            __bf16* dst = a.C + (size_t)token * ENC_H + wfeat;
            *(u32x4*)(dst + 8 * h) = piece_exchange(o[0], o[1]);
            *(u32x4*)(dst + 16 + 8 * h) = piece_exchange(o[2], o[3]);
        }
        __syncthreads();
        flush_stats(stages / SPT - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- attention -----------------------------------------------------------------------------------
// One workgroup (4 waves) per (sequence, head); a wave owns 32-query tiles. With S <= 256 the whole
// score row lives in registers: S^T = K Q^T puts the query on the lane, so the row maximum and sum
// are in-lane reductions plus one exchange between lane halves; P (bf16) then feeds O^T = V^T P^T as the
// second MFMA operand with no lane movement (the k order of that product is the accumulator's
// register order, which is the order V was stored in by the QKV epilogue), so the query is still on the
// lane in the output, 1/l needs no shuffle and a context row leaves as two 16-byte stores per lane.
struct AttnArgs {
    const __bf16* q;   // [B][heads][S/32][2][64][8] (MFMA operand order), scaled by log2(e)/sqrt(32)
    const __bf16* k;   // the same order
    const __bf16* vf;  // [B][heads][S/32][2][64][8]
    const int* lens;   // [B]
    __bf16* ctx;       // [B*S][384]
    int S;
};

// Two waves per SIMD are asked for so that the register budget is 256: the compiler then keeps the MFMA
// results in ordinary registers (with a 512-register budget it puts them in the accumulator half and
// copies every score tile out and every initial value in, 26 register moves per MFMA here).
#define ATT_NOSHIFT_BOUND 100.0f  // |q| |k| up to which the scores go through exp2 unshifted
template <int NKT>
__global__ __launch_bounds__(256, 2) void k_attention(AttnArgs a) {
    // K and V of this (sequence, head), both stored by the QKV projection in the order the MFMAs take them
    // (fragment reads are lane-linear), shared by the four waves and by both passes
    __shared__ __attribute__((aligned(16))) char lds[2 * NKT * 32 * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int h = lane >> 5, i = lane & 31;
    const int bh = blockIdx.x;
    const int b = bh / ENC_HEADS, head = bh % ENC_HEADS;
    constexpr int S = NKT * 32;
    int len = a.lens[b];
    if (len > S) len = S;
    const int nkt = len > 0 ? (len + 31) >> 5 : 1;  // key tiles that hold at least one valid key
    const char* qb = (const char*)(a.q + (size_t)bh * S * ENC_DH);
    const char* kb = (const char*)(a.k + (size_t)bh * S * ENC_DH);
    const char* vb = (const char*)(a.vf + (size_t)bh * S * ENC_DH);
    char* klds = lds;
    char* vlds = lds + NKT * 32 * 64;
    for (int pc = tid; pc < nkt * 128; pc += 256) {  // 16-byte pieces, both already in fragment order: a straight copy
        *(uint4*)(klds + pc * 16) = *(const uint4*)(kb + pc * 16);
        *(uint4*)(vlds + pc * 16) = *(const uint4*)(vb + pc * 16);
    }
    __syncthreads();
    const int koff0 = lane * 16, koff1 = 1024 + lane * 16;  // the two k-steps of a key tile: lane-linear, conflict-free
    // The largest key norm of this (sequence, head): |q . k| <= |q| |k| bounds every score of a query without computing
    // one.  Where that bound is small enough for exp2 to take the scores as they are -- it nearly always is: scores are
    // in units of log2 e / sqrt(32) -- the row maximum is not needed and pass 1 below is skipped (softmax is invariant
    // under the shift; P = 2^s stays within 2^+-100, its sum over 256 keys and its products with V within fp32).
    __shared__ float kmax_w[4];
    {
        float k2 = 0.f;
        if (tid < nkt * 32) {  // thread <-> key: its 32 values are four 16-byte pieces of the operand order
            const int kt = tid >> 5, j = tid & 31;
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const bf16x8 kv = *(const bf16x8*)(klds + ((kt * 2 + (pc >> 1)) * 64 + j + 32 * (pc & 1)) * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) k2 = fmaf((float)kv[e], (float)kv[e], k2);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) k2 = fmaxf(k2, __shfl_xor(k2, d));
        if (lane == 0) kmax_w[wv] = k2;
    }
    __syncthreads();
    const float kmax2 = fmaxf(fmaxf(kmax_w[0], kmax_w[1]), fmaxf(kmax_w[2], kmax_w[3]));

    for (int qt = wv; qt < NKT; qt += 4) {
        // the key masks of a partial tile are recomputed where they are used: hoisted out of this loop they
        // are 128 lane masks that no longer fit the scalar registers
        asm volatile("" : "+s"(len));
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(qb + ((qt * 2 + ks) * 64 + lane) * 16);
        // this lane's query: half of its 32 values are here, the other half on lane ^ 32
        float q2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) q2 = fmaf((float)qf[ks][e], (float)qf[ks][e], q2);
        q2 += __shfl_xor(q2, 32);
        float bound2 = q2 * kmax2;  // (|q| |k|)^2 of the tile's largest query
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) bound2 = fmaxf(bound2, __shfl_xor(bound2, d));
        const bool shifted = !(__builtin_amdgcn_readfirstlane(__float_as_int(bound2)) <= __float_as_int(ATT_NOSHIFT_BOUND * ATT_NOSHIFT_BOUND));
        // pass 1 (only where the bound does not settle it): row maximum.  Scores are recomputed in pass 2 (two MFMAs per
        // tile) instead of keeping the 256-wide row in registers, which would leave one wave per SIMD and no latency hiding.
        float m = shifted ? -1e30f : 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (shifted && kt < nkt) {
                f32x16 sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(klds + kt * 2048 + koff0), qf[0], sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(klds + kt * 2048 + koff1), qf[1], sc, 0, 0, 0);
                if (kt * 32 + 32 <= len) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, sc[r]);
                } else {
                    const int lim = len - kt * 32 - 4 * h;  // register r holds key kt*32 + (r&3) + 8(r>>2) + 4h
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, (r & 3) + 8 * (r >> 2) < lim ? sc[r] : -1e30f);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        // pass 2: the accumulator starts at -m (0 without pass 1), so the MFMA result is already s - m
        float l = 0.f;
        f32x16 o, negm;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt < nkt) {
                f32x16 sc;
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(klds + kt * 2048 + koff0), qf[0], negm, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(klds + kt * 2048 + koff1), qf[1], sc, 0, 0, 0);
                if (kt * 32 + 32 <= len) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sc[r] = __builtin_amdgcn_exp2f(sc[r]);
                        l += sc[r];
                    }
                } else {
                    const int lim = len - kt * 32 - 4 * h;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sc[r] = (r & 3) + 8 * (r >> 2) < lim ? __builtin_amdgcn_exp2f(sc[r]) : 0.f;
                        l += sc[r];
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (__bf16)sc[8 * s2 + e];
                    // O^T = V^T P^T: the query stays on the lane, the registers run over the head's 32 columns
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(vlds + ((kt * 2 + s2) * 64 + lane) * 16), pf, o, 0, 0, 0);
                }
            }
        }
        l += __shfl_xor(l, 32);
        const float inv = l > 0.f ? 1.0f / l : 0.f;  // per query = per lane pair (i, i + 32)
        // o[r] = O[query qt*32 + i][column (r&3) + 8(r>>2) + 4h]: a lane pair holds the 64 bytes of its query's
        // context row for this head; two 16-byte stores per lane after the exchange between the halves
        bf16x4 ob[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[g][e] = (__bf16)(o[4 * g + e] * inv);
        }
        __bf16* dst = a.ctx + ((size_t)b * S + qt * 32 + i) * ENC_H + head * ENC_DH;
        *(u32x4*)(dst + 8 * h) = piece_exchange(ob[0], ob[1]);
        *(u32x4*)(dst + 16 + 8 * h) = piece_exchange(ob[2], ob[3]);
    }
}

// ---- row-wise kernels ----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per token: 384 features = 6 per lane.  The embedding LayerNorm is left to the readers of x (as the
// block-output LayerNorm is, see k_gemm_down): the row goes out as the bf16 sum of its three embeddings, with the
// (sum, sum of squares) of those rounded values as the first of its three partial statistics (the others zero).
__global__ __launch_bounds__(256) void k_embed_sum(const int* __restrict__ ids, const float* __restrict__ word,
                                                   const float* __restrict__ pos, const float* __restrict__ type0, int M,
                                                   int Mreal, int S, int vocab, __bf16* __restrict__ x,
                                                   float2* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= M) return;
    int id = t < Mreal ? ids[t] : 0;
    if (id < 0 || id >= vocab) id = 0;
    const int sp = t % S;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int c = lane + 64 * j;
        const __bf16 o = (__bf16)(word[(size_t)id * ENC_H + c] + pos[(size_t)sp * ENC_H + c] + type0[c]);
        x[(size_t)t * ENC_H + c] = o;
        const float r = (float)o;
        s1 += r;
        s2 = fmaf(r, r, s2);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane < 3) stats[(size_t)lane * M + t] = lane == 0 ? make_float2(s1, s2) : make_float2(0.f, 0.f);
}

// (sum, sum of squares) of bf16 rows, for the path whose down projection is the generic GEMM: one wave per row.
__global__ __launch_bounds__(256) void k_row_stats(const __bf16* __restrict__ x, int M, float2* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= M) return;
    float s1 = 0.f, s2 = 0.f;
    if (lane < 48) {
        const bf16x8 r = *(const bf16x8*)(x + (size_t)t * ENC_H + lane * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (float)r[j];
            s1 += v;
            s2 = fmaf(v, v, s2);
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane < 3) stats[(size_t)lane * M + t] = lane == 0 ? make_float2(s1, s2) : make_float2(0.f, 0.f);
}

// The last block's LayerNorm, masked mean over the sequence, L2 normalisation.  One block of 384 threads (6 waves)
// per sequence: wave w sums (z - mean) rstd over the tokens w, w + 6, ... with 16-byte reads (lanes 0..47 hold 8
// features each), the six partial rows meet in LDS, then thread c owns feature c: the mean over the tokens of
// LayerNorm(z) is g_c / len * that sum + b_c.
__global__ __launch_bounds__(384) void k_pool_norm(const __bf16* __restrict__ x, const float2* __restrict__ stats, int M,
                                                   const float* __restrict__ g, const float* __restrict__ bta, float eps,
                                                   const int* __restrict__ lens, int S, float* __restrict__ out) {
    __shared__ float part[6][ENC_H];
    __shared__ float red[6];
    const int b = blockIdx.x, c = threadIdx.x;
    const int lane = c & 63, wv = c >> 6;
    int len = lens[b];
    if (len > S) len = S;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (lane < 48) {
        for (int t = wv; t < len; t += 6) {
            const size_t tok = (size_t)b * S + t;
            const bf16x8 r = *(const bf16x8*)(x + tok * ENC_H + lane * 8);
            const float2 mr = row_mean_rstd(stats[tok], stats[(size_t)M + tok], stats[2 * (size_t)M + tok], eps);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += fmaf((float)r[j], mr.x, mr.y);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wv][lane * 8 + j] = acc[j];
    }
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 6; ++w) s += part[w][c];
    const float mean = len > 0 ? fmaf(g[c], s / (float)len, bta[c]) : 0.f;
    float ss = wave_sum(mean * mean);
    if (lane == 0) red[wv] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) tot += red[j];
    const float nrm = fmaxf(sqrtf(tot), 1e-12f);
    out[(size_t)b * ENC_H + c] = mean / nrm;
}

// W [N, 384] fp32, LayerNorm weights g / b of the rows W is applied to -> Wg = bf16(W g), csum = row sums of the
// ROUNDED Wg (what the matrix cores will multiply the row mean with), bias_out = bias + W b.  One wave per row.
__global__ __launch_bounds__(256) void k_fold_ln(const float* __restrict__ W, const float* __restrict__ g,
                                                 const float* __restrict__ b, const float* __restrict__ bias, int N,
                                                 __bf16* __restrict__ Wg, float* __restrict__ csum, float* __restrict__ bias_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float cs = 0.f, wb = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int k = lane + 64 * j;
        const float w = W[(size_t)n * ENC_H + k];
        const __bf16 o = (__bf16)(w * g[k]);
        Wg[(size_t)n * ENC_H + k] = o;
        cs += (float)o;
        wb = fmaf(w, b[k], wb);
    }
    cs = wave_sum(cs);
    wb = wave_sum(wb);
    if (lane == 0) {
        csum[n] = cs;
        bias_out[n] = bias[n] + wb;
    }
}

__global__ void k_to_bf16(const float* __restrict__ in, int64_t n, __bf16* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = (__bf16)in[idx];
}

// ---- host side -----------------------------------------------------------------------------------
struct hrs_encoder {
    hrs_encoder_config cfg;
    // fp32 tables / vectors (owned copies)
    float *word = nullptr, *pos = nullptr, *type0 = nullptr, *emb_g = nullptr, *emb_b = nullptr;
    struct Layer {
        __bf16 *wqkv = nullptr, *wo = nullptr, *wi = nullptr, *wd = nullptr;  // wqkv: W g of the LayerNorm in front of it
        float *bqkv = nullptr, *bo = nullptr, *bi = nullptr, *bd = nullptr;   // bqkv: bias + W b of that LayerNorm
        float* cqkv = nullptr;                                                // row sums of wqkv
        float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
    } layer[16];
    std::vector<void*> owned;
    hrs_workspace ws;
};

static int dev_alloc(hrs_encoder* e, void** p, size_t bytes) {
    hipError_t err = hipMalloc(p, bytes);
    if (err != hipSuccess) {
        hrs_set_error("encoder allocation of %zu bytes failed: %s", bytes, hipGetErrorString(err));
        return HRS_ERR_NOMEM;
    }
    e->owned.push_back(*p);
    return HRS_OK;
}

static int copy_f32(hrs_encoder* e, float** dst, const float* src_dev, size_t n, hipStream_t st) {
    int rc = dev_alloc(e, (void**)dst, n * sizeof(float));
    if (rc) return rc;
    HRS_CHECK_HIP(hipMemcpyAsync(*dst, src_dev, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return HRS_OK;
}

static int conv_bf16(hrs_encoder* e, __bf16* dst, const float* src_dev, size_t n, hipStream_t st) {
    (void)e;
    hipLaunchKernelGGL(k_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src_dev, (int64_t)n, dst);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

extern "C" int hrs_encoder_load(hrs_ctx* ctx, const hrs_encoder_config* cfg, const float* const* w, void* stream,
                                hrs_encoder** out) {
    HRS_REQUIRE(ctx && cfg && w && out, "hrs_encoder_load: null argument");
    HRS_REQUIRE(cfg->hidden == ENC_H && cfg->heads == ENC_HEADS, "hrs_encoder_load: only hidden=384, heads=12 (MiniLM-L6/L12 geometry) is built");
    HRS_REQUIRE(cfg->inter % G_BN == 0 && cfg->inter % G_BK == 0 && cfg->inter <= 3072, "hrs_encoder_load: intermediate size %d must be a multiple of 192", cfg->inter);
    HRS_REQUIRE(cfg->layers >= 1 && cfg->layers <= 16, "hrs_encoder_load: layers=%d outside [1,16]", cfg->layers);
    HRS_REQUIRE(cfg->vocab > 0 && cfg->max_pos >= 32, "hrs_encoder_load: bad vocabulary / position table size");
    const int nt = 5 + 16 * cfg->layers;
    for (int t = 0; t < nt; ++t) HRS_REQUIRE(w[t] != nullptr, "hrs_encoder_load: weight tensor %d is null", t);
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    hrs_encoder* e = new hrs_encoder();
    e->cfg = *cfg;
    const size_t H = ENC_H, I = cfg->inter;
    int rc = 0;
#define TRY(x)                          \
    do {                                \
        rc = (x);                       \
        if (rc) goto fail;              \
    } while (0)
    TRY(copy_f32(e, &e->word, w[0], (size_t)cfg->vocab * H, st));
    TRY(copy_f32(e, &e->pos, w[1], (size_t)cfg->max_pos * H, st));
    TRY(copy_f32(e, &e->type0, w[2], H, st));
    TRY(copy_f32(e, &e->emb_g, w[3], H, st));
    TRY(copy_f32(e, &e->emb_b, w[4], H, st));
    for (int l = 0; l < cfg->layers; ++l) {
        const float* const* p = w + 5 + 16 * l;  // q_w q_b k_w k_b v_w v_b o_w o_b ln1_g ln1_b i_w i_b d_w d_b ln2_g ln2_b
        hrs_encoder::Layer& L = e->layer[l];
        TRY(dev_alloc(e, (void**)&L.wqkv, 3 * H * H * sizeof(__bf16)));
        TRY(dev_alloc(e, (void**)&L.bqkv, 3 * H * sizeof(float)));
        TRY(dev_alloc(e, (void**)&L.cqkv, 3 * H * sizeof(float)));
        {
            // The LayerNorm in front of this block (the embedding's, or the block before's second) is folded into the
            // projection that reads the block's input (see GemmWsArgs::stats_in).
            const float* in_g = l == 0 ? w[3] : w[5 + 16 * (l - 1) + 14];
            const float* in_b = l == 0 ? w[4] : w[5 + 16 * (l - 1) + 15];
            for (int j = 0; j < 3; ++j)
                hipLaunchKernelGGL(k_fold_ln, dim3((unsigned)((H + 3) / 4)), dim3(256), 0, st, p[2 * j], in_g, in_b, p[2 * j + 1], (int)H,
                                   L.wqkv + j * H * H, L.cqkv + j * H, L.bqkv + j * H);
            if (hipGetLastError() != hipSuccess) {
                hrs_set_error("hrs_encoder_load: folding the LayerNorm into the QKV weights failed");
                rc = HRS_ERR_HIP;
                goto fail;
            }
        }
        TRY(dev_alloc(e, (void**)&L.wo, H * H * sizeof(__bf16)));
        TRY(conv_bf16(e, L.wo, p[6], H * H, st));
        TRY(copy_f32(e, &L.bo, p[7], H, st));
        TRY(copy_f32(e, &L.ln1g, p[8], H, st));
        TRY(copy_f32(e, &L.ln1b, p[9], H, st));
        TRY(dev_alloc(e, (void**)&L.wi, I * H * sizeof(__bf16)));
        TRY(conv_bf16(e, L.wi, p[10], I * H, st));
        TRY(copy_f32(e, &L.bi, p[11], I, st));
        TRY(dev_alloc(e, (void**)&L.wd, H * I * sizeof(__bf16)));
        TRY(conv_bf16(e, L.wd, p[12], H * I, st));
        TRY(copy_f32(e, &L.bd, p[13], H, st));
        TRY(copy_f32(e, &L.ln2g, p[14], H, st));
        TRY(copy_f32(e, &L.ln2b, p[15], H, st));
    }
#undef TRY
    *out = e;
    return HRS_OK;
fail:
    for (void* p : e->owned) (void)hipFree(p);
    delete e;
    return rc;
}

extern "C" int hrs_encoder_release(hrs_ctx* ctx, hrs_encoder* e) {
    if (!e) return HRS_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (void* p : e->owned) (void)hipFree(p);
    if (e->ws.ptr) (void)hipFree(e->ws.ptr);
    delete e;
    return HRS_OK;
}

template <int EPI>
static int launch_gemm(hrs_ctx* ctx, const GemmArgs& g, hipStream_t st) {
    static hrs_per_device_once attr;
    const int lds = 2 * G_STAGE_BYTES;
    if (attr.need(ctx->device)) {
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr.done(ctx->device);
    }
    const int tiles = (g.M / G_BM) * (g.N / G_BN);
    int grid = ctx->num_cu & ~7;
    if (grid < 8) grid = 8;
    if (tiles < grid) grid = (tiles + 7) & ~7;
    hipLaunchKernelGGL((k_gemm<EPI>), dim3(grid), dim3(512), lds, st, g);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

template <int EPI>
static int launch_gemm_ws(hrs_ctx* ctx, const GemmWsArgs& g, hipStream_t st) {
    static hrs_per_device_once attr;
    const int lds = WS_LDS_BYTES;
    if (attr.need(ctx->device)) {
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_ws<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr.done(ctx->device);
    }
    const int ngroups = g.N / 384;
    int grid = ctx->num_cu / ngroups * ngroups;
    const int tiles = g.M / 64;
    if (tiles * ngroups < grid) grid = tiles * ngroups;
    hipLaunchKernelGGL((k_gemm_ws<EPI>), dim3(grid), dim3(768), lds, st, g);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

static int launch_gemm_down(hrs_ctx* ctx, const GemmDownArgs& g, hipStream_t st) {
    static hrs_per_device_once attr;
    const int lds = 4 * 32 * 1024 + 128 * 4 + 2 * 4 * 32 * 8 + 4 * 2048;  // ring, biases, row sums, residual rows
    if (attr.need(ctx->device)) {
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_down, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr.done(ctx->device);
    }
    int grid = ctx->num_cu / 3 * 3;
    const int tiles = g.M / 32;
    if (tiles * 3 < grid) grid = tiles * 3;
    hipLaunchKernelGGL(k_gemm_down, dim3(grid), dim3(256), lds, st, g);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

static int launch_attention(const AttnArgs& a, int B, hipStream_t st) {
    const int nkt = a.S / 32;
    const dim3 grid(B * ENC_HEADS), block(256);
    switch (nkt) {
        case 1: hipLaunchKernelGGL((k_attention<1>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((k_attention<2>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((k_attention<3>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((k_attention<4>), grid, block, 0, st, a); break;
        case 5: hipLaunchKernelGGL((k_attention<5>), grid, block, 0, st, a); break;
        case 6: hipLaunchKernelGGL((k_attention<6>), grid, block, 0, st, a); break;
        case 7: hipLaunchKernelGGL((k_attention<7>), grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL((k_attention<8>), grid, block, 0, st, a); break;
        default: hrs_set_error("attention: sequence length %d not a multiple of 32 in [32, 256]", a.S); return HRS_ERR_ARG;
    }
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

extern "C" int hrs_encode(hrs_ctx* ctx, hrs_encoder* e, const int32_t* ids_dev, const int32_t* lens_dev, int B, int S,
                          float* out_dev, void* stream) {
    HRS_REQUIRE(ctx && e && ids_dev && lens_dev && out_dev, "hrs_encode: null argument");
    HRS_REQUIRE(B >= 1, "hrs_encode: B=%d < 1", B);
    HRS_REQUIRE(S >= 32 && S <= 256 && S % 32 == 0 && S <= e->cfg.max_pos, "hrs_encode: S=%d must be a multiple of 32 in [32, 256]", S);
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int H = ENC_H, I = e->cfg.inter;
    const int Mreal = B * S;
    // whole sequences up to a multiple of 256 rows (8 sequences of 32 tokens = 256)
    const int Bpad = (B + 7) & ~7;
    const int M = Bpad * S;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    // x: a block's input / output rows WITHOUT their LayerNorm, st: their partial row statistics (see GemmWsArgs::stats_in)
    const size_t o_x = carve((size_t)M * H * 2), o_y = carve((size_t)M * H * 2), o_st = carve((size_t)3 * M * sizeof(float2));
    const size_t o_ctx = carve((size_t)M * H * 2), o_q = carve((size_t)M * H * 2), o_k = carve((size_t)M * H * 2);
    const size_t o_v = carve((size_t)M * H * 2), o_h = carve((size_t)M * I * 2), o_len = carve((size_t)Bpad * 4);
    int rc = hrs_ws_reserve(ctx, &e->ws, off);
    if (rc) return rc;
    char* ws = (char*)e->ws.ptr;
    __bf16 *x = (__bf16*)(ws + o_x), *y = (__bf16*)(ws + o_y), *cx = (__bf16*)(ws + o_ctx);
    __bf16 *q = (__bf16*)(ws + o_q), *k = (__bf16*)(ws + o_k), *vf = (__bf16*)(ws + o_v), *hbuf = (__bf16*)(ws + o_h);
    float2* stats = (float2*)(ws + o_st);
    int* lens = (int*)(ws + o_len);
    HRS_CHECK_HIP(hipMemsetAsync(lens, 0, (size_t)Bpad * 4, st));
    HRS_CHECK_HIP(hipMemcpyAsync(lens, lens_dev, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_embed_sum, dim3((M + 3) / 4), dim3(256), 0, st, ids_dev, e->word, e->pos, e->type0, M, Mreal, S,
                       e->cfg.vocab, x, stats);
    for (int l = 0; l < e->cfg.layers; ++l) {
        const hrs_encoder::Layer& L = e->layer[l];
        // the LayerNorm that x still needs: the embedding's, or the second one of the block before
        const float* in_g = l == 0 ? e->emb_g : e->layer[l - 1].ln2g;
        const float* in_b = l == 0 ? e->emb_b : e->layer[l - 1].ln2b;
        GemmArgs g;
        GemmWsArgs gw;
        memset(&gw, 0, sizeof(gw));
        gw.M = M;
        gw.S = S;
        gw.eps = e->cfg.eps;
        gw.A = x;
        gw.W = L.wqkv;
        gw.bias = L.bqkv;
        gw.csum = L.cqkv;
        gw.stats_in = stats;
        gw.N = 3 * H;
        gw.q = q;
        gw.k = k;
        gw.vf = vf;
        gw.qscale = 0.17677669529663687f * 1.4426950408889634f;  // 1/sqrt(32) * log2(e): softmax in base 2
        if ((rc = launch_gemm_ws<EPI_QKV>(ctx, gw, st))) return rc;
        AttnArgs at;
        at.q = q;
        at.k = k;
        at.vf = vf;
        at.lens = lens;
        at.ctx = cx;
        at.S = S;
        if ((rc = launch_attention(at, Bpad, st))) return rc;
        gw.A = cx;
        gw.W = L.wo;
        gw.bias = L.bo;
        gw.csum = nullptr;
        gw.C = y;
        gw.N = H;
        gw.resid = x;
        gw.rin_g = in_g;
        gw.rin_b = in_b;
        gw.ln_g = L.ln1g;
        gw.ln_b = L.ln1b;
        if ((rc = launch_gemm_ws<EPI_LN>(ctx, gw, st))) return rc;  // attention output projection + residual + LayerNorm
        gw.stats_in = nullptr;  // y is normalised
        gw.A = y;
        gw.W = L.wi;
        gw.bias = L.bi;
        gw.C = hbuf;
        gw.N = I;
        gw.frag_out = I == 1536;  // k_gemm_down takes the hidden activations in its own operand order
        if ((rc = launch_gemm_ws<EPI_GELU>(ctx, gw, st))) return rc;
        if (I == 1536) {
            GemmDownArgs gd;
            gd.A = hbuf;
            gd.W = L.wd;
            gd.bias = L.bd;
            gd.resid = y;
            gd.C = x;
            gd.stats = stats;
            gd.M = M;
            if ((rc = launch_gemm_down(ctx, gd, st))) return rc;
        } else {
            memset(&g, 0, sizeof(g));
            g.M = M;
            g.S = S;
            g.C = x;
            g.N = H;
            g.A = hbuf;
            g.W = L.wd;
            g.bias = L.bd;
            g.resid = y;
            g.K = I;
            if ((rc = launch_gemm<EPI_RESID>(ctx, g, st))) return rc;
            hipLaunchKernelGGL(k_row_stats, dim3((M + 3) / 4), dim3(256), 0, st, x, M, stats);
        }
    }
#ifdef HRS_ENC_DEBUG
    if (const char* dump = getenv("HRS_ENC_DUMP")) {  // debug builds only: the last block's output rows, their statistics, y
        (void)hipStreamSynchronize(st);
        std::vector<char> hb((size_t)M * H * 2), hy((size_t)M * H * 2), hs((size_t)3 * M * 8);
        (void)hipMemcpy(hb.data(), x, hb.size(), hipMemcpyDeviceToHost);
        (void)hipMemcpy(hy.data(), y, hy.size(), hipMemcpyDeviceToHost);
        (void)hipMemcpy(hs.data(), stats, hs.size(), hipMemcpyDeviceToHost);
        FILE* f = fopen(dump, "wb");
        if (f) {
            fwrite(&M, 4, 1, f);
            fwrite(hb.data(), 1, hb.size(), f);
            fwrite(hy.data(), 1, hy.size(), f);
            fwrite(hs.data(), 1, hs.size(), f);
            fclose(f);
        }
    }
#endif
    {
        const hrs_encoder::Layer& Ll = e->layer[e->cfg.layers - 1];
        hipLaunchKernelGGL(k_pool_norm, dim3(B), dim3(384), 0, st, x, stats, M, Ll.ln2g, Ll.ln2b, e->cfg.eps, lens, S, out_dev);
    }
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}
