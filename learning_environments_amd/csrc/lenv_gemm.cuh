// lenv_gemm.cuh -- workgroup-cooperative, LDS-tiled fp32 GEMM on the f32-input matrix cores with the canonical
// (k-ascending fmaf chain) reduction order.
//
//   C[i][j] = epi(i, j, sum_{r<R} P[i*sPi + r*sPr] * Q[j*sQj + r*sQr])       one block: I <= MAXI (128 or 256), J <= 128, any R;
//                                                                           larger outputs run block by block (gemm_run_queue)
//
// 512 threads (8 waves).  The reduction is staged through LDS 64 deep (Ps/Qs, r-major rows); the
// output is cut into 32x32 tiles, wave w owns tiles w, w+8, ... and each tile is accumulated with
// v_mfma_f32_32x32x2_f32.  That instruction is bit-for-bit D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) (one rounding per
// product, k ascending), so every output is exactly the sequential fmaf chain of oracle/lenv_oracle.h starting from 0,
// whatever the strides -- the same routine serves the forward (r = input feature), input-gradient (r = output unit) and
// weight-gradient (r = sample) products of the big-net agents (DuelingDDQN, TD3).  An odd last k is added with one
// scalar fmaf per output (a zero-padded k would turn a -0 accumulator into +0).
// Why MFMA when the peak rate equals v_pk_fma_f32's: a 4x8 register tile needs 24.6 KB of LDS reads per k for the
// workgroup (LDS-bound at 2.7x the FMA time, measured); the MFMA operands are one dword per lane (3 KB per k).
// ONE instance (descriptor-driven epilogue) serves every product of a kernel (see 'GEMM programs').  The routine is instruction-issue
// bound around the MFMAs (2 waves per SIMD), so every descriptor field is made wave-uniform (readfirstlane -> SGPRs,
// scalar branches), global operands use 32-bit offsets off a scalar base, and the per-output switches are hoisted out
// of the register loops.  Operands that live in LDS (a few small vectors) take a generic-pointer path (GEMM_GENERIC_*).
#pragma once

#include "lenv_device.cuh"

namespace lenv {

#ifndef LENV_DNT
#define LENV_DNT 512
#endif
constexpr int DNT = LENV_DNT;     // threads per chain (8 or 16 waves)
constexpr int DNW = DNT / 64;     // waves per chain
constexpr int SKQ = 2048 / DNT;   // float4 pieces per thread and 128-row block of a 64-deep stage
constexpr int GT_I = 128, GT_J = 128, GT_RB = 64, GT_LD = 132;   // 128x128 outputs, 64-deep stages, padded LDS rows

enum { EPI_STORE = 0,       // out = acc
       EPI_BIAS_ACT = 1,    // out = act(acc + bias[j])
       EPI_BIAS = 2,        // out = acc + bias[j]
       EPI_ACT_BWD = 3,     // out = act'(aux[i][j]) * acc      (aux = the layer's activation)
       EPI_ACCUM = 4,       // out = out + acc
       EPI_BIAS_TANH = 5,   // t = tanh(acc + bias[j]); out2 = t (optional); out = t * scale
       EPI_BIAS_ADD = 6 };  // out = aux[i][j] + (acc + bias[j])     (residual connection)

enum { GEMM_GENERIC_P = 1, GEMM_GENERIC_OUT = 2 };   // the P operand / the output may live in LDS (generic pointers)

struct GemmEpi {
    int kind;
    float *out; int ldo, ocol;          // out[i*ldo + ocol + j]
    const float *bias;                  // global
    const float *aux; int ldaux;        // EPI_ACT_BWD / EPI_BIAS_ADD, global
    float *out2; int ldo2;              // EPI_BIAS_TANH, global
    int act; float prelu, scale;
    int flags;
};

struct GemmOp {
    const float *P; int sPi, sPr;
    const float *Q; int sQj, sQr;       // Q is always global
    int I, J, R;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(3))) float lfloat;

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float unif(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
template <class T> __device__ __forceinline__ T *uni_ptr(T *p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)uni((int)(uint32_t)v), hi = (uint32_t)uni((int)(uint32_t)(v >> 32));
    return reinterpret_cast<T *>(((uint64_t)hi << 32) | lo);
}

// memory views: global (scalar base + 32-bit offset) or generic
template <bool G> struct MemView;
template <> struct MemView<true> {
    typedef __attribute__((address_space(1))) char gchar;
    const gchar *p;                                        // uniform base; 32-bit byte offsets -> `global_* v, v, s[..]` forms
    __device__ __forceinline__ explicit MemView(const float *q) : p((const gchar *)q) {}
    __device__ __forceinline__ float ld(int idx) const { return *reinterpret_cast<const gfloat *>(p + (uint32_t)(idx << 2)); }
    __device__ __forceinline__ f32x4 ld4(int idx) const { return *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(p + (uint32_t)(idx << 2)); }
    __device__ __forceinline__ void st(int idx, float v) const { *reinterpret_cast<gfloat *>(const_cast<gchar *>(p) + (uint32_t)(idx << 2)) = v; }
};
template <> struct MemView<false> {
    const float *p;
    __device__ __forceinline__ explicit MemView(const float *q) : p(q) {}
    __device__ __forceinline__ float ld(int idx) const { return p[idx]; }
    __device__ __forceinline__ f32x4 ld4(int idx) const { return *reinterpret_cast<const f32x4 *>(p + idx); }
    __device__ __forceinline__ void st(int idx, float v) const { const_cast<float *>(p)[idx] = v; }
};

__device__ __forceinline__ int pow2_shift(int n) { return n <= 1 ? 0 : 32 - __builtin_clz(n - 1); }   // ceil(log2 n)

// Copy src[i*sI + (r0+r)*sR], i < nI, r < rb, into dst[r*ld + i] (all scalars wave-uniform).  A thread's global reads
// are issued before its LDS writes (staging is latency bound).  Rows i >= nI are not needed: they only feed outputs that
// the epilogue drops.
template <bool G>
__device__ __forceinline__ void stage_operand(const float *src_, int sI, int sR, int nI, int r0, int rb, int R, lfloat *dst, int ld, int tid, int lane, int wave)
{
    const MemView<G> src(src_);
    const bool al16 = (reinterpret_cast<uintptr_t>(src_) & 15) == 0;
    if (G && sR == 1 && al16 && (sI & 3) == 0 && (R & 3) == 0) {
        // row-major along r: float4 = 4 consecutive r of one row; a wave instruction covers 16 rows x 64 B
        for (int ib = 0; ib < nI; ib += 128) {
            f32x4 v[SKQ];
            // unit u = tid + k*DNT of the 2048 float4 pieces: lane bits 0-1 and bits 9-10 of u -> the r quad, the rest -> the row
            const int i = ib + ((lane >> 2) | ((wave & 7) << 4));
            const bool rowok = i < nI;
#pragma unroll
            for (int k = 0; k < SKQ; ++k) {
                const int q = (lane & 3) | ((((wave >> 3) + (DNW >> 3) * k) & 3) << 2);
                v[k] = (rowok && 4 * q < rb) ? src.ld4(i * sI + r0 + 4 * q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int k = 0; k < SKQ; ++k) {
                const int q = (lane & 3) | ((((wave >> 3) + (DNW >> 3) * k) & 3) << 2);
                if (rowok && 4 * q < rb) {
                    lfloat *d = dst + (4 * q) * ld + i;
                    d[0] = v[k].x; d[ld] = v[k].y; d[2 * ld] = v[k].z; d[3 * ld] = v[k].w;
                }
            }
        }
    } else if (G && sI == 1 && al16 && (sR & 3) == 0) {
        // contiguous along i: float4 = 4 consecutive i of one r; a wave instruction covers 2 r x 512 B
        for (int ib = 0; ib < nI; ib += 128) {
            f32x4 v[SKQ];
            const int i4 = ib + 4 * (lane & 31);
            const bool colok = i4 < nI;
#pragma unroll
            for (int k = 0; k < SKQ; ++k) {
                const int r = (lane >> 5) | (wave << 1) | (k * 2 * DNW);
                v[k] = (colok && r < rb) ? src.ld4((r0 + r) * sR + i4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int k = 0; k < SKQ; ++k) {
                const int r = (lane >> 5) | (wave << 1) | (k * 2 * DNW);
                if (colok && r < rb) *reinterpret_cast<__attribute__((address_space(3))) f32x4 *>(dst + r * ld + i4) = v[k];
            }
        }
    } else if (sR == 1 || (sI != 1 && sR < sI)) {
        // small / unaligned operands, r fastest across lanes: e -> (i = e >> sh, r = e & mask)
        const int sh = pow2_shift(rb), mask = (1 << sh) - 1, total = nI << sh;
        for (int e0 = tid; e0 < total; e0 += 4 * DNT) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * DNT, i = e >> sh, r = e & mask;
                v[u] = (e < total && r < rb) ? src.ld(i * sI + (r0 + r) * sR) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * DNT, i = e >> sh, r = e & mask;
                if (e < total && r < rb) dst[r * ld + i] = v[u];
            }
        }
    } else {
        // i fastest across lanes: e -> (r = e >> sh, i = e & mask)
        const int sh = pow2_shift(nI), mask = (1 << sh) - 1, total = rb << sh;
        for (int e0 = tid; e0 < total; e0 += 4 * DNT) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * DNT, r = e >> sh, i = e & mask;
                v[u] = (e < total && i < nI) ? src.ld(i * sI + (r0 + r) * sR) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * DNT, r = e >> sh, i = e & mask;
                if (e < total && i < nI) dst[r * ld + i] = v[u];
            }
        }
    }
}

__device__ __forceinline__ void act_fwd16(int act, float prelu, float (&z)[16])
{
    switch (act) {                                         // uniform: one arm runs
    case LENV_ACT_RELU:
#pragma unroll
        for (int v = 0; v < 16; ++v) z[v] = z[v] > 0.0f ? z[v] : 0.0f;
        break;
    case LENV_ACT_LEAKYRELU:
#pragma unroll
        for (int v = 0; v < 16; ++v) z[v] = z[v] > 0.0f ? z[v] : z[v] * 0.01f;
        break;
    case LENV_ACT_TANH:
#pragma unroll
        for (int v = 0; v < 16; ++v) z[v] = det_tanhf(lenv_tanh_table, z[v]);
        break;
    case LENV_ACT_PRELU:
#pragma unroll
        for (int v = 0; v < 16; ++v) z[v] = z[v] > 0.0f ? z[v] : prelu * z[v];
        break;
    default: break;
    }
}

// res = act'(a) * g  (see act_bwd)
__device__ __forceinline__ void act_bwd16(int act, float prelu, const float (&a)[16], float (&g)[16])
{
    switch (act) {
    case LENV_ACT_RELU:
#pragma unroll
        for (int v = 0; v < 16; ++v) g[v] = a[v] > 0.0f ? g[v] : 0.0f;
        break;
    case LENV_ACT_LEAKYRELU:
#pragma unroll
        for (int v = 0; v < 16; ++v) g[v] = a[v] > 0.0f ? g[v] : g[v] * 0.01f;
        break;
    case LENV_ACT_TANH:
#pragma unroll
        for (int v = 0; v < 16; ++v) g[v] = g[v] * fma32(-a[v], a[v], 1.0f);
        break;
    case LENV_ACT_PRELU:
#pragma unroll
        for (int v = 0; v < 16; ++v) g[v] = a[v] > 0.0f ? g[v] : prelu * g[v];
        break;
    default: break;
    }
}

// Epilogue of one 32x32 tile.  MFMA D layout: lane l holds column j = l % 32 and, in register v, row
// 8*(v/4) + 4*(l/32) + v%4.  Every global read (bias / aux / old value) is issued before the first store; a store
// instruction writes two 128-B row pieces.  `ep` fields are wave-uniform.
template <bool G, bool full>
__device__ __forceinline__ void tile_epilogue(const f32x16 acc, int i0, int j0, int I, int J, const GemmEpi &ep, int lane)
{
    const int j = j0 + (lane & 31), ib = i0 + 4 * (lane >> 5);
    const bool jok = full || j < J;
    const int kind = ep.kind;
    const MemView<G> out(ep.out + ep.ocol);
    const int ldo = ep.ldo;
    float x[16], res[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) res[v] = acc[v];
    if (kind == EPI_ACT_BWD) {
        const MemView<true> aux(ep.aux);
#pragma unroll
        for (int v = 0; v < 16; ++v) { const int i = ib + 8 * (v >> 2) + (v & 3); x[v] = (full || (jok && i < I)) ? aux.ld(ib * ep.ldaux + j + (8 * (v >> 2) + (v & 3)) * ep.ldaux) : 0.0f; }
        act_bwd16(ep.act, ep.prelu, x, res);
    } else if (kind == EPI_ACCUM) {
#pragma unroll
        for (int v = 0; v < 16; ++v) { const int i = ib + 8 * (v >> 2) + (v & 3); x[v] = (full || (jok && i < I)) ? out.ld(ib * ldo + j + (8 * (v >> 2) + (v & 3)) * ldo) : 0.0f; }
#pragma unroll
        for (int v = 0; v < 16; ++v) res[v] = x[v] + res[v];
    } else if (kind != EPI_STORE) {
        const MemView<true> bias(ep.bias);
        const float bv = jok ? bias.ld(j) : 0.0f;
#pragma unroll
        for (int v = 0; v < 16; ++v) res[v] = res[v] + bv;
        if (kind == EPI_BIAS_ACT) act_fwd16(ep.act, ep.prelu, res);
        else if (kind == EPI_BIAS_ADD) {
            const MemView<true> aux(ep.aux);
#pragma unroll
            for (int v = 0; v < 16; ++v) { const int i = ib + 8 * (v >> 2) + (v & 3); x[v] = (full || (jok && i < I)) ? aux.ld(ib * ep.ldaux + j + (8 * (v >> 2) + (v & 3)) * ep.ldaux) : 0.0f; }
#pragma unroll
            for (int v = 0; v < 16; ++v) res[v] = x[v] + res[v];
        } else if (kind == EPI_BIAS_TANH) {
#pragma unroll
            for (int v = 0; v < 16; ++v) res[v] = det_tanhf(lenv_tanh_table, res[v]);
            if (ep.out2) {
                const MemView<true> o2(ep.out2);
#pragma unroll
                for (int v = 0; v < 16; ++v) { const int i = ib + 8 * (v >> 2) + (v & 3); if (full || (jok && i < I)) o2.st(ib * ep.ldo2 + j + (8 * (v >> 2) + (v & 3)) * ep.ldo2, res[v]); }
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) res[v] = res[v] * ep.scale;
        }
    }
    if (full) {
#pragma unroll
        for (int v = 0; v < 16; ++v) out.st(ib * ldo + j + (8 * (v >> 2) + (v & 3)) * ldo, res[v]);
    } else if (jok) {
#pragma unroll
        for (int v = 0; v < 16; ++v) { const int i = ib + 8 * (v >> 2) + (v & 3); if (i < I) out.st(ib * ldo + j + (8 * (v >> 2) + (v & 3)) * ldo, res[v]); }
    }
}

// ---- GEMM programs -------------------------------------------------------------------------------------------------
// A kernel queues the products of a phase (a forward pass, a backward chain) as commands in LDS and runs the queue with
// ONE out-of-line call: the GEMM body is instantiated once per kernel and inlined at a single site, so no per-product
// argument marshalling / callee-save traffic (measured 5-7 k cycles per product when every product was a call).
// Commands run in order, a workgroup barrier after each (the epilogue's global writes are visible to the next staging).

enum { CMD_GEMM = 0, CMD_COLSUM = 1 };

struct GemmCmd {                                           // 32 dwords, lives in LDS
    int kind;                                              // CMD_GEMM | CMD_COLSUM (out[j] = sum_i P[i*sPi + j], i < I, j < J)
    int sPi, sPr, sQj, sQr, I, J, R;
    const float *P, *Q;
    int ekind, ldo, ocol, ldaux, ldo2, act, flags;
    float prelu, scale;
    float *out, *out2;
    const float *bias, *aux;
    int pad_;
};

template <int MAXI> struct GemmShape {
    static constexpr int LDP = MAXI + 4;                   // Ps row stride (floats): r-major rows of MAXI operand rows
    static constexpr int NTW = (MAXI / 32) * 4 / DNW;      // 32x32 tiles per wave
    static constexpr int PS_FLOATS = GT_RB * LDP, QS_FLOATS = GT_RB * GT_LD;
};

template <int MAXI>
__device__ __forceinline__ void gemm_body(const GemmOp &op, const GemmEpi &ep, lfloat *Ps, lfloat *Qs, int tid, int lane, int wave)
{
    constexpr int LDP = GemmShape<MAXI>::LDP, NTW = GemmShape<MAXI>::NTW;
    constexpr int UNROLL_R = NTW >= 4 ? 1 : 8 / NTW;         // k-steps unrolled in the full-stage MFMA loop
    const int I = op.I, J = op.J, R = op.R;
    const int nbi = (I + 31) >> 5, nbj = (J + 31) >> 5, ntile = nbi * nbj;      // <= 8*NTW tiles of 32x32
    // t / nbi == (t * inv) >> 10 for t < 32, nbi <= 8
    const int inv = nbi == 1 ? 1024 : (nbi == 2 ? 512 : (nbi == 3 ? 342 : (nbi == 4 ? 256 : (nbi == 5 ? 205 : (nbi == 6 ? 171 : (nbi == 7 ? 147 : 128))))));
    bool has[NTW];
    int bi[NTW], bj[NTW];
    const lfloat *pa[NTW], *pb[NTW];
    f32x16 acc[NTW];
    // A operand: lane l supplies P[row l%32][k = l/32]; B operand: Q[col l%32][k = l/32]; both one LDS dword, r-major rows
#pragma unroll
    for (int m = 0; m < NTW; ++m) {
        const int t = wave + DNW * m;
        has[m] = t < ntile;
        bj[m] = (t * inv) >> 10; bi[m] = t - bj[m] * nbi;
        pa[m] = Ps + (lane >> 5) * LDP + (lane & 31) + 32 * bi[m];
        pb[m] = Qs + (lane >> 5) * GT_LD + (lane & 31) + 32 * bj[m];
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[m][v] = 0.0f;
    }
    for (int r0 = 0; r0 < R; r0 += GT_RB) {
        const int rb = R - r0 < GT_RB ? R - r0 : GT_RB;
        if (r0 > 0) __syncthreads();                       // previous stage fully consumed (the queue barriers cover r0 == 0)
#ifndef LENV_DIAG_SKIP_STAGE
        if (ep.flags & GEMM_GENERIC_P) stage_operand<false>(op.P, op.sPi, op.sPr, I, r0, rb, R, Ps, LDP, tid, lane, wave);
        else stage_operand<true>(op.P, op.sPi, op.sPr, I, r0, rb, R, Ps, LDP, tid, lane, wave);
        stage_operand<true>(op.Q, op.sQj, op.sQr, J, r0, rb, R, Qs, GT_LD, tid, lane, wave);
#endif
        __syncthreads();
#ifndef LENV_DIAG_SKIP_COMPUTE
        const int rb2 = rb & ~1;
        if (has[NTW - 1]) {                                // every tile slot of this wave is live: the common full-size case
            if (rb == GT_RB) {
#pragma unroll UNROLL_R
                for (int r = 0; r < GT_RB; r += 2)
#pragma unroll
                    for (int m = 0; m < NTW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[m][r * LDP], pb[m][r * GT_LD], acc[m], 0, 0, 0);
            } else {
                for (int r = 0; r < rb2; r += 2)
#pragma unroll
                    for (int m = 0; m < NTW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[m][r * LDP], pb[m][r * GT_LD], acc[m], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < NTW - 1; ++m)
                if (has[m]) {
                    if (rb == GT_RB) {
#pragma unroll 8
                        for (int r = 0; r < GT_RB; r += 2) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[m][r * LDP], pb[m][r * GT_LD], acc[m], 0, 0, 0);
                    } else {
                        for (int r = 0; r < rb2; r += 2) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[m][r * LDP], pb[m][r * GT_LD], acc[m], 0, 0, 0);
                    }
                }
        }
        if (rb & 1) {                                      // odd last k: one scalar fmaf per output
            const lfloat *prow = Ps + rb2 * LDP + 4 * (lane >> 5), *qrow = Qs + rb2 * GT_LD + (lane & 31);
#pragma unroll
            for (int m = 0; m < NTW; ++m)
                if (has[m]) {
                    const float q = qrow[32 * bj[m]];
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[m][v] = fma32(prow[32 * bi[m] + 8 * (v >> 2) + (v & 3)], q, acc[m][v]);
                }
        }
#endif
    }
#ifdef LENV_DIAG_SKIP_EPI
    if (R > 0) return;
#endif
#pragma unroll
    for (int m = 0; m < NTW; ++m) {
        if (!has[m]) continue;
        if (ep.flags & GEMM_GENERIC_OUT) tile_epilogue<false, false>(acc[m], 32 * bi[m], 32 * bj[m], I, J, ep, lane);
        else if (32 * bi[m] + 32 <= I && 32 * bj[m] + 32 <= J) tile_epilogue<true, true>(acc[m], 32 * bi[m], 32 * bj[m], I, J, ep, lane);
        else tile_epilogue<true, false>(acc[m], 32 * bi[m], 32 * bj[m], I, J, ep, lane);
    }
}

// out[j] = sum_i d[i*ld + j] (i ascending), j < n: the bias gradient of a Linear layer
__device__ __forceinline__ void wg_colsum(const float *d, int rows, int ld, int n, float *out)
{
    for (int k = (int)threadIdx.x; k < n; k += DNT) {
        float s = 0.0f;
        int b = 0;
        for (; b + 8 <= rows; b += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = d[(int64_t)(b + u) * ld + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = s + x[u];
        }
        for (; b < rows; ++b) s = s + d[(int64_t)b * ld + k];
        out[k] = s;
    }
}

template <int MAXI>
__device__ __noinline__ void gemm_run_queue(const GemmCmd *cmds_, int n_, float *Ps_, float *Qs_)
{
    typedef __attribute__((address_space(3))) const GemmCmd LCmd;
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int n = uni(n_);
    LCmd *cmds = (LCmd *)uni_ptr(cmds_);
    lfloat *Ps = (lfloat *)uni_ptr(Ps_), *Qs = (lfloat *)uni_ptr(Qs_);
    for (int k = 0; k < n; ++k) {
        LCmd &c = cmds[k];
        GemmOp op; GemmEpi ep;
        const int kind = uni(c.kind);
        op.P = uni_ptr(c.P); op.sPi = uni(c.sPi); op.sPr = uni(c.sPr); op.Q = uni_ptr(c.Q); op.sQj = uni(c.sQj); op.sQr = uni(c.sQr);
        op.I = uni(c.I); op.J = uni(c.J); op.R = uni(c.R);
        ep.kind = uni(c.ekind); ep.out = uni_ptr(c.out); ep.ldo = uni(c.ldo); ep.ocol = uni(c.ocol); ep.bias = uni_ptr(c.bias);
        ep.aux = uni_ptr(c.aux); ep.ldaux = uni(c.ldaux); ep.out2 = uni_ptr(c.out2); ep.ldo2 = uni(c.ldo2); ep.act = uni(c.act);
        ep.prelu = unif(c.prelu); ep.scale = unif(c.scale); ep.flags = uni(c.flags);
        if (kind == CMD_COLSUM) wg_colsum(op.P, op.I, op.sPi, op.J, ep.out);
        else if (op.I <= MAXI && op.J <= GT_J) gemm_body<MAXI>(op, ep, Ps, Qs, tid, lane, wave);
        else {
            // outputs larger than one MAXI x 128 block (the *_vary agents: batch / width up to 3x the configured size):
            // block by block, each with the whole reduction -- every output is still its own k-ascending chain
            const int I = op.I, J = op.J;
            const float *P0 = op.P, *Q0 = op.Q, *bias0 = ep.bias, *aux0 = ep.aux;
            float *out0 = ep.out, *out20 = ep.out2;
            for (int j0 = 0; j0 < J; j0 += GT_J)
                for (int i0 = 0; i0 < I; i0 += MAXI) {
                    if (i0 | j0) __syncthreads();
                    op.P = P0 + (int64_t)i0 * op.sPi; op.Q = Q0 + (int64_t)j0 * op.sQj;
                    op.I = I - i0 < MAXI ? I - i0 : MAXI; op.J = J - j0 < GT_J ? J - j0 : GT_J;
                    ep.out = out0 + (int64_t)i0 * ep.ldo + j0;
                    ep.bias = bias0 ? bias0 + j0 : nullptr;
                    ep.aux = aux0 ? aux0 + (int64_t)i0 * ep.ldaux + j0 : nullptr;
                    ep.out2 = out20 ? out20 + (int64_t)i0 * ep.ldo2 + j0 : nullptr;
                    gemm_body<MAXI>(op, ep, Ps, Qs, tid, lane, wave);
                }
        }
        __syncthreads();
    }
}

// ---- two idioms that keep ROCm 7.2's backend away from "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base" ----
// Where the compiler can SEE that a generic pointer was derived from an LDS array, it folds (a) the aperture test of __builtin_amdgcn_is_shared and
// (b) the null test inside a generic -> LDS address-space cast into a compare of the aperture register with a constant; under scalar-register
// pressure the backend then moves that compare to the vector unit with an operand it may not have there, and the build fails.  (a): the pointer
// goes through an empty asm first (the test stays a run-time compare of the pointer's high word); (b): lds_offset_ptr takes the low 32 bits of
// the generic address -- they ARE the LDS offset -- instead of casting.
__device__ __forceinline__ bool ptr_in_lds(const void *p)
{
#ifdef __HIP_DEVICE_COMPILE__
    uint64_t v = reinterpret_cast<uint64_t>(p);
    asm volatile("" : "+v"(v));
    return __builtin_amdgcn_is_shared(reinterpret_cast<const void *>(v));
#else
    return false;
#endif
}
__device__ __forceinline__ lfloat *lds_offset_ptr(float *p)
{
#ifdef __HIP_DEVICE_COMPILE__
    return (lfloat *)(uint32_t)(uintptr_t)uni_ptr(p);
#else
    return nullptr;
#endif
}

// Per-thread handle of the command queue (the count is uniform; thread 0 writes the records).
struct GemmQueue {
    GemmCmd *cmds;
    int n;
    __device__ __forceinline__ explicit GemmQueue(GemmCmd *c) : cmds(c), n(0) {}
    __device__ __forceinline__ void gemm(const float *P, int sPi, int sPr, const float *Q, int sQj, int sQr, int I, int J, int R, const GemmEpi &ep)
    {
        if (threadIdx.x == 0) {
            GemmCmd &c = cmds[n];                          // written field by field straight into LDS
            int flags = ep.flags;
#ifdef __HIP_DEVICE_COMPILE__
            if (ptr_in_lds(P)) flags |= GEMM_GENERIC_P;
            if (ptr_in_lds(ep.out)) flags |= GEMM_GENERIC_OUT;
#endif
            c.kind = CMD_GEMM; c.sPi = sPi; c.sPr = sPr; c.sQj = sQj; c.sQr = sQr; c.I = I; c.J = J; c.R = R; c.P = P; c.Q = Q;
            c.ekind = ep.kind; c.ldo = ep.ldo; c.ocol = ep.ocol; c.ldaux = ep.ldaux; c.ldo2 = ep.ldo2; c.act = ep.act; c.flags = flags;
            c.prelu = ep.prelu; c.scale = ep.scale; c.out = ep.out; c.out2 = ep.out2; c.bias = ep.bias; c.aux = ep.aux;
        }
        ++n;
    }
    // out[j] = sum_{i<rows} d[i*ld + j], j < cols
    __device__ __forceinline__ void colsum(const float *d, int rows, int ld, int cols, float *out)
    {
        if (threadIdx.x == 0) {
            GemmCmd &c = cmds[n];
            c.kind = CMD_COLSUM; c.P = d; c.I = rows; c.sPi = ld; c.J = cols; c.out = out;
        }
        ++n;
    }
    template <int MAXI> __device__ __forceinline__ void run(float *Ps, float *Qs)
    {
        __syncthreads();                                   // records written; every earlier global write of the workgroup visible
        gemm_run_queue<MAXI>(cmds, n, Ps, Qs);
        n = 0;
    }
};

constexpr int GEMM_QUEUE_MAX = 32;                         // commands per queue run (LDS: 32 x 128 B)

__device__ __forceinline__ GemmEpi epi_store(float *out, int ldo, int ocol = 0) { GemmEpi e{}; e.kind = EPI_STORE; e.out = out; e.ldo = ldo; e.ocol = ocol; return e; }
__device__ __forceinline__ GemmEpi epi_accum(float *out, int ldo) { GemmEpi e{}; e.kind = EPI_ACCUM; e.out = out; e.ldo = ldo; return e; }
__device__ __forceinline__ GemmEpi epi_bias(float *out, int ldo, int ocol, const float *bias) { GemmEpi e{}; e.kind = EPI_BIAS; e.out = out; e.ldo = ldo; e.ocol = ocol; e.bias = bias; return e; }
__device__ __forceinline__ GemmEpi epi_bias_act(float *out, int ldo, const float *bias, int act, float prelu) { GemmEpi e{}; e.kind = EPI_BIAS_ACT; e.out = out; e.ldo = ldo; e.bias = bias; e.act = act; e.prelu = prelu; return e; }
__device__ __forceinline__ GemmEpi epi_act_bwd(float *out, int ldo, const float *aux, int ldaux, int act, float prelu) { GemmEpi e{}; e.kind = EPI_ACT_BWD; e.out = out; e.ldo = ldo; e.aux = aux; e.ldaux = ldaux; e.act = act; e.prelu = prelu; return e; }
__device__ __forceinline__ GemmEpi epi_bias_add(float *out, int ldo, const float *bias, const float *aux, int ldaux) { GemmEpi e{}; e.kind = EPI_BIAS_ADD; e.out = out; e.ldo = ldo; e.bias = bias; e.aux = aux; e.ldaux = ldaux; return e; }
__device__ __forceinline__ GemmEpi epi_bias_act_ld(float *out, int ldo, const float *bias, int act) { return epi_bias_act(out, ldo, bias, act, 0.25f); }
__device__ __forceinline__ GemmEpi epi_bias_tanh(float *out, int ldo, int ocol, const float *bias, float scale, float *out2, int ldo2) { GemmEpi e{}; e.kind = EPI_BIAS_TANH; e.out = out; e.ldo = ldo; e.ocol = ocol; e.bias = bias; e.scale = scale; e.out2 = out2; e.ldo2 = ldo2; return e; }


// ---- elementwise passes over a chain's HBM arena: 4 elements per thread with all reads issued before the first write
// (the arrays may alias as far as the compiler knows; one element at a time costs one HBM round trip each) ----

struct AdamConsts { float neg_step, bc2_sqrt, w1, w2, beta2, eps; };

// one element of torch.optim.Adam's single-tensor step (+ Polyak update of the target value t with the new parameter)
__device__ __forceinline__ void adam_elem(float g, float &m, float &v, float &w, float &t, const AdamConsts c, float tau, float omt)
{
    m = fma32(c.w1, g - m, m);
    v = fma32(c.w2 * g, g, v * c.beta2);
    const float denom = __builtin_sqrtf(v) / c.bc2_sqrt + c.eps;
    w = w + (c.neg_step * m) / denom;
    t = tau * w + omt * t;
}

// torch.optim.Adam single-tensor step on [p0, p0+n) (+ optional Polyak update of `target` with the new parameter).
// The body moves float4 pieces (the arena arrays are 16-byte aligned; 2 pieces x 5 arrays in flight per thread: the pass
// is bound by the latency of one workgroup's loads, not by arithmetic); a misaligned head and the tail go element-wise.
__device__ __forceinline__ void wg_adam_t(float *params, float *adam_m, float *adam_v, const float *grad, int p0, int n, const AdamConsts c,
                                          float *target, float tau, float omt, int tid, int nthreads)
{
    typedef __attribute__((address_space(1))) f32x4 gf4;
    const int end = p0 + n;
    const int b0 = (p0 + 3) & ~3, b1 = end & ~3;           // float4 body [b0, b1)
    if (b1 > b0) {
        const int npieces = (b1 - b0) >> 2;
        for (int q0 = tid; q0 < npieces; q0 += 2 * nthreads) {
            f32x4 g[2], m[2], v[2], w[2], t[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = q0 + u * nthreads;
                const bool ok = q < npieces;
                const int p = b0 + 4 * (ok ? q : q0);
                g[u] = *((const gf4 *)(grad + p)); m[u] = *((const gf4 *)(adam_m + p));
                v[u] = *((const gf4 *)(adam_v + p)); w[u] = *((const gf4 *)(params + p));
                t[u] = target ? *((const gf4 *)(target + p)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float mm = m[u][k], vv = v[u][k], ww = w[u][k], tt = t[u][k];
                    adam_elem(g[u][k], mm, vv, ww, tt, c, tau, omt);
                    m[u][k] = mm; v[u][k] = vv; w[u][k] = ww; t[u][k] = tt;
                }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = q0 + u * nthreads;
                if (q < npieces) {
                    const int p = b0 + 4 * q;
                    *((gf4 *)(adam_m + p)) = m[u]; *((gf4 *)(adam_v + p)) = v[u];
                    *((gf4 *)(params + p)) = w[u];
                    if (target) *((gf4 *)(target + p)) = t[u];
                }
            }
        }
    }
    // element-wise: [p0, min(b0, end)) and [max(b1, b0), end)  (at most 3 + 3 elements, or everything when n < 4)
    const int head_end = b0 < end ? b0 : end, tail_start = b1 > b0 ? b1 : head_end;
    const int nhead = head_end - p0, nrest = nhead + (end - tail_start);
    if (tid < nrest) {
        const int p = tid < nhead ? p0 + tid : tail_start + (tid - nhead);
        float m = adam_m[p], v = adam_v[p], w = params[p], t = target ? target[p] : 0.0f;
        adam_elem(grad[p], m, v, w, t, c, tau, omt);
        adam_m[p] = m; adam_v[p] = v; params[p] = w;
        if (target) target[p] = t;
    }
}

__device__ __forceinline__ void wg_adam(float *params, float *adam_m, float *adam_v, const float *grad, int p0, int n, const AdamConsts c,
                                        float *target, float tau, float omt)
{
    wg_adam_t(params, adam_m, adam_v, grad, p0, n, c, target, tau, omt, (int)threadIdx.x, DNT);
}

// target = tau * params + (1 - tau) * target on [0, n)
__device__ __forceinline__ void wg_polyak(const float *params, float *target, int n, float tau, float omt)
{
    typedef __attribute__((address_space(1))) f32x4 gf4;
    const int n4 = n >> 2, tid = (int)threadIdx.x;
    for (int q0 = tid; q0 < n4; q0 += 4 * DNT) {
        f32x4 w[4], t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int q = q0 + u * DNT; const int p = 4 * (q < n4 ? q : q0); w[u] = *((const gf4 *)(params + p)); t[u] = *((const gf4 *)(target + p)); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * DNT;
            if (q < n4) {
                f32x4 r;
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = tau * w[u][k] + omt * t[u][k];
                *((gf4 *)(target + 4 * q)) = r;
            }
        }
    }
    const int p = 4 * n4 + tid;
    if (p < n) target[p] = tau * params[p] + omt * target[p];
}

}  // namespace lenv
