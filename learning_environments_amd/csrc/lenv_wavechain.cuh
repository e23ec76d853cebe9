// lenv_wavechain.cuh -- "wave-chain" building blocks for the 128-wide MLPs of the big-net agents (DuelingDDQN / TD3) on gfx950.
//
// The GEMM-queue routine (lenv_gemm.cuh) treats every Linear of every pass as an independent product: operands staged from the
// HBM arena through LDS, outputs written back, five barriers each -- about 11 k cycles of fixed cost per product and 64 KB of
// weights re-staged for every pass that shares them.  Here the unit of work is a BLOCK OF 32 SAMPLES OWNED BY ONE WAVE:
//
//   * the product is computed transposed, out^T[unit][sample] = sum_k W[unit][k] * in^T[k][sample], on v_mfma_f32_32x32x2_f32
//     (bit for bit a k-ascending fmaf chain = the canonical order of oracle/lenv_oracle.h): the A operand is the weight matrix,
//     read from an LDS image that ALL waves share (staged once per layer and pass), the B operand is the wave's own activation
//     block held in 64 VGPRs;
//   * the 32x32 result tiles (lane = sample, register = unit) become the next layer's B operand with 32 v_permlane32_swap --
//     activations never leave the register file between the layers of a pass; what the backward pass needs is dumped to the
//     HBM arena in register order (16-byte stores, 1 KB per wave instruction) and read back by the same lanes;
//   * weight gradients (a reduction over the samples) read two LDS images, [sample][unit] of the upstream gradient and
//     [sample][unit] of the layer input, written from the register tiles with 16-byte stores;
//   * thin products (the 1-row greedy action and the T-row lock-step test forward) take their A operand straight from the
//     K-major weight arrays in the arena (coalesced, each weight read once) and exchange activations through a 16 KB image.
//
// LDS images are 128 x 128 floats without padding; element (row r, col c) sits at r*128 + (c ^ 4*(r & 7)): the XOR keeps
// 16-byte groups intact, makes the tile -> image stores (8 lanes = 8 rows per LDS cycle) and the transposing weight stores
// conflict-free, and leaves the MFMA operand reads (32 consecutive columns of one row) a permutation of the 32 banks.
#pragma once

#include "lenv_gemm.cuh"

namespace lenv {
namespace wc {

constexpr int NT = 512, NW = 8;           // threads / waves per chain
constexpr int W = 128;                    // units per layer = rows and columns of an image
constexpr int IMG = W * W;                // floats per image (64 KB)
constexpr int BLK = 32 * W;               // floats of one wave's register dump of a [32 samples x 128 units] block

typedef __attribute__((address_space(1))) f32x4 gf4;
typedef __attribute__((address_space(3))) f32x4 lf4;

// Workgroup barrier for hand-offs through LDS only.  __syncthreads() also drains the wave's global loads AND stores (s_waitcnt
// vmcnt(0)): behind a burst of dump stores that is several thousand cycles of write-acknowledge latency which nothing in the next
// phase depends on.  Use it only where every datum the next phase reads from another wave went through LDS; global data written
// before it must pass a later __syncthreads() before another wave reads it.
__device__ __forceinline__ void barrier_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- team barrier: a chain that runs on G co-resident workgroups (td3_wavechain.hip, dueling_wavechain.hip).  One monotonically
// increasing counter per chain (`bar`, zeroed by a reset kernel in front of the launch); every wave waits for the acknowledgement of
// its own arena stores (__syncthreads() alone is s_barrier without a vmcnt wait on gfx950), then thread 0 releases, arrives, waits
// for the epoch's count and acquires.  When all members sit on ONE XCD (same_xcd, established once by the kernels) they share its
// L2 and the vector L1 writes through: the release needs no L2 write-back and the acquire only drops this CU's L1 lines; otherwise
// the agent-scope fences run.  The launch is only made when all its workgroups fit the device at once (host: lenv_team_grid_resident),
// but a foreign kernel may hold CUs: a member that waits longer than LENV_TEAM_GIVEUP_TICKS gives up for good and raises the LAUNCH's
// give-up word (`launch_dead`: one per launch, polled by every waiting member), so the whole launch drains within that time with
// status -10 and the caller repeats it with one workgroup per chain.  Returns through ts.dead (uniform in the workgroup).
struct TeamSync {
    unsigned *bar, *launch_dead;
    // two LDS words, zero at kernel start: [0] the give-up flag, [1] the barrier epoch of team_barrier<true>.  The TD3 kernel keeps both in
    // LDS rather than in this record: a kernel-lifetime register of a kernel that calls out-of-line routines can end up in scratch memory,
    // and there the barrier started with a memory round trip to fetch its own counter (configs[4] shard: 493 -> 481 ms).  The DuelingDDQN
    // kernel, whose body spills next to nothing, is 2 % FASTER with the counter in `epoch` (measured both ways): team_barrier<false>.
    volatile __attribute__((address_space(3))) int *lds_flag;
    unsigned epoch;
    int G;
    bool dead, same_xcd;
};
__device__ __forceinline__ bool team_is_dead(const TeamSync &ts) { return ts.lds_flag[0] != 0; }
template <bool LDS_EPOCH>
__device__ __forceinline__ void team_barrier(TeamSync &ts, int tid)
{
    if (ts.G == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!LDS_EPOCH) ++ts.epoch;
    if (tid == 0 && (LDS_EPOCH ? ts.lds_flag[0] == 0 : !ts.dead)) {
        unsigned epoch = ts.epoch;
        if (LDS_EPOCH) { epoch = (unsigned)(ts.lds_flag[1] + 1); ts.lds_flag[1] = (int)epoch; }
        if (ts.same_xcd) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ts.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * (unsigned)ts.G;
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz
        unsigned spins = 0;
        while (__hip_atomic_load(ts.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0u) {          // every 64th poll: has anybody in the launch given up / is it this member's turn to?
                if (__hip_atomic_load(ts.launch_dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { *ts.lds_flag = 1; break; }
                if (__builtin_amdgcn_s_memrealtime() - w0 > (epoch <= 1u ? LENV_TEAM_GIVEUP_TICKS : LENV_TEAM_GIVEUP_TICKS_RUN)) {     // (epoch 1 = the team assembles)
                    __hip_atomic_store(ts.launch_dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (system scope: past every XCD's L2)
                    *ts.lds_flag = 1;
                    break;
                }
            }
        }
        // acquire: drop this CU's L1 lines.  The invalidate must have COMPLETED before the workgroup barrier below lets the other waves
        // load (a buffer_inv only orders the issuing wave's own later loads; its completion is counted by vmcnt)
        if (ts.same_xcd) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        else { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
    if (!LDS_EPOCH) { if (*ts.lds_flag) ts.dead = true; }
}

__device__ __forceinline__ int img_pos(int r, int c) { return r * W + (c ^ ((r & 7) << 2)); }

struct Lane {
    int tid, lane, wave, li, h;
    int colsw[4];                         // swizzled column (li) of rows 2t+h with (t & 3) = q
    __device__ __forceinline__ void init()
    {
        tid = (int)threadIdx.x; lane = tid & 63; wave = uni(tid >> 6); li = lane & 31; h = lane >> 5;
#pragma unroll
        for (int q = 0; q < 4; ++q) colsw[q] = li ^ ((2 * q + h) << 2);
    }
    // Make the lane coordinates opaque to the optimiser at this point.  Every per-lane address of a phase (swizzled image slots,
    // staging pieces, tile rows) is a function of them and of compile-time constants; without the cut LLVM treats the whole set as
    // loop invariants of the enclosing layer loop, hoists it in front of the loop and spills it (100+ registers), where
    // recomputing an address costs one or two VALU instructions next to its use.
    __device__ __forceinline__ void refresh()
    {
        asm volatile("" : "+v"(tid), "+v"(lane), "+v"(li), "+v"(h));
#pragma unroll
        for (int q = 0; q < 4; ++q) colsw[q] = li ^ ((2 * q + h) << 2);
    }
};

// register index of the B operand of k-step t (k = 2t, 2t+1) inside a [4 tiles x 16] register block after tile_to_operand
__device__ __forceinline__ constexpr int breg_of(int t) { return (t & ~3) | ((t & 1) << 1) | ((t >> 1) & 1); }

// D layout of a 32x32 tile: lane (sample li, half h), register v -> unit 8*(v/4) + 4*h + v%4.  After the swaps register
// 4g+{0,2,1,3} holds, in its lower / upper lane half, units (8g, 8g+1), (8g+2, 8g+3), (8g+4, 8g+5), (8g+6, 8g+7): the B operand
// of four consecutive k-steps (see breg_of).
__device__ __forceinline__ void tile_to_operand(float (&r)[64])
{
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const int a = 2 * p, b = 2 * p + 1;                    // (4g, 4g+1) and (4g+2, 4g+3)
        auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[a]), __float_as_uint(r[b]), false, false);
        r[a] = __uint_as_float(s[0]); r[b] = __uint_as_float(s[1]);
    }
}

// acc[jt] += sum over 64 k-steps: A = image rows 2t+h (all 128 columns = 4 tiles), B = the wave's operand registers
__device__ __forceinline__ void chain128(const float *img_, const Lane &L, const float (&b)[64], f32x16 (&acc)[4])
{
    const lfloat *img = (const lfloat *)img_;
    const lfloat *ab[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ab[q] = img + L.h * W + L.colsw[q];
#pragma unroll
    for (int t = 0; t < 64; ++t) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
            acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[t & 3][2 * t * W + 32 * jt], b[breg_of(t)], acc[jt], 0, 0, 0);
    }
}

__device__ __forceinline__ void acc_zero(f32x16 (&acc)[4])
{
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[jt][v] = 0.0f;
}

// r[16 jt + 4 g + c] = acc[jt][4 g + c] + bias[32 jt + 8 g + 4 h + c]  (bias: 128 floats in LDS), optional ReLU-family activation
template <int ACT>
__device__ __forceinline__ void tile_bias_act(const f32x16 (&acc)[4], const float *bias_, const Lane &L, float prelu, float (&r)[64])
{
    const lfloat *bias = (const lfloat *)bias_;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *(const lf4 *)(bias + 32 * jt + 8 * g + 4 * L.h);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float z = acc[jt][4 * g + c] + bv[c];
                if (ACT == LENV_ACT_RELU) z = z > 0.0f ? z : 0.0f;
                else if (ACT == LENV_ACT_LEAKYRELU) z = z > 0.0f ? z : z * 0.01f;
                else if (ACT == LENV_ACT_PRELU) z = z > 0.0f ? z : prelu * z;
                else if (ACT == LENV_ACT_TANH) z = det_tanhf(lenv_tanh_table, z);
                r[16 * jt + 4 * g + c] = z;
            }
        }
}

// register dump of a block: piece (jt, g) of lane l at dump[((4 jt + g) * 64 + l) * 4 .. +3]
__device__ __forceinline__ void dump_store(float *dump, const Lane &L, const float (&r)[64])
{
    gf4 *d = (gf4 *)dump + L.lane;
#pragma unroll
    for (int p = 0; p < 16; ++p) d[p * 64] = f32x4{r[4 * p], r[4 * p + 1], r[4 * p + 2], r[4 * p + 3]};
}
__device__ __forceinline__ void dump_load(const float *dump, const Lane &L, float (&r)[64])
{
    const gf4 *d = (const gf4 *)dump + L.lane;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const f32x4 v = d[p * 64];
        r[4 * p] = v[0]; r[4 * p + 1] = v[1]; r[4 * p + 2] = v[2]; r[4 * p + 3] = v[3];
    }
}
// element (sample i of the block, unit u) of a dump: for the few consumers that walk a dump unit-wise (head weight gradients)
__device__ __forceinline__ int dump_index(int i, int u) { return ((((u >> 5) * 4 + ((u >> 3) & 3)) * 64 + i + 32 * ((u >> 2) & 1)) << 2) + (u & 3); }

// [sample][unit] image rows 32 blk .. 32 blk + 31 from a register block (D layout): 16 conflict-free 16-byte stores
__device__ __forceinline__ void tile_to_image(float *img_, int blk, const Lane &L, const float (&r)[64])
{
    lfloat *img = (lfloat *)img_;
    const int row = 32 * blk + L.li;
    lfloat *base = img + row * W;
    const int sw = (row & 7) << 2;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = (32 * jt + 8 * g + 4 * L.h) ^ sw;
            *(lf4 *)(base + col) = f32x4{r[16 * jt + 4 * g], r[16 * jt + 4 * g + 1], r[16 * jt + 4 * g + 2], r[16 * jt + 4 * g + 3]};
        }
}

// ---- split passes (a block of 32 samples on FOUR waves, wave jt = output tile jt of every layer; the quad exchanges the operand
// registers through a 16 KB LDS buffer: 64 k-step registers x 64 lanes) ----
__device__ __forceinline__ void tile16_to_operand(float (&r)[16])
{
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        auto s_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[2 * p]), __float_as_uint(r[2 * p + 1]), false, false);
        r[2 * p] = __uint_as_float(s_[0]); r[2 * p + 1] = __uint_as_float(s_[1]);
    }
}
__device__ __forceinline__ void xch_put(float *xch_, int jt, int lane, const float (&r)[16])
{
    lfloat *x = (lfloat *)xch_ + (16 * jt) * 64 + lane;
#pragma unroll
    for (int v = 0; v < 16; ++v) x[v * 64] = r[v];
}
__device__ __forceinline__ void xch_get(const float *xch_, int lane, float (&r)[64])
{
    const lfloat *x = (const lfloat *)xch_ + lane;
#pragma unroll
    for (int v = 0; v < 64; ++v) r[v] = x[v * 64];
}
// one 32-unit tile of a 128 -> 128 layer: acc += sum over 64 k-steps, A = image rows 2t+h, columns of tile jt
__device__ __forceinline__ void chain_tile(const float *img_, int jt, const Lane &L, const float (&b)[64], f32x16 &acc)
{
    const lfloat *img = (const lfloat *)img_;
    const lfloat *ab[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ab[q] = img + L.h * W + 32 * jt + L.colsw[q];
#pragma unroll
    for (int t = 0; t < 64; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[t & 3][2 * t * W], b[breg_of(t)], acc, 0, 0, 0);
}

// ---- weight images.  The arena keeps every 128x128 matrix K-MAJOR: Wt[k][unit] (so the forward image is a straight copy and the
// thin products read it coalesced); the input-gradient products reduce over the units and need image[unit][k] = the transpose.
struct StageRegs { f32x4 v[8]; };

// straight copy: piece p = tid + 512 u -> row p >> 5, float4 column p & 31
__device__ __forceinline__ void stage_load_direct(const float *Wt, const Lane &L, StageRegs &s)
{
    const gf4 *src = (const gf4 *)Wt + L.tid;
#pragma unroll
    for (int u = 0; u < 8; ++u) s.v[u] = src[u * NT];
}
__device__ __forceinline__ void stage_store_direct(float *img_, const Lane &L, const StageRegs &s)
{
    lfloat *img = (lfloat *)img_;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = L.tid + u * NT, r = p >> 5, c = (p & 31) << 2;
        *(lf4 *)(img + r * W + (c ^ ((r & 7) << 2))) = s.v[u];
    }
}
// transposing copy: a wave instruction takes 16 rows k x 4 float4 columns (units 4 c4 .. 4 c4 + 3) of Wt and writes image rows
// (units) 4 c4 + cc, column k: 2-way bank conflicts at most (free for ds_write_b32)
__device__ __forceinline__ void stage_load_transposed(const float *Wt, const Lane &L, StageRegs &s)
{
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        // wave w, instruction u: k block = (w & 7), c4 block = u  -> k = 16 w + lane / 4, c4 = 4 u + lane % 4
        const int k = 16 * L.wave + (L.lane >> 2), c4 = 4 * u + (L.lane & 3);
        s.v[u] = *((const gf4 *)Wt + k * 32 + c4);
    }
}
__device__ __forceinline__ void stage_store_transposed(float *img_, const Lane &L, const StageRegs &s)
{
    lfloat *img = (lfloat *)img_;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = 16 * L.wave + (L.lane >> 2), c4 = 4 * u + (L.lane & 3);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int r = 4 * c4 + cc;
            img[r * W + (k ^ ((r & 7) << 2))] = s.v[u][cc];
        }
    }
}

// out[j] = sum_i image[i][j] (i ascending, plain adds): the bias gradient of a layer from the [sample][unit] image of its
// upstream gradient; one thread per unit
__device__ __forceinline__ float image_colsum(const float *img_, int j, int rows)
{
    const lfloat *img = (const lfloat *)img_;
    float s = 0.0f;
    for (int i0 = 0; i0 < rows; i0 += 32) {                // rows is a multiple of 32: 32 reads in flight, then the ordered adds
        float x[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) x[u] = img[(i0 + u) * W + (j ^ ((u & 7) << 2))];      // (i0 + u) & 7 == u & 7
#pragma unroll
        for (int u = 0; u < 32; ++u) s = s + x[u];
    }
    return s;
}

// Weight gradient of a 128 -> 128 layer: gWt[k][j] = sum_i in[i][k] * dz[i][j] (i ascending over `rows` samples, rows even).
// A = image of the layer input (rows i, columns k), B = image of the upstream gradient (rows i, columns j); wave w owns the
// tiles (kt = w / 2, jt = 2 (w % 2) + {0, 1}).  Output rows are 128-byte segments of the K-major gradient array.
__device__ __forceinline__ void wgrad_tiles(const float *img_in_, const float *img_dz_, int rows, const Lane &L, float *gWt)
{
    const lfloat *img_in = (const lfloat *)img_in_, *img_dz = (const lfloat *)img_dz_;
    const int kt = L.wave >> 1, jt0 = (L.wave & 1) << 1;
    f32x16 acc0, acc1;
#pragma unroll
    for (int v = 0; v < 16; ++v) { acc0[v] = 0.0f; acc1[v] = 0.0f; }
    const lfloat *pa[4], *pb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { pa[q] = img_in + L.h * W + 32 * kt + L.colsw[q]; pb[q] = img_dz + L.h * W + 32 * jt0 + L.colsw[q]; }
    const int steps = rows >> 1;                               // rows is a multiple of 8 here (whole sample blocks)
#pragma unroll 2
    for (int t4 = 0; t4 < steps; t4 += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = pa[q][2 * (t4 + q) * W];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[q][2 * (t4 + q) * W], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[q][2 * (t4 + q) * W + 32], acc1, 0, 0, 0);
        }
    }
    gfloat *out = (gfloat *)gWt + (32 * kt + 4 * L.h) * W + 32 * jt0 + L.li;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int r = 8 * (v >> 2) + (v & 3);
        out[r * W] = acc0[v];
        out[r * W + 32] = acc1[v];
    }
}

// ---- optimizer step as the EPILOGUE of a 32 x 32 gradient tile (shared by the TD3 and DuelingDDQN team paths) ----
// Adam + Polyak on a 32 x 32 tile of a K-major array whose gradient sits in `acc` (D layout: lane = column, register = row
// 8 (v / 4) + 4 h + v % 4): transposed through the wave's LDS tile so that every arena access is 16 bytes per lane.  Rows >= kmax are
// left alone (W1t's zero rows).  wu != null: the updated tile also goes to the unit-major copy (element (k, j) at wu[j * W + k]).
// The tile's state (parameter, target, Adam m / v: 16 float4 per lane) is loaded by t3v_tile_state BEFORE the gradient chain runs.
struct T3vTileState { f32x4 w[4], m[4], v[4], t[4]; };
__device__ __forceinline__ void t3v_tile_state(const Lane &L, const float *w_, const float *m_, const float *v_, const float *t_, int kmax, T3vTileState &st)
{
    const int kr = L.lane >> 3, j4 = (L.lane & 7) << 2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int k = kr + 8 * p, off = (k < kmax ? k : 0) * W + j4;       // (rows >= kmax: a harmless in-range load, never written back)
        st.w[p] = *(const gf4 *)((const gfloat *)w_ + off); st.m[p] = *(const gf4 *)((const gfloat *)m_ + off);
        st.v[p] = *(const gf4 *)((const gfloat *)v_ + off); st.t[p] = *(const gf4 *)((const gfloat *)t_ + off);
    }
}
__device__ __forceinline__ void t3v_tile_adam(const f32x16 &acc, T3vTileState &st, float *tile_, float *tileT_, const Lane &L, float *w_, float *m_, float *v_, float *t_,
                                              int kmax, float *wu_, const AdamConsts ac, float tau, float omt)
{
    lfloat *tile = (lfloat *)tile_;
#pragma unroll
    for (int v = 0; v < 16; ++v) tile[(8 * (v >> 2) + 4 * L.h + (v & 3)) * 32 + L.li] = acc[v];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int kr = L.lane >> 3, j4 = (L.lane & 7) << 2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int k = kr + 8 * p, off = k * W + j4;
        const f32x4 g = *(const lf4 *)(tile + k * 32 + j4);
        if (k < kmax) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float mm = st.m[p][c], v2 = st.v[p][c], ww = st.w[p][c], tt = st.t[p][c];
                adam_elem(g[c], mm, v2, ww, tt, ac, tau, omt);
                st.m[p][c] = mm; st.v[p][c] = v2; st.w[p][c] = ww; st.t[p][c] = tt;
            }
            *(gf4 *)((gfloat *)m_ + off) = st.m[p]; *(gf4 *)((gfloat *)v_ + off) = st.v[p];
            *(gf4 *)((gfloat *)w_ + off) = st.w[p]; *(gf4 *)((gfloat *)t_ + off) = st.t[p];
        }
    }
    if (wu_) {
        lfloat *tt = (lfloat *)tileT_;                     // [j][33]: the updated tile, transposed
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) tt[(j4 + c) * 33 + kr + 8 * p] = st.w[p][c];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int j = kr + 8 * p;                      // (kr, j4) now address (row j, columns k = j4 .. j4 + 3)
            const f32x4 o = { tt[j * 33 + j4], tt[j * 33 + j4 + 1], tt[j * 33 + j4 + 2], tt[j * 33 + j4 + 3] };
            *(gf4 *)((gfloat *)wu_ + j * W + j4) = o;
        }
    }
}

// the same in two halves: the four state words requested before the gradient is computed, the step once it is there (a job whose
// gradient is a long reduction does not pay the state's memory round trip behind it)
struct T3vAdam1State { float w, m, v, t; };
__device__ __forceinline__ T3vAdam1State t3v_adam1_load(const float *w_, const float *m_, const float *v_, const float *t_, int off)
{
    return T3vAdam1State{ ((const gfloat *)w_)[off], ((const gfloat *)m_)[off], ((const gfloat *)v_)[off], ((const gfloat *)t_)[off] };
}
__device__ __forceinline__ void t3v_adam1_step(float g, T3vAdam1State st, float *w_, float *m_, float *v_, float *t_, int off, const AdamConsts ac, float tau, float omt)
{
    adam_elem(g, st.m, st.v, st.w, st.t, ac, tau, omt);
    ((gfloat *)m_)[off] = st.m; ((gfloat *)v_)[off] = st.v; ((gfloat *)w_)[off] = st.w; ((gfloat *)t_)[off] = st.t;
}

// one parameter of a small vector / matrix: Adam + Polyak in place (4-byte accesses: a few hundred elements per network)
__device__ __forceinline__ void t3v_adam1(float g, float *w_, float *m_, float *v_, float *t_, int off, const AdamConsts ac, float tau, float omt)
{
    float w = ((gfloat *)w_)[off], m = ((gfloat *)m_)[off], v = ((gfloat *)v_)[off], t = ((gfloat *)t_)[off];
    adam_elem(g, m, v, w, t, ac, tau, omt);
    ((gfloat *)m_)[off] = m; ((gfloat *)v_)[off] = v; ((gfloat *)w_)[off] = w; ((gfloat *)t_)[off] = t;
}


// Weight gradient over more samples than one image holds (batch 192 = two half-batches of 96 rows): the accumulators live across the
// calls, the reduction stays one i-ascending chain.
__device__ __forceinline__ void wgrad_accum(const float *img_in_, const float *img_dz_, int rows, const Lane &L, f32x16 &acc0, f32x16 &acc1)
{
    const lfloat *img_in = (const lfloat *)img_in_, *img_dz = (const lfloat *)img_dz_;
    const int kt = L.wave >> 1, jt0 = (L.wave & 1) << 1;
    const lfloat *pa[4], *pb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { pa[q] = img_in + L.h * W + 32 * kt + L.colsw[q]; pb[q] = img_dz + L.h * W + 32 * jt0 + L.colsw[q]; }
    const int steps = rows >> 1;
#pragma unroll 2
    for (int t4 = 0; t4 < steps; t4 += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = pa[q][2 * (t4 + q) * W];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[q][2 * (t4 + q) * W], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[q][2 * (t4 + q) * W + 32], acc1, 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void wgrad_store(const Lane &L, float *gWt, const f32x16 &acc0, const f32x16 &acc1)
{
    const int kt = L.wave >> 1, jt0 = (L.wave & 1) << 1;
    gfloat *out = (gfloat *)gWt + (32 * kt + 4 * L.h) * W + 32 * jt0 + L.li;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int r = 8 * (v >> 2) + (v & 3);
        out[r * W] = acc0[v];
        out[r * W + 32] = acc1[v];
    }
}

// The same with torch.optim.Adam's step + the Polyak update of the target net applied to the tile right away (element-wise:
// adam_elem is the routine wg_adam uses): the four state arrays of the tile are loaded BEFORE the MFMA loop, so their latency
// hides behind it, and the gradient never makes the round trip through the arena.
__device__ __forceinline__ void wgrad_tiles_adam(const float *img_in_, const float *img_dz_, int rows, const Lane &L, float *Wt, float *Mt, float *Vt,
                                                 float *Tt, const AdamConsts c, float tau, float omt)
{
    const lfloat *img_in = (const lfloat *)img_in_, *img_dz = (const lfloat *)img_dz_;
    const int kt = L.wave >> 1, jt0 = (L.wave & 1) << 1;
    // uniform (SGPR) array bases + ONE 32-bit per-lane offset: the 128 element addresses of a lane differ by compile-time constants
    const MemView<true> pw(Wt), pm(Mt), pv(Vt), pt(Tt);
    const int off = (32 * kt + 4 * L.h) * W + 32 * jt0 + L.li;
    // tile 0's state is loaded before the MFMA loop (latency hidden behind it), tile 1's while tile 0 is being updated
    float w0[16], m0[16], v0[16], t0[16], w1[16], m1[16], v1[16], t1[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int o = off + (8 * (e >> 2) + (e & 3)) * W;
        w0[e] = pw.ld(o); m0[e] = pm.ld(o); v0[e] = pv.ld(o); t0[e] = pt.ld(o);
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int x = 0; x < 16; ++x) { acc0[x] = 0.0f; acc1[x] = 0.0f; }
    const lfloat *pa[4], *pb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { pa[q] = img_in + L.h * W + 32 * kt + L.colsw[q]; pb[q] = img_dz + L.h * W + 32 * jt0 + L.colsw[q]; }
    const int steps = rows >> 1;
#pragma unroll 2
    for (int t4 = 0; t4 < steps; t4 += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = pa[q][2 * (t4 + q) * W];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[q][2 * (t4 + q) * W], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[q][2 * (t4 + q) * W + 32], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int o = off + (8 * (e >> 2) + (e & 3)) * W + 32;
        w1[e] = pw.ld(o); m1[e] = pm.ld(o); v1[e] = pv.ld(o); t1[e] = pt.ld(o);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int o = off + (8 * (e >> 2) + (e & 3)) * W;
        adam_elem(acc0[e], m0[e], v0[e], w0[e], t0[e], c, tau, omt);
        pm.st(o, m0[e]); pv.st(o, v0[e]); pw.st(o, w0[e]); pt.st(o, t0[e]);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int o = off + (8 * (e >> 2) + (e & 3)) * W + 32;
        adam_elem(acc1[e], m1[e], v1[e], w1[e], t1[e], c, tau, omt);
        pm.st(o, m1[e]); pv.st(o, v1[e]); pw.st(o, w1[e]); pt.st(o, t1[e]);
    }
}

// ---- thin products: I <= 32 samples, activations in [unit][32] images (hT[k * 32 + sample]) ----
// wave w < 4 computes units 32 w .. 32 w + 31 of out^T = act(Wt^T . in^T + bias): A straight from the K-major array in the arena
// (lane (unit li, half h) reads Wt[(2t+h) * 128 + 32 w + li]: two 128-byte segments per instruction, every weight read once)
template <int ACT>
__device__ __forceinline__ void thin_layer(const float *Wt, const float *bias, const float *in_img_, float *out_img_, int jt, const Lane &L, float prelu)
{
    const lfloat *in_img = (const lfloat *)in_img_;
    lfloat *out_img = (lfloat *)out_img_;
    const gfloat *wa = (const gfloat *)Wt + L.h * W + 32 * jt + L.li;
    const lfloat *xb = in_img + L.h * 32 + L.li;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll 16
    for (int t = 0; t < 64; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * t * W], xb[2 * t * 32], acc, 0, 0, 0);
    const gfloat *bg = (const gfloat *)bias + 32 * jt + 4 * L.h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *(const gf4 *)(bg + 8 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float z = acc[4 * g + c] + bv[c];
            if (ACT == LENV_ACT_RELU) z = z > 0.0f ? z : 0.0f;
            else if (ACT == LENV_ACT_LEAKYRELU) z = z > 0.0f ? z : z * 0.01f;
            else if (ACT == LENV_ACT_PRELU) z = z > 0.0f ? z : prelu * z;
            else if (ACT == LENV_ACT_TANH) z = det_tanhf(lenv_tanh_table, z);
            out_img[(32 * jt + 8 * g + 4 * L.h + c) * 32 + L.li] = z;
        }
    }
}

// The same for I <= 16 samples on v_mfma_f32_16x16x4_f32 (also a k-ascending fmaf chain, four k per instruction; 32 cycles per
// instruction instead of 64 for a tile that wastes no sample columns): activations in [unit][16] images, wave w computes units
// 16 w .. 16 w + 15.  NOUT = 2: two layers that share the input (the value and advantage streams), interleaved accumulators.
// A operand of unit tile ut for all 32 k-steps: 32 registers, loaded long before the product needs them (the weights do not depend
// on the activations: a thin forward issues the loads of ALL its layers up front and runs the chains back to back)
__device__ __forceinline__ void thin_load16(const float *Wt, int ut, const Lane &L, float (&a)[32])
{
    const gfloat *wa = (const gfloat *)Wt + (L.lane >> 4) * W + 16 * ut + (L.lane & 15);
#pragma unroll
    for (int t = 0; t < 32; ++t) a[t] = wa[4 * t * W];
}
// (thin_bias16: the four bias values of this lane's output rows -- a loop over many forwards of the same net loads them once)
__device__ __forceinline__ f32x4 thin_bias16(const float *bias, int ut, const Lane &L) { return *(const gf4 *)((const gfloat *)bias + 16 * ut + 4 * (L.lane >> 4)); }
template <int ACT, int NOUT>
__device__ __forceinline__ void thin_layer16v(const float (&a0)[32], const f32x4 bv0, float *out0_, const float (&a1)[32], const f32x4 bv1, float *out1_,
                                              const float *in_img_, int ut, const Lane &L, float prelu)
{
    const lfloat *in_img = (const lfloat *)in_img_;
    const int l16 = L.lane & 15, q = L.lane >> 4;
    const lfloat *xb = in_img + q * 16 + l16;
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const float x = xb[4 * t * 16];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], x, acc0, 0, 0, 0);
        if (NOUT == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], x, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const f32x4 acc = o == 0 ? acc0 : acc1;
        const f32x4 bv = o == 0 ? bv0 : bv1;
        lfloat *out = (lfloat *)(o == 0 ? out0_ : out1_) + (16 * ut + 4 * q) * 16 + l16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float z = acc[c] + bv[c];
            if (ACT == LENV_ACT_RELU) z = z > 0.0f ? z : 0.0f;
            else if (ACT == LENV_ACT_LEAKYRELU) z = z > 0.0f ? z : z * 0.01f;
            else if (ACT == LENV_ACT_PRELU) z = z > 0.0f ? z : prelu * z;
            else if (ACT == LENV_ACT_TANH) z = det_tanhf(lenv_tanh_table, z);
            out[c * 16] = z;
        }
    }
}
template <int ACT, int NOUT>
__device__ __forceinline__ void thin_layer16(const float (&a0)[32], const float *bias0, float *out0_, const float (&a1)[32], const float *bias1, float *out1_,
                                             const float *in_img_, int ut, const Lane &L, float prelu)
{
    const lfloat *in_img = (const lfloat *)in_img_;
    const int l16 = L.lane & 15, q = L.lane >> 4;
    const lfloat *xb = in_img + q * 16 + l16;
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const float x = xb[4 * t * 16];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], x, acc0, 0, 0, 0);
        if (NOUT == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], x, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const f32x4 acc = o == 0 ? acc0 : acc1;
        const f32x4 bv = *(const gf4 *)((const gfloat *)(o == 0 ? bias0 : bias1) + 16 * ut + 4 * q);
        lfloat *out = (lfloat *)(o == 0 ? out0_ : out1_) + (16 * ut + 4 * q) * 16 + l16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float z = acc[c] + bv[c];
            if (ACT == LENV_ACT_RELU) z = z > 0.0f ? z : 0.0f;
            else if (ACT == LENV_ACT_LEAKYRELU) z = z > 0.0f ? z : z * 0.01f;
            else if (ACT == LENV_ACT_PRELU) z = z > 0.0f ? z : prelu * z;
            else if (ACT == LENV_ACT_TANH) z = det_tanhf(lenv_tanh_table, z);
            out[c * 16] = z;
        }
    }
}

}  // namespace wc
}  // namespace lenv
