// qnet_td.hip -- DDQN TD forward over replay minibatches (K3 + K4-forward of SURVEY.md §2a) for gfx950.
//
// Replaces the forward half of DDQN.learn (agents/DDQN.py:63-85) with ReplayBuffer.sample's gather
// (utils.py:34-45) fused in: workgroup = (chain, 256-sample tile); the chain's online and target Critic_DQN
// parameters are staged in LDS as per-hidden-unit records [W1[j,:], b1[j], W2[:,j]] so every lane reads them
// as broadcasts; thread = sample runs the three forwards (online(s), online(s'), target(s')) with the hidden
// loop in registers and writes q(s)[a] and y = r + gamma*Q_target(s')[argmax Q(s')]*(1-done).
// hidden_layer == 1 only (all BASELINE DDQN configs); canonical arithmetic order of oracle/lenv_oracle.h.
#include "lenv_device.cuh"

namespace lenv {

constexpr int TD_NT = 256;
constexpr int TD_MAX_IN = 8, TD_MAX_OUT = 4;

struct TdArgs {
    lenv_mlp_desc q;
    int64_t P;
    const float *online, *target, *replay; int64_t replay_cap; int32_t row_stride;
    const int32_t *idx; int32_t batch; float gamma32;
    float *q_sa, *y;
    int RP;
};

__global__ __launch_bounds__(TD_NT) void qnet_td_kernel(const TdArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const int64_t chain = blockIdx.y;
    const int S = a.q.in_dim, A = a.q.out_dim, H = a.q.hidden, RP = a.RP;
    float *onl = lds, *tgt = lds + H * RP + ((A + 3) & ~3);
    // stage both nets as packed records
    for (int p = tid; p < a.P; p += TD_NT) {
        int off, r = p;
        if (r < H * S) { int j = r / S; off = j * RP + (r - j * S); }
        else if ((r -= H * S) < H) off = r * RP + S;
        else if ((r -= H) < A * H) { int aa = r / H; off = (r - aa * H) * RP + S + 1 + aa; }
        else off = H * RP + (r - A * H);
        onl[off] = a.online[chain * a.P + p];
        tgt[off] = a.target[chain * a.P + p];
    }
    __syncthreads();
    const int b = blockIdx.x * TD_NT + tid;
    if (b >= a.batch) return;
    const int64_t id = a.idx[chain * a.batch + b];
    const float *row = a.replay + (chain * a.replay_cap + id) * a.row_stride;
    float s[TD_MAX_IN], s2[TD_MAX_IN];
    for (int i = 0; i < S; ++i) { s[i] = row[i]; s2[i] = row[S + 1 + i]; }
    const int act = (int)row[S];
    const float r = row[2 * S + 1], d = row[2 * S + 2];
    float qs[TD_MAX_OUT], qn[TD_MAX_OUT], qt[TD_MAX_OUT];
    for (int aa = 0; aa < A; ++aa) { qs[aa] = 0.0f; qn[aa] = 0.0f; qt[aa] = 0.0f; }
    for (int j = 0; j < H; ++j) {
        const float *ro = onl + j * RP, *rt = tgt + j * RP;
        float z0 = 0.0f, z1 = 0.0f, z2 = 0.0f;
        for (int i = 0; i < S; ++i) { z0 = fma32(s[i], ro[i], z0); z1 = fma32(s2[i], ro[i], z1); z2 = fma32(s2[i], rt[i], z2); }
        const float h0 = act_fwd(a.q.act, a.q.prelu, z0 + ro[S]);
        const float h1 = act_fwd(a.q.act, a.q.prelu, z1 + ro[S]);
        const float h2 = act_fwd(a.q.act, a.q.prelu, z2 + rt[S]);
        for (int aa = 0; aa < A; ++aa) {
            qs[aa] = fma32(h0, ro[S + 1 + aa], qs[aa]);
            qn[aa] = fma32(h1, ro[S + 1 + aa], qn[aa]);
            qt[aa] = fma32(h2, rt[S + 1 + aa], qt[aa]);
        }
    }
    int am = 0;
    float best = qn[0] + onl[H * RP];
    float q_act = 0.0f, q_tgt = qt[0] + tgt[H * RP];
    for (int aa = 0; aa < A; ++aa) {
        const float vn = qn[aa] + onl[H * RP + aa];
        if (aa > 0 && vn > best) { best = vn; am = aa; }
        if (aa == act) q_act = qs[aa] + onl[H * RP + aa];
    }
    for (int aa = 0; aa < A; ++aa) if (aa == am) q_tgt = qt[aa] + tgt[H * RP + aa];
    const float t1 = a.gamma32 * q_tgt;
    const float t2 = 1.0f - d;
    a.q_sa[chain * a.batch + b] = q_act;
    a.y[chain * a.batch + b] = r + t1 * t2;
}

}  // namespace lenv

using namespace lenv;

extern "C" int lenv_qnet_td_forward(const lenv_mlp_desc *q, const float *online, const float *target, const float *replay,
                                    int64_t replay_cap, int32_t row_stride, const int32_t *idx, int64_t chains,
                                    int32_t batch, double gamma, float *q_sa, float *y, void *stream)
{
    if (!q || !online || !target || !replay || !idx || !q_sa || !y || chains < 0 || batch < 1 || replay_cap < 1) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    if (q->use_layer_norm) return LENV_ERR_UNSUPPORTED;
    if (q->layers != 1 || q->in_dim > TD_MAX_IN || q->out_dim > TD_MAX_OUT || q->in_dim < 1 || q->out_dim < 1) return LENV_ERR_UNSUPPORTED;
    if (row_stride < 2 * q->in_dim + 3) return LENV_ERR_INVALID;
    TdArgs a;
    a.q = *q;
    a.P = (int64_t)q->in_dim * q->hidden + q->hidden + (int64_t)q->hidden * q->out_dim + q->out_dim;
    a.online = online; a.target = target; a.replay = replay; a.replay_cap = replay_cap; a.row_stride = row_stride;
    a.idx = idx; a.batch = batch; a.gamma32 = (float)gamma; a.q_sa = q_sa; a.y = y;
    a.RP = (q->in_dim + 1 + q->out_dim + 3) & ~3;
    const size_t lds_bytes = 2 * ((size_t)q->hidden * a.RP + ((q->out_dim + 3) & ~3)) * sizeof(float);
    if (lds_bytes > 160 * 1024 || chains > 65535) return LENV_ERR_UNSUPPORTED;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(qnet_td_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return LENV_ERR_LAUNCH;
    hipLaunchKernelGGL(qnet_td_kernel, dim3((unsigned)((batch + TD_NT - 1) / TD_NT), (unsigned)chains), dim3(TD_NT), lds_bytes,
                       static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}
