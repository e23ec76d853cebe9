// se_step.hip -- population-batched VirtualEnv step (K1 + K10 of SURVEY.md §2a) for gfx950.
//
// Replaces EnvWrapper.step -> VirtualEnv.step (envs/env_wrapper.py:16-47, envs/virtual_env.py:43-54) for
// `chains` perturbed synthetic environments in one launch: workgroup c stages W_c = theta + sign[c]*eps[worker[c]]
// (agents/GTN_worker.py:165-175) into LDS once (coalesced reads of the noise row), then evaluates the three
// MLPs (state / reward / done) on [onehot(action), state] for each of its n_per_chain inputs.
// Lane = hidden unit for the hidden layers; the output layer is a sequential fmaf chain per output
// (canonical order of oracle/lenv_oracle.h), so results are bit-identical to the oracle.
#include "lenv_device.cuh"

namespace lenv {

constexpr int SE_NT = 256;

struct SeArgs {
    lenv_mlp_desc net[3];
    int64_t net_off[3];      // parameter offset of each net inside theta
    int64_t P;               // total parameters
    const float *theta, *eps; const int32_t *worker; const float *sign;
    int32_t n_per_chain;
    const float *state; const int32_t *action;
    float *next_state, *reward, *done;
    int S, A;
};

__global__ __launch_bounds__(SE_NT) void se_step_kernel(const SeArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const int64_t chain = blockIdx.x;
    float *W = lds;                                  // [P] canonical flat layout
    const int H = a.net[0].hidden, L = a.net[0].layers, K = a.S + a.A;
    float *x = lds + ((a.P + 3) & ~(int64_t)3);      // [K]
    float *h0 = x + ((K + 3) & ~3);                  // [3][H] ping
    float *h1 = h0 + 3 * H;                          // [3][H] pong
    float *stat = h1 + 3 * H;                        // [3][2] mean, 1 / sqrt(var + eps) of a LayerNorm position

    const float sg = a.eps ? a.sign[chain] : 0.0f;
    const float *e = a.eps ? a.eps + (int64_t)a.worker[chain] * a.P : nullptr;
    for (int64_t i = tid; i < a.P; i += SE_NT) W[i] = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i];

    for (int n = 0; n < a.n_per_chain; ++n) {
        const int64_t row = chain * a.n_per_chain + n;
        __syncthreads();
        if (tid < K) {
            const int act = a.action[row];
            x[tid] = tid < a.A ? (tid == act ? 1.0f : 0.0f) : a.state[row * a.S + (tid - a.A)];
        }
        __syncthreads();
        const float *in = x;
        int n_in = K;
        float *hout = h0;
        // `use_layer_norm` (models/model_utils.py:22-37): ONE shared nn.LayerNorm(hidden) behind every hidden Linear but the first, its
        // weight | bias behind the second Linear's bias (Module.parameters() order); the three nets come from one config section but
        // each carries its own flag.  layer_off / ln_off: per net.
        int64_t layer_off[3] = { 0, 0, 0 }, ln_off[3] = { 0, 0, 0 };
        for (int l = 0; l < L; ++l) {
            const bool any_ln = l >= 1 && (a.net[0].use_layer_norm || a.net[1].use_layer_norm || a.net[2].use_layer_norm);
            for (int u = tid; u < 3 * H; u += SE_NT) {
                const int net = u / H, j = u - net * H;
                const float *w = W + a.net_off[net] + layer_off[net] + (int64_t)j * n_in;
                const float *b = W + a.net_off[net] + layer_off[net] + (int64_t)H * n_in;
                const float *xin = (l == 0) ? in : in + net * H;
                float z = 0.0f;
                for (int k = 0; k < n_in; ++k) z = fma32(xin[k], w[k], z);
                z = z + b[j];
                hout[u] = (l >= 1 && a.net[net].use_layer_norm) ? z : act_fwd(a.net[net].act, a.net[net].prelu, z);
            }
            __syncthreads();
            for (int net = 0; net < 3; ++net) {
                layer_off[net] += (int64_t)H * n_in + H;
                if (l == 1 && a.net[net].use_layer_norm) { ln_off[net] = layer_off[net]; layer_off[net] += 2 * H; }
            }
            if (any_ln) {
                // sequential mean / biased variance per row (one thread per net), then y = fma((z - mean) * rstd, w, b) and the activation
                // -- the oracle's mlp_forward_one_ex
                if (tid < 3 && a.net[tid].use_layer_norm) {
                    const float *zr = hout + tid * H;
                    float sm = 0.0f, sv = 0.0f;
                    for (int j = 0; j < H; ++j) sm = sm + zr[j];
                    const float mean = sm / (float)H;
                    for (int j = 0; j < H; ++j) { const float dj = zr[j] - mean; sv = fma32(dj, dj, sv); }
                    stat[2 * tid] = mean; stat[2 * tid + 1] = 1.0f / __builtin_sqrtf(sv / (float)H + 1e-5f);
                }
                __syncthreads();
                for (int u = tid; u < 3 * H; u += SE_NT) {
                    const int net = u / H, j = u - net * H;
                    if (a.net[net].use_layer_norm) {
                        const float *lw = W + a.net_off[net] + ln_off[net], *lb = lw + H;
                        hout[u] = act_fwd(a.net[net].act, a.net[net].prelu, fma32((hout[u] - stat[2 * net]) * stat[2 * net + 1], lw[j], lb[j]));
                    }
                }
                __syncthreads();
            }
            in = hout;
            hout = (hout == h0) ? h1 : h0;
            n_in = H;
        }
        const int n_out_total = a.S + 2;
        if (tid < n_out_total) {
            const int net = tid < a.S ? 0 : (tid == a.S ? 1 : 2);
            const int o = tid < a.S ? tid : 0;
            const int n_out = a.net[net].out_dim;
            const float *w = W + a.net_off[net] + layer_off[net] + (int64_t)o * H;
            const float *b = W + a.net_off[net] + layer_off[net] + (int64_t)n_out * H;
            const float *hin = in + net * H;
            float acc = 0.0f;
            for (int j = 0; j < H; ++j) acc = fma32(hin[j], w[j], acc);
            acc = acc + b[o];
            if (net == 0) a.next_state[row * a.S + o] = acc;
            else if (net == 1) a.reward[row] = acc;
            else a.done[row] = acc;
        }
    }
}

}  // namespace lenv

using namespace lenv;

extern "C" int64_t lenv_mlp_num_params(const lenv_mlp_desc *d)
{
    if (!d || d->layers < 1) return -1;
    const int64_t H = d->hidden;
    return (int64_t)d->in_dim * H + H + (int64_t)(d->layers - 1) * (H * H + H) + H * d->out_dim + d->out_dim
           + ((d->use_layer_norm && d->layers >= 2) ? 2 * H : 0);      // the shared nn.LayerNorm's weight + bias
}

extern "C" int lenv_se_step_population(const lenv_mlp_desc *sn, const lenv_mlp_desc *rn, const lenv_mlp_desc *dn,
                                       const float *theta, const float *eps, const int32_t *worker, const float *sign,
                                       int64_t chains, int32_t n_per_chain, const float *state, const int32_t *action,
                                       float *next_state, float *reward, float *done, void *stream)
{
    if (!sn || !rn || !dn || !theta || !state || !action || !next_state || !reward || !done) return LENV_ERR_INVALID;
    if (eps && (!worker || !sign)) return LENV_ERR_INVALID;
    if (chains < 0 || n_per_chain < 1) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    // the three nets share input, width and depth (envs/virtual_env.py:23-31)
    if (sn->in_dim != rn->in_dim || sn->in_dim != dn->in_dim || sn->hidden != rn->hidden || sn->hidden != dn->hidden ||
        sn->layers != rn->layers || sn->layers != dn->layers || rn->out_dim != 1 || dn->out_dim != 1)
        return LENV_ERR_INVALID;
    SeArgs a;
    a.net[0] = *sn; a.net[1] = *rn; a.net[2] = *dn;
    a.S = sn->out_dim; a.A = sn->in_dim - sn->out_dim;
    if (a.A < 1 || a.S < 1 || a.S + 2 > SE_NT || sn->in_dim > SE_NT || sn->layers < 1) return LENV_ERR_UNSUPPORTED;
    a.net_off[0] = 0;
    a.net_off[1] = lenv_mlp_num_params(sn);
    a.net_off[2] = a.net_off[1] + lenv_mlp_num_params(rn);
    a.P = a.net_off[2] + lenv_mlp_num_params(dn);
    a.theta = theta; a.eps = eps; a.worker = worker; a.sign = sign; a.n_per_chain = n_per_chain;
    a.state = state; a.action = action; a.next_state = next_state; a.reward = reward; a.done = done;
    const size_t lds_floats = ((a.P + 3) & ~(int64_t)3) + ((sn->in_dim + 3) & ~3) + 6 * (size_t)sn->hidden + 16;
    const size_t lds_bytes = lds_floats * sizeof(float);
    if (lds_bytes > 160 * 1024) return LENV_ERR_UNSUPPORTED;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(se_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return LENV_ERR_LAUNCH;
    hipLaunchKernelGGL(se_step_kernel, dim3((unsigned)chains), dim3(SE_NT), lds_bytes, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}
