// mlp_forward.hip -- batched forward of one MLP built by build_nn_from_config (models/model_utils.py:4-39): the generic
// entry behind Critic_DQN / Actor_TD3 / Critic_Q / reward_net calls of the one-step Python API (models/actor_critic.py,
// envs/reward_env.py:81-110).  One workgroup per input row; lane = output unit of the current layer, every dot product a
// sequential fmaf chain in index order with the bias added last (canonical order of oracle/lenv_oracle.h).
#include "lenv_device.cuh"

namespace lenv {

constexpr int MF_NT = 256;

__global__ __launch_bounds__(MF_NT) void mlp_forward_kernel(lenv_mlp_desc d, const float *params, const float *x, int64_t rows, float *y)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int W = d.hidden > d.in_dim ? d.hidden : d.in_dim;
    float *a0 = lds, *a1 = lds + ((W + 3) & ~3);
    for (int i = tid; i < d.in_dim; i += MF_NT) a0[i] = x[row * d.in_dim + i];
    __syncthreads();
    const float *p = params;
    int n_in = d.in_dim;
    float *in = a0, *out = a1;
    const float *ln_w = nullptr, *ln_b = nullptr;
    for (int l = 0; l < d.layers; ++l) {
        const float *Wl = p, *bl = p + (int64_t)d.hidden * n_in;
        const bool ln = d.use_layer_norm && l >= 1;        // model_utils.py:33-36: norm after every hidden Linear but the first
        for (int j = tid; j < d.hidden; j += MF_NT) {
            float acc = 0.0f;
            for (int k = 0; k < n_in; ++k) acc = fma32(in[k], Wl[(int64_t)j * n_in + k], acc);
            out[j] = ln ? acc + bl[j] : act_fwd(d.act, d.prelu, acc + bl[j]);
        }
        __syncthreads();
        p += (int64_t)d.hidden * n_in + d.hidden;
        if (ln) {
            // nn.LayerNorm(hidden), eps 1e-5, biased variance; sums in index order (every thread the same values), then
            // y = fma((z - mean) * rstd, w, b) and the activation -- oracle: mlp_forward_one
            if (l == 1) { ln_w = p; ln_b = p + d.hidden; p += 2 * d.hidden; }
            float sm = 0.0f, sv = 0.0f;
            for (int j = 0; j < d.hidden; ++j) sm = sm + out[j];
            const float mean = sm / (float)d.hidden;
            for (int j = 0; j < d.hidden; ++j) { const float dj = out[j] - mean; sv = fma32(dj, dj, sv); }
            const float rstd = 1.0f / __builtin_sqrtf(sv / (float)d.hidden + 1e-5f);
            __syncthreads();
            for (int j = tid; j < d.hidden; j += MF_NT) out[j] = act_fwd(d.act, d.prelu, fma32((out[j] - mean) * rstd, ln_w[j], ln_b[j]));
            __syncthreads();
        }
        n_in = d.hidden;
        float *t = in; in = out; out = t;
    }
    const float *Wo = p, *bo = p + (int64_t)d.out_dim * n_in;
    for (int o = tid; o < d.out_dim; o += MF_NT) {
        float acc = 0.0f;
        for (int k = 0; k < n_in; ++k) acc = fma32(in[k], Wo[(int64_t)o * n_in + k], acc);
        y[row * d.out_dim + o] = acc + bo[o];
    }
}

// HalfCheetah-v3 STAND-IN reset / step for n instances (tools/gen_cheetah_standin.py)
__global__ void cheetah_reset_kernel(const uint64_t *keys, const int64_t *episode, int64_t n, double *state, float *obs, int32_t *elapsed)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * 17) return;
    const int64_t i = e / 17; const int k = (int)(e - i * 17);
    const double v = -0.1 + 0.2 * u64_to_unit(rng_u64(keys[i], STREAM_TEST_RESET, (uint64_t)(episode[i] * 17 + k)));
    state[e] = v; obs[e] = (float)v;
    if (k == 0) elapsed[i] = 0;
}

__global__ void cheetah_step_kernel(int max_steps, int64_t n, const float *action, double *state, int32_t *elapsed, float *obs,
                                    float *reward, float *done)
{
    __shared__ double nx[17];
    const int64_t i = blockIdx.x;
    const int k = threadIdx.x;
    if (k < 17) nx[k] = cheetah_row(k, state + i * 17, action + i * 6);
    __syncthreads();
    if (k < 17) { state[i * 17 + k] = nx[k]; obs[i * 17 + k] = (float)nx[k]; }
    if (k == 0) {
        double ctrl = 0.0;
        for (int j = 0; j < 6; ++j) ctrl = ctrl + (double)action[i * 6 + j] * (double)action[i * 6 + j];
        reward[i] = (float)(nx[8] - 0.1 * ctrl);
        const int el = elapsed[i] + 1;
        elapsed[i] = el;
        done[i] = el >= max_steps ? 1.0f : 0.0f;
    }
}

// one env instance per workgroup of 64 threads, any ContEnv
template <int ENV>
__global__ void cont_env_reset_kernel(const uint64_t *keys, const int64_t *episode, int64_t n, double *state, float *obs, int32_t *elapsed)
{
    using E = ContEnv<ENV>;
    __shared__ double st[E::SD];
    const int64_t i = blockIdx.x;
    const int k = threadIdx.x;
    if (k < E::SD) { st[k] = E::reset_word(keys[i], STREAM_TEST_RESET, episode[i], k); state[i * E::SD + k] = st[k]; }
    __syncthreads();
    if (k < E::S) obs[i * E::S + k] = E::obs(k, st);
    if (k == 0) elapsed[i] = 0;
}

template <int ENV>
__global__ void cont_env_step_kernel(int max_steps, int64_t n, const float *action, double *state, int32_t *elapsed, float *obs,
                                     float *reward, float *done)
{
    using E = ContEnv<ENV>;
    __shared__ double nx[E::SD];
    const int64_t i = blockIdx.x;
    const int k = threadIdx.x;
    double pre = 0.0;
    if (k < E::SD) nx[k] = E::step_word(k, state + i * E::SD, action + i * E::A);
    if (k == 0) pre = E::reward_pre(state + i * E::SD, action + i * E::A);
    __syncthreads();
    if (k < E::SD) state[i * E::SD + k] = nx[k];
    if (k < E::S) obs[i * E::S + k] = E::obs(k, nx);
    if (k == 0) {
        reward[i] = (float)E::reward_post(nx, pre);
        const int el = elapsed[i] + 1;
        elapsed[i] = el;
        done[i] = (E::done(nx) || el >= max_steps) ? 1.0f : 0.0f;
    }
}

}  // namespace lenv

using namespace lenv;

extern "C" int lenv_cont_env_reset(int32_t env_id, const uint64_t *keys, const int64_t *episode, int64_t n, double *state, float *obs,
                                   int32_t *elapsed, void *stream)
{
    if (!keys || !episode || !state || !obs || !elapsed || n < 0) return LENV_ERR_INVALID;
    if (env_id != LENV_ENV_CHEETAH_STANDIN && env_id != LENV_ENV_PENDULUM && env_id != LENV_ENV_CMC) return LENV_ERR_UNSUPPORTED;
    if (n == 0) return LENV_OK;
    if (env_id == LENV_ENV_PENDULUM)
        hipLaunchKernelGGL(cont_env_reset_kernel<LENV_ENV_PENDULUM>, dim3((unsigned)n), dim3(64), 0, static_cast<hipStream_t>(stream), keys, episode, n, state, obs, elapsed);
    else if (env_id == LENV_ENV_CMC)
        hipLaunchKernelGGL(cont_env_reset_kernel<LENV_ENV_CMC>, dim3((unsigned)n), dim3(64), 0, static_cast<hipStream_t>(stream), keys, episode, n, state, obs, elapsed);
    else
        hipLaunchKernelGGL(cont_env_reset_kernel<LENV_ENV_CHEETAH_STANDIN>, dim3((unsigned)n), dim3(64), 0, static_cast<hipStream_t>(stream), keys, episode, n, state, obs, elapsed);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_cont_env_step(int32_t env_id, int32_t max_steps, int64_t n, const float *action, double *state, int32_t *elapsed,
                                  float *obs, float *reward, float *done, void *stream)
{
    if (!action || !state || !elapsed || !obs || !reward || !done || n < 0) return LENV_ERR_INVALID;
    if (env_id != LENV_ENV_CHEETAH_STANDIN && env_id != LENV_ENV_PENDULUM && env_id != LENV_ENV_CMC) return LENV_ERR_UNSUPPORTED;
    if (n == 0) return LENV_OK;
    if (env_id == LENV_ENV_PENDULUM)
        hipLaunchKernelGGL(cont_env_step_kernel<LENV_ENV_PENDULUM>, dim3((unsigned)n), dim3(64), 0, static_cast<hipStream_t>(stream), (int)max_steps, n,
                           action, state, elapsed, obs, reward, done);
    else if (env_id == LENV_ENV_CMC)
        hipLaunchKernelGGL(cont_env_step_kernel<LENV_ENV_CMC>, dim3((unsigned)n), dim3(64), 0, static_cast<hipStream_t>(stream), (int)max_steps, n,
                           action, state, elapsed, obs, reward, done);
    else
        hipLaunchKernelGGL(cont_env_step_kernel<LENV_ENV_CHEETAH_STANDIN>, dim3((unsigned)n), dim3(64), 0, static_cast<hipStream_t>(stream), (int)max_steps, n,
                           action, state, elapsed, obs, reward, done);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_mlp_forward(const lenv_mlp_desc *d, const float *params, const float *x, int64_t rows, float *y, void *stream)
{
    if (!d || !params || !x || !y || rows < 0) return LENV_ERR_INVALID;
    if (d->layers < 1 || d->hidden < 1 || d->in_dim < 1 || d->out_dim < 1) return LENV_ERR_INVALID;
    if (rows == 0) return LENV_OK;
    const int W = d->hidden > d->in_dim ? d->hidden : d->in_dim;
    const size_t lds_bytes = 2 * (size_t)((W + 3) & ~3) * sizeof(float) + 16;
    if (lds_bytes > 160 * 1024) return LENV_ERR_UNSUPPORTED;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return LENV_ERR_LAUNCH;
    hipLaunchKernelGGL(mlp_forward_kernel, dim3((unsigned)rows), dim3(MF_NT), lds_bytes, static_cast<hipStream_t>(stream), *d, params, x, rows, y);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_cheetah_standin_reset(const uint64_t *keys, const int64_t *episode, int64_t n, double *state, float *obs,
                                          int32_t *elapsed, void *stream)
{
    if (!keys || !episode || !state || !obs || !elapsed || n < 0) return LENV_ERR_INVALID;
    if (n == 0) return LENV_OK;
    hipLaunchKernelGGL(cheetah_reset_kernel, dim3((unsigned)((n * 17 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), keys,
                       episode, n, state, obs, elapsed);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_cheetah_standin_step(int32_t max_steps, int64_t n, const float *action, double *state, int32_t *elapsed, float *obs,
                                         float *reward, float *done, void *stream)
{
    if (!action || !state || !elapsed || !obs || !reward || !done || n < 0) return LENV_ERR_INVALID;
    if (n == 0) return LENV_OK;
    hipLaunchKernelGGL(cheetah_step_kernel, dim3((unsigned)n), dim3(64), 0, static_cast<hipStream_t>(stream), (int)max_steps, n, action,
                       state, elapsed, obs, reward, done);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}
