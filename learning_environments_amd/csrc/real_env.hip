// real_env.hip -- batched real-environment reset/step on device (K13 of SURVEY.md §2a) for gfx950.
//
// Replaces gym==0.17.3 CartPole-v0 / Acrobot-v1 `reset`/`step` + TimeLimit as called through
// EnvWrapper.reset/step on a real env (envs/env_wrapper.py:49-70,72-85).  The same device functions are
// inlined into the fused inner loop's scoring rollouts; this standalone entry serves the one-env-one-step API.
// Thread = environment instance.  State is float64 like gym's; observations are cast to fp32 exactly where
// EnvWrapper does (`torch.tensor(state, dtype=float32)`).
#include "lenv_device.cuh"

namespace lenv {

__global__ void real_env_reset_kernel(int env_id, const uint64_t *keys, const int64_t *episode, int64_t n, double *state, float *obs,
                                      int32_t *elapsed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double st[4];
    real_env_reset_draw(env_id, keys[i], STREAM_TEST_RESET, episode[i], st);
    for (int k = 0; k < 4; ++k) state[i * 4 + k] = st[k];
    elapsed[i] = 0;
    const int S = env_id == LENV_ENV_CARTPOLE ? 4 : (env_id == LENV_ENV_MOUNTAINCAR ? 2 : 6);
    real_env_obs(env_id, st, obs + i * S);
}

__global__ void real_env_step_kernel(int env_id, int max_steps, int64_t n, const int32_t *action, double *state, int32_t *elapsed,
                                     float *obs, float *reward, float *done)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double st[4] = { state[i * 4], state[i * 4 + 1], state[i * 4 + 2], state[i * 4 + 3] };
    double rew; int dn;
    real_env_step(env_id, st, action[i], rew, dn);
    for (int k = 0; k < 4; ++k) state[i * 4 + k] = st[k];
    const int el = elapsed[i] + 1;
    elapsed[i] = el;
    if (el >= max_steps) dn = 1;                       // gym.wrappers.TimeLimit
    const int S = env_id == LENV_ENV_CARTPOLE ? 4 : (env_id == LENV_ENV_MOUNTAINCAR ? 2 : 6);
    real_env_obs(env_id, st, obs + i * S);
    reward[i] = (float)rew;
    done[i] = dn ? 1.0f : 0.0f;
}

}  // namespace lenv

using namespace lenv;

extern "C" int lenv_real_env_reset(int32_t env_id, const uint64_t *keys, const int64_t *episode, int64_t n, double *state, float *obs,
                                   int32_t *elapsed, void *stream)
{
    if (!keys || !episode || !state || !obs || !elapsed || n < 0) return LENV_ERR_INVALID;
    if (env_id != LENV_ENV_CARTPOLE && env_id != LENV_ENV_ACROBOT && env_id != LENV_ENV_MOUNTAINCAR) return LENV_ERR_UNSUPPORTED;
    if (n == 0) return LENV_OK;
    hipLaunchKernelGGL(real_env_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (int)env_id, keys, episode, n, state, obs, elapsed);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_real_env_step(int32_t env_id, int32_t max_steps, int64_t n, const int32_t *action, double *state, int32_t *elapsed,
                                  float *obs, float *reward, float *done, void *stream)
{
    if (!action || !state || !elapsed || !obs || !reward || !done || n < 0) return LENV_ERR_INVALID;
    if (env_id != LENV_ENV_CARTPOLE && env_id != LENV_ENV_ACROBOT && env_id != LENV_ENV_MOUNTAINCAR) return LENV_ERR_UNSUPPORTED;
    if (n == 0) return LENV_OK;
    hipLaunchKernelGGL(real_env_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (int)env_id, (int)max_steps, n, action, state, elapsed, obs, reward, done);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}
