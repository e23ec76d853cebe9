// lenv_wavechain_host.h -- host-side entry points of the wave-chain kernels (library-internal; the C-ABI stays include/lenv_hip.h)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lenv_hip.h"

// index of the wave-chain shape `cfg` matches exactly (0 = none: the GEMM-queue kernel runs it)
int lenv_wc_dueling_shape(const lenv_ddqn_cfg *cfg);
// floats of one chain's arena in the wave-chain layout
int64_t lenv_wc_dueling_arena_floats(const lenv_ddqn_cfg *cfg, int shape, int64_t rb_cap, int RS, int P_se);
int lenv_wc_dueling_team(const lenv_ddqn_cfg *cfg, int shape, int64_t chains);      // workgroups per chain the launch will use (1 or 2)
int lenv_wc_dueling_launch(int shape, const lenv_ddqn_cfg *cfg, const float *theta, const float *eps, const int32_t *worker, const float *sign,
                           const float *agent_init, const uint64_t *rng_keys, int64_t chains, float *arena, int64_t arena_stride, int64_t rb_cap,
                           int RS, int P, int P_se, const int *se_net_size, const lenv_inner_out *out, hipStream_t stream);

// the same for the TD3 kernel (td3_wavechain.hip): 1 when `cfg` is the published BASELINE configs[4] shape
int lenv_wc_td3_shape(const lenv_td3_cfg *cfg);
int64_t lenv_wc_td3_arena_floats(const lenv_td3_cfg *cfg, int64_t rb_cap, int RS);
int lenv_wc_td3_team(const lenv_td3_cfg *cfg, int64_t chains);                      // workgroups per chain the launch will use (1, 2, 3 or 6)
int lenv_wc_td3_launch(const lenv_td3_cfg *cfg, const float *theta, const float *eps, const int32_t *worker, const float *sign, const float *agent_init,
                       const uint64_t *rng_keys, int64_t chains, float *arena, int64_t arena_stride, int64_t rb_cap, int RS, int P, int Pa, int Pc, int P_rn,
                       const lenv_td3_out *out, hipStream_t stream);
