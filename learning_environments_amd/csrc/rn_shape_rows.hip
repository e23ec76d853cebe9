// rn_shape_rows.hip -- RewardEnv._calc_reward (reference envs/reward_env.py:68-133) for rows of a vector-state real env,
// all 11 reward types: the one-step RewardEnv.step API of config 5 (the fused TD3 kernel has the same arithmetic inline).
//   0: r   1: g*phi(s') - phi(s)   2: r + g*phi(s') - phi(s)   3/4: as 1/2 on [s | info]   5: phi(s')   6: r + phi(s')
//   7/8: as 5/6 on [s' | info]   101: w.info   102: r + w.info            (fp32, left to right, like the oracle)
// One 64-lane wave per row: lane = hidden unit (strided), sequential fmaf chains, output chain on lane 0.  Reward nets of 1-4 hidden layers
// (build_nn_from_config, models/model_utils.py:16-37), with the shared LayerNorm of `use_layer_norm` nets (its weight | bias behind the second
// Linear in theta, the layout of lenv_mlp_desc; reduced by lane 0 in the oracle's order).
#include <hip/hip_runtime.h>
#include "../../include/lenv_hip.h"
#include "lenv_device.cuh"

namespace lenv {

constexpr int RN_MAXH = 256, RN_MAXIN = 64;

struct RnRowsArgs {
    int type, S, info_dim, H, act, L, ln;
    float prelu, gamma;
    const float *theta, *s, *s2, *info, *r;
    float *out;
    int64_t rows;
};

__global__ __launch_bounds__(64) void rn_shape_rows_kernel(const RnRowsArgs a)
{
    __shared__ float x[2][RN_MAXIN], hbuf[2][RN_MAXH], phi[2], stat[2];
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x, t = a.type, S = a.S, nI = a.info_dim;
    const bool info_in = t == 3 || t == 4 || t == 7 || t == 8;
    const int D = info_in ? S + nI : S;
    const float r32 = a.r[row];
    if (t == 0) { if (lane == 0) a.out[row] = r32; return; }
    if (t > 100) {
        if (lane == 0) {
            float acc = 0.0f;
            for (int k = 0; k < nI; ++k) acc = fma32(a.info[row * nI + k], a.theta[k], acc);
            a.out[row] = t == 101 ? acc : r32 + acc;
        }
        return;
    }
    for (int k = lane; k < D; k += 64) {
        x[0][k] = k < S ? a.s[row * S + k] : a.info[row * nI + (k - S)];
        x[1][k] = k < S ? a.s2[row * S + k] : a.info[row * nI + (k - S)];
    }
    __syncthreads();
    const int H = a.H;
    const float *W0 = a.theta, *b0 = W0 + H * D;
    const bool ln = a.ln != 0 && a.L >= 2;
    const float *lnw = b0 + H + H * H + H, *lnb = lnw + H;        // behind the second Linear (Module.parameters() order)
    const bool need_s = t == 1 || t == 2 || t == 3 || t == 4;
    for (int which = need_s ? 0 : 1; which < 2; ++which) {
        float *h = hbuf[0], *hn = hbuf[1];
        for (int j = lane; j < H; j += 64) {
            float z = 0.0f;
            for (int k = 0; k < D; ++k) z = fma32(x[which][k], W0[j * D + k], z);
            h[j] = act_fwd(a.act, a.prelu, z + b0[j]);
        }
        __syncthreads();
        const float *Wl = b0 + H;
        for (int l = 1; l < a.L; ++l) {
            const float *bl = Wl + H * H;
            for (int j = lane; j < H; j += 64) {
                float z = 0.0f;
                for (int k = 0; k < H; ++k) z = fma32(h[k], Wl[j * H + k], z);
                z = z + bl[j];
                hn[j] = ln ? z : act_fwd(a.act, a.prelu, z);
            }
            __syncthreads();
            if (ln) {
                if (lane == 0) {
                    float sm = 0.0f, sv = 0.0f;
                    for (int j = 0; j < H; ++j) sm = sm + hn[j];
                    const float mean = sm / (float)H;
                    for (int j = 0; j < H; ++j) { const float dj = hn[j] - mean; sv = fma32(dj, dj, sv); }
                    stat[0] = mean; stat[1] = 1.0f / __builtin_sqrtf(sv / (float)H + 1e-5f);
                }
                __syncthreads();
                for (int j = lane; j < H; j += 64) hn[j] = act_fwd(a.act, a.prelu, fma32((hn[j] - stat[0]) * stat[1], lnw[j], lnb[j]));
                __syncthreads();
            }
            float *t2 = h; h = hn; hn = t2;
            Wl = bl + H + ((ln && l == 1) ? 2 * H : 0);
        }
        const float *Wo = Wl, *bo = Wo + H;
        if (lane == 0) {
            float acc = 0.0f;
            for (int j = 0; j < H; ++j) acc = fma32(h[j], Wo[j], acc);
            phi[which] = acc + bo[0];
        }
        __syncthreads();
    }
    if (lane == 0) {
        const float phi_s = phi[0], phi_s2 = phi[1];
        float shaped;
        switch (t) {
        case 1: case 3: shaped = a.gamma * phi_s2 - phi_s; break;
        case 2: case 4: shaped = (r32 + a.gamma * phi_s2) - phi_s; break;
        case 5: case 7: shaped = phi_s2; break;
        default: shaped = r32 + phi_s2; break;             // 6, 8
        }
        a.out[row] = shaped;
    }
}

}  // namespace lenv

using namespace lenv;

static bool rn_type_known(int t) { return (t >= 0 && t <= 8) || t == 101 || t == 102; }

extern "C" int64_t lenv_rn_num_params(int32_t type, int32_t state_dim, int32_t info_dim, int32_t hidden, int32_t layers)
{
    if (!rn_type_known(type)) return LENV_ERR_UNSUPPORTED;
    if (type == 0) return 0;
    if (type > 100) return info_dim;
    const int in = (type == 3 || type == 4 || type == 7 || type == 8) ? state_dim + info_dim : state_dim;
    lenv_mlp_desc d = { in, hidden, layers, 1, 0, 0.0f, 0 };
    return lenv_mlp_num_params(&d);
}

extern "C" int lenv_rn_shape_rows(int32_t type, const lenv_mlp_desc *rn, int32_t state_dim, int32_t info_dim, double gamma,
                                  const float *theta, const float *s, const float *s2, const float *info, const float *r,
                                  int64_t rows, float *out, void *stream)
{
    if (!rn_type_known(type)) return LENV_ERR_UNSUPPORTED;                 // reward_env.py:49,58: NotImplementedError
    if (!s || !s2 || !r || !out || rows < 0 || state_dim < 1) return LENV_ERR_INVALID;
    const bool uses_info = type == 3 || type == 4 || type == 7 || type == 8 || type > 100;
    if (uses_info && (!info || info_dim < 1)) return LENV_ERR_INVALID;     // reward_env.py:96,113,126: ValueError('No info dict ...')
    if (type != 0 && !theta) return LENV_ERR_INVALID;
    RnRowsArgs a{};
    a.type = type; a.S = state_dim; a.info_dim = info_dim; a.gamma = (float)gamma;
    if (type >= 1 && type <= 8) {
        if (!rn) return LENV_ERR_INVALID;
        const int D = (type == 3 || type == 4 || type == 7 || type == 8) ? state_dim + info_dim : state_dim;
        if (rn->layers < 1 || rn->layers > 4 || rn->out_dim != 1 || rn->in_dim != D || rn->hidden < 1 || rn->hidden > RN_MAXH || D > RN_MAXIN) return LENV_ERR_UNSUPPORTED;
        a.H = rn->hidden; a.act = rn->act; a.prelu = rn->prelu; a.L = rn->layers; a.ln = rn->use_layer_norm;
    }
    a.theta = theta; a.s = s; a.s2 = s2; a.info = info; a.r = r; a.out = out; a.rows = rows;
    if (rows == 0) return LENV_OK;
    hipLaunchKernelGGL(rn_shape_rows_kernel, dim3((unsigned)rows), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}
