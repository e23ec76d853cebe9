// nes_update.hip -- GTN worker/master aggregation on device (K11 + K12 of SURVEY.md §2a) for gfx950.
//
//   lenv_nes_worker_best : GTN_Worker.calc_best_score   agents/GTN_worker.py:234-254
//   lenv_nes_rank_update : GTN_Master.score_transform   agents/GTN_master.py:197-265
//                        + GTN_Master.update_env        agents/GTN_master.py:267-298
//
// Every rank runs these redundantly on bit-identical gathered scores, so theta stays identical on all ranks
// with no broadcast.  Ranking is O(pop^2) compare-count with a documented stable order (ties: lower worker id
// first); the rank-only utilities (types 1,2,3) come from a host table computed with the reference's numpy
// formulas, score-dependent types (0,4,5,6,7) are plain fp64 arithmetic here.  The theta update streams the
// noise tensor eps[pop,P] coalesced along P, accumulating sequentially over workers in id order in fp32,
// exactly like the reference's python loop.
#include "lenv_device.cuh"

namespace lenv {

__global__ void worker_best_kernel(const double *chain_scores, int64_t pop, int mirrored, double *result)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pop) return;
    const double orig = chain_scores[p * 3], add = chain_scores[p * 3 + 1], sub = chain_scores[p * 3 + 2];
    double best = add, sign = 1.0;
    if (mirrored) {
        best = add > sub ? add : sub;        // max(score_add, score_sub)
        sign = sub > add ? -1.0 : 1.0;       // invert_eps() iff score_sub > score_add
    }
    result[p * 4] = best; result[p * 4 + 1] = orig; result[p * 4 + 2] = sign; result[p * 4 + 3] = 0.0;
}

// statistics.mean of python floats = the correctly rounded exact mean; same double-double accumulation as the oracle's
__device__ __forceinline__ double exact_mean(const double *x, int n)
{
    double hi = 0.0, lo = 0.0;
    for (int i = 0; i < n; ++i) {
        const double s = hi + x[i], bb = s - hi, err = (hi - (s - bb)) + (x[i] - bb);
        hi = s; lo = lo + err;
    }
    const double s = hi + lo, e = lo - (s - hi);
    const double q = s / (double)n, r = __builtin_fma(-q, (double)n, s) + e;
    return q + r / (double)n;
}

// num_grad_evals = G evaluations per direction: chain_scores [pop, 1+2G] = (orig, add_1..add_G, sub_1..sub_G);
// grad_eval_type 0 'mean', 1 'minmax' (the reference takes min() of BOTH lists, GTN_worker.py:238-240)
__global__ void worker_best_multi_kernel(const double *chain_scores, int64_t pop, int G, int mirrored, int type, double *result)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pop) return;
    const double *row = chain_scores + p * (1 + 2 * G);
    double add, sub;
    if (type == 0) { add = exact_mean(row + 1, G); sub = exact_mean(row + 1 + G, G); }
    else {
        add = row[1]; sub = row[1 + G];
        for (int i = 1; i < G; ++i) { if (row[1 + i] < add) add = row[1 + i]; if (row[1 + G + i] < sub) sub = row[1 + G + i]; }
    }
    double best = add, sign = 1.0;
    if (mirrored) { best = add > sub ? add : sub; sign = sub > add ? -1.0 : 1.0; }
    result[p * 4] = best; result[p * 4 + 1] = row[0]; result[p * 4 + 2] = sign; result[p * 4 + 3] = 0.0;
}

constexpr int RT_NT = 1024;

// single workgroup: weights_out[i] = score_transform(scores)[i]
__global__ __launch_bounds__(RT_NT) void score_transform_kernel(int type, const double *gathered, const double *rank_table,
                                                                int64_t pop, double *weights_out)
{
    __shared__ double red[RT_NT];
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    auto score = [&](int64_t i) { return gathered[i * 4]; };
    auto orig = [&](int64_t i) { return gathered[i * 4 + 1]; };

    if (type == 1 || type == 2 || type == 3) {
        for (int64_t i = tid; i < pop; i += RT_NT) {
            const double si = score(i);
            int64_t rank = 0;
            if (type == 1) { for (int64_t j = 0; j < pop; ++j) { double sj = score(j); rank += (sj < si) || (sj == si && j < i); } }
            else { for (int64_t j = 0; j < pop; ++j) { double sj = score(j); rank += (sj > si) || (sj == si && j < i); } }
            weights_out[i] = rank_table[rank];
        }
        if (type == 1) return;
        // types 2/3: rank_table holds the raw utilities max(0, log(n/2+1) - log(rank)); the normalisations run in
        // worker order like the reference's `scores / sum(scores)`, `scores /= max(scores)` (GTN_master.py:221-227)
        __syncthreads();
        if (tid == 0) {
            double sm = 0.0;
            for (int64_t i = 0; i < pop; ++i) sm += weights_out[i];
            double mx = 0.0;
            for (int64_t i = 0; i < pop; ++i) {
                double w = weights_out[i] / sm;
                if (type == 2) w -= 1.0 / (double)pop;
                weights_out[i] = w;
                if (i == 0 || w > mx) mx = w;
            }
            sh[0] = mx;
        }
        __syncthreads();
        const double mx = sh[0];
        for (int64_t i = tid; i < pop; i += RT_NT) weights_out[i] = weights_out[i] / mx;
        return;
    }
    // sequential reductions by thread 0 keep python's sum()/min()/max() order
    if (tid == 0) {
        double mn = score(0), mx = score(0), so = 0.0;
        int64_t am = 0, cnt = 0;
        for (int64_t i = 0; i < pop; ++i) {
            const double s = score(i);
            if (s < mn) mn = s;
            if (s > mx) { mx = s; am = i; }
            so += orig(i);
        }
        const double avg = so / (double)pop;
        for (int64_t i = 0; i < pop; ++i) cnt += score(i) > avg + 1e-6;
        sh[0] = mn; sh[1] = mx; sh[2] = avg; sh[3] = (double)am;
        red[0] = (double)cnt;
    }
    __syncthreads();
    const double mn = sh[0], mx = sh[1], avg = sh[2];
    const int64_t am = (int64_t)sh[3], cnt = (int64_t)red[0];
    __syncthreads();
    if (type == 0) {
        for (int64_t i = tid; i < pop; i += RT_NT) weights_out[i] = (score(i) - mn) / (mx - mn + 1e-9);
    } else if (type == 4 || (type == 5 && cnt > 0)) {
        for (int64_t i = tid; i < pop; i += RT_NT) weights_out[i] = i == am ? 1.0 : 0.0;
    } else if ((type == 6 || type == 7) && cnt > 0) {
        for (int64_t i = tid; i < pop; i += RT_NT) {
            const double idx = score(i) > avg + 1e-6 ? 1.0 : 0.0;
            weights_out[i] = idx * (score(i) - avg) / (mx - avg + 1e-9);
        }
        __syncthreads();
        if (tid == 0) {
            double m2 = weights_out[0], s2 = 0.0;
            for (int64_t i = 0; i < pop; ++i) { double w = weights_out[i]; if (w > m2) m2 = w; s2 += w; }
            sh[0] = type == 6 ? m2 : s2;
        }
        __syncthreads();
        const double dv = sh[0];
        for (int64_t i = tid; i < pop; i += RT_NT) weights_out[i] = weights_out[i] / dv;
    } else {
        for (int64_t i = tid; i < pop; i += RT_NT) weights_out[i] = 0.0;
    }
}

__global__ void update_env_kernel(float *theta, const float *eps, const double *gathered, const double *weights, int64_t pop,
                                  int64_t P, double ss, float decay, float *theta_prev, int64_t *generation)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && generation) generation[0] += 1;                 // captured generations: the next replay draws generation + 1
    if (i >= P) return;
    if (theta_prev) theta_prev[i] = theta[i];                     // the master saves / keeps the pre-update theta (GTN_master.py:95-101)
    float t = theta[i] * decay;                                   // weight decay, GTN_master.py:281-286
    for (int64_t w = 0; w < pop; ++w) {
        const float c = (float)(ss * weights[w]);                 // (ss * score_transform) * eps
        const float sg = (float)gathered[w * 4 + 2];              // mirrored sampling: eps := -eps
        t = t + c * (sg * eps[w * P + i]);
    }
    theta[i] = t;
}

// One generation's stochastic inputs in ONE launch (replaces torch.randn * noise_std, torch.rand -> agent init, the host
// computation + upload of the chain keys): element e of the flat index space
//   [0, pop*P)                      eps[w][i]        = (float)N(0,1) * noise_std   GTN_Worker.get_random_noise (GTN_worker.py:156-163)
//   [.., + chains*p_agent)          agent_init[c][i] = (2u - 1) * bound[i]        nn.Linear default init of a fresh agent
//   [.., + chains)                  rng_keys[c]      = lenv_chain_key(seed, generation, worker(c), kind(c))
// all from the counter RNG (same functions as the oracle's orc_nes_draw), so every rank regenerates identical tensors and
// the CPU oracle can reproduce a whole generation bit for bit.
constexpr uint64_t NES_EPS_DOMAIN = 0x6e65735f657073ULL;      // "nes_eps": separates the noise keys from the chain keys

__device__ __host__ __forceinline__ uint64_t chain_key_dev(uint64_t seed, uint64_t generation, uint64_t worker, uint64_t kind)
{
    uint64_t k = mix64(seed + 0x9e3779b97f4a7c15ULL);
    k = mix64(k ^ (generation + 0x9e3779b97f4a7c15ULL * 2));
    k = mix64(k ^ (worker + 0x9e3779b97f4a7c15ULL * 3));
    k = mix64(k ^ (kind + 0x9e3779b97f4a7c15ULL * 4));
    return k;
}

__global__ void nes_draw_kernel(uint64_t seed, uint64_t generation, const int64_t *generation_dev, int64_t pop, int64_t P, float noise_std, float *eps,
                                int64_t chains, int64_t cpw, int64_t worker_lo, int64_t p_agent, const float *bounds, float *agent_init,
                                uint64_t *rng_keys)
{
    if (generation_dev) generation = (uint64_t)generation_dev[0];      // device-resident counter (graph replays)
    const int64_t n_eps = eps ? pop * P : 0, n_init = agent_init ? chains * p_agent : 0, n_keys = rng_keys ? chains : 0;
    const int64_t total = n_eps + n_init + n_keys;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < n_eps) {
            const int64_t w = e / P, i = e - w * P;
            const uint64_t key = chain_key_dev(seed ^ NES_EPS_DOMAIN, generation, (uint64_t)w, 0);
            eps[e] = (float)det_normal(key, STREAM_NES_EPS, (uint64_t)i) * noise_std;
        } else if (e < n_eps + n_init) {
            const int64_t q = e - n_eps, c = q / p_agent, i = q - c * p_agent;
            const uint64_t key = chain_key_dev(seed, generation, (uint64_t)(worker_lo + c / cpw), (uint64_t)(c % cpw));
            const float u = (float)u64_to_unit(rng_u64(key, STREAM_AGENT_INIT, (uint64_t)i));
            agent_init[q] = (u * 2.0f - 1.0f) * bounds[i];
        } else {
            const int64_t c = e - n_eps - n_init;
            rng_keys[c] = chain_key_dev(seed, generation, (uint64_t)(worker_lo + c / cpw), (uint64_t)(c % cpw));
        }
    }
}

// out[c][i] = (2u - 1) * bounds[i], u = unit(rng(keys[c], rng_stream, i)): fresh nn.Linear-initialised parameter vectors
// keyed by the chains' counter-RNG keys (the agent_init draw of nes_draw_kernel for any stream: ICM modules use their own)
__global__ void chain_uniform_init_kernel(const uint64_t *keys, int64_t chains, uint32_t rng_stream, int64_t P, const float *bounds, float *out)
{
    const int64_t total = chains * P;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = e / P, i = e - c * P;
        const float u = (float)u64_to_unit(rng_u64(keys[c], rng_stream, (uint64_t)i));
        out[e] = (u * 2.0f - 1.0f) * bounds[i];
    }
}

// column 3 of the per-worker records = this rank's worst chain status (min over `n` int32), so a failure travels through the
// fitness all-gather (one thread; n is a few hundred)
__global__ void status_fold_kernel(const int32_t *status, int64_t n, double *result, int64_t pop)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int32_t m = 0;
        for (int64_t i = 0; i < n; ++i) m = status[i] < m ? status[i] : m;
        for (int64_t w = 0; w < pop; ++w) result[w * 4 + 3] = (double)m;
    }
}

}  // namespace lenv

using namespace lenv;

static int nes_draw_launch(uint64_t seed, uint64_t generation, const int64_t *generation_dev, int64_t pop, int64_t p_theta, float noise_std, float *eps,
                           int64_t chains, int32_t chains_per_worker, int64_t worker_lo, int64_t p_agent, const float *bounds,
                           float *agent_init, uint64_t *rng_keys, void *stream)
{
    if (pop < 0 || chains < 0 || chains_per_worker < 1 || (eps && p_theta < 1)) return LENV_ERR_INVALID;
    if (agent_init && (!bounds || p_agent < 1)) return LENV_ERR_INVALID;
    const int64_t total = (eps ? pop * p_theta : 0) + (agent_init ? chains * p_agent : 0) + (rng_keys ? chains : 0);
    if (total == 0) return LENV_OK;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(nes_draw_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), seed, generation, generation_dev, pop, p_theta,
                       noise_std, eps, chains, (int64_t)chains_per_worker, worker_lo, p_agent, bounds, agent_init, rng_keys);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_nes_draw(uint64_t seed, uint64_t generation, int64_t pop, int64_t p_theta, float noise_std, float *eps,
                             int64_t chains, int32_t chains_per_worker, int64_t worker_lo, int64_t p_agent, const float *bounds,
                             float *agent_init, uint64_t *rng_keys, void *stream)
{
    return nes_draw_launch(seed, generation, nullptr, pop, p_theta, noise_std, eps, chains, chains_per_worker, worker_lo, p_agent, bounds, agent_init,
                           rng_keys, stream);
}

extern "C" int lenv_nes_draw_dev(uint64_t seed, const int64_t *generation_dev, int64_t pop, int64_t p_theta, float noise_std, float *eps,
                                 int64_t chains, int32_t chains_per_worker, int64_t worker_lo, int64_t p_agent, const float *bounds,
                                 float *agent_init, uint64_t *rng_keys, void *stream)
{
    if (!generation_dev) return LENV_ERR_INVALID;
    return nes_draw_launch(seed, 0, generation_dev, pop, p_theta, noise_std, eps, chains, chains_per_worker, worker_lo, p_agent, bounds, agent_init,
                           rng_keys, stream);
}

extern "C" int lenv_chain_uniform_init(const uint64_t *rng_keys, int64_t chains, uint32_t rng_stream, int64_t p, const float *bounds, float *out,
                                       void *stream)
{
    if (!rng_keys || !bounds || !out || chains < 0 || p < 1) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    const int64_t total = chains * p;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(chain_uniform_init_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), rng_keys, chains, rng_stream, p, bounds, out);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_nes_worker_best(const double *chain_scores, int64_t pop, int32_t mirrored, double *result, void *stream)
{
    if (!chain_scores || !result || pop < 0) return LENV_ERR_INVALID;
    if (pop == 0) return LENV_OK;
    hipLaunchKernelGGL(worker_best_kernel, dim3((unsigned)((pop + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       chain_scores, pop, mirrored, result);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_nes_worker_best_multi(const double *chain_scores, int64_t pop, int32_t num_grad_evals, int32_t mirrored,
                                          int32_t grad_eval_type, double *result, void *stream)
{
    if (!chain_scores || !result || pop < 0 || num_grad_evals < 1) return LENV_ERR_INVALID;
    if (grad_eval_type != 0 && grad_eval_type != 1) return LENV_ERR_UNSUPPORTED;       // GTN_worker.py:242 NotImplementedError
    if (pop == 0) return LENV_OK;
    hipLaunchKernelGGL(worker_best_multi_kernel, dim3((unsigned)((pop + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       chain_scores, pop, num_grad_evals, mirrored, grad_eval_type, result);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_nes_status_fold(const int32_t *status, int64_t n, double *result, int64_t pop, void *stream)
{
    if (!status || !result || n < 0 || pop < 0) return LENV_ERR_INVALID;
    if (pop == 0) return LENV_OK;
    hipLaunchKernelGGL(status_fold_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), status, n, result, pop);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

static int rank_update_launch(int32_t type, const double *gathered, const double *rank_table, int64_t pop, float *theta,
                              const float *eps, int64_t p_theta, double step_size, int32_t nes_step_size,
                              double weight_decay, double *weights_out, float *theta_prev, int64_t *generation_dev, void *stream)
{
    if (!gathered || !weights_out || pop < 1) return LENV_ERR_INVALID;
    if (type < 0 || type > 7) return LENV_ERR_INVALID;            // ValueError("Unknown rank transform type") GTN_master.py:263
    if ((type >= 1 && type <= 3) && !rank_table) return LENV_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(score_transform_kernel, dim3(1), dim3(RT_NT), 0, st, (int)type, gathered, rank_table, pop, weights_out);
    if (hipGetLastError() != hipSuccess) return LENV_ERR_LAUNCH;
    if (theta) {
        if (!eps || p_theta < 1) return LENV_ERR_INVALID;
        double ss = step_size;
        if (nes_step_size) ss = ss / (double)pop;
        const float decay = (float)(1.0 - weight_decay);
        hipLaunchKernelGGL(update_env_kernel, dim3((unsigned)((p_theta + 255) / 256)), dim3(256), 0, st, theta, eps, gathered,
                           weights_out, pop, p_theta, ss, decay, theta_prev, generation_dev);
        if (hipGetLastError() != hipSuccess) return LENV_ERR_LAUNCH;
    }
    return LENV_OK;
}

extern "C" int lenv_nes_rank_update(int32_t type, const double *gathered, const double *rank_table, int64_t pop, float *theta,
                                    const float *eps, int64_t p_theta, double step_size, int32_t nes_step_size,
                                    double weight_decay, double *weights_out, void *stream)
{
    return rank_update_launch(type, gathered, rank_table, pop, theta, eps, p_theta, step_size, nes_step_size, weight_decay, weights_out, nullptr,
                              nullptr, stream);
}

extern "C" int lenv_nes_rank_update_keep(int32_t type, const double *gathered, const double *rank_table, int64_t pop, float *theta,
                                         const float *eps, int64_t p_theta, double step_size, int32_t nes_step_size,
                                         double weight_decay, double *weights_out, float *theta_prev, int64_t *generation_dev, void *stream)
{
    if (!theta) return LENV_ERR_INVALID;
    return rank_update_launch(type, gathered, rank_table, pop, theta, eps, p_theta, step_size, nes_step_size, weight_decay, weights_out, theta_prev,
                              generation_dev, stream);
}
