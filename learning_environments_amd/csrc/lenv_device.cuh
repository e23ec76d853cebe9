// lenv_device.cuh -- device-side building blocks shared by the gfx950 kernels.
//
// Floating-point contract: compiled with -ffp-contract=off; an FMA exists only where
// __builtin_fmaf/__builtin_fma is written.  '/' and sqrtf are IEEE correctly rounded (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).  The transcendental routines below are fixed polynomial
// sequences (no ocml calls), so a gcc/x86 build of the same sequence is bitwise identical.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lenv_hip.h"
#include "lenv_tanh_table.h"
#include "lenv_cheetah_standin.h"

#define LENV_WAVE 64
// A member of a team of workgroups (a chain on several CUs) that waits this long for the others gives up: ticks of the constant
// 100 MHz clock (s_memrealtime), 0.25 s.  All team workgroups of a launch are resident at once on an idle device (host-side
// occupancy check, lenv_team_grid_resident), so only a foreign kernel holding CUs can make a member wait.
#define LENV_TEAM_GIVEUP_TICKS 25000000ull
// ... and once the team HAS assembled (its first barrier passed: every member is resident and stays so), a member is only ever late because
// its CU is time-sliced or stalled (another process's kernel, a debugger, a profiler): the later barriers and hand-overs wait 5 s before
// they call the launch off, so that such a stall does not silently turn a run into one-workgroup launches.
#define LENV_TEAM_GIVEUP_TICKS_RUN 500000000ull

// Host side of the team launches: can `grid` workgroups of `kern` (threads per workgroup, dynamic LDS bytes) all be resident at the
// same time on the current device?  (occupancy API x CU count; false without a device)
static inline bool lenv_team_grid_resident(const void *kern, int threads, size_t lds_bytes, long long grid)
{
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds_bytes) != hipSuccess) return false;
    return grid <= (long long)per_cu * cus;
}

namespace lenv {

__device__ __forceinline__ float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Canonical tanh v4 (same sequence as the oracle's orc_tanhf): t = min(|x|, TMAX), u = t + 2^19 (the addition rounds t to the
// grid of 1/16, the ulp of 2^19), i = bits(u) - bits(2^19), d = t - (u - 2^19) (exact, |d| <= 1/32),
// tanh(x) = copysign(((c3 d + c2) d + c1) d + c0, x) with the cubic (c0..c3)_i of lenv_tanh_table.h.
// 9 VALU instructions + one 16-byte gather; as a PAIR 15 (three packed adds serve both units).
// Table images: `tab` = one copy, entry i at tab[4 i] (the global copy below, or a plain LDS copy); `tab16` = SIXTEEN
// interleaved copies in LDS, LENV_TANH_N slots of 256 B: slot = entry index, copy c at byte 16 c of the slot.  A lane that
// reads copy (lane & 15) owns bank quad (lane & 15) of the 64 LDS banks, and the 16 lanes a ds_read_b128 services per cycle
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... -- MI355X_MICROARCH.md, LDS) have distinct (lane & 15): every gather is
// conflict-free.  Address = (bits(u) << 8) + 16 (lane & 15): the bits of 2^19 (0x49000000) leave the 32-bit word in the
// shift, so the address is ONE v_lshl_or_b32 and the image base rides in the DS instruction's immediate offset.
static __device__ const float lenv_tanh_table[LENV_TANH_N * 4] = LENV_TANH_TABLE_INIT;
constexpr int LENV_TANH16_FLOATS = LENV_TANH_N * 64, LENV_TANH1_FLOATS = LENV_TANH_N * 4;
static_assert(LENV_TANH_MAGIC_BITS == 0x49000000u && LENV_TANH_N <= 255, "tab16 addressing: bits(2^19) << 8 == 0 (mod 2^32), index in the low byte");

struct TanhArg { float d; int idx; uint32_t bits; };
__device__ __forceinline__ TanhArg det_tanh_arg(float x)
{
    const float ax = __builtin_fabsf(x);
    const float t = ax < LENV_TANH_TMAX ? ax : LENV_TANH_TMAX;
    const float u = t + LENV_TANH_MAGIC;
    const uint32_t b = __float_as_uint(u);
    TanhArg r;
    r.bits = b;
    r.d = t - (u - LENV_TANH_MAGIC);
    r.idx = (int)(b - LENV_TANH_MAGIC_BITS);
    return r;
}
__device__ __forceinline__ float det_tanh_poly(const float4 k, float d, float x)
{
    float p = fma32(k.w, d, k.z);
    p = fma32(p, d, k.y);
    p = fma32(p, d, k.x);
    return __builtin_copysignf(p, x);
}
__device__ __forceinline__ float det_tanhf(const float *tab, float x)
{
    const TanhArg a = det_tanh_arg(x);
    return det_tanh_poly(*reinterpret_cast<const float4 *>(tab + 4 * a.idx), a.d, x);
}
// Addressing of an LDS image with 16 copies (slot stride 256 B) or, when LDS is short, ONE copy (slot stride 16 B, gathers
// then conflict like any random 16-byte access): byte offset = (bits(u) << shift) + base, where base carries the lane's copy
// and cancels what is left of bits(2^19) after the shift (nothing for shift 8).  shift is wave-uniform.
struct TanhLds {
    uint32_t shift, base;
    __device__ __forceinline__ static TanhLds make(bool sixteen, int lane)
    {
        TanhLds t;
        t.shift = sixteen ? 8u : 4u;
        t.base = (sixteen ? 16u * (uint32_t)(lane & 15) : 0u) - (LENV_TANH_MAGIC_BITS << t.shift);
        return t;
    }
    __device__ __forceinline__ uint32_t off(uint32_t bits) const { return (bits << shift) + base; }
};
// byte offset of this lane's coefficients inside the image
__device__ __forceinline__ uint32_t det_tanh_lds_off(const TanhArg &a, const TanhLds &t) { return t.off(a.bits); }
// 16-byte gather at LDS byte address `img_addr + off`.  The address is formed as an INTEGER in the LDS address space: with
// the image at LDS address 0 the whole computation is v_lshrrev + v_and_or and the DS instruction needs no base add.
__device__ __forceinline__ float4 det_tanh_lds_gather(uint32_t img_addr, uint32_t off)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const f4v lds_f4;
    const f4v v = *reinterpret_cast<lds_f4 *>(static_cast<uintptr_t>(img_addr + off));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t lds_addr_of(const float *p)
{
    return (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const float *)p);
}
__device__ __forceinline__ float det_tanhf_lds(uint32_t img_addr, const TanhLds &t, float x)
{
    const TanhArg a = det_tanh_arg(x);
    return det_tanh_poly(det_tanh_lds_gather(img_addr, det_tanh_lds_off(a, t)), a.d, x);
}
// fill the LDS image (all threads of the workgroup; caller synchronises)
__device__ __forceinline__ void det_tanh_lds_stage(float *img, bool sixteen, int tid, int nthreads)
{
    const int per_slot = sixteen ? 64 : 4;
    for (int e = tid; e < LENV_TANH_N * per_slot; e += nthreads) img[e] = lenv_tanh_table[4 * (e / per_slot) + (e & 3)];
}

__device__ __forceinline__ double det_ksin(double x)
{
    double z = x * x;
    double p = 1.58969099521155010221e-10;
    p = fma64(p, z, -2.50507602534068634195e-08);
    p = fma64(p, z, 2.75573137070700676789e-06);
    p = fma64(p, z, -1.98412698298579493134e-04);
    p = fma64(p, z, 8.33333333332248946124e-03);
    p = fma64(p, z, -1.66666666666666324348e-01);
    return fma64(x * z, p, x);
}

__device__ __forceinline__ double det_kcos(double x)
{
    double z = x * x;
    double p = -1.13596475577881948265e-11;
    p = fma64(p, z, 2.08757232129817482790e-09);
    p = fma64(p, z, -2.75573143513906633035e-07);
    p = fma64(p, z, 2.48015872894767294178e-05);
    p = fma64(p, z, -1.38888888888741095749e-03);
    p = fma64(p, z, 4.16666666666666019037e-02);
    return 1.0 - fma64(-z * z, p, 0.5 * z);
}

__device__ __forceinline__ void det_reduce(double x, double &r, int &q)
{
    double kf = __builtin_rint(x * 6.36619772367581382433e-01);
    double t = fma64(-kf, 1.57079632673412561417e+00, x);
    t = fma64(-kf, 6.07710050630396597660e-11, t);
    t = fma64(-kf, 2.02226624871116645580e-21, t);
    r = t;
    q = (int)((long long)kf & 3);
}

__device__ __forceinline__ double det_sin(double x)
{
    double r; int q;
    det_reduce(x, r, q);
    double s = det_ksin(r), c = det_kcos(r);
    double v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}

__device__ __forceinline__ double det_cos(double x)
{
    double r; int q;
    det_reduce(x, r, q);
    double s = det_ksin(r), c = det_kcos(r);
    double v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}

// counter RNG: value = mix(mix(key + G*((stream<<56)^n)) ^ key); two splitmix64 finalisers
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

__host__ __device__ __forceinline__ uint64_t rng_u64(uint64_t key, uint32_t stream, uint64_t n)
{
    uint64_t x = key + 0x9e3779b97f4a7c15ULL * (((uint64_t)stream << 56) ^ n);
    return mix64(mix64(x) ^ key);
}

// ReplayBuffer.sample index draw n of a chain (utils.py:35 np.random.randint(0, size)): one 32-bit murmur3 finaliser over
// (n, key) -- this draw runs once per minibatch sample per learn step, where the 64-bit double mix above cost ~350 cycles
// of quarter-rate integer multiplies per wave; the mapping to [0, size) is the multiply-high of u64_to_below.
__host__ __device__ __forceinline__ uint32_t rng_replay_below(uint64_t key, uint64_t n, uint32_t size)
{
    uint32_t h = (uint32_t)n * 0x9e3779b1u + (uint32_t)key;
    h ^= (uint32_t)(key >> 32);
    h ^= h >> 16; h *= 0x85ebca6bu;
    h ^= h >> 13; h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return (uint32_t)(((uint64_t)h * (uint64_t)size) >> 32);
}

__device__ __forceinline__ double u64_to_unit(uint64_t u) { return (double)(u >> 11) * (1.0 / 9007199254740992.0); }
__device__ __forceinline__ uint32_t u64_to_below(uint64_t u, uint32_t n) { return (uint32_t)(((u >> 32) * (uint64_t)n) >> 32); }

enum { STREAM_EPS = 0, STREAM_ACTION = 1, STREAM_REPLAY = 2, STREAM_TRAIN_RESET = 3, STREAM_TEST_RESET = 4,
       STREAM_TD3_RAND_ACTION = 5, STREAM_TD3_ACT_NOISE = 6, STREAM_TD3_TEST_NOISE = 7, STREAM_TD3_POLICY_NOISE = 8,
       STREAM_NES_EPS = 9, STREAM_AGENT_INIT = 10,
       STREAM_VARY_HP = 11,     // host side only (agents/vary.py): the four hyper-parameter draws of a *_vary agent
       STREAM_ICM_INIT = 12 };  // fresh ICMModel parameters of an ICM agent (lenv_chain_uniform_init)

// natural log, same sequence as the oracle's orc_log (fdlibm scheme, fma Horner); used by the counter-mode Box-Muller
__device__ __forceinline__ double det_log(double x)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)u);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = 1.479819860511658591e-01;
    p = fma64(p, z, 1.531383769920937332e-01);
    p = fma64(p, z, 1.818357216161805012e-01);
    p = fma64(p, z, 2.222219843214978396e-01);
    p = fma64(p, z, 2.857142874366239149e-01);
    p = fma64(p, z, 3.999999999940941908e-01);
    p = fma64(p, z, 6.666666666666735130e-01);
    const double R = z * p;
    const double hfsq = 0.5 * f * f;
    const double de = (double)e;
    return de * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + de * 1.90821492927058770002e-10)) - f);
}

// deterministic expf (oracle: orc_expf, the same sequence): 2^n * p(r), n = rint(x*log2e), r = x - n*ln2 (two constants),
// p = degree-6 Horner in fmaf.  Used for the softmax of the ICM inverse model (arguments <= 0).
__device__ __forceinline__ float det_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = fma32(-n, 0.693145751953125f, x);
    r = fma32(-n, 1.42860682030941723212e-6f, r);
    float p = 1.0f / 720.0f;
    p = fma32(p, r, 1.0f / 120.0f);
    p = fma32(p, r, 1.0f / 24.0f);
    p = fma32(p, r, 1.0f / 6.0f);
    p = fma32(p, r, 0.5f);
    p = fma32(p, r, 1.0f);
    p = fma32(p, r, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

// N(0,1) = sqrt(-2 ln u1) * cos(2 pi u2) on counter draws 2n, 2n+1 (oracle: orc_normal)
__device__ __forceinline__ double det_normal(uint64_t key, uint32_t stream, uint64_t n)
{
    const double u1 = (double)((rng_u64(key, stream, 2 * n) >> 11) + 1) * (1.0 / 9007199254740992.0);
    const double u2 = u64_to_unit(rng_u64(key, stream, 2 * n + 1));
    return __builtin_sqrt(-2.0 * det_log(u1)) * det_cos(6.283185307179586 * u2);
}

// HalfCheetah-v3 STAND-IN constants (tools/gen_cheetah_standin.py; a build decision, not reference behaviour)
static __device__ const double lenv_cheetah_A[17 * 17] = LENV_CHEETAH_A_INIT;
static __device__ const double lenv_cheetah_B[17 * 6] = LENV_CHEETAH_B_INIT;
static __device__ const double lenv_cheetah_c[17] = LENV_CHEETAH_C_INIT;

// row i of x' = clip(c + A x + B a, -10, 10), accumulated left to right without FMA (oracle: orc_cheetah_step)
__device__ __forceinline__ double cheetah_row(int i, const double *x, const float *a)
{
    double acc = lenv_cheetah_c[i];
    for (int j = 0; j < 17; ++j) acc = acc + lenv_cheetah_A[i * 17 + j] * x[j];
    for (int k = 0; k < 6; ++k) acc = acc + lenv_cheetah_B[i * 6 + k] * (double)a[k];
    return acc < -10.0 ? -10.0 : (acc > 10.0 ? 10.0 : acc);
}

// ---- the continuous real envs behind the TD3 path (oracle: td3_env_* in lenv_oracle_td3.inc) ----
// S / A = observation / action dims, SD = fp64 words of the env's own state (= width of the reset tapes), INFO = entries of
// the step's info dict.  reset_word: word i of the reset state of episode `row`; step_word: word i of the next state;
// obs: fp32 observation i of a state; the reward is split into the part that sees the OLD state (reward_pre, computed
// before the state is overwritten in place) and the part that sees the new one (reward_post).
template <int ENV> struct ContEnv;
template <> struct ContEnv<LENV_ENV_CHEETAH_STANDIN> {
    static constexpr int S = 17, A = 6, SD = 17, INFO = 4;
    static constexpr bool TERMINATES = false;              // episodes end only through TimeLimit
    static __device__ __forceinline__ bool done(const double *) { return false; }
    static __device__ __forceinline__ double reset_word(uint64_t key, uint32_t stream, int64_t row, int i)
    {
        return -0.1 + 0.2 * u64_to_unit(rng_u64(key, stream, (uint64_t)(row * 17 + i)));
    }
    static __device__ __forceinline__ double step_word(int i, const double *x, const float *a) { return cheetah_row(i, x, a); }
    static __device__ __forceinline__ float obs(int i, const double *x) { return (float)x[i]; }
    static __device__ __forceinline__ double reward_pre(const double *, const float *a)      // control cost sum a^2
    {
        double ctrl = 0.0;
        for (int k = 0; k < 6; ++k) ctrl = ctrl + (double)a[k] * (double)a[k];
        return ctrl;
    }
    static __device__ __forceinline__ double reward_post(const double *x_new, double pre) { return x_new[8] - 0.1 * pre; }
};
// gym==0.17.3 Pendulum-v0 (classic_control/pendulum.py; third party, restated; oracle: orc_pendulum_step): state = (theta,
// theta_dot); the torque arrives as fp32 and its square in the cost is an fp32 square
template <> struct ContEnv<LENV_ENV_PENDULUM> {
    static constexpr int S = 3, A = 1, SD = 2, INFO = 0;
    static constexpr bool TERMINATES = false;
    static __device__ __forceinline__ bool done(const double *) { return false; }
    static __device__ __forceinline__ double reset_word(uint64_t key, uint32_t stream, int64_t row, int i)
    {
        const double pi = 3.141592653589793, u = u64_to_unit(rng_u64(key, stream, (uint64_t)(row * 2 + i)));
        return i == 0 ? -pi + (2 * pi) * u : -1.0 + 2.0 * u;
    }
    static __device__ __forceinline__ float torque(const float *a) { return a[0] < -2.0f ? -2.0f : (a[0] > 2.0f ? 2.0f : a[0]); }
    static __device__ __forceinline__ double step_word(int i, const double *x, const float *a)
    {
        const double g = 10.0, m = 1.0, l = 1.0, dt = 0.05, pi = 3.141592653589793;
        const double u = (double)torque(a);
        double newthdot = x[1] + (-3 * g / (2 * l) * det_sin(x[0] + pi) + 3. / (m * (l * l)) * u) * dt;
        if (i == 0) return x[0] + newthdot * dt;
        return newthdot < -8.0 ? -8.0 : (newthdot > 8.0 ? 8.0 : newthdot);
    }
    static __device__ __forceinline__ float obs(int i, const double *x) { return i == 0 ? (float)det_cos(x[0]) : (i == 1 ? (float)det_sin(x[0]) : (float)x[1]); }
    static __device__ __forceinline__ double reward_pre(const double *x, const float *a)
    {
        const double pi = 3.141592653589793;
        const float u32 = torque(a);
        const double usq = (double)(u32 * u32);
        double an = fmod(x[0] + pi, 2 * pi);               // python float %: the result takes the divisor's sign
        if (an != 0.0 && an < 0.0) an += 2 * pi;
        an = an - pi;
        return -(an * an + .1 * (x[1] * x[1]) + .001 * usq);
    }
    static __device__ __forceinline__ double reward_post(const double *, double pre) { return pre; }
};

// gym==0.17.3 MountainCarContinuous-v0 (classic_control/continuous_mountain_car.py; third party, restated; oracle: orc_cmc_step):
// state = (position, velocity); force = the fp32 action clipped to [-1, 1], the action cost uses the unclipped action; the episode
// ends at the flag (position >= 0.45 while moving right) with a bonus of 100
template <> struct ContEnv<LENV_ENV_CMC> {
    static constexpr int S = 2, A = 1, SD = 2, INFO = 0;
    static constexpr bool TERMINATES = true;
    static __device__ __forceinline__ double reset_word(uint64_t key, uint32_t stream, int64_t row, int i)
    {
        return i == 0 ? -0.6 + 0.2 * u64_to_unit(rng_u64(key, stream, (uint64_t)(row * 2))) : 0.0;
    }
    static __device__ __forceinline__ double step_word(int i, const double *x, const float *a)
    {
        double position = x[0], velocity = x[1];
        const float f32 = a[0] < -1.0f ? -1.0f : (a[0] > 1.0f ? 1.0f : a[0]);
        velocity = velocity + ((double)f32 * 0.0015 - 0.0025 * det_cos(3 * position));
        if (velocity > 0.07) velocity = 0.07;
        if (velocity < -0.07) velocity = -0.07;
        position = position + velocity;
        if (position > 0.6) position = 0.6;
        if (position < -1.2) position = -1.2;
        if (position == -1.2 && velocity < 0) velocity = 0;
        return i == 0 ? position : velocity;
    }
    static __device__ __forceinline__ float obs(int i, const double *x) { return (float)x[i]; }
    static __device__ __forceinline__ bool done(const double *x) { return x[0] >= 0.45 && x[1] >= 0.0; }
    static __device__ __forceinline__ double reward_pre(const double *, const float *a) { return ((double)a[0] * (double)a[0]) * 0.1; }
    static __device__ __forceinline__ double reward_post(const double *x_new, double pre) { return (done(x_new) ? 100.0 : 0.0) - pre; }
};

__device__ __forceinline__ float act_fwd(int act, float prelu, float z)
{
    switch (act) {
    case LENV_ACT_RELU: return z > 0.0f ? z : 0.0f;
    case LENV_ACT_LEAKYRELU: return z > 0.0f ? z : z * 0.01f;
    case LENV_ACT_TANH: return det_tanhf(lenv_tanh_table, z);
    case LENV_ACT_PRELU: return z > 0.0f ? z : prelu * z;
    default: return z;
    }
}

// upstream gradient g through the activation, given the activation value a (sign(a) == sign(z) for the
// piecewise-linear activations, slopes are positive)
__device__ __forceinline__ float act_bwd(int act, float prelu, float a, float g)
{
    switch (act) {
    case LENV_ACT_RELU: return a > 0.0f ? g : 0.0f;
    case LENV_ACT_LEAKYRELU: return a > 0.0f ? g : g * 0.01f;
    case LENV_ACT_TANH: return g * fma32(-a, a, 1.0f);
    case LENV_ACT_PRELU: return a > 0.0f ? g : prelu * g;
    default: return g;
    }
}

// gym==0.17.3 CartPole-v0 dynamics (third party; oracle: orc_cartpole_step)
__device__ __forceinline__ void cartpole_step(double st[4], int action, double &reward, int &done)
{
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1, length = 0.5, force_mag = 10.0, tau = 0.02;
    const double total_mass = masspole + masscart;
    const double polemass_length = masspole * length;
    const double theta_thr = 12 * 2 * 3.141592653589793 / 360;
    const double x_thr = 2.4;
    double x = st[0], x_dot = st[1], theta = st[2], theta_dot = st[3];
    double force = action == 1 ? force_mag : -force_mag;
    double costheta = det_cos(theta), sintheta = det_sin(theta);
    double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
    double thetaacc = (gravity * sintheta - costheta * temp) /
                      (length * (4.0 / 3.0 - masspole * (costheta * costheta) / total_mass));
    double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    st[0] = x; st[1] = x_dot; st[2] = theta; st[3] = theta_dot;
    done = (x < -x_thr || x > x_thr || theta < -theta_thr || theta > theta_thr) ? 1 : 0;
    reward = 1.0;
}

// gym==0.17.3 MountainCar-v0 (classic_control/mountain_car.py; third party, restated; oracle: orc_mountaincar_step):
// state = (position, velocity)
__device__ __forceinline__ void mountaincar_step(double st[4], int action, double &reward, int &done)
{
    double position = st[0], velocity = st[1];
    velocity = velocity + ((double)(action - 1) * 0.001 + det_cos(3 * position) * (-0.0025));
    velocity = velocity < -0.07 ? -0.07 : (velocity > 0.07 ? 0.07 : velocity);
    position = position + velocity;
    position = position < -1.2 ? -1.2 : (position > 0.6 ? 0.6 : position);
    if (position == -1.2 && velocity < 0) velocity = 0;
    done = (position >= 0.5 && velocity >= 0) ? 1 : 0;
    reward = -1.0;
    st[0] = position; st[1] = velocity;
}

__device__ __forceinline__ void acrobot_dsdt(const double s[5], double out[5])
{
    const double m1 = 1., m2 = 1., l1 = 1., lc1 = .5, lc2 = .5, I1 = 1., I2 = 1., g = 9.8, pi = 3.141592653589793;
    double a = s[4], theta1 = s[0], theta2 = s[1], dtheta1 = s[2], dtheta2 = s[3];
    double c2 = det_cos(theta2), s2 = det_sin(theta2);
    double d1 = m1 * (lc1 * lc1) + m2 * (l1 * l1 + lc2 * lc2 + 2 * l1 * lc2 * c2) + I1 + I2;
    double d2 = m2 * (lc2 * lc2 + l1 * lc2 * c2) + I2;
    double phi2 = m2 * lc2 * g * det_cos(theta1 + theta2 - pi / 2.);
    double phi1 = -m2 * l1 * lc2 * (dtheta2 * dtheta2) * s2 - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * s2 +
                  (m1 * lc1 + m2 * l1) * g * det_cos(theta1 - pi / 2) + phi2;
    double ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * (dtheta1 * dtheta1) * s2 - phi2) /
                      (m2 * (lc2 * lc2) + I2 - (d2 * d2) / d1);
    double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
    out[0] = dtheta1; out[1] = dtheta2; out[2] = ddtheta1; out[3] = ddtheta2; out[4] = 0.;
}

__device__ __forceinline__ void acrobot_step(double st[4], int action, double &reward, int &done)
{
    const double pi = 3.141592653589793, dt = .2, dt2 = .2 / 2.0;
    const double max_vel1 = 4 * pi, max_vel2 = 9 * pi;
    double y0[5] = { st[0], st[1], st[2], st[3], (double)(action - 1) };
    double k1[5], k2[5], k3[5], k4[5], tmp[5], ns[5];
    acrobot_dsdt(y0, k1);
    for (int i = 0; i < 5; ++i) tmp[i] = y0[i] + dt2 * k1[i];
    acrobot_dsdt(tmp, k2);
    for (int i = 0; i < 5; ++i) tmp[i] = y0[i] + dt2 * k2[i];
    acrobot_dsdt(tmp, k3);
    for (int i = 0; i < 5; ++i) tmp[i] = y0[i] + dt * k3[i];
    acrobot_dsdt(tmp, k4);
    for (int i = 0; i < 5; ++i) ns[i] = y0[i] + dt / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    for (int i = 0; i < 2; ++i) {
        double x = ns[i], diff = pi - (-pi);
        while (x > pi) x = x - diff;
        while (x < -pi) x = x + diff;
        ns[i] = x;
    }
    ns[2] = __builtin_fmin(__builtin_fmax(ns[2], -max_vel1), max_vel1);
    ns[3] = __builtin_fmin(__builtin_fmax(ns[3], -max_vel2), max_vel2);
    for (int i = 0; i < 4; ++i) st[i] = ns[i];
    int terminal = (-det_cos(st[0]) - det_cos(st[1] + st[0]) > 1.) ? 1 : 0;
    done = terminal;
    reward = terminal ? 0. : -1.;
}

// ---- the discrete-action real envs behind one interface (env_id: include/lenv_hip.h) ----
__device__ __forceinline__ void real_env_step(int env_id, double st[4], int action, double &reward, int &done)
{
    if (env_id == LENV_ENV_CARTPOLE) cartpole_step(st, action, reward, done);
    else if (env_id == LENV_ENV_MOUNTAINCAR) mountaincar_step(st, action, reward, done);
    else acrobot_step(st, action, reward, done);
}

// fp32 observation of the fp64 state (CartPole: the state; Acrobot: cos/sin of the angles + velocities; MountainCar: the state)
__device__ __forceinline__ void real_env_obs(int env_id, const double st[4], float *obs)
{
    if (env_id == LENV_ENV_CARTPOLE) { for (int i = 0; i < 4; ++i) obs[i] = (float)st[i]; }
    else if (env_id == LENV_ENV_MOUNTAINCAR) { obs[0] = (float)st[0]; obs[1] = (float)st[1]; }
    else {
        obs[0] = (float)det_cos(st[0]); obs[1] = (float)det_sin(st[0]); obs[2] = (float)det_cos(st[1]); obs[3] = (float)det_sin(st[1]);
        obs[4] = (float)st[2]; obs[5] = (float)st[3];
    }
}

// reset state from the counter RNG: CartPole U(-0.05, 0.05)^4, Acrobot U(-0.1, 0.1)^4, MountainCar (U(-0.6, -0.4), 0); draws
// rng(key, stream, 4 * episode + i) -- oracle: draw_reset
__device__ __forceinline__ void real_env_reset_draw(int env_id, uint64_t key, uint32_t stream, int64_t episode, double st[4])
{
    if (env_id == LENV_ENV_MOUNTAINCAR) {
        st[0] = -0.6 + 0.2 * u64_to_unit(rng_u64(key, stream, (uint64_t)(episode * 4)));
        st[1] = st[2] = st[3] = 0.0;
        return;
    }
    const double lim = env_id == LENV_ENV_CARTPOLE ? 0.05 : 0.1;
    for (int i = 0; i < 4; ++i) st[i] = -lim + (2 * lim) * u64_to_unit(rng_u64(key, stream, (uint64_t)(episode * 4 + i)));
}

// BaseAgent.env_solved (agents/base_agent.py:49-62) over the reward meter's list (AverageMeter.get_mean / get_mean_last / _mean,
// utils.py:94-105: a slice sum over (len + 1e-9)); the caller checks episode >= init_episodes (base_agent.py:141).  virtual_rule:
// break_env is a VirtualEnv (train(env, test_env=None) on one, lenv_ddqn_cfg::test_mode 1): the mean of the last `num` entries
// against the mean of the `num` before them; otherwise the real rule avg >= solved_reward.  Oracle: meter_env_solved (lenv_oracle.c).
__device__ __forceinline__ int meter_env_solved_inl(const double *meter, int n, int num, bool virtual_rule, double solved_reward, double virtual_diff,
                                                    int episode, int init_episodes)
{
    int lo = n - num; if (lo < 0) lo = 0;
    double sm = 0.0;
    for (int i = lo; i < n; ++i) sm += meter[i];
    const double avg = sm / ((double)(n - lo) + 1e-9);
    if (!virtual_rule) return avg >= solved_reward;
    int hi2 = n - num; if (hi2 < 0) hi2 = 0;
    int lo2 = n - 2 * num; if (lo2 < 0) lo2 = 0;
    double s2 = 0.0;
    for (int i = lo2; i < hi2; ++i) s2 += meter[i];
    const double last = s2 / ((double)(hi2 - lo2) + 1e-9);
    return __builtin_fabs(avg - last) / (__builtin_fabs(last) + 1e-9) < virtual_diff && episode >= init_episodes + num;
}
// the out-of-line form (the GEMM-queue kernels: one thread per chain calls it once per training episode)
[[maybe_unused]] static __device__ __noinline__ int meter_env_solved(const double *meter, int n, int num, bool virtual_rule, double solved_reward, double virtual_diff,
                                                                     int episode, int init_episodes)
{
    return meter_env_solved_inl(meter, n, num, virtual_rule, solved_reward, virtual_diff, episode, init_episodes);
}

}  // namespace lenv
