// occupy_cus.hip -- TEST AID, not product: liblenv_diag.so (built by tools/diag/Makefile; tests/ and tools/diag/*.py load it with ctypes).
// Holds `blocks` compute units for `ticks` of the constant 100 MHz clock (s_memrealtime) on `stream`: every block asks for `lds_bytes` of
// LDS (>= 82 KiB: one block per CU) and spins.  The stand-in for "a foreign kernel occupies part of the device" in the tests of the team
// launches' give-up path (status -10, include/lenv_hip.h lenv_ddqn_cfg::team_size).  It lived in liblenv_hip.so's ABI until round 5.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void occupy_cus_kernel(long long ticks)
{
    extern __shared__ float occupy_lds[];
    if (threadIdx.x == 0) occupy_lds[0] = 0.0f;           // (the allocation is what matters)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int lenv_diag_occupy_cus(int32_t blocks, int32_t lds_bytes, int64_t ticks, void *stream)
{
    if (blocks < 1 || lds_bytes < 0 || lds_bytes > 160 * 1024 || ticks < 0) return -1;
    void (*kern)(long long) = occupy_cus_kernel;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -4;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), (size_t)lds_bytes, static_cast<hipStream_t>(stream), (long long)ticks);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
