// Which XCD does workgroup i of a launch land on?  (The team launches of td3_wavechain.hip / dueling_wavechain.hip put the members of a
// chain at block indices that agree mod 8 and check HW_REG_XCC_ID at run time.)  Prints the id of the first 32 blocks and whether
// id == blockIdx % 8 held for all blocks of a 256-block launch of one-per-CU workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out)
{
    extern __shared__ float lds[];
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    lds[threadIdx.x] = 0.0f;
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 150 * 1024, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int ok = 1;
        for (int i = 0; i < 256; ++i) ok &= (h[i] == (unsigned)(i & 7));
        printf("rep %d: xcc ids of blocks 0..31:", rep);
        for (int i = 0; i < 32; ++i) printf(" %u", h[i]);
        printf("\n        id == block %% 8 for all 256 blocks: %s\n", ok ? "yes" : "NO");
    }
    return 0;
}
