// Resident workgroups per CU of the fused attention forward by LDS footprint (hipOccupancyMaxActiveBlocksPerMultiprocessor), and the LDS the device reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../meta_tts_amd/csrc/attention.h"
using namespace mtts;
template <int LCAP> void one() {
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_kernel<LCAP, 16>, 256, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)attn_fwd_kernel<LCAP, 16>);
    printf("attn_fwd_kernel<%d,16>: static LDS %zu B, regs %d, resident workgroups per CU %d (%s)\n", LCAP, fa.sharedSizeBytes, fa.numRegs, n, hipGetErrorString(e));
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, CUs %d\n", p.gcnArchName, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.multiProcessorCount);
    one<128>(); one<352>(); one<480>(); one<608>(); one<1024>();
    return 0;
}
