#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    int x = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    int* d; hipMalloc(&d, 1024 * 4);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 160 * 1024 - 64, 0, d);  // one block per CU (LDS-limited)
    int h[256]; hipMemcpy(h, d, 256 * 4, hipMemcpyDeviceToHost);
    int cnt[16] = {0}; int same = 0;
    for (int i = 0; i < 256; ++i) { cnt[h[i] & 15]++; same += (h[i] == (i & 7)); }
    for (int i = 0; i < 8; ++i) printf("xcc %d: %d blocks\n", i, cnt[i]);
    printf("blocks with xcc == b %% 8: %d of 256\n", same);
    return 0;
}
