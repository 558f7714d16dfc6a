// Probe of ds_read_b64_tr_b16 on gfx950: which LDS element lands in which lane / slot.
// LDS holds one [32 k][16 x] bf16 subtile (row = 32 B), element value = k*16 + x.  Lane l = (s = l&15, q = l>>4) supplies the
// address of the 8-byte chunk (row 4q + (s>>2), chunk s&3); expected result: slot j = element [k = 4q + j][x = s].
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int rowoff) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[32 * 16];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, s = l & 15, q = l >> 4;
    auto p = reinterpret_cast<__attribute__((address_space(3))) s16x4*>(
        (__attribute__((address_space(3))) char*)lds + (rowoff + 4 * q + (s >> 2)) * 32 + (s & 3) * 8);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    int bad = 0;
    for (int rowoff = 0; rowoff <= 16; rowoff += 16) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rowoff);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int kk = rowoff + 4 * (l >> 4) + j, x = l & 15;
                if (h[l * 4 + j] != kk * 16 + x) {
                    if (bad < 8) printf("lane %d slot %d: got k=%d x=%d, expected k=%d x=%d\n", l, j, h[l * 4 + j] / 16, h[l * 4 + j] % 16, kk, x);
                    ++bad;
                }
            }
    }
    printf("tr_read_probe: %d mismatches\n", bad);
    return bad != 0;
}
