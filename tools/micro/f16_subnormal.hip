// MFMA f16 input subnormals on MI355X: v_mfma_f32_16x16x32_f16 keeps them (2^-24 x 1 comes out exact; each result below is 4 x the
// product because the operand sits in all four k groups).  hipcc --offload-arch=gfx950 -O2 f16_subnormal.hip -o f16dn && ./f16dn
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float av, float bv, float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0; b[i] = 0; }
    a[0] = (_Float16)av; b[0] = (_Float16)bv;   // lane l: A row l&15, k = 8*(l>>4)+i ; B col l&15
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x * 4 + 0] = c[0];
    if (threadIdx.x == 0) { out[256] = (float)a[0]; out[257] = (float)b[0]; }
}
int main() {
    float* d; hipMalloc(&d, 4096);
    float h[260];
    const float as[] = {1.0f, 9.5367431640625e-07f /*2^-20*/, 5.9604644775390625e-08f /*2^-24*/, 3.0517578125e-05f /*2^-15*/, 60000.f};
    const float bs[] = {1.0f, 1.0f, 1.0f, 1.0f, 60000.f};
    for (int t = 0; t < 5; ++t) {
        k<<<1, 64>>>(as[t], bs[t], d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> c[0,0]=%.10g  (a as f16=%g)\n", as[t], bs[t], h[0], h[256]);
    }
    return 0;
}
