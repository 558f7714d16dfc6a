// Philox4x32-10 counter streams shared by csrc/rng.hip (the stand-alone draws) and the kernels that draw INSIDE their consumer
// (round 6: reparameterisation noise, the class prior, the word-dropout mask) - element i of a draw of (seed, offset) is lane i % 4 of
// philox(seed, offset + base + i / 4, stream), whoever evaluates it: a consumer that draws for itself produces the numbers the
// stand-alone launch would have written.  Streams: 0 normal, 1 uniform f32, 2 Bernoulli / one-hot, 3 uniform f64.
#pragma once
#include "cpg_common.h"

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32(uint64_t seed, uint64_t ctr, uint32_t stream, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), stream, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)


// the four N(0,1) values of quad q of a normal draw (rng_normal_kernel's arithmetic)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t ctr, float (&v)[4]) {
    uint32_t r[4];
    philox4x32(seed, ctr, 0u, r);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float rad = sqrtf(-2.f * logf(u01(r[2 * h])));
        const float ang = 6.283185307179586f * u01(r[2 * h + 1]);
        v[2 * h] = rad * cosf(ang);
        v[2 * h + 1] = rad * sinf(ang);
    }
}
