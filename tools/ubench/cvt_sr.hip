// What does v_cvt_sr_bf16_f32 (gfx950 stochastic-rounding conversion) do with its "seed" operand?  The 16-bit stores of the
// bf16 handles use it as a DETERMINISTIC dither: the seed is a function of the output row, not a random number,
// and rn_create / the tests restate the conversion on the host -- so the exact rule matters.  Candidates, counted over random
// floats (both signs, several binades, zeros, subnormals) x random and structured seeds:
//   A: (bits + (seed & 0xffff)) >> 16     B: (bits + (seed >> 16)) >> 16      C: round-to-nearest-even (seed ignored)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/cvt_sr.hip -o tools/ubench/cvt_sr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, const unsigned* seed, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bf16x2 o = {0, 0};
    o = __builtin_amdgcn_cvt_sr_bf16_f32(o, x[i], seed[i], false);
    o = __builtin_amdgcn_cvt_sr_bf16_f32(o, -x[i], seed[i], true);
    out[i] = __builtin_bit_cast(unsigned, o);
}
static unsigned bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
int main() {
    const int n = 1 << 20;
    std::vector<float> x(n);
    std::vector<unsigned> s(n), o(n);
    std::mt19937 rng(1);
    for (int i = 0; i < n; ++i) {
        unsigned u = rng();
        const int kind = i & 7;
        if (kind == 0) u = (u & 0x807fffffu) | (static_cast<unsigned>(120 + (i >> 3) % 14) << 23);   // 2^-7 .. 2^6
        else if (kind == 1) u &= 0x807fffffu;                                                          // subnormals
        else if (kind == 2) u = (i & 8) ? 0u : 0x80000000u;                                            // zeros
        else u = (u & 0x807fffffu) | (static_cast<unsigned>(110 + u % 30) << 23);
        std::memcpy(&x[i], &u, 4);
        const unsigned r = rng();
        s[i] = (i & 16) ? r : ((i & 32) ? (r & 0xffffu) : (r << 16));
        if ((i & 0x3c0) == 0) s[i] = (i & 1) ? 0x8000u : ((i & 2) ? 0x2aaau : 0xd555u);
    }
    float* dx; unsigned *ds, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dout, n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    long bad[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    int shown = 0;
    for (int i = 0; i < n; ++i) {
        for (int half = 0; half < 2; ++half) {
            const unsigned b = bits(half ? -x[i] : x[i]);
            const unsigned got = half ? (o[i] >> 16) : (o[i] & 0xffffu);
            const unsigned a = (b + (s[i] & 0xffffu)) >> 16, bb = (b + (s[i] >> 16)) >> 16, c = (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
            bad[0][half] += got != (a & 0xffffu);
            bad[1][half] += got != (bb & 0xffffu);
            bad[2][half] += got != (c & 0xffffu);
            if (got != (a & 0xffffu) && shown < 12) {
                printf("  x bits %08x seed %08x -> %04x (A %04x, B %04x, RNE %04x)\n", b, s[i], got, a & 0xffffu, bb & 0xffffu, c & 0xffffu);
                ++shown;
            }
        }
    }
    printf("mismatches of %d (lo half / hi half): A (seed & 0xffff) %ld / %ld   B (seed >> 16) %ld / %ld   RNE %ld / %ld\n", n, bad[0][0], bad[0][1],
           bad[1][0], bad[1][1], bad[2][0], bad[2][1]);
    return 0;
}
