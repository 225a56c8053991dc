// Microbenchmark: issue cost (cycles per instruction per wave) of the VALU instructions the stage
// epilogue is made of, alone and while another wave of the same SIMD runs an MFMA chain.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using i32x4 = __attribute__((ext_vector_type(4))) int;

#define R8(op)                                                                                        \
    asm volatile(op(0, 1) op(1, 2) op(2, 3) op(3, 4) op(4, 5) op(5, 6) op(6, 7) op(7, 0)               \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]))
// independent: destination i, source i (8 separate chains, dependency distance 8 instructions)
#define OP_ADD(i, j) "v_add_f32 %" #i ", %" #i ", %" #i "\n\t"
#define OP_ADD_WSHL(i, j) "v_add_f32_dpp %" #i ", %" #i ", %" #i " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_MOV_WSHL(i, j) "v_mov_b32_dpp %" #i ", %" #i " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_ADD_RSHL(i, j) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_ADD_RSHL2(i, j) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_MED3(i, j) "v_med3_f32 %" #i ", %" #i ", 0, %" #j "\n\t"
#define OP_FMA(i, j) "v_fma_f32 %" #i ", %" #i ", %" #j ", %" #i "\n\t"
#define OP_SWAP(i, j) "v_permlane32_swap_b32_e32 %" #i ", %" #j "\n\t"
#define OP_CVT(i, j) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %" #j "\n\t"
// dependent chain through DPP: each instruction reads the previous one's result
#define OP_ADD_WSHL_DEP(i, j) "v_add_f32_dpp %" #j ", %" #i ", %" #i " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
#define OP_ADD_DEP(i, j) "v_add_f32 %" #j ", %" #i ", %" #i "\n\t"

#define R4PK(op)                                                                                      \
    asm volatile(op(0, 1) op(1, 2) op(2, 3) op(3, 0) op(0, 1) op(1, 2) op(2, 3) op(3, 0)               \
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]))
#define OP_PKADD(i, j) "v_pk_add_f32 %" #i ", %" #i ", %" #j "\n\t"
#define OP_PKFMA(i, j) "v_pk_fma_f32 %" #i ", %" #i ", %" #j ", %" #i "\n\t"
#define OP_PKMUL(i, j) "v_pk_mul_f32 %" #i ", %" #i ", %" #j "\n\t"

enum { ADD, ADD_WSHL, MOV_WSHL, ADD_RSHL, ADD_RSHL2, MED3, FMA, SWAP, CVT, ADD_WSHL_DEP, ADD_DEP, PKADD, PKFMA, PKMUL, NKIND };
const char* names[] = {"v_add_f32", "v_add_f32_dpp wave_shl:1", "v_mov_b32_dpp wave_shl:1", "v_add_f32_dpp row_shl:1",
                       "v_add_f32_dpp row_shl:2", "v_med3_f32", "v_fma_f32", "v_permlane32_swap", "v_cvt_pk_bf16_f32",
                       "dep. add_dpp wave_shl + s_nop 1", "dep. v_add_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32"};

template <int KIND>
__device__ __forceinline__ void valu_block(float (&x)[8]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    if constexpr (KIND == ADD) R8(OP_ADD);
    if constexpr (KIND == ADD_WSHL) R8(OP_ADD_WSHL);
    if constexpr (KIND == MOV_WSHL) R8(OP_MOV_WSHL);
    if constexpr (KIND == ADD_RSHL) R8(OP_ADD_RSHL);
    if constexpr (KIND == ADD_RSHL2) R8(OP_ADD_RSHL2);
    if constexpr (KIND == MED3) R8(OP_MED3);
    if constexpr (KIND == FMA) R8(OP_FMA);
    if constexpr (KIND == SWAP) R8(OP_SWAP);
    if constexpr (KIND == CVT) R8(OP_CVT);
    if constexpr (KIND == ADD_WSHL_DEP) R8(OP_ADD_WSHL_DEP);
    if constexpr (KIND == ADD_DEP) R8(OP_ADD_DEP);
    if constexpr (KIND >= PKADD) {
        f2 p[4] = {{x[0], x[1]}, {x[2], x[3]}, {x[4], x[5]}, {x[6], x[7]}};
        if constexpr (KIND == PKADD) R4PK(OP_PKADD);
        if constexpr (KIND == PKFMA) R4PK(OP_PKFMA);
        if constexpr (KIND == PKMUL) R4PK(OP_PKMUL);
        for (int i = 0; i < 4; ++i) x[2 * i] = p[i][0], x[2 * i + 1] = p[i][1];
    }
}

// MODE 0: all waves run the VALU stream.  MODE 1: waves 0..3 run a dependent MFMA chain, waves 4..7 the VALU stream.
// MODE 2: all waves run the MFMA chain (reference).
template <int KIND, int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + threadIdx.x * 1e-6f + i;
    f32x16 acc = {};
    i32x4 wa = {0x3f803f80, 0x3f803f80, 0x3f803f80, (int)threadIdx.x}, wb = {0x3c003c00, 1, 2, (int)threadIdx.x};
    const bool do_mfma = MODE == 2 || (MODE == 1 && wave < 4);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa), __builtin_bit_cast(bf16x8, wb), acc, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) valu_block<KIND>(x);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int KIND, int MODE, int WAVES>
void run() {
    const int iters = 400, nb = 256;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, nb * 64 * WAVES * 4);
    hipMalloc(&cyc, nb * WAVES * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, MODE, WAVES>), dim3(nb), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * WAVES);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> valu, mfma;
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < WAVES; ++w) {
            const bool is_mfma = MODE == 2 || (MODE == 1 && w < 4);
            (is_mfma ? mfma : valu).push_back(h[b * WAVES + w] / (is_mfma ? iters * 16.0 : iters * 64.0));
        }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-34s mode %d waves/CU %d: %6.2f cycles/VALU-instr   %6.2f cycles/MFMA\n", names[KIND], MODE, WAVES, med(valu), med(mfma));
    hipFree(out);
    hipFree(cyc);
}

template <int KIND>
void run_kind() {
    run<KIND, 0, 4>();
    run<KIND, 0, 8>();
    run<KIND, 1, 8>();
}

int main() {
    run<ADD, 2, 4>();
    run<ADD, 2, 8>();
    run_kind<ADD>();
    run_kind<ADD_WSHL>();
    run_kind<MOV_WSHL>();
    run_kind<ADD_RSHL>();
    run_kind<ADD_RSHL2>();
    run_kind<MED3>();
    run_kind<FMA>();
    run_kind<SWAP>();
    run_kind<CVT>();
    run_kind<ADD_WSHL_DEP>();
    run_kind<ADD_DEP>();
    run_kind<PKADD>();
    run_kind<PKFMA>();
    run_kind<PKMUL>();
    return 0;
}
