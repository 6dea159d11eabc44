// Probe: in which order does v_mfma_f32_16x16x4_f32 accumulate its four k-steps, and does every step round like fmaf?
// (v_mfma_f32_32x32x2_f32 is known here to be the chain acc = fmaf(a0, b0, acc); acc = fmaf(a1, b1, acc): control.)
// build: hipcc -O2 --offload-arch=gfx950 -o mfma_order tools/ubench/mfma_order.hip ; run: ./mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A[16][4], B[4][16], C[16][16] row-major -> D[16][16]
__global__ void k16(const float *A, const float *B, const float *C, float *D) {
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * kq + r) * 16 + i];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + kq], B[kq * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + i] = acc[r];
}

// two chained instructions (K = 8): is the chain across instructions k = 0..3 then 4..7?
__global__ void k16x2(const float *A, const float *B, const float *C, float *D) {
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * kq + r) * 16 + i];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 8 + kq], B[kq * 16 + i], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 8 + 4 + kq], B[(4 + kq) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + i] = acc[r];
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> mant(1.0f, 2.0f);
    std::uniform_int_distribution<int> expo(-12, 12), sign(0, 1), zero(0, 5);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 16 * 8 * 4); hipMalloc(&dB, 8 * 16 * 4); hipMalloc(&dC, 256 * 4); hipMalloc(&dD, 256 * 4);
    long n = 0, bad_seq = 0, bad_rev = 0, bad_pair = 0, bad_exact = 0, bad_seq8 = 0, bad_01 = 0, n01 = 0;
    for (int trial = 0; trial < 400; ++trial) {
        const bool unit_a = trial >= 200;          // A in {0, 1}: the adjacency case
        std::vector<float> A(16 * 8), B(8 * 16), C(256), D(256), D2(256);
        auto val = [&]() { return (sign(rng) ? -1.0f : 1.0f) * std::ldexp(mant(rng), expo(rng)); };
        for (auto &v : A) v = unit_a ? (zero(rng) ? 1.0f : 0.0f) : val();
        for (auto &v : B) v = val();
        for (auto &v : C) v = trial % 3 == 0 ? 0.0f : val();
        std::vector<float> A4(16 * 4);
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A4[i * 4 + k] = A[i * 8 + k];
        hipMemcpy(dA, A4.data(), 64 * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), 64 * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 256 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(dA, A.data(), 128 * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), 128 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16x2, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D2.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        for (int m = 0; m < 16; ++m)
            for (int c = 0; c < 16; ++c) {
                const float c0 = C[m * 16 + c];
                float seq = c0, rev = c0, seq8 = c0, add01 = c0;
                for (int k = 0; k < 4; ++k) seq = fmaf(A[m * 8 + k], B[k * 16 + c], seq);
                for (int k = 3; k >= 0; --k) rev = fmaf(A[m * 8 + k], B[k * 16 + c], rev);
                for (int k = 0; k < 8; ++k) seq8 = fmaf(A[m * 8 + k], B[k * 16 + c], seq8);
                // the adjacency form written as the reference writes it: acc = acc + y for the edges that exist
                for (int k = 0; k < 8; ++k) if (A[m * 8 + k] != 0.0f) add01 = add01 + B[k * 16 + c];
                const float p01 = fmaf(A[m * 8 + 1], B[16 + c], A[m * 8] * B[c]);
                const float p23 = fmaf(A[m * 8 + 3], B[48 + c], A[m * 8 + 2] * B[32 + c]);
                const float pair = c0 + (p01 + p23);
                long double ex = c0;
                for (int k = 0; k < 4; ++k) ex += (long double)A[m * 8 + k] * (long double)B[k * 16 + c];
                const float got = D[m * 16 + c];
                ++n;
                bad_seq += bits(got) != bits(seq);
                bad_rev += bits(got) != bits(rev);
                bad_pair += bits(got) != bits(pair);
                bad_exact += bits(got) != bits((float)ex);
                bad_seq8 += bits(D2[m * 16 + c]) != bits(seq8);
                if (unit_a) { ++n01; bad_01 += bits(D2[m * 16 + c]) != bits(add01); }
            }
    }
    printf("v_mfma_f32_16x16x4_f32 over %ld outputs: mismatches vs sequential fmaf k=0..3: %ld | reversed: %ld | pairwise: %ld | "
           "exact-then-round: %ld\n", n, bad_seq, bad_rev, bad_pair, bad_exact);
    printf("two chained instructions vs sequential fmaf k=0..7: %ld mismatches of %ld\n", bad_seq8, n);
    printf("A in {0,1}: two chained instructions vs `acc = acc + y` over the existing edges: %ld mismatches of %ld\n", bad_01, n01);
    return (bad_seq == 0 && bad_seq8 == 0 && bad_01 == 0) ? 0 : 1;
}
