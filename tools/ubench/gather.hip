// Micro-benchmark: how fast can one CU gather 256-B row segments from an L2-resident table, by load shape?
//   A: buffer_load_dword   64 lanes x 4 B  = 1 row  per wave-instruction   (what the rspmm kernels do today)
//   B: buffer_load_dwordx4 64 lanes x 16 B = 4 rows per wave-instruction   (registers)
//   C: global_load_lds_dwordx4             = 4 rows per wave-instruction   (LDS-DMA, then ds_read_b32 per row)
// build: hipcc -O3 --offload-arch=gfx950 -o gather tools/ubench/gather.hip ; run: ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// table shape: default FB15k237 (14541 rows, stride 1024 floats); `./gather big` = 10M rows of 64 floats (2.56 GB, DRAM-bound)
#ifdef BIG
constexpr int kRows = 10000000;
constexpr int kRowStride = 64;
#else
constexpr int kRows = 14541;
constexpr int kRowStride = 1024;
#endif
#ifdef BIG
constexpr int kShift = 8;
#else
constexpr int kShift = 17;
#endif
constexpr int kBlock = 1024;
constexpr int kPerWave = 4096;        // rows gathered per wave

__global__ __launch_bounds__(kBlock) void gather_a(const float* tab, const int* idx, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const int* my = idx + (size_t)wave * kPerWave;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (unsigned)((size_t)kRows * kRowStride * 4 > 0xffffffffull ? 0xffffffffu : (size_t)kRows * kRowStride * 4), 0x00020000);
    float acc = 0;
    for (int i = 0; i < kPerWave; i += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, ((unsigned)my[i + u] /*shift*/ >> kShift) * (kRowStride * 4), 0));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    out[(size_t)wave * 64 + lane] = acc;
}

typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(kBlock) void gather_b(const float* tab, const int* idx, float* out) {
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const int* my = idx + (size_t)wave * kPerWave;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (unsigned)((size_t)kRows * kRowStride * 4 > 0xffffffffull ? 0xffffffffu : (size_t)kRows * kRowStride * 4), 0x00020000);
    float acc = 0;
    for (int i = 0; i < kPerWave; i += 32) {
        uint4v v[8];
        const int mine = (int)((unsigned)my[i + (lane & 31)] /*shift*/ >> kShift);   // one vector load of 32 indices
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = __shfl(mine, u * 4 + q, 64);
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, row * (kRowStride * 4) + j * 16, 0, AUX);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += __builtin_bit_cast(float, v[u].x) + __builtin_bit_cast(float, v[u].y) + __builtin_bit_cast(float, v[u].z) + __builtin_bit_cast(float, v[u].w);
    }
    out[(size_t)wave * 64 + lane] = acc;
}

// H: eight 128-B half rows per wave-instruction (8 lanes x 16 B each): what a 32-column tile would gather
__global__ __launch_bounds__(kBlock) void gather_h(const float* tab, const int* idx, float* out, int half) {
    const int lane = threadIdx.x & 63;
    const int q = lane >> 3, j = lane & 7;
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const int* my = idx + (size_t)wave * kPerWave;
    const char* base = (const char*)tab + half * 128 + j * 16;
    float acc = 0;
    for (int i = 0; i < kPerWave; i += 64) {
        uint4v v[8];
        const unsigned mine = (unsigned)my[i + lane] >> kShift;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned row = (unsigned)__shfl((int)mine, u * 8 + q, 64);
            v[u] = *(const uint4v*)(base + (unsigned long long)row * (kRowStride * 4));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += __builtin_bit_cast(float, v[u].x) + __builtin_bit_cast(float, v[u].y) + __builtin_bit_cast(float, v[u].z) + __builtin_bit_cast(float, v[u].w);
    }
    out[(size_t)wave * 64 + lane] = acc;
}

// LDS-DMA: each wave owns 2 KiB of LDS (two 1-KiB slots = 8 rows in flight), rows land as [4 rows][64 floats]
__global__ __launch_bounds__(kBlock) void gather_c(const float* tab, const int* idx, float* out) {
    __shared__ __attribute__((aligned(16))) float stage[16 * 512];
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, j = lane & 15;
    const int wl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const int* my = idx + (size_t)wave * kPerWave;
    float* slot = stage + wl * 512;
    float acc = 0;
    for (int i = 0; i < kPerWave; i += 32) {
        const int mine = (int)((unsigned)my[i + (lane & 31)] /*shift*/ >> kShift);
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            const int row0 = __shfl(mine, u * 4 + q, 64);
            const int row1 = __shfl(mine, u * 4 + 4 + q, 64);
            __builtin_amdgcn_global_load_lds(tab + (size_t)row0 * kRowStride + j * 4, (__attribute__((address_space(3))) void*)(slot), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(tab + (size_t)row1 * kRowStride + j * 4, (__attribute__((address_space(3))) void*)(slot + 256), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += slot[k * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    out[(size_t)wave * 64 + lane] = acc;
}


// A2: variant A + what the rspmm kernel adds per edge: packed word (shift + mul for the offset, and-mask for the
//     relation row), one ds_read_b32 from a 121-KB LDS table, multiply-add.
__global__ __launch_bounds__(kBlock) void gather_a2(const float* tab, const int* idx, float* out) {
    extern __shared__ float rel[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 474 * 64; i += kBlock) rel[i] = 1.0f + (i & 7);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const unsigned* my = (const unsigned*)idx + (size_t)wave * kPerWave;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (unsigned)((size_t)kRows * kRowStride * 4 > 0xffffffffull ? 0xffffffffu : (size_t)kRows * kRowStride * 4), 0x00020000);
    const char* rl = (const char*)rel + lane * 4;
    float acc = 0;
    for (int i = 0; i < kPerWave; i += 8) {
        float v[8], rv[8]; unsigned m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m[u] = my[i + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, (m[u] /*shift*/ >> kShift) * (kRowStride * 4), 0));
#pragma unroll
        for (int u = 0; u < 8; ++u) rv[u] = *(const float*)(rl + (m[u] & (kShift == 17 ? 0x1ff00u : 0x0u)));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u] * rv[u];
    }
    out[(size_t)wave * 64 + lane] = acc;
}

// A3: A2 cut into chunks of 88 edges: a dependent descriptor load in front of every chunk, a row store behind it.
__global__ __launch_bounds__(kBlock) void gather_a3(const float* tab, const int* idx, const int4* desc, float* out, float* rows) {
    extern __shared__ float rel[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 474 * 64; i += kBlock) rel[i] = 1.0f + (i & 7);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (unsigned)((size_t)kRows * kRowStride * 4 > 0xffffffffull ? 0xffffffffu : (size_t)kRows * kRowStride * 4), 0x00020000);
    const char* rl = (const char*)rel + lane * 4;
    float tot = 0;
    const int nchunk = kPerWave / 88;
    for (int c = 0; c < nchunk; ++c) {
        const int4 d = desc[(size_t)wave * nchunk + c];
        const unsigned* my = (const unsigned*)idx + d.x;
        float acc = 0;
        for (int i = 0; i < d.y; i += 8) {
            float v[8], rv[8]; unsigned m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) m[u] = my[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, (m[u] /*shift*/ >> kShift) * (kRowStride * 4), 0));
#pragma unroll
            for (int u = 0; u < 8; ++u) rv[u] = *(const float*)(rl + (m[u] & (kShift == 17 ? 0x1ff00u : 0x0u)));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u] * rv[u];
        }
        rows[(size_t)d.z * kRowStride + lane] = acc;
        tot += acc;
    }
    out[(size_t)wave * 64 + lane] = tot;
}

int main() {
    int n_cu = 256;
    const int blocks = n_cu, waves = blocks * 16;
    std::vector<int> h((size_t)waves * kPerWave);
    srand(1);
    for (auto& v : h) {
#ifdef BIG
        v = (int)((((unsigned)rand() * 32768u + (unsigned)rand()) % kRows) << 8);   // row << 8 (no relation field)
#else
        v = ((rand() % kRows) << 17) | ((rand() % 474) << 8);   // packed: row | relation
#endif
    }
    float *tab, *out; int* idx;
    hipMalloc(&tab, (size_t)kRows * kRowStride * 4);
    hipMemset(tab, 0, (size_t)kRows * kRowStride * 4);
    hipMalloc(&out, (size_t)waves * 64 * 4);
    hipMalloc(&idx, h.size() * 4);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double bytes = (double)waves * kPerWave * 256;
    const int nchunk = kPerWave / 88;
    std::vector<int4> hd((size_t)waves * nchunk);
    for (int w = 0; w < waves; ++w) for (int c = 0; c < nchunk; ++c) hd[(size_t)w * nchunk + c] = make_int4(w * kPerWave + c * 88, 88, rand() % kRows, 0);
    int4* desc; hipMalloc(&desc, hd.size() * sizeof(int4)); hipMemcpy(desc, hd.data(), hd.size() * sizeof(int4), hipMemcpyHostToDevice);
    float* rows; hipMalloc(&rows, (size_t)kRows * kRowStride * 4);
    hipFuncSetAttribute((const void*)gather_a2, hipFuncAttributeMaxDynamicSharedMemorySize, 474 * 256);
    hipFuncSetAttribute((const void*)gather_a3, hipFuncAttributeMaxDynamicSharedMemorySize, 474 * 256);
    for (int pol = 0; pol < 7; ++pol) {     // cache policy bits of the gather: 1 = sc0, 2 = nt, 16 = sc1
        const int aux[7] = {0, 1, 2, 3, 16, 17, 18};
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            switch (pol) {
                case 0: hipLaunchKernelGGL(gather_b<0>, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out); break;
                case 1: hipLaunchKernelGGL(gather_b<1>, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out); break;
                case 2: hipLaunchKernelGGL(gather_b<2>, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out); break;
                case 3: hipLaunchKernelGGL(gather_b<3>, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out); break;
                case 4: hipLaunchKernelGGL(gather_b<16>, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out); break;
                case 5: hipLaunchKernelGGL(gather_b<17>, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out); break;
                case 6: hipLaunchKernelGGL(gather_b<18>, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out); break;
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("dwordx4 gather, cache policy aux=%d: %.3f ms  %.2f TB/s  %.1f GB/s/CU (%s)\n", aux[pol], ms, bytes / ms / 1e9, bytes / ms / 1e6 / n_cu, hipGetErrorString(hipGetLastError()));
        }
    }
    for (int half = 0; half < 2; ++half) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(gather_h, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out, half);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("variant H (128-B half rows, half %d): %.3f ms  %.2f TB/s of requested bytes (%s)\n", half, ms, bytes / 2 / ms / 1e9, hipGetErrorString(hipGetLastError()));
        }
    }
    for (int which = 0; which < 5; ++which) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (which == 0) hipLaunchKernelGGL(gather_a, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out);
            if (which == 1) hipLaunchKernelGGL(gather_b<0>, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out);
            if (which == 2) hipLaunchKernelGGL(gather_c, dim3(blocks), dim3(kBlock), 0, 0, tab, idx, out);
            if (which == 3) hipLaunchKernelGGL(gather_a2, dim3(blocks), dim3(kBlock), 474 * 256, 0, tab, idx, out);
            if (which == 4) hipLaunchKernelGGL(gather_a3, dim3(blocks), dim3(kBlock), 474 * 256, 0, tab, idx, desc, out, rows);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("variant %d: %.3f ms  %.2f TB/s  %.1f GB/s/CU (%s)\n", which, ms, bytes / ms / 1e9, bytes / ms / 1e6 / n_cu, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
