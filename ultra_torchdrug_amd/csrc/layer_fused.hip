// layer_fused.hip -- one whole Bellman-Ford layer of the entity stack in inference as ONE launch (fourth translation unit of
// libultra_rspmm.so): the rspmm of csrc/rowgroup.inc with the layer's epilogue INSIDE its row loop.
//
//     out = [input +] relu(LayerNorm(Linear_{128->64}(cat[input, rspmm_{add,mul}(adjacency, relation, input) + boundary])))
//
// = GeneralizedRelationalConv*.message_and_aggregate + combine (/root/reference/ultra/layer.py:298-392, the rspmm call at :357,
// `update + boundary` at :358, combine at :386-392) + the caller's shortcut (ultra/model.py:126-127).  As two launches
// (ultra_rspmm_forward_boundary_f32, ultra_combine_forward_f32) the (N, B, 64) tensor `update` is written by the first and read
// by the second: 2 of the 5 row-sized streams of a layer ("fusing into the rspmm row tile removes one full read
// + write of (N, F) per layer").  On S-stress (10 M nodes, B = 1: 2 queries) that is 10.2 of 72.9 GB per layer.
//
// The row loop is rowgroup_kernel's (one destination row per 16 / 32 / 64-lane group, the whole 16-edge window of gathers in
// flight, one exposed round trip per row); sum = add, mul = mul, forward only.  A finished row of a group is G / 16 rows of the
// epilogue (one per query block of 64 columns): a wave finishes FOUR epilogue rows per iteration, lanes 16 e .. 16 e + 15
// holding row e.  They are staged, with the rows' own `input` segments, in a wave-private LDS tile; every fourth iteration the
// wave owns 16 complete epilogue rows and runs the epilogue on them:
//   * GEMM on the exact-f32 matrix cores, v_mfma_f32_16x16x4_f32 with K = (in[s], up[s], in[s+1], up[s+1]): the instruction is
//     a sequential fmaf chain (tools/ubench/mfma_order.hip), so every output element is the chain  bias, in[0], up[0], in[1],
//     up[1], ...  of combine_kernel (v_mfma_f32_32x32x2_f32, K = (in[s], up[s])) and of oracle_combine_forward: the same bits;
//     the 64 x 128 weight lives in LDS once per workgroup, already in B-operand order (one ds_read_b128 = four steps);
//   * LayerNorm as combine_kernel does it: two lanes per row, each the sequential sum of 32 columns, added once; relu; shortcut;
//   * the 16 finished rows leave with four coalesced 1-KiB buffer stores (each lane stores its own rows: the offsets are the
//     ones the rspmm kernel would have stored `update` at).
// Results are bit-identical to the two launches (tests/test_layer_fused_gpu.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "relgraph_dense.h"
#include "ultra_rspmm.h"

extern thread_local int ultra_detail_last_hip_error;

namespace {

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) {                         \
            ultra_detail_last_hip_error = (int)_e;      \
            (void)hipGetLastError();                    \
            return ULTRA_ERR_HIP;                       \
        }                                               \
    } while (0)

typedef float qf4 __attribute__((ext_vector_type(4)));
typedef float qf2 __attribute__((ext_vector_type(2)));
typedef uint32_t qu4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const qf4 *lds_qf4_ptr;
typedef __attribute__((address_space(3))) qf4 *lds_qf4_wptr;
typedef __attribute__((address_space(3))) qf2 *lds_qf2_wptr;
typedef __attribute__((address_space(3))) float *lds_f_wptr;
typedef __attribute__((address_space(3))) const float *lds_f_ptr;
typedef __attribute__((address_space(3))) char *lds_char_ptr;

// LDS accesses at integer byte addresses (dynamic LDS starts at 0 in this kernel: no static __shared__; the kernel traps if
// that ever changes) -- table bases stay in the instructions' immediate offsets
__device__ __forceinline__ qf4 lds_read4(uint32_t byte_addr) { return *(lds_qf4_ptr)byte_addr; }
__device__ __forceinline__ float lds_read1f(uint32_t byte_addr) { return *(lds_f_ptr)byte_addr; }
__device__ __forceinline__ void lds_write4(uint32_t byte_addr, qf4 v) { *(lds_qf4_wptr)byte_addr = v; }
__device__ __forceinline__ void lds_write2(uint32_t byte_addr, qf2 v) { *(lds_qf2_wptr)byte_addr = v; }
__device__ __forceinline__ void lds_write1(uint32_t byte_addr, float v) { *(lds_f_wptr)byte_addr = v; }

constexpr int kXcd = 8;
constexpr int kMaxLdsBytes = 156 * 1024;
constexpr int kLfBlock = 512;                    // 8 waves per CU, as rowgroup_kernel
constexpr int kLfWaves = kLfBlock / 64;
// a staged epilogue row: in[even columns] 32 | in[odd] 32 | up[even] 32 | up[odd] 32 | pad 4 floats -- lane group k of the
// MFMA's A operand (k = 0: in[s], 1: up[s], 2: in[s + 1], 3: up[s + 1], s even) reads 32 CONTIGUOUS floats
constexpr int kLfStride = 132;
constexpr uint32_t kLfTileBytes = 16 * kLfStride * 4;
constexpr uint32_t kOffW = 16;                                  // [4 n][8 t8][64 lanes][4] floats: the B operands
constexpr uint32_t kOffGamma = kOffW + 64 * 128 * 4;
constexpr uint32_t kOffBeta = kOffGamma + 256;
constexpr uint32_t kOffBias = kOffBeta + 256;
constexpr uint32_t kOffTiles = kOffBias + 256;
constexpr uint32_t kOffRel = kOffTiles + kLfWaves * kLfTileBytes;        // relation rows held in LDS (if any)
static_assert(kOffRel % 16 == 0, "16-byte aligned LDS tables");

constexpr int kRelL2 = 0, kRelLds = 1, kRelPart = 2;

struct LayerParams {
    const int32_t *row_ptr;     // [n_rows + 1]
    const int32_t *col;         // [E] gathered row of every edge
    const int32_t *rel;         // [E]
    const float *weight;        // [E] or NULL
    const float *relation;      // [n_rel, F]
    const float *gather;        // [n_rows, F]: the layer's input (gathered AND read at the row itself)
    const int32_t *bnode;       // sparse boundary: node of every query, or NULL (no boundary term)
    const float *bvec;          // [F]
    float *out;                 // [n_rows, F], not aliasing `gather`
    const float *lin_w;         // [64, 128]
    const float *lin_b, *gamma, *beta;
    float eps;
    int relu, shortcut;
    long long F;
    int n_rows, n_rel, n_rel_lds, n_tiles, split, n_slots, blocks_per_label;
    // SCORE (the LAST layer of full-batch evaluation): the score head runs on the finished rows inside the same flush and only
    // the scores leave -- s_w1 [128, 128] (its hidden half [:, :64] is used here), s_qbias [n_query, 128] = the queries' share of
    // the head's first layer (score_qbias_kernel), s_w2 [128], s_b2 [1], score [n_query, n_rows]
    const float *s_w1, *s_qbias, *s_w2;
    const float *s_b2;      // [1]
    float *score;
    int n_query;
};

// LDS of the SCORE form, behind the tiles (it reads its relation rows through L2)
constexpr uint32_t kOffSW1 = kOffRel;                       // [8 n][4 t4][64 lanes][4] floats: B operands of the head's first layer
constexpr uint32_t kOffSW2 = kOffSW1 + 128 * 64 * 4;        // [128]
constexpr uint32_t kOffSMeta = kOffSW2 + 512;               // [waves][16] byte offset of every staged row's score (or ~0)
constexpr uint32_t kOffSC = kOffSMeta + kLfWaves * 64;      // [n_query][128] the queries' share
constexpr int kScoreMaxQueries = 32;

template <bool UNIT_W, int REL, int G, bool SCORE = false>
__global__ __launch_bounds__(kLfBlock) void rowgroup_layer_kernel(const LayerParams p) {
    static_assert(!SCORE || REL == kRelL2, "the score form keeps its LDS for the head's weights");
    constexpr int U = 16;
    constexpr bool REL_LDS = REL == kRelLds, REL_PART = REL == kRelPart;
    constexpr int W = 4 * G;                                // columns per tile
    constexpr int GPW = 64 / G;                             // groups (destination rows) per wave
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    if ((uint32_t)(uintptr_t)(lds_char_ptr)lds_raw != 0u) __builtin_trap();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane & (G - 1);
    const int gw = lane / G;                                // group inside the wave
    const int bp_base = (lane & (64 - G)) * 4;              // byte address (ds_bpermute) of the group's lane 0
    const int label = blockIdx.x % kXcd;
    const int bl = blockIdx.x / kXcd;
    const int nb = p.blocks_per_label;
    const long long F = p.F;
    const qf4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t tile = kOffTiles + (uint32_t)wave * kLfTileBytes;
    const int i16 = lane & 15, kq = lane >> 4;

    // ---- once per workgroup: the weight in B-operand order, gamma | beta | bias
    for (int idx = threadIdx.x; idx < 64 * 128; idx += kLfBlock) {
        const int u = idx & 3, l = (idx >> 2) & 63, t8 = (idx >> 8) & 7, n = idx >> 11;
        const int k = l >> 4, j = l & 15;
        const int s = 2 * (4 * t8 + u) + (k >> 1);
        lds_raw[kOffW / 4 + idx] = p.lin_w[(16 * n + j) * 128 + 64 * (k & 1) + s];
    }
    if (threadIdx.x < 64) {
        lds_raw[kOffGamma / 4 + threadIdx.x] = p.gamma != nullptr ? p.gamma[threadIdx.x] : 1.0f;
        lds_raw[kOffBeta / 4 + threadIdx.x] = p.gamma != nullptr ? p.beta[threadIdx.x] : 0.0f;
        lds_raw[kOffBias / 4 + threadIdx.x] = p.lin_b[threadIdx.x];
    }
    if constexpr (SCORE) {
        for (int idx = threadIdx.x; idx < 128 * 64; idx += kLfBlock) {
            const int u = idx & 3, l = (idx >> 2) & 63, t4 = (idx >> 8) & 3, n = idx >> 10;
            const int k = l >> 4, j = l & 15;
            // step t of the chain  c, hid[0], hid[32], hid[1], hid[33], ...  (score_kernel's order): K = (hid[s], hid[32 + s], hid[s + 1],
            // hid[33 + s]), s = 2 t
            lds_raw[kOffSW1 / 4 + idx] = p.s_w1[(16 * n + j) * 128 + 32 * (k & 1) + (k >> 1) + 2 * (4 * t4 + u)];
        }
        if (threadIdx.x < 128) lds_raw[kOffSW2 / 4 + threadIdx.x] = p.s_w2[threadIdx.x];
        if (threadIdx.x < kLfWaves * 16) reinterpret_cast<uint32_t *>(lds_raw)[kOffSMeta / 4 + threadIdx.x] = 0xffffffffu;
        for (int idx = threadIdx.x; idx < p.n_query * 128; idx += kLfBlock) lds_raw[kOffSC / 4 + idx] = p.s_qbias[idx];
    }
    __syncthreads();

    for (int s = label; s < p.n_slots; s += kXcd) {
        const int tile_id = s / p.split;
        const int part = s - tile_id * p.split;
        const long long col0 = (long long)tile_id * W + sub * 4;         // F % W == 0 (dispatch condition): every lane is active
        const uint32_t row_bytes = (uint32_t)F * 4u;
        const char *gather_lane = reinterpret_cast<const char *>(p.gather + col0);
        const char *relation_lane = reinterpret_cast<const char *>(p.relation + col0);
        const uint32_t lds_lane = kOffRel + (uint32_t)sub * 16u;
        if constexpr (REL_LDS || REL_PART) {
            __syncthreads();                                // the previous slot's readers are done
            const int total = p.n_rel_lds * W;
            for (int i = threadIdx.x; i < total; i += kLfBlock) {
                const int r = i / W;
                lds_raw[kOffRel / 4 + i] = p.relation[(long long)r * F + (long long)tile_id * W + (i % W)];
            }
            __syncthreads();
        }
        int b_node = -1;
        qf4 b_val = zero4;
        if (p.bnode != nullptr) {
            b_node = p.bnode[col0 / 64];
            b_val = *reinterpret_cast<const qf4 *>(p.bvec + col0);
        }
        const long long rows_per_part = ((long long)p.n_rows + p.split - 1) / p.split;
        const long long row_begin = (long long)part * rows_per_part;
        const long long row_end = row_begin + rows_per_part < (long long)p.n_rows ? row_begin + rows_per_part : (long long)p.n_rows;
        const long long stride = (long long)nb * (kLfBlock / G);
        // the wave's first row is wave-uniform: the loop and the epilogue run with all 64 lanes, groups past the end carry empty rows
        long long wrow = row_begin + (long long)bl * (kLfBlock / G) + (long long)wave * GPW;

        int beg, end, beg1, end1, n_col = 0, n_rel_id = 0;
        float n_w = 1.0f;
        auto load_ptrs = [&](long long r_, int &b_, int &e_) {
            b_ = 0; e_ = 0;
            if (r_ < row_end) {
                b_ = p.row_ptr[r_];
                e_ = p.row_ptr[r_ + 1];
            }
        };
        auto load_window = [&](int first, int last, int &c, int &r, float &w) {
            const int e = first + sub;
            c = 0; r = 0; w = 1.0f;
            if (sub < 16 && e < last) {
                c = p.col[e];
                r = p.rel[e];
                if constexpr (!UNIT_W) w = p.weight[e];
            }
        };
        load_ptrs(wrow + gw, beg, end);
        load_ptrs(wrow + gw + stride, beg1, end1);
        load_window(beg, end, n_col, n_rel_id, n_w);
        const unsigned long long part_bytes = (unsigned long long)(row_end - row_begin) * row_bytes;
        const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char *>(p.out) + (unsigned long long)row_begin * row_bytes, 0, (int)(uint32_t)part_bytes, 0x00020000);
        const uint32_t lane_off = (uint32_t)(col0 * 4);
        __amdgpu_buffer_rsrc_t rsrc_score = rsrc_out;
        if constexpr (SCORE)
            rsrc_score = __builtin_amdgcn_make_buffer_rsrc(p.score, 0, (int)(uint32_t)((unsigned long long)p.n_query * p.n_rows * 4ull), 0x00020000);
        const int tile_q0 = tile_id * (W / 64);             // first query block of this column tile
        const uint32_t meta = kOffSMeta + (uint32_t)wave * 64u;
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): the loop is entered with no load pending
        qf4 xv[U], rv[U];
        float wv[U];
        uint32_t cid[U], rid[U];
        auto issue = [&](int n_here, int w_col, int w_rel, float w_w) {
            // (an opaque copy of the group's base: `base + 4 u` then folds into the instruction's offset field; derived outside
            // the loop the sixteen addresses were hoisted into sixteen registers)
            int bpb = bp_base;
            asm volatile("" : "+v"(bpb));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cid[u] = (uint32_t)__builtin_amdgcn_ds_bpermute(bpb + 4 * u, w_col);
                rid[u] = (uint32_t)__builtin_amdgcn_ds_bpermute(bpb + 4 * u, w_rel);
                wv[u] = 1.0f;
                if constexpr (!UNIT_W)
                    wv[u] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bpb + 4 * u, __builtin_bit_cast(int, w_w)));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < n_here) {
                    xv[u] = *reinterpret_cast<const qf4 *>(gather_lane + (unsigned long long)cid[u] * row_bytes);
                    if constexpr (REL_LDS) rv[u] = lds_read4(lds_lane + rid[u] * (uint32_t)(W * 4));
                    else if constexpr (REL_PART) {
                        if (rid[u] >= (uint32_t)p.n_rel_lds)
                            rv[u] = *reinterpret_cast<const qf4 *>(relation_lane + (unsigned long long)rid[u] * row_bytes);
                    } else rv[u] = *reinterpret_cast<const qf4 *>(relation_lane + (unsigned long long)rid[u] * row_bytes);
                }
            }
        };
        auto reduce = [&](int n_here, qf4 &acc) {
            if constexpr (REL_PART) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (u < n_here && rid[u] < (uint32_t)p.n_rel_lds) rv[u] = lds_read4(lds_lane + rid[u] * (uint32_t)(W * 4));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < n_here) {
                    qf4 y = rv[u] * xv[u];
                    if constexpr (!UNIT_W) y = wv[u] * y;
                    acc = acc + y;
                }
            }
        };

        // ---- the epilogue of the wave's 16 staged rows (k_filled x 4 of them are real; the others hold stale rows whose
        // results are stored nowhere: an epilogue row depends on its own staged row only)
        uint32_t st_off[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};      // (only ever indexed by constants: registers)
        const uint32_t s_addr = tile + (uint32_t)(kq * kLfStride + 2 * i16) * 4u;                             // staging: rows 4 k + kq
        auto flush = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the staging writes have landed
            // per-lane LDS base addresses, derived HERE from an opaque copy of the tile address: written out per access (or
            // derived outside) the compiler hoists forty of them out of the row loop and spills the gather window for them;
            // everything else is an immediate offset of the instruction
            uint32_t tl = tile;
            asm volatile("" : "+v"(tl));
            uint32_t bias_addr = kOffBias + (uint32_t)i16 * 4u;
            asm volatile("" : "+v"(bias_addr));
            const uint32_t a_addr = tl + (uint32_t)(i16 * kLfStride + 64 * (kq & 1) + 32 * (kq >> 1)) * 4u;     // A operand: row i16, group kq
            uint32_t w_addr = kOffW + (uint32_t)lane * 16u;                                                     // B operand
            asm volatile("" : "+v"(w_addr));
            const uint32_t d_addr = tl + (uint32_t)(4 * kq * kLfStride + i16) * 4u;                             // D: rows 4 kq + r, column i16
            const uint32_t zrow = tl + (uint32_t)((lane >> 1) * kLfStride + 64 + 32 * (lane & 1)) * 4u;         // LayerNorm: row lane / 2, half lane % 2
            const uint32_t irow = tl + (uint32_t)((lane >> 1) * kLfStride + 16 * (lane & 1)) * 4u;
            uint32_t gb_addr = kOffGamma + (uint32_t)(32 * (lane & 1)) * 4u;
            asm volatile("" : "+v"(gb_addr));
            const uint32_t o_addr = tl + (uint32_t)(kq * kLfStride + 4 * i16) * 4u;                             // rows 4 q + kq, this lane's columns
            qf4 acc[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float b = lds_read1f(bias_addr + (uint32_t)(64 * n));
                acc[n] = qf4{b, b, b, b};
            }
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
                const qf4 a = lds_read4(a_addr + (uint32_t)t8 * 16u);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const qf4 b = lds_read4(w_addr + (uint32_t)((n * 8 + t8) * 64) * 16u);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[n], 0, 0, 0);
                }
                // (keeps the scheduler from hoisting all 32 B-operand reads -- 128 VGPRs -- in front of the first MFMA)
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // every A read done before `up` is overwritten
            // z = Linear(cat[in, up]) in natural column order over the consumed `up` half: acc[n][r] = z[row 4 kq + r][16 n + i16]
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lds_write1(d_addr + (uint32_t)((r * kLfStride + 64 + 16 * n) * 4), acc[n][r]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // LayerNorm + ReLU + shortcut: two lanes per row, 32 columns each, sums in column order (combine_kernel's expressions)
            if (lane < 32) {
                // streamed from the tile in three passes (sum, squared deviations, finish): a row's 32 values held in registers
                // through all of it would not fit beside the gather window that is in flight around this call
                const int ln_half = lane & 1;
                float mean = 0.0f, inv = 1.0f;
                if (p.gamma != nullptr) {
                    float sm = 0.0f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const qf4 d = lds_read4(zrow + (uint32_t)q * 16u);
                        sm = sm + d.x; sm = sm + d.y; sm = sm + d.z; sm = sm + d.w;
                    }
                    const float so = __shfl_xor(sm, 1, 64);
                    mean = (ln_half == 0 ? sm + so : so + sm) * (1.0f / 64.0f);
                    float ss = 0.0f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const qf4 d = lds_read4(zrow + (uint32_t)q * 16u);
                        float dlt = d.x - mean; ss = ss + dlt * dlt;
                        dlt = d.y - mean; ss = ss + dlt * dlt;
                        dlt = d.z - mean; ss = ss + dlt * dlt;
                        dlt = d.w - mean; ss = ss + dlt * dlt;
                    }
                    const float sso = __shfl_xor(ss, 1, 64);
                    const float var = (ln_half == 0 ? ss + sso : sso + ss) * (1.0f / 64.0f);
                    inv = 1.0f / sqrtf(var + p.eps);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    qf4 v = lds_read4(zrow + (uint32_t)q * 16u);
                    if (p.gamma != nullptr) {
                        const qf4 gq = lds_read4(gb_addr + (uint32_t)(16 * q));
                        const qf4 bq = lds_read4(gb_addr + (uint32_t)(kOffBeta - kOffGamma + 16 * q));
                        v.x = ((v.x - mean) * inv) * gq.x + bq.x;
                        v.y = ((v.y - mean) * inv) * gq.y + bq.y;
                        v.z = ((v.z - mean) * inv) * gq.z + bq.z;
                        v.w = ((v.w - mean) * inv) * gq.w + bq.w;
                    }
                    if (p.relu) {
                        v.x = !(v.x <= 0.0f) ? v.x : 0.0f; v.y = !(v.y <= 0.0f) ? v.y : 0.0f;
                        v.z = !(v.z <= 0.0f) ? v.z : 0.0f; v.w = !(v.w <= 0.0f) ? v.w : 0.0f;
                    }
                    if (p.shortcut) {
                        // the row's input: even columns at [0, 32), odd columns at [32, 64) of the staged row; columns 4 q .. 4 q + 3
                        // of this half are even entries 2 q, 2 q + 1 and odd entries 2 q, 2 q + 1
                        const qf2 ev = *(__attribute__((address_space(3))) const qf2 *)(irow + (uint32_t)(8 * q));
                        const qf2 od = *(__attribute__((address_space(3))) const qf2 *)(irow + 128u + (uint32_t)(8 * q));
                        v.x = v.x + ev.x; v.y = v.y + od.x; v.z = v.z + ev.y; v.w = v.w + od.y;
                    }
                    lds_write4(zrow + (uint32_t)q * 16u, v);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (SCORE) {
                // ---- the score head on the 16 finished rows (score_kernel's arithmetic, csrc/dense.inc):
                //   h[o] = relu(c[query][o] + W1[o, :64] . hidden)   chain  c, hid[0], hid[32], hid[1], hid[33], ...  on the matrix cores
                //   score = b2 + sum over o ascending of h[o] w2[o]   one lane per row, fmaf chain
                const uint32_t a2_addr = tl + (uint32_t)(i16 * kLfStride + 64 + 32 * (kq & 1)) * 4u;
                uint32_t w1_addr = kOffSW1 + (uint32_t)lane * 16u;
                uint32_t c_addr = kOffSC + (uint32_t)(tile_q0 * 128 + i16) * 4u;
                asm volatile("" : "+v"(w1_addr), "+v"(c_addr));            // (one base each + immediates: see `tl`)
                const bool odd = (kq >> 1) != 0;
                // two passes of 64 outputs (16 accumulator registers at a time beside the gather window in flight); the first pass
                // writes h[0 .. 63] over the rows' consumed input / update halves, the second h[64 .. 127] over the hidden rows it
                // has just read
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    qf4 acc2[4];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r)        // D row 4 kq + r = iteration kq, staged row r: query block r % (G / 16) of the tile
                            acc2[n][r] = lds_read1f(c_addr + (uint32_t)(((r % (G / 16)) * 128 + 64 * half + 16 * n) * 4));
#pragma unroll
                    for (int m = 0; m < 8; ++m) {             // steps t = 2 m, 2 m + 1: hidden columns 4 m .. 4 m + 3 of the lane's half
                        const qf4 a = lds_read4(a2_addr + (uint32_t)m * 16u);
                        const float a0 = odd ? a.y : a.x, a1 = odd ? a.w : a.z;
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            const qf2 b = *(__attribute__((address_space(3))) const qf2 *)(w1_addr + (uint32_t)((((4 * half + n) * 4 + (m >> 1)) * 64) * 16 + (m & 1) * 8));
                            acc2[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b.x, acc2[n], 0, 0, 0);
                            acc2[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b.y, acc2[n], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (second pass: the hidden rows are consumed)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc2[n][r];
                            lds_write1(tl + (uint32_t)((4 * kq + r) * kLfStride + 64 * half + 16 * n + i16) * 4u, !(v <= 0.0f) ? v : 0.0f);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < 16) {
                    float sc = p.s_b2[0];
                    const uint32_t hrow = tl + (uint32_t)(lane * kLfStride) * 4u;
#pragma unroll 8
                    for (int o4 = 0; o4 < 32; ++o4) {
                        const qf4 v = lds_read4(hrow + (uint32_t)o4 * 16u), w2 = lds_read4(kOffSW2 + (uint32_t)o4 * 16u);
                        sc = __builtin_fmaf(v.x, w2.x, sc);
                        sc = __builtin_fmaf(v.y, w2.y, sc);
                        sc = __builtin_fmaf(v.z, w2.z, sc);
                        sc = __builtin_fmaf(v.w, w2.w, sc);
                    }
                    const uint32_t off = *(__attribute__((address_space(3))) const uint32_t *)(meta + (uint32_t)lane * 4u);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, sc), rsrc_score, off, 0, 0);
                    *(__attribute__((address_space(3))) uint32_t *)(meta + (uint32_t)lane * 4u) = 0xffffffffu;      // a ragged last batch must not store it again
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                return;
            }
            // the finished rows: iteration q's four rows in one 1-KiB store, every lane at the offset its own row has in `out`
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const qf4 d = lds_read4(o_addr + (uint32_t)((4 * q * kLfStride + 64) * 4));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(qu4, d), rsrc_out, st_off[q], 0, 2 /* nt */);
                st_off[q] = 0xffffffffu;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the tile is rewritten by the next iterations
        };

        int k_filled = 0;
        while (wrow < row_end) {
            const int cur_beg = beg, cur_end = end;
            int w_col = n_col, w_rel = n_rel_id;
            float w_w = n_w;
            const long long cur_row = wrow + gw;
            wrow += stride;
            int beg2, end2;
            load_ptrs(wrow + gw + stride, beg2, end2);      // row i+2 (zeros past the end)
            load_window(beg1, end1, n_col, n_rel_id, n_w);  // row i+1
            beg = beg1; end = end1;
            beg1 = beg2; end1 = end2;
            const bool live = cur_row < row_end;
            // the row's own input segment (the epilogue's `input` operand) travels with the gathers
            const qf4 inv4 = *reinterpret_cast<const qf4 *>(gather_lane + (unsigned long long)(uint32_t)(live ? cur_row : row_begin) * row_bytes);
            qf4 acc = zero4;
            {
                const int n_here = min(16, cur_end - cur_beg);
                issue(n_here, w_col, w_rel, w_w);
                // the epilogue of the previous four iterations' rows runs HERE, under this row's gathers: the matrix phase is a
                // few microseconds during which the wave would otherwise have nothing in flight, and its four stores are younger
                // than the gathers (vmcnt retires in order: the reduction below does not wait for their acknowledgements)
                if (k_filled == 4) {
                    flush();
                    k_filled = 0;
                }
                reduce(n_here, acc);
            }
            for (int e0 = cur_beg + 16; e0 < cur_end; e0 += 16) {
                load_window(e0, cur_end, w_col, w_rel, w_w);
                const int n_here = min(16, cur_end - e0);
                issue(n_here, w_col, w_rel, w_w);
                reduce(n_here, acc);
            }
            if (p.bnode != nullptr) acc = acc + (((int)cur_row == b_node) ? b_val : zero4);
            // stage the epilogue row (4 k + kq): in even | in odd | up even | up odd
            const uint32_t srow = s_addr + (uint32_t)k_filled * (uint32_t)(4 * kLfStride * 4);
            lds_write2(srow, qf2{inv4.x, inv4.z});
            lds_write2(srow + 128u, qf2{inv4.y, inv4.w});
            lds_write2(srow + 256u, qf2{acc.x, acc.z});
            lds_write2(srow + 384u, qf2{acc.y, acc.w});
            const uint32_t off_now = live ? (uint32_t)((unsigned long long)(cur_row - row_begin) * row_bytes) + lane_off : 0xffffffffu;
#pragma unroll
            for (int q = 0; q < 4; ++q) st_off[q] = (q == k_filled) ? off_now : st_off[q];
            if constexpr (SCORE) {
                // where the staged row's score goes: score[query, node] (the first lane of every 16-lane group says it)
                if (i16 == 0)
                    *(__attribute__((address_space(3))) uint32_t *)(meta + (uint32_t)(4 * k_filled + kq) * 4u) =
                        live ? (uint32_t)(((unsigned long long)(col0 / 64) * (unsigned long long)p.n_rows + (unsigned long long)cur_row) * 4ull) : 0xffffffffu;
            }
            ++k_filled;
        }
        if (k_filled > 0) flush();
    }
}

// ---- ultra_second_layer_sources: a bitmap of the listed nodes, one unlisted node, the remapped source of every edge
__global__ void sources_zero_kernel(uint32_t *bits, long long n_words) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) bits[i] = 0u;
}

__global__ void sources_mark_kernel(const int32_t *row_list, const int32_t *list_count, int list_len, int n_query, uint32_t *bits) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int count = list_count[0] < list_len ? list_count[0] : list_len;
    if (slot >= count) return;
    const int row = row_list[slot];                        // node * n_query + query, or -1
    if (row < 0) return;
    const int node = row / n_query;
    atomicOr(bits + (node >> 5), 1u << (node & 31));
}

__global__ void sources_pick_kernel(const uint32_t *bits, int n_cand, int32_t *c_node) {
    if (threadIdx.x != 0) return;
    int pick = 0;
    for (int node = 0; node < n_cand; ++node)
        if (!((bits[node >> 5] >> (node & 31)) & 1u)) { pick = node; break; }
    c_node[0] = pick;
}

__global__ __launch_bounds__(256) void sources_remap_kernel(const int32_t *col, long long n_edges, long long total, const uint32_t *bits,
                                                            const int32_t *c_node, int32_t *sources) {
    const int c = c_node[0];
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        int v = 0;                                          // (the slack behind the last edge: readable, never used)
        if (e < n_edges) {
            const int u = col[e];
            v = ((bits[u >> 5] >> (u & 31)) & 1u) ? u : c;
        }
        sources[e] = v;
    }
}

// the queries' share of the score head's first layer: c[b, o] = b1[o] + W1[o, 64:] . query[b], the fmaf chain of
// score_query_bias_kernel (csrc/dense.inc): k = 64, 96, 65, 97, ... from the bias
__global__ __launch_bounds__(128) void score_qbias_kernel(const float *query, const float *w1, const float *b1, float *qbias) {
    const int b = blockIdx.x, o = threadIdx.x;
    const float *w = w1 + o * 128 + 64;
    const float *q = query + (long long)b * 64;
    float acc = b1[o];
    for (int s = 0; s < 32; ++s) {
        acc = __builtin_fmaf(q[s], w[s], acc);
        acc = __builtin_fmaf(q[32 + s], w[32 + s], acc);
    }
    qbias[(long long)b * 128 + o] = acc;
}

template <int G, bool SCORE = false>
int launch_layer_g(const LayerParams &p, bool unit_w, int rel, int grid, size_t lds, hipStream_t stream) {
#define ULTRA_LF(UW, RL)                                                                                          \
    do {                                                                                                          \
        auto kern = rowgroup_layer_kernel<UW, RL, G, SCORE>;                                                      \
        static bool attr_set[16] = {};                                                                            \
        int dev = 0;                                                                                              \
        HIP_TRY(hipGetDevice(&dev));                                                                              \
        if (dev >= 0 && dev < 16 && !attr_set[dev]) {                                                             \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        kMaxLdsBytes));                                                           \
            attr_set[dev] = true;                                                                                 \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kLfBlock), lds, stream, p);                                     \
        HIP_TRY(hipGetLastError());                                                                               \
        return ULTRA_OK;                                                                                          \
    } while (0)
    if constexpr (!SCORE) {
        if (unit_w) {
            if (rel == kRelLds) ULTRA_LF(true, kRelLds);
            if (rel == kRelPart) ULTRA_LF(true, kRelPart);
        } else {
            if (rel == kRelLds) ULTRA_LF(false, kRelLds);
            if (rel == kRelPart) ULTRA_LF(false, kRelPart);
        }
    }
    if (unit_w) ULTRA_LF(true, kRelL2);
    ULTRA_LF(false, kRelL2);
#undef ULTRA_LF
}

int gcd_int(int a, int b) {
    while (b) { const int t = a % b; a = b; b = t; }
    return a;
}

bool abi_ok(const ultra_segments *s) {
    return s->struct_bytes == (uint32_t)sizeof(ultra_segments) && s->abi_version == (uint32_t)ULTRA_RSPMM_ABI_VERSION;
}

}  // namespace

extern "C" {

int ultra_layer_forward_supported(const ultra_segments *fwd, int64_t n_query, int64_t n_rel) {
    if (fwd == nullptr || !abi_ok(fwd)) return 0;
    if (fwd->row_ptr == nullptr || fwd->n_pieces != 0 || fwd->n_rows <= 0 || fwd->n_rows > 0x7fffffffLL) return 0;
    if (n_query <= 0 || n_rel <= 0 || n_query * 64 >= (1LL << 30)) return 0;
    return 1;
}

// shared by the two entries: tiling, LDS budget, launch.  `score` != NULL: the last layer with the score head inside.
static int layer_launch(const ultra_segments *fwd, LayerParams &q, int64_t n_query, int64_t n_rel, bool score, hipStream_t s,
                        const int32_t *col_override = nullptr) {
    int n_cu = 0;
    int rc = ultra_detail::persistent_cus(&n_cu);
    if (rc) return rc;
    const long long F = n_query * 64;
    q.row_ptr = fwd->row_ptr; q.col = col_override != nullptr ? col_override : fwd->node_a; q.rel = fwd->rel; q.weight = fwd->weight;
    q.F = F; q.n_rows = (int)fwd->n_rows; q.n_rel = (int)n_rel; q.n_query = (int)n_query;
    // groups as wide as the row allows when the gathered matrix lives in DRAM (launch_rowgroup's rule), a whole number of tiles
    const bool dram = ultra_detail::wide_groups_forced() || (double)fwd->n_rows * (double)F * 4.0 > 256.0 * 1024 * 1024;
    const int group = (dram && F % 256 == 0) ? 64 : ((dram && F % 128 == 0) ? 32 : 16);
    const int width = 4 * group;
    q.n_tiles = (int)(F / width);
    q.split = kXcd / gcd_int(q.n_tiles, kXcd);
    while (((long long)q.n_rows + q.split - 1) / q.split * F * 4 >= (1LL << 32) - 65536 && q.split < (1 << 20)) q.split *= 2;
    q.n_slots = q.n_tiles * q.split;
    q.blocks_per_label = (n_cu + kXcd - 1) / kXcd;
    const int grid = q.blocks_per_label * kXcd;
    const bool unit_w = fwd->weight == nullptr;
    if (score) {
        const size_t lds = kOffSC + (size_t)n_query * 128 * sizeof(float);
        q.n_rel_lds = 0;
        if (group == 64) return launch_layer_g<64, true>(q, unit_w, kRelL2, grid, lds, s);
        if (group == 32) return launch_layer_g<32, true>(q, unit_w, kRelL2, grid, lds, s);
        return launch_layer_g<16, true>(q, unit_w, kRelL2, grid, lds, s);
    }
    // relation rows in the LDS the epilogue leaves: all of them, or the first ones when that is at least a quarter of the table
    const size_t room = (size_t)kMaxLdsBytes - kOffRel;
    const size_t lds_need = (size_t)n_rel * width * sizeof(float);
    int rel = kRelL2;
    q.n_rel_lds = 0;
    size_t lds = kOffRel;
    if (lds_need <= room) {
        rel = kRelLds; q.n_rel_lds = (int)n_rel; lds += lds_need;
    } else {
        const int part_rows = (int)(room / ((size_t)width * sizeof(float)));
        if ((long long)part_rows * 4 >= n_rel) { rel = kRelPart; q.n_rel_lds = part_rows; lds += (size_t)part_rows * width * sizeof(float); }
    }
    if (group == 64) return launch_layer_g<64>(q, unit_w, rel, grid, lds, s);
    if (group == 32) return launch_layer_g<32>(q, unit_w, rel, grid, lds, s);
    return launch_layer_g<16>(q, unit_w, rel, grid, lds, s);
}

static int layer_args_ok(const ultra_segments *fwd, const float *relation, const float *input, const int32_t *boundary_node,
                         const float *boundary_value, int64_t n_query, const float *weight, const float *bias,
                         const float *ln_weight, const float *ln_bias, int64_t n_rel) {
    if (fwd == nullptr) return ULTRA_ERR_NULL_POINTER;
    if (!abi_ok(fwd)) return ULTRA_ERR_ABI;
    if (!ultra_layer_forward_supported(fwd, n_query, n_rel)) return ULTRA_ERR_BAD_SHAPE;
    if (relation == nullptr || input == nullptr || weight == nullptr || bias == nullptr) return ULTRA_ERR_NULL_POINTER;
    if ((boundary_node == nullptr) != (boundary_value == nullptr)) return ULTRA_ERR_NULL_POINTER;
    if (ln_weight != nullptr && ln_bias == nullptr) return ULTRA_ERR_NULL_POINTER;
    if ((reinterpret_cast<uintptr_t>(input) | reinterpret_cast<uintptr_t>(relation) | reinterpret_cast<uintptr_t>(boundary_value)) & 15u)
        return ULTRA_ERR_BAD_SHAPE;
    return ULTRA_OK;
}

int ultra_layer_forward_f32(const ultra_segments *fwd, const float *relation, const float *input, const int32_t *boundary_node,
                            const float *boundary_value, int64_t n_query, const float *weight, const float *bias,
                            const float *ln_weight, const float *ln_bias, float ln_eps, int relu, int shortcut, float *out,
                            int64_t n_rel, void *stream) {
    int rc = layer_args_ok(fwd, relation, input, boundary_node, boundary_value, n_query, weight, bias, ln_weight, ln_bias, n_rel);
    if (rc) return rc;
    if (out == nullptr) return ULTRA_ERR_NULL_POINTER;
    if (out == input || (reinterpret_cast<uintptr_t>(out) & 15u)) return ULTRA_ERR_BAD_SHAPE;      // other workgroups still gather from the input
    LayerParams q{};
    q.relation = relation; q.gather = input; q.bnode = boundary_node; q.bvec = boundary_value; q.out = out;
    q.lin_w = weight; q.lin_b = bias; q.gamma = ln_weight; q.beta = ln_bias; q.eps = ln_eps; q.relu = relu; q.shortcut = shortcut;
    return layer_launch(fwd, q, n_query, n_rel, false, static_cast<hipStream_t>(stream));
}

// The same layer with the gathered row of every edge taken from `sources` instead of the plan (see ultra_second_layer_sources).
int ultra_layer_forward_sources_f32(const ultra_segments *fwd, const int32_t *sources, const float *relation, const float *input,
                                    const int32_t *boundary_node, const float *boundary_value, int64_t n_query, const float *weight,
                                    const float *bias, const float *ln_weight, const float *ln_bias, float ln_eps, int relu,
                                    int shortcut, float *out, int64_t n_rel, void *stream) {
    int rc = layer_args_ok(fwd, relation, input, boundary_node, boundary_value, n_query, weight, bias, ln_weight, ln_bias, n_rel);
    if (rc) return rc;
    if (out == nullptr || sources == nullptr) return ULTRA_ERR_NULL_POINTER;
    if (out == input || (reinterpret_cast<uintptr_t>(out) & 15u)) return ULTRA_ERR_BAD_SHAPE;
    LayerParams q{};
    q.relation = relation; q.gather = input; q.bnode = boundary_node; q.bvec = boundary_value; q.out = out;
    q.lin_w = weight; q.lin_b = bias; q.gamma = ln_weight; q.beta = ln_bias; q.eps = ln_eps; q.relu = relu; q.shortcut = shortcut;
    return layer_launch(fwd, q, n_query, n_rel, false, static_cast<hipStream_t>(stream), sources);
}

// ---- sources of the SECOND layer.  After the sparse first layer (ultra_first_layer_sparse_f32) every row of the layer's output is
// ONE constant vector except the LISTED rows (a few dozen on a graph like S-stress).  The second layer gathers that output once
// per edge -- 100 M random 512-byte rows of which all but a few hundred hold the same bytes.  Pointing every edge whose source is
// NOT listed at ONE fixed unlisted row makes those gathers cache hits: identical values, identical sums, no DRAM gather.
int ultra_second_layer_sources(const int32_t *col, int64_t n_edges, int64_t slack, const int32_t *row_list, const int32_t *list_count,
                               int64_t list_len, int64_t n_query, int64_t n_node, uint32_t *bitmap, int32_t *c_node,
                               int32_t *sources, void *stream) {
    if (n_edges < 0 || slack < 0 || list_len < 0 || n_query <= 0 || n_node <= 0 || n_node > 0x7fffffffLL || n_edges > 0x7fffffffLL)
        return ULTRA_ERR_BAD_SHAPE;
    if (list_len + 1 >= n_node) return ULTRA_ERR_BAD_SHAPE;            // (some node among 0 .. list_len must be unlisted)
    if (col == nullptr || row_list == nullptr || list_count == nullptr || bitmap == nullptr || c_node == nullptr || sources == nullptr)
        return ULTRA_ERR_NULL_POINTER;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n_words = (n_node + 31) / 32;
    hipLaunchKernelGGL(sources_zero_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, bitmap, n_words);
    hipLaunchKernelGGL(sources_mark_kernel, dim3((unsigned)((list_len + 255) / 256 > 0 ? (list_len + 255) / 256 : 1)), dim3(256), 0, s,
                       row_list, list_count, (int)list_len, (int)n_query, bitmap);
    hipLaunchKernelGGL(sources_pick_kernel, dim3(1), dim3(64), 0, s, bitmap, (int)(list_len + 1), c_node);
    const long long total = n_edges + slack;
    const unsigned blocks = (unsigned)((total + 256 * 4 - 1) / (256 * 4) < 65535 * 16 ? (total + 256 * 4 - 1) / (256 * 4) : 65535 * 16);
    hipLaunchKernelGGL(sources_remap_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, col, (long long)n_edges, total, bitmap, c_node, sources);
    HIP_TRY(hipGetLastError());
    return ULTRA_OK;
}

int ultra_layer_score_supported(const ultra_segments *fwd, int64_t n_query, int64_t n_rel) {
    return ultra_layer_forward_supported(fwd, n_query, n_rel) && n_query <= kScoreMaxQueries &&
           n_query * fwd->n_rows * 4 < (1LL << 32) - 65536;
}

int ultra_layer_score_forward_f32(const ultra_segments *fwd, const float *relation, const float *input, const int32_t *boundary_node,
                                  const float *boundary_value, int64_t n_query, const float *weight, const float *bias,
                                  const float *ln_weight, const float *ln_bias, float ln_eps, int relu, int shortcut,
                                  const float *query, const float *w1, const float *b1, const float *w2, const float *b2,
                                  float *qbias, float *score, int64_t n_rel, void *stream) {
    int rc = layer_args_ok(fwd, relation, input, boundary_node, boundary_value, n_query, weight, bias, ln_weight, ln_bias, n_rel);
    if (rc) return rc;
    if (!ultra_layer_score_supported(fwd, n_query, n_rel)) return ULTRA_ERR_BAD_SHAPE;
    if (query == nullptr || w1 == nullptr || b1 == nullptr || w2 == nullptr || b2 == nullptr || qbias == nullptr || score == nullptr)
        return ULTRA_ERR_NULL_POINTER;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(score_qbias_kernel, dim3((unsigned)n_query), dim3(128), 0, s, query, w1, b1, qbias);
    HIP_TRY(hipGetLastError());
    LayerParams q{};
    q.relation = relation; q.gather = input; q.bnode = boundary_node; q.bvec = boundary_value; q.out = score;      // (`out` only backs an unused descriptor)
    q.lin_w = weight; q.lin_b = bias; q.gamma = ln_weight; q.beta = ln_bias; q.eps = ln_eps; q.relu = relu; q.shortcut = shortcut;
    q.s_w1 = w1; q.s_qbias = qbias; q.s_w2 = w2; q.s_b2 = b2; q.score = score;
    return layer_launch(fwd, q, n_query, n_rel, true, s);
}

}  // extern "C"
