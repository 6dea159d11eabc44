// relgraph_dense.hip -- sum aggregation over a DENSE relation graph on the exact-f32 matrix cores (third translation unit of
// libultra_rspmm.so; rspmm_kernels.hip's run_plan reaches it through csrc/relgraph_dense.h).
//
// construct_relation_graph (/root/reference/ultra/rel_model.py:99-143) multiplies incidence matrices: the graph of relations
// has 2R nodes, 4 edge types, unit weights, and is dense by nature (the FB15k237-shaped one is COMPLETE: 474 x 474 x 4 =
// 898 704 edges).  Walking it as a sparse edge list (quad_kernel<.., X_LDS>: both operands in LDS) is bound by LDS bandwidth at
// 84 us per layer + a 7 us fix-up of the 8 pieces every row is split into -- a quarter of an evaluation batch.  Here the same
// sums are a product with the plan's 0/1 matrix (ultra_segments.dense):
//
//     out[v, c] = sum over sources u ascending, types t ascending of A[v][u][t] * y[u][t][c],   y = relation[t][c] (*|+) x[u][c]
//
// v_mfma_f32_16x16x4_f32 computes acc = fmaf(a0, b0, acc); ...; acc = fmaf(a3, b3, acc) -- sequential, every step rounded like
// fmaf (tools/ubench/mfma_order.hip, run on an MI355X) -- so ONE instruction per source node with K = the 4 edge types adds that
// node's messages in relation order: fmaf(1, y, acc) = acc + y, fmaf(0, y, acc) = acc (y finite).  The message y is rounded by a
// VALU multiply first (-ffp-contract=off), exactly as the edge-list kernels and the oracle do.  Every output element is therefore
// the strictly sequential (source, relation) sum: the REFERENCE order (oracle `piece = 0`) for every row, no pieces, no fix-up.
//
// A wave owns a 16-row x 16-column tile of `out` and walks all sources: a chain of n_src dependent MFMAs (40 cycles each), two
// waves per SIMD keep the pipe (32 cycles per instruction) busy.  The 0/1 matrix is one BYTE per entry (v_cvt_f32_ubyte<k> makes
// the operand); dense_rows_kernel says how the operands reach the lanes.
//   d_input:    the same kernel over the by_src plan's matrix (rows = sources, gathered = output_grad): ((g * 1) * rel) = rel * g.
//   d_relation: its own documented order (include/ultra_rspmm.h): S_t = A_t . X on the matrix cores (K = 4 consecutive sources, no
//               multiply at all), then grad * S summed over the tile's rows in a fixed order, tiles added by a second tiny kernel.
#include <hip/hip_runtime.h>

#include <cstdint>

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

constexpr int kXcd = 8;
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
using ultra_detail::DenseCall;
constexpr int KIND_FWD = 0, KIND_DX = 1, KIND_DREL = 2;       // rspmm_kernels.hip's enum Kind

typedef float dn4 __attribute__((ext_vector_type(4)));
constexpr int kDnU = 8;          // forward / d_input: sources per block = 2 matrix words (4 sources each) + 2 loads of gathered rows
constexpr int kDnQ = 4;          // d_relation: source quads per block = 16 sources = 1 matrix word per type + 4 loads
constexpr int kDnWaves = 4;      // waves per workgroup: four adjacent 16-column tiles (one 64-column query block) of one row tile

struct DenseParams {
    const uint32_t *adj;     // ultra_segments.dense: one BYTE (0 / 1) per matrix entry, four entries of a lane per word
    const float *relation;   // [4, F]
    const float *gather;     // forward: input; d_input: output_grad; d_relation: input      [n_cols, F]
    const float *grad;       // d_relation: output_grad [n_rows, F]
    const float *add_rows;   // forward / d_input: [n_rows, F] or NULL
    const int32_t *bnode;    // forward: sparse boundary (see KParams)
    const float *bvec;
    int bdim;
    float *out;              // forward / d_input: [n_rows, F];  d_relation: tile sums [n_vt][4][F]
    long long F;
    int n_rows, n_cols, n_vt, n_ct;      // n_ct = F / 16 column tiles
    int cols_pad;            // columns a slab covers (kind 0: multiple of 2 kDnU; kind 1: multiple of 8 kDnQ)
    int n_items, per_xcd;    // workgroups with work; per XCD label
};

__device__ __forceinline__ bool dense_item(const DenseParams &p, int &vt, int &ct) {
    // XCD label k (blockIdx % 8) takes a contiguous range of items, i.e. whole row tiles: its L2 then holds those tiles' slabs
    // of the matrix and the gathered rows (1.9 MB for 474 nodes at F = 1 024)
    const int j = blockIdx.x / kXcd;
    const int item = (blockIdx.x % kXcd) * p.per_xcd + j;
    if (j >= p.per_xcd || item >= p.n_items) return false;
    const int n_qb = (p.n_ct + kDnWaves - 1) / kDnWaves;
    vt = item / n_qb;
    ct = (item - vt * n_qb) * kDnWaves + uniform((int)(threadIdx.x >> 6));
    return ct < p.n_ct;
}

// timing experiments only (wrong results): 1 = matrix operand constant (no conversion, no matrix loads), 2 = + no hand-round,
// 3 = + no multiply, 4 = + no gathered rows (the bare MFMA chain)
#ifndef ULTRA_DN_VARIANT
#define ULTRA_DN_VARIANT 0
#endif

// byte b (a compile-time 0..3) of a matrix word as 0.0f / 1.0f: one v_cvt_f32_ubyte<b>
template <int B>
__device__ __forceinline__ float entry(uint32_t word) {
#if ULTRA_DN_VARIANT >= 1
    return 1.0f;
#else
    return (float)((word >> (8 * B)) & 0xffu);
#endif
}

// forward (HAS_REL, MUL as given) and d_input (mul = mul: HAS_REL; mul = add: the gathered gradient itself).
//
// The first build issued, per source and wave, one dword load of the wave's 16 x values (replicated over the four 16-lane
// groups) and one of its 64 fp32 matrix entries: two wave-loads of 256 B per MFMA, and ran at 2.2 x the MFMA time (37 us for
// 474 nodes x 16 queries, 118 us for 64 queries: bound by the number of vector-memory instructions the L1 serves, not by the
// matrix pipe; profiles/r05_dense_crossover.json).  Now a lane of group kq loads x[u0 + kq] -- ONE wave-load brings the x values of
// FOUR sources -- and ds_bpermute (the LDS crossbar, no LDS memory) hands group j's value to all four groups for MFMA j; the
// matrix is one byte per entry, a lane's entries for four consecutive sources in one word: 0.5 vector-memory instructions per
// MFMA instead of 2.
template <int MUL, bool HAS_REL>
__device__ __forceinline__ dn4 dense_rows_sum(const DenseParams &p, const int vt, const int ct, const int lane) {
    const int i = lane & 15, kq = lane >> 4;
    const long long F = p.F;
    const int col = ct * 16 + i;
    float relv = 0.0f;
    if constexpr (HAS_REL) relv = p.relation[(long long)kq * F + col];
    const uint32_t row_bytes = (uint32_t)F * 4u;
    // gathered rows: the whole offset travels in the VGPR, so rows past the end are dropped by the descriptor's range check
    // (they meet zero matrix entries: fmaf(0, 0, acc) = acc)
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gather), 0, (int)((uint32_t)p.n_cols * row_bytes), 0x00020000);
    const uint32_t slab_words = (uint32_t)(p.cols_pad / 4) * 64u;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(p.adj + (long long)vt * slab_words), 0, (int)((slab_words + 4u * 64u) * 4u), 0x00020000);
    const uint32_t voff_x0 = (uint32_t)kq * row_bytes + (uint32_t)col * 4u, voff_a = (uint32_t)lane * 4u;
    const int pick[4] = {4 * i, 4 * (i + 16), 4 * (i + 32), 4 * (i + 48)};      // ds_bpermute addresses: lane i of group j

    auto load = [&](int u0, float (&x)[2], uint32_t (&a)[2]) {        // block of 8 sources starting at u0
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            x[h] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, voff_x0 + (uint32_t)(u0 + 4 * h) * row_bytes, 0, 0));
            a[h] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, voff_a, (uint32_t)(u0 / 4 + h) * 256u, 0);
        }
    };
    auto spread = [&](const float (&x)[2], float (&xs)[kDnU]) {       // x value of source u0 + j in every group
#pragma unroll
        for (int j = 0; j < kDnU; ++j) {
#if ULTRA_DN_VARIANT >= 4
            xs[j] = 1.5f;
#elif ULTRA_DN_VARIANT >= 2
            xs[j] = x[j >> 2];
#else
            xs[j] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pick[j & 3], __builtin_bit_cast(int, x[j >> 2])));
#endif
        }
    };
    dn4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    auto message = [&](float xv) -> float {
#if ULTRA_DN_VARIANT >= 3
        return xv;
#endif
        if constexpr (!HAS_REL) return xv;
        return (MUL == ULTRA_MUL_MUL) ? relv * xv : relv + xv;
    };
    auto compute = [&](const float (&xs)[kDnU], const uint32_t (&a)[2]) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<0>(a[0]), message(xs[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<1>(a[0]), message(xs[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<2>(a[0]), message(xs[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<3>(a[0]), message(xs[3]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<0>(a[1]), message(xs[4]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<1>(a[1]), message(xs[5]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<2>(a[1]), message(xs[6]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<3>(a[1]), message(xs[7]), acc, 0, 0, 0);
    };
    // Three stages in flight: loads of block t + 2, the crossbar hand-round of block t + 1, the MFMAs of block t.  Every load is
    // issued UNCONDITIONALLY (cols_pad is a multiple of two blocks, the matrix carries slack behind its last slab, gathered rows
    // past the end are range-checked): behind a branch that MAY have issued loads the compiler must assume the fewest in flight,
    // and its s_waitcnt for the current block then also waits for most of the block just requested -- the first build did
    // exactly that and ran the memory latency in series with the MFMAs.
    float x0[2], x1[2], s0[kDnU], s1[kDnU];
    uint32_t a0[2], a1[2], c0[2], c1[2];
    load(0, x0, a0);
    load(kDnU, x1, a1);
    spread(x0, s0);
    c0[0] = a0[0]; c0[1] = a0[1];
    for (int u0 = 0; u0 < p.cols_pad; u0 += 2 * kDnU) {
        load(u0 + 2 * kDnU, x0, a0);          // block t + 2
        spread(x1, s1);                       // block t + 1
        c1[0] = a1[0]; c1[1] = a1[1];
        compute(s0, c0);                      // block t
        load(u0 + 3 * kDnU, x1, a1);          // block t + 3
        spread(x0, s0);                       // block t + 2
        c0[0] = a0[0]; c0[1] = a0[1];
        compute(s1, c1);                      // block t + 1
    }
    return acc;
}

template <int MUL, bool HAS_REL>
__global__ __launch_bounds__(kDnWaves * 64) void dense_rows_kernel(const DenseParams p) {
    int vt, ct;
    if (!dense_item(p, vt, ct)) return;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const long long F = p.F;
    const int col = ct * 16 + i;
    const dn4 acc = dense_rows_sum<MUL, HAS_REL>(p, vt, ct, lane);

    // acc[r] = D[row 4 kq + r][column i]; the epilogue the reference applies right after the call (layer.py:156,358)
    int b_node = -1;
    float b_val = 0.0f;
    if (p.add_rows == nullptr && p.bnode != nullptr) {
        b_node = p.bnode[col / p.bdim];
        b_val = p.bvec[col];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = vt * 16 + 4 * kq + r;
        if (row < p.n_rows) {
            float v = acc[r];
            if (p.add_rows != nullptr) v = v + p.add_rows[(long long)row * F + col];
            else if (p.bnode != nullptr) v = v + (row == b_node ? b_val : 0.0f);
            p.out[(long long)row * F + col] = v;
        }
    }
}

// The WHOLE layer of a relation-graph Bellman-Ford in inference, one launch:
//     out = [input +] relu(LayerNorm(Linear_{128->64}(cat[input, rspmm(input) + boundary])))
// = GeneralizedRelationalConvNBF.forward (/root/reference/ultra/layer.py:111-190) + the caller's shortcut
// (ultra/rel_model.py:371-372).  A workgroup's four waves hold the sums of 16 nodes x the 64 columns of ONE query -- 16 complete
// rows of the epilogue -- so the 128 -> 64 product follows on the same matrix cores without the rows leaving the chip: wave w
// computes outputs 16 w .. 16 w + 15 with K = 4 per instruction = (in[s], up[s], in[s+1], up[s+1]), i.e. the fmaf chain
// bias, in[0], up[0], in[1], up[1], ... of combine_kernel / oracle_combine_forward; LayerNorm as two sequential 32-column half
// sums added once, relu, shortcut: the same expressions.  Bit-identical to dense_rows_kernel + combine_kernel
// (tests/test_relgraph_dense_gpu.py), one launch and one (N, Q, 64) round trip through memory less per layer.
struct DenseLayerParams {
    DenseParams d;           // the sums (d.out unused); d.gather is also the layer's `input` [n_rows, Q, 64]
    const float *weight;     // [64, 128]
    const float *bias, *gamma, *beta;
    float eps;
    int relu, shortcut;
    float *out;              // [n_rows, Q, 64], must not alias the input (other workgroups still gather from it)
};
constexpr int kDlStride = 68;        // floats per staged row: 16-byte aligned, rows 4 banks apart

__global__ __launch_bounds__(kDnWaves * 64) void dense_layer_kernel(const DenseLayerParams q) {
    const DenseParams &p = q.d;
    __shared__ __attribute__((aligned(16))) float t_in[16 * kDlStride], t_up[16 * kDlStride], t_z[16 * kDlStride];
    int vt, ct;
    if (!dense_item(p, vt, ct)) return;          // F % 64 == 0: a workgroup's four waves all have a tile or none has
    const int lane = threadIdx.x & 63, wave = ct & 3;
    const int i = lane & 15, kq = lane >> 4;
    const long long F = p.F;
    const int col = ct * 16 + i;
    // B fragments of the epilogue, requested before the walk: lane (i, kq) feeds W[16 w + i][64 (kq & 1) + (kq >> 1) + 2 j]
    float wf[32];
    {
        const float *wrow = q.weight + (long long)(16 * wave + i) * 128 + 64 * (kq & 1) + (kq >> 1);
#pragma unroll
        for (int j = 0; j < 32; ++j) wf[j] = wrow[2 * j];
    }
    const float bias = q.bias[16 * wave + i];
    // the layer's input rows of this tile (rows 4 w .. 4 w + 3 by this wave: four 256-byte rows per load)
    {
        const int row = vt * 16 + 4 * wave + kq;
        dn4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (row < p.n_rows) v = *reinterpret_cast<const dn4 *>(p.gather + (long long)row * F + (ct & ~3) * 16 + 4 * i);
        *reinterpret_cast<dn4 *>(t_in + (4 * wave + kq) * kDlStride + 4 * i) = v;
    }
    const dn4 acc = dense_rows_sum<ULTRA_MUL_MUL, true>(p, vt, ct, lane);
    {
        const int b_node = p.bnode[col / 64];
        const float b_val = p.bvec[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = vt * 16 + 4 * kq + r;
            t_up[(4 * kq + r) * kDlStride + 16 * wave + i] = acc[r] + (row == b_node ? b_val : 0.0f);
        }
    }
    __syncthreads();
    dn4 z = {bias, bias, bias, bias};
    {
        const float *src = ((kq & 1) ? t_up : t_in) + i * kDlStride + (kq >> 1);
#pragma unroll
        for (int j = 0; j < 32; ++j) z = __builtin_amdgcn_mfma_f32_16x16x4f32(src[2 * j], wf[j], z, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) t_z[(4 * kq + r) * kDlStride + 16 * wave + i] = z[r];
    __syncthreads();
    // LayerNorm, relu, shortcut: wave w finishes rows 4 w .. 4 w + 3, 16 lanes x 4 columns per row; the two half sums of a row are
    // taken SEQUENTIALLY (columns 0..31, 32..63) by the row's lanes 0 and 8 and handed to the others
    const int lrow = 4 * wave + kq;
    const float *zr = t_z + lrow * kDlStride;
    dn4 v = *reinterpret_cast<const dn4 *>(zr + 4 * i);
    if (q.gamma != nullptr) {
        const int half = (i >> 3) & 1;
        float s = 0.0f;
        if ((i & 7) == 0)
            for (int c = 0; c < 32; ++c) s = s + zr[32 * half + c];
        const float s0 = __shfl(s, 16 * kq, 64), s1 = __shfl(s, 16 * kq + 8, 64);
        const float mean = (s0 + s1) * (1.0f / 64.0f);
        float ss = 0.0f;
        if ((i & 7) == 0)
            for (int c = 0; c < 32; ++c) { const float dlt = zr[32 * half + c] - mean; ss = ss + dlt * dlt; }
        const float q0 = __shfl(ss, 16 * kq, 64), q1 = __shfl(ss, 16 * kq + 8, 64);
        const float var = (q0 + q1) * (1.0f / 64.0f);
        const float inv = 1.0f / sqrtf(var + q.eps);
        const dn4 g = *reinterpret_cast<const dn4 *>(q.gamma + 4 * i), b = *reinterpret_cast<const dn4 *>(q.beta + 4 * i);
        v.x = ((v.x - mean) * inv) * g.x + b.x; v.y = ((v.y - mean) * inv) * g.y + b.y;
        v.z = ((v.z - mean) * inv) * g.z + b.z; v.w = ((v.w - mean) * inv) * g.w + b.w;
    }
    if (q.relu) {
        v.x = !(v.x <= 0.0f) ? v.x : 0.0f; v.y = !(v.y <= 0.0f) ? v.y : 0.0f; v.z = !(v.z <= 0.0f) ? v.z : 0.0f; v.w = !(v.w <= 0.0f) ? v.w : 0.0f;
    }
    if (q.shortcut) {
        const dn4 x = *reinterpret_cast<const dn4 *>(t_in + lrow * kDlStride + 4 * i);
        v.x = v.x + x.x; v.y = v.y + x.y; v.z = v.z + x.z; v.w = v.w + x.w;
    }
    const int row = vt * 16 + lrow;
    if (row < p.n_rows) *reinterpret_cast<dn4 *>(q.out + (long long)row * F + (ct & ~3) * 16 + 4 * i) = v;
}

// d_relation, first pass: tile sums T[tile][type][column] (order: include/ultra_rspmm.h).  K = 4 consecutive sources: the B
// operand IS the gathered row (no multiply, no hand-round); per block of 16 sources 4 row loads + 4 matrix words (one per type,
// a lane's entries of four consecutive source quads in one word) feed 16 MFMAs on four independent accumulators.
__global__ __launch_bounds__(kDnWaves * 64) void dense_drel_kernel(const DenseParams p) {
    int vt, ct;
    if (!dense_item(p, vt, ct)) return;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const long long F = p.F;
    const int col = ct * 16 + i;
    const uint32_t row_bytes = (uint32_t)F * 4u;
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gather), 0, (int)((uint32_t)p.n_cols * row_bytes), 0x00020000);
    const uint32_t n_h = (uint32_t)p.cols_pad / 16u;          // matrix words per lane, type and row tile
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(p.adj), 0, (int)((4u * (uint32_t)p.n_vt * n_h + 2u) * 256u), 0x00020000);
    const uint32_t voff_a = (uint32_t)lane * 4u;
    uint32_t base_t[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) base_t[t] = ((uint32_t)(t * p.n_vt + vt) * n_h) * 256u;
    const uint32_t voff_x0 = (uint32_t)kq * row_bytes + (uint32_t)col * 4u;

    auto load = [&](int h, float (&x)[kDnQ], uint32_t (&a)[4]) {       // block h: sources 16 h .. 16 h + 15
#pragma unroll
        for (int j = 0; j < kDnQ; ++j) {
            x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, voff_x0 + (uint32_t)(16 * h + 4 * j) * row_bytes, 0, 0));
#if ULTRA_DN_VARIANT >= 4
            x[j] = 1.5f;
#endif
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, voff_a, base_t[t] + (uint32_t)h * 256u, 0);
    };
    dn4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = dn4{0.0f, 0.0f, 0.0f, 0.0f};
    auto compute = [&](const float (&x)[kDnQ], const uint32_t (&a)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<0>(a[t]), x[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<1>(a[t]), x[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<2>(a[t]), x[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(entry<3>(a[t]), x[3], acc[t], 0, 0, 0);
        }
    };
    float xa[kDnQ], xb[kDnQ];
    uint32_t aa[4], ab[4];
    load(0, xa, aa);
    const int n_blocks = p.cols_pad / 16;
    for (int h = 0; h < n_blocks; h += 2) {       // unconditional loads, as in dense_rows_kernel
        load(h + 1, xb, ab);
        compute(xa, aa);
        load(h + 2, xa, aa);
        compute(xb, ab);
    }

    // acc[t][r] = S_t[row 4 kq + r][column i];  P = grad * S;  q_k = ((P0 + P1) + P2) + P3 in the lane, the tile's sum
    // ((q0 + q1) + q2) + q3 over the four lanes of the column
    float gv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = vt * 16 + 4 * kq + r;
        gv[r] = row < p.n_rows ? p.grad[(long long)row * F + col] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float q = ((gv[0] * acc[t][0] + gv[1] * acc[t][1]) + gv[2] * acc[t][2]) + gv[3] * acc[t][3];
        const int qi = __builtin_bit_cast(int, q);
        const float q0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * i, qi));
        const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (i + 16), qi));
        const float q2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (i + 32), qi));
        const float q3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (i + 48), qi));
        const float tile = ((q0 + q1) + q2) + q3;
        if (kq == 0) p.out[((long long)vt * 4 + t) * F + col] = tile;
    }
}

// d_relation, second pass: d_relation[type][column] = sequential sum of the tile sums, tiles ascending
__global__ __launch_bounds__(256) void dense_drel_reduce_kernel(const float *tiles, float *d_relation, long long F, int n_vt) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * F) return;
    const long long t = idx / F, c = idx - t * F;
    float acc = 0.0f;
    // 16 tile sums in flight, added in tile order (one dependent load per tile made this 10 us for 30 tiles)
    for (int v0 = 0; v0 < n_vt; v0 += 16) {
        float part[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) part[j] = tiles[((long long)min(v0 + j, n_vt - 1) * 4 + t) * F + c];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (v0 + j < n_vt) acc = acc + part[j];
    }
    d_relation[idx] = acc;
}

// one thread per edge sets its byte (distinct edges -> distinct bytes; a byte store needs no read-modify-write of the word)
__global__ __launch_bounds__(256) void dense_build_kernel(const int32_t *row, const int32_t *node_a, const int32_t *node_b,
                                                          long long n_edges, int kind, int n_vt, int cols_pad, uint8_t *dense,
                                                          const int32_t *rel) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    long long idx;
    if (kind == 0) {        // [tile][col / 4][type][i][col % 4]
        const int v = row[e], u = node_a[e], t = rel[e];
        if (t < 0 || t >= 4) return;
        idx = ((((long long)(v >> 4) * (cols_pad / 4) + (u >> 2)) * 4 + t) * 16 + (v & 15)) * 4 + (u & 3);
    } else {                // [type][tile][col / 16][col % 4][i][(col / 4) % 4]
        const int t = row[e], u = node_a[e], v = node_b[e];
        if (t < 0 || t >= 4) return;
        idx = (((((long long)t * n_vt + (v >> 4)) * (cols_pad / 16) + (u >> 4)) * 4 + (u & 3)) * 16 + (v & 15)) * 4 + ((u >> 2) & 3);
    }
    dense[idx] = 1;
}

// zero fill as a kernel (a memset NODE inside a captured graph has misbehaved here, see ultra_rspmm_frontier_f32; the build is
// not captured today, but the rule costs nothing)
__global__ __launch_bounds__(256) void dense_zero_kernel(uint32_t *dense, long long n_words) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_words) dense[k] = 0u;
}

// columns a slab covers: whole rounds of the kernels' two register sets (kind 0: 2 x 8 sources; kind 1: 2 x 16)
inline int dense_cols_pad(long long n_cols, int kind) {
    const int round = kind == 0 ? 2 * kDnU : 8 * kDnQ;
    return (int)((n_cols + round - 1) / round * round);
}
// bytes the kernels may read past the last slab (the loads of blocks that are never used)
inline long long dense_slack_bytes(int kind) { return kind == 0 ? 2 * kDnU * 64 : 2 * 256; }

}  // namespace

namespace ultra_detail {

// Is the plan's dense form usable for this call?  (what the header lists as preconditions; the caller has looked at the knobs)
bool dense_applies(const DenseCall &c) {
    const ultra_segments *seg = c.seg;
    if (seg->dense == nullptr || seg->weight != nullptr || c.sum_op != ULTRA_SUM_ADD || c.n_rel != 4) return false;
    if (c.F % 16 != 0 || c.F * 4 >= (1LL << 24)) return false;
    if (seg->dense_cols != c.gather_rows || c.gather_rows * c.F * 4 >= (1LL << 31)) return false;
    if (c.kind == KIND_DREL) {
        if (c.mul_op != ULTRA_MUL_MUL || seg->n_rows != 4 || seg->dense_rows != c.gather2_rows) return false;
        const long long n_vt = (seg->dense_rows + 15) / 16;
        if (c.workspace == nullptr || c.workspace_bytes < (size_t)(n_vt * 4 * c.F) * sizeof(float)) return false;
    } else {
        if (seg->dense_rows != seg->n_rows) return false;
        if (c.kind == KIND_FWD && c.add_rows == nullptr && c.bnode != nullptr && c.bdim <= 0) return false;
    }
    return true;
}

int dense_launch(const DenseCall &c, hipStream_t stream) {
    const ultra_segments *seg = c.seg;
    const long long F = c.F;
    DenseParams q{};
    q.adj = seg->dense;
    q.relation = c.relation;
    q.gather = (c.kind == KIND_DX) ? c.grad : c.input;
    q.grad = c.grad;
    q.add_rows = (c.kind == KIND_DREL) ? nullptr : c.add_rows;
    q.bnode = (c.kind == KIND_FWD) ? c.bnode : nullptr;
    q.bvec = c.bvec;
    q.bdim = c.bdim;
    q.out = (c.kind == KIND_DREL) ? static_cast<float *>(c.workspace) : c.out;
    q.F = F;
    q.n_rows = (int)seg->dense_rows;
    q.n_cols = (int)seg->dense_cols;
    q.n_vt = (int)((seg->dense_rows + 15) / 16);
    q.n_ct = (int)(F / 16);
    q.cols_pad = dense_cols_pad(seg->dense_cols, c.kind == KIND_DREL ? 1 : 0);
    const int n_qb = (q.n_ct + kDnWaves - 1) / kDnWaves;
    q.n_items = q.n_vt * n_qb;
    q.per_xcd = (q.n_items + kXcd - 1) / kXcd;
    const dim3 grid((unsigned)(q.per_xcd * kXcd)), block(kDnWaves * 64);
    if (c.kind == KIND_DREL) {
        hipLaunchKernelGGL(dense_drel_kernel, grid, block, 0, stream, q);
        HIP_TRY(hipGetLastError());
        const long long n = 4 * F;
        hipLaunchKernelGGL(dense_drel_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           static_cast<const float *>(c.workspace), c.out, F, q.n_vt);
    } else if (c.kind == KIND_FWD) {
        if (c.mul_op == ULTRA_MUL_MUL) hipLaunchKernelGGL((dense_rows_kernel<ULTRA_MUL_MUL, true>), grid, block, 0, stream, q);
        else hipLaunchKernelGGL((dense_rows_kernel<ULTRA_MUL_ADD, true>), grid, block, 0, stream, q);
    } else {
        if (c.mul_op == ULTRA_MUL_MUL) hipLaunchKernelGGL((dense_rows_kernel<ULTRA_MUL_MUL, true>), grid, block, 0, stream, q);
        else hipLaunchKernelGGL((dense_rows_kernel<ULTRA_MUL_ADD, false>), grid, block, 0, stream, q);
    }
    HIP_TRY(hipGetLastError());
    return ULTRA_OK;
}

}  // namespace ultra_detail

namespace {

// ---------------------------------------------------------------------------------------------------- calibration
// ultra_calibrate_gather_f32: the bare gather of random 256-byte rows (see include/ultra_rspmm.h)
constexpr int kCalRowsPerWave = 2048;
typedef uint32_t cu4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void calibrate_gather_kernel(const float *table, uint32_t table_bytes, const int32_t *index,
                                                               long long n_waves, float *out) {
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, j = lane & 15;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= n_waves) return;
    const int32_t *mine = index + wave * kCalRowsPerWave;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, (int)table_bytes, 0x00020000);
    float acc = 0.0f;
    int ids = mine[lane & 31];
    for (int k = 0; k < kCalRowsPerWave; k += 32) {
        const int nxt = (k + 32 < kCalRowsPerWave) ? mine[k + 32 + (lane & 31)] : 0;
        cu4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t row = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (u * 4 + q), ids);
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row * 256u + (uint32_t)j * 16u, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc += (__builtin_bit_cast(float, v[u].x) + __builtin_bit_cast(float, v[u].y)) +
                   (__builtin_bit_cast(float, v[u].z) + __builtin_bit_cast(float, v[u].w));
        ids = nxt;
    }
    out[wave * 64 + lane] = acc;
}

}  // namespace

extern "C" {

size_t ultra_relcsr_dense_bytes(int64_t n_rows, int64_t n_cols, int kind) {
    if (n_rows <= 0 || n_cols <= 0 || (kind != 0 && kind != 1) || n_rows > (1 << 20) || n_cols > (1 << 20)) return 0;
    const long long n_vt = (n_rows + 15) / 16;
    const long long bytes = n_vt * 64 * dense_cols_pad(n_cols, kind) + dense_slack_bytes(kind);       // 16 rows x 4 types x columns
    if (bytes >= (1LL << 31)) return 0;        // the kernels address the matrix with 32-bit byte offsets
    return (size_t)bytes;
}

int ultra_relcsr_dense(const ultra_segments *plan, int64_t n_rows, int64_t n_cols, int kind, uint32_t *dense, void *stream) {
    if (plan == nullptr) return ULTRA_ERR_NULL_POINTER;
    if (plan->struct_bytes != (uint32_t)sizeof(ultra_segments) || plan->abi_version != (uint32_t)ULTRA_RSPMM_ABI_VERSION) return ULTRA_ERR_ABI;
    if (plan->n_edges < 0 || plan->n_edges > 0x7fffffffLL) return ULTRA_ERR_BAD_SHAPE;
    if (plan->n_edges > 0 && (plan->row == nullptr || plan->node_a == nullptr || plan->rel == nullptr)) return ULTRA_ERR_NULL_POINTER;
    const size_t bytes = ultra_relcsr_dense_bytes(n_rows, n_cols, kind);
    if (bytes == 0) return ULTRA_ERR_BAD_SHAPE;
    if (dense == nullptr) return ULTRA_ERR_NULL_POINTER;
    if (plan->weight != nullptr) return ULTRA_ERR_BAD_SHAPE;                  // unit weights only
    if (kind == 0 && plan->n_rows != n_rows) return ULTRA_ERR_BAD_SHAPE;
    if (kind == 1 && (plan->n_rows != 4 || (plan->n_edges > 0 && plan->node_b == nullptr))) return ULTRA_ERR_BAD_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n_words = (long long)(bytes / 4);
    hipLaunchKernelGGL(dense_zero_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, dense, n_words);
    HIP_TRY(hipGetLastError());
    if (plan->n_edges > 0) {
        hipLaunchKernelGGL(dense_build_kernel, dim3((unsigned)((plan->n_edges + 255) / 256)), dim3(256), 0, s, plan->row,
                           plan->node_a, plan->node_b, (long long)plan->n_edges, kind, (int)((n_rows + 15) / 16),
                           dense_cols_pad(n_cols, kind), reinterpret_cast<uint8_t *>(dense), plan->rel);
        HIP_TRY(hipGetLastError());
    }
    return ULTRA_OK;
}

int ultra_dense_layer_supported(const ultra_segments *fwd, int64_t n_query) {
    if (fwd == nullptr || fwd->struct_bytes != (uint32_t)sizeof(ultra_segments) || fwd->abi_version != (uint32_t)ULTRA_RSPMM_ABI_VERSION) return 0;
    if (fwd->dense == nullptr || fwd->weight != nullptr || n_query <= 0) return 0;
    if (fwd->dense_rows != fwd->n_rows || fwd->dense_rows != fwd->dense_cols) return 0;        // a layer maps the nodes onto themselves
    const long long F = n_query * 64;
    return (F * 4 < (1LL << 24) && fwd->dense_cols * F * 4 < (1LL << 31)) ? 1 : 0;
}

int ultra_dense_layer_forward_f32(const ultra_segments *fwd, const float *relation, const float *input,
                                  const int32_t *boundary_node, const float *boundary_value, int64_t n_query, const float *weight,
                                  const float *bias, const float *ln_weight, const float *ln_bias, float ln_eps, int relu,
                                  int shortcut, float *out, void *stream) {
    if (fwd != nullptr && (fwd->struct_bytes != (uint32_t)sizeof(ultra_segments) || fwd->abi_version != (uint32_t)ULTRA_RSPMM_ABI_VERSION))
        return ULTRA_ERR_ABI;
    if (!ultra_dense_layer_supported(fwd, n_query)) return ULTRA_ERR_BAD_SHAPE;
    if (relation == nullptr || input == nullptr || boundary_node == nullptr || boundary_value == nullptr || weight == nullptr ||
        bias == nullptr || out == nullptr)
        return ULTRA_ERR_NULL_POINTER;
    if (ln_weight != nullptr && ln_bias == nullptr) return ULTRA_ERR_NULL_POINTER;
    if (out == input) return ULTRA_ERR_BAD_SHAPE;
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(input) | reinterpret_cast<uintptr_t>(ln_weight) |
         reinterpret_cast<uintptr_t>(ln_bias)) & 15u)
        return ULTRA_ERR_BAD_SHAPE;
    DenseLayerParams q{};
    DenseParams &d = q.d;
    const long long F = n_query * 64;
    d.adj = fwd->dense; d.relation = relation; d.gather = input; d.bnode = boundary_node; d.bvec = boundary_value; d.bdim = 64;
    d.F = F; d.n_rows = (int)fwd->dense_rows; d.n_cols = (int)fwd->dense_cols; d.n_vt = (int)((fwd->dense_rows + 15) / 16);
    d.n_ct = (int)(F / 16); d.cols_pad = dense_cols_pad(fwd->dense_cols, 0);
    d.n_items = d.n_vt * (d.n_ct / kDnWaves); d.per_xcd = (d.n_items + kXcd - 1) / kXcd;
    q.weight = weight; q.bias = bias; q.gamma = ln_weight; q.beta = ln_bias; q.eps = ln_eps; q.relu = relu; q.shortcut = shortcut;
    q.out = out;
    hipLaunchKernelGGL(dense_layer_kernel, dim3((unsigned)(d.per_xcd * kXcd)), dim3(kDnWaves * 64), 0, static_cast<hipStream_t>(stream), q);
    HIP_TRY(hipGetLastError());
    return ULTRA_OK;
}

int ultra_calibrate_gather_f32(const float *table, int64_t n_rows, const int32_t *index, int64_t n_index, float *out,
                               int64_t *n_waves_host, void *stream) {
    if (n_rows <= 0 || n_index < 0 || n_rows * 256 >= (1LL << 32)) return ULTRA_ERR_BAD_SHAPE;
    const long long n_waves = n_index / kCalRowsPerWave;
    if (n_waves_host != nullptr) *n_waves_host = n_waves;
    if (out == nullptr || n_waves == 0) return ULTRA_OK;
    if (table == nullptr || index == nullptr) return ULTRA_ERR_NULL_POINTER;
    hipLaunchKernelGGL(calibrate_gather_kernel, dim3((unsigned)((n_waves + 7) / 8)), dim3(512), 0, static_cast<hipStream_t>(stream),
                       table, (uint32_t)(n_rows * 256), index, n_waves, out);
    HIP_TRY(hipGetLastError());
    return ULTRA_OK;
}

}  // extern "C"
