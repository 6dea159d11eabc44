// relcsr_build.hip -- native builder of the reduction plans (second translation unit of libultra_rspmm.so).
//
// What torchdrug does inside EVERY generalized_rspmm call -- sparse.coalesce() (sort the COO triples, merge
// duplicates) and coo2csr (/root/reference/ultra/layer.py:127,328 hand it an un-coalesced adjacency) -- plus the chunk
// schedule and packed edge words of this library, done once per graph on the device with rocPRIM primitives
// (radix sort, scans) and a few small kernels.  ultra_torchdrug_amd/relcsr.py keeps an equivalent builder written
// with torch ops (used for CPU tensors and as the cross-check in tests): both must produce identical arrays.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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

constexpr int kThreads = 256;
inline unsigned grid_for(long long n) { return (unsigned)((n + kThreads - 1) / kThreads > 0 ? (n + kThreads - 1) / kThreads : 1); }

// bump allocator over the caller's scratch buffer (256-B aligned pieces)
struct Arena {
    char *base;
    size_t off, cap;
    bool ok;
    template <typename T>
    T *take(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (off + bytes > cap) { ok = false; return nullptr; }
        T *p = reinterpret_cast<T *>(base + off);
        off += bytes;
        return p;
    }
};
inline size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

int bit_length(unsigned long long v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

// ------------------------------------------------------------------------------------------- coalesce kernels
__global__ void make_keys_kernel(const int64_t *row, const int64_t *col, const int64_t *rel, unsigned long long *keys,
                                 int64_t *idx, long long n, long long n_cols, long long n_rel) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((unsigned long long)row[i] * (unsigned long long)n_cols + (unsigned long long)col[i]) *
                  (unsigned long long)n_rel + (unsigned long long)rel[i];
    idx[i] = i;
}

__global__ void head_flags_kernel(const unsigned long long *keys, int *flags, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ void coalesce_scatter_kernel(const unsigned long long *keys, const int64_t *idx, const int *flags, const int *gid,
                                        const float *weight, int32_t *out_row, int32_t *out_col, int32_t *out_rel,
                                        float *out_weight, int64_t *edge_of_input, int *all_unit, long long n,
                                        long long n_cols, long long n_rel) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = gid[i] - 1;
    edge_of_input[idx[i]] = g;
    if (!flags[i]) return;
    unsigned long long k = keys[i];
    out_rel[g] = (int32_t)(k % (unsigned long long)n_rel);
    k /= (unsigned long long)n_rel;
    out_col[g] = (int32_t)(k % (unsigned long long)n_cols);
    out_row[g] = (int32_t)(k / (unsigned long long)n_cols);
    // duplicates of one triple: sequential fp32 sum in input order (the sort is stable)
    float acc = weight ? weight[idx[i]] : 1.0f;
    for (long long j = i + 1; j < n && !flags[j]; ++j) acc = acc + (weight ? weight[idx[j]] : 1.0f);
    out_weight[g] = acc;
    if (acc != 1.0f) *all_unit = 0;
}

// ------------------------------------------------------------------------------------------- schedule kernels
__global__ void row_ptr_kernel(const int32_t *row, int *row_ptr, long long n_edges, long long n_rows) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    long long lo = 0, hi = n_edges;            // first edge whose row >= r
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (row[mid] < r) lo = mid + 1; else hi = mid;
    }
    row_ptr[r] = (int)lo;
}

__global__ void start_flags_kernel(const int *row_ptr, int *start, long long n_rows, int chunk_edges, int chunk_rows,
                                   int piece_len) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int s = 1;
    if (r > 0) {
        const bool long_r = (row_ptr[r + 1] - row_ptr[r]) > piece_len;
        const bool long_p = (row_ptr[r] - row_ptr[r - 1]) > piece_len;
        s = (row_ptr[r] / chunk_edges != row_ptr[r - 1] / chunk_edges) || (r / chunk_rows != (r - 1) / chunk_rows) ||
            long_r || long_p;
    }
    start[r] = s;
}

// one entry per group: first row, and what the group contributes to the three exclusive scans
__global__ void group_info_kernel(const int *row_ptr, const int *start, const int *gid_row, int *g_first, int *n_piece,
                                  int *is_long, int *is_normal, long long n_rows, int piece_len) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows || !start[r]) return;
    const int g = gid_row[r] - 1;
    const int deg = row_ptr[r + 1] - row_ptr[r];
    const int lng = deg > piece_len;
    g_first[g] = (int)r;
    is_long[g] = lng;
    is_normal[g] = !lng;
    n_piece[g] = lng ? (deg + piece_len - 1) / piece_len : 0;
}

__global__ void emit_chunks_kernel(const int *row_ptr, const int *g_first, const int *n_piece, const int *is_long,
                                   const int *first_slot, const int *long_index, const int *normal_index, int4 *unsorted,
                                   unsigned *cost_key, int *order, int32_t *long_rows, long long n_groups,
                                   long long n_rows, int n_pieces_total, int piece_len) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int r0 = g_first[g];
    const int r1 = (g + 1 < n_groups) ? g_first[g + 1] : (int)n_rows;
    if (is_long[g]) {
        const int e0 = row_ptr[r0], e1 = row_ptr[r0 + 1];
        const int slot0 = first_slot[g];
        for (int k = 0; k < n_piece[g]; ++k) {
            const int a = e0 + k * piece_len;
            const int b = (a + piece_len < e1) ? a + piece_len : e1;
            const int slot = slot0 + k;
            unsorted[slot] = make_int4(a, b, r0, -(slot + 1));
            cost_key[slot] = 0xFFFFFFFFu - (unsigned)((b - a) * 4 + 1);       // ascending key == descending cost
            order[slot] = slot;
        }
        const int li = long_index[g];
        long_rows[li * 3 + 0] = r0;
        long_rows[li * 3 + 1] = slot0;
        long_rows[li * 3 + 2] = n_piece[g];
    } else {
        const int at = n_pieces_total + normal_index[g];
        const int a = row_ptr[r0], b = row_ptr[r1];
        unsorted[at] = make_int4(a, b, r0, r1);
        cost_key[at] = 0xFFFFFFFFu - (unsigned)((b - a) * 4 + (r1 - r0));
        order[at] = at;
    }
}

__global__ void gather_chunks_kernel(const int4 *unsorted, const int *order, int4 *chunks, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) chunks[i] = unsorted[order[i]];
}

// packed word per edge: row - chunk.row_begin | relation << 8 | node << (8 + bits_rel); wide: no node field
__global__ void packed_kernel_build(const int32_t *row, const int32_t *node_a, const int32_t *rel, const int *gid_row,
                                    const int *g_first, int32_t *packed, long long n_edges, int bits_rel, int wide,
                                    long long slack) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges + slack) return;
    if (e >= n_edges) { packed[e] = 0; return; }
    const int r = row[e];
    const unsigned delta = (unsigned)(r - g_first[gid_row[r] - 1]);
    unsigned w = delta;
    if (bits_rel > 0) w |= ((unsigned)rel[e]) << 8;
    if (!wide) w |= ((unsigned)node_a[e]) << (8 + bits_rel);
    packed[e] = (int32_t)w;
}

}  // namespace

extern "C" {

size_t ultra_relcsr_coalesce_temp_bytes(int64_t n_edges) {
    if (n_edges <= 0) return 256;
    const size_t n = (size_t)n_edges;
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (int64_t *)nullptr, (int64_t *)nullptr, n, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, (int *)nullptr, (int *)nullptr, n, rocprim::plus<int>(), (hipStream_t)0);
    const size_t prim = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    return padded(prim) + 2 * padded(n * 8) + 2 * padded(n * 8) + 2 * padded(n * 4) + 512;
}

// Sort (row, col, rel) triples, merge duplicates by summing their weights.  Outputs (capacity n_edges each):
// out_row / out_col / out_rel / out_weight of the unique triples in sorted order, edge_of_input[i] = position of input
// edge i in that list.  *n_unique_host and *unit_weight_host are written after a stream synchronisation.
int ultra_relcsr_coalesce(const int64_t *row, const int64_t *col, const int64_t *rel, const float *weight, int64_t n_edges,
                          int64_t n_rows, int64_t n_cols, int64_t n_rel, int32_t *out_row, int32_t *out_col,
                          int32_t *out_rel, float *out_weight, int64_t *edge_of_input, int64_t *n_unique_host,
                          int *unit_weight_host, void *temp, size_t temp_bytes, void *stream) {
    if (n_unique_host == nullptr || unit_weight_host == nullptr) return ULTRA_ERR_NULL_POINTER;
    *n_unique_host = 0;
    *unit_weight_host = 1;
    if (n_edges < 0 || n_rows < 0 || n_cols < 0 || n_rel < 0 || n_edges > 0x7fffffffLL) return ULTRA_ERR_BAD_SHAPE;
    if (n_edges == 0) return ULTRA_OK;
    if (row == nullptr || col == nullptr || rel == nullptr || out_row == nullptr || out_col == nullptr ||
        out_rel == nullptr || out_weight == nullptr || edge_of_input == nullptr || temp == nullptr)
        return ULTRA_ERR_NULL_POINTER;
    if (temp_bytes < ultra_relcsr_coalesce_temp_bytes(n_edges)) return ULTRA_ERR_WORKSPACE;
    const long long nr = n_rel > 0 ? n_rel : 1;
    const long double span = (long double)n_rows * (long double)n_cols * (long double)nr;
    if (span >= 9.0e18L) return ULTRA_ERR_BAD_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)n_edges;

    Arena a{static_cast<char *>(temp), 0, temp_bytes, true};
    unsigned long long *keys_in = a.take<unsigned long long>(n), *keys_out = a.take<unsigned long long>(n);
    int64_t *idx_in = a.take<int64_t>(n), *idx_out = a.take<int64_t>(n);
    int *flags = a.take<int>(n), *gid = a.take<int>(n);
    int *small = a.take<int>(64);                      // [0] all_unit
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, keys_in, keys_out, idx_in, idx_out, n, 0, 64, s);
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, flags, gid, n, rocprim::plus<int>(), s);
    size_t prim_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    void *prim = a.take<char>(prim_bytes);
    if (!a.ok) return ULTRA_ERR_WORKSPACE;

    hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(n_edges)), dim3(kThreads), 0, s, row, col, rel, keys_in, idx_in,
                       (long long)n_edges, (long long)n_cols, nr);
    const int end_bit = bit_length((unsigned long long)(span > 1 ? span - 1 : 1));
    HIP_TRY(rocprim::radix_sort_pairs(prim, sort_bytes, keys_in, keys_out, idx_in, idx_out, n, 0,
                                      end_bit > 0 ? end_bit : 1, s));
    hipLaunchKernelGGL(head_flags_kernel, dim3(grid_for(n_edges)), dim3(kThreads), 0, s, keys_out, flags, (long long)n_edges);
    HIP_TRY(rocprim::inclusive_scan(prim, scan_bytes, flags, gid, n, rocprim::plus<int>(), s));
    const int one = 1;
    HIP_TRY(hipMemcpyAsync(small, &one, sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(coalesce_scatter_kernel, dim3(grid_for(n_edges)), dim3(kThreads), 0, s, keys_out, idx_out, flags, gid,
                       weight, out_row, out_col, out_rel, out_weight, edge_of_input, small, (long long)n_edges,
                       (long long)n_cols, nr);
    HIP_TRY(hipGetLastError());
    int n_unique = 0, unit = 1;
    HIP_TRY(hipMemcpyAsync(&n_unique, gid + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&unit, small, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *n_unique_host = n_unique;
    *unit_weight_host = unit;
    return ULTRA_OK;
}

size_t ultra_relcsr_plan_temp_bytes(int64_t n_edges, int64_t n_rows, int64_t piece_len) {
    const size_t e = (size_t)(n_edges > 0 ? n_edges : 1), r = (size_t)(n_rows > 0 ? n_rows : 1);
    const size_t c = r + e / (size_t)(piece_len > 0 ? piece_len : 1) + 2;    // upper bound on the number of chunks
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned *)nullptr, (unsigned *)nullptr, (int *)nullptr,
                                    (int *)nullptr, c, 0, 32, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, (int *)nullptr, (int *)nullptr, r + 1, rocprim::plus<int>(), (hipStream_t)0);
    const size_t prim = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    return padded(prim) + 10 * padded((r + 2) * 4) + padded(c * 16) + 4 * padded(c * 4) + 1024;
}

// Chunk schedule + packed words of one plan.  `row` must be non-decreasing.  Capacities: chunks [cap_chunks][4] with
// cap_chunks >= n_rows + n_edges / piece_len + 2, long_rows [n_edges / piece_len + 1][3], packed [n_edges + slack] or NULL.
// counts_host[4] = {n_chunks, n_long_rows, n_pieces, packed_src_shift (0: no packed words, 32: wide ids)}.
int ultra_relcsr_plan(const int32_t *row, const int32_t *node_a, const int32_t *rel, int64_t n_edges, int64_t n_rows,
                      int64_t n_node_a, int64_t n_rel, int is_relation_plan, int wide_ids, int balance,
                      int64_t chunk_edges, int64_t chunk_rows, int64_t piece_len, int32_t *chunks, int64_t cap_chunks,
                      int32_t *long_rows, int64_t cap_long, int32_t *packed, int64_t packed_slack,
                      int64_t *counts_host, void *temp, size_t temp_bytes, void *stream) {
    if (counts_host == nullptr) return ULTRA_ERR_NULL_POINTER;
    counts_host[0] = counts_host[1] = counts_host[2] = counts_host[3] = 0;
    if (n_edges < 0 || n_rows < 0 || n_edges > 0x7fffffffLL || n_rows > 0x7fffffffLL || chunk_edges <= 0 ||
        chunk_rows <= 0 || piece_len <= 0)
        return ULTRA_ERR_BAD_SHAPE;
    if (n_rows == 0) return ULTRA_OK;
    if (chunks == nullptr || long_rows == nullptr || temp == nullptr || (n_edges > 0 && (row == nullptr || node_a == nullptr || rel == nullptr)))
        return ULTRA_ERR_NULL_POINTER;
    if (temp_bytes < ultra_relcsr_plan_temp_bytes(n_edges, n_rows, piece_len)) return ULTRA_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t R = (size_t)n_rows;

    Arena a{static_cast<char *>(temp), 0, temp_bytes, true};
    int *row_ptr = a.take<int>(R + 1), *start = a.take<int>(R), *gid_row = a.take<int>(R);
    int *g_first = a.take<int>(R + 1), *n_piece = a.take<int>(R + 1), *is_long = a.take<int>(R + 1), *is_normal = a.take<int>(R + 1);
    int *first_slot = a.take<int>(R + 1), *long_index = a.take<int>(R + 1), *normal_index = a.take<int>(R + 1);
    const size_t C = R + (size_t)n_edges / (size_t)piece_len + 2;
    int4 *unsorted = a.take<int4>(C);
    unsigned *key_in = a.take<unsigned>(C), *key_out = a.take<unsigned>(C);
    int *ord_in = a.take<int>(C), *ord_out = a.take<int>(C);
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, key_in, key_out, ord_in, ord_out, C, 0, 32, s);
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, start, gid_row, R + 1, rocprim::plus<int>(), s);
    const size_t prim_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    void *prim = a.take<char>(prim_bytes);
    if (!a.ok) return ULTRA_ERR_WORKSPACE;

    hipLaunchKernelGGL(row_ptr_kernel, dim3(grid_for(n_rows + 1)), dim3(kThreads), 0, s, row, row_ptr, (long long)n_edges,
                       (long long)n_rows);
    hipLaunchKernelGGL(start_flags_kernel, dim3(grid_for(n_rows)), dim3(kThreads), 0, s, row_ptr, start, (long long)n_rows,
                       (int)chunk_edges, (int)chunk_rows, (int)piece_len);
    HIP_TRY(rocprim::inclusive_scan(prim, scan_bytes, start, gid_row, R, rocprim::plus<int>(), s));
    int n_groups = 0;
    HIP_TRY(hipMemcpyAsync(&n_groups, gid_row + (R - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const size_t G = (size_t)n_groups;
    hipLaunchKernelGGL(group_info_kernel, dim3(grid_for(n_rows)), dim3(kThreads), 0, s, row_ptr, start, gid_row, g_first,
                       n_piece, is_long, is_normal, (long long)n_rows, (int)piece_len);
    HIP_TRY(rocprim::exclusive_scan(prim, scan_bytes, n_piece, first_slot, 0, G, rocprim::plus<int>(), s));
    HIP_TRY(rocprim::exclusive_scan(prim, scan_bytes, is_long, long_index, 0, G, rocprim::plus<int>(), s));
    HIP_TRY(rocprim::exclusive_scan(prim, scan_bytes, is_normal, normal_index, 0, G, rocprim::plus<int>(), s));
    int last[6];
    HIP_TRY(hipMemcpyAsync(&last[0], first_slot + (G - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&last[1], n_piece + (G - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&last[2], long_index + (G - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&last[3], is_long + (G - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&last[4], normal_index + (G - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&last[5], is_normal + (G - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const long long n_pieces = (long long)last[0] + last[1];
    const long long n_long = (long long)last[2] + last[3];
    const long long n_normal = (long long)last[4] + last[5];
    const long long n_chunks = n_pieces + n_normal;
    if (n_chunks > cap_chunks || n_long > cap_long || (size_t)n_chunks > C) return ULTRA_ERR_WORKSPACE;

    hipLaunchKernelGGL(emit_chunks_kernel, dim3(grid_for(n_groups)), dim3(kThreads), 0, s, row_ptr, g_first, n_piece, is_long,
                       first_slot, long_index, normal_index, unsorted, key_in, ord_in, long_rows, (long long)n_groups,
                       (long long)n_rows, (int)n_pieces, (int)piece_len);
    if (balance && n_chunks > 1) {
        HIP_TRY(rocprim::radix_sort_pairs(prim, sort_bytes, key_in, key_out, ord_in, ord_out, (size_t)n_chunks, 0, 32, s));
        hipLaunchKernelGGL(gather_chunks_kernel, dim3(grid_for(n_chunks)), dim3(kThreads), 0, s, unsorted, ord_out,
                           reinterpret_cast<int4 *>(chunks), n_chunks);
    } else {
        HIP_TRY(hipMemcpyAsync(chunks, unsorted, (size_t)n_chunks * sizeof(int4), hipMemcpyDeviceToDevice, s));
    }

    long long shift = 0;
    if (packed != nullptr && n_edges > 0 && chunk_rows <= 256) {
        const int bits_rel = is_relation_plan ? 0 : (bit_length((unsigned long long)(n_rel > 1 ? n_rel - 1 : 1)));
        const int bits_a = bit_length((unsigned long long)(n_node_a > 1 ? n_node_a - 1 : 1));
        const bool fits = 8 + bits_rel + bits_a <= 32;
        const bool wide = (!fits || wide_ids) && !is_relation_plan && n_rel <= (1LL << 24);
        if ((fits && !wide) || wide) {
            hipLaunchKernelGGL(packed_kernel_build, dim3(grid_for(n_edges + packed_slack)), dim3(kThreads), 0, s, row, node_a,
                               rel, gid_row, g_first, packed, (long long)n_edges, bits_rel, wide ? 1 : 0,
                               (long long)packed_slack);
            shift = wide ? 32 : 8 + bits_rel;
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    counts_host[0] = n_chunks;
    counts_host[1] = n_long;
    counts_host[2] = n_pieces;
    counts_host[3] = shift;
    return ULTRA_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------- graph of relations
// construct_relation_graph (/root/reference/ultra/rel_model.py:91-143): relations r1, r2 are joined by an edge of type
// hh / tt / ht / th when some entity is the head (tail) of an r1 edge and the head (tail) of an r2 edge.  The reference
// multiplies (2R x N) by (N x 2R) sparse incidence matrices and keeps the INDICES of the product; an spgemm over 2R rows
// of ~E / 2R entries each parallelises badly (117 s for S-stress: 10 M entities, 1 000 relations).  Only the pattern is
// wanted, so: one wave per entity, the entity's DISTINCT head relations A and tail relations B, every pair marked in a
// byte matrix -- identical stores from many waves, no atomics, no order to define.
namespace {

__global__ void marks_zero_kernel(uint32_t *words, long long n_words) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) words[i] = 0u;
}

__device__ __forceinline__ void mark_once(uint8_t *m, long long idx) {
    // a stale 0 read from another XCD's copy only costs a redundant store of the same byte
    if (m[idx] == 0) m[idx] = 1;
}

// lanes form an 8 x 8 tile of (i, j) pairs: no division in the pair loop
__global__ __launch_bounds__(256) void relation_marks_kernel(const int32_t *head_ptr, const int32_t *head_rel,
                                                             const int32_t *tail_ptr, const int32_t *tail_rel,
                                                             long long n_node, long long n_rel, uint8_t *marks) {
    const int lane = threadIdx.x & 63;
    const int li = lane >> 3, lj = lane & 7;
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const long long plane = n_rel * n_rel;
    for (long long e = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); e < n_node; e += n_waves) {
        const int h0 = head_ptr[e], nh = head_ptr[e + 1] - h0;
        const int t0 = tail_ptr[e], nt = tail_ptr[e + 1] - t0;
        for (int i0 = 0; i0 < nh; i0 += 8) {                      // A = heads
            const int i = i0 + li;
            const long long a = i < nh ? head_rel[h0 + i] : -1;
            for (int j0 = 0; j0 < nh; j0 += 8) {                  // hh
                const int j = j0 + lj;
                if (a >= 0 && j < nh) mark_once(marks, a * n_rel + head_rel[h0 + j]);
            }
            for (int j0 = 0; j0 < nt; j0 += 8) {                  // ht, and th as its transpose
                const int j = j0 + lj;
                if (a >= 0 && j < nt) {
                    const long long b = tail_rel[t0 + j];
                    mark_once(marks + 2 * plane, a * n_rel + b);
                    mark_once(marks + 3 * plane, b * n_rel + a);
                }
            }
        }
        for (int i0 = 0; i0 < nt; i0 += 8) {                      // tt
            const int i = i0 + li;
            const long long a = i < nt ? tail_rel[t0 + i] : -1;
            for (int j0 = 0; j0 < nt; j0 += 8) {
                const int j = j0 + lj;
                if (a >= 0 && j < nt) mark_once(marks + plane, a * n_rel + tail_rel[t0 + j]);
            }
        }
    }
}

}  // namespace

extern "C" {

int ultra_relation_graph_marks(const int32_t *head_ptr, const int32_t *head_rel, const int32_t *tail_ptr, const int32_t *tail_rel,
                               int64_t n_node, int64_t n_rel, uint8_t *marks, void *stream) {
    if (n_node < 0 || n_rel <= 0 || n_rel > (1LL << 15) || n_node > 0x7fffffffLL) return ULTRA_ERR_BAD_SHAPE;
    if (marks == nullptr || (reinterpret_cast<uintptr_t>(marks) & 3u)) return ULTRA_ERR_NULL_POINTER;
    if (n_node > 0 && (head_ptr == nullptr || tail_ptr == nullptr)) return ULTRA_ERR_NULL_POINTER;     // (empty lists: the rel arrays may be NULL)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n_words = (4 * n_rel * n_rel + 3) / 4;
    hipLaunchKernelGGL(marks_zero_kernel, dim3(grid_for(n_words)), dim3(kThreads), 0, s, reinterpret_cast<uint32_t *>(marks), n_words);
    HIP_TRY(hipGetLastError());
    if (n_node > 0) {
        const long long want = (n_node + 3) / 4;
        const unsigned grid = (unsigned)(want < 8192 ? (want > 0 ? want : 1) : 8192);
        hipLaunchKernelGGL(relation_marks_kernel, dim3(grid), dim3(256), 0, s, head_ptr, head_rel, tail_ptr, tail_rel,
                           (long long)n_node, (long long)n_rel, marks);
        HIP_TRY(hipGetLastError());
    }
    return ULTRA_OK;
}

}  // extern "C"
