/*
 * include/ultra_rspmm.h -- C ABI of libultra_rspmm.so (MI355X / gfx950).
 *
 * Drop-in boundary for the one native operator on the reference's hot path:
 *
 *     torchdrug.layers.functional.generalized_rspmm(sparse, relation, input, sum=..., mul=...)
 *
 * called from /root/reference/ultra/layer.py:134-167 (relation-graph stack) and :336-369 (entity
 * stack).  In torchdrug that Python function dispatches to the C++/CUDA extension entry points
 * rspmm_{add,min,max}_{mul,add}_{forward,backward}_{cpu,cuda}(sparse, relation, input[, output,
 * output_grad]) (torchdrug/layers/functional/extension/rspmm.{h,cpp,cu}, un-vendored).
 * The functions below are what a binding for that path would bind instead: plain device pointers and
 * sizes, no torch types.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host;
 *  - all tensors are dense row-major fp32, indices are int32;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only enqueues
 *    work on that stream: no allocation, no host synchronisation, hipGraph-capture safe;
 *  - return value: 0 on success, otherwise an ultra_status code (ultra_rspmm_status_string()).
 *
 * Reduction plan ("segments").  One call reduces, for every target row t, an ordered list of
 * contributions.  The three plans of a graph differ only in what a row and a contribution are:
 *    forward      rows = destination nodes   sorted by (dst, src, rel)   (the coalesced CSR torchdrug builds)
 *    d_input      rows = source nodes        sorted by (src, dst, rel)
 *    d_relation   rows = relations           sorted by (rel, dst, src)
 * A plan cuts the sorted edge list into chunks: a chunk is either a run of whole rows or one piece
 * (<= piece_len consecutive edges) of a row longer than piece_len.  Pieces are summed into a
 * workspace and added in piece order by a second kernel, so results never depend on scheduling.
 */
#ifndef ULTRA_RSPMM_H
#define ULTRA_RSPMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared here are exported
 * (tests/test_host_logic.py compares `nm -D` of the built library with this header). */
#pragma GCC visibility push(default)

#define ULTRA_RSPMM_ABI_VERSION 8

/* sum= / mul= of generalized_rspmm (layer.py:134-167: sum in {"add","max","min"}, mul in {"mul","add"}) */
enum ultra_sum_op { ULTRA_SUM_ADD = 0, ULTRA_SUM_MIN = 1, ULTRA_SUM_MAX = 2 };
enum ultra_mul_op { ULTRA_MUL_MUL = 0, ULTRA_MUL_ADD = 1 };

enum ultra_status {
    ULTRA_OK = 0,
    ULTRA_ERR_BAD_OP = 1,        /* unknown sum/mul code (the Python side raises ValueError first) */
    ULTRA_ERR_BAD_SHAPE = 2,     /* F <= 0, negative sizes, index range beyond int32 */
    ULTRA_ERR_NULL_POINTER = 3,
    ULTRA_ERR_WORKSPACE = 4,     /* workspace smaller than ultra_rspmm_workspace_bytes() */
    ULTRA_ERR_HIP = 5,           /* a HIP runtime call failed; see ultra_rspmm_last_hip_error() */
    ULTRA_ERR_NO_DEVICE = 6,     /* no gfx950 device visible */
    ULTRA_ERR_ABI = 7            /* an ultra_segments whose struct_bytes / abi_version are not this library's (a binding   */
                                 /* written against another header): nothing of the struct beyond those two fields is read */
};

/* One ordered reduction plan over the edge list (all arrays in device memory, built once per graph). */
typedef struct ultra_segments {
    /* ABI 8: the fence.  A binding sets struct_bytes = sizeof(ultra_segments) of the header it was written against  */
    /* (== ultra_segments_bytes() of the library it loads) and abi_version = ULTRA_RSPMM_ABI_VERSION; every entry    */
    /* point that takes a plan checks both FIRST and returns ULTRA_ERR_ABI otherwise -- a struct declared from an   */
    /* older header (a field short) is refused instead of being read past its end.                                   */
    uint32_t struct_bytes;
    uint32_t abi_version;
    int64_t n_rows;            /* number of target rows of this plan                                   */
    int64_t n_edges;           /* coalesced edge count E                                                */
    const int32_t *row;        /* [E] target row of each edge, non-decreasing                           */
    const int32_t *node_a;     /* [E] forward: src node | d_input: dst node | d_relation: src node      */
    const int32_t *node_b;     /* [E] forward: unused   | d_input: unused   | d_relation: dst node      */
    const int32_t *rel;        /* [E] relation id                                                       */
    const float *weight;       /* [E] edge weight, or NULL when every weight is exactly 1.0f            */
    int64_t n_chunks;          /* number of schedule entries                                            */
    const int32_t *chunks;     /* [n_chunks][4]: {edge_begin, edge_end, row_begin, row_end}; a piece of */
                               /*   a long row stores row_end = -(piece_slot + 1)                       */
    int64_t n_long_rows;       /* rows split into pieces                                                */
    const int32_t *long_rows;  /* [n_long_rows][3]: {row, first_piece_slot, n_pieces}                   */
    int64_t n_pieces;          /* total piece slots (workspace rows)                                    */
    int64_t piece_len;         /* contributions per piece                                               */
    /* optional packed edge words for the fast path (NULL when the ids do not fit in 32 bits):           */
    /*   bits [0,8) row - chunk row_begin | bits [8, src_shift) relation id | bits [src_shift,32) node_a */
    /*   packed_src_shift == 32: the word holds row delta | relation << 8 only, node ids are read from   */
    /*   node_a (which then carries 16 readable words of slack after the last edge)                      */
    const uint32_t *packed;
    int64_t packed_src_shift;
    /* optional hot-row cache (n_hot > 0): the node field of `packed` then holds the cache slot (< n_hot) for   */
    /* the n_hot most frequently gathered nodes and n_hot + node id for all others                               */
    int64_t n_hot;
    const int32_t *hot_nodes;  /* [n_hot] */
    /* optional row pointers [n_rows + 1] (first edge of every target row); with them, plans of big graphs (node ids  */
    /* outside the packed word) that have no split rows run one row per 16-lane group (csrc/rowgroup.inc)           */
    const int32_t *row_ptr;
    /* optional DENSE form (ABI 7; ultra_relcsr_dense): the plan's edges as a 0/1 byte matrix, for unit-weight plans over  */
    /* exactly 4 relation types -- the relation graphs of /root/reference/ultra/rel_model.py:99-143, which multiply       */
    /* incidence matrices and come out dense (the FB15k237-shaped one is complete: 474 x 474 x 4).  With it the sum        */
    /* aggregations run on the exact-f32 matrix cores (csrc/relgraph_dense.hip); NULL = walk the edge list.                */
    const uint32_t *dense;
    int64_t dense_rows;        /* nodes along the matrix's 16-row tiles (forward: n_dst, d_input: n_src, d_relation: n_dst) */
    int64_t dense_cols;        /* gathered nodes (forward: n_src, d_input: n_dst, d_relation: n_src)                        */
    /* optional (ABI 7; ultra_edge_removal_marks): a copy of `packed` in which bit 31 is set for every edge whose weight is     */
    /* exactly 0, for plans whose other weights are all exactly 1 (the training step's edge removal on a unit-weight graph,    */
    /* /root/reference/ultra/model.py:57-74): the sum / mul kernels then run their unit-weight form on these words -- a marked   */
    /* edge issues its gathers past the buffer descriptors and contributes the (+-0) a zero weight contributes -- and never      */
    /* read `weight`; kernels without that form use `packed` + `weight` as before.  NULL = none.                              */
    const uint32_t *packed_dead;
} ultra_segments;

int ultra_rspmm_abi_version(void);
/* sizeof(ultra_segments) as THIS library was compiled (ABI 8): what a binding's own declaration of the struct must measure. */
size_t ultra_segments_bytes(void);
const char *ultra_rspmm_status_string(int status);
/* hipError_t of the last failing HIP call on this thread (0 if none). */
int ultra_rspmm_last_hip_error(void);

/* Fills n_cu (compute units), lds_bytes (per workgroup limit), arch (e.g. "gfx950", buffer >= 32 B). */
int ultra_rspmm_device_info(int device, int *n_cu, int *lds_bytes, char *arch_host, size_t arch_len);

/*
 * Measurement aid (bench.py): the NEXT forward/backward plan launched from this thread records `start_event`
 * right before and `stop_event` right after its main segment kernel, on the stream of that call (hipEvent_t
 * handles owned by the caller, passed as void*; NULL = off).  One-shot: cleared by the call that uses it.
 */
int ultra_rspmm_profile_next(void *start_event, void *stop_event);
/* hipEvent_t helpers for the hook above, so that the events come from the HIP runtime this library runs on
 * (a process may hold more than one copy of libamdhip64).  elapsed_ms synchronises on stop_event first. */
int ultra_rspmm_event_create(void **event_host);
int ultra_rspmm_event_destroy(void *event);
int ultra_rspmm_event_elapsed_ms(void *start_event, void *stop_event, float *ms_host);

/* Test/bench knob (process-wide): bit 0 forces the general kernel where the packed fast paths apply, bit 1 keeps
 * them from staging a small gathered matrix in LDS, bit 2 selects one chunk per wave (packed_kernel) where four
 * chunks per wave (quad_kernel) would run, bit 3 the chunked kernels where one row per 16-lane group (rowgroup_kernel)
 * would run, bit 4 the wide-group forms of that kernel (32 / 64 lanes per row, column tiles of 128 / 256) on inputs small
 * enough to be cache-resident, bit 5 makes quad_kernel walk a label's column tiles one after the other where it would
 * work on several at once (small graphs), bit 6 walks the edge list of a plan that carries a dense form, bit 7 runs the weighted kernels where a plan carries marked words (packed_dead).  Bit 0 also selects the L2-row form of the first-layer frontier kernel where the
 * LDS-message form would run.  All paths return identical bits. */
int ultra_rspmm_force_general_path(int on);

/* Process-wide: size every persistent grid for `n` compute units fewer than the device has (default 0).  The plan kernels run
 * one workgroup per compute unit for the life of a launch; a collective overlapped with them on a side stream (RCCL's kernels
 * during the phased training step) then finds no free compute unit until workgroups retire.  Affects launches (and hipGraph
 * captures) made AFTER the call; results never depend on it. */
int ultra_rspmm_reserve_cus(int n);

/* Scratch bytes a call over `seg` with row width F needs (piece partial sums). */
size_t ultra_rspmm_workspace_bytes(const ultra_segments *seg_host, int64_t F);

/*
 * out[v, :] = SUM_{(u, v, r, w) in row v}  w * (relation[r, :] MUL input[u, :])
 * replaces rspmm_{sum}_{mul}_forward_cuda(sparse, relation, input)      [layer.py:134-167,336-369]
 *   fwd       : forward plan;  relation [n_rel, F];  input [n_src, F];  out [fwd->n_rows, F]
 *   n_src     : rows of `input` (bounds the 32-bit offsets of the packed fast path)
 *   add_rows  : optional [n_rows, F] or NULL.  When given, the epilogue the reference applies right
 *               after the call is fused: sum=add -> out + add_rows (layer.py:156,358),
 *               sum=max -> max(out, add_rows) (layer.py:162,364), sum=min -> min(out, add_rows).
 * Empty rows give 0 (add), FLT_MAX (min), -FLT_MAX (max): torchdrug's NaryMin / NaryMax start from
 * std::numeric_limits<scalar_t>::max() / lowest(), finite values.
 */
int ultra_rspmm_forward_f32(const ultra_segments *fwd_host, const float *relation, const float *input,
                            const float *add_rows, float *out, void *workspace, size_t workspace_bytes,
                            int64_t n_src, int64_t n_rel, int64_t F, int sum_op, int mul_op, void *stream);

/*
 * The same operator straight from a coalesced CSR, without a plan (SURVEY.md 8b's torch-free entry): rows = destination
 * nodes, row_ptr int32 [N + 1], src / rel int32 [E] sorted by (row, src, rel), w fp32 [E] or NULL (all ones).
 * Every row is reduced strictly sequentially in that order -- the reference order for EVERY row (no pieces), so long
 * rows cost their length on one 16-lane group: meant for the standalone bench, for raw-CSR callers
 * (torch.ops.ultra_mi.rspmm_fwd) and for graphs of short rows; skewed KGs should use a plan.
 * relation [R, F], x [n_src, F], out [N, F] fp32, 16-byte aligned rows (F % 4 == 0).
 */
int ultra_rspmm_fwd_f32(const int32_t *row_ptr, const int32_t *src, const int32_t *rel, const float *w,
                        const float *relation, const float *x, float *out, int64_t N, int64_t E, int64_t R, int64_t F,
                        int sum_op, int mul_op, void *stream);

/* Forward with the Bellman-Ford boundary in its sparse form (same kernels, no dense [n_rows, F] read per layer).
 * The reference builds `boundary = zeros(N, B, D); boundary.scatter_add_(0, h_index, query)` (ultra/model.py:106-107,
 * ultra/rel_model.py:114-115) and applies `update + boundary` / `max(update, boundary)` after every rspmm
 * (ultra/layer.py:156,162,358,364).  Here: column c belongs to query block c / block; row boundary_node[c / block]
 * holds boundary_value[c] in that column and every other element of the boundary is 0.
 *   boundary_node  : int32 [F / block]     boundary_value : fp32 [F]     block : columns per query (64), F % block == 0
 * Result identical to ultra_rspmm_forward_f32 with the dense tensor as add_rows. */
int ultra_rspmm_forward_boundary_f32(const ultra_segments *fwd, const float *relation, const float *input,
                                     const int32_t *boundary_node, const float *boundary_value, int64_t block, float *out,
                                     void *workspace, size_t workspace_bytes, int64_t n_src, int64_t n_rel, int64_t F,
                                     int sum_op, int mul_op, void *stream);

/* The FIRST Bellman-Ford layer: `input` IS the boundary (ultra/model.py:116-120, ultra/rel_model.py:365-369), zero outside
 * row boundary_node[b] of query block b.  For sum = add, mul = mul (DistMult messages, summed: the shipped configuration)
 * every edge whose source row is zero contributes w * (rel * 0) = +-0 and can be skipped without changing one bit of the
 * sums, so only the out-edges of the boundary nodes are visited: identical result to
 * ultra_rspmm_forward_boundary_f32(fwd, relation, <dense boundary>, boundary_node, boundary_value, ...) for finite
 * relation values, from deg_out(boundary nodes) edges instead of E.
 *   by_src    : the d_input plan of the graph (edges sorted by (src, dst, rel)); its `weight` / `piece_len` are used
 *   src_ptr   : int32 [n_src + 1] first edge of every source node in that order
 *   fwd_rank  : int32 [E] position of each of those edges inside its destination row in FORWARD-plan order (which piece
 *               of a split row the edge belongs to: rank / piece_len)
 *   block     : must be 64;  out [n_dst, F] is written completely (zero fill included). */
int ultra_rspmm_frontier_f32(const ultra_segments *by_src, const int32_t *src_ptr, const int32_t *fwd_rank,
                             const float *relation, const int32_t *boundary_node, const float *boundary_value,
                             int64_t block, float *out, int64_t n_dst, int64_t n_rel, int64_t F, void *stream);

/* The WHOLE first layer of a Bellman-Ford in inference (ABI 6), sparse end to end:
 *     out = [boundary +] relu(LayerNorm(Linear_{128->64}(cat[boundary, rspmm(boundary) + boundary])))
 * = GeneralizedRelationalConv*.forward on `input = boundary` (/root/reference/ultra/model.py:116-127, ultra/layer.py:298-392)
 * -- ultra_rspmm_frontier_f32 followed by ultra_combine_forward_boundary_f32, bit for bit.  A row (v, q) that no out-edge of
 * boundary_node[q] reaches (and v != boundary_node[q]) has a zero input row and a zero update row, so its output is ONE vector
 * for the whole layer, relu(LayerNorm(bias)): a fill kernel derives it in the epilogue kernel's own order and broadcasts
 * it, the frontier kernel writes its rows over it AND lists them, and the epilogue runs on the listed rows only
 * (~a fifth of the rows on hub-heavy FB15k237-shaped batches, a few per cent on average).
 *   run_prefix  : int32 [E], by_src order: number of (source, destination) runs that start at or before each edge (inclusive
 *                 prefix count of `src or dst differs from the previous edge`); fixes every listed row's slot, so the list
 *                 is written without atomics or counters and the same way on every launch
 *   out         : [n_dst, n_query, 64], written completely
 *   row_list    : int32 scratch of row_list_len entries
 *   max_runs    : the largest number of (source, destination) runs any ONE source node has in the plan, i.e. of distinct
 *                 destinations (a property of the graph, like run_prefix); row_list_len < n_query * (max_runs + 1) is refused
 *                 with ULTRA_ERR_BAD_SHAPE -- a shorter list could not hold every query's slots, and rows without a slot would
 *                 keep their raw sums with no epilogue applied
 *   list_offset : int32 scratch [n_query + 1]
 * Sum aggregation of DistMult messages with a FINITE relation table (as ultra_rspmm_frontier_f32).
 * ultra_first_layer_sparse_supported: the shapes this entry takes (message table of n_rel rows in LDS, n_query <= 128,
 * n_dst * n_query * 256 B < 4 GiB); otherwise call the two entries it fuses. */
int ultra_first_layer_sparse_supported(int64_t n_dst, int64_t n_rel, int64_t n_query);
int ultra_first_layer_sparse_f32(const ultra_segments *by_src, const int32_t *src_ptr, const int32_t *fwd_rank,
                                 const int32_t *run_prefix, const float *relation, const int32_t *boundary_node,
                                 const float *boundary_value, int64_t n_query, const float *weight, const float *bias,
                                 const float *ln_weight, const float *ln_bias, float ln_eps, int relu, int shortcut,
                                 float *out, int32_t *row_list, int64_t row_list_len, int64_t max_runs, int32_t *list_offset,
                                 int64_t n_dst, int64_t n_rel, void *stream);

/* The first layer of an entity Bellman-Ford in TRAINING, sparse end to end (ABI 7; csrc/first_layer_train.inc).  Forward:
 * ultra_first_layer_sparse_f32 with the frontier's raw rows kept -- `update` [n_dst, n_query, 64] receives them at the LISTED rows
 * (what the epilogue's backward recomputes from; every other row of `update` stays unwritten), `out` receives the constant row
 * everywhere and the epilogue's result at the listed rows; row_list / list_offset as there (list_offset[n_query] = slots in use).
 * Same bits in `out` as ultra_rspmm_frontier_f32 + ultra_combine_forward_boundary_f32.
 * Backward of that layer's epilogue, ultra_first_layer_epilogue_backward_f32: the listed rows of input / update / grad_out are
 * gathered into dense buffers, run through the one-pass backward of ultra_combine_backward_fused_f32 (which reads the row count
 * from the device) and their d_input / d_update rows scattered back -- d_input [rows, 64] is zero elsewhere, d_update is written
 * at the listed rows only (the first layer's rspmm backward reads it there only) -- and every other row's share of d_bias /
 * d_ln_weight / d_ln_bias, a fixed linear map J(bias, LayerNorm, relu) of the COLUMN SUMS of grad_out over those rows, is added.
 *   list_count : device pointer to the number of slots in use (list_offset + n_query);  list_cap: slots row_list holds
 *   workspace  : ultra_first_layer_epilogue_backward_workspace(device, list_cap) bytes
 * Gradients equal the dense backward's up to the association of the sums.  ultra_column_sum_f32: deterministic column sums of a
 * (rows, 64) matrix (rows_dev: optional device-side row count; partial_workspace: ultra_column_sum_blocks() x 64 floats). */
int ultra_first_layer_sparse_train_f32(const ultra_segments *by_src, const int32_t *src_ptr, const int32_t *fwd_rank,
                                       const int32_t *run_prefix, const float *relation, const int32_t *boundary_node,
                                       const float *boundary_value, int64_t n_query, const float *weight, const float *bias,
                                       const float *ln_weight, const float *ln_bias, float ln_eps, int relu, int shortcut,
                                       float *update, float *out, int32_t *row_list, int64_t row_list_len, int64_t max_runs,
                                       int32_t *list_offset, int64_t n_dst, int64_t n_rel, void *stream);
size_t ultra_first_layer_epilogue_backward_workspace(int device, int64_t list_cap);
int ultra_first_layer_epilogue_backward_f32(const float *input, const float *update, const float *grad_out, const int32_t *row_list,
                                            const int32_t *list_count, int64_t list_cap, const float *weight, const float *bias,
                                            const float *ln_weight, const float *ln_bias, float ln_eps, int relu, int shortcut,
                                            float *d_input, float *d_update, float *d_weight, float *d_bias, float *d_ln_weight,
                                            float *d_ln_bias, float *workspace, size_t workspace_bytes, int64_t rows, void *stream);
int ultra_column_sum_blocks(void);
int ultra_column_sum_f32(const float *x, int64_t rows, const int32_t *rows_dev, float *partial_workspace, float *out, void *stream);

/* d_input of the FIRST layer's rspmm in training, at the rows that are used: that layer's input is the boundary
 * (/root/reference/ultra/model.py:106-107,116-120), whose gradient autograd consumes at row (boundary_node[q], query block q)
 * only.  For every query q:
 *     d_input[boundary_node[q], q*64 .. q*64+63] += sum over out-edges (u -> v, r) of u = boundary_node[q]:
 *                                                   (output_grad[v, same columns] * w) [* relation[r, same columns]]
 * -- the out-edges of one node per query instead of the d_input pass over all E edges for all queries.  by_src / src_ptr:
 * as for ultra_rspmm_frontier_f32; F = n_query * 64; mul_op = add has no relation factor.  Added IN PLACE (d_input holds the
 * epilogue's share already); the other rows of d_input do not receive their edge gradient. */
size_t ultra_rspmm_backward_boundary_rows_workspace(int64_t n_query);      /* bytes of scratch for the call below */
int ultra_rspmm_backward_boundary_rows_f32(const ultra_segments *by_src, const int32_t *src_ptr, const float *relation,
                                           const float *output_grad, const int32_t *boundary_node, float *d_input,
                                           float *workspace, size_t workspace_bytes, int64_t n_src, int64_t n_query, int64_t F,
                                           int mul_op, void *stream);

/*
 * Gradients of the call above w.r.t. input and relation
 * replaces rspmm_{sum}_{mul}_backward_cuda(sparse, relation, input, output, output_grad).
 *   by_src / by_rel : the d_input / d_relation plans of the same graph (n_src / n_dst: rows of input / output)
 *   output          : forward result WITHOUT add_rows fused (only read for min/max)
 *   output_grad     : dL/d(output) [n_dst, F]
 *   d_input [n_src, F], d_relation [n_rel, F]: either may be NULL to skip it.
 */
int ultra_rspmm_backward_f32(const ultra_segments *by_src_host, const ultra_segments *by_rel_host,
                             const float *relation, const float *input, const float *output,
                             const float *output_grad, float *d_input, float *d_relation, void *workspace,
                             size_t workspace_bytes, int64_t n_src, int64_t n_dst, int64_t n_rel, int64_t F,
                             int sum_op, int mul_op, void *stream);

/* The same with d_input = d_input_add + (gradient through the edges): the rows of `input` usually receive a second
 * gradient from the layer's dense epilogue (cat[input, update] -> Linear, plus the shortcut), which autograd would add in a
 * pass of its own; here it rides in the kernel's row epilogue.  sum = add only; d_input_add [n_src, F] may be the same
 * buffer as d_input (every element is read before it is written); NULL: exactly ultra_rspmm_backward_f32. */
int ultra_rspmm_backward_accumulate_f32(const ultra_segments *by_src_host, const ultra_segments *by_rel_host,
                                        const float *relation, const float *input, const float *output,
                                        const float *output_grad, const float *d_input_add, float *d_input,
                                        float *d_relation, void *workspace, size_t workspace_bytes, int64_t n_src,
                                        int64_t n_dst, int64_t n_rel, int64_t F, int sum_op, int mul_op, void *stream);

/* Backward of sum-aggregation when the caller KNOWS which rows carry gradient (ABI 6).  Same results as
 * ultra_rspmm_backward_accumulate_f32(sum_op = add) bit for bit; the knowledge only lets the kernels skip gathers:
 *   dst_active_bits : [F / 64][active_words] one bitmap over the destination nodes per 64-column tile (query block), bit v of
 *                     tile b set where output_grad[v, tile b] may be non-zero -- the caller's promise that every other row of
 *                     that tile IS zero (the LAST layer of a training step: the score head reads the layer's output at the
 *                     candidate entities' rows only, /root/reference/ultra/model.py:177-183, so its gradient is zero elsewhere;
 *                     ultra_node_bitmap builds the bitmaps from the candidate index grid); NULL: no such knowledge
 *   src_active_node : [F / 64] for d_relation with mul = mul: the one source node per tile whose `input` row is non-zero (the
 *                     FIRST layer: its input is the boundary, model.py:106-107,116-120); NULL: none
 * An inactive edge contributes (+-0): its gathers are issued past the end of their buffer descriptors and return 0.0 without
 * touching memory.  Used by the d_relation kernels (two gathers per edge: 30-45 % faster); d_input, which is bound by its row
 * epilogue rather than by its gathers, and plans the masked kernels do not cover compute everything, as the unmasked entry does.
 * ultra_node_bitmap: bits[b][w] over n_node nodes from t_index int64 [n_batch, per_row] (ids outside [0, n_node) are ignored);
 * at most 512 Ki nodes (ULTRA_ERR_BAD_SHAPE beyond: go without the mask). */
int ultra_rspmm_backward_active_f32(const ultra_segments *by_src, const ultra_segments *by_rel, const float *relation,
                                    const float *input, const float *output_grad, const float *d_input_add, float *d_input,
                                    float *d_relation, void *workspace, size_t workspace_bytes, int64_t n_src, int64_t n_dst,
                                    int64_t n_rel, int64_t F, int mul_op, const uint32_t *dst_active_bits, int64_t active_words,
                                    const int32_t *src_active_node, void *stream);
int ultra_node_bitmap(const int64_t *t_index, int64_t n_batch, int64_t per_row, int64_t n_node, uint32_t *bits, void *stream);

/* d_relation of the FIRST layer (sum aggregation, mul = mul) from the boundary nodes' out-edges alone: the layer's input is zero outside
 * row src_node[b] of 64-column block b (/root/reference/ultra/model.py:106-107,116-120), so only that node's out-edges contribute to
 * block b -- a few thousand of the E edges the d_relation plan holds; ultra_rspmm_backward_active_f32(src_active_node) skips their
 * gathers but still walks all E edge words.  Same bits as that entry (the plan's order: within a relation row by (dst, src), split
 * rows in pieces of piece_len summed from 0 and added in piece order; a skipped edge would have added +0).
 *   items      : int32 [n_items][3] {begin, end, target}, one per piece of a split row (target = -(piece_slot + 1)) and one per
 *                unsplit row (target = the row), [begin, end) in the plan's edge order; n_items = n_pieces + n_rel - n_long_rows
 *   src_ptr    : int32 [n_src + 1] first out-edge of every source node; src_relpos: int32 [E], for each source node the positions
 *                of its out-edges in the plan's edge order, ascending
 *   input      : the layer's input [n_src, F] (row src_node[b] is read at block b); output_grad [n_dst, F]; d_relation [n_rel, F]
 *   workspace  : n_pieces * F floats
 * Built once per graph by the host side (relcsr.RelCSR.boundary_relation_index); per-step edge weights come from by_rel->weight. */
int ultra_rspmm_drelation_boundary_f32(const ultra_segments *by_rel, const int32_t *items, int64_t n_items, const int32_t *src_ptr,
                                       const int32_t *src_relpos, const int32_t *src_node, const float *input,
                                       const float *output_grad, float *d_relation, void *workspace, size_t workspace_bytes,
                                       int64_t n_src, int64_t n_rel, int64_t F, void *stream);

/*
 * d_weight[e] = sum_f output_grad[dst_e, f] * [out == y] * (relation[r_e, f] MUL input[src_e, f])
 * (the value gradient torchdrug returns when sparse.requires_grad), edges in forward-plan order.
 */
int ultra_rspmm_backward_weight_f32(const ultra_segments *fwd_host, const float *relation, const float *input,
                                    const float *output, const float *output_grad, float *d_weight,
                                    int64_t n_rel, int64_t F, int sum_op, int mul_op, void *stream);


/*
 * Dense epilogue of one Bellman-Ford layer, fused (dim must be 64, the shipped architecture):
 *     out = [input +]  relu?( LayerNorm?( Linear( cat[input, update] ) ) )
 * replaces GeneralizedRelationalConv*.combine (/root/reference/ultra/layer.py:184-190, :386-392: cat, nn.Linear,
 * nn.LayerNorm, relu) and, when `shortcut` is set, the caller's `hidden + layer_input`
 * (ultra/model.py:126-127, ultra/rel_model.py:371-372).
 *   input, update, out : [rows, 64] fp32 (a row = one (node, query) pair);  weight [64, 128] = nn.Linear.weight;
 *                        `out` may be the same buffer as `update` (each 32-row tile is read before it is written);
 *   bias [64];  ln_weight / ln_bias [64] or both NULL (no LayerNorm);  relu, shortcut: 0 / 1.
 * relu here and in every other forward entry of this header is torch.relu's: a NaN stays a NaN (`!(v <= 0) ? v : 0`).
 * Forward of the epilogue; its backward (training) is ultra_combine_backward_f32 + ultra_combine_dxdu_f32 below.
 */
int ultra_combine_forward_f32(const float *input, const float *update, const float *weight, const float *bias,
                              const float *ln_weight, const float *ln_bias, float ln_eps, int relu, int shortcut,
                              float *out, float *z_out, int64_t rows, int64_t dim, void *stream);
/* z_out (NULL in inference): [rows, 64], receives z = Linear(cat[input, update]) -- the LayerNorm's input -- for
 * ultra_combine_backward_fused_f32, which then loads it instead of recomputing it (a third of its matrix work). */

/* The same epilogue for the FIRST layer of a Bellman-Ford, whose `input` is the boundary itself (ultra/model.py:116-120):
 * row (v, q) of `input` -- rows = n_nodes * n_query, q fastest -- is boundary_value[q, :] where v == boundary_node[q] and
 * +0 elsewhere (what zeros + scatter_add_ build, ultra/model.py:106-107).  The kernel synthesises those rows instead of
 * reading an (N, B, 64) tensor that is zero outside B rows; identical bits to ultra_combine_forward_f32 on the dense tensor.
 *   boundary_node : int32 [n_query]     boundary_value : fp32 [n_query, 64] (16-byte aligned) */
int ultra_combine_forward_boundary_f32(const int32_t *boundary_node, const float *boundary_value, int64_t n_query,
                                       const float *update, const float *weight, const float *bias, const float *ln_weight,
                                       const float *ln_bias, float ln_eps, int relu, int shortcut, float *out, int64_t rows,
                                       int64_t dim, void *stream);


/*
 * Backward of the fused epilogue above (training): replaces what autograd derives for the reference's
 * cat -> nn.Linear -> nn.LayerNorm -> relu chain (/root/reference/ultra/layer.py:386-392).
 *   grad_out                 : dL/d(out) [rows, 64]  (the shortcut's own pass-through, d_input += grad_out, is the caller's)
 *   d_z                      : OUT  dL/dz [rows, 64], z = Linear(cat[input, update]);  d_input = d_z . W[:, :64],
 *                              d_update = d_z . W[:, 64:]  are plain GEMMs
 *   d_ln_weight_partial,
 *   d_ln_bias_partial        : OUT  [n_ln_waves, 64] per-wave partial sums (sum over dim 0 gives the gradients)
 *   d_weight_partial         : OUT  [n_wgrad_waves, 64 * 128] per-wave partial slabs of d_weight
 *   d_bias_partial           : OUT  [n_wgrad_waves, 64] per-wave column sums of d_z (d_bias partials), or NULL
 * ultra_combine_backward_waves() gives the two wave counts for `rows` on `device`.
 */
int ultra_combine_backward_waves(int device, int64_t rows, int *n_ln_waves, int *n_wgrad_waves);
int ultra_combine_backward_f32(const float *input, const float *update, const float *weight, const float *bias,
                               const float *ln_weight, const float *ln_bias, float ln_eps, int relu,
                               const float *grad_out, float *d_z, float *d_ln_weight_partial,
                               float *d_ln_bias_partial, float *d_weight_partial, float *d_bias_partial, int64_t rows,
                               int64_t dim, void *stream);

/* d_input and d_update of the same backward in one pass over d_z:
 *     d_input = [grad_out +] d_z . weight[:, :64]      d_update = d_z . weight[:, 64:]
 * (what autograd derives for cat -> nn.Linear, ultra/layer.py:386-387, plus the shortcut's pass-through,
 * ultra/model.py:126-127); grad_out NULL when the layer has no shortcut.  All [rows, 64] fp32; weight [64, 128]. */
int ultra_combine_dxdu_f32(const float *d_z, const float *weight, const float *grad_out, float *d_input, float *d_update,
                           int64_t rows, int64_t dim, void *stream);


/* The same backward in ONE pass over the rows (csrc/combine_fused_bwd.inc): a wave keeps a 32-row tile in LDS through the
 * three GEMMs (recompute z, d_weight, d_input | d_update), so the rows are read once (input, update, grad_out) and written
 * once (d_input, d_update) instead of the 8 reads + 3 writes of the two calls above; d_input / d_update carry the same
 * bits as ultra_combine_dxdu_f32, the parameter gradients come out finished (partials added in wave order).
 *   shortcut          : d_input += grad_out (the caller's `hidden + layer_input`, ultra/model.py:126-127)
 *   d_weight [64,128], d_bias [64] or NULL, d_ln_weight / d_ln_bias [64] (ignored without LayerNorm)
 *   partial_workspace : ultra_combine_backward_fused_waves() * (64 * 128 + 192) floats of scratch
 *   z                 : NULL, or the z_out of the layer's ultra_combine_forward_f32 (same bits as the recomputation)
 *   tile_list, n_list : NULL / 0, or the 32-row tiles (row / 32, ascending, < 0 = padding) outside which grad_out is zero by
 *                       the caller's word -- the LAST layer of a Bellman-Ford in training, whose output is read at the candidate
 *                       entities' rows only (/root/reference/ultra/model.py:177-183): d_input / d_update are zero-filled and
 *                       only the listed tiles are computed (z is ignored in this form) */
int ultra_combine_backward_fused_waves(int device, int64_t rows, int *n_waves);
int ultra_combine_backward_fused_f32(const float *input, const float *update, const float *weight, const float *bias,
                                     const float *ln_weight, const float *ln_bias, float ln_eps, int relu, int shortcut,
                                     const float *grad_out, const float *z, float *d_input, float *d_update, float *d_weight,
                                     float *d_bias, float *d_ln_weight, float *d_ln_bias, float *partial_workspace,
                                     size_t workspace_bytes, const int32_t *tile_list, int64_t n_list, int64_t rows, int64_t dim, void *stream);


/*
 * out[rows, out_dim] = relu?( input[rows, in_dim] . weight[out_dim, in_dim]^T + bias ) in a documented summation order
 * (k-ordered fmaf chain from the bias: k = 0, in_dim/2, 1, in_dim/2 + 1, ...; for out_dim == 1: k ascending), so that
 * the small nn.Linear layers around the Bellman-Ford stacks -- relation projection 64 -> 64 -> 64
 * (/root/reference/ultra/layer.py:228,318-319) and score head 128 -> 128 -> 1 (ultra/model.py:53,193) -- give the same
 * bits on the GPU and in the CPU oracle.  Shapes: (64, 64), (128, 128), (in_dim % 4 == 0, 1).  Forward only.
 */
int ultra_linear_forward_f32(const float *input, const float *weight, const float *bias, float *out, int64_t rows,
                             int64_t in_dim, int64_t out_dim, int relu, void *stream);


/*
 * Score head of full-batch evaluation, fused:  out[b, n] = w2 . relu(W1 . cat[hidden[n, b, :], query[b, :]] + b1) + b2
 * replaces cat (/root/reference/ultra/model.py:134-138), transpose + gather of all candidate tails (:177-183, the
 * identity permutation when every entity is a candidate: ultra/task.py:249-253) and the 128 -> 128 -> 1 mlp (:193).
 * hidden [n_node, batch, 64], query [batch, 64], w1 [128, 128], b1 [128], w2 [128], b2 [1], out [batch, n_node];
 * query_bias [batch, 128]: scratch the call fills with the queries' share of the first layer,
 *     c[b, o] = b1[o] + w1[o, 64:] . query[b]                     (fmaf chain k = 64, 96, 65, 97, ... from the bias)
 *     h[o]    = relu(c[b, o] + w1[o, :64] . hidden[n, b])          (chain k = 0, 32, 1, 33, ... from c)
 *     out     = b2 + sum_o h[o] w2[o]                              (o ascending)
 * -- the order oracle/rspmm_oracle.c restates (ABI 5; up to ABI 4 the 128-wide product ran per row).
 */
int ultra_score_forward_f32(const float *hidden, const float *query, const float *w1, const float *b1, const float *w2,
                            const float *b2, float *query_bias, float *out, int64_t n_node, int64_t batch, void *stream);

/* The relation projections of all entity layers in one launch:
 *     out[l][r, b, :] = w2[l] . relu( w1[l] . relation[b, r, :] + b1[l] ) + b2[l]
 * = `relation_projection` (2-layer MLP) of GeneralizedRelationalConvNBFMod followed by the (B, R, D) -> (R, B*D)
 * transpose (ultra/layer.py:228,318-319,325-326), which ultra/model.py:120-130 triggers once per layer.
 *   relation : fp32, row (b, r) = 64 floats at relation + stride_b * b + stride_r * r (contiguous [batch, n_rel, 64]:
 *   n_rel * 64 and 64; the relation stack's own [n_rel, batch, 64] output: 64 and batch * 64; multiples of 4);
 *   w1 / b1 / w2 / b2 / out : HOST arrays of n_layers DEVICE pointers
 *   (weights [64, 64] as nn.Linear.weight, biases [64], out[l] [n_rel, batch * repeat, 64]).  Bit-identical to two
 *   ultra_linear_forward_f32 calls + the transpose.  repeat >= 1: every projected row (b, r) is written to the query
 *   blocks b, b + batch, ...: the tables of `torch.cat([relation] * repeat)` (full-batch evaluation scores the tail and
 *   the head queries of a batch over the same relation representations, ultra/task.py:249-259) at the cost of one. */
int ultra_relation_project_f32(const float *relation, int64_t stride_b, int64_t stride_r, const float *const *w1,
                               const float *const *b1, const float *const *w2, const float *const *b2, float *const *out,
                               int64_t n_layers, int64_t batch, int64_t repeat, int64_t n_rel, int64_t dim, void *stream);

/* The index glue between the relation stack and the entity stack for one evaluation batch, in one launch: from the B
 * triples batch[b] = (h, t, r) (int64 [B, 3], the reference's column order, ultra/task.py:123) and the conditioned
 * relation representations rel_rep [B, n_rel2, 64] (n_rel2 = 2 * n_base_rel relations incl. inverses), the 2B tail-form
 * queries of full-batch evaluation (ultra/task.py:249-259, ultra/model.py:76-83,101-114):
 *     q <  B : anchor = h[q],     relation = r[q]                    q >= B : anchor = t[q - B], relation = r[q - B] + n_base_rel
 *     query[q, :] = rel_rep[q mod B, relation[q], :]
 * Outputs: anchor int64 [2B], anchor32 int32 [2B], relation int64 [2B], query fp32 [2B, 64].  Plain copies: what the
 * reference's cat / add / arange / index kernels produce.  rel_rep row (b, r) = 64 floats at stride_b * b + stride_r * r
 * (as in ultra_relation_project_f32: the relation stack's [n_rel2, B, 64] output is read where it lies). */
int ultra_prepare_queries(const int64_t *batch, const float *rel_rep, int64_t stride_b, int64_t stride_r, int64_t n_batch,
                          int64_t n_rel2, int64_t n_base_rel, int64_t *anchor, int32_t *anchor32, int64_t *relation,
                          float *query, void *stream);

/* What the relation stack (RelNBFNet, /root/reference/ultra/rel_model.py:351-378) needs before its first layer, in one
 * launch instead of stack / expand / ones / cast kernels:
 *     tables[l, r, q, :] = weights[l][r, :]   the layers' relation embeddings (nn.Embedding(4, 64), ultra/layer.py:143-151)
 *                                             tiled over the n_query query blocks: the (n_rel, n_query * 64) operand of rspmm
 *     ones[q, :] = 1                          the boundary value of every query (rel_model.py:355)
 *     node32[q] = (int32) h_index[q * h_stride]   the boundary nodes (the batch's relations: column 2 of the (B, 3)
 *                                             batch is read in place with h_stride = 3)
 * weights: HOST array of n_layers (<= 8) DEVICE pointers, each [n_rel, 64]. */
int ultra_relation_stack_inputs(const float *const *weights, int64_t n_layers, int64_t n_rel, int64_t n_query,
                                const int64_t *h_index, int64_t h_stride, float *tables, float *ones, int32_t *node32,
                                void *stream);

/* Training, last layer: the distinct 32-row tiles (row / 32) of an [n_rows = N * n_query, 64] activation that hold the rows
 * (t_index[b, j] * n_query + b) -- where the gradient of hidden[t_index, arange(B)] (/root/reference/ultra/model.py:177-183) is
 * non-zero -- ascending, -1 padded to n_batch * per_row entries, in one launch: the tile_list of
 * ultra_combine_backward_fused_f32.  n_rows <= 32 Mi (ULTRA_ERR_BAD_SHAPE above: the caller builds the list itself); t_index
 * entries in [0, n_rows / n_query), unchecked. */
int ultra_candidate_tiles(const int64_t *t_index, int64_t n_batch, int64_t per_row, int64_t n_query, int64_t n_rows, int32_t *out,
                          void *stream);

/* out[q, :] = rows[node[q], q, :] for rows [n_node, n_query, 64]: the gradient of the boundary VALUES out of a layer's d_update
 * (backward of the sparse boundary epilogue of ultra_rspmm_forward_boundary_f32; /root/reference/ultra/model.py:106-107's
 * scatter_add_ backward) in one launch. */
int ultra_gather_boundary_rows_f32(const float *rows, const int32_t *node, int64_t n_query, float *out, void *stream);

/* The score head on the candidate rows of a training step, forward and backward (csrc/score_rows.inc):
 *     score[b, j] = w2 . relu( w1 . cat[ hidden[t_index[b, j], b, :], query[b, :] ] + b1 ) + b2
 * = /root/reference/ultra/model.py:177-183,193 (gather of the candidate tails, concatenation with the query, the 128 -> 128 -> 1
 * mlp) with the candidates picked before the concatenation.  hidden [n_node, n_batch, 64], query [n_batch, 64], t_index int64
 * [n_batch, per_row], w1 [128, 128], b1 [128], w2 [128], b2 [1]; h and in_rows [n_batch * per_row, 128] receive relu(.) and the
 * gathered cat[.] rows for the backward.  Backward: grad [n_batch, per_row]; scratch d_pre [n_batch * per_row, 128] and partial
 * [16 * 129 * 129]; d_hidden [n_node, n_batch, 64] is zero-filled and
 * the candidate rows written (rows repeated inside a query are added in row order); d_query, d_w1, d_b1, d_w2, d_b2 come out
 * finished, every sum in a fixed order (the backward does not read `hidden`: NULL is accepted there).  per_row <= 160 (the
 * backward keeps a query's rows in LDS).  t_index entries must lie
 * in [0, n_node): like the reference's gather they are not checked on the device (engine.validate_triples and the strict
 * negative sampler guarantee it). */
int ultra_score_rows_forward_f32(const float *hidden, const float *query, const int64_t *t_index, const float *w1, const float *b1,
                                 const float *w2, const float *b2, float *h, float *in_rows, float *score, int64_t n_batch,
                                 int64_t per_row, void *stream);
int ultra_score_rows_backward_f32(const float *hidden, const float *query, const int64_t *t_index, const float *w1, const float *w2,
                                  const float *h, const float *in_rows, const float *grad, float *d_pre, float *partial,
                                  float *d_hidden, float *d_query, float *d_w1, float *d_b1, float *d_w2, float *d_b2,
                                  int64_t n_node, int64_t n_batch, int64_t per_row, void *stream);

/* Training metrics: norm, mean and unbiased standard deviation of the values { a[0 .. n_a) } together with every b[0 .. n_b)
 * taken b_repeat times, in two launches with double-precision accumulation:  out[0..2] = (norm, mean, std).
 * The reference logs them in every training forward for the relation representations (`query_*`,
 * /root/reference/ultra/model.py:158-160) and for node_feature = cat[hidden, query] (`output_*`, :178-181; the query half
 * is B vectors repeated for every node: b, b_repeat = n_node, never materialised).  partials: scratch of
 * 2 * ultra_statistics_blocks(n_a) doubles; a 16-byte aligned. */
int ultra_statistics_blocks(int64_t n_a);
int ultra_statistics_f32(const float *a, int64_t n_a, const float *b, int64_t n_b, int64_t b_repeat, double *partials,
                         float *out, void *stream);

/* The training criterion and its gradient in one launch (/root/reference/ultra/task.py:169-180: binary cross entropy with
 * logits over (positive | K negatives), self-adversarial negative weights softmax(pred[:, 1:] / T) without gradient -- or
 * 1 / K when temperature <= 0 --, weighted mean per row):  pred fp32 [rows, cols] with the positive in column 0;
 * loss_rows [rows]; dpred [rows, cols] = d loss_rows[row] / d pred[row, col]. */
int ultra_bce_adversarial_f32(const float *pred, int64_t rows, int64_t cols, float temperature, float *loss_rows,
                              float *dpred, void *stream);

/* Backward of ultra_relation_project_f32 for all layers in one launch (training):
 *     d_w1[l], d_b1[l], d_w2[l], d_b2[l]   gradients of the layer's four parameters ([64, 64] / [64], overwritten)
 *     d_relation_layers[l, b * n_rel + r, :]   the layer's gradient of relation[b, r, :]; the input feeds every layer,
 *                                              so the caller sums over l
 * from grad[l] = the gradient of out[l] ([n_rel, batch, 64]; a NULL entry = no gradient reached that table: zeros).
 * Replaces the autograd chain of 2 x n_layers nn.Linear + relu + transpose the reference runs for
 * `relation_projection` (ultra/layer.py:228,318-319,325-326): 6 GEMMs + 2 bias reductions + elementwise passes per
 * layer.  The hidden activation is recomputed with the forward kernel's chain.  n_layers <= 8.
 *   w1 / b1 / w2 / grad / d_w1 / d_b1 / d_w2 / d_b2 : HOST arrays of n_layers DEVICE pointers.
 *   workspace : device scratch of n_layers * blocks * (2 * 64 * 64 + 128) floats, `blocks` from
 *   ultra_relation_project_backward_blocks (depends on the device's CU count and the shape only). */
int ultra_relation_project_backward_blocks(int device, int64_t batch, int64_t n_rel, int64_t n_layers, int64_t *blocks);
int ultra_relation_project_backward_f32(const float *relation, const float *const *w1, const float *const *b1,
                                        const float *const *w2, const float *const *grad, float *d_relation_layers,
                                        float *const *d_w1, float *const *d_b1, float *const *d_w2, float *const *d_b2,
                                        void *workspace, size_t workspace_bytes, int64_t n_layers, int64_t batch,
                                        int64_t n_rel, int64_t dim, void *stream);

/* Filtered ranking on the device, from filter LISTS instead of dense masks.
 * Replaces: get_ranking, ultra/task.py:307-315 -- `sum((pos_pred <= pred) & mask, -1) + 1` -- together with the dense
 * (B, N) boolean masks of ultra/task.py:65-100 (`mask[pos_index, truth_index] = 0`) that feed it.
 *   pred       : fp32 [n_query, row_stride], the first n_cand entries of a row are the candidate scores
 *   target     : int64 [n_query] index of the positive candidate
 *   filt_ptr   : int32 [n_query + 1], filt_node : int32 [filt_ptr[n_query]] -- per query the DISTINCT candidates the
 *                mask would set to 0 (known truths, the positive included when it is one); NULL: unfiltered ranking
 *   rank       : int64 [n_query]  (1-based) */
int ultra_filtered_rank(const float *pred, int64_t n_query, int64_t n_cand, int64_t row_stride, const int64_t *target,
                        const int32_t *filt_ptr, const int32_t *filt_node, int64_t *rank, void *stream);


/*
 * Per-step index work from SORTED KEY ARRAYS instead of dense (B, N) masks (csrc/sampler.inc).  A graph keeps, per
 * direction, the sorted DISTINCT int64 keys  (anchor * n_rel + rel) * n_node + other  of its triples: for tail
 * prediction anchor = head, other = tail; for head prediction anchor = tail, other = head.  All three calls only
 * enqueue work: no allocation, no host synchronisation, capturable into a hipGraph.
 *
 * ultra_filtered_rank_keys: get_ranking (ultra/task.py:307-315) with the filter mask of ultra/task.py:65-100 looked up
 *   in `keys`:  rank = 1 + #{c : pos <= pred[c]} - #{c completes (anchor_q, rel_q, ?) : pos <= pred[c]}.
 *   Query q reads pred + q * row_stride (n_cand = n_node scores), target[q * target_stride], anchor / rel
 *   [q * index_stride] and writes rank[q * rank_stride]; keys == NULL: unfiltered ranking.  Rows of more than 64 K
 *   candidates are counted by several workgroups each (ABI 8: one workgroup per query took 32 ms for two rows of 10 M
 *   candidates): a first launch writes 1 - (filtered count), a second adds the slices' counts with integer atomics -- the
 *   same int64 whatever the order.
 * ultra_strict_negative: strict negative sampling (ultra/task.py:102-118) without mask.nonzero(): out[q, s] is the
 *   floor(rand[q, s] * n_free_q)-th entity in ascending order that does NOT complete (anchor_q, rel_q, ?) -- the entity
 *   torchdrug's variadic_sample picks for the same uniform numbers.  rand fp32 [n_query, n_sample] in [0, 1).
 * ultra_edge_removal_weights: remove_easy_edges (ultra/model.py:57-74, remove_one_hop = False) as edge weights: the
 *   three plans of the graph WITH inverse edges get weight arrays (n_edges + slack floats each) equal to their own
 *   weights (1.0 when NULL, 1.0 in the slack) except 0.0 at every edge (h, t, r) / (t, h, r + n_base_rel) of the
 *   n_pattern triples that exists -- duplicates of a triple are one coalesced edge, so all of them go, as in the
 *   reference.  Triples that are not edges (the negatives of the batch) change nothing.
 */
int ultra_filtered_rank_keys(const float *pred, int64_t n_query, int64_t n_cand, int64_t row_stride, const int64_t *target,
                             int64_t target_stride, const int64_t *keys, int64_t n_keys, const int64_t *anchor,
                             const int64_t *rel, int64_t index_stride, int64_t n_rel, int64_t *rank, int64_t rank_stride,
                             void *stream);
int ultra_strict_negative(const int64_t *keys, int64_t n_keys, const int64_t *anchor, const int64_t *rel, int64_t n_query,
                          int64_t n_rel, int64_t n_node, const float *rand, int64_t n_sample, int64_t *out, void *stream);
int ultra_edge_removal_weights(const ultra_segments *fwd, const ultra_segments *by_src, const ultra_segments *by_rel,
                               const int64_t *h, const int64_t *t, const int64_t *r, int64_t n_pattern,
                               int64_t n_base_rel, float *w_fwd, float *w_src, float *w_rel, int64_t slack, void *stream);
/* As ultra_edge_removal_weights, and ALSO the marked word copies that ultra_segments.packed_dead takes (any of them NULL: that plan
 * gets none): words_x = the plan's packed words (n_edges + slack of them) with bit 31 set where w_x is set to 0.  Only for plans
 * WITHOUT weights of their own (every weight exactly 1) whose packed words leave bit 31 free (node ids inside the word, id range
 * below 2^(31 - packed_src_shift)): ULTRA_ERR_BAD_SHAPE otherwise.  One fill launch for all six arrays + the search launch. */
int ultra_edge_removal_marks(const ultra_segments *fwd, const ultra_segments *by_src, const ultra_segments *by_rel,
                             const int64_t *h, const int64_t *t, const int64_t *r, int64_t n_pattern, int64_t n_base_rel,
                             float *w_fwd, float *w_src, float *w_rel, int64_t slack, uint32_t *words_fwd, uint32_t *words_src,
                             uint32_t *words_rel, int64_t n_node, void *stream);


/*
 * Native plan builder (rocPRIM radix sort + scans), replaces sparse.coalesce() + coo2csr that torchdrug runs inside
 * every generalized_rspmm call (/root/reference/ultra/layer.py:127,328 pass an un-coalesced adjacency).
 *
 * ultra_relcsr_coalesce: sort the (row, col, rel) triples, merge duplicates by summing their weights (weight may be
 *   NULL = ones).  Outputs have capacity n_edges; *n_unique_host / *unit_weight_host (every merged weight == 1.0f)
 *   are valid on return (the call synchronises the stream).  edge_of_input[i] = index of input edge i in the output.
 * ultra_relcsr_plan: chunk schedule + packed words for ONE ordered plan (`row` non-decreasing).  Capacities:
 *   chunks >= n_rows + n_edges / piece_len + 2 entries of 4 ints, long_rows >= n_edges / piece_len + 1 entries of
 *   3 ints, packed n_edges + packed_slack ints (or NULL).  counts_host[4] = {n_chunks, n_long_rows, n_pieces,
 *   packed_src_shift (0: none, 32: node ids stay in node_a)}.  is_relation_plan: rows are relations (no relation field).
 * The *_temp_bytes functions give the scratch size each call needs.
 */
size_t ultra_relcsr_coalesce_temp_bytes(int64_t n_edges);
int ultra_relcsr_coalesce(const int64_t *row, const int64_t *col, const int64_t *rel, const float *weight,
                          int64_t n_edges, int64_t n_rows, int64_t n_cols, int64_t n_rel, int32_t *out_row,
                          int32_t *out_col, int32_t *out_rel, float *out_weight, int64_t *edge_of_input,
                          int64_t *n_unique_host, int *unit_weight_host, void *temp, size_t temp_bytes, void *stream);
size_t ultra_relcsr_plan_temp_bytes(int64_t n_edges, int64_t n_rows, int64_t piece_len);
int ultra_relcsr_plan(const int32_t *row, const int32_t *node_a, const int32_t *rel, int64_t n_edges, int64_t n_rows,
                      int64_t n_node_a, int64_t n_rel, int is_relation_plan, int wide_ids, int balance,
                      int64_t chunk_edges, int64_t chunk_rows, int64_t piece_len, int32_t *chunks,
                      int64_t cap_chunks, int32_t *long_rows, int64_t cap_long, int32_t *packed,
                      int64_t packed_slack, int64_t *counts_host, void *temp, size_t temp_bytes, void *stream);

/*
 * Dense form of a plan (ABI 7).  construct_relation_graph (/root/reference/ultra/rel_model.py:99-143) builds the graph of
 * relations from products of incidence matrices: 2R nodes, 4 edge types, unit weights, and dense by nature.  For such a
 * plan the sum aggregation is a matrix product with a 0/1 matrix, and v_mfma_f32_16x16x4_f32 computes
 *     acc = fmaf(a0, b0, acc); acc = fmaf(a1, b1, acc); acc = fmaf(a2, b2, acc); acc = fmaf(a3, b3, acc)
 * (sequential, every step rounded like fmaf: tools/ubench/mfma_order.hip).  With a_k in {0, 1} and b_k = the ROUNDED message
 * relation[k] (*|+) input[u] of edge type k, one instruction per gathered node u adds that node's (up to) four messages in
 * relation order -- fmaf(1, y, acc) = acc + y, fmaf(0, y, acc) = acc for finite y -- so a row's sum is the strictly sequential
 * (node, relation) sum of the reference for EVERY row: no pieces, no fix-up pass.
 *   kind 0 (forward / d_input plan: rows are nodes):  byte [tile][col / 4][type][i][col % 4]                  = 1 iff edge (row 16 tile + i, node_a col, type)
 *   kind 1 (d_relation plan: rows are the 4 types):   byte [type][tile][col / 16][col % 4][i][(col / 4) % 4] = 1 iff edge (node_b 16 tile + i, node_a col, type)
 * (one byte per entry, the four entries a lane needs for consecutive MFMAs in one 32-bit word; columns padded to whole rounds of
 * the kernels' register sets + a few hundred bytes the kernels may read past the end.)
 * d_relation in this form has its own documented order (the reference's is a sum over all (dst, src) pairs of one type):
 *     S[v][t] = sequential sum over sources u ascending of input[u] where edge (u -> v, t) exists   (exact adds)
 *     tile sum T[tile][t] = ((q0 + q1) + q2) + q3,  q_k = ((P[4k] + P[4k+1]) + P[4k+2]) + P[4k+3],  P[j] = grad[16 tile + j] * S[16 tile + j][t]
 *     d_relation[t] = sequential sum over tiles ascending of T[tile][t]
 * (oracle/rspmm_oracle.c restates it; within rounding of the reference order, tests/test_relgraph_dense_gpu.py).
 * Preconditions checked at dispatch, otherwise the edge list is walked: weight == NULL, n_rel == 4, sum = add, F % 16 == 0
 * (d_relation: mul = mul and a workspace of ultra_rspmm_workspace_bytes()).  Inputs must be finite for exact equivalence
 * (0 * inf = NaN where the edge list would skip a missing edge).
 * ultra_relcsr_dense_bytes: size of the matrix (0 when the shape is not supported);  ultra_relcsr_dense: fill it from the
 * plan's edge arrays (zero fill + one byte store per edge; the caller then sets plan->dense / dense_rows / dense_cols).
 */
size_t ultra_relcsr_dense_bytes(int64_t n_rows, int64_t n_cols, int kind);
int ultra_relcsr_dense(const ultra_segments *plan, int64_t n_rows, int64_t n_cols, int kind, uint32_t *dense, void *stream);

/*
 * One whole layer of the relation-graph Bellman-Ford in inference on a plan that carries its dense form (ABI 7):
 *     out = [input +] relu(LayerNorm(Linear_{128->64}(cat[input, rspmm_{add,mul}(input) + boundary])))
 * = GeneralizedRelationalConvNBF.forward (/root/reference/ultra/layer.py:111-190) + the shortcut of ultra/rel_model.py:371-372
 * -- ultra_rspmm_forward_boundary_f32 followed by ultra_combine_forward_f32, bit for bit, in one launch: the workgroup that sums
 * 16 nodes x one query's 64 columns holds 16 complete rows of the epilogue and finishes them on the same matrix cores.
 *   input [n, n_query, 64] (also the gathered matrix [n, n_query * 64]);  relation [4, n_query * 64];  out [n, n_query, 64],
 *   not aliasing input;  boundary as in ultra_rspmm_forward_boundary_f32 with block = 64;  weight [64, 128].
 * ultra_dense_layer_supported: 1 where the entry applies (dense form present, square adjacency, sizes within 32-bit offsets).
 */
int ultra_dense_layer_supported(const ultra_segments *fwd, int64_t n_query);
int ultra_dense_layer_forward_f32(const ultra_segments *fwd, const float *relation, const float *input,
                                  const int32_t *boundary_node, const float *boundary_value, int64_t n_query, const float *weight,
                                  const float *bias, const float *ln_weight, const float *ln_bias, float ln_eps, int relu,
                                  int shortcut, float *out, void *stream);

/*
 * One whole layer of the ENTITY Bellman-Ford in inference as one launch (ABI 8; csrc/layer_fused.hip), for plans that run one row
 * per lane group (big graphs: row_ptr present, no split rows -- BASELINE config 5, S-stress):
 *     out = [input +] relu(LayerNorm(Linear_{128->64}(cat[input, rspmm_{add,mul}(input) + boundary])))
 * = GeneralizedRelationalConv*.message_and_aggregate + combine (/root/reference/ultra/layer.py:298-392: rspmm :357, + boundary :358,
 * combine :386-392) + the shortcut of ultra/model.py:126-127 -- ultra_rspmm_forward_boundary_f32 followed by
 * ultra_combine_forward_f32, bit for bit, with the epilogue INSIDE the rspmm's row loop: a wave stages the four epilogue rows it
 * finishes per iteration (and their own input segments) in LDS and runs the 128 -> 64 product on the exact-f32 matrix cores every
 * fourth iteration (v_mfma_f32_16x16x4_f32, K = (in[s], up[s], in[s+1], up[s+1]): the chain of ultra_combine_forward_f32), so the
 * (N, B, 64) tensor `update` is never written or read: 2 of a layer's 5 row-sized streams.
 *   fwd: the forward plan (row_ptr != NULL, n_pieces == 0: ultra_layer_forward_supported);  input [n_rows, n_query, 64] (the
 *   gathered matrix AND the epilogue's `input`);  relation [n_rel, n_query * 64];  boundary as in ultra_rspmm_forward_boundary_f32
 *   with block = 64 (both NULL: no boundary term);  weight [64, 128];  out [n_rows, n_query, 64], NOT aliasing input.
 */
int ultra_layer_forward_supported(const ultra_segments *fwd, int64_t n_query, int64_t n_rel);
int ultra_layer_forward_f32(const ultra_segments *fwd, const float *relation, const float *input, const int32_t *boundary_node,
                            const float *boundary_value, int64_t n_query, const float *weight, const float *bias,
                            const float *ln_weight, const float *ln_bias, float ln_eps, int relu, int shortcut, float *out,
                            int64_t n_rel, void *stream);

/*
 * The SECOND entity layer of a Bellman-Ford in inference on such plans (ABI 8).  After ultra_first_layer_sparse_f32 every row of
 * the first layer's output is ONE constant vector except the rows it LISTED (a few dozen on a graph like S-stress: the boundary
 * nodes' out-neighbours), and the second layer's rspmm gathers that output once per edge.  ultra_second_layer_sources writes, for
 * every edge of the forward plan, the row to gather: the edge's own source if it is listed, ONE fixed unlisted row otherwise --
 * the same values, the same sums (results bit for bit), and all but a few hundred of the 100 M gathers hit the cache.
 *   col: the forward plan's node_a [n_edges (+ slack readable words)];  row_list / list_count / list_len: as written by
 *   ultra_first_layer_sparse_f32 (row ids node * n_query + query, -1 = empty; the count on the device);  bitmap: uint32
 *   [(n_node + 31) / 32] scratch;  c_node: int32 [1] scratch;  sources: int32 [n_edges + slack] out.  Needs list_len + 1 < n_node.
 * ultra_layer_forward_sources_f32 = ultra_layer_forward_f32 gathering row sources[e] for edge e.
 */
int ultra_second_layer_sources(const int32_t *col, int64_t n_edges, int64_t slack, const int32_t *row_list, const int32_t *list_count,
                               int64_t list_len, int64_t n_query, int64_t n_node, uint32_t *bitmap, int32_t *c_node,
                               int32_t *sources, void *stream);
int ultra_layer_forward_sources_f32(const ultra_segments *fwd, const int32_t *sources, const float *relation, const float *input,
                                    const int32_t *boundary_node, const float *boundary_value, int64_t n_query, const float *weight,
                                    const float *bias, const float *ln_weight, const float *ln_bias, float ln_eps, int relu,
                                    int shortcut, float *out, int64_t n_rel, void *stream);

/*
 * The LAST entity layer of full-batch evaluation with the score head inside the same launch (ABI 8; csrc/layer_fused.hip):
 *     score[q, n] = w2 . relu(W1 . cat[hidden_L[n, q], query[q]] + b1) + b2,   hidden_L = ultra_layer_forward_f32(...)
 * = ultra_layer_forward_f32 followed by ultra_score_forward_f32 (/root/reference/ultra/model.py:134-138,177-193 after the last
 * layer of :120-130), bit for bit: the 16 finished rows of a flush go through the head's 64 -> 128 product on the matrix cores
 * (the chain  c[q], hid[0], hid[32], hid[1], hid[33], ...  of score_kernel), relu and the w2 dot (one lane per row, o ascending)
 * before anything leaves: the last layer's (N, Q, 64) output is neither written nor read again, only (Q, N) scores are stored.
 *   query [n_query, 64];  w1 [128, 128], b1 [128], w2 [128], b2 [1];  qbias: fp32 [n_query, 128]
 *   scratch (the queries' share of the head's first layer, written by a small launch in front);  score [n_query, n_rows].
 *   n_query <= 32, n_query * n_rows * 4 B < 4 GiB (ultra_layer_score_supported).
 */
int ultra_layer_score_supported(const ultra_segments *fwd, int64_t n_query, int64_t n_rel);
int ultra_layer_score_forward_f32(const ultra_segments *fwd, const float *relation, const float *input, const int32_t *boundary_node,
                                  const float *boundary_value, int64_t n_query, const float *weight, const float *bias,
                                  const float *ln_weight, const float *ln_bias, float ln_eps, int relu, int shortcut,
                                  const float *query, const float *w1, const float *b1, const float *w2, const float *b2,
                                  float *qbias, float *score, int64_t n_rel, void *stream);

/*
 * The graph of relations, natively (ABI 8): construct_relation_graph, /root/reference/ultra/rel_model.py:91-143.  The reference
 * multiplies the (2R x N) and (N x 2R) incidence matrices of the graph with inverse edges four ways -- Eh^T Eh, Et^T Et, Eh^T Et,
 * Et^T Eh -- and keeps the INDICES of each product (`block.coalesce().indices()`, :131-139): relations r1, r2 get an edge of
 * type 0 (head-head), 1 (tail-tail), 2 (head-tail), 3 (tail-head) iff some entity is the head / tail of an r1 edge and the
 * head / tail of an r2 edge.  Here, from the per-entity lists of DISTINCT incident relations:
 *   head_ptr int32 [n_node + 1], head_rel int32 [head_ptr[n_node]]   relations entity e is the head of (any order inside e)
 *   tail_ptr, tail_rel                                               ... the tail of
 *   marks uint8 [4][n_rel][n_rel], 4-byte aligned, written completely: marks[type][r1][r2] = 1 iff that edge exists
 * One wave per entity marks the pairs of its lists with plain byte stores of the same value (no atomics, nothing to order).
 * `marks[type].nonzero()` is the block's index list in the reference's (row, column) order.  n_rel <= 32 768.
 */
int ultra_relation_graph_marks(const int32_t *head_ptr, const int32_t *head_rel, const int32_t *tail_ptr, const int32_t *tail_rel,
                               int64_t n_node, int64_t n_rel, uint8_t *marks, void *stream);

/*
 * On-box calibration for the HBM roofline line of bench.py ("confirm on the box with a copy / gather
 * calibration, report both"): the bare gather of `n_index` random 256-byte rows of `table` (n_rows x 64 fp32) -- four rows
 * per wave-instruction (buffer_load_dwordx4), eight instructions in flight per wave, every lane's values summed into
 * out[n_waves * 64] so that nothing is dropped -- i.e. the rowgroup kernel's access pattern with the relation operand, the
 * row pointers, the reduction structure and the output rows taken away.  `*n_waves_host` receives the number of waves
 * (out must hold 64 floats per wave; query it with out == NULL).  index values must be < n_rows; n_index is rounded down to
 * a multiple of 2 048 per wave.
 */
int ultra_calibrate_gather_f32(const float *table, int64_t n_rows, const int32_t *index, int64_t n_index, float *out,
                               int64_t *n_waves_host, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* ULTRA_RSPMM_H */
