/*
 * sgnn_hip.h — C ABI of libsgnn_hip.so, the MI355X (gfx950) implementation of
 * SGNN's sparse propagation path:
 *
 *     A_hat = D^-1/2 (A + I) D^-1/2            (adjacency build)
 *     Y     = A_hat^K . X                      (repeated CSR SpMM)
 *
 * The reference (tavogus55/SGNN) is pure Python and has no FFI layer of its
 * own; each entry point below names the reference code it replaces
 * (paths relative to the reference tree).  The Python surfaces that the
 * reference's callers see (`csr_to_sparse_Tensor`,
 * `process_data_with_adjacency_high_order`, `SGConv`, `GCNConv`) live in
 * sgnn_amd/ and call these functions through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every buffer; the library never allocates or frees
 *     device memory (scratch comes in through `workspace`);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and the
 *     call returns without synchronising;
 *   - return value 0 = success, negative = error; the message for the calling
 *     thread's last error is returned by sgnn_last_error();
 *   - indices in CSR are int32 (n and nnz must stay below 2^31), values fp32.
 */
#ifndef SGNN_HIP_H
#define SGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGNN_OK                0
#define SGNN_ERR_INVALID      -1   /* bad argument (null pointer, negative size, n >= 2^31 ...) */
#define SGNN_ERR_WORKSPACE    -2   /* workspace too small */
#define SGNN_ERR_HIP          -3   /* a HIP call failed; see sgnn_last_error() */
#define SGNN_ERR_UNSUPPORTED  -4

/* build flags */
#define SGNN_STRIP_DIAG        1   /* drop (i,i) entries of the input first: model/SGNN.py:497-501 */
#define SGNN_DROP_ZEROS        2   /* drop entries whose merged value is 0 (scipy's sparse subtraction does) */
#define SGNN_SYMMETRIZE        4   /* every input (r, c, v) also counts as (c, r, v): adj.T + adj of the loaders,
                                      data_loader.py:451,476,490,507 (diagonal input entries count twice, as there) */
#define SGNN_CLAMP_ONE         8   /* merged input weight = min(sum, 1), before any self-loop is added:
                                      adj.minimum(1), data_loader.py:452,477,491,508 */
/* self-loop modes */
#define SGNN_LOOPS_NONE        0
#define SGNN_LOOPS_ADD         1   /* A + I, diagonal value += 1: utils.py:83-86 */
#define SGNN_LOOPS_REMAINING   2   /* diagonal = weight of the LAST diagonal input entry, else 1: torch_geometric gcn_norm's
                                      add_remaining_self_loops, reached from model/SGC.py:14, model/GCN.py:14,16 */

/* Library version (major*10000 + minor*100 + patch).  No reference counterpart. */
int sgnn_version(void);

/* Override a kernel-selection policy for experiments (process-wide, not thread
 * safe).  Keys: "tile" = feature-slice width in floats owned by one wavefront,
 * "unroll" = row fetches issued back to back (2 or 4), "rg_seg" = merged columns per
 * row-group work item, "rg_R" = rows per group, 8 or 16, "rg_order" = width of the column blocks the work
 * items are ordered by (all three affect plans built afterwards), "tg_seg" = stages per
 * tile-group work item, "tg_order" = order of the tile-group work items (0: chosen by the
 * graph, the default; 1: longest first; -1: neighbours together; > 1: by column blocks of
 * this width), "tg_combine_gy" = workgroups per split group of the tile-group combine pass,
 * "tg_split_blocks" / "tg_split_threads" = launch shape of the tile-group split pass,
 * "tg_no_sample" = 1 takes the per-feature maxima from every row instead of a sample first,
 * "tg_mode" = diagnostic switches of the tile-group kernel,
 * "time_kernel" = 1/0 turns the kernel timer on/off (see sgnn_kernel_time_ms); value 0
 * restores the built-in policy.  No reference counterpart. */
int sgnn_tune_set(const char* key, int value);

/* With sgnn_tune_set("time_kernel", 1) every SpMM entry point brackets its main kernel
 * (k_tg_spmm, k_rg_spmm or k_spmm; not the scaling, split or combine kernels around it)
 * with a pair of HIP events on the launch stream.  This call waits for the recorded
 * launches, returns the sum of their durations in milliseconds and their number, and
 * starts a new count (at most 8192 launches are kept between two reads).  The
 * reference's counterpart is the wall-clock print at GNN_tasks.py:133-140; this is the
 * per-kernel figure a roofline needs.  Not thread safe. */
int sgnn_kernel_time_ms(double* total_ms, int64_t* launches);

/* Message of the calling thread's most recent failure ("" if none).
 * The reference raises Python exceptions; the host layer turns a negative
 * return code plus this string into a RuntimeError. */
const char* sgnn_last_error(void);

/* Bytes of device scratch sgnn_build_norm_csr needs for an n x n matrix given
 * as nnz_in COO entries (pass 2 * nnz_in when building with SGNN_SYMMETRIZE). */
size_t sgnn_build_workspace_bytes(int64_t n, int64_t nnz_in);

/*
 * COO (int64 row/col, fp32 val, duplicates allowed, any order) ->
 * canonical CSR of the matrix the propagation multiplies by, its weighted row
 * degree and dinv = 1/sqrt(deg).
 *
 * Replaces: utils.py:54-62 (scipy -> torch COO upload), utils.py:82-89
 * (self-loop COO, A + I, torch.sparse.sum row degree, sqrt, reciprocal — all
 * redone by the reference on every call) and, with SGNN_STRIP_DIAG,
 * model/SGNN.py:497-501.
 *
 *   val        nullable: every entry has weight 1
 *   rowptr     [n+1] int32 out
 *   colidx     [nnz_in + n] int32 out (capacity; *nnz_out entries are valid);
 *              [2 * nnz_in + n] with SGNN_SYMMETRIZE
 *   vals       same capacity, fp32 out
 *   deg        [n] fp32 out: row sums in fp32 (64 strided partial sums folded in halves;
 *              exact, hence order-free, for integer-valued weights)
 *   dinv       [n] fp32 out: 1.0f / sqrtf(deg) (two roundings, as utils.py:88-89);
 *              with zero_inf != 0 an infinite result is replaced by 0 (gcn_norm)
 *   nnz_out    device int64 out
 * Duplicates are summed in fp32 in input order (with SGNN_SYMMETRIZE: the entries as
 * given first, then their mirror images in input order); columns are sorted inside a row.
 */
int sgnn_build_norm_csr(int64_t n, int64_t nnz_in,
                        const int64_t* row, const int64_t* col, const float* val,
                        int flags, int self_loops, int zero_inf,
                        int32_t* rowptr, int32_t* colidx, float* vals,
                        float* deg, float* dinv, int64_t* nnz_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Bytes of device scratch sgnn_csr_transpose needs. */
size_t sgnn_transpose_workspace_bytes(int64_t n, int64_t nnz);

/*
 * CSR of the transpose of an n x n canonical CSR (columns sorted inside a row).
 * Needed for the backward of GCNConv (A_hat^T . dY, model/GCN.py:14-16 under
 * autograd) when the adjacency is not symmetric.
 */
int sgnn_csr_transpose(int64_t n, int64_t nnz,
                       const int32_t* rowptr, const int32_t* colidx, const float* vals,
                       int32_t* t_rowptr, int32_t* t_colidx, float* t_vals,
                       void* workspace, size_t workspace_bytes, void* stream);

/* epilogue modes of sgnn_spmm_csr_f32 */
#define SGNN_EPI_SCALE_ROW     0   /* Y_i = dinv_row[i] * s_i                      (utils.py:95) */
#define SGNN_EPI_SCALE_ROW2    1   /* Y_i = (dinv_row[i] * s_i) * dinv_row[i]: utils.py:95 of one hop
                                      followed by utils.py:93 of the next, same two roundings */

/* Bytes of device scratch one sgnn_spmm_csr_f32 / sgnn_spmm_chain_f32 call needs
 * (partial sums of rows that are split over several wavefronts). */
size_t sgnn_spmm_workspace_bytes(int64_t n_rows, int64_t nnz, int64_t F);

/*
 * Y[n_rows, F] = diag(dinv_row) . CSR . diag(dinv_col) . X      (fp32)
 *
 * Replaces one trip of the loop at utils.py:92-95 (row-scale, adj.mm,
 * row-scale).  X has n_cols rows (n_cols = n_rows except for a row block of a
 * partitioned matrix), nnz = rowptr[n_rows] (the host knows it; it sizes the
 * scratch layout), leading dimensions ldx/ldy in elements.
 *   vals      nullable: all stored entries are 1
 *   dinv_col  nullable: X is already scaled (or no column scaling wanted)
 *   dinv_row  nullable: no row scaling
 * Each output row is reduced in an order that depends only on that row's
 * entries, so results are reproducible run to run and identical for a row
 * whichever row block it is computed in.  X and Y must not overlap.
 */
int sgnn_spmm_csr_f32(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t F,
                      const int32_t* rowptr, const int32_t* colidx, const float* vals,
                      const float* dinv_row, const float* dinv_col,
                      const float* X, int64_t ldx, float* Y, int64_t ldy,
                      int epilogue, void* workspace, size_t workspace_bytes, void* stream);

/*
 * sgnn_spmm_csr_f32 for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): same arguments without
 * the workspace, same kernels, same results bit for bit.  It has no counterpart in the
 * reference.  The scratch (segment lists and partial sums of rows longer than 1024
 * entries) lives in a 32 MiB device array inside the library, one copy per GPU, held
 * until the process ends; the call neither allocates nor frees anything and returns
 * SGNN_ERR_WORKSPACE when sgnn_spmm_workspace_bytes(n_rows, nnz, F) exceeds 32 MiB
 * (about nnz * F <= 4e9).  The device the kernels run on must be the current device.
 * The first call on a device looks the array's address up (make it outside graph capture).
 * All calls on one GPU share the array: calls that overlap in time on one GPU (two
 * streams, two host threads) race; calls on one stream are ordered and safe.
 */
int sgnn_spmm_csr_noalloc_f32(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t F,
                              const int32_t* rowptr, const int32_t* colidx, const float* vals,
                              const float* dinv_row, const float* dinv_col,
                              const float* X, int64_t ldx, float* Y, int64_t ldy,
                              int epilogue, void* stream);

/*
 * Y = A_hat^K . X for a square matrix: K hops of the kernel above with the
 * normalisation between hops folded into the epilogue (SGNN_EPI_SCALE_ROW2), so
 * no separate scaling pass touches the intermediates.  tmp is [n, F] fp32
 * scratch (ld = F) and is used only when K > 1.
 * Replaces utils.py:91-96 (whole loop) and the K x propagate of SGConv
 * (model/SGC.py:11,14).
 */
int sgnn_spmm_chain_f32(int K, int64_t n, int64_t nnz, int64_t F,
                        const int32_t* rowptr, const int32_t* colidx, const float* vals,
                        const float* dinv,
                        const float* X, int64_t ldx, float* Y, int64_t ldy, float* tmp,
                        void* workspace, size_t workspace_bytes, void* stream);

/*
 * Tile-group SpMM: the adj.mm of utils.py:94 with its row scalings (utils.py:93, 95) for
 * pattern-only adjacencies (all stored off-diagonal values equal) whose rows share many
 * columns -- the Reddit shape.  Groups of 512 degree-ranked rows share every fetched row of
 * X through LDS; the sums run on the matrix cores (v_mfma_f32_32x32x16_f16, fp32
 * accumulation) over X split into two fp16 planes, hi + lo / 2048, with one power-of-two
 * scale per feature column (columns never mix in A . X).  What the planes do to the
 * arithmetic, as enforced on the device: an element within 2^29 of its column's largest
 * magnitude is reproduced to 2^-22 relative, a smaller one to 2^-50 of that largest
 * magnitude; an operand with a non-finite element, an element of at least twice its
 * column's bound, or more than max(16, elements / 2^14) nonzero elements so far below their
 * column's largest magnitude (2^-33) that fewer than 17 of their bits survive (the zero
 * crossings of a continuous distribution leave about one in 10^6), is refused:
 * a device flag goes up, the matrix-core kernel does nothing, and the plain CSR kernel
 * launched behind it computes the result from the fp32 operand (bit-identical to
 * sgnn_spmm_csr_f32), with no host synchronisation.  Accepted operands give results as far
 * from the exact product as sgnn_spmm_csr_f32's plus at most 2^-21 of |A||X|, plus 2^-49 of
 * the column's largest magnitude times the row's sum of weights, per hop, element by element
 * (on ordinary data the two kernels agree to 2^-20 of |A||X|), and are reproducible run to
 * run.
 *
 * Plan (once per matrix), two steps because its size depends on the matrix:
 *   sgnn_tilegroup_analyze  sorts and counts into `workspace`
 *                           (sgnn_tilegroup_build_workspace_bytes); when the stream has
 *                           drained, the first six int64 of `workspace` hold {merged
 *                           (group, column) pairs, stages, steps, work items, partial
 *                           slots, split groups}
 *   sgnn_tilegroup_fill     writes the plan (sgnn_tilegroup_plan_bytes of those counts)
 *                           from the same, untouched workspace.  The plan records its own
 *                           counts; an SpMM launch whose counts differ raises the flag
 *                           (bit 4) instead of reading the plan with the wrong layout.
 * Per operand, `state` = sgnn_tilegroup_state_bytes(F) of device memory: the fall-back flag
 * (second 32-bit word), the count of
 * tiny elements (fifth word), the work counter of the SpMM launch, and per feature the
 * measured largest magnitude and the bound its scale derives from:
 *   sgnn_tilegroup_amax_f32   clears `state` and measures max |scale_row[i] * X[i, f]| per feature
 *   sgnn_tilegroup_planes_f32 planes <- fp16 hi / lo of (scale_row[i] * X[i, :]) at the scales the
 *                           state's maxima give (scale_row nullable; utils.py:93 when it is
 *                           dinv).  `plane_rows` = rows per 128-feature slice of `planes`
 *                           (>= n: the n rows may be part of a larger operand, a rank's rows
 *                           of a partitioned matrix, whose ranks have agreed on the maxima
 *                           in between); `operand_rows` sizes the limit of tiny elements
 *   sgnn_tilegroup_split_f32  both, for an operand that is split whole.  From 65536 rows on the
 *                           scales are provisional: the maxima of every 32nd row, times 8,
 *                           stand in for the maxima of all rows (one pass over the operand
 *                           less); the split tests every element against its bound, and one
 *                           beyond it (an outlier the sample missed) makes the exact passes,
 *                           launched behind the provisional ones in any case, run instead of
 *                           returning at once (third word of the state: they ran; fourth: the
 *                           maxima are a sample's).  The bound of a feature then lies between
 *                           half and eight times its largest magnitude.
 *                           sgnn_tune_set("tg_no_sample", 1) switches the sampling off
 *   sgnn_spmm_tilegroup_f32   Y = diag(dinv_row) . (w_off * (pattern . planes + addend) + diag(corr) . x'),
 *                           epilogue as sgnn_spmm_csr_f32.  corr (nullable) = stored diagonal
 *                           value - w_off per row: the loaders' A + A^T carries 2.0 off the
 *                           diagonal (data_loader.py:75) and utils.py:83-86 puts 1.0 on it; x'
 *                           is row x_row_offset + i of the operand, taken from X (times
 *                           dinv_col) in fp32 where X is given, rebuilt from the planes
 *                           otherwise.  addend (nullable, [n_rows, ld_addend]): raw sums of
 *                           another column block of the same rows (a row partition
 *                           multiplies its own-columns block while remote rows travel).
 *                           rowptr / colidx / vals / dinv_col / X / ldx / csr_workspace
 *                           describe the same product for the fall-back (X and dinv_col as
 *                           given to the split call, csr_workspace of
 *                           sgnn_spmm_workspace_bytes); rowptr == NULL: no fall-back is
 *                           launched and the output is left untouched when the flag is set.
 *                           planes_out != NULL: instead of Y the rows are written as the
 *                           NEXT hop's operand -- fp16 planes of row i at node
 *                           out_row_offset + i of a buffer of planes_out_rows rows per slice,
 *                           `state_out` cleared and filled for it (its flag also inherits a
 *                           refusal of this hop).  The next operand's scale per feature is
 *                           derived from this operand's measured maximum (adapt = 1) or from
 *                           this operand's own bound again (adapt = 0: ranks of a partition
 *                           need no agreement): valid because dinv_row[i]^2 times row i's
 *                           weights sum to at most one for the normalised adjacency
 *                           (utils.py:88-95), so no element of D^-1 (A+I) x' exceeds the
 *                           largest of x'; an operand that breaks the bound is caught by the
 *                           element check and refused.
 *                           The launch is one workgroup per CU; the workgroups take (work
 *                           item, feature slice) pairs off a counter kept in `state` (zero
 *                           when the state is made, zeroed again by the last workgroup of
 *                           every launch): two SpMM calls that share a `state` must not
 *                           overlap (calls in one stream never do).
 *   sgnn_spmm_tilegroup_slices_f32  the same product for the operand's 128-feature slices
 *                           slice_begin .. slice_begin + slice_count - 1 only (slice_count <= 0:
 *                           to the last): a row partition whose operand arrives slice by slice
 *                           multiplies a slice while the next one travels.  All pointers and F
 *                           describe the whole operand as in sgnn_spmm_tilegroup_f32; ranges of
 *                           one operand go in ascending order on one stream (the range that
 *                           holds slice 0 clears state_out); a range that is not the whole
 *                           operand has no fall-back (rowptr must be NULL).  reserve_cus: the
 *                           launch leaves that many CUs without a workgroup, for kernels of
 *                           other streams (a collective) that must make progress beside it --
 *                           a workgroup of this kernel holds its CU until the launch's work
 *                           is done.
 *   sgnn_tilegroup_chain_f32  Y = A_hat^K . X (utils.py:91-96, model/SGC.py:11): one split of X,
 *                           K hops of which only the last writes fp32 -- the others write the
 *                           next hop's planes from their epilogue, so no intermediate is
 *                           stored as fp32, measured or split again -- then the plain CSR
 *                           chain on the fp32 X behind the last operand's flag (its
 *                           intermediates reuse the planes' memory).  planes_a / planes_b:
 *                           sgnn_tilegroup_planes_bytes(n, F) each (planes_b only for K > 1);
 *                           states: K blocks of sgnn_tilegroup_state_bytes(F).
 */
/* Diagnostics: cycle counters of tile-group runs made with tuning key "tg_mode" bit 16 set (kernel,
 * stage waits, next-stage issue, step loops, wavefronts, steps, stages; summed over wavefronts);
 * reading clears them.  No counterpart in the reference. */
int sgnn_tilegroup_debug_counters(unsigned long long* out8);
/* Diagnostics: eight words per unit of work (item, feature slice) -- wall clock (100 MHz) at the
 * start and the end of its stage loop, CU id, stages walked, wall clock when wavefront 0 had
 * written its rows, at entry, when the whole workgroup had fetched its next unit, one spare -- for
 * the first `count` (at most 16384) units of the last tile-group SpMM launch made with tuning
 * key tg_mode bit 16 set; out is host memory of count * 64 bytes.  No reference counterpart. */
int sgnn_tilegroup_debug_spans(unsigned long long* out, int64_t count);
size_t sgnn_tilegroup_build_workspace_bytes(int64_t n_rows, int64_t nnz);
size_t sgnn_tilegroup_plan_bytes(int64_t n_rows, int64_t stages, int64_t steps, int64_t items, int64_t long_groups);
int sgnn_tilegroup_analyze(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                           void* workspace, size_t workspace_bytes, void* stream);
int sgnn_tilegroup_fill(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t stages, int64_t steps, int64_t items,
                        int64_t long_groups, void* plan, size_t plan_bytes, void* workspace, size_t workspace_bytes,
                        void* stream);
size_t sgnn_tilegroup_planes_bytes(int64_t n_cols, int64_t F);
size_t sgnn_tilegroup_state_bytes(int64_t F);
int sgnn_tilegroup_amax_f32(int64_t n, int64_t F, const float* scale_row, const float* X, int64_t ldx, void* state,
                            void* stream);
int sgnn_tilegroup_planes_f32(int64_t n, int64_t F, const float* scale_row, const float* X, int64_t ldx, void* planes,
                              int64_t plane_rows, int64_t operand_rows, void* state, void* stream);
int sgnn_tilegroup_split_f32(int64_t n_cols, int64_t F, const float* scale_row, const float* X, int64_t ldx,
                             void* planes, size_t planes_bytes, void* state, void* stream);
size_t sgnn_spmm_tilegroup_workspace_bytes(int64_t partial_slots, int64_t F);
int sgnn_spmm_tilegroup_f32(int64_t n_rows, int64_t n_cols, int64_t F, const void* plan, size_t plan_bytes,
                            int64_t stages, int64_t steps, int64_t items, int64_t partial_slots, int64_t long_groups,
                            const float* dinv_row, float w_off, const float* corr, const void* planes,
                            size_t planes_bytes, void* state, float* Y, int64_t ldy, int epilogue, void* workspace,
                            size_t workspace_bytes, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                            const float* vals, const float* dinv_col, const float* X, int64_t ldx,
                            void* csr_workspace, size_t csr_workspace_bytes, int64_t x_row_offset, const float* addend,
                            int64_t ld_addend, void* planes_out, int64_t planes_out_rows, int64_t out_row_offset,
                            void* state_out, int adapt, void* stream);
int sgnn_spmm_tilegroup_slices_f32(int64_t n_rows, int64_t n_cols, int64_t F, const void* plan, size_t plan_bytes,
                                   int64_t stages, int64_t steps, int64_t items, int64_t partial_slots, int64_t long_groups,
                                   const float* dinv_row, float w_off, const float* corr, const void* planes,
                                   size_t planes_bytes, void* state, float* Y, int64_t ldy, int epilogue, void* workspace,
                                   size_t workspace_bytes, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                                   const float* vals, const float* dinv_col, const float* X, int64_t ldx,
                                   void* csr_workspace, size_t csr_workspace_bytes, int64_t x_row_offset, const float* addend,
                                   int64_t ld_addend, void* planes_out, int64_t planes_out_rows, int64_t out_row_offset,
                                   void* state_out, int adapt, int64_t slice_begin, int64_t slice_count, int reserve_cus,
                                   void* stream);
int sgnn_tilegroup_chain_f32(int K, int64_t n, int64_t F, const void* plan, size_t plan_bytes, int64_t stages,
                             int64_t steps, int64_t items, int64_t partial_slots, int64_t long_groups, const float* dinv,
                             float w_off, const float* corr, const float* X, int64_t ldx, float* Y, int64_t ldy,
                             void* planes_a, void* planes_b, size_t planes_bytes, void* states, size_t states_bytes,
                             void* workspace, size_t workspace_bytes, int64_t nnz, const int32_t* rowptr,
                             const int32_t* colidx, const float* vals, void* csr_workspace, size_t csr_workspace_bytes,
                             void* stream);

/*
 * Row-group form of a CSR's sparsity pattern, and the SpMM kernel that uses it.
 *
 * Rows are ranked by degree and cut into groups of R (8 or 16); a group stores
 * one merged column list with an R-bit mask per column.  The kernel fetches each
 * listed row of X once for the whole group, which on power-law graphs removes
 * 40-60 % of the row fetches that bound the plain CSR kernel.  Same arithmetic
 * as sgnn_spmm_csr_f32 with vals == NULL (utils.py:93-95); every row is summed
 * in ascending column order inside a segment of its group's list and the
 * segments are added in order, so results are reproducible run to run (the
 * order differs from the plain kernel's, results agree to fp32 rounding).
 *
 * sgnn_rowgroup_config   R to build the plan with for feature width F
 *                        (0: this F is not supported, use sgnn_spmm_csr_f32)
 * sgnn_rowgroup_config_for  the same choice made with the graph's size known: a graph
 *                        with n_rows / 16 + nnz / 64 < 8192 takes R = 8 at every supported
 *                        width (too few work items at R = 16 to occupy the chip); either
 *                        R is valid for every supported F
 * sgnn_rowgroup_build    fills `plan` (device, sgnn_rowgroup_plan_bytes) from the
 *                        CSR; when the stream has drained, the first four int64
 *                        of `plan` hold {merged entries, launch slots, partial
 *                        slots, split groups} — the caller reads them once and
 *                        passes the last three to every SpMM call as items, slots,
 *                        n_long.  Launch slots are the work items laid out for the
 *                        grid: on graphs whose rows share columns the items are dealt
 *                        to the 8 XCDs in contiguous, work-balanced runs padded to
 *                        equal length, so the count can exceed the item count.
 * sgnn_spmm_rowgroup_f32 workspace >= sgnn_spmm_rowgroup_workspace_bytes(slots, R, F).
 *                        X must already carry the column scaling of utils.py:93
 *                        (sgnn_row_scale_f32; dinv_col NULL) and its rows must
 *                        start on 16-byte boundaries (X 16-byte aligned, ldx % 4 == 0):
 *                        the kernel reads them with 16-byte loads; SGNN_ERR_UNSUPPORTED
 *                        otherwise, and for F < 33.  With R = 8, no group weights and X
 *                        below 4 GiB, dinv_col may be given instead: the kernel then scales
 *                        every fetched row itself (one launch and one pass over X less,
 *                        which is what a graph that fits the caches is bound by; larger
 *                        graphs are faster with X scaled once up front).  Widths above 640 run as equal
 *                        feature slices of at most 640 columns, one launch each.
 *                        The plan stores the sparsity pattern only.  An adjacency whose
 *                        off-diagonal entries all carry one weight w_off (Reddit's A + A^T
 *                        has 2.0, data_loader.py:75) is handled in the epilogue:
 *                        row i = w_off * (pattern sum) + corr[i] * x'_i with
 *                        corr[i] = (diagonal value of row i) - w_off where the row stores
 *                        a diagonal entry, else 0; x'_i is row (i + x_row_offset) of X.
 *                        Unit weights: w_off = 1, corr = NULL.
 * sgnn_rowgroup_weights  for adjacencies whose stored values differ (edge weights passed
 *                        to GCNConv, weighted scipy matrices): fills `group_weights`
 *                        (device, sgnn_rowgroup_weights_bytes(merged, R)) with the value of
 *                        every (merged column, row of the group) pair, 0 where the row has
 *                        no such entry.  Passed to sgnn_spmm_rowgroup_f32 it makes the
 *                        kernel compute the weighted sum entry by entry (fmaf in ascending
 *                        column order per row); w_off and corr are then ignored.  NULL =
 *                        pattern sum + uniform-weight epilogue as described above.
 * No reference counterpart for the plan itself: the reference keeps only the
 * COO tensor (utils.py:54-62).
 */
int sgnn_rowgroup_config(int64_t F);
int sgnn_rowgroup_config_for(int64_t n_rows, int64_t nnz, int64_t F);
size_t sgnn_rowgroup_plan_bytes(int64_t n_rows, int64_t nnz, int R);
size_t sgnn_rowgroup_workspace_bytes(int64_t n_rows, int64_t nnz, int R);
int sgnn_rowgroup_build(int64_t n_rows, int64_t n_cols, int64_t nnz,
                        const int32_t* rowptr, const int32_t* colidx, int R,
                        void* plan, size_t plan_bytes,
                        void* workspace, size_t workspace_bytes, void* stream);
size_t sgnn_spmm_rowgroup_workspace_bytes(int64_t slots, int R, int64_t F);
size_t sgnn_rowgroup_weights_bytes(int64_t merged, int R);
int sgnn_rowgroup_weights(int64_t n_rows, int64_t nnz,
                          const int32_t* rowptr, const int32_t* colidx, const float* vals, int R,
                          const void* plan, size_t plan_bytes, int64_t merged,
                          float* group_weights, size_t group_weights_bytes, void* stream);
int sgnn_spmm_rowgroup_f32(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t F, int R,
                           const void* plan, size_t plan_bytes, const float* group_weights,
                           int64_t items, int64_t slots, int64_t n_long,
                           const float* dinv_row, const float* dinv_col,
                           float w_off, const float* corr, int64_t x_row_offset,
                           const float* X, int64_t ldx, float* Y, int64_t ldy,
                           int epilogue, void* workspace, size_t workspace_bytes, void* stream);

/* out[i, 0:F] = scale[i] * in[i, 0:F]  — utils.py:93 as a stand-alone pass (used
 * to pre-scale the row block a rank contributes to the all-gather). */
int sgnn_row_scale_f32(int64_t n, int64_t F, const float* scale,
                       const float* in, int64_t ldin, float* out, int64_t ldout, void* stream);

/*
 * out[i, :] = ((a[i, :] + b[i, :]) * scale[i]) (* scale[i] again when twice != 0).
 * The row-partitioned path multiplies a rank's own-columns block while the other ranks' rows
 * of X travel and its remote-columns block when they have landed; this adds the two blocks'
 * raw sums in that fixed order and applies the row scaling of utils.py:95 (twice: followed by
 * utils.py:93 of the next hop).  No counterpart in the reference, whose ranks are replicas
 * (main.py:39-41).
 */
int sgnn_add_scale_f32(int64_t n, int64_t F, const float* a, int64_t lda, const float* b, int64_t ldb,
                       const float* scale, int twice, float* out, int64_t ldout, void* stream);

/* Measurement aid, no counterpart in the reference: dst[0, count) <- src, 16 bytes per lane, as a
 * yardstick of what the card streams (read + write) beside the vendor's HBM peak.  count must be
 * a multiple of 4, both buffers 16-byte aligned and disjoint. */
int sgnn_stream_copy_f32(const float* src, float* dst, int64_t count, void* stream);

/* sampling flags */
#define SGNN_SAMPLE_SKIP_SELF  1   /* write 0 where idx[i] == idx[j]: sample the matrix without its diagonal */

/*
 * out[i, j] = A[idx[i], idx[j]] for i, j < b, dense row-major fp32 with row
 * stride ldo — the b x b block the auto-encoder layer reconstructs per Adam
 * iteration.  Replaces model/SGNN.py:167-170 (adjacency_npy[idx, :][:, idx],
 * .todense(), torch.Tensor(...), .to(device)) and, called on the CSR of the
 * overlook matrix, model/SGNN.py:172-175.
 *   rowptr/colidx/vals : CSR with column indices ascending within each row
 *                        (what sgnn_build_norm_csr writes); vals may be null = all 1
 *   idx                : int64[b] node ids; a repeated id repeats the row/column as
 *                        fancy indexing does; an id outside [0, n_rows) gives a zero row
 *   flags              : SGNN_SAMPLE_SKIP_SELF to read a CSR that stores A + I as
 *                        the diagonal-free A the reference keeps (model/SGNN.py:497-501)
 */
int sgnn_sample_block_f32(int64_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                          const float* vals, int64_t b, const int64_t* idx, int flags,
                          float* out, int64_t ldo, void* stream);

/*
 * out[0:b] = the first b entries of a pseudo-random permutation of [0, n): distinct
 * int64 node ids, as torch.randperm(n)[:b] gives the layers per optimiser iteration
 * (model/SGNN.py:115-117), without sorting n keys.  `state` is uint64[3] in device
 * memory, {seed, counter, 0}; the third word is internal, zero on entry and left zero.
 * The call uses and then increments the counter on the stream, so a captured HIP graph
 * that contains it draws a new batch on every replay.  The permutation is a 4-round Feistel network with cycle walking: a
 * bijection for every (seed, counter), not a cryptographic one.  With `map` (int64[n] in
 * device memory, may be NULL) the entries are map[.] of the drawn positions: a batch of the
 * training nodes, as node ids (model/SGNN.py:412-421 samples rows of processed_X[mask]).
 */
int sgnn_batch_indices(int64_t n, int64_t b, uint64_t* state, const int64_t* map, int64_t* out, void* stream);

/*
 * Reconstruction loss of the graph auto-encoder layer on one batch (model/SGNN.py:138-164,
 * 209-226, 682-702), in two launches around the caller's GEMM S = Z Z^T.
 *
 * sgnn_gae_targets_f32: for the batch `idx` (int64[b]) writes, dense b x b row-major,
 *   target = A[idx][:, idx] with 0 where idx[i] == idx[j] (model/SGNN.py:167-170 on the
 *            diagonal-free adjacency; the CSR passed may store A + I),
 *   retain = 1 - max(random symmetric ignore mask without diagonal, overlook block)
 *            (model/SGNN.py:157-159, 172-175, 209-226): a pair is ignored with
 *            probability 2*mask_rate - mask_rate^2 (either of the reference's two draws),
 *            drawn from `state` = uint64[3] {seed, counter, 0} (read only here: the counter is
 *            advanced by the sgnn_gae_loss_grad_f32 call that completes the pair); the overlook matrix is its unit diagonal
 *            plus, optionally, the pattern of a second CSR (overlook_rowptr/colidx, sorted
 *            columns; both NULL = diagonal only),
 *   row_pos[i] = sum_j target*retain, row_ret[i] = sum_j retain.
 * sgnn_gae_loss_grad_f32: from S (row stride lds) and the four arrays above writes
 *   G = dloss/dS (b x b) and row_loss[i] (the loss is their sum, added in row order) for
 *   loss = mean(-ratio * target * log(max(P, 1e-6)) * retain) [if num_pos > 0]
 *        + mean(-(1 - target) * log(max(1 - P, 1e-6)) * retain),
 *   P = sigmoid(clamp(S, -40, 40)), num_pos = sum(row_pos), ratio = sum(row_ret)/num_pos - 1.
 *   `state` (may be NULL): the state passed to sgnn_gae_targets_f32; its counter is advanced
 *   here, so that the next pair (or the next replay of a captured pair) draws another mask.
 * The caller completes dZ = (G + G^T) Z.
 */
int sgnn_gae_targets_f32(int64_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                         const float* vals, const int32_t* overlook_rowptr, const int32_t* overlook_colidx,
                         int64_t b, const int64_t* idx, float mask_rate, uint64_t* state,
                         float* target, float* retain, float* row_pos, float* row_ret, void* stream);
int sgnn_gae_loss_grad_f32(int64_t b, const float* S, int64_t lds, const float* target, const float* retain,
                           const float* row_pos, const float* row_ret, float* G, float* row_loss,
                           uint64_t* state, void* stream);

/*
 * The dense tail of one optimiser iteration of a decoupled layer, one launch each.
 *
 * sgnn_adam_step_f32: one Adam update (torch.optim.Adam as model/SGNN.py:417-432 uses it:
 *   no amsgrad, weight decay added to the gradient) of up to 16 contiguous fp32 tensors:
 *     g = grad + weight_decay[k] * p + l1_decay[k] * sign(p);
 *     m += (1 - beta1) * (g - m);  v = beta2 * v + (1 - beta2) * g * g;
 *     p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps),   t = steps so far + 1.
 *   params / grads / exp_avg / exp_avg_sq / sizes / weight_decay are HOST arrays of `count`
 *   entries (device pointers, element counts, per-tensor penalties; a NULL penalty array = 0).
 *   `state` is uint64[SGNN_ADAM_STATE_WORDS] in device memory, all zero before the first
 *   step: word 0 = steps taken, the rest internal (left zero); the launch advances it, so a
 *   captured HIP graph keeps counting on replay.
 *   With weight_decay = 2 * lam on W, b and U this also covers the ridge penalty of
 *   model/SGNN.py:89-92, with l1_decay = lam the LASSO penalty of model/SGNN.py:93-96
 *   (their gradients; the penalty's value is not added to any reported loss).
 * sgnn_softmax_xent_f32: loss[0] = mean_i( logsumexp(z_i) - z_i[y_i] ) and
 *   dlogits = (softmax(z) - onehot(y)) / B for logits z (B x C, row stride ldz) —
 *   torch's CrossEntropyLoss with its backward (model/SGNN.py:397-399).
 *   y_i = labels[idx[i]] when idx (int64[B]) is given, else labels[i]; labels is
 *   int64[n_labels]; a row whose label is out of range contributes no loss term.
 * sgnn_sq_err_f32: loss[0] = scale * sum_ij (a_ij - t_ij)^2 and grad = 2 * scale * (a - t)
 *   for a (B x D, row stride lda) — the embedding-target term eta * ||a - t||_F^2 / B of
 *   model/SGNN.py:80-86, 393.  Row i of t is target[idx[i]] when idx is given, else
 *   target[i]; target has target_rows rows of stride ldt.
 * Both losses take `workspace` (sgnn_loss_workspace_bytes(B) bytes of device memory, contents
 * irrelevant): the per-row terms go through it and are added in a fixed order by a second
 * small launch, so the loss value does not depend on scheduling.
 */
#define SGNN_ADAM_STATE_WORDS 1026
int sgnn_adam_step_f32(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* sizes, const float* weight_decay,
                       const float* l1_decay, float lr, float beta1, float beta2, float eps, uint64_t* state,
                       void* stream);
/* out[0] = factor * (x[0] + ... + x[n-1]), one wavefront, fixed order (lane l adds l, l + 64, ...,
 * then the lanes fold in halves): the per-row loss terms of a batch into the scalar loss. */
int sgnn_sum_f32(int64_t n, const float* x, float factor, float* out, void* stream);
size_t sgnn_loss_workspace_bytes(int64_t B);
int sgnn_softmax_xent_f32(int64_t B, int64_t C, const float* logits, int64_t ldz, const int64_t* labels,
                          int64_t n_labels, const int64_t* idx, float* loss, float* dlogits, int64_t ldd,
                          void* workspace, size_t workspace_bytes, void* stream);
int sgnn_sq_err_f32(int64_t B, int64_t D, const float* a, int64_t lda, const float* target, int64_t ldt,
                    int64_t target_rows, const int64_t* idx, float scale, float* loss, float* grad,
                    int64_t ldg, void* workspace, size_t workspace_bytes, void* stream);

/*
 * A layer's projection of one batch with the row gather and the inner activation folded
 * into the operand load (model/SGNN.py:73-78 applied to the rows model/SGNN.py:106-121
 * picked):
 *   sgnn_batch_project_f32            E  (b x N) = act(X[idx]) . W        W  is F x N, row stride ldw
 *   sgnn_batch_project_grad_f32       dW (F x N) = act(X[idx])^T . dE     dE is b x N, row stride ldde
 *   sgnn_batch_project_input_grad_f32 dA (b x F) = (dE . W^T) * act'(pre) pre is b x F or NULL (no factor)
 * act(x) = x > 0 ? x : slope * x (slope 1 = linear, 0 = relu, else leaky_relu), act' its
 * derivative (1 or slope); X has n_rows rows of F floats with row stride ldx; idx is int64[b],
 * an id outside [0, n_rows) reads as a zero row, idx NULL = rows 0..b-1 of X (a dense operand).
 * With U the identity the first two are the whole layer; while U is trained they serve
 * X[idx] . U / dU (slope 1) and, on the dense XU, act(XU) . W / dW, and the third gives dXU.
 * fp32 throughout, fused multiply-adds in a fixed order: the reduction dimension is cut into
 * chunks handled by different workgroups whose partial tiles go through `workspace`
 * (sgnn_batch_project_workspace_bytes bytes, contents irrelevant; `which` = 0, 1, 2 for the
 * three calls in the order above) and are added in chunk order by a second launch.
 */
size_t sgnn_batch_project_workspace_bytes(int64_t b, int64_t F, int64_t N, int which);
int sgnn_batch_project_f32(int64_t n_rows, int64_t F, const float* X, int64_t ldx, int64_t b, const int64_t* idx,
                           float slope, const float* W, int64_t ldw, int64_t N, float* E, int64_t lde,
                           void* workspace, size_t workspace_bytes, void* stream);
int sgnn_batch_project_grad_f32(int64_t n_rows, int64_t F, const float* X, int64_t ldx, int64_t b,
                                const int64_t* idx, float slope, const float* dE, int64_t ldde, int64_t N,
                                float* dW, int64_t lddw, void* workspace, size_t workspace_bytes, void* stream);
int sgnn_batch_project_input_grad_f32(int64_t b, int64_t F, int64_t N, const float* dE, int64_t ldde,
                                      const float* W, int64_t ldw, const float* pre, int64_t ldpre, float slope,
                                      float* dA, int64_t ldda, void* workspace, size_t workspace_bytes,
                                      void* stream);

/*
 * The label head on one batch in two launches (model/SGNN.py:397-399 with its backward):
 *   logits = E . Wt,  loss[0] = mean_i( logsumexp(logits_i) - logits_i[y_i] ),
 *   dE = dz . Wt^T,  dWt = E^T . dz,   dz = (softmax(logits) - onehot(y)) / B.
 * E is B x D (row stride lde), Wt is D x C (row stride ldw), D <= 256, C <= 64 and
 * 4 (D C + 8 D + 8 C) <= 64 KB of LDS (SGNN_ERR_UNSUPPORTED beyond); y_i = labels[idx[i]] when idx is given, else labels[i].
 * `workspace`: sgnn_head_xent_workspace_bytes(B, D, C) bytes, contents irrelevant.  Sums are
 * formed in a fixed order.
 */
size_t sgnn_head_xent_workspace_bytes(int64_t B, int64_t D, int64_t C);
int sgnn_head_xent_f32(int64_t B, int64_t D, int64_t C, const float* E, int64_t lde, const float* Wt, int64_t ldw,
                       const int64_t* labels, int64_t n_labels, const int64_t* idx, float* loss,
                       float* dE, int64_t ldde, float* dWt, int64_t lddw,
                       void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGNN_HIP_H */
