/* tcmi.h -- C ABI of libtcmi.so, the MI355X (gfx950) executor for the tensorcircuit-ng
 * state-vector / expectation hot path.
 *
 * The reference is pure Python: its "FFI" for this path is the backend plug-in
 * (tensorcircuit/backends/backend_factory.py:26-59) whose tensordot/transpose/reshape methods are
 * called by the contractor loop (tensorcircuit/cons.py:937-960).  A reference maintainer binds this
 * library with ctypes from a new backend class; every entry point below names
 * the reference interface it replaces.
 *
 * Multi-GPU: one process per GPU, every entry point acts on the calling rank's device.  The path's exchange steps are one
 * packed all-reduce of [value || gradients] per step (plus, in a sliced contraction, one all-gather of the
 * slice-invariant roots and one all-reduce of their cotangents).  The package's Python host issues them through the host
 * framework's process group (torch.distributed, backend nccl = RCCL); a host without one uses tcmi_comm_* /
 * tcmi_allreduce_sum below (RCCL opened at first use, not linked).
 *
 * Conventions: all pointers are DEVICE pointers unless the name ends in _host; the caller owns every
 * buffer; `stream` is a hipStream_t passed as void* (NULL = default stream); every function returns
 * 0 (TCMI_OK) or a negative error code and never throws; tcmi_last_error() gives the message of the
 * calling thread's last failure.  dtype: TCMI_C64 = complex64 (interleaved float re,im),
 * TCMI_C128 = complex128.  State index convention: qubit 0 is the most significant bit of the flat
 * index (reference tests/test_circuit.py:47-53); a batched state is [batch][2^n] with
 * `state_stride` amplitudes between batch elements.
 */
#ifndef TCMI_H
#define TCMI_H

#ifdef __cplusplus
extern "C" {
#endif

#define TCMI_VERSION 1
#define TCMI_OK 0
#define TCMI_ERR_ARG (-1)
#define TCMI_ERR_HIP (-2)
#define TCMI_C64 0
#define TCMI_C128 1
#define TCMI_F32 2   /* real dtypes: tcmi_allreduce_sum only */
#define TCMI_F64 3

/* library / runtime probes */
int tcmi_version(void);
const char* tcmi_last_error(void);
int tcmi_device_count(void);

/* |0...0> for every batch element.
 * Replaces: Circuit.__init__ -> BaseCircuit.all_zero_nodes (tensorcircuit/circuit.py:44-131,
 * tensorcircuit/basecircuit.py:51-66). */
int tcmi_init_zero_state(void* state, long long state_stride, int batch, int n, int dtype, void* stream);

/* Host-side planner helper (no GPU work): exact dynamic programme over the subsets of the k <= 16 frontier tensors of a
 * contraction subtree.  `masks` = k index sets as W 64-bit words each; `lw` = log2 of the index dimensions by bit
 * (NULL: all dimensions 2); intermediates with more than `cap` (log2 elements) are not allowed; cost of a step =
 * 2^|union| + alpha (elements read + written).  Writes split[S] (the larger half A of the best bipartition of subset S,
 * 0 if none) for all 2^k subsets and the cost of the full set.  Bit-identical to the Python loop of
 * tcmi/tn.py::reconfigure_path it accelerates.
 * Replaces: cotengra's subtree reconfiguration behind tensorcircuit/cons.py:1168-1190 (`optimizer_reconf`) and
 * experimental.py `slicing_reconf_opts`. */
int tcmi_subtree_dp(int k, int W, const unsigned long long* masks, const double* lw, double cap, double alpha,
                    int* split, double* best_full);

/* Host code, no device work: the WHOLE subtree-reconfiguration loop of tcmi/tn.py::reconfigure_path around
 * tcmi_subtree_dp -- for every internal node (most expensive first) the subtree below it is cut off at `subtree_size`
 * intermediates (the most expensive internal node expanded first) and re-contracted in the order the dynamic programme
 * finds cheapest, passes repeated until nothing improves (at most `max_passes` passes, `max_evals` subtree
 * optimisations).  masks: the `ntensors` leaf index sets (W words each); ssa_pairs: the path in SSA numbering (step s
 * creates node ntensors + s); lw / cap / alpha as tcmi_subtree_dp.  Result: the internal nodes of the new tree,
 * nodes_out[i] with children kids_out[2 i], kids_out[2 i + 1] (ids >= 2 ntensors - 1 are nodes created by the rebuilds;
 * the root keeps its id), *nnodes_out of them (at most max_nodes_out).  Same tree as the Python loop, node for node.
 * Replaces: cotengra's subtree reconfiguration behind tensorcircuit/cons.py:1168-1190 (`optimizer_reconf`) and
 * experimental.py `slicing_reconf_opts`. */
int tcmi_reconfigure_path(int ntensors, int W, const unsigned long long* masks, const int* ssa_pairs, const double* lw,
                          double cap, double alpha, int subtree_size, int max_passes, int max_evals, int* nodes_out,
                          int* kids_out, int max_nodes_out, int* nnodes_out);

/* Host code, no device work: greedy slicing of a FIXED contraction tree of dimension-2 indices
 * (tcmi/tn.py::ContractionTree._slice_fixed): um / km = per step the union of the operand indices / the indices the step
 * keeps (W words each), outm = the output indices, labels[bit] = the caller's index label (tie-break); indices are removed
 * one at a time -- always the candidate (of the `max_candidates` highest-scoring indices of the oversize intermediates) that
 * leaves the smallest (total oversize, flops) -- until every intermediate has at most `target_bits` indices.  Writes the
 * removed bits in order, their number (-1: not reachable within `max_slices` slices) and sum_steps 2^|union| of the sliced tree.
 * Replaces: cotengra's slicing (reference experimental.py:936-953 `slicing_opts`). */
int tcmi_slice_fixed(int nsteps, int W, const unsigned long long* um, const unsigned long long* km,
                     const unsigned long long* outm, const long long* labels, int target_bits, long long max_slices,
                     int max_candidates, int* sliced_bits_out, int max_sliced_out, int* nsliced_out, double* flops_out);

/* Host code, no device work: the random-greedy pairwise path of a circuit network (every index has dimension 2 and at most
 * two ends; one end = an output index) -- opt_einsum's RandomGreedy, which the reference reaches through cotengra's
 * "greedy" method (tensorcircuit/cons.py:1168-1190).  masks: ntensors x W 64-bit words (bit e = index e), outmask: the
 * output indices; alpha: the size-difference weight; temperature > 0: Boltzmann choice among the nbranch best candidates,
 * one pre-drawn uniform number per step (`uniforms`).  ssa[2 * step + {0, 1}] = the contracted pair in SSA numbering.
 * Returns the number of steps (ntensors - 1 for a connected network) or a negative error. */
int tcmi_greedy_path(int ntensors, int W, const unsigned long long* masks, const unsigned long long* outmask,
                     double alpha, double temperature, int nbranch, const double* uniforms, int nuniforms, int* ssa);


/* Gate-table builder: per batch element, turn the flat real parameter vector into the dense gate
 * matrices  M = C0 + cos(k*theta+o) C1 + sin(k*theta+o) C2  and the diagonal phase coefficients the
 * pass programs reference.  `ginfo` = int32[nrec][8] records {kind, out_slot, param_index, dim,
 * cpool_offset, 0,0,0}; `cpool` = float64 constants; params are float (C64) / double (C128).
 * Replaces: the gate factories rx/ry/rz/exp1/rzz/phase/... evaluated per call
 * (tensorcircuit/gates.py:584-603, 692-743, 920-978). */
int tcmi_build_tables(const int* ginfo, int nrec, const double* cpool, const void* params,
                      long long params_stride, void* ptab, long long ptab_stride, int batch,
                      int dtype, void* stream);

/* Weights of the cut contraction (tcmi/cut.py): a shallow circuit is cut between two qubit groups, every gate across
 * the cut is written as a sum of products (left factor x right factor x coefficient), and the wavefunction is
 * sum_k w[k] L_k (x) R_k with  w[b][k] = prod_j coef_j(digit_j(k), theta_b).  `tab_i` = int32[nb*rmax] entries
 * kind | (param_index << 2) (kind 0 constant, 1 cos, 2 sin); `tab_f` = float64[nb*rmax][4] {scale, offset, const_re,
 * const_im}; `digits` = uint8[K][nb]; params float (C64) / double (C128) [batch][params_stride]; `w` = complex
 * [batch][K] of the state's precision.  One launch for what is otherwise a dozen elementwise launches per step.
 * Replaces: the coefficient arithmetic of the cross-cut gates inside the contraction that Circuit.wavefunction
 * executes (tensorcircuit/circuit.py:701-721, gate factories tensorcircuit/gates.py:692-743). */
int tcmi_cut_weights(const void* params, long long params_stride, int batch, const int* tab_i, const double* tab_f,
                     const unsigned char* digits, int K, int nb, int rmax, void* w, int dtype, void* stream);

/* The 4 x 4 epilogue of a cut contraction whose LAST crossing gate is applied after the join (tcmi/cut.py, `Epilogue`):
 * X[b] = prod_g (c0_g + cos(a_g) c1_g + sin(a_g) c2_g), later factors on the left, a_g = scale_g * theta_b[param_g] +
 * offset_g, every factor a 4 x 4 over (last qubit of the left half, first qubit of the right half).  `tab_i` =
 * int32[nfac] parameter indices (-1: constant factor); `tab_f` = float64[nfac][98] {scale, offset, c0[16], c1[16],
 * c2[16]} (complex entries as re, im; row-major [out][in]); params float (TCMI_C64) / double (TCMI_C128)
 * [batch][params_stride]; `x` = complex64 [batch][16], the operand of tcmi_cgemm_split_epi.  Replaces: the gate
 * matrices of those gates inside the contraction Circuit.wavefunction executes (tensorcircuit/circuit.py:701-721,
 * gate factories tensorcircuit/gates.py:692-743). */
int tcmi_cut_epilogue(const void* params, long long params_stride, int batch, const int* tab_i, const double* tab_f,
                      int nfac, void* x, int dtype, void* stream);

/* One pass of a compiled plan over the (batched) state, in place: every workgroup loads a tile of
 * 2^(R+LT) amplitudes, applies the pass program `desc` (int32 words, layout in
 * tensorcircuit-ng_amd/csrc/tcmi_vm.h) and stores the tile back.  `ctab` = shared constant table,
 * `ptab` = per-batch table written by tcmi_build_tables (real values of the state's precision).
 * Replaces: the pairwise loop  tn.contract_between -> backend.tensordot  plus the final
 * Node.reorder_edges -> backend.transpose  (tensorcircuit/cons.py:937-960), i.e. what
 * Circuit.wavefunction executes (tensorcircuit/circuit.py:701-721).
 * Measurement passes (programs containing EXPECT ops, built by tcmi.plan.compile_measure_plan)
 * accumulate the Pauli-string values <psi|P_t|psi> into eout[batch][2*t] (re, im; float64; the
 * caller zeroes it) instead of storing the tile.  They replace the 2n-1 separate
 * contractor([psi, psi*, op...]) reductions of Circuit.expectation
 * (tensorcircuit/circuit.py:899-902, basecircuit.py:393-447); eout may be NULL otherwise.
 * eout is replicated `ecopies` times (`ecopy_stride` doubles apart; workgroup w adds into copy
 * w % ecopies) so that same-address atomics spread over the L2 channels; the caller sums the copies. */
int tcmi_run_pass(void* state, long long state_stride, int batch, int n, int R, int LT,
                  const int* desc, const void* ctab, const void* ptab, long long ptab_stride,
                  double* eout, long long eout_stride, int ecopies, long long ecopy_stride, int dtype,
                  void* stream);

/* ---- reverse mode (value_and_grad) -----------------------------------------------------------------
 * Replaces the framework AD the reference relies on -- backend.value_and_grad / vvag
 * (tensorcircuit/backends/abstract_backend.py:2262-2293, 2541-2591; jax_backend.py:854-952;
 * pytorch_backend.py:775-878) -- for functions of the form  params -> Circuit -> expectation. */

/* (sum_t w[b][t] P_t) |in>  ->  out, both [batch][2^n]; terms = int32[nterms][3] {xmask, zmask, nY}
 * sorted by xmask (bit p = physical bit p of the flat index), w = float64[batch][nterms].
 * With w = 2 Re(dL/d<P_t>) this is the cotangent of psi for L = f(<psi|P_t|psi>): the backward of
 * Circuit.expectation (tensorcircuit/circuit.py:899-902). */
int tcmi_apply_pauli_sum(const void* in, void* out, long long state_stride, int batch, int n,
                         const int* terms, int nterms, const double* weights, long long weights_stride,
                         int dtype, void* stream);

/* The same product as tile passes (csrc/tcmi_hsum.hip): ONE pass over tiles spanned by `tilepos` (T =
 * tcmi_pauli_sum_tile_bits(dtype) ascending physical index bits, tilepos[0] must be 0) that applies every given term --
 * all X masks must lie inside the tile bits.  terms = int32[nterms][4] {X mask in tile-index space (bit j = tile bit j),
 * Z/Y sign mask over physical bits, number of Y | emask << 8, parity(popcount(X & Z)) with X the physical mask}; emask =
 * the sign mask restricted to the element bits of a thread: bit 0 = tile bit 0, bit 1 + b = tile bit 9 + b.  The first
 * ndiag rows are the Z-only strings sorted by emask, the others are sorted by X mask; w = float64[batch][nterms] in row
 * order.  accumulate != 0: out += ..., else out = ....  eout (may be NULL): float64[batch][ecopies], zeroed by the
 * caller; receives Re <in | this pass's part of out> spread over the copies -- summed over the passes of a Hermitian sum
 * that is sum_t w_t <P_t>, the energy whose cotangent `out` is (reference circuit.py:833-913 for the value, :899-902
 * under value_and_grad for the cotangent; n <= 32). */
int tcmi_pauli_sum_tile_bits(int dtype);
int tcmi_apply_pauli_sum_tiled(const void* in, void* out, long long state_stride, int batch, int n, const int* tilepos,
                               const int* terms, int nterms, int ndiag, const double* weights, long long weights_stride,
                               int accumulate, double* eout, long long eout_stride, int ecopies, int dtype,
                               void* stream);

/* Adjoint-sweep tables: U^dagger and K = (dU/dtheta) U^dagger per gate (records as in
 * tcmi_build_tables with kinds TCMI_BK_UDAG / TCMI_BK_KMAT / TCMI_BK_COEF). */
int tcmi_build_adjoint_tables(const int* ginfo, int nrec, const double* cpool, const void* params,
                              long long params_stride, void* ptab, long long ptab_stride, int batch,
                              int dtype, void* stream);

/* One pass of the reversed plan over (psi, lambda), both updated in place:
 * gout[batch][slot] += Re <lambda| K_g |psi> for every parametrised gate g of the pass (float64,
 * caller zeroes it), then psi <- U_g^dagger psi, lambda <- U_g^dagger lambda.  After the last pass
 * psi is the circuit's input state and gout holds dL/dtheta per gate slot (replicated `gcopies`
 * times like eout above; the caller sums the copies).
 * opset: the op set the descriptors were compiled for -- TCMI_OPSET_GENERIC (every backward op of tcmi_vm.h) or
 * TCMI_OPSET_PACKED (complex64 plans of one-qubit gate ops and table-form diagonal flushes only, run by the
 * packed-f32 kernel; tiles (R, LT) = (4, 8), (4, 9), (5, 8)). */
#define TCMI_OPSET_GENERIC 0
#define TCMI_OPSET_PACKED 1
int tcmi_run_adjoint_pass(void* psi, void* lam, long long state_stride, int batch, int n, int R, int LT,
                          const int* desc, const void* ctab, const void* ptab, long long ptab_stride,
                          double* gout, long long gout_stride, int gcopies, long long gcopy_stride,
                          int dtype, int opset, void* stream);

/* ---- pairwise contraction engine (closed networks, sliced contraction) -------------------------------
 * tn.contract_between(a, b) -> backend.tensordot(a, b, axes) (tensorcircuit/cons.py:396,413,450,948 via
 * tensornetwork) is lowered by tcmi/tn.py to  permute(A) -> [M x K],  permute(B) -> [K x N],
 * C = A.B  (axes free_A + free_B, no output permute). */

/* Axis permutation of a tensor whose axes all have dimension 2 (rank <= 34), out-of-place:
 * out[o] = in[src(o)], src(o) = OR_b ((o >> b) & 1) << srcbit[b].  srcbit = int32[rank] on the device.
 * Replaces Node.reorder_edges -> backend.transpose (tensorcircuit/cons.py:425,462,761,924,960). */
int tcmi_permute_bits(const void* in, void* out, int rank, const int* srcbit, int batch,
                      long long batch_stride, int dtype, void* stream);

/* Contraction of a big [2]^rank tensor with a small one over `nk` (1..5, or 6..8 when n <= 16) of its axes at ARBITRARY bit positions
 * (pos = host array, ascending bit indices, bit j of the small operand's row index <-> pos[j]; small = [2^nk][n]
 * row-major, 2^nk * n <= 4096, n a power of two):  out[f][n] (big_first) or out[n][f], f = the free bits of the big
 * tensor in their stored order.  The big-tensor x gate-like-tensor steps of a contraction tree
 * (tensorcircuit/cons.py:948 contract_between on a boundary tensor) read the big operand once and write the
 * result once instead of permute + skinny GEMM. */
int tcmi_contract_scattered(const void* big, int rank, const int* pos, int nk, const void* small_operand, long long n,
                            void* out, int big_first, int dtype, void* stream);

/* tensordot(a, b, axes=(axes_a, axes_b)) of two contiguous [2]^rank complex64 tensors straight from their stored
 * layouts -- no operand is permuted: c has the axes (free axes of a in order, free axes of b in order), the
 * contract_between convention of tensornetwork (reference tensorcircuit/cons.py:948 -> backend.tensordot).  MFMA
 * kernel with bit-deposit operand addressing; splits K over the grid when the output has few tiles.  rank <= 31. */
int tcmi_tensordot_bits(const void* a, int rank_a, const void* b, int rank_b, const int* axes_a, const int* axes_b,
                        int nk, void* c, int dtype, void* stream);

/* The same with the elementwise and index work of a reverse-mode step fused in: flags bit 0 / 1 = use conj(a) /
 * conj(b); out_axes (NULL: natural order) = the result is stored as permute(natural, out_axes), natural = (free axes of
 * a, free axes of b).  The two VJPs of c = tensordot(a, b) -- gA = tensordot(g, conj b) moved to a's axis order, gB
 * likewise (JAX's transpose rule of dot_general, which the reference gets through value_and_grad of contract_core,
 * tensorcircuit/experimental.py:1182-1211) -- are then ONE launch each instead of three.  out_axes works for every
 * shape (the tile kernels store through the bit permutation); the conjugation flags only for the shapes
 * tcmi_tensordot_bits_small_ok() accepts (both ranks <= 12, result rank <= 12, nk <= 8: the thousands of gate-sized
 * steps of a circuit network) -- other shapes return TCMI_ERR_ARG for flags != 0 (a reverse sweep that carries the
 * CONJUGATED cotangent needs no conjugation at all: conj(gA) = tensordot(conj g, b)). */
int tcmi_tensordot_bits_ex(const void* a, int rank_a, const void* b, int rank_b, const int* axes_a, const int* axes_b,
                           int nk, const int* out_axes, int flags, void* c, int dtype, void* stream);
int tcmi_tensordot_bits_small_ok(int rank_a, int rank_b, int nk);

/* The same gate-sized steps, MANY per launch: tcmi_tensordot_small_desc() writes the 64-word job descriptor of one
 * tcmi_tensordot_bits_ex() call (host memory; the addresses a, b, c are baked in), tcmi_tensordot_small_batch() runs
 * `count` consecutive descriptors that live in DEVICE memory in one launch (jobs of one launch must not depend on each
 * other; max_log2_out = the largest log2(result elements) among them).  The steps of one level of a contraction tree
 * are independent: a circuit network's slice-invariant part and its reverse sweep become one launch per level instead
 * of one per contract_between (reference tensorcircuit/cons.py:948, called ~1000 times per contraction of a 30-qubit
 * ladder). */
int tcmi_tensordot_small_desc(const void* a, int rank_a, const void* b, int rank_b, const int* axes_a, const int* axes_b,
                              int nk, const int* out_axes, int flags, void* c, int* desc);
int tcmi_tensordot_small_batch(const int* desc_dev, int count, int max_log2_out, void* stream);

/* Batched complex GEMM C[M x N] = A[M x K] . B[K x N], row-major interleaved complex, strides in
 * elements between batch members; trans_a != 0: A is stored [K x M] (k-major).  complex64 runs on the f32 MFMA pipe
 * (v_mfma_f32_32x32x2_f32, exact f32 FMA), complex128 on the f64 MFMA pipe (v_mfma_f64_16x16x4_f64); both issue
 * Gauss's three real products per complex multiply.  Replaces backend.tensordot's GEMM (numpy/jax/torch BLAS in
 * the reference, cons.py:948 -> backend.tensordot). */
int tcmi_cgemm(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch,
               long long strideA, long long strideB, long long strideC, int trans_a, int dtype,
               void* stream);

/* The same product for complex64 with a k-major A (trans_a form: A [K x M], B [K x N]; M, N multiples of 128, K of 32,
 * 16-byte aligned operands, even strides) on the bf16 matrix pipe at f32 accuracy: every f32 operand value is cut exactly
 * into three bf16 pieces and a real product is the six piece products of order <= 2^-16 accumulated in f32
 * (v_mfma_f32_32x32x16_bf16); the dropped pieces are below one f32 rounding of the product (tcmi_gemm_split.hip; error
 * against a float64 product measured equal to tcmi_cgemm's).  2.7 x the MFMA rate of the exact-f32 pipe that tcmi_cgemm
 * uses.  The join of the cut contraction (reference circuit.py:701-721 -> cons.py:948). */
int tcmi_cgemm_split(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch,
                     long long strideA, long long strideB, long long strideC, void* stream);

/* tcmi_cgemm_split with a 4 x 4 gate applied to the product before it is stored: with P[b] = A[b]^T B[b] (M x N), u the
 * lowest bit of the row index and v the lowest bit of P's column index,
 *     C[b][(m, u')][col(c, v')] = sum_{u, v} X[b][2 u' + v'][2 u + v] P[b][(m, u)][(c, v)],   col(c, v) = c + v N / 2:
 * P's column index is C's column index rotated left by one bit (B's columns come from a half-circuit whose first qubit
 * is labelled last, so that the gate's right-hand qubit is bit 0).  `X` = complex64 [batch][16] (tcmi_cut_epilogue).  The
 * four (u, v) results sit in one thread of the MFMA result layout: 16 complex multiply-adds in registers.  Same
 * argument rules as tcmi_cgemm_split.  The join of a cut contraction whose last crossing gate is deferred: half the
 * bond dimension for ZZ / CNOT / CZ crossings (reference circuit.py:701-721 -> cons.py:948). */
int tcmi_cgemm_split_epi(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch,
                         long long strideA, long long strideB, long long strideC, const void* X, void* stream);

/* tcmi_cgemm_split (X = NULL) / tcmi_cgemm_split_epi (X given) with operands of KNOWN magnitude on the f16 matrix pipe:
 * every operand value times a power of two the caller passes (`scale_a` for A, `scale_b` for B) is cut into TWO f16
 * pieces, h = the nearest f16 and l = the nearest f16 of what h left (|scale x - h - l| <= 2^-22 |scale x|), and a real
 * product is the three piece products h h' + h l' + l h', each exact in the f32 accumulator of v_mfma_f32_32x32x16_f16 --
 * half the matrix instructions and two thirds of the LDS traffic of the three-piece bf16 cut.  What is dropped is at most
 * 3 * 2^-22 of |x y| per product; against a float64 product the kernel is measured at the error of the exact-f32 MFMA
 * kernel (tests/test_gpu_gemm_split.py).  The caller guarantees |scale_a re|, |scale_a im|, |scale_a (re + im)| < 65504
 * for every element of A (the same for B): an operand beyond that becomes inf and the product NaN -- loud, not wrong.
 * Values below 2^-14 / scale lose their low piece's bits gradually (f16 subnormals), an absolute error of 2^-25 / scale.
 * The cut contraction calls it with the bound it derives from its half-circuits (unitary gates and the operator-Schmidt
 * factors of the crossing gates: |amplitude| <= prod of spectral norms); other callers use tcmi_cgemm_split.
 * (reference circuit.py:701-721 -> cons.py:948 backend.tensordot of complex64 operands.) */
int tcmi_cgemm_split_f16(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch,
                         long long strideA, long long strideB, long long strideC, const void* X, float scale_a, float scale_b,
                         void* stream);

/* <a|b> = sum conj(a_i) b_i per batch element (states [batch][2^n], stride elements apart), accumulated in
 * float64 into `copies` replicated {re, im} pairs: out[b * out_batch_stride + 2 * copy + {0,1}] += ...; the
 * caller zeroes `out` and sums the copies.  Used for <psi|P|psi> of Pauli strings with more than two X/Y
 * factors (reference circuit.py:833-913 expectation of arbitrary operator lists): P|psi> by
 * tcmi_apply_pauli_sum, then this. */
int tcmi_vdot(const void* a, const void* b, double* out, long long state_stride, int batch, int n, int copies,
              long long out_batch_stride, int dtype, void* stream);

/* ---- MPS / TEBD (K7): reference tensorcircuit/mps_base.py:33-175 (FiniteMPS.apply_two_site_gate),
 * mpscircuit.py:35-64 (split_tensor), backend.svd truncation rule (backends/jax_backend.py:62-112),
 * backend.qr (tensornetwork decompositions). ---- */

/* Workspace size in bytes for tcmi_svd_trunc_batched (-1 on bad arguments). */
long long tcmi_svd_work_bytes(int m, int n, int batch, int dtype);

/* Batched thin SVD with the reference's truncation rule, one launch per <=256 workgroups.
 *   a   [batch][m][n] row-major complex, m <= n (callers pass the transpose otherwise)
 *   u   [batch][m][kmax], s [batch][m] real (descending, all m values), vh [batch][kmax][n]
 *   keep_out [batch] int  : min(max_singular_values (<=0: none), #{k : sqrt(sum_{j>=k} s_j^2) > err})
 *                           with err = max_truncation_err (* s_0 if relative); max_truncation_err < 0: none
 *   tw2_out  [batch] real : sum of the squared discarded singular values (fidelity factor 1 - tw2)
 *   absorb 0: u, vh isometries; 1: u <- u*s; 2: vh <- s*vh.   max_sweeps <= 0: default 30.
 * keep_out / tw2_out may be NULL.  One-sided Jacobi; work from tcmi_svd_work_bytes. */
int tcmi_svd_trunc_batched(const void* a, void* u, void* s, void* vh, int* keep_out, void* tw2_out, int m, int n,
                           int kmax, int batch, int max_singular_values, double max_truncation_err, int relative,
                           int absorb, int max_sweeps, void* work, long long work_bytes, int dtype, void* stream);

long long tcmi_qr_work_bytes(int m, int n, int batch, int dtype);

/* Batched Householder QR: a [batch][m][n] -> q [batch][m][K], r [batch][K][n], K = min(m, n);
 * q is a complete isometry also for rank-deficient a. */
int tcmi_qr_batched(const void* a, void* q, void* r, int m, int n, int batch, void* work, long long work_bytes,
                    int dtype, void* stream);

/* theta[l,a',b',r] = sum_{a,b} gate[a',b',a,b] t[l,a,b,r]; t, out [batch][L][2][2][R]; gate [16] complex per
 * batch element (gate_stride elements apart, 0 = shared). */
int tcmi_mps_gate_mix(const void* t, const void* gate, void* out, int L, int R, int batch, long long gate_stride,
                      int dtype, void* stream);

/* ---- plan-specialised pass kernels ---------------------------------------------------------------------
 * A pass descriptor can be compiled ahead of its first hot use into a straight-line gfx950 kernel (generator:
 * tensorcircuit-ng_amd/tcmi/specialize.py; same tables, same arithmetic bodies as the interpreting kernels behind
 * tcmi_run_pass / tcmi_run_adjoint_pass, which remain the fallback).  The library loads such a code object and
 * launches it; it does not compile.
 * Replaces: the compiled executable behind backend.jit (jax.jit of the circuit function,
 * tensorcircuit/backends/jax_backend.py `jit`; benchmarks/scripts/vqe_tc.py:136-141 jits the VQE step): compiled once
 * per circuit structure, replayed for every parameter value.
 * tcmi_spec_load: `path_host` = code-object file, `kernel_name_host` = its kernel, `lds_bytes` = dynamic LDS per
 * workgroup; the handle comes back through `handle_out_host` and belongs to the current device.
 * tcmi_spec_run_pass / _adjoint_pass: arguments as tcmi_run_pass / tcmi_run_adjoint_pass without the descriptor;
 * T = tile bits, LT = log2(threads per workgroup) of the plan the kernel was generated from. */
int tcmi_spec_load(const char* path_host, const char* kernel_name_host, int lds_bytes, void** handle_out_host);
int tcmi_spec_unload(void* handle);
/* What the generator knows about the kernel and the launchers must respect (0 after tcmi_spec_load):
 * TCMI_SPEC_FLAG_SRC = generated with the "src" option: its argument buffer is tcmi_spec_run_pass_from's; that launcher
 * refuses handles without the flag and tcmi_spec_run_pass refuses handles with it. */
#define TCMI_SPEC_FLAG_SRC 1
int tcmi_spec_set_flags(void* handle, int flags);
/* `live_mask` (over the n - T bits of the tile index; 0xffffffff = every tile): only the tiles whose index is zero outside
 * the mask get a workgroup.  A circuit started from |0...0> leaves every amplitude whose index has a 1 on a qubit no pass
 * has had in its tile yet exactly zero, and a pass acts inside its tiles: those tiles are zero before and after it, so
 * the first passes of a state (and the last passes of the reverse sweep, where psi is back to that shape and a zero psi
 * tile contributes to no gradient) run on the few tiles that can be non-zero (tcmi/executor.py live_masks).
 * `zero_bits` (forward pass; physical index bits inside the tile, 0 = none): amplitudes whose index has one of these bits
 * set are zero for the same reason and are not read -- they need not even have been written: a state whose passes are
 * all launched this way needs no zero fill, only its amplitude 0 set to 1 (CompiledCircuit.zero_start). */
int tcmi_spec_run_pass(void* handle, void* state, long long state_stride, int batch, int n, int T, int LT,
                       const void* ctab, const void* ptab, long long ptab_stride, unsigned live_mask, unsigned zero_bits,
                       void* stream);
/* The first pass of a plan whose input states are read from ANOTHER batch: state[b] <- pass(scale[b] * src[b >> src_shift])
 * (`scale` complex, one per output state, or NULL).  Replaces: the replication of the shared prefix states over the bond
 * digits of a cut contraction and their multiplication by the bond weights (reference circuit.py:701-721 contracts the two
 * halves of the network and joins them; here the K variants of a half share prefixes, tcmi/executor.py _HalfBatch) --
 * formerly an elementwise launch of its own that wrote the whole batch before this pass read it again.  The handle must
 * come from a kernel generated with the "src" option (tcmi/specialize.py). */
int tcmi_spec_run_pass_from(void* handle, void* state, long long state_stride, int batch, int n, int T, int LT,
                            const void* ctab, const void* ptab, long long ptab_stride, const void* src,
                            long long src_stride, int src_shift, const void* scale, void* stream);
int tcmi_spec_run_adjoint_pass(void* handle, void* psi, void* lam, long long state_stride, int batch, int n, int T,
                               int LT, const void* ctab, const void* ptab, long long ptab_stride, double* gout,
                               long long gout_stride, int gcopies, long long gcopy_stride, unsigned live_mask,
                               void* stream);

/* ---- the path's collective ---------------------------------------------------------------------------
 * Replaces: the sum over devices of the per-rank partial [value || gradients] of a sharded vmap batch or of the slices of
 * a sliced contraction (reference tensorcircuit/experimental.py:1145-1152 `jnp.sum(device_values, axis=0)` after the pmap
 * of 1125-1143; examples/slicing_auto_pmap_vqa.py:60-72) -- for hosts that have no process group of their own.
 * tcmi_comm_load: optional, names the librccl to open (NULL / not called: the one already in the process, else
 * "librccl.so", "librccl.so.1", the ROCm installation's).  tcmi_comm_unique_id: 128 bytes (TCMI_COMM_ID_BYTES) to be
 * created on ONE rank and handed to the others out of band (ncclGetUniqueId).  tcmi_comm_init: every rank, after
 * hipSetDevice, with the same id (ncclCommInitRank: collective, blocks until all `world` ranks have called).
 * tcmi_allreduce_sum: in place on `buf` (device), `count` elements of `dtype` (TCMI_F32 / TCMI_F64 / TCMI_C64 / TCMI_C128),
 * ordered on `stream` like every other entry point.  tcmi_comm_destroy: ncclCommDestroy. */
#define TCMI_COMM_ID_BYTES 128
int tcmi_comm_load(const char* librccl_path_host);
int tcmi_comm_unique_id(void* id_out_host);
int tcmi_comm_init(const void* id_host, int rank, int world, void** comm_out_host);
int tcmi_allreduce_sum(void* comm, void* buf, long long count, int dtype, void* stream);
int tcmi_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* TCMI_H */
