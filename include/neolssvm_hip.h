/*
 * neolssvm_hip.h - C ABI of the MI355X (gfx950) implementation of the neo-ls-svm fit/predict hot path.
 *
 * The reference (lsorber/neo-ls-svm, pure Python) has no FFI; its only seams for this path are two
 * private solver methods and three public inference methods.  Each entry point below names the
 * reference interface it replaces (file:line into the reference tree, src/neo_ls_svm/...):
 *
 *   nls_featuremap          RandomFourierFeatures.transform              _feature_maps.py:153-203
 *                           (AffineFeatureMap.transform inside it)       _affine_feature_map.py:72-92
 *   nls_gram_only           first product of _optimize_beta_gamma        _neo_ls_svm.py:110-114,127
 *   nls_rotate_only         leverage / numerator products of the same    _neo_ls_svm.py:128-143
 *   nls_eigh_only           eigh(A / c) (primal), eigh(sn K sn) (dual)   _neo_ls_svm.py:120, 265
 *   nls_tridiag_only        (its Householder tridiagonalisation stage)
 *   nls_primal_fit          NeoLSSVM._optimize_beta_gamma(phi, y, s, C)  _neo_ls_svm.py:77-189
 *                           fused with the transform that feeds it       _neo_ls_svm.py:386,401-402
 *   nls_primal_predict      decision_function / predict_std (primal)     _neo_ls_svm.py:661-665, 464-469,477
 *   nls_dual_fit            NeoLSSVM._optimize_alpha_gamma(X, y, s)      _neo_ls_svm.py:191-325
 *   nls_bin_stats(_labels)  per-bin weighted medians / deviations        _affine_normalizer.py:72-79
 *   nls_rank_codes          np.unique(y, return_inverse) of the quantiser _quantizer.py:246-253
 *   nls_dual_predict        decision_function / predict_std (dual)       _neo_ls_svm.py:666-671, 470-477
 *   nls_factor_create       cho_solve(self.L_, .) state of predict_std   _neo_ls_svm.py:464-469 (the factor kept on the device)
 *   nls_primal_fit_grid     the gamma x sigma grid (extension): the fit above once per sigma, one call
 *   nls_comm_*, nls_group_* (no counterpart: the reference is single-process; SURVEY.md 8(b) "multi-GPU is internal to the ctx", 8(e))
 *
 * Conventions
 *   - Every call returns 0 on success, non-zero on failure; nls_last_error() gives the message.
 *     NLS_ERR_ARG    bad argument               (the Python mirror raises ValueError)
 *     NLS_ERR_HIP    HIP / rocBLAS failure      (RuntimeError)
 *     NLS_ERR_LINALG rocSOLVER info != 0        (numpy.linalg.LinAlgError, like scipy's cho_factor)
 *     NLS_ERR_COMM   a collective failed: the all-reduce hook or RCCL returned an error, ANOTHER rank of a sharded call failed (the message
 *                    names it), or a collective did not complete within the deadline (RuntimeError)
 *   - All matrices are dense, C-contiguous (row-major) float64 unless stated; complex values are
 *     interleaved (re, im) pairs, i.e. numpy complex128.
 *   - Bulk inputs (X, y, s, Xt, Xq) may be HOST or DEVICE pointers; the library asks the HIP runtime
 *     which (hipPointerGetAttributes).  Device-resident inputs are used in place, host inputs are
 *     staged with one H2D copy.  Small parameter arrays (shift, scale, B, gammas, beta, L, alpha) and
 *     all outputs are host pointers.  Nothing is retained after a call returns.
 *   - Calls are blocking; one host thread per context.  Several contexts of one process may run fits on the same GPU at the same time (each
 *     owns its streams, handles and workspace): tests/test_gpu_two_contexts.py holds two contexts with fits in flight to the bits of the same fits
 *     run alone.  (Round 3's failures in this mode were rocsolver_zpotrf, which is not safe on two handles at once in this ROCm build -
 *     profiles/r04_two_contexts.md; the fit path no longer calls it.  NLS_POTRF=rocsolver, a diagnostic knob, brings it back: single context only.)
 *     The library never uses a CPU fallback.
 *   - Multi-GPU: one context per GPU - the contexts of N launched processes (nls_comm_init_rank, rows sharded by the caller), or the member
 *     contexts of ONE process's group (nls_group_create: rows sharded inside nls_group_primal_fit).  Either way the path exchanges data at
 *     four points: {sum s, sum s*y, n}, the Hermitian block A||b (sum all-reduce), the eigenvectors (rank 0 runs the
 *     tridiagonal eigensolver and broadcasts; every rank back-transforms one column block; all-gather) and the
 *     per-gamma error vectors (sum all-reduce).  The collectives are RCCL calls on the library's own stream once
 *     nls_comm_init_rank has joined the context to a communicator (librccl is loaded on first use; no PyTorch
 *     anywhere); alternatively a caller-supplied all-reduce hook (nls_set_allreduce: CPU tests over gloo, several
 *     ranks sharing one GPU).  Without either the context is single-rank.
 */
#ifndef NEOLSSVM_HIP_H
#define NEOLSSVM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLS_OK 0
#define NLS_ERR_ARG 1
#define NLS_ERR_HIP 2
#define NLS_ERR_LINALG 3
#define NLS_ERR_COMM 4

#define NLS_ABI_VERSION 4
#define NLS_NUM_TIMINGS 24

typedef struct nls_ctx nls_ctx;

/* Sum-all-reduce of `count` doubles at device address `buf`, in place, over all ranks; return 0 on
 * success.  The library has synchronised its stream before the call and expects the result to be
 * complete (visible to any stream) on return. */
typedef int (*nls_allreduce_fn)(void* buf, size_t count, void* user);

/* ---- context ---------------------------------------------------------------------------------- */
int nls_abi_version(void);
int nls_ctx_create(int device, nls_ctx** ctx);
void nls_ctx_destroy(nls_ctx* ctx);
/* Message of the last failure on this context (ctx == NULL: last failure of nls_ctx_create). */
const char* nls_last_error(const nls_ctx* ctx);
/* Register the collective hook; world == 1 or fn == NULL resets to single-rank. */
int nls_set_allreduce(nls_ctx* ctx, nls_allreduce_fn fn, void* user, int rank, int world);
/* Upper bound, in bytes, on the device workspace the context may hold (0 = default: 60% of HBM).  Row chunks are
 * planned against it; a workspace request that would take the total past an explicitly set limit (the n x n
 * buffers of the eigendecomposition / dual path cannot be chunked) fails with NLS_ERR_ARG instead of allocating. */
int nls_set_workspace_limit(nls_ctx* ctx, size_t bytes);
/* Free the context's workspace buffers of at least min_bytes each (0 = all of them) and return the bytes still held.
 * The workspace is a grow-only arena between calls (no hipMalloc in steady state); this is the trim. */
int nls_ws_release(nls_ctx* ctx, size_t min_bytes, size_t* still_held);

/* ---- multi-GPU: native RCCL communicator (one process per GPU) ----------------------------------------------
 * Rank 0 calls nls_comm_get_unique_id and hands the NLS_COMM_ID_BYTES bytes to the other ranks by any means (file,
 * socket, environment); every rank then calls nls_comm_init_rank on its own context (collective, blocks until all
 * ranks arrive).  From then on nls_primal_fit treats X, y, s as this rank's row block.  nls_comm_allreduce is the
 * small utility collective a driver needs around the library (barrier, max of a timing): count doubles at a HOST
 * address (at most NLS_COMM_UTIL_MAX), op 0 = sum, 1 = max. */
#define NLS_COMM_ID_BYTES 128
#define NLS_COMM_UTIL_MAX 8192
int nls_comm_get_unique_id(void* id);
int nls_comm_init_rank(nls_ctx* ctx, const void* id, int rank, int world);
int nls_comm_destroy(nls_ctx* ctx);
int nls_comm_allreduce(nls_ctx* ctx, double* host_values, size_t count, int op);
/* Failure of ONE rank of a sharded call is an error on EVERY rank, never a hang (the reference has nothing distributed; SURVEY.md section 5:
 * "surface HIP/RCCL errors as status codes"):
 *   - before every exchange the ranks all-reduce one status slot per rank.  A rank whose local work failed (allocation, launch, argument,
 *     factorisation) does not leave early: it skips to that vote, and all ranks return from the call at the same point - the failed rank
 *     with its own code, the others with NLS_ERR_COMM and a message naming the rank and its code (NLS_ERR_LINALG, a property of the shared
 *     problem, is returned as NLS_ERR_LINALG everywhere).  The communicator stays usable.
 *   - the host never blocks in the runtime behind a collective: it polls the stream, and gives the communicator up (ncclCommAbort) on an
 *     asynchronous RCCL error or when the collective has not completed after the time-out (a peer process died or is stuck): NLS_ERR_COMM.
 *     The context then refuses collective work until it joins a new communicator (nls_comm_init_rank) or leaves (nls_comm_destroy).
 *     A rank that failed must NOT be restarted by replacing its process image (exec) once it has touched the GPU: exit non-zero, start a child.
 * nls_comm_set_timeout: seconds > 0, or 0 for the default (environment NLS_COMM_TIMEOUT_S, else 300).  The first collective of a call also
 * waits for ranks that enter the call late - choose it above the largest skew between the ranks' arrivals.
 * nls_comm_abort: gives the context's communicator up now (e.g. from a signal handler's deferred work when the job is being cancelled).
 * nls_comm_state: 0 no communicator, 1 joined, 2 aborted (collective calls fail until nls_comm_init_rank / nls_comm_destroy). */
/* Measurement hook (bench.py --as-rank r --of W; no production use): on a context that is the ONLY rank of a native communicator,
 * world > 1 makes nls_primal_fit do the share of rank `rank` of `world` ranks on the rows it is given - the tridiagonal eigensolver only
 * as rank 0, its own column block of the back-transformation, and every exchange with its real payload pushed through the one-rank
 * communicator (the enqueue and the local pass of the collective are timed; no link is).  The other ranks' eigenvector blocks (and, for
 * rank != 0, rank 0's eigenpairs) come from a copy that a complete call on the SAME inputs with capture = 1 (world <= 1) left on the
 * device, so the fitted results are those of the complete call.  world <= 1, capture = 0: off. */
int nls_comm_set_virtual_rank(nls_ctx* ctx, int rank, int world, int capture);
int nls_comm_set_timeout(nls_ctx* ctx, double seconds);
int nls_comm_abort(nls_ctx* ctx);
int nls_comm_state(const nls_ctx* ctx);

/* ---- device memory plumbing (so a host without a GPU array library can keep inputs resident) -- */
int nls_device_malloc(nls_ctx* ctx, size_t bytes, void** dptr);
int nls_device_free(nls_ctx* ctx, void* dptr);
int nls_memcpy_h2d(nls_ctx* ctx, void* dst, const void* src, size_t bytes);
int nls_memcpy_d2h(nls_ctx* ctx, void* dst, const void* src, size_t bytes);
int nls_synchronize(nls_ctx* ctx);
/* Page-lock / release a HOST buffer that the caller will pass as a large output more than once (the L_ of the fits: 268 MB at D = 4096, 800 MB
 * at n = 10^4): into page-locked memory the finished block columns of the factor travel with asynchronous copies beside the factorisation at the
 * full PCIe rate; into pageable memory they are staged (~ 6-10 GB/s).  Registering costs ~ 10 ms per 268 MB - once per buffer, not per call
 * (the Python mirror registers the buffers of its output pool, neo_ls_svm_amd/_hostpool.py).  Unregister BEFORE the memory is unmapped; ctx may be
 * NULL there (the registering context may already be destroyed). */
int nls_host_register(nls_ctx* ctx, void* ptr, size_t bytes);
int nls_host_unregister(nls_ctx* ctx, void* ptr);
/* Device description: name (<=255 chars), CU count, HBM bytes. */
int nls_device_info(nls_ctx* ctx, char* name, int name_len, int* compute_units, size_t* hbm_bytes);

/* ---- K1: feature map --------------------------------------------------------------------------
 * phi[i, j<D] = exp(-i * t_ij) / sqrt(D),  t = ((X - shift) / scale) @ B;  phi[i, D] = 1.
 * X: n x d.  shift, scale: d.  B: d x D (the separator matrix with Z already folded in,
 * _feature_maps.py:150).  phi: n x (D+1) complex128, host or device pointer. */
int nls_featuremap(nls_ctx* ctx, const double* X, int64_t n, int d, const double* shift,
                   const double* scale, const double* B, int D, double* phi);

/* ---- K2: weighted Hermitian normal equations (test / bench hook) ------------------------------
 * s is normalised by its (global) sum inside, as _neo_ls_svm.py:110.  A: (D+1) x (D+1) complex128
 * row-major, full Hermitian; b: (D+1) complex128.  With a collective hook the result is the
 * all-reduced block on every rank. */
int nls_gram_only(nls_ctx* ctx, const double* X, const double* y, const double* s, int64_t n, int d,
                  const double* shift, const double* scale, const double* B, int D, double* A,
                  double* b);

/* ---- K4: rotation with fused epilogue (test / bench hook) ---------------------------------------
 * P = phi(X) Q;  U = Re(P o v^T),  Gm = |P|^2  (the two n x (D+1) real matrices the gamma sweep contracts,
 * _neo_ls_svm.py:128-143 in the one-rotation form).  Q: (D+1) x (D+1) complex128 row-major, v: (D+1)
 * complex128, both host.  U, Gm: n x (D+1) float64 host outputs, either may be NULL (timing only). */
int nls_rotate_only(nls_ctx* ctx, const double* X, int64_t n, int d, const double* shift, const double* scale,
                    const double* B, int D, const double* Q, const double* v, double* U, double* Gm);

/* ---- P4 / D2: eigendecomposition (test / bench hooks) -------------------------------------------
 * The serial section of both fits: scipy.linalg.eigh(A / c) of the (D+1) x (D+1) Hermitian normal matrix
 * (_neo_ls_svm.py:120) and numpy.linalg.eigh(sn K sn) of the n x n kernel (:265).  A: n x n column-major, host,
 * complex128 (is_complex = 1) or float64 (0); only the lower triangle is read.
 * nls_tridiag_only: A = Q T Q^H with LAPACK zhetrd / dsytrd (uplo = 'L') conventions: d[n], e[n-1], tau[n-1]
 *   (complex128 or float64) and the reflectors written below the sub-diagonal of A (d, e on its diagonals).
 * nls_eigh_only: eigenvalues ascending in lam[n], eigenvectors in the columns of A.  NLS_EVD=rocsolver in the
 *   environment selects rocSOLVER's zheevd / dsyevd for every eigendecomposition of the library. */
int nls_tridiag_only(nls_ctx* ctx, void* A, int n, int is_complex, double* d, double* e, void* tau);
int nls_eigh_only(nls_ctx* ctx, void* A, int n, int is_complex, double* lam);
/* The eigendecompositions above take a TWO-STAGE reduction for large real matrices (n >= 4500; dense -> band of width bw -> tridiagonal, two
 * back-transformations; csrc/nls_sb.h, nls_chase.h, nls_q2.h) and the one-stage panel otherwise (complex matrices always): NLS_EVD=twostage /
 * onestage forces / forbids it, NLS_TWOSTAGE_MIN moves the size rule (both arithmetics), NLS_SB_BW = 32 / 64 the band width of real matrices
 * (default 32; complex: 32).  nls_twostage_stage runs ONE stage of it on host data (tests, profiling):
 *   stage 1: A (n x n column-major, lower) -> band in its bw sub-diagonals + block reflectors below; aux = tau1[n]; info = {failure flag
 *            (a panel that could not be orthogonalised: the library then repeats the reduction with perturbed panels, then falls back to the
 *            one-stage panel), columns reduced};
 *   stage 2: the band held in A's bw sub-diagonals -> d[n], e[n-1]; aux = the chase reflectors V2 (n x n); info[0] != 0: time-out;
 *   stage 3: aux (n x ncols, column-major) <- Q2 aux with the chase reflectors V2 handed in as A.
 * nls_twostage_rescues: eigendecompositions of this context whose band reduction met a panel it could not orthogonalise (columns dependent to
 * working precision) and went through at the second attempt, every panel perturbed by 1e-13 of its norm (csrc/nls_sb.h, k_sb_perturb);
 * nls_twostage_fallbacks: those that failed again (zero panels) and fell back from the two-stage to the one-stage reduction. */
/* The dual fit's own Cholesky factorisation (csrc/nls_potrf.h; cho_factor(gamma* diag(sn^-2) + K), _neo_ls_svm.py:313-314) on host data (tests,
 * profiling): A (n x n column-major doubles, lower triangle) is overwritten by L; *info = 0 or the 1-based index of the first non-positive pivot. */
int nls_cholesky_only(nls_ctx* ctx, double* A, int n, int* info);
/* The primal fit's own complex Cholesky factorisation (csrc/nls_zpotrf.h; cho_factor(gamma* C + A), _neo_ls_svm.py:176-177) on host data (tests,
 * profiling): A (n x n column-major, interleaved re / im, lower triangle) is overwritten by L; *info as above. */
int nls_zcholesky_only(nls_ctx* ctx, double* A, int n, int* info);
int nls_twostage_stage(nls_ctx* ctx, int stage, void* A, int n, int is_complex, int bw, void* aux, double* d, double* e, int ncols,
                       int* info);
long nls_twostage_fallbacks(const nls_ctx* ctx);
long nls_twostage_rescues(const nls_ctx* ctx);
/* Stage times (milliseconds, HIP events on the library's stream) of the context's most recent eigendecomposition (P4 `_neo_ls_svm.py:120`,
 * D2 `:265`).  out[0..7] = {reduction to tridiagonal form (one-stage panel) or to band form (two-stage), bulge chase (two-stage only, else 0),
 * tridiagonal eigensolver (stedc, incl. its broadcast in a collective fit), second back-transformation Q2 (two-stage only, else 0), first
 * back-transformation Q1, their sum, n, kind} with kind = 1: one-stage real, 2: one-stage complex, 3: two-stage real, 4: two-stage complex,
 * 5: rocSOLVER heevd / syevd in one call (only the sum is filled), 0: no eigendecomposition has run (returns NLS_ERR_ARG). */
int nls_evd_stage_ms(nls_ctx* ctx, double* out8);
/* The tridiagonal eigensolver of both eigendecompositions alone, on host data (tests, profiling): the library's own divide and conquer
 * (csrc/nls_stedc.h; NLS_STEDC=rocsolver: rocsolver_dstedc).  d[n]: diagonal in, eigenvalues ascending out; e[n - 1]: off-diagonal; Q: n x n
 * column-major eigenvectors out.  LAPACK's dstedc with compz = 'I' (what scipy / numpy eigh run under _neo_ls_svm.py:120 / :265). */
int nls_stedc_only(nls_ctx* ctx, double* d, const double* e, int n, double* Q);

/* ---- primal fit ------------------------------------------------------------------------------- */
typedef struct nls_primal_fit_args {
  /* inputs */
  const double* X;      /* n x d (local rows)                                  host | device */
  const double* y;      /* n     targets: float for regression, +-1 for classification       */
  const double* s;      /* n     sample weights, un-normalised (>= 0)                        */
  const double* shift;  /* d                                                   host          */
  const double* scale;  /* d     (non-zero)                                    host          */
  const double* B;      /* d x D folded projection                             host          */
  const double* gammas; /* G     regularisation grid (reference: logspace(1e-6, 20, 1024))   */
  int64_t n;
  int32_t d, D, G;
  int32_t is_classifier;  /* 0 regressor, 1 classifier (residual clipping + hinge selection)  */
  int32_t gamma_index_in; /* >= 0 forces the selected grid index, -1 = argmin as the reference */
  int32_t flags;          /* NLS_FIT_* bits                                                    */
  const double* Cmat;     /* complexity matrix C of the penalty gamma beta^H C beta: NULL = identity (the reference's */
                          /* fast diagonal approximation, _feature_maps.py:129-135), else (D+1) x (D+1) real      */
                          /* symmetric positive definite, row-major, host: the generalised-EVD branch            */
                          /* eigh(A, b=C) of _neo_ls_svm.py:122-124,131,139                                      */
  double finish_below;    /* NLS_FIT_FINISH_IF_BELOW: threshold on the selected objective         */
  /* outputs (host; any may be NULL) */
  double* beta;          /* 2 (D+1)   fitted weights, complex128                              */
  double* L;             /* 2 (D+1)^2 cho_factor(gamma* C + A) as scipy returns it: upper; only that triangle is defined */
                         /*           triangular factor U (A = U^H U), row-major, lower=False */
  double* lam;           /* D+1       eigenvalues of A / c (ascending)                        */
  double* loo_errors;    /* G         s @ |e_loo(gamma)|                  (loo_errors_gammas_) */
  double* objective;     /* G         the vector whose argmin selects gamma                   */
  double* loo_residuals; /* n         column of the selected gamma                            */
  double* loo_leverage;  /* n                                                                 */
  double* loo_std;       /* n                                                                 */
  double* residuals;     /* n         Re(phi beta) - y with the RETURNED beta (_neo_ls_svm.py:178-182; clipped for a classifier).     */
                         /*           When beta is the Cholesky re-solve (L requested, or a sharded fit) one more pass over the       */
                         /*           feature planes evaluates it (8-13 ms at n = 10^6, D = 4096); NLS_FIT_RESIDUALS_FROM_SWEEP takes */
                         /*           the sweep table's column instead (the eigendecomposition's beta(gamma*): equal to              */
                         /*           cond(gamma* C + A) eps, ~1e-9 relative).  With L == NULL on one rank beta IS that vector.       */
  double* loo_score;     /* 1         weighted accuracy / R^2 of the LOO predictions          */
  int32_t* gamma_index;  /* 1         selected grid index                                     */
  int32_t* finished;     /* 1         1 when P8 / P9 ran (beta, L, residuals, row outputs written), else 0 */
  double* timings;       /* NLS_NUM_TIMINGS seconds per stage, see NLS_T_* (HIP events)       */
} nls_primal_fit_args;

/* flags */
#define NLS_FIT_SWEEP_ONLY 1      /* stop after the gamma selection (P1-P7): no Cholesky re-solve, no residuals / L / beta */
#define NLS_FIT_FINISH_IF_BELOW 2 /* run P8 / P9 only when objective[selected] < finish_below: a gamma x sigma grid    */
                                  /* finishes only the sigmas that beat the incumbent (the others need the curve only) */
#define NLS_FIT_RESIDUALS_FROM_SWEEP 4 /* residuals: the sweep table's column, no extra pass (see `residuals` above)   */

/* indices into timings[] */
#define NLS_T_TOTAL 0
#define NLS_T_UPLOAD 1
#define NLS_T_FEATUREMAP 2   /* all K1 launches                                  */
#define NLS_T_GRAM 3         /* K2 launches incl. slab reduction                 */
#define NLS_T_ALLREDUCE 4
#define NLS_T_EVD 5          /* assembly of A / c + eigendecomposition (own tridiagonalisation, stedc, back-transformation) + rotation planes */
#define NLS_T_ROTATE 6       /* K4: P = phi Q with fused |P|^2, Re(P v) epilogue  */
#define NLS_T_SWEEP 7        /* K5: the two gamma-sweep GEMMs                     */
#define NLS_T_LOO 8          /* LOO residual epilogue + weighted reductions       */
#define NLS_T_CHOLESKY 9     /* the Cholesky factor L_ (side stream) + the re-solve beta = cho_solve(L_, b)      */
#define NLS_T_RESIDUALS 10   /* Re(phi beta) - y                                  */
#define NLS_T_DOWNLOAD 11
#define NLS_T_ROTATE_LAUNCHES 12 /* number of K4 launches (for the per-launch roofline)       */
#define NLS_T_GRAM_LAUNCHES 13
#define NLS_T_SWEEP_LAUNCHES 14
#define NLS_T_FEATUREMAP_LAUNCHES 15
#define NLS_T_ROTATE_FLOPS 16    /* algorithmic flops of all K4 launches: 8 n (D+1)^2          */
#define NLS_T_GRAM_FLOPS 17      /* 4 n (D+1)^2                                               */
#define NLS_T_SWEEP_FLOPS 18     /* 4 n (D+1) G                                               */
#define NLS_T_FEATUREMAP_FLOPS 19 /* 2 n d D per pass                                         */
#define NLS_T_ROW_CHUNK 20       /* rows per chunk used                                       */

int nls_primal_fit(nls_ctx* ctx, const nls_primal_fit_args* args);

/* ---- gamma x sigma leave-one-out grid (BASELINE config 5; SURVEY.md 8(b): the `sigmas[Sg]` / `sigma_idx` / `loo_errors[Sg G]` arguments of
 * the nls_primal_fit row, 8(d): its definition) ----------------------------------------------------------------------------------------
 * The reference fixes the kernel bandwidth in closed form (_affine_separator.py:200-209); the grid extends the search with multipliers
 * sigma_k that divide the folded projection (T / sigma_k, i.e. B / sigma_k).  For every sigma one fit runs P2-P7 on args->gammas - ONE
 * eigendecomposition per sigma is the factorisation all gammas reuse (_neo_ls_svm.py:120,146-150) - and the (sigma, gamma) pair with the
 * smallest selection objective wins.  What the call does, in this order:
 *   1. this rank's sigmas (k = rank, rank + world, ...) are visited nearest to 1 first (|ln sigma_k| ascending, ties by index): sigma = 1
 *      is the separator's own bandwidth, so the incumbent is good from the start;
 *   2. the first sigma is finished unconditionally (P8 / P9: Cholesky re-solve, row outputs, L); a later one only when its selected
 *      objective is STRICTLY below the incumbent's (NLS_FIT_FINISH_IF_BELOW) - its outputs then replace the incumbent's in args' buffers;
 *   3. with world > 1 and a merge context the Sg x G tables (each sigma owned by one rank, zeros elsewhere) are summed over the ranks -
 *      after a status vote on the merge communicator: a rank whose own fits failed takes every rank out of the call (see nls_comm_*);
 *      without one the rows of the other ranks' sigmas are NaN;
 *   4. sigma_index = first minimum over the owned sigmas of min_g objective[k][g] (numpy.argmin: ties go to the SMALLEST index); unmerged,
 *      a tie that includes the finished incumbent goes to the incumbent (it carries the full result); gamma_index = argmin_g of that row.
 * args: as for nls_primal_fit, with args->B the UNSCALED projection; args->gamma_index_in must be -1 and args->flags 0.  The row / factor /
 * beta outputs of args hold the winner's full result iff *best_valid == 1 (always on one rank; after a merge on the rank that owns the
 * winning sigma); args->loo_errors / objective / gamma_index / lam receive the winning sigma's curve, selected index and spectrum when it
 * was fitted by this rank.  merge: a context that has joined a communicator (nls_comm_init_rank) and is NOT the fitting context - a fitting
 * context inside a communicator makes every fit a row-sharded collective - or NULL. */
typedef struct nls_sigma_grid {
  const double* sigmas;    /* Sg   multipliers, > 0                                                     */
  int32_t Sg;
  int32_t rank, world;     /* sigma sharding: this call fits k = rank, rank + world, ... (0, 1: all)     */
  nls_ctx* merge;          /* communicator-only context for the merge of the small tables, or NULL       */
  double* loo_errors;      /* Sg x G   s @ |e_loo| per (sigma, gamma)                  (may be NULL)     */
  double* objective;       /* Sg x G   the selection objective                         (may be NULL)     */
  double* seconds;         /* Sg       wall time of each sigma's fit                   (may be NULL)     */
  int32_t* sigma_index;    /* 1        winning sigma                                                     */
  int32_t* gamma_index;    /* 1        winning gamma index of that sigma                                 */
  int32_t* best_valid;     /* 1        1: args' beta / L / row outputs are the winner's full result      */
  int32_t* finished_count; /* 1        how many of this rank's sigmas ran P8 / P9      (may be NULL)     */
  double* timings;         /* NLS_NUM_TIMINGS  stage seconds summed over this rank's sigmas (may be NULL) */
} nls_sigma_grid;
int nls_primal_fit_grid(nls_ctx* ctx, const nls_primal_fit_args* args, const nls_sigma_grid* grid);
/* Test hooks of the grid's bookkeeping (host arithmetic only, no GPU needed).  nls_grid_visiting_order: step 1 - order[] receives this
 * rank's sigma indices in visiting order, the return value is their count.  nls_grid_select: step 4 on an Sg x G objective table with
 * owned[k] != 0 marking the rows that count; incumbent >= 0: the unmerged tie rule (a tie that includes it goes to it), -1: merged. */
int nls_grid_visiting_order(const double* sigmas, int Sg, int rank, int world, int32_t* order);
int nls_grid_select(const double* objective, const unsigned char* owned, int Sg, int G, int incumbent, int32_t* sigma_index, int32_t* gamma_index);

/* ---- several GPUs behind ONE handle and ONE host call (`nls_ctx_create(const int* devs, int ndev, ...)`, "multi-GPU is
 * internal to the ctx ... the Python surface is identical at 1 and 8 GPUs"; 8(e)) -------------------------------------------------------
 * A group owns one context per listed device and, for ndev > 1, an RCCL communicator that joins them (several ranks of ONE process, one
 * per device: librccl is loaded on first use).  Every group call fans out to one short-lived host thread per device, each driving its own
 * context, stream and collectives - exactly the per-rank code path of the process-per-GPU launch (nls_comm_init_rank + nls_primal_fit per
 * rank), so the two deployments cannot diverge.
 *   nls_group_primal_fit     args as nls_primal_fit with n = ALL rows and HOST pointers X, y, s (device pointers only when every member
 *                            context sits on the device that holds them); rank r takes the contiguous row block [n r / ndev, n (r + 1) / ndev)
 *                            (_neo_ls_svm.py:110: s is normalised by the GLOBAL sum; one all-reduce of the packed Hermitian block A || b
 *                            before the eigendecomposition); row outputs land in the caller's n-vectors at the block's offset, the
 *                            replicated outputs (beta, L, lam, curves, score, timings) are rank 0's.  n >= ndev.
 *   nls_group_primal_fit_grid  the gamma x sigma grid with the SIGMAS dealt over the devices (every device holds all rows, no collective in
 *                            the data path; the tables are merged on the host); grid->rank / world / merge must be 0 / 1 / NULL.  Ranks > 0
 *                            keep their incumbent's full result in host buffers of the group (4 n + 2 (D+1)^2 [+ 2 (D+1)] doubles each,
 *                            allocated on first use and reused); the winner's is copied into the caller's buffers at the end.
 *   nls_group_primal_predict query rows sharded the same way; factor: a group factor (U^-1 resident on every member) or NULL with L.
 * The dual path does not shard (its n x n eigendecomposition): use nls_group_ctx(group, 0) with nls_dual_fit ("replicas only").
 * devices may name one device several times ONLY with a communication library that allows it (the test stand-in of tests/csrc/rccl_shim.cpp;
 * RCCL itself refuses two ranks on one device). */
typedef struct nls_group nls_group;
typedef struct nls_group_factor nls_group_factor;
int nls_group_create(const int* devices, int ndev, nls_group** group);
void nls_group_destroy(nls_group* group);
/* Message of the last failure of a group call (group == NULL: of nls_group_create); names the rank that failed first.  A group call
 * returns - it does not hang - when one member fails: through the status votes above, or, for a failure outside them (an exception on
 * the member's thread, a failed RCCL call), through the group's abort flag, which the other members' waits poll.  The code returned is
 * the failing member's own.  If the failure cost members their communicator the group joins a fresh one at its next sharded call. */
const char* nls_group_last_error(const nls_group* group);
int nls_group_size(const nls_group* group);
/* Member context of rank r (owned by the group): pre-step statistics on rank 0, workspace limits, the dual path. */
nls_ctx* nls_group_ctx(nls_group* group, int rank);
int nls_group_primal_fit(nls_group* group, const nls_primal_fit_args* args);
int nls_group_primal_fit_grid(nls_group* group, const nls_primal_fit_args* args, const nls_sigma_grid* grid);
int nls_group_factor_create(nls_group* group, const double* L, int D, nls_group_factor** factor);
int nls_group_factor_destroy(nls_group* group, nls_group_factor* factor);
int nls_group_primal_predict(nls_group* group, const double* X, int64_t m, int d, const double* shift, const double* scale, const double* B,
                             int D, const double* beta, const double* L, const nls_group_factor* factor, double* yhat, double* sigma);

/* Test hook of the compressed gamma sweep (host arithmetic only, no GPU needed).  For a strictly increasing positive grid of
 * more than 256 points spanning at most e^17.2 (the reference's: 2e7) nls_primal_fit evaluates the rational functions of _neo_ls_svm.py:146-149 at NLS_SWEEP_NODES
 * Chebyshev nodes in ln(gamma) and interpolates:  1 / (gammas[g] + lam) = sum_q W[q][g] / (nodes[q] + lam)  for every
 * lam >= 0, to rounding.  nodes: NLS_SWEEP_NODES, W: NLS_SWEEP_NODES x G row-major.  *applies = 0 when the grid takes the
 * direct product instead (nodes / W untouched). */
#define NLS_SWEEP_NODES 128
int nls_sweep_weights(const double* gammas, int G, double* nodes, double* W, int* applies);

/* ---- primal inference -------------------------------------------------------------------------
 * A factor handle keeps U^-1 of one fitted Cholesky factor on the device (as the B-operand planes of the rotation
 * kernel), so that repeated predict_std calls skip the (D+1)^2 upload and the triangular inversion.  The handle is
 * explicit state owned by the caller: create it from L_ after a fit, destroy it (or the context) when done.
 * L: (D+1) x (D+1) complex128, scipy cho_factor(lower=False) layout as returned by nls_primal_fit, host or device. */
typedef struct nls_factor nls_factor;
int nls_factor_create(nls_ctx* ctx, const double* L, int D, nls_factor** factor);
int nls_factor_destroy(nls_ctx* ctx, nls_factor* factor);

/* yhat[i] = Re(phi(x_i) . beta); sigma[i] = sqrt(Re phi_i (U^H U)^-1 phi_i^H).  Either output may be NULL.
 * sigma needs the factor: a handle (preferred), or L itself (inverted on the fly, nothing kept). */
int nls_primal_predict(nls_ctx* ctx, const double* X, int64_t m, int d, const double* shift,
                       const double* scale, const double* B, int D, const double* beta,
                       const double* L, const nls_factor* factor, double* yhat, double* sigma);

/* ---- supervised normaliser statistics (next row after the hot path) ------------
 * Per class bin b and input column j: the weighted median of X[bin b, j] (weighted_quantile(.., 0.5), the average of
 * the lower- and upper-cumulative-weight interpolants, _weighted_quantile.py:35-63) and the weighted mean absolute
 * deviation about it - the two statistics AffineNormalizer.fit builds shift_/scale_ from (_affine_normalizer.py:72-79).
 * perm: the n row indices grouped by bin (stable order), bin_off: nbins + 1 offsets into perm.  X, s: host | device.
 * centers, spreads: nbins x d, host. */
int nls_bin_stats(nls_ctx* ctx, const double* X, const double* s, int64_t n, int d, const int32_t* perm,
                  const int64_t* bin_off, int nbins, double* centers, double* spreads);
/* The same statistics from the per-row bin labels (0 .. nbins - 1, host): the grouping (numpy's argsort(labels, kind="stable") + bincount /
 * cumsum in the call above's caller) runs on the device - one stable radix sort of (label, row) pairs - and yields the same permutation, hence
 * bit-identical statistics. */
int nls_bin_stats_labels(nls_ctx* ctx, const double* X, const double* s, int64_t n, int d, const int32_t* labels, int nbins,
                         double* centers, double* spreads);
/* Rank codes of the targets: inverse[i] = rank of y[i] among the distinct values of y, *nunique = their number - what
 * numpy.unique(y, return_inverse=True) gives the target quantiser (sample_bins_quantized_ecdf, _quantizer.py:246-253).  y: n finite doubles,
 * host | device; inverse: n int64, host.  (-0.0 and +0.0 are one value, as numpy compares them.) */
int nls_rank_codes(nls_ctx* ctx, const double* y, int64_t n, int64_t* inverse, int64_t* nunique);

/* ---- dual fit --------------------------------------------------------------------------------- */
typedef struct nls_dual_fit_args {
  const double* Xt;     /* n x r  affine-transformed training rows (X_)        host | device */
  const double* y;      /* n */
  const double* s;      /* n      strictly positive weights (zero-weight rows dropped by caller,
                                  _neo_ls_svm.py:388-389) */
  const double* gammas; /* G      (reference: logspace(1e-6, 20, 128)) */
  int64_t n;
  int32_t r, G;
  int32_t is_classifier;
  int32_t gamma_index_in;
  double* alpha;         /* n */
  double* L;             /* n x n  cho_factor(gamma* diag(sn^-2) + K), upper, row-major; only that triangle is defined.  A pageable host
                            buffer is filled by short-lived helper threads of the call (its pages touched behind the eigendecomposition, the
                            finished block columns copied beside the factorisation), a page-locked one (nls_host_register) by asynchronous
                            copies; NULL: no factorisation (alpha is then the eigendecomposition's) */
  double* lam;           /* n */
  double* loo_errors;    /* G */
  double* objective;     /* G */
  double* loo_residuals; /* n */
  double* loo_std;       /* n */
  double* residuals;     /* n */
  double* loo_score;     /* 1 */
  int32_t* gamma_index;  /* 1 */
  double* timings;       /* NLS_NUM_TIMINGS */
} nls_dual_fit_args;

int nls_dual_fit(nls_ctx* ctx, const nls_dual_fit_args* args);

/* yhat = k(Xq, Xt) alpha + sum(alpha); sigma = sqrt(1 - sum K o cho_solve(L, K^T)^T). */
int nls_dual_predict(nls_ctx* ctx, const double* Xq, int64_t m, const double* Xt, int64_t n, int r,
                     const double* alpha, const double* L, double* yhat, double* sigma);

#ifdef __cplusplus
}
#endif
#endif /* NEOLSSVM_HIP_H */
