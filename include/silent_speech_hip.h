/* silent_speech_hip.h -- C ABI of libsilent_speech_hip.so: the MI355X (gfx950) implementation of the
 * EMG->mel transduction TRAINING hot path of dgaddy/silent_speech (plus, per SURVEY section 8f, the CTC loss of the
 * recognition trainer and the input-conditioning kernel of the device-side pipeline).
 *
 * The reference is pure Python/PyTorch and has no FFI seam; its seam for this path is the set of
 * torch operator calls made by architecture.py / transformer.py / transduction_model.py / align.py /
 * data_utils.py.  Each entry point below replaces the reference call sites cited next to it
 * (file:line are relative to the reference repository root).  Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer (HBM) unless marked [host];
 *   - `stream` is a hipStream_t passed as void*; nothing here synchronises the device;
 *   - `dtype`: SS_F32 (0) or SS_BF16 (1) is the activation/compute type; statistics, losses,
 *     gradients of parameters and optimizer state are always f32;
 *   - return 0 on success; non-zero on error, with a message available from ss_last_error();
 *   - inputs are never modified unless documented (the reference's in-place EMG shift,
 *     architecture.py:67-68, is such a case and is mirrored by ss_emg_prepare's `shift`).
 * The Python binding a maintainer of the reference would add is a ctypes stub.
 */
#ifndef SILENT_SPEECH_HIP_H
#define SILENT_SPEECH_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SS_F32 0
#define SS_BF16 1
#define SS_F32X3 3   /* ss_gemm dtype_in only: f32 operands in memory, arithmetic on three bf16 MFMAs per product (operands split
                        hi + lo in registers, f32 accumulate: 16-17 significant bits per operand instead of 24 / 8) */

const char* ss_last_error(void);
/* Library/ABI version and the GPU architecture the kernels were compiled for ("gfx950"). */
/* Bumped whenever a struct layout or an entry-point signature changes (3: ss_gemm_epilogue column-statistics fields, the plan /
 * attention-image / LayerNorm-workspace entry points, ss_dtw_cumulative; 4: ss_loss_index_tables; 5: gate recomputation arguments of ss_bn_backward_sums / ss_bn_backward_apply; 7: ss_split_planes / ss_gemm_planes, ss_dw_job.flags, up to 24 jobs per grouped launch; 8: ss_stft_logmel_fft, the rejected-frame counter behind the matrix of ss_phoneme_confusion; 9: planes_hi / planes_lo / planes_only of ss_gemm_epilogue, dropout groups of the GEMM epilogue are row-major).  Bindings must compare it with SS_ABI_VERSION at load time: a stale
 * library paired with newer headers would otherwise read garbage struct fields instead of failing. */
#define SS_ABI_VERSION 9
int ss_abi_version(void);
const char* ss_target_arch(void);

/* ---------------------------------------------------------------------------------------------
 * Row addressing: logical row i starts at element
 *      base + (i / rows_per_batch) * batch_stride + (i % rows_per_batch) * row_stride
 * (rows_per_batch <= 0 means "one batch").  Expresses plain matrices, the zero-padded (B, T+2, C)
 * activation buffers, the overlapping 3C-wide im2col rows of the k=3 convolutions and strided
 * scatter of their input gradients. */
typedef struct ss_rowmap {
    int64_t base;
    int64_t batch_stride;
    int64_t row_stride;
    int32_t rows_per_batch;
} ss_rowmap;

typedef struct ss_gemm_epilogue {
    const float* bias;      /* [N] f32 or NULL                                                        */
    const void* gate;       /* out-typed tensor addressed like C or NULL: out = gate>0 ? out*gate_scale : 0
                               (ReLU + inverted-dropout backward from the SAVED activation)           */
    float gate_scale;
    float alpha;            /* scale applied to the accumulator first                                 */
    int32_t relu;           /* F.relu  (architecture.py:32; transformer.py:57)                        */
    float dropout_p;        /* nn.Dropout in training mode (transformer.py:57); counter-based hash RNG  */
    uint64_t seed;
    uint32_t rng_stream;    /* dropout site id; element (row, col) is element row * N + col of the stream (4 consecutive elements share a draw group) */
    int32_t mode;           /* 0 store, 1 C += v, 2 atomicAdd(C, v) (f32 out; required for split_k>1) */
    int32_t col_mod, col_mul, col_div_mul; /* optional output column permutation
                               col -> (col % col_mod)*col_mul + (col / col_mod)*col_div_mul           */
    float log_clamp;        /* > 0: v = log(max(v, log_clamp)) (data_utils.py:29-30 dynamic range compression) */
    void* c2;               /* optional second copy of the result, element (row, col) at
                               c2[rowmap2(row) + col*col_stride2] -- the [b][col][t] transposed copies
                               the attention kernels read as MFMA B operands                          */
    ss_rowmap cmap2;
    int64_t col_stride2;
    /* optional per-column statistics of the STORED result (f32, accumulated with atomics; the caller zeroes them):
       col_sum[n] += sum_m (C[m][n] - shift[n]),  col_sumsq[n] += sum_m (C[m][n] - shift[n])^2   (shift NULL = 0, col_sumsq may be NULL).
       Serves nn.Linear bias gradients (column sums of dY) and the BatchNorm batch statistics of a convolution output
       (architecture.py:19,21,25) without a separate pass over the tensor.  Only the 8-wave kernel computes them: ask
       ss_gemm_fuses_column_stats() first and fall back to ss_colsum / ss_bn_stats_sums when it answers 0. */
    float* col_sum;
    float* col_sumsq;
    const float* col_shift;
    /* ss_gemm_planes only (ABI 9): the stored result ALSO leaves as hi / lo bf16 planes -- hi = bf16(v), lo = bf16(v - hi), the arithmetic of
       ss_split_planes on the f32 value the kernel stores -- addressed like C (same row map, 2-byte elements), so that a consumer that is a plane
       GEMM / plane attention needs no split pass over C.  planes_only != 0: C itself is not written (C is still the address the row map refers to
       and must be non-NULL).  Column statistics, gate and accumulation apply before the split, as they do before the store.  NULL = off. */
    void* planes_hi;
    void* planes_lo;
    int32_t planes_only;
    /* The sign of a stored activation as ONE BIT per element (ABI 9), for the backward of ReLU (+ inverted dropout) -- transformer.py:57 --:
       sign_out (producer, bf16 results): byte [row * sign_pitch + col / 8], bit col % 8 = [stored C(row, col) > 0], rows = the GEMM's logical rows.
       gate_bits (consumer): out = bit ? out * gate_scale : 0 -- what `gate` does from the saved tensor, from 1/16 of its bytes; a thread requests the bits of
       its whole tile at once, so the 18 exposed load round trips per tile of the tensor form shrink to one.  N % 8 == 0.  Only some kernels of the 8-wave
       family carry these paths: ask ss_gemm_sign_bits_supported() (producer) / ss_gemm_fuses_column_stats() with gate_bits set (consumer: it rides the
       column-sum epilogue) and keep `gate` otherwise; ss_gemm refuses a launch that would ignore them. */
    void* sign_out;
    int64_t sign_pitch;
    const void* gate_bits;
    int64_t gate_bits_pitch;
} ss_gemm_epilogue;

#define SS_OP_KC 0   /* reduction index contiguous:  elem(o, r) = p[rowmap(o) + r] */
#define SS_OP_OC 1   /* outer index contiguous:      elem(o, r) = p[rowmap(r) + o] */

/* C[m][n] (+)= alpha * sum_k A(m,k) * B(n,k) on the MFMA units, f32 accumulate.
 * Replaces: nn.Conv1d k=3/k=1 (architecture.py:18,20,24 -> F.conv1d) as implicit GEMM over padded
 * activation buffers; nn.Linear (architecture.py:51,55,59; transformer.py:32,34); the per-head
 * einsum projections 'tbf,hfa->bhta' / 'bhta,haf->tbf' (transformer.py:96-98,111); and the autograd
 * backward of all of them (transduction_model.py:209): dX = dY.W (b_mode = OC), dW = dY^T.X
 * (a_mode = b_mode = OC, split_k > 1 with atomic f32 accumulation).
 * dtype_in: SS_BF16 (bf16 MFMA), SS_F32 (exact f32 MFMA 16x16x4) or SS_F32X3 (f32 operands, bf16 x 3 MFMA: ~5x the f32 rate at
 * ~1e-5 relative error per product, the "parity-grade fast mode" of the training plan, ss_plan_set_option 5). */
int ss_gemm(int dtype_in, int dtype_out, int a_mode, int b_mode, const void* A, const void* B, void* C,
            int M, int N, int K, const ss_rowmap* amap, const ss_rowmap* bmap, const ss_rowmap* cmap,
            const ss_gemm_epilogue* epilogue, int split_k, void* stream);

/* f32 operands as two bf16 planes (round 6: the parity-grade arithmetic of SS_F32X3 on the 8-wave bf16 kernel).
 * ss_split_planes: hi[i] = bf16(x[i]), lo[i] = bf16(x[i] - hi[i])  (x = hi + lo to 2^-17 relative); an elementwise pass, 16-byte aligned buffers.
 * ss_gemm_planes:  C[m][n] = epilogue(sum_k A(m,k) B(n,k)) with A = A_hi + A_lo, B = B_hi + B_lo, both operands K-contiguous and both planes
 *   of an operand addressed by the SAME row map, computed as ONE bf16 contraction of length 3 K,  [A_lo | A_hi | A_hi] . [B_hi | B_lo | B_hi]^T
 *   (three bf16 MFMAs per product, f32 accumulate; the a_lo.b_lo term, 2^-18 of the product, is dropped -- the arithmetic of SS_F32X3).
 *   f32 output only; runs on the 8-wave kernel or not at all: ss_gemm_planes_supported() answers 1 when it does (K % 64 == 0, 16-byte
 *   rows, no transposed second output, no log-clamp), and the call fails otherwise (the caller keeps ss_gemm(SS_F32X3) for such shapes).
 * Replaces, in the f32-storage plan: the f32 matmuls of architecture.py:18-24,51,55,59 and transformer.py:32,34,96-98,111 and their
 * input gradients.  Weight gradients of that mode: ss_gemm_dw_grouped with three jobs per gradient (hi.hi, hi.lo, lo.hi; flags bit 0). */
int ss_split_planes(const float* x, void* hi, void* lo, int64_t n, void* stream);
int ss_gemm_planes(int dtype_out, const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, void* C, int M, int N, int K,
                   const ss_rowmap* amap, const ss_rowmap* bmap, const ss_rowmap* cmap, const ss_gemm_epilogue* epilogue, void* stream);
/* [host] 1 when ss_gemm with these arguments (epilogue->sign_out set or not) runs on the kernel that writes sign bits (8-wave kernel, bf16 results, no column
 * statistics, default schedule, N % 8 == 0). */
int ss_gemm_sign_bits_supported(int dtype_in, int dtype_out, int a_mode, int b_mode, const void* C, int M, int N, int K, const ss_rowmap* amap,
                                const ss_rowmap* bmap, const ss_rowmap* cmap, const ss_gemm_epilogue* epilogue, int split_k);
int ss_gemm_planes_supported(int dtype_out, const void* C, int M, int N, int K, const ss_rowmap* amap, const ss_rowmap* bmap, const ss_rowmap* cmap,
                             const ss_gemm_epilogue* epilogue); /* [host] */

/* Persistent-grid size of ss_gemm for the calling thread: 2 (default) or 1 workgroup per CU.  The weight-gradient GEMMs that
 * run on a side stream use 1 so that the dependent chain on the main stream can co-reside on every CU.  Returns the old value. */
int ss_gemm_set_blocks_per_cu(int n); /* [host] */
/* Which kernel the calling thread's last ss_gemm launch used: 0 register-staged gemm_kernel, 1 gemm_glds_kernel,
 * 2 gemm_w2_kernel, 3 / 4 gemm8_kc_kernel with 256 / 288-row tiles, 5 gemm_smallk_kernel (K <= 32, no LDS) (so that a profiler can attribute per-launch timings to
 * the kernel names rocprofv3 reports, and tests can assert which variant they exercised). */
int ss_gemm_last_kernel(void); /* [host] */
/* 1 if ss_gemm with these arguments runs a kernel that honours epilogue.col_sum / col_sumsq (the 8-wave kernel, the K <= 32 kernel); 0 otherwise (the call then fails
 * with those fields set).  Same decision procedure as ss_gemm itself (legality + cost model + knobs). */
int ss_gemm_fuses_column_stats(int dtype_in, int dtype_out, int a_mode, int b_mode, const void* C, int M, int N, int K, const ss_rowmap* amap,
                               const ss_rowmap* bmap, const ss_rowmap* cmap, const ss_gemm_epilogue* epilogue, int split_k); /* [host] */
/* Kernel-selection knobs of ss_gemm (process-wide): what = 0 two-wave kernel (0 never / 1 cost model / 2 whenever legal),
 * 1 its tile height (128 / 144, 0 = cost model), 2 eight-wave kernel (0 / 1 / 2 as above), 3 its tile height in 16-row units
 * per M-wave (8 / 9, 0 = cost model), 4 its LDS reads / DMA pieces spread between MFMA groups (0 / 1, 2 = per tile height), 5 ablation mask for kernel tuning (results are
 * wrong when non-zero), 6 the LDS-free K <= 32 kernel on / off (SS_GEMM_SMALLK).  value < 0 restores the default (environment SS_GEMM_W2, SS_GEMM_W2_BM, SS_GEMM8, SS_GEMM8_NI, SS_GEMM8_PIN,
 * SS_GEMM_DEBUG).  Returns the previous value, -1 for a bad `what`. */
int ss_gemm_set_option(int what, int value); /* [host] */

/* Grouped weight-gradient GEMMs: for every job  C[m][n] += sum_k A(k, m) * B(k, n)  (bf16 operands, f32 C, reduction over the
 * K = B*T frame rows; both operands outer-contiguous: element (k, m) of A at A[amap(k) + m]).  Replaces the autograd backward
 * of nn.Linear / nn.Conv1d / the per-head einsum projections w.r.t. their weights (transduction_model.py:209 through
 * architecture.py:18-24,51 and transformer.py:32,34,96-98,111): dW = dY^T X.  ONE persistent launch covers up to 24 jobs (e.g.
 * the four weight gradients of an encoder layer), so the K split that fills the 256 CUs -- and with it the number of f32
 * atomic accumulations -- is chosen for the group, not per GEMM.  M, N multiples of 8; amap / bmap must cut the rows into
 * batches of equal length (rows_per_batch). */
typedef struct ss_dw_job {
    const void* A;          /* dY: K rows of M contiguous bf16 */
    const void* B;          /* X : K rows of N contiguous bf16 */
    float* C;               /* M x N f32, row stride ldc; accumulated into */
    ss_rowmap amap, bmap;
    int64_t ldc;
    int32_t M, N, K;
    int32_t flags;          /* bit 0: other jobs of this launch accumulate into the same C (always atomic updates) */
} ss_dw_job;
int ss_gemm_dw_grouped(int n_jobs, const ss_dw_job* jobs /* [host] */, void* stream);
/* Tuning knobs of ss_gemm_dw_grouped for the calling thread: what = 0 K split override (0 = automatic), 1 scheduling fences,
 * 2 XCD-contiguous item order on / off (default on: neighbouring tiles of one problem share an L2). */
int ss_gemm_dw_set_option(int what, int value); /* [host] */

/* out[a][b][c] (contiguous, dims d0 x d1 x d2) (+)= scale * in[a*s0 + b*s1 + c*s2] for b < valid1 and
 * c < valid2, else 0 (zero padding).  Converts between the reference's parameter layouts (state_dict:
 * conv (O,I,k) architecture.py:18; w_q/w_k/w_v (H,d,dh), w_o (H,dh,d) transformer.py:71-74; embeddings
 * (H,2D-1,dh,1) transformer.py:159) and the GEMM-ready, head-dim-padded layouts, in both directions
 * (weights forward, gradients back).  in/out dtype: SS_F32 or SS_BF16. */
int ss_permute3d(const void* in, int in_dtype, void* out, int out_dtype, int d0, int d1, int d2,
                 int64_t s0, int64_t s1, int64_t s2, int valid1, int valid2, float scale, int accumulate,
                 void* stream);

/* The same for a whole table of jobs in one launch (per-step weight re-layout, gradient un-layout).
 * jobs_dev: device array of
 *   struct { const void* in; void* out; int64 s0,s1,s2 (input strides), o0,o1 (output strides of dims 0,1; dim 2 contiguous);
 *            int32 d0,d1,d2, valid1,valid2, in_dtype,out_dtype, accumulate; float scale; int32 first_block, nblocks, pad; }
 * job_of_block_dev[b] = job index of workgroup b (workgroups first_block .. first_block+nblocks-1 grid-stride over the job).
 * all_f32 != 0 promises that every job is f32 -> f32 (the gradient un-layout): a kernel specialised for read-modify-write runs. */
int ss_permute3d_batch(const void* jobs_dev, const int32_t* job_of_block_dev, int total_blocks, int all_f32, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DTW alignment (align.py:5-14 time_warp + align.py:16-34 align_from_distances; call site
 * transduction_model.py:126,131 and :88).  Batched: one workgroup per matrix.
 * desc_dev: device array [n][10] of int64:
 *   {N, M, cost_off (elements from `costs`), stride_i, stride_j (elements), sk_off, dirs_off, bnd_off
 *    (BYTE offsets into `workspace`, sizes from ss_dtw_workspace_bytes), res_off (elements into
 *    `results`), 0}.
 * results[res_off + i], i < N: the reference's `results` list (smallest j visited in row i; 0 for
 * row 0 and for degenerate 1xM / Nx1 inputs).  Bit-exact: f32, one add per cell, first-minimum tie
 * order (up, left, diag).  The cumulative matrix itself is never written (2-bit directions are).
 * Where the recurrence takes its costs from is decided per matrix (ss_dtw_source): a matrix with one unit-stride axis of at most 1025
 * cells is read IN PLACE (2: row-major -- the lanes own columns, the kernel sweeps the transposed problem; 1: column-major, e.g. the
 * costs.T view of transduction_model.py:126 -- the lanes own rows); anything else (0: both strides > 1, a unit-stride axis longer than
 * one strip, fewer than 5 cells on it) is first copied into skewed strips inside `workspace`. */
int ss_dtw_source(int n, int m, int64_t stride_i, int64_t stride_j); /* [host] */
int64_t ss_dtw_workspace_bytes(int n, int m, int64_t* sk_bytes, int64_t* dirs_bytes, int64_t* bnd_bytes); /* [host] */
int ss_dtw_align(const float* costs, const int64_t* desc_dev, int n, int max_n, int max_m, void* workspace,
                 int32_t* results, void* stream);
/* Same, for costs already in the skewed strip layout (written by ss_silent_cost_skewed); results must
 * have been zeroed by the producer. */
int ss_dtw_align_skewed(const int64_t* desc_dev, int n, void* workspace, int32_t* results, void* stream);
/* align.py:5-14 `time_warp` itself: the dense cumulative matrix dtw_out[N][M] (contiguous) of ONE cost matrix whose element (i, j)
 * sits at costs + i*stride_i + j*stride_j (elements), in the dtype of the input (SS_F32 or SS_F64: `zeros_like(costs)`,
 * align.py:6); dtw_out[N-1][M-1] is the alignment cost.  results (optional, N ints): the reference's backtrace on that matrix
 * (align.py:16-34) -- this is how float64 input is aligned without rounding it to float32.  Not on the training path (one
 * workgroup, one barrier per anti-diagonal); batches of f32 matrices go through ss_dtw_align. */
int ss_dtw_cumulative(int dtype, const void* costs, int64_t stride_i, int64_t stride_j, int N, int M, void* dtw_out,
                      int32_t* results, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm1d fused with ReLU / the ResBlock residual add (architecture.py:19,21,25 nn.BatchNorm1d,
 * :32 relu(bn1(conv1)), :33 bn2(conv2), :36 res_norm(residual_path), :40 relu(x + res)).
 * Activations are (B, T + 2*pad, C), C contiguous, pad in {0,1}: padded buffers carry a zero row on
 * each side of every sequence (the conv halo); pad rows of outputs are written as zeros.
 * Training: per-channel batch mean / 1/sqrt(biased var + eps) over B*T, running stats updated in place
 * (momentum, unbiased variance) as torch does; eval: from the running stats.
 * scratch: ss_bn_scratch_floats(B,T,C) floats. */
int64_t ss_bn_scratch_floats(int B, int T, int C); /* [host] */
/* Two-phase statistics so that data-parallel ranks can all-reduce between the phases (the batch
 * statistics of the reference span the WHOLE batch):  sums[3][C] = { sum(x-s), sum (x-s)^2, s } with
 * s = shift (a [C] vector identical on all ranks, e.g. running_mean) or, if NULL, the first row. */
int ss_bn_stats_sums(int dtype, const void* x, int B, int T, int C, int pad, float* scratch, const float* shift, float* sums, void* stream);
int ss_bn_finalize(const float* sums, double n_total, int C, float* mean, float* invstd, float* running_mean, float* running_var,
                   float momentum, float eps, int training, void* stream);
/* The same with the shift vector passed separately (sums = [2][C]): for sums produced by a GEMM epilogue (ss_gemm_epilogue.col_sum /
 * col_sumsq with col_shift = shift, e.g. the running mean BEFORE this update; shift may alias running_mean). */
int ss_bn_finalize_shift(const float* sums, const float* shift, double n_total, int C, float* mean, float* invstd, float* running_mean,
                         float* running_var, float momentum, float eps, int training, void* stream);
/* y = act( bn_a(xa) [+ bn_b(xb)] ),  act = ReLU if relu */
int ss_bn_apply(int dtype, const void* xa, const float* mean_a, const float* invstd_a, const float* gamma_a, const float* beta_a, int pad_xa,
                const void* xb, const float* mean_b, const float* invstd_b, const float* gamma_b, const float* beta_b, int pad_xb,
                void* y, int pad_y, int B, int T, int C, int relu, void* stream);
/* autograd backward of the above (transduction_model.py:209), again in two phases:
 * sums[3][C] = { sum g, sum g*xhat_a, sum g*xhat_b } with g = dy*1[y>0]; dgamma/dbeta are ACCUMULATED (+=);
 * apply: dx = gamma*invstd*(g - sums0/n - xhat*sums{1,2}/n).
 * The ReLU gate 1[y>0] is read from the saved output y, or -- gate_beta_a != NULL -- RECOMPUTED as the sign of the forward's own
 * expression (xa - mean_a) gamma_a invstd_a + beta_a [+ the b branch] from the inputs both passes read anyway (y may then be NULL):
 * one tensor less per pass (the plan does this). */
int ss_bn_backward_sums(int dtype, const void* dy, int pad_dy, const void* y, int pad_y,
                        const void* xa, int pad_xa, const float* mean_a, const float* invstd_a,
                        const void* xb, int pad_xb, const float* mean_b, const float* invstd_b,
                        float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b,
                        float* scratch, float* sums, int B, int T, int C, int relu,
                        const float* gate_gamma_a, const float* gate_beta_a, const float* gate_gamma_b, const float* gate_beta_b, void* stream);
int ss_bn_backward_apply(int dtype, const void* dy, int pad_dy, const void* y, int pad_y,
                         const void* xa, int pad_xa, const float* mean_a, const float* invstd_a, const float* gamma_a,
                         const void* xb, int pad_xb, const float* mean_b, const float* invstd_b, const float* gamma_b,
                         const float* sums, double n_total, void* dxa, int pad_dxa, void* dxb, int pad_dxb,
                         int B, int T, int C, int relu, const float* gate_beta_a, const float* gate_beta_b, void* stream);
/* out[c] += sum_r x[r][c]  (bias gradients of nn.Linear / nn.Conv1d) */
int64_t ss_colsum_scratch_floats(int rows, int C); /* [host] */
int ss_colsum(int dtype, const void* x, int rows, int C, int64_t ld, float* scratch, float* out_accum, void* stream);

/* Post-norm residual block of the encoder layer (transformer.py:55-56, 58-59):
 *   z = x + dropout(branch);  y = LayerNorm(z) * gamma + beta        (eps 1e-5)
 * branch_inout is overwritten with z (saved for backward); mean/rstd: [rows] f32. */
int ss_add_dropout_layernorm_forward(int dtype, const void* x, void* branch_inout, const float* gamma, const float* beta, void* y,
                                     float* mean, float* rstd, int rows, int C, float eps, float dropout_p, uint64_t seed,
                                     uint32_t rng_stream, void* stream);
/* dres = dL/dz (gradient of the residual stream; may alias dy), dbranch = dres * keep/(1-p) (may be NULL);
 * dgamma/dbeta accumulated. */
int ss_layernorm_backward(int dtype, const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                          void* dres, void* dbranch, float* dgamma, float* dbeta, int rows, int C, float dropout_p, uint64_t seed,
                          uint32_t rng_stream, void* stream);

/* The same, also accumulating dbranch_colsum[c] += sum_rows dbranch[row][c] (as stored, i.e. rounded to dtype): the bias gradient of
 * the nn.Linear that produced the branch (linear2 of transformer.py:58), without a separate pass over dbranch. */
int ss_layernorm_backward_bias(int dtype, const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                               void* dres, void* dbranch, float* dgamma, float* dbeta, float* dbranch_colsum, int rows, int C, float dropout_p,
                               uint64_t seed, uint32_t rng_stream, void* stream);
/* The 16-waves-per-CU form (C = 256, 512 or 768; dbranch required): selected by passing a scratch buffer of
 * ss_layernorm_backward_scratch_floats(rows, C) floats (0 for other widths; scratch == NULL runs the round-1 form).  The column sums leave
 * as one atomic per column and workgroup (#CU workgroups); with SS_LN_DIRECT=0 they go through the scratch buffer and a second small kernel
 * instead (deterministic summation order). */
int64_t ss_layernorm_backward_scratch_floats(int rows, int C); /* [host] */
int ss_layernorm_backward_ws(int dtype, const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                             void* dres, void* dbranch, float* dgamma, float* dbeta, float* dbranch_colsum, float* scratch, int64_t scratch_floats,
                             int rows, int C, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream);
/* The parity-grade mode (f32 data): the LayerNorm output / the branch gradient ALSO leave as hi / lo bf16 planes (ss_split_planes' arithmetic on the stored f32
 * value), so that the plane GEMMs that consume them need no split pass (ABI 9).  NULL planes = the plain forms above. */
int ss_add_dropout_layernorm_forward_planes(int dtype, const void* x, void* branch_inout, const float* gamma, const float* beta, void* y, void* y_hi, void* y_lo,
                                            float* mean, float* rstd, int rows, int C, float eps, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream);
int ss_layernorm_backward_ws_planes(int dtype, const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                                    void* dres, void* dbranch, void* dbranch_hi, void* dbranch_lo, float* dgamma, float* dbeta, float* dbranch_colsum, float* scratch,
                                    int64_t scratch_floats, int rows, int C, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream);

/* EMG input conditioning: training-time shift augmentation (architecture.py:64-68: x[:, :-r] = x[:, r:],
 * x[:, -r:] = 0), cast to the compute dtype and zero halo rows: x_raw (B,T0,Cin) f32 ->
 * out_padded (B,T0+2,Cin).  shifted_copy (optional, f32 (B,T0,Cin)) receives the shifted signal so the
 * caller can mirror the reference's in-place mutation of its input. */
int ss_emg_prepare(int dtype, const float* x_raw, void* out_padded, float* shifted_copy, int B, int T0, int Cin, int shift, void* stream);

/* ---------------------------------------------------------------------------------------------
 * dtw_loss (transduction_model.py:98-157).  `head` = [pred (n_mel) | phoneme logits (n_phone)] per
 * packed frame, f32, row stride ld.  All kernels produce value AND gradient (dhead, same layout,
 * must be zero-initialised; scaled by inv_total = 1/sum(T2)); loss_accum/correct_accum are += . */
/* Per-frame index tables of one batch (the decollate_tensor + zip bookkeeping of transduction_model.py:101-111) expanded on the
 * device from one row of 8 int64 per utterance: [n_pred, n_tgt, silent, first packed pred row, first target row, offset into
 * `results`, offset into the voiced tables, offset into the silent tables].  Voiced utterance: vo_pred / vo_tgt [vo_off + i] =
 * pred_row0 + i / tgt_row0 + i; silent: si_tgt = tgt_row0 + i, si_base = pred_row0, si_res = res_off + i (i < n_tgt).  The tables
 * are what ss_voiced_loss / ss_silent_loss take; they may be sized exactly (sum of voiced n_pred / silent n_tgt, at least 1). */
int ss_loss_index_tables(const int64_t* utt_dev, int n_utt, int32_t* vo_pred, int32_t* vo_tgt, int32_t* si_tgt, int32_t* si_base,
                         int32_t* si_res, void* stream);
int ss_frame_lse(const float* head, int64_t ld, int col0, int ncls, int rows, float* lse, int32_t* argmax, void* stream);
int ss_voiced_loss(const float* head, int64_t ld, int n_mel, int n_phone, const float* lse, const int32_t* argmax, const float* Y,
                   const int64_t* phones, const int32_t* pred_row, const int32_t* tgt_row, int nframes, float lam, float inv_total,
                   float* dhead, float* loss_accum, int32_t* correct_accum, void* stream);
/* Cost matrices of the silent utterances written directly in the DTW strip layout (never dense);
 * desc as for ss_dtw_align with fields [2],[3] = first packed pred row, first target row. */
int ss_silent_cost_skewed(const float* head, int64_t ld, int n_mel, const float* lse, const float* Y, const int64_t* phones,
                          const int64_t* desc_dev, int n, int max_n, int max_m, float lam, void* workspace, int32_t* results, void* stream);
int ss_silent_loss(const float* head, int64_t ld, int n_mel, int n_phone, const float* lse, const int32_t* argmax, const float* Y,
                   const int64_t* phones, const int32_t* results, const int32_t* tgt_row, const int32_t* pred_base,
                   const int32_t* res_idx, int nframes, float lam, float inv_total, float* dhead, float* loss_accum,
                   int32_t* correct_accum, void* stream);
/* Phoneme confusion matrix of the evaluation path (transduction_model.py:130-137 silent, :147-152 voiced), accumulated ON the device:
 * confusion[pred * n_phone + target] += 1 for every target frame, pred = argmax (ss_frame_lse) of the aligned prediction row -- through
 * `results` (ss_dtw_align_skewed) for silent utterances.  Index tables as produced by ss_loss_index_tables; int32 matrix, += .
 * `confusion` holds n_phone * n_phone + 1 ints: the last one counts the frames whose prediction or target label lies outside [0, n_phone) (the
 * reference's numpy indexing raises IndexError for those, :134-137); they are left out of the matrix and the caller decides (DeviceConfusion raises). */
int ss_phoneme_confusion(const int32_t* argmax, const int64_t* phones, const int32_t* results, const int32_t* vo_pred, const int32_t* vo_tgt,
                         int n_voiced, const int32_t* si_tgt, const int32_t* si_base, const int32_t* si_res, int n_silent_frames,
                         int32_t* confusion, int n_phone, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CTC loss of the recognition trainer ("next" row N1): replaces F.log_softmax + pad_sequence(decollate_tensor(...))
 * + F.ctc_loss(..., blank, reduction='mean') of recognition_model.py:96-101 on the PACKED (rows, ld) f32 logits.
 * lse = per-frame log-sum-exp over the V classes (ss_frame_lse with col0=0).  desc (device, int64 x5 per utterance,
 * sorted by frame0): first packed frame, T, first target index, S, offset (floats) of the utterance's T*(2S+1)
 * block in alpha_ws / beta_ws.  Writes nll[n_utt], loss[0] = mean_u nll_u / max(S_u, 1) and the complete
 * d loss / d logits (all `rows` x ld entries; zero outside the utterances). */
int ss_ctc_loss(const float* logits, int64_t ld, int V, int blank, const float* lse, const int64_t* desc_dev, int n_utt, int max_target_len,
                int64_t rows, const int32_t* targets, float* alpha_ws, float* beta_ws, float* nll, float* dlogits, float* loss, void* stream);

/* AdamW over a flat f32 arena (transduction_model.py:178,210).  step is 1-based. */
int ss_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, float grad_scale, void* stream);
int ss_cast_f32(const float* in, void* out, int out_dtype, int64_t n, void* stream);
/* Input conditioning of the dataset on the device ("next" row N3): out = limit * tanh(((x - mean[c]) / std[c]) / pre_div / limit)
 * with mean/std optional (both NULL = no affine; C = row length they index) and limit <= 0 = no clipping.
 * raw EMG: pre_div 20, limit 50 (read_emg.py:227-228); EMG features: FeatureNormalizer + limit 8 (read_emg.py:231-233);
 * mel targets: FeatureNormalizer only (read_emg.py:231).  In place is allowed. */
int ss_soft_clip(const float* x, float* out, int64_t n, int C, const float* mean, const float* stdv, float pre_div, float limit, void* stream);

/* combine_fixed_length (data_utils.py:158-167; call sites transduction_model.py:200-202) as ONE gather: concatenates n device
 * blobs and zero-fills the tail of `out` (total_bytes).  table_dev = n source pointers followed by n + 1 cumulative byte offsets
 * (int64 each, device memory); granule (16, 8, 4 or 1) must divide every offset, pointer and total_bytes. */
int ss_concat_pad(const void* table_dev, int n, void* out, int64_t total_bytes, int granule, void* stream);

/* mel target extraction (data_utils.py:39-62): reflect pad (:51); the STFT itself is ss_gemm against a
 * windowed DFT matrix with overlapping hop-strided rows; magnitude (:57); mel matmul + log clamp (:59-60)
 * is ss_gemm with epilogue.log_clamp. */
int ss_reflect_pad(const float* y, float* out, int B, int L, int pad, int64_t ld_out, void* stream);
/* Ragged batch (data_utils.py:76 np.clip + :51 reflect pad for EVERY utterance of a batch in one launch): utterance b is
 * y[offsets_dev[b] .. + lengths_dev[b]); row b of out[B][ld_out] gets its clipped (clip != 0), reflect-padded signal and zeros
 * behind it.  min_len = the shortest length (host-known; reflect needs pad < length). */
int ss_reflect_pad_ragged(const float* y, const int64_t* offsets_dev, const int32_t* lengths_dev, float* out, int B, int min_len,
                          int pad, int64_t ld_out, int clip, void* stream);
int ss_stft_magnitude(const float* spec, int64_t ld_spec, int n_bins, float* mag, int64_t ld_mag, int rows, void* stream);
/* The whole of data_utils.py:51-60 (and :76's np.clip when clip != 0) for n_fft = 1024 as ONE kernel (csrc/mel.hip: an LDS radix-8 FFT per frame, one
 * wave per frame), reading the caller's signals in place: signal b is y[offsets_dev[b] .. + lengths_dev[b]) (a ragged batch) or, with both arrays NULL,
 * row b of a [B][uniform_len] matrix.  Frame f (< F) of a signal covers its reflect-padded samples f * hop - pad .. + 1024 (F.pad(..., 'reflect'), :51;
 * pad < length; samples behind the padded signal count as 0, as in the zero-filled rows of ss_reflect_pad_ragged), windowed by window[1024]
 * (torch.hann_window, :49), one-sided spectrum, sqrt(re^2 + im^2 + 1e-9) (:57), the mel filterbank (:59) in SPARSE form -- band m = sum_j
 * band_w[band_off[m] + j] * mag[band_lo[m] + j], j < band_cnt[m]: the non-zero run of row m of librosa.filters.mel, padded with zero weights to a
 * multiple of 4 (band_lo[m] + band_cnt[m] <= 520), n_w packed weights in all; lane_bands[2][64] deals the bands to the 64 lanes (-1 = none) --,
 * log(clamp(., log_clamp)) (:60, spectral_normalize_torch) written to out[b * stride_b + f * stride_f + m * stride_m].  n_mels <= 128, n_w <= 4096. */
int ss_stft_logmel_fft(const float* y, const int64_t* offsets_dev, const int32_t* lengths_dev, int64_t uniform_len, int B, int F, int pad, int clip,
                       int n_fft, int hop, const float* window,
                       const int32_t* band_lo, const int32_t* band_cnt, const int32_t* band_off, const float* band_w, const int32_t* lane_bands, int n_mels, int n_w,
                       float log_clamp, float* out, int64_t stride_b, int64_t stride_f, int64_t stride_m, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head self-attention with learned relative-position logits (transformer.py:87-112 and
 * :162-297; closed form in csrc/attention.hip), fused: QK^T, Q.E^T (banded, |k-q| <= D-1, -1e8
 * outside), softmax, dropout (transformer.py:109), P.V.  Operands (compute dtype):
 *   qkv  [B*T][3*H*dp] (q|k|v, head, d) with the head dim zero-padded to dp (multiple of 32, <=128);
 *   qkvT [B][3*H*dp][Tp] the same transposed per sequence (2nd output of the QKV ss_gemm);
 *   E    [H][2D-1][dp];  ET [H][dp][MPt], MPt = roundup(2D-1, 32)  (ss_permute3d of the embeddings);
 *   out  [B*T][H*dp];  lse [B][H][T] f32 (saved for backward).  scale = 1/sqrt(d_qkv) (unpadded).
 * backward (transduction_model.py:209): dqkv [B*T][3*H*dp] = (dQ|dK|dV); dO/dOT like out / its
 * transposed copy [B][H*dp][Tp]; Dscratch [B][H][T] f32.  The embeddings receive no gradient
 * (transformer.py:214-218). */
/* 1 if (dtype, T, dp, D) runs the per-tile attention kernels, which read the per-sequence transposed copies qkvT / dOT;
 * 0 if the LDS-resident kernels run (bf16, T <= 208, operands fit the 160 KB LDS): then qkvT and dOT may be NULL and the
 * producing GEMMs need not emit them. */
int ss_relpos_attention_needs_transposed(int dtype, int T, int dp, int D); /* [host] */
/* Which kernels (dtype, T, dp, D) runs: 0 = per-tile (qkvT / dOT needed), 1 = LDS-resident 16 x 16 tiles, 2 = transposed 32 x 32 score
 * tiles (bf16, T <= 224: csrc/attention_t.hip).  Family 2 reads the embeddings from a prepared table instead of E / ET:
 * ss_relpos_attention_table_bytes() bytes, filled by ss_relpos_attention_prepare_tables from the f32 parameter
 * (transformer.py:172-176 `embeddings`, [H][2D-1][dh] contiguous; scale = 1/sqrt(d_qkv) as in the calls below) -- E / scale in
 * MFMA-fragment order, so that Q.E accumulates in the same accumulator as Q.K and the table streams from L2 as whole KiB.
 * `tab` may be NULL for families 0 and 1; E / ET / qkvT / dOT may be NULL for family 2. */
int ss_relpos_attention_family(int dtype, int T, int dp, int D); /* [host] */
int64_t ss_relpos_attention_table_bytes(int H, int dp, int D); /* [host] */
int ss_relpos_attention_prepare_tables(const float* emb, void* tab, int H, int D, int dh, int dp, float scale, void* stream);
int ss_relpos_attention_forward(int dtype, const void* qkv, const void* qkvT, const void* E, const void* tab, void* out, float* lse,
                                int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed,
                                uint32_t rng_stream, void* stream);
int ss_relpos_attention_backward(int dtype, const void* qkv, const void* qkvT, const void* E, const void* ET, const void* tab, const void* out,
                                 const float* lse, const void* dO, const void* dOT, float* Dscratch, void* dqkv,
                                 int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed,
                                 uint32_t rng_stream, void* stream);

/* The same with SAVED PROBABILITIES: the LDS-resident forward can leave its normalised probabilities (bf16, accumulator layout,
 * dropout decision in the sign bit; csrc/attention.hip "the P image") in `pimg`, ss_relpos_attention_saved_bytes() bytes; the
 * backward then reads them instead of recomputing both logit products, the skew, the exponentials and the dropout draws.
 * saved_bytes is 0 for shapes that run the per-tile kernels; pimg may be NULL in both calls (= the functions above), except that the
 * family-2 backward works ONLY from the saved probabilities (its forward may still run without pimg: inference). */
int64_t ss_relpos_attention_saved_bytes(int dtype, int B, int H, int T, int dp, int D); /* [host] */
int ss_relpos_attention_forward_p(int dtype, const void* qkv, const void* qkvT, const void* E, const void* tab, void* out, float* lse, void* pimg,
                                  int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed,
                                  uint32_t rng_stream, void* stream);
int ss_relpos_attention_backward_p(int dtype, const void* qkv, const void* qkvT, const void* E, const void* ET, const void* tab, const void* out,
                                   const float* lse, const void* dO, const void* dOT, float* Dscratch, void* dqkv, const void* pimg,
                                   int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed,
                                   uint32_t rng_stream, void* stream);

/* The parity-grade mode of the attention (round 6): f32 operands as hi / lo bf16 planes (ss_split_planes), every product of the kernels above
 * on three bf16 MFMAs (a_lo.b_hi + a_hi.b_lo + a_hi.b_hi, f32 accumulate) -- the transposed-score kernels' formulation, so bf16 rows of up to
 * 224 frames, d_head <= 96 (ss_relpos_attention_x3_supported; other shapes keep SS_F32X3 of the entry points above).  qkv / dO arrive as
 * plane pairs, out / dqkv LEAVE as plane pairs (their consumers are ss_gemm_planes / the grouped weight-gradient jobs); the table holds E / scale
 * as [hi | lo] (ss_relpos_attention_x3_prepare_tables, ss_relpos_attention_x3_table_bytes), the saved probabilities two images [hi | lo]
 * (ss_relpos_attention_x3_saved_bytes; required by the backward).  Dropout masks are those of kernel family 2.
 * Replaces the f32 arithmetic of transformer.py:87-112, :229-297 and its autograd backward. */
int ss_relpos_attention_x3_supported(int T, int dp, int D);                               /* [host] */
int64_t ss_relpos_attention_x3_saved_bytes(int B, int H, int T, int dp, int D);           /* [host] */
int64_t ss_relpos_attention_x3_table_bytes(int H, int dp, int D);                         /* [host] */
int ss_relpos_attention_x3_prepare_tables(const float* emb, void* tab, int H, int D, int dh, int dp, float scale, void* stream);
int ss_relpos_attention_x3_forward(const void* qkv_hi, const void* qkv_lo, const void* tab, void* out_hi, void* out_lo, float* lse, void* pimg,
                                   int B, int H, int T, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream);
int ss_relpos_attention_x3_backward(const void* qkv_hi, const void* qkv_lo, const void* tab, const void* out_hi, const void* out_lo, const void* dO_hi, const void* dO_lo,
                                    float* Dscratch, void* dqkv_hi, void* dqkv_lo, const void* pimg,
                                    int B, int H, int T, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Offline EMG conditioning ("next" row N4): the zero-phase IIR cascade of read_emg.py:27-38 (7 x filtfilt(iirnotch(60 h, 30)) then
 * filtfilt(butter(3, 2 Hz, 'highpass')), scipy.signal.filtfilt defaults: odd extension of 3 max(len a, len b) samples, lfilter_zi
 * initial conditions) and the np.interp resampling of read_emg.py:40-44, in f64 like numpy.  x, y: (T, C) f64, time-major.
 * coef [host]: per filter 13 doubles { b[4], a[4] (a[0] = 1, zero padded), zi[3] (scipy.signal.lfilter_zi), order n, padlen }. */
int64_t ss_iir_filtfilt_workspace_bytes(int T, int C, int max_padlen); /* [host] */
int ss_iir_filtfilt(const double* x, double* y, int T, int C, int n_filt, const double* coef, void* workspace, int64_t workspace_bytes, void* stream);
/* y[i] = np.interp(i / new_freq, arange(T) / old_freq, x[:, c]) for i < T_out (the caller sizes T_out = len(arange(0, (T-1)/old_freq, 1/new_freq))) */
int ss_linear_resample(const double* x, double* y, int T, int C, double old_freq, double new_freq, int T_out, void* stream);
/* The same for a RAGGED BATCH of recordings (one launch sequence for all of them): x / y are the recordings back to back, packed
 * (sum T_u, C) f64; lengths_host = the R lengths (host memory, like coef).  Work items are (chunk, channel) pairs over all recordings. */
int64_t ss_iir_filtfilt_batch_workspace_bytes(const int32_t* lengths_host, int R, int C, int max_padlen, int n_filt); /* [host] */
int ss_iir_filtfilt_batch(const double* x, double* y, const int32_t* lengths_host, int R, int C, int n_filt, const double* coef,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* np.interp of every recording onto its own grid; table_dev: int64 [R][4] = {first input row, T, first output row, T_out}. */
int ss_linear_resample_batch(const double* x, double* y, const int64_t* table_dev, int R, int C, double old_freq, double new_freq,
                             int64_t total_out_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The execution plan of the transduction model as native code: ONE call enqueues the whole forward pass of Model.forward
 * (architecture.py:61-84: shift augmentation, 3 ResBlocks :29-40, w_raw_in, the post-norm relative-position encoder layers
 * transformer.py:43-60,87-112, both heads) and ONE the whole backward pass that loss.backward() (transduction_model.py:209)
 * triggers through autograd -- ~450 kernel launches per training step without a host-language round trip per kernel.
 * The plan owns no memory: device pointers (parameters, the GEMM-ready weight copies, gradient buffers, the job tables of the
 * gradient un-layout launches) are bound to NAMED slots (ss_plan_slot_name lists them; names follow the reference's
 * state_dict keys), activations live in a caller-provided workspace, the saved-for-backward pointers in a caller-kept host
 * struct of ss_plan_ctx_bytes() bytes.  dtype: SS_BF16 (MFMA bf16, f32 accumulate) or SS_F32 (exact f32 MFMA). */
typedef struct ss_model_dims {
    int32_t d_model, n_layers, n_head, d_qkv, dp /* head dim padded to 32 */, max_rel /* relative_positional_distance */, ff;
    int32_t n_head_cols /* num_outs + num_aux_outs rounded up to 8: the fused output heads */, dtype;
    float ln_eps;
} ss_model_dims;
typedef struct ss_plan ss_plan;
/* Data-parallel hook: called between the two phases of every training-mode BatchNorm (forward: n_floats = 2C sums, backward:
 * 3C) with the device pointer of the per-channel sums and the local row count; must all-reduce the sums in place on `stream`
 * and return the GLOBAL row count. */
typedef double (*ss_reduce_hook)(void* user, float* sums_dev, int n_floats, double n_local, void* stream);
/* Called from ss_plan_backward when a group of parameter gradients is final on `stream`: what = 4 + l: encoder layer l (fired
 * layer by layer, last layer first, while the backward of the earlier layers still runs), 0: the fused heads + w_raw_in (after it every
 * non-convolutional gradient is final), 1..3: ResBlock 2, 1, 0.  Lets the caller start a bucketed gradient all-reduce while the
 * rest of backward still runs. */
typedef void (*ss_event_hook)(void* user, int what, void* stream);
ss_plan* ss_plan_create(const ss_model_dims* dims);                 /* [host] NULL on error */
void ss_plan_destroy(ss_plan* plan);                                /* [host] */
int ss_plan_slot_count(const ss_plan* plan);                        /* [host] */
const char* ss_plan_slot_name(const ss_plan* plan, int slot);       /* [host] */
int ss_plan_bind(ss_plan* plan, int slot, void* device_ptr_or_value); /* [host] slots named *.total / *.all_f32 / *.bytes take integers */
int ss_plan_set_option(ss_plan* plan, int what, int value);         /* [host] 0 side stream on/off, 1 grouped dW on/off, 2 side-stream blocks per CU, 3 BatchNorm statistics / bias column sums from GEMM epilogues on/off, 4 BatchNorm backward recomputes the ReLU gate (on) or reads the saved output (off), 5 an SS_F32 plan runs its GEMMs as SS_F32X3 (bf16 x 3 MFMA on f32 operands) on/off (default off = exact f32), 6 the training-mode forward leaves x_raw untouched and only hands the shifted signal out in shifted_scratch (default off = written back in place like architecture.py:67-68), 7 an SS_F32X3 plan runs every GEMM the 8-wave kernel can take on hi / lo bf16 planes (ss_split_planes / ss_gemm_planes; default on), 8 the bound EF tables are the [hi | lo] tables of ss_relpos_attention_x3_prepare_tables: such a plan also runs its attention on planes (default off), 9 plane GEMMs whose consumers take planes (qkv, the FFN hidden activation and its gradient, dO) write them from their epilogue instead of a split pass on first use (default on), 10 a bf16 training plan keeps [FFN hidden > 0] as sign bits (ss_gemm_epilogue.sign_out) and the FFN input gradient gates from them (gate_bits; default on) */
int ss_plan_set_reduce_hook(ss_plan* plan, ss_reduce_hook fn, void* user); /* [host] */
int ss_plan_set_event_hook(ss_plan* plan, ss_event_hook fn, void* user);   /* [host] */
int64_t ss_plan_ctx_bytes(void);                                    /* [host] */
int64_t ss_plan_workspace_bytes(ss_plan* plan, int B, int T0, int training); /* [host] forward (+ backward temporaries if training) */
/* x_raw (B, T0, 8) f32 -> head [B*T0/8][n_head_cols] f32 (pred | aux logits | zero padding).  training: BatchNorm batch statistics
 * and running-stat updates, dropout (counter-based, keyed by seed), the shift augmentation by shift_r samples, whose result is also
 * written back to x_raw (the reference mutates its input, architecture.py:67-68; shifted_scratch: B*T0*8 floats, may be NULL when
 * shift_r == 0).  ctx_out [host, ss_plan_ctx_bytes()] must be kept, with the workspace, until ss_plan_backward has run. */
int ss_plan_forward(ss_plan* plan, const float* x_raw, float* shifted_scratch, void* workspace, int64_t workspace_bytes, int B, int T0,
                    int training, int shift_r, float dropout_p, uint64_t seed, float* head, void* ctx_out, void* stream);
/* Accumulates (+=) the gradient of every bound parameter; dhead [B*T0/8][n_head_cols] f32.  The weight-gradient GEMMs, bias column sums
 * and gradient un-layouts run on side_stream (may equal stream or be NULL), joined into `stream` before the call returns. */
int ss_plan_backward(ss_plan* plan, void* ctx, const float* dhead, void* stream, void* side_stream);
/* Per-launch timing of the plan's kernels (HIP events on the launch stream; use with the side stream switched off so that
 * durations are exclusive).  ss_plan_profile_read synchronises, aggregates the records gathered since the last read by
 * kernel, clears them and returns the number of rows written. */
typedef struct ss_profile_row { char name[64]; int64_t calls; double seconds, flops /* algorithmic */, bytes /* algorithmic HBM traffic */; } ss_profile_row;
int ss_plan_profile(ss_plan* plan, int enable);                                  /* [host] returns the previous setting */
int ss_plan_profile_read(ss_plan* plan, ss_profile_row* rows, int max_rows);      /* [host] */
/* counters[i][0] += delta for n <= 16 device int64 counters (BatchNorm num_batches_tracked, architecture.py:19,21,25) in one launch */
int ss_counters_add(int n, int64_t** counters /* [host] array of device pointers */, int64_t delta, void* stream);

#ifdef __cplusplus
}
#endif
#endif
