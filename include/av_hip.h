/* C-ABI of libavhip.so — the MI355X (gfx950) kernels behind the audio-visual CTC path.
 *
 * The reference (limeorange1102/multimodal-av-model) has no FFI of its own: every device op is a PyTorch /
 * HuggingFace call.  Each entry point below therefore cites the reference call site (file:line, `hf:` =
 * transformers/models/wav2vec2/modeling_wav2vec2.py 5.15.0, `torch:` = torch/nn 2.10) whose arithmetic it
 * replaces.  Conventions:
 *   - plain pointers + sizes only, no torch types; all buffers are owned by the caller (PyTorch caching
 *     allocator) and outlive the call; workspaces are passed in;
 *   - every call enqueues on the given hipStream_t (passed as void*) and returns immediately;
 *   - return 0 on success, non-zero on error with a message in av_last_error(); never abort();
 *   - dtype codes: AV_F32 = 0 (parity mode), AV_BF16 = 1 (perf mode).  Accumulation is always fp32.
 */
#ifndef AV_HIP_H
#define AV_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define AV_OK 0
#define AV_ERR_ARG 1
#define AV_ERR_LAUNCH 2

#define AV_F32 0
#define AV_BF16 1

/* A-operand addressing modes of av_gemm */
#define AV_A_ROWMAJOR 0 /* A[m][k] at A + m*lda + k                                   */
#define AV_A_TRANS 1    /* A[m][k] at A + k*lda + m           (dW = dY^T X)           */
#define AV_A_CONV2D 2   /* implicit im2col of an NHWC image (ResNet convs, pos-conv)  */
#define AV_A_CONV3D1 3  /* implicit im2col of a 1-channel [N,T,H,W] volume (front-end) */
/* B-operand modes */
#define AV_B_NK 0 /* B[n][k] at B + n*ldb + k  (nn.Linear weight layout)  */
#define AV_B_KN 1 /* B[k][n] at B + k*ldb + n                             */
/* epilogue activation */
#define AV_ACT_NONE 0
#define AV_ACT_GELU 1          /* v = gelu(v)                     hf:565-572, hf:291-299 */
#define AV_ACT_MUL_GELU_GRAD 2 /* v = v * gelu'(aux[m][n])        backward of the above  */
#define AV_ACT_GELU_GF 3       /* m = dropout multiplier; C2 = gelu'(v) * m (the "gradient factor"), v = gelu(v) * m: the
                                  backward of this activation + dropout site is then ONE multiply, AV_ACT_MUL_AUX        */
#define AV_ACT_MUL_AUX 4       /* v = v * aux[m][n]               backward of AV_ACT_GELU_GF (aux = its C2)              */

const char* av_last_error(void);
int av_version(void);

/* ---- MFMA GEMM family: C = epilogue(alpha * A x B) ------------------------------------------------------
 * replaces: nn.Linear (hf:522-527,546,565-572,429-434; model/fusion_module.py:57-58,63; model/decoder.py:24),
 * F.conv1d of the wav2vec2 feature encoder layers 1-6 as a strided GEMM (hf:291-299), the grouped positional
 * conv (hf:360-368), nn.Conv2d of the ResNet trunk (model/encoder.py:9-12,36), nn.Conv3d front-end
 * (model/encoder.py:61), and all the dX / dW products of their backward passes.
 * epilogue, in order: v = alpha*acc; v += bias[n]; C2[m][n] = v (optional pre-activation copy);
 * act; optional dropout; v += R[m][n] (fp32 residual, may alias C); C[m][n] = v;  optional per-column sum / sum-of-squares
 * partials for train-mode BatchNorm statistics (model/trainer.py:54): stats[blockRow][0/1][n].               */
typedef struct av_gemm_args {
    const void* A;
    const void* B;
    void* C;
    void* C2;          /* optional, same dtype/ld as C */
    const float* bias; /* optional [N] */
    const float* R;    /* optional fp32 residual, ld = ldr */
    const void* aux;   /* AV_ACT_MUL_GELU_GRAD / AV_ACT_MUL_AUX operand, dtype = aux_dtype, ld = ldc */
    float* stats;      /* optional [ceil(M/128)][2][N] fp32 partials (batch must be 1) */
    int M, N, K, batch;
    long long lda, ldb, ldc, ldr;
    long long sA, sB, sC, sR, sBias; /* batch strides in elements */
    int a_mode, b_mode, in_dtype, out_dtype, aux_dtype, act;
    float alpha;
    /* implicit-im2col geometry (a_mode 2/3): input [img][T][H][W][Ctot] (T = 1 for 2-D) */
    int cT, cH, cW, cCtot, cCin, cCoff;
    int cKt, cKh, cKw, cSh, cSw, cPt, cPh, cPw, cOh, cOw;
    /* optional two-level batch: z = zo*batch_inner + zi; offsets zo*o? + zi*s? (batch_inner = 0: one level) */
    int batch_inner;
    long long oA, oB, oC;
    /* optional dropout in the epilogue (hf:628-633,565-572 hidden / activation dropout): after the activation and BEFORE the
     * residual, v *= mask(seed, stream, m*ldc + n) in {0, 1/(1-p)}; with AV_ACT_MUL_GELU_GRAD the same mask multiplies the
     * gradient.  drop_p = 0 disables it. */
    float drop_p;
    unsigned int drop_stream;
    unsigned long long drop_seed;
    /* optional split-K over the (one-level) batch: when k_total > 0, batch z multiplies k in [z*K, min((z+1)*K, k_total)).
     * Only with both operands k-major (a_mode 1, b_mode 1, bf16 fast kernel): dW = dY^T X with few output tiles; the
     * partial products C + z*sC are summed with av_sum_slices. */
    int k_total;
    /* position-major pixel order for 2-D implicit-im2col products (a_mode 2, bf16 fast path).  Images are taken in blocks of cNF (the
     * image count must be a multiple of it); inside a block the rows are ordered [pixel position][image of the block]: image q's pixel
     * `pos` of a P-pixel map is row ((q / cNF) * P + pos) * cNF + q % cNF.  cPM bit 0: the INPUT is in that order (P = cH * cW), bit 1:
     * the OUTPUT is (P = cOh * cOw).  With cNF a multiple of the row tile (256) every row tile lies at one output position, so filter taps
     * that fall outside the image for that position are skipped as whole K-tiles (3 x 3 images under a 3 x 3 filter, model/encoder.py:36
     * layer4: 49 of 81 taps do work), and consecutive row tiles re-read the same images' pixels from L2.  0 / 0 = frame-major order. */
    int cNF, cPM;
} av_gemm_args;
int av_gemm(const av_gemm_args* args, void* stream);
/* out[i] = (accumulate ? out[i] : 0) + alpha * sum_s parts[s*stride + i], i < n (fp32): the split-K partials of av_gemm */
int av_sum_slices(const float* parts, int n_slices, long long n, long long stride, float alpha, float* out, int accumulate,
                  void* stream);
/* out[C][Rpad] = in[R][C]^T (zero-filled for r >= R): brings dX / dW products to the fast K-contiguous form */
int av_transpose(const void* in, int idt, void* out, int odt, int R, int C, long long ldi, int Rpad, void* stream);

/* ---- row kernels (one wavefront per row, shuffle reductions) ------------------------------------------- */
/* nn.LayerNorm over the last dim (hf:297,431,638,644,791), eps 1e-5, optional exact-erf GELU (hf:298).
 * x [rows][cols] (dtype xdt), y (dtype ydt); mean/rstd [rows] fp32 optional (saved for backward). */
int av_layernorm_fwd(const void* x, int xdt, const float* gamma, const float* beta, void* y, int ydt,
                     float* mean, float* rstd, long long rows, int cols, float eps, int act, void* stream);
/* dx = dres + LN'(dy) (dres optional fp32); dgamma/dbeta partials [nblk][2][cols] (optional; reduce with
 * av_colsum). x fp32/bf16 (xdt), dy dtype dydt, dx fp32; dx_bf16 (optional): bf16 copy of dx for the next GEMM. */
int av_layernorm_bwd(const void* x, int xdt, const void* dy, int dydt, const float* gamma, const float* mean,
                     const float* rstd, const float* dres, float* dx, float* dgb_partial, int nblk,
                     long long rows, int cols, void* dx_bf16, void* stream);
/* the same with dropout folded into the bf16 copy: dx_bf16 = bf16(dx o mask / (1 - p)), mask = Philox(seed, stream, element index):
 * the backward of a dropout that sits between this LayerNorm's input and the next dX GEMM (hf:633,648: hidden dropout) */
int av_layernorm_bwd_drop(const void* x, int xdt, const void* dy, int dydt, const float* gamma, const float* mean,
                          const float* rstd, const float* dres, float* dx, float* dgb_partial, int nblk,
                          long long rows, int cols, void* dx_bf16, float drop_p, unsigned long long drop_seed,
                          unsigned int drop_stream, void* stream);
/* F.log_softmax(dim=-1) (model/decoder.py:25) and its backward: dx = dy - exp(y) * sum(dy) */
int av_log_softmax_fwd(const void* x, int xdt, float* y, long long rows, int cols, void* stream);
int av_log_softmax_bwd(const float* y, const float* dy, void* dx, int dxdt, long long rows, int cols, void* stream);
/* the same with a row stride ldx >= cols for dx; columns [cols, ldx) are written as zeros, so that dx can feed a GEMM whose K is padded to a
 * multiple of 64 (the CTC head's dX product: vocabulary 800 -> 832) */
int av_log_softmax_bwd_ld(const float* y, const float* dy, void* dx, int dxdt, long long rows, int cols, int ldx, void* stream);
/* column sums of a [rows][cols] matrix (bias gradients): out[cols] (+)= sum_rows x */
int av_colsum(const void* x, int xdt, float* out, long long rows, int cols, long long ld, int accumulate,
              void* stream);
/* elementwise */
int av_cast(const void* x, int xdt, void* y, int ydt, long long n, void* stream);
/* y = cast(x * dropout_mask(seed, stream, i)) — forward dropout and its backward (same mask) in one kernel; p = 0: plain cast */
int av_cast_dropout(const void* x, int xdt, void* y, int ydt, long long n, float p, unsigned long long seed, unsigned int stream_id,
                    void* stream);
/* debug / test: u[i] = the uniform number behind the mask of element i */
int av_dropout_uniform(float* u, long long n, unsigned long long seed, unsigned int stream_id, void* stream);
/* SpecAugment time masking (hf:1272-1296): x[row][:] = embed[:] where mask[row] != 0 */
/* fused cross-attention block of CrossAttentionFusion (model/fusion_module.py:57-61; nn.MultiheadAttention need-weights path,
 * torch:functional.py:6206,6576-6606): packed in-projection + attention core of one (batch item, head) per workgroup.
 * a, v [B, T, 512] bf16 (audio / visual projections); w_in [1536, 512] bf16, b_in [1536] fp32 (in_proj_weight / in_proj_bias);
 * o [B, T, 512] bf16 = concat_h softmax(scale q_h k_h^T) v_h; q_out [B, T, 512], kv_out [B, T, 2, 512], lse [B, H, T] are optional
 * (what the backward reads).  embed_dim 512, 4 heads x 128, T <= 112 */
int av_fusion_xattn_fwd(const void* a, const void* v, const void* w_in, const float* b_in, void* q_out, void* kv_out, void* o,
                        float* lse, int B, int T, int E, int H, float scale, void* stream);
/* backward of that attention core (P recomputed from the row LSE): q, o, dout [B, T, 512], kv [B, T, 2, 512] bf16, lse [B, H, T] ->
 * dq [B, T, 512], dkv [B, T, 2, 512]; one workgroup per (item, head), every operand staged once, no atomics (T <= 112) */
int av_fusion_xattn_bwd(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv, int B, int T,
                        int E, int H, float scale, void* stream);
/* SpecAugment feature-axis masking (hf:1298-1316): x[b, t, c] = 0 for all t where mask[b * H + c] != 0 */
int av_zero_feature_cols(void* x, int xdt, const unsigned char* mask, int B, int T, int H, void* stream);
int av_overwrite_rows(void* x, int xdt, const unsigned char* mask, const float* embed, long long rows, int cols, void* stream);
int av_axpby(float a, const void* x, int xdt, float b, float* y, long long n, void* stream); /* y = a*x + b*y */
int av_mask_rows(void* x, int xdt, const unsigned char* keep, long long rows, int cols, void* stream); /* hf:752-755 */
int av_mul_scalar_dev(const float* x, const float* scalar, float* y, long long n, void* stream); /* y = scalar[0]*x, scalar on device */

/* ---- fused attention (flash-style forward) ---------------------------------------------------------------
 * O = softmax(scale*Q K^T + key-padding mask) V, LSE saved.  Replaces hf:438-463 (sdpa, 16 heads x 64) and the
 * need_weights path of nn.MultiheadAttention (torch:functional.py:6576-6606; fusion_module.py:61, 4 heads x 128).
 * Element (b, t, h, d) of Q at q + b*q_bs + t*q_rs + h*D + d (same for K, V, O).  klen[b] = number of valid
 * keys (NULL: all Tk).  lse [B][H][Tq] fp32 (optional).  D in {16,32,64,128}. */
int av_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B, int H, int Tq,
                     int Tk, int D, long long q_bs, long long q_rs, long long k_bs, long long k_rs, long long v_bs,
                     long long v_rs, long long o_bs, long long o_rs, const int* klen, float scale, float drop_p,
                     unsigned long long drop_seed, unsigned int drop_stream, void* stream);
/* fused (flash-style) attention backward, bf16: recomputes P from Q, K and the forward's LSE; dq/dk/dv written in place.
 * strides[16] = (batch stride, row stride) of q, k, v, o, dout, dq, dk, dv (elements; head stride = D);
 * delta_ws: B*H*Tq floats of workspace.  Backward of hf:438-463 / fusion_module.py:61. */
int av_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                     float* delta_ws, void* dq, void* dk, void* dv, int B, int H, int Tq, int Tk, int D,
                     const long long* strides, const int* klen, float scale, float drop_p, unsigned long long drop_seed,
                     unsigned int drop_stream, void* stream);
/* Attention-probability dropout with PRECOMPUTED keep bits (whole-sequence bf16 kernels only: head_dim 64, Tq, Tk <= 256).
 * av_attention_dropmask evaluates the (seed, stream, index) Philox masks once: mask = [B][H][ceil(Tq / 16)][64] 64-bit words, 32-byte
 * aligned; bit 4 t + e of lane (r, g)'s word of query tile qt <-> (query 16 qt + r, key 16 t + 4 g + e).  The _mask forms of the
 * forward and the backward read 1 bit per probability instead of evaluating the generator (once in the forward, twice in the
 * backward).  Results are bit-identical to the calls without a mask. */
int av_attention_dropmask(void* mask, int B, int H, int Tq, int Tk, float drop_p, unsigned long long drop_seed, unsigned int drop_stream,
                          void* stream);
int av_attention_fwd_mask(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B, int H, int Tq,
                          int Tk, int D, long long q_bs, long long q_rs, long long k_bs, long long k_rs, long long v_bs,
                          long long v_rs, long long o_bs, long long o_rs, const int* klen, float scale, float drop_p,
                          unsigned long long drop_seed, unsigned int drop_stream, const void* drop_mask, void* stream);
int av_attention_bwd_mask(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                          float* delta_ws, void* dq, void* dk, void* dv, int B, int H, int Tq, int Tk, int D,
                          const long long* strides, const int* klen, float scale, float drop_p, unsigned long long drop_seed,
                          unsigned int drop_stream, const void* drop_mask, void* stream);
/* rows of the (unfused, fp32 parity mode) attention backward: P = softmax(scale*S) with key mask; dS = scale * P o (dP - sum(dP o P)) */
int av_softmax_rows(const float* s, void* p, int pdt, long long rows, int cols, float scale, const int* klen,
                    int rows_per_batch, int ld, void* stream);
int av_softmax_bwd_rows(const void* p, int pdt, const float* dp, void* ds, int dsdt, long long rows, int cols,
                        float scale, int ld, void* stream);

/* ---- wav2vec2 feature-encoder layer 0, fused Conv1d(1->C,k,stride)+bias -> LayerNorm(C) -> GELU (hf:291-299).
 * wav [B][T_in] fp32 -> out [B][L_out][C] channel-last in out_dtype; w [C][k] fp32. */
int av_conv0_ln_gelu(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, void* out,
                     int out_dtype, int B, int T_in, int L_out, int C, int k, int stride, float eps, void* stream);

/* ---- bidirectional LSTM time steps (nn.LSTM(512,512,2,bidirectional), model/fusion_module.py:21-27,64) ----
 * Time-major internal buffers; both directions advance in one launch (step s: fwd chain at time s, reverse
 * chain at T-1-s).  gx = x W_ih^T + b_ih + b_hh for all steps comes from av_gemm.  fwd: h,c,(gates) of step s;
 * bwd (s counts from the END of each chain): dgates[t] from dout[t] + dgates[t_next] W_hh, running dc in place. */
int av_lstm_fwd_step(const float* gx, const void* whh, void* hseq, float* cseq, void* gates, void* out_bt, int dtype,
                     int T, int B, int H, int s, void* stream);
int av_lstm_bwd_step(const void* dout, int dout_dtype, long long do_bs, long long do_ts, void* dgates, const void* whhT,
                     const void* gates, const float* cseq, float* dc, int dtype, int T, int B, int H, int s, void* stream);

/* persistent form (bf16, H = 512): ONE launch per layer for all T steps of both directions.  Grid = 32 unit tiles x 2 directions x
 * up to 4 row groups of 16 batch rows; every (direction, row group) is an independent chain of 32 resident workgroups whose steps
 * are separated by an arrival counter (agent-coherent stores / loads, bounded spins: counters[2] is set on timeout).  W_hh slices
 * stay in registers for the whole sequence.  counters: AV_LSTM_COUNTER_INTS ints of device workspace (zeroed by the call).
 * Same buffers as the step kernels; results agree with them to fp32 rounding of the K-quarter partial sums. */
#define AV_LSTM_COUNTER_INTS 1024
int av_lstm_fwd_layer(const float* gx, const void* whh, void* hseq, float* cseq, void* gates, void* out_bt, int* counters,
                      int T, int B, int H, void* stream);
int av_lstm_bwd_layer(const void* dout, int dout_dtype, long long do_bs, long long do_ts, void* dgates, const void* whhT,
                      const void* gates, const float* cseq, float* dc, int* counters, int T, int B, int H, void* stream);

/* ---- fusion glue without host syncs (model/fusion_module.py:40-55,66; model/trainer.py:98,102) ------------- */
int av_mask_downsample(const long long* mask, long long* out, int B, int Tin, int Tout, void* stream);
/* ws_i32: B*Ta + B + groups ints (compaction index, counts, per-group batch max) kept for the backward;
 * the reference's "pad to the batch maximum" is evaluated per group of B/groups consecutive items */
int av_fusion_gather_lerp_fwd(const float* feat, const long long* mask, int* ws_i32, float* out, long long* mask_out,
                              long long* lens, int B, int Ta, int Tv, int D, int groups, void* stream);
int av_fusion_gather_lerp_bwd(const float* dout, const int* ws_i32, float* dfeat, int B, int Ta, int Tv, int D, int groups,
                              void* stream);
int av_permute_bt(const void* in, int idt, void* out, int odt, int B, int T, int D, void* stream); /* [B,T,D]->[T,B,D] */
int av_gather_rows(const void* src, int sdt, const long long* idx, void* out, int odt, long long n, int D, void* stream);
int av_scatter_rows(const float* src, const long long* idx, float* out, long long n, int D, float alpha, int accumulate,
                    void* stream);
/* contrastive.py:24-26 class order (anchors = mask 1, positives = 2, negatives = 0, anything else last), stable: order[] = what a stable
 * argsort of the class rank gives; int64 in, int64 out, one launch */
int av_class_order(const long long* mask, long long n, long long* order, void* stream);

/* ---- lip-frame encoder glue (model/encoder.py:6-75): train-mode BatchNorm, PReLU, pooling; NHWC ------------ */
/* bf16 fast path of the Conv3d(1->64,(5,7,7),(1,2,2),(2,3,3)) front-end (model/encoder.py:61): x [B*T][H][W] fp32,
 * w bf16 [64][288] with k = (kt*7+ky)*8+kx (kx padded to 8, K padded to 288), y bf16 [B*T][H/2][W/2][64],
 * stats [B*T*(H/16)*(W/32)][2][64] BatchNorm partials (optional). */
int av_conv3d_front(const float* x, const void* w, void* y, float* stats, int B, int T, int H, int W, void* stream);
/* the same convolution fused with the first half of model/encoder.py:62-64 (BatchNorm3d + PReLU + MaxPool3d((1,3,3),(1,2,2),(0,1,1))):
 * instead of the conv output it writes, per 3 x 3 / stride 2 / pad 1 pooling window and channel, the MAXIMUM and the MINIMUM of the
 * (bf16-rounded) conv output: ymax, ymin bf16 [B*T][H/4][W/4][64].  Train-mode BatchNorm needs the whole batch before it can be applied,
 * but bn followed by prelu is monotone or V-shaped per channel, so max over the window of prelu(bn(x)) = max(prelu(bn(max x)), prelu(bn(min x)))
 * exactly; av_bn_prelu_minmax evaluates that once scale / shift exist.  stats as above (partials of the UNPOOLED output). */
int av_conv3d_front_pool(const float* x, const void* w, void* ymax, void* ymin, float* stats, int B, int T, int H, int W, void* stream);
/* out[i] = max(prelu(ymax[i] * scale[c] + shift[c]), prelu(ymin[i] * scale[c] + shift[c])), c = i % 64; bf16, n elements (a multiple of 64) */
int av_bn_prelu_minmax(const void* ymax, const void* ymin, const float* scale, const float* shift, const float* slope, void* out, long long n,
                       void* stream);
/* bf16 fast path of the 3x3 / stride 1 / pad 1, 64 -> 64 channel convolutions of ResNet-18 layer1 (model/encoder.py:44-57):
 * x bf16 NHWC [n_img][H][W][64], w bf16 [64][9*64] with k = (ky*3+kx)*64 + c, y bf16 [n_img*H*W][64],
 * stats [ceil(n_img*H*W/256)][2][64] BatchNorm partials (optional).  Weights stay in LDS, the input is staged once per filter row. */
int av_conv3x3_c64(const void* x, const void* w, void* y, float* stats, int n_img, int H, int W, const float* in_scale,
                   const float* in_shift, const float* in_slope, void* stream);
/* ws: 2C + 1 doubles of workspace (2C accumulators + an arrival ticket: train mode is ONE launch, the last workgroup finalizes).
 * ws_zeroed = 1: the caller guarantees ws is zero on entry (the kernel leaves it zero on exit, so one buffer zeroed once serves every
 * BatchNorm of a stream); 0: it is cleared here first. */
int av_bn_finalize(const float* partial, int nblk, long long count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, int training, float* scale,
                   float* shift, int C, double* ws, int ws_zeroed, void* stream);
int av_bn_act(const void* x, const float* scale, const float* shift, const void* res, const float* rscale,
              const float* rshift, const float* slope, void* out, int dtype, long long n, int C, void* stream);
int av_bn_prelu_maxpool(const void* x, const float* scale, const float* shift, const float* slope, void* out, int dtype,
                        long long N, int H, int W, int C, void* stream);
int av_avgpool(const void* x, int dtype, float* out, long long N, int HW, int C, int pm_block, void* stream);   /* pm_block: 0 = frame-major, else images per position-major block (av_gemm_args.cNF) */

/* ---- loss / optimizer side (contrastive.py:8-44; torch.optim.Adam, model/trainer.py:34-39) ------------------ */
int av_l2norm_fwd(const float* x, float* y, float* nrm, long long rows, int cols, float eps, void* stream);
int av_l2norm_bwd(const float* y, const float* dy, const float* nrm, float* dx, long long rows, int cols, float eps,
                  void* stream);
int av_lse_rows(const float* s, float* lse, float* rowsum, long long rows, int cols, int ld, void* stream);
int av_contrastive_dsim(const float* s, const float* lse, void* out, int odt, long long rows, int cols, int ld, float coef,
                        void* stream);
/* column-chunked forms (the N1 x N2 similarity matrix of contrastive.py:30-43 is never materialised whole: S is produced and
 * consumed one [row chunk] x [column chunk] block at a time and recomputed in the backward): av_lse_rows_chunk merges the chunk's
 * row LSE / row sum into the running values when accumulate != 0; av_contrastive_dsim_chunk uses the row LSE over ALL columns
 * and 1 / total_cols */
int av_lse_rows_chunk(const float* s, float* lse, float* rowsum, long long rows, int cols, int ld, int accumulate, void* stream);
int av_contrastive_dsim_chunk(const float* s, const float* lse, void* out, int odt, long long rows, int cols, int ld, float coef,
                              int total_cols, void* stream);
int av_reduce_sum(const float* x, long long n, float* out, float scale, int accumulate, void* stream);
/* loss combination of the trainer (model/trainer.py:111-119): out3 = {sum_i nll[i] w[i] + half_lambda (c1 + c2), 2 x first-half sum,
 * 2 x second-half sum}; nll, w fp32 [n] (n even: speaker 1 then speaker 2), c1 / c2 optional device scalars */
int av_loss_combine(const float* nll, const float* w, const float* c1, const float* c2, float half_lambda, int n, float* out3,
                    void* stream);
/* greedy CTC decoding (beam_search.py:2-48; the reference's beam search returns the per-frame argmax path): log_probs fp32
 * [B][T][V], lengths optional int64 [B] (frames to decode); out_ids int32 [B][T] = collapsed ids padded with -1, out_len int32 [B] */
int av_ctc_greedy(const float* log_probs, const long long* lengths, int* out_ids, int* out_len, int B, int T, int V, int blank,
                  void* stream);
/* device side of the input pipeline (dataset/multi_speaker_dataset.py:13-59; decoding wav / npy files stays on the host):
 * av_lip_gray_resize: src [T][Hs][Ws][C] (uint8 if src_is_u8 else fp32) -> dst fp32 [T][Hd][Wd] = bilinear(mean over C) / divisor
 *   (:49-58: .astype(float32).mean(-1), cv2.resize INTER_LINEAR law, / 255), float32 operation order of the reference;
 * av_mix_pair: mixed[i] = (a1[i] + a2[i]) / (max|a1 + a2| + 1e-6) over n = max(len1, len2) samples with zero padding (:21-32),
 *   mask1 / mask2 int64 [n]: 1 = both speakers, 2 = only this speaker, 0 = otherwise (:35-45); peak_ws: 4 bytes of workspace */
int av_lip_gray_resize(const void* src, int src_is_u8, float* dst, int T, int Hs, int Ws, int C, int Hd, int Wd, float divisor,
                       void* stream);
int av_mix_pair(const float* a1, long long len1, const float* a2, long long len2, float* mixed, long long* mask1, long long* mask2,
                unsigned* peak_ws, void* stream);
int av_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                 int step, float grad_scale, void* stream);
/* multi-tensor form (every optimizer step of the trainer): ptrs [n_tensors][5] device pointers {param, grad, exp_avg, exp_avg_sq,
 * shadow}; shadow = optional (0 = none) bf16 copy of the updated parameter, same element order (the perf path's cached compute-dtype
 * weight); hyper [n_tensors][4] floats {lr, beta1, beta2, eps} (torch.optim.Adam param-group values, model/trainer.py:34-39); chunk c =
 * elements [chunk_start[c], +chunk_elems) of tensor chunk_tensor[c]; steps = device int32 table of PER-TENSOR step counts
 * (torch.optim.Adam's state[p]['step']: a parameter without a gradient in some step - LayerDrop - falls behind the others), tensor t
 * owns steps[step_slot[t]]: read for the bias corrections, incremented by a trailing one-block launch.  All arrays live on the device.
 * scaler_state != NULL = the same step under loss scaling (replaces torch.amp.GradScaler.step + .update, model/trainer.py:40,121-123
 * of the reference; torch/amp/grad_scaler.py:126-129 defaults): 5 device floats {scale, 1/scale, found_inf, growth tracker, steps
 * taken}; non-finite check over every gradient -> Adam with grad * grad_scale / scale, skipped (no counter advances) when found_inf
 * -> scale update (x backoff on overflow, x growth after growth_interval clean steps).  No host synchronisation in either form */
int av_adam_multi(const void* ptrs, const long long* sizes, const float* hyper, const int* chunk_tensor, const long long* chunk_start,
                  int n_chunks, int chunk_elems, int* steps, const int* step_slot, int n_tensors, float grad_scale,
                  float* scaler_state, float growth, float backoff, int growth_interval, void* stream);

/* ---- legacy mel + GRU model (reference: "이전 버전/multimodal_ctc_korean.py":8-55, train loop
 * "이전 버전/train_ctc_korea.py":82-109).  Its convolutions (nn.Conv2d 3x3, :12,15) and all input / weight-gradient products run on
 * av_gemm; these entry points are the rest of it:
 * av_nchw_to_nhwc:      frames [N][C][H][W] fp32 (the (B*T, C, H, W) view of :23) -> [N][H][W][Cp] channel-last, zero padded to Cp
 * av_relu_maxpool2_fwd: y = MaxPool2d(2)(ReLU(x)) (:13-14,16-17), x [N][H][W][C]; y [N][H/2][W/2][C] or, nchw_out = 1,
 *                       [N][C][H/2][W/2] = the (C, H', W') flatten order the GRU input uses (:25)
 * av_relu_maxpool2_bwd: dx = dy routed to the first maximum of each 2x2 window where x > 0 (torch max_pool2d / relu backward)
 * av_im2col3:           cols [(n,h,w)][(ky,kx,c)] of a 3x3 / stride 1 / pad 1 window: k-major operand of dW = dY^T cols
 * av_gru_fwd_step / av_gru_bwd_step: nn.GRU(hidden, 2 layers, bidirectional) time steps (:19,32; gate order r, z, n;
 *   h = (1 - z) n + z h_prev), both directions per launch, time-major buffers as av_lstm_*_step: gx [T][B][2][3H] = x W_ih^T + b_ih
 *   from av_gemm; hseq [T][B][2H] compute dtype; hf [T][B][2][H] fp32 state; gates [T][B][2][4H] = (r, z, n, W_hn h + b_hn);
 *   backward (s counts from the END of each chain): dgi / dgh [T][B][2][3H] from dout[t] + dgh[t_next] W_hh + the direct path
 *   dh o z carried in dhc [2][B][H] */
int av_nchw_to_nhwc(const float* in, void* out, int out_dtype, long long N, int C, int H, int W, int Cp, void* stream);
int av_relu_maxpool2_fwd(const void* x, void* y, int dtype, long long N, int H, int W, int C, int nchw_out, void* stream);
int av_relu_maxpool2_bwd(const void* x, const void* dy, void* dx, int dtype, long long N, int H, int W, int C, int nchw_dy, void* stream);
int av_im2col3(const void* x, void* cols, int dtype, long long N, int H, int W, int C, void* stream);
int av_gru_fwd_step(const float* gx, const void* whh, const float* bhh, void* hseq, float* hf, float* gates, void* out_bt, int dtype,
                    int T, int B, int H, int s, void* stream);
int av_gru_bwd_step(const void* dout, int dout_dtype, long long do_bs, long long do_ts, void* dgi, void* dgh, const void* whhT,
                    const float* gates, const float* hf, float* dhc, int dtype, int T, int B, int H, int s, void* stream);

/* file decoding + resampling of the input pipeline (dataset/multi_speaker_dataset.py:15-19 `librosa.load(path, sr=16000)`):
 * av_wav_info / av_wav_read_mono_f32 are HOST-side file I/O (RIFF/WAVE: PCM 8/16/24/32 bit, IEEE float 32/64; float32 mono =
 *   mean over channels of libsndfile-scaled samples, i.e. what librosa holds before it resamples); `out` is HOST memory;
 * av_resample_sinc (device): Kaiser-windowed-sinc interpolation with a linearly interpolated filter table (win / delta [nwin],
 *   num_table entries per zero crossing, already scaled by min(1, sr_out / sr_in)); n_out = ceil(n_in * sr_out / sr_in).
 *   librosa resamples with soxr_hq, whose filter is not published as a formula: this stage is "parity unpinned" (DESIGN 0, (f)-2). */
int av_wav_info(const char* path, int* sample_rate, int* channels, long long* frames, int* bits, int* is_float);
int av_wav_read_mono_f32(const char* path, long long frame0, long long n, float* out);
int av_resample_sinc(const float* x, long long n_in, float* y, long long n_out, const float* win, const float* delta, int nwin, int num_table,
                     int sr_in, int sr_out, void* stream);

/* ---- one wav2vec2 encoder layer, forward, as ONE call ------------------------------------------------------------------------------------
 * replaces: Wav2Vec2EncoderLayerStableLayerNorm.forward (hf:611-640: layer_norm -> attention (hf:438-463: q/k/v projections, softmax(QK^T)V with
 * probability dropout, out_proj) -> hidden dropout -> + residual -> final_layer_norm -> feed_forward (hf:565-572: intermediate_dense, GELU,
 * activation dropout, output_dense, hidden dropout) -> + residual), 16-bit compute type, fp32 residual stream.
 * The entry point enqueues the same seven kernels with the same arguments as av_layernorm_fwd / av_gemm / av_attention_fwd_mask called one by one
 * (bit-identical results); it exists because at small batches the step is bound by host round trips, not by the device.
 * Buffers (caller-allocated, M = B * T rows): h [M][hidden] fp32 in; x1, x2 [M][hidden], qkv [M][3 hidden] (= [B][T][3][heads][hidden / heads]),
 * ao [M][hidden], g [M][inter], u [M][inter] (optional: the pre-activation, or with gf != 0 the saved gradient factor gelu'(.) o mask / keep)
 * in the 16-bit type; h2, h3 [M][hidden] fp32; mu1 / rs1 / mu2 / rs2 [M] and lse [B][heads][T] fp32 (optional: saved for the backward).
 * Weights in nn.Linear layout in the 16-bit type (w_qkv = the packed [3 hidden][hidden] q / k / v projection), biases and LayerNorm parameters fp32.
 * Dropout masks: the counter-hash streams stream_base + 0 (after out_proj), + 1 (after GELU), + 2 (after output_dense), + 3 (attention
 * probabilities; amask = optional precomputed keep bits of av_attention_dropmask) under `seed`; a probability of 0 disables a site. */
typedef struct av_w2v2_layer_args {
    int B, T, hidden, heads, inter;
    int lp;                      /* dtype code of the 16-bit tensors (AV_BF16 = "the library's 16-bit type") */
    int gf;                      /* != 0: FFN-up runs AV_ACT_GELU_GF (u receives the gradient factor) instead of AV_ACT_GELU (u receives the pre-activation) */
    int stream_base;
    float eps, scale;            /* LayerNorm epsilon; softmax scale (head_dim^-0.5) */
    float hd_p, at_p, ac_p;      /* hidden / attention / activation dropout probabilities */
    unsigned long long seed;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *b_qkv, *b_o, *b_1, *b_2;
    const void *w_qkv, *w_o, *w_1, *w_2;
    const float* h;
    const int* klen;             /* [B] valid key counts or NULL */
    const void* amask;           /* optional keep bits for the attention probabilities */
    void *x1, *qkv, *ao, *x2, *u, *g;
    float *mu1, *rs1, *lse, *h2, *mu2, *rs2, *h3;
} av_w2v2_layer_args;
int av_w2v2_layer_fwd(const av_w2v2_layer_args* args, void* stream);

/* Backward of the same layer when its weights take no gradient (frozen layers above the lowest trainable one: the gradient only passes through):
 * dh [M][hidden] fp32 = gradient of the layer's output h3; dh_lp = optional 16-bit copy of it with THIS layer's FFN-output dropout mask already
 * applied (written by the LayerNorm backward of the layer above; NULL: the entry point makes it in dh3_t).  Saved tensors of av_w2v2_layer_fwd:
 * h, mu1, rs1, qkv, ao, lse, amask, h2, mu2, rs2, u (+ gf).  Weights TRANSPOSED to the K-contiguous form of the dX products, 16-bit:
 * w_2t [inter][hidden] = output_dense^T, w_1t [hidden][inter] = intermediate_dense^T, w_ot [hidden][hidden] = out_proj^T, w_qkvt [hidden][3 hidden].
 * Work / output buffers (caller-allocated): du [M][inter], dx2, dh2_lp, dao, dx1 [M][hidden], dqkv [M][3 hidden] 16-bit; dh2, dh_out [M][hidden] fp32;
 * delta [B][heads][T] fp32; dh_out_lp (optional) = 16-bit copy of dh_out with the dropout mask of stream `lower_stream` (the FFN-output site of the
 * layer below) applied.  Same kernels, same arguments as the per-kernel path of model/w2v2.py::_layer_backward: bit-identical. */
typedef struct av_w2v2_layer_bwd_args {
    int B, T, hidden, heads, inter;
    int lp, gf, stream_base, lower_stream;
    float scale, hd_p, at_p, ac_p;
    unsigned long long seed;
    const float *ln1_g, *ln2_g;
    const void *w_2t, *w_1t, *w_ot, *w_qkvt;
    const float *dh, *h, *mu1, *rs1, *lse, *h2, *mu2, *rs2;
    const void *dh_lp, *qkv, *ao, *amask, *u;
    const int* klen;
    void *dh3_t, *du, *dx2, *dh2_lp, *dao, *dqkv, *dx1, *dh_out_lp;
    float *dh2, *delta, *dh_out;
} av_w2v2_layer_bwd_args;
int av_w2v2_layer_bwd_dx(const av_w2v2_layer_bwd_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif
