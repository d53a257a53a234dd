// One wav2vec2 encoder layer (pre-LN form, hf:611-640) as ONE entry point: the seven launches of its forward - LayerNorm, packed QKV
// projection, whole-sequence attention, output projection (+ hidden dropout + residual), LayerNorm, FFN up (GELU [+ saved gradient factor]
// + activation dropout), FFN down (+ hidden dropout + residual) - enqueued from native code.  The kernels and their arguments are exactly
// those the per-op entry points (av_layernorm_fwd, av_gemm, av_attention_fwd_mask) receive from the Python layer loop: the results are
// bit-identical; what goes away is six of the seven host round trips (argument marshalling) per layer and pass.
#include "../../include/av_hip.h"
#include "av_common.h"

namespace {

int linear(const void* x, const void* w, const float* bias, void* y, int M, int N, int K, int lp, int out_dtype, int act, const float* R, void* C2,
           float drop_p, unsigned long long seed, unsigned stream_id, void* stream, const void* aux = nullptr) {
    av_gemm_args a = {};
    a.A = x; a.B = w; a.C = y; a.C2 = C2; a.bias = bias; a.R = R; a.aux = aux;
    a.M = M; a.N = N; a.K = K; a.batch = 1;
    a.lda = K; a.ldb = K; a.ldc = N; a.ldr = N;
    a.a_mode = 0; a.b_mode = 0; a.in_dtype = lp; a.out_dtype = out_dtype; a.aux_dtype = aux ? lp : 0; a.act = act;
    a.alpha = 1.0f;
    if (drop_p > 0.f) { a.drop_p = drop_p; a.drop_seed = seed; a.drop_stream = stream_id; }
    return av_gemm(&a, stream);
}

}  // namespace

extern "C" int av_w2v2_layer_fwd(const av_w2v2_layer_args* p, void* stream) {
    AV_CHECK(p && p->h && p->x1 && p->qkv && p->ao && p->h2 && p->x2 && p->g && p->h3, "av_w2v2_layer_fwd: null pointer");
    AV_CHECK(p->B > 0 && p->T > 0 && p->heads > 0 && p->hidden % p->heads == 0 && p->inter > 0, "av_w2v2_layer_fwd: bad dims B=%d T=%d hidden=%d heads=%d inter=%d",
             p->B, p->T, p->hidden, p->heads, p->inter);
    AV_CHECK(p->lp == AV_BF16, "av_w2v2_layer_fwd: the 16-bit compute type only (fp32 parity mode keeps the per-op path)");
    const int M = p->B * p->T, Hd = p->hidden, hd = Hd / p->heads;
    const long long es = 2;                                                  // bytes of the 16-bit type
    int rc;
    if ((rc = av_layernorm_fwd(p->h, AV_F32, p->ln1_g, p->ln1_b, p->x1, p->lp, p->mu1, p->rs1, M, Hd, p->eps, AV_ACT_NONE, stream))) return rc;
    if ((rc = linear(p->x1, p->w_qkv, p->b_qkv, p->qkv, M, 3 * Hd, Hd, p->lp, p->lp, AV_ACT_NONE, nullptr, nullptr, 0.f, 0, 0, stream))) return rc;
    {
        // qkv [B][T][3][heads][hd]: q / k / v are the three slices of the third axis
        const char* q = (const char*)p->qkv;
        const long long bs = (long long)p->T * 3 * Hd, rs = 3LL * Hd;
        if ((rc = av_attention_fwd_mask(q, q + (long long)Hd * es, q + 2LL * Hd * es, p->ao, p->lse, p->lp, p->B, p->heads, p->T, p->T, hd, bs, rs, bs, rs, bs, rs,
                                        (long long)p->T * Hd, Hd, p->klen, p->scale, p->at_p, p->seed, (unsigned)(p->stream_base + 3),
                                        p->at_p > 0.f ? p->amask : nullptr, stream)))
            return rc;
    }
    if ((rc = linear(p->ao, p->w_o, p->b_o, p->h2, M, Hd, Hd, p->lp, AV_F32, AV_ACT_NONE, p->h, nullptr, p->hd_p, p->seed, (unsigned)(p->stream_base + 0), stream))) return rc;
    if ((rc = av_layernorm_fwd(p->h2, AV_F32, p->ln2_g, p->ln2_b, p->x2, p->lp, p->mu2, p->rs2, M, Hd, p->eps, AV_ACT_NONE, stream))) return rc;
    if ((rc = linear(p->x2, p->w_1, p->b_1, p->g, M, p->inter, Hd, p->lp, p->lp, p->gf ? AV_ACT_GELU_GF : AV_ACT_GELU, nullptr, p->u, p->ac_p, p->seed,
                     (unsigned)(p->stream_base + 1), stream)))
        return rc;
    return linear(p->g, p->w_2, p->b_2, p->h3, M, Hd, p->inter, p->lp, AV_F32, AV_ACT_NONE, p->h2, nullptr, p->hd_p, p->seed, (unsigned)(p->stream_base + 2), stream);
}

// The backward of the same layer for a layer WITHOUT weight gradients (frozen weights: the gradient only passes through - 25 of the 32 layer
// backwards of a benchmark step): saved-factor / GELU-gradient dX of the FFN, LayerNorm backward (+ the 16-bit copy with the next dropout site's
// mask), output-projection dX, attention backward, QKV dX, LayerNorm backward - the launches of model/w2v2.py::_layer_backward with tr = False.
extern "C" int av_w2v2_layer_bwd_dx(const av_w2v2_layer_bwd_args* p, void* stream) {
    AV_CHECK(p && p->dh && p->h && p->h2 && p->qkv && p->ao && p->lse && p->u && p->du && p->dx2 && p->dh2 && p->dh2_lp && p->dao && p->dqkv && p->delta &&
             p->dx1 && p->dh_out, "av_w2v2_layer_bwd_dx: null pointer");
    AV_CHECK(p->dh_lp || p->dh3_t, "av_w2v2_layer_bwd_dx: neither a 16-bit copy of dh nor a buffer to make one");
    AV_CHECK(p->lp == AV_BF16 && p->hidden % p->heads == 0, "av_w2v2_layer_bwd_dx: 16-bit compute type only; hidden %% heads == 0");
    const int M = p->B * p->T, Hd = p->hidden, I = p->inter, hd = Hd / p->heads;
    const long long es = 2;
    int nblk = (M + 15) / 16;                                    // ops.layernorm_bwd's block count: max(1, min(512, ceil(rows / 16)))
    nblk = nblk > 512 ? 512 : (nblk < 1 ? 1 : nblk);
    int rc;
    const void* dh3_t = p->dh_lp;
    if (!dh3_t) {                                                // the FFN-output dropout site of THIS layer, applied to the incoming gradient
        rc = p->hd_p > 0.f ? av_cast_dropout(p->dh, AV_F32, p->dh3_t, p->lp, (long long)M * Hd, p->hd_p, p->seed, (unsigned)(p->stream_base + 2), stream)
                           : av_cast(p->dh, AV_F32, p->dh3_t, p->lp, (long long)M * Hd, stream);
        if (rc) return rc;
        dh3_t = p->dh3_t;
    }
    // du = (dh3 W2) o saved factor   |   o gelu'(u) o activation-dropout mask
    if ((rc = linear(dh3_t, p->w_2t, nullptr, p->du, M, I, Hd, p->lp, p->lp, p->gf ? AV_ACT_MUL_AUX : AV_ACT_MUL_GELU_GRAD, nullptr, nullptr,
                     p->gf ? 0.f : p->ac_p, p->seed, (unsigned)(p->stream_base + 1), stream, p->u)))
        return rc;
    if ((rc = linear(p->du, p->w_1t, nullptr, p->dx2, M, Hd, I, p->lp, p->lp, AV_ACT_NONE, nullptr, nullptr, 0.f, 0, 0, stream))) return rc;
    rc = p->hd_p > 0.f ? av_layernorm_bwd_drop(p->h2, AV_F32, p->dx2, p->lp, p->ln2_g, p->mu2, p->rs2, p->dh, p->dh2, nullptr, nblk, M, Hd, p->dh2_lp, p->hd_p, p->seed,
                                              (unsigned)(p->stream_base + 0), stream)
                       : av_layernorm_bwd(p->h2, AV_F32, p->dx2, p->lp, p->ln2_g, p->mu2, p->rs2, p->dh, p->dh2, nullptr, nblk, M, Hd, p->dh2_lp, stream);
    if (rc) return rc;
    if ((rc = linear(p->dh2_lp, p->w_ot, nullptr, p->dao, M, Hd, Hd, p->lp, p->lp, AV_ACT_NONE, nullptr, nullptr, 0.f, 0, 0, stream))) return rc;
    {
        const char* q = (const char*)p->qkv;
        char* dq = (char*)p->dqkv;
        const long long bs3 = (long long)p->T * 3 * Hd, rs3 = 3LL * Hd, bs1 = (long long)p->T * Hd, rs1 = Hd;
        const long long st[16] = {bs3, rs3, bs3, rs3, bs3, rs3, bs1, rs1, bs1, rs1, bs3, rs3, bs3, rs3, bs3, rs3};     // q k v o do dq dk dv
        if ((rc = av_attention_bwd_mask(q, q + (long long)Hd * es, q + 2LL * Hd * es, p->ao, p->dao, p->lse, p->delta, dq, dq + (long long)Hd * es, dq + 2LL * Hd * es,
                                        p->B, p->heads, p->T, p->T, hd, st, p->klen, p->scale, p->at_p, p->seed, (unsigned)(p->stream_base + 3),
                                        p->at_p > 0.f ? p->amask : nullptr, stream)))
            return rc;
    }
    if ((rc = linear(p->dqkv, p->w_qkvt, nullptr, p->dx1, M, Hd, 3 * Hd, p->lp, p->lp, AV_ACT_NONE, nullptr, nullptr, 0.f, 0, 0, stream))) return rc;
    if (p->dh_out_lp && p->hd_p > 0.f)
        return av_layernorm_bwd_drop(p->h, AV_F32, p->dx1, p->lp, p->ln1_g, p->mu1, p->rs1, p->dh2, p->dh_out, nullptr, nblk, M, Hd, p->dh_out_lp, p->hd_p, p->seed,
                                     (unsigned)p->lower_stream, stream);
    return av_layernorm_bwd(p->h, AV_F32, p->dx1, p->lp, p->ln1_g, p->mu1, p->rs1, p->dh2, p->dh_out, nullptr, nblk, M, Hd, p->dh_out_lp, stream);
}
