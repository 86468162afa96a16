// fp32-VALU path of the fused  shift -> window_partition -> attention -> window_reverse -> shift_back
// kernel: any power-of-two window size in [4, 256] and any head_dim <= 128 (padded to a power of two), fp32 or bf16
// activations, all arithmetic in fp32.  This is the production path for fp32 models (exact-fp32
// scores, as the reference computes them) and the fallback for shapes the MFMA path does not cover.
//
// Layout: one thread per (image, shifted position, head) "row"; a workgroup covers max(64, Ws)
// consecutive shifted positions (= whole windows) of one head.  K and V rows of the covered windows sit
// in LDS (row stride HD+1 floats: conflict-free both for the broadcast reads of the score loop and for
// the per-thread row writes); every thread keeps its own query row, running max/sum and output row
// in registers and walks the window's keys with an online softmax, four keys per rescale.
#include "window_attn.h"

namespace hs {
HS_DEFINE_SEED_EPOCH_SETTER(set_seed_epoch_attn_generic)
namespace {

constexpr float kNormEps = 1e-12f;   // F.normalize eps, swin_hp_transformer.py:143
constexpr float kMaskValue = -100.f; // hp_shifting.py:25

template <typename T, int HD>
__global__ void __launch_bounds__(256) attn_fwd_generic_kernel(AttnParams p) {
    apply_seed_epoch(p);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = HD + 1;
    const int nthr = blockDim.x, t = threadIdx.x, h = blockIdx.y, Ws = p.Ws, hd = p.hd;
    float* k_s = smem;
    float* v_s = k_s + nthr * LD;
    int* lab_s = (int*)(v_s + nthr * LD);

    const int64_t g = (int64_t)blockIdx.x * nthr + t;  // global shifted row: b * N + j
    const bool valid = g < (int64_t)p.B * p.N;
    const bool cosine = (p.flags & HS_ATTN_COSINE) != 0;
    const float hscale = p.head_scale[h];

    float q[HD];
    int64_t tok = 0, j = 0;
    int b = 0;
    if (valid) {
        b = (int)(g / p.N);
        j = g - (int64_t)b * p.N;
        tok = (int64_t)b * p.N + shifted_source(p, j);
        const int64_t base = tok * 3 * p.C + (int64_t)h * hd;
        float qn = 0.f, kn = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const bool in = d < hd;
            const float qv = in ? io<T>::load(p.qkv, base + d) : 0.f;
            const float kv = in ? io<T>::load(p.qkv, base + p.C + d) : 0.f;
            const float vv = in ? io<T>::load(p.qkv, base + 2 * p.C + d) : 0.f;
            q[d] = qv;
            k_s[t * LD + d] = kv;
            v_s[t * LD + d] = vv;
            qn += qv * qv;
            kn += kv * kv;
        }
        if (cosine) {
            const float qi = 1.f / fmaxf(sqrtf(qn), kNormEps), ki = 1.f / fmaxf(sqrtf(kn), kNormEps);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                q[d] *= qi;
                k_s[t * LD + d] *= ki;
            }
        }
        lab_s[t] = p.labels ? (int)p.labels[j] : 0;
    }
    __syncthreads();
    if (!valid) return;

    const int i = (int)(j % Ws);
    const int t0 = t - i;  // first row of this thread's window inside the workgroup
    const int my_lab = lab_s[t];
    const float* bias_row = p.bias ? p.bias + ((int64_t)h * Ws + i) * Ws : nullptr;

    float m = -INFINITY, l = 0.f;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    const bool dropping = p.drop_p > 0.f;
    const DropRng rng(p, ((int64_t)b * p.nH + h) * p.N + j);

    for (int j0 = 0; j0 < Ws; j0 += 4) {
        float s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* kr = k_s + (t0 + j0 + u) * LD;
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) dot = fmaf(q[d], kr[d], dot);
            float sv = hscale * dot;
            if (bias_row) sv += bias_row[j0 + u];
            if (lab_s[t0 + j0 + u] != my_lab) sv += kMaskValue;
            s[u] = sv;
        }
        const float m_new = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), m);
        const float corr = expf(m - m_new);
        l *= corr;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= corr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float e = expf(s[u] - m_new);
            l += e;  // the softmax denominator is not affected by dropout
            const float ed = dropping ? e * rng.mult(j0 + u) : e;
            const float* vr = v_s + (t0 + j0 + u) * LD;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] = fmaf(ed, vr[d], o[d]);
        }
        m = m_new;
    }
    const float linv = 1.f / l;
    const int64_t obase = tok * p.C + (int64_t)h * hd;
#pragma unroll
    for (int d = 0; d < HD; ++d)
        if (d < hd) io<T>::store(p.out, obase + d, o[d] * linv);
    if (p.lse) p.lse[((int64_t)b * p.nH + h) * p.N + j] = m + logf(l);
}

// Backward.  Phase A: thread = query row (dq, bias/scale gradients).  Phase B: thread = key row
// (dk, dv), re-deriving the probabilities from the saved log-sum-exp.
template <typename T, int HD>
__global__ void __launch_bounds__(256) attn_bwd_generic_kernel(AttnParams p) {
    apply_seed_epoch(p);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = HD + 1;
    const int nthr = blockDim.x, t = threadIdx.x, h = blockIdx.y, Ws = p.Ws, hd = p.hd;
    float* q_s = smem;              // normalised (cosine) or raw q rows
    float* k_s = q_s + nthr * LD;   // normalised (cosine) or raw k rows
    float* v_s = k_s + nthr * LD;
    float* do_s = v_s + nthr * LD;
    float* lse_s = do_s + nthr * LD;
    float* dsum_s = lse_s + nthr;   // D_i = dO_i . O_i
    int* lab_s = (int*)(dsum_s + nthr);

    const int64_t g = (int64_t)blockIdx.x * nthr + t;
    const bool valid = g < (int64_t)p.B * p.N;
    const bool cosine = (p.flags & HS_ATTN_COSINE) != 0;
    const float hscale = p.head_scale[h];

    int64_t tok = 0, j = 0;
    int b = 0;
    float q_ninv = 1.f, k_ninv = 1.f, q_norm = 1.f, k_norm = 1.f;
    if (valid) {
        b = (int)(g / p.N);
        j = g - (int64_t)b * p.N;
        tok = (int64_t)b * p.N + shifted_source(p, j);
        const int64_t base = tok * 3 * p.C + (int64_t)h * hd;
        const int64_t obase = tok * p.C + (int64_t)h * hd;
        float qn = 0.f, kn = 0.f, dsum = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const bool in = d < hd;
            const float qv = in ? io<T>::load(p.qkv, base + d) : 0.f;
            const float kv = in ? io<T>::load(p.qkv, base + p.C + d) : 0.f;
            const float vv = in ? io<T>::load(p.qkv, base + 2 * p.C + d) : 0.f;
            const float dov = in ? io<T>::load(p.dout, obase + d) : 0.f;
            const float ov = in ? io<T>::load(p.out, obase + d) : 0.f;
            q_s[t * LD + d] = qv;
            k_s[t * LD + d] = kv;
            v_s[t * LD + d] = vv;
            do_s[t * LD + d] = dov;
            qn += qv * qv;
            kn += kv * kv;
            dsum = fmaf(dov, ov, dsum);
        }
        if (cosine) {
            q_norm = sqrtf(qn);
            k_norm = sqrtf(kn);
            q_ninv = 1.f / fmaxf(q_norm, kNormEps);
            k_ninv = 1.f / fmaxf(k_norm, kNormEps);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                q_s[t * LD + d] *= q_ninv;
                k_s[t * LD + d] *= k_ninv;
            }
        }
        lse_s[t] = p.lse[((int64_t)b * p.nH + h) * p.N + j];
        dsum_s[t] = dsum;
        lab_s[t] = p.labels ? (int)p.labels[j] : 0;
    }
    __syncthreads();

    float dscale_acc = 0.f;
    if (valid) {
        const int i = (int)(j % Ws);
        const int t0 = t - i;
        const int my_lab = lab_s[t];
        const int64_t base = tok * 3 * p.C + (int64_t)h * hd;

        // ---------------- phase A: this thread is query row i
        {
            float qh[HD], dov[HD], dq[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                qh[d] = q_s[t * LD + d];
                dov[d] = do_s[t * LD + d];
                dq[d] = 0.f;
            }
            const float my_lse = lse_s[t], my_dsum = dsum_s[t];
            const bool dropping = p.drop_p > 0.f;
            const DropRng rng(p, ((int64_t)b * p.nH + h) * p.N + j);
            const float* bias_row = p.bias ? p.bias + ((int64_t)h * Ws + i) * Ws : nullptr;
            float* dbias_row = p.dbias ? p.dbias + ((int64_t)h * Ws + i) * Ws : nullptr;
            for (int jj = 0; jj < Ws; ++jj) {
                const float* kr = k_s + (t0 + jj) * LD;
                const float* vr = v_s + (t0 + jj) * LD;
                float dot = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dot = fmaf(qh[d], kr[d], dot);
                    dp = fmaf(dov[d], vr[d], dp);
                }
                float s = hscale * dot;
                if (bias_row) s += bias_row[jj];
                if (lab_s[t0 + jj] != my_lab) s += kMaskValue;
                const float pr = expf(s - my_lse);
                if (dropping) dp *= rng.mult(jj);  // d(out)/d(P) passes through the mask
                const float ds = pr * (dp - my_dsum);
                const float dsk = ds * hscale;
#pragma unroll
                for (int d = 0; d < HD; ++d) dq[d] = fmaf(dsk, kr[d], dq[d]);
                if (dbias_row) atomicAdd(dbias_row + jj, ds);
                dscale_acc = fmaf(ds, dot, dscale_acc);
            }
            if (cosine) {  // through q_hat = q / max(|q|, eps)
                float proj = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) proj = fmaf(qh[d], dq[d], proj);
                const bool clamped = q_norm <= kNormEps;
#pragma unroll
                for (int d = 0; d < HD; ++d) dq[d] = (clamped ? dq[d] : dq[d] - qh[d] * proj) * q_ninv;
            }
#pragma unroll
            for (int d = 0; d < HD; ++d)
                if (d < hd) io<T>::store(p.dqkv, base + d, dq[d]);
        }
        // ---------------- phase B: this thread is key row i
        {
            float kh[HD], vv[HD], dk[HD], dv[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                kh[d] = k_s[t * LD + d];
                vv[d] = v_s[t * LD + d];
                dk[d] = 0.f;
                dv[d] = 0.f;
            }
            const float* bias_col = p.bias ? p.bias + (int64_t)h * Ws * Ws + i : nullptr;
            const bool dropping = p.drop_p > 0.f;
            const int64_t row0 = ((int64_t)b * p.nH + h) * p.N + (j - i);  // first query row of this window
            for (int ii = 0; ii < Ws; ++ii) {
                const float* qr = q_s + (t0 + ii) * LD;
                const float* dor = do_s + (t0 + ii) * LD;
                float dot = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dot = fmaf(qr[d], kh[d], dot);
                    dp = fmaf(dor[d], vv[d], dp);
                }
                float s = hscale * dot;
                if (bias_col) s += bias_col[(int64_t)ii * Ws];
                if (lab_s[t0 + ii] != my_lab) s += kMaskValue;
                const float pr = expf(s - lse_s[t0 + ii]);
                float prd = pr;  // dropped probability (what multiplied V in the forward)
                if (dropping) {
                    const float mlt = DropRng(p, row0 + ii).mult(i);
                    dp *= mlt;
                    prd *= mlt;
                }
                const float ds = pr * (dp - dsum_s[t0 + ii]);
                const float dsq = ds * hscale;
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dk[d] = fmaf(dsq, qr[d], dk[d]);
                    dv[d] = fmaf(prd, dor[d], dv[d]);
                }
            }
            if (cosine) {
                float proj = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) proj = fmaf(kh[d], dk[d], proj);
                const bool clamped = k_norm <= kNormEps;
#pragma unroll
                for (int d = 0; d < HD; ++d) dk[d] = (clamped ? dk[d] : dk[d] - kh[d] * proj) * k_ninv;
            }
#pragma unroll
            for (int d = 0; d < HD; ++d)
                if (d < hd) {
                    io<T>::store(p.dqkv, base + p.C + d, dk[d]);
                    io<T>::store(p.dqkv, base + 2 * p.C + d, dv[d]);
                }
        }
    }
    if (cosine && p.dhead_scale) {  // uniform branch
        const float tot = wave_sum(dscale_acc);
        if ((t & 63) == 0) atomicAdd(p.dhead_scale + h, tot);
    }
}

template <typename T, int HD>
int launch_fwd(const AttnParams& p, hipStream_t stream) {
    const int nthr = p.Ws > 64 ? p.Ws : 64;
    const size_t smem = (size_t)nthr * (2 * (HD + 1) + 1) * sizeof(float);
    if (smem > 160 * 1024) return fail(HS_ERR_UNSUPPORTED, "window %d x head_dim %d needs %zu B of LDS", p.Ws, p.hd, smem);
    auto kern = attn_fwd_generic_kernel<T, HD>;
    if (smem > 48 * 1024) HS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int64_t rows = (int64_t)p.B * p.N;
    dim3 grid((unsigned)((rows + nthr - 1) / nthr), (unsigned)p.nH);
    hipLaunchKernelGGL(kern, grid, dim3(nthr), smem, stream, p);
    HS_LAUNCH_CHECK("attn_fwd_generic");
    return HS_OK;
}

template <typename T, int HD>
int launch_bwd(const AttnParams& p, hipStream_t stream) {
    const int nthr = p.Ws > 64 ? p.Ws : 64;
    const size_t smem = (size_t)nthr * (4 * (HD + 1) + 3) * sizeof(float);
    if (smem > 160 * 1024) return fail(HS_ERR_UNSUPPORTED, "window %d x head_dim %d needs %zu B of LDS (backward)", p.Ws, p.hd, smem);
    auto kern = attn_bwd_generic_kernel<T, HD>;
    if (smem > 48 * 1024) HS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int64_t rows = (int64_t)p.B * p.N;
    dim3 grid((unsigned)((rows + nthr - 1) / nthr), (unsigned)p.nH);
    hipLaunchKernelGGL(kern, grid, dim3(nthr), smem, stream, p);
    HS_LAUNCH_CHECK("attn_bwd_generic");
    return HS_OK;
}

template <typename T>
int dispatch_fwd(const AttnParams& p, hipStream_t s) {
    const int hd = p.hd;
    if (hd <= 2) return launch_fwd<T, 2>(p, s);
    if (hd <= 4) return launch_fwd<T, 4>(p, s);
    if (hd <= 8) return launch_fwd<T, 8>(p, s);
    if (hd <= 16) return launch_fwd<T, 16>(p, s);
    if (hd <= 32) return launch_fwd<T, 32>(p, s);
    if (hd <= 64) return launch_fwd<T, 64>(p, s);
    if (hd <= 128) return launch_fwd<T, 128>(p, s);
    return fail(HS_ERR_UNSUPPORTED, "head_dim %d > 128", hd);
}
template <typename T>
int dispatch_bwd(const AttnParams& p, hipStream_t s) {
    const int hd = p.hd;
    if (hd <= 2) return launch_bwd<T, 2>(p, s);
    if (hd <= 4) return launch_bwd<T, 4>(p, s);
    if (hd <= 8) return launch_bwd<T, 8>(p, s);
    if (hd <= 16) return launch_bwd<T, 16>(p, s);
    if (hd <= 32) return launch_bwd<T, 32>(p, s);
    if (hd <= 64) return launch_bwd<T, 64>(p, s);
    if (hd <= 128) return launch_bwd<T, 128>(p, s);
    return fail(HS_ERR_UNSUPPORTED, "head_dim %d > 128 (backward)", hd);
}

}  // namespace

int launch_attn_fwd_generic(const AttnParams& p, int dtype, hipStream_t stream) {
    return dtype == HS_BF16 ? dispatch_fwd<bf16_t>(p, stream) : dispatch_fwd<float>(p, stream);
}
int launch_attn_bwd_generic(const AttnParams& p, int dtype, hipStream_t stream) {
    return dtype == HS_BF16 ? dispatch_bwd<bf16_t>(p, stream) : dispatch_bwd<float>(p, stream);
}

}  // namespace hs
