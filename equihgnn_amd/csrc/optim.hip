// Optimiser-side helpers of the graphed training step (main.py:137-140: torch.optim.Adam(lr, weight_decay)).
//
// eqh_adam_step: Adam over ONE flat fp32 tensor (the trainer keeps all parameters in one buffer, their
//   gradients in another).  torch's fused multi-tensor Adam covers a single tensor with one 512-thread
//   block per 64 Ki elements -- 40 blocks for the 2.6 M parameters of egnn_equihnns, 46 us; an ordinary
//   grid-stride elementwise kernel needs 12.  Same arithmetic as torch.optim.Adam (amsgrad = False,
//   L2 weight decay folded into the gradient, bias corrections from a step counter), fp32 throughout.
//   The learning rate and the step counter live in device memory, so a captured hipGraph keeps working
//   when a scheduler changes the rate; the counter is advanced by the last workgroup to finish (a ticket
//   in the state block), after every workgroup has read it.
// eqh_copy_many: up to 64 small device-to-device copies in one launch (gradients that autograd allocated,
//   packed into the flat gradient buffer).
#include "common.h"

namespace {

struct AdamState {       // device memory, 16 bytes, zero-initialised by the caller
    int64_t step;
    unsigned int ticket;
    unsigned int pad;
};

__global__ void __launch_bounds__(256)
k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
       const float* __restrict__ lr_ptr, float beta1, float beta2, float eps, float wd, float gscale,
       AdamState* __restrict__ st, int zero_grad, float* __restrict__ zero_also, int64_t zero_also_n) {
    // AD_U quads per thread and pass, all sixteen 16-byte loads of a pass requested before anything else -- before the bias
    // corrections are worked out, too (the step counter's load, two powf and a barrier otherwise sit in front of every memory
    // request).  Round 6: one quad per thread on 1340 workgroups moved 44 MB in 23 us (1.9 TB/s, four loads in flight per
    // thread and 1340 tickets on one address); a quarter of the workgroups with four times the bytes in flight each.
    constexpr int AD_U = 4;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * AD_U;
    int64_t i0 = (int64_t)blockIdx.x * blockDim.x * AD_U + threadIdx.x;
    float4 pp[AD_U], gg[AD_U], mm[AD_U], vv[AD_U];
    auto load = [&](int64_t base) {
#pragma unroll
        for (int u = 0; u < AD_U; ++u) {
            const int64_t i = base + (int64_t)u * blockDim.x;
            if (i < n4) {
                pp[u] = reinterpret_cast<float4*>(p)[i];
                gg[u] = reinterpret_cast<const float4*>(g)[i];
                mm[u] = reinterpret_cast<float4*>(m)[i];
                vv[u] = reinterpret_cast<float4*>(v)[i];
            }
        }
    };
    load(i0);
    // the bias corrections once per workgroup (two powf per THREAD were most of the kernel's instructions)
    __shared__ float s_corr[2];
    if (threadIdx.x == 0) {
        const float t = (float)(st->step + 1);
        s_corr[0] = *lr_ptr / (1.0f - powf(beta1, t));
        s_corr[1] = sqrtf(1.0f - powf(beta2, t));
    }
    __syncthreads();
    const float step_size = s_corr[0];
    const float bc2_sqrt = s_corr[1];
    for (; i0 < n4; i0 += stride) {
#pragma unroll
        for (int u = 0; u < AD_U; ++u) {
            const int64_t i = i0 + (int64_t)u * blockDim.x;
            if (i >= n4) continue;
            float* pe = &pp[u].x; float* ge = &gg[u].x; float* me = &mm[u].x; float* ve = &vv[u].x;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float gr = ge[c] * gscale;
                if (wd != 0.f) gr = fmaf(wd, pe[c], gr);
                me[c] = beta1 * me[c] + (1.0f - beta1) * gr;
                ve[c] = beta2 * ve[c] + (1.0f - beta2) * gr * gr;
                const float denom = sqrtf(ve[c]) / bc2_sqrt + eps;
                pe[c] -= step_size * (me[c] / denom);
            }
            reinterpret_cast<float4*>(p)[i] = pp[u];
            reinterpret_cast<float4*>(m)[i] = mm[u];
            reinterpret_cast<float4*>(v)[i] = vv[u];
            if (zero_grad) reinterpret_cast<float4*>(g)[i] = f4_zero();   // the next step's kernels accumulate into zeros
        }
        if (i0 + stride < n4) load(i0 + stride);
    }
    // a second buffer cleared on the way (accumulators that are not parameter gradients: the merged weights' scratch)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (zero_also_n >> 2); i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(zero_also)[i] = f4_zero();
    if (blockIdx.x == 0) {  // ragged tail (n not a multiple of 4)
        for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) {
            float gr = g[i] * gscale;
            if (wd != 0.f) gr = fmaf(wd, p[i], gr);
            const float mi = beta1 * m[i] + (1.0f - beta1) * gr;
            const float vi = beta2 * v[i] + (1.0f - beta2) * gr * gr;
            m[i] = mi;
            v[i] = vi;
            p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
            if (zero_grad) g[i] = 0.f;
        }
    }
    // the last workgroup to finish advances the step counter: every workgroup read it before finishing
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int ticket = atomicAdd(&st->ticket, 1u);
        if (ticket == gridDim.x - 1) {
            st->step += 1;
            st->ticket = 0;
        }
    }
}

constexpr int COPY_MAX = 64;
struct CopyBatch {
    const float* src[COPY_MAX];
    float* dst[COPY_MAX];
    int n[COPY_MAX];
};

__global__ void __launch_bounds__(256) k_copy_many(CopyBatch b) {
    const float* __restrict__ s = b.src[blockIdx.x];
    float* __restrict__ d = b.dst[blockIdx.x];
    const int n = b.n[blockIdx.x];
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {  // float4 body, scalar tail
        const int n4 = n >> 2;
        for (int i = blockIdx.y * 256 + threadIdx.x; i < n4; i += gridDim.y * 256)
            reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
        if (blockIdx.y == 0)
            for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) d[i] = s[i];
    } else {
        for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) d[i] = s[i];
    }
}

__global__ void __launch_bounds__(1024)
k_mse(const float* __restrict__ pred, const float* __restrict__ target, int n, float* __restrict__ loss,
      float* __restrict__ grad) {
    __shared__ float s_sum[1024];
    const float inv_n = 1.0f / (float)n;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {   // fixed assignment and tree: reproducible
        const float d = pred[i] - target[i];
        acc = fmaf(d, d, acc);
        grad[i] = 2.0f * d * inv_n;
    }
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_sum[threadIdx.x] += s_sum[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = s_sum[0] * inv_n;
}

}  // namespace

extern "C" int eqh_mse_fwd_bwd(const float* pred, const float* target, int32_t n, float* loss, float* grad,
                               void* stream_) {
    if (n <= 0 || n > 65536 || !pred || !target || !loss || !grad) return EQH_ERR_ARG;
    hipLaunchKernelGGL(k_mse, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream_), pred, target, (int)n, loss, grad);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int eqh_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             const float* lr, float beta1, float beta2, float eps, float weight_decay,
                             float grad_scale, void* state, int32_t zero_grad, float* zero_also, int64_t zero_also_n,
                             void* stream_) {
    if (n < 0 || !lr || !state) return EQH_ERR_ARG;
    if (n == 0) return EQH_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return EQH_ERR_ARG;
    if (zero_also_n < 0 || (zero_also_n > 0 && !zero_also) || (zero_also_n & 3) || !eqh_aligned16(zero_also)) return EQH_ERR_ARG;
    if (!eqh_aligned16(param) || !eqh_aligned16(grad) || !eqh_aligned16(exp_avg) || !eqh_aligned16(exp_avg_sq) ||
        ((uintptr_t)state & 7))
        return EQH_ERR_ALIGN;
    hipLaunchKernelGGL(k_adam, dim3(eqh_grid_for(n / 16 + 1, 256, 2048)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_scale,
                       static_cast<AdamState*>(state), (int)zero_grad, zero_also, zero_also_n);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int eqh_copy_many(int32_t count, const float* const* src, float* const* dst, const int64_t* n,
                             void* stream_) {
    if (count < 0 || (count > 0 && (!src || !dst || !n))) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    for (int i0 = 0; i0 < count; i0 += COPY_MAX) {
        CopyBatch b;
        const int m = (count - i0 < COPY_MAX) ? count - i0 : COPY_MAX;
        int64_t most = 0;
        for (int i = 0; i < m; ++i) {
            if (n[i0 + i] < 0 || n[i0 + i] >= ((int64_t)1 << 31) || (n[i0 + i] > 0 && (!src[i0 + i] || !dst[i0 + i])))
                return EQH_ERR_ARG;
            b.src[i] = src[i0 + i];
            b.dst[i] = dst[i0 + i];
            b.n[i] = (int)n[i0 + i];
            if (n[i0 + i] > most) most = n[i0 + i];
        }
        if (most == 0) continue;
        const int by = (int)((most + 4095) / 4096 < 256 ? (most + 4095) / 4096 : 256);  // 16 floats per thread and pass
        hipLaunchKernelGGL(k_copy_many, dim3(m, by), dim3(256), 0, stream, b);
        EQH_CHECK_LAUNCH();
    }
    return EQH_OK;
}
