// optim.hip -- global-norm gradient clipping + AdamW over the flat parameter arena (HBM-bound, 28 B/param).
//
// Replaces torch.nn.utils.clip_grad_norm_(loss_model.parameters(), max_grad_norm) followed by
// torch.optim.AdamW(lr, betas=(0.9,0.999), eps=1e-8, weight_decay on all but bias/LayerNorm).step()
// and optimizer.zero_grad() as SentenceTransformer.fit runs them (sentence-transformers 2.2.2;
// reference call site /root/reference/training/main.py:128-148).
// The clip coefficient is computed on the device from the reduced norm: no host sync in the step.
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

constexpr int kNormBlocks = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, int64_t n4, float* partial) {
    float s = 0.f;
    const f32x4* g4 = (const f32x4*)g;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = g4[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void norm_finish_kernel(const float* partial, int n, float grad_scale, float* norm_out) {
    __shared__ float red[16];
    float s = threadIdx.x < n ? partial[threadIdx.x] : 0.f;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? red[threadIdx.x] : 0.f;
        v = wave_sum(v);
        if (threadIdx.x == 0) norm_out[0] = sqrtf(v) * fabsf(grad_scale);
    }
}

// Device-side schedule (graph-capturable step: no per-step host value reaches a kernel argument). One thread advances
// the step counter and derives this step's learning rate (transformers.get_linear_schedule_with_warmup, as
// trainer.warmup_linear_lr) and Adam bias corrections into dyn[0..2] = {lr, 1-beta1^t, sqrt(1-beta2^t)}.
struct SchedArgs { int64_t* step; float base_lr, beta1, beta2; int64_t warmup, total; float* dyn; };

__global__ __launch_bounds__(1024) void norm_finish_sched_kernel(const float* partial, int n, float grad_scale,
                                                                 float* norm_out, SchedArgs sc) {
    __shared__ float red[16];
    float s = threadIdx.x < n ? partial[threadIdx.x] : 0.f;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? red[threadIdx.x] : 0.f;
        v = wave_sum(v);
        if (threadIdx.x == 0) {
            norm_out[0] = sqrtf(v) * fabsf(grad_scale);
            const int64_t t = sc.step[0] + 1;                   // 1-based optimiser step
            sc.step[0] = t;
            const int64_t k = t - 1;                            // scheduler steps already taken
            float lr = sc.base_lr;
            if (sc.total > 0) {
                if (k < sc.warmup) lr = sc.base_lr * (float)((double)k / (double)(sc.warmup > 1 ? sc.warmup : 1));
                else {
                    const int64_t den = sc.total - sc.warmup > 1 ? sc.total - sc.warmup : 1;
                    const double f = (double)(sc.total - k) / (double)den;
                    lr = sc.base_lr * (float)(f > 0.0 ? f : 0.0);
                }
            }
            sc.dyn[0] = lr;
            sc.dyn[1] = (float)(1.0 - pow((double)sc.beta1, (double)t));
            sc.dyn[2] = (float)sqrt(1.0 - pow((double)sc.beta2, (double)t));
        }
    }
}

// Mixed precision (qst_clip_adamw_step_amp, include/qst.h): torch.cuda.amp.GradScaler's unscale_ / step / update and ST fit()'s
// "skip the scheduler when the scale changed", decided by one thread from the reduced norm of the SCALED gradients.
// dyn[0..4] = {lr, 1-beta1^t, sqrt(1-beta2^t), unscale factor (grad_scale / scale), 1.0 if this step is skipped}.
struct AmpArgs { float* scaler; float growth, backoff; int interval; };

__global__ __launch_bounds__(1024) void norm_finish_amp_kernel(const float* partial, int n, float grad_scale,
                                                               float* norm_out, SchedArgs sc, AmpArgs am) {
    __shared__ float red[16];
    float s = threadIdx.x < n ? partial[threadIdx.x] : 0.f;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? red[threadIdx.x] : 0.f;
        v = wave_sum(v);
        if (threadIdx.x == 0) {
            const float scale = am.scaler[0];
            const float unscale = fabsf(grad_scale) / scale;
            const float norm = sqrtf(v) * unscale;              // inf / nan in any gradient -> not finite
            norm_out[0] = norm;
            const bool ok = fabsf(norm) < 3.0e38f;          // (false for inf and for nan)
            float new_scale = scale, tracker = am.scaler[1];
            if (!ok) {
                if (am.interval > 0) new_scale = scale * am.backoff;
                tracker = 0.f;
            } else {
                tracker += 1.f;
                if (am.interval > 0 && tracker >= (float)am.interval) { new_scale = scale * am.growth; tracker = 0.f; }
            }
            am.scaler[0] = new_scale; am.scaler[1] = tracker;
            am.scaler[2] = ok ? 0.f : 1.f;
            if (!ok) am.scaler[3] += 1.f;
            const int64_t t = sc.step[0] + (ok ? 1 : 0);       // optimiser steps (bias correction); a skipped step does not count
            sc.step[0] = t;
            const int64_t k = sc.step[1];                       // scheduler steps already taken
            if (new_scale == scale) sc.step[1] = k + 1;         // ST fit(): skip_scheduler = scaler.get_scale() != scale_before_step
            float lr = sc.base_lr;
            if (sc.total > 0) {
                if (k < sc.warmup) lr = sc.base_lr * (float)((double)k / (double)(sc.warmup > 1 ? sc.warmup : 1));
                else {
                    const int64_t den = sc.total - sc.warmup > 1 ? sc.total - sc.warmup : 1;
                    const double f = (double)(sc.total - k) / (double)den;
                    lr = sc.base_lr * (float)(f > 0.0 ? f : 0.0);
                }
            }
            const int64_t tt = t > 0 ? t : 1;
            sc.dyn[0] = lr;
            sc.dyn[1] = (float)(1.0 - pow((double)sc.beta1, (double)tt));
            sc.dyn[2] = (float)sqrt(1.0 - pow((double)sc.beta2, (double)tt));
            sc.dyn[3] = unscale;
            sc.dyn[4] = ok ? 0.f : 1.f;
        }
    }
}

__global__ void amp_scaler_init_kernel(float* sc, float init_scale) { sc[0] = init_scale; sc[1] = 0.f; sc[2] = 0.f; sc[3] = 0.f; }

struct AdamArgs {
    float* p; float* g; float* m; float* v;
    const uint8_t* chunk_decay;     // one flag per 256-element chunk of the arena
    const float* norm;              // device scalar (pre-clip global norm), or null
    const float* dyn;               // device {lr, bc1, bc2_sqrt} (device-side schedule), or null: the by-value fields
    int64_t n4;
    float lr, beta1, beta2, eps, wd, max_norm, grad_scale, bc1, bc2_sqrt;
    int amp;                        // dyn[3] = the gradient factor (grad_scale / loss scale), dyn[4] != 0: skip this step
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    if (a.dyn) { a.lr = a.dyn[0]; a.bc1 = a.dyn[1]; a.bc2_sqrt = a.dyn[2]; }
    float coef = a.grad_scale;
    if (a.amp) {
        coef = a.dyn[3];
        if (a.dyn[4] != 0.f) {      // an overflowed step (GradScaler.step skips optimizer.step()): only optimizer.zero_grad()
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4* g4 = (f32x4*)a.g;
            for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (int64_t)gridDim.x * 256) st_stream(g4 + i, z);
            return;
        }
    }
    if (a.norm && a.max_norm > 0.f) {
        const float c = a.max_norm / (a.norm[0] + 1e-6f);      // clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
        coef *= fminf(c, 1.0f);
    }
    const float step = a.lr / a.bc1;
    f32x4* p4 = (f32x4*)a.p; f32x4* g4 = (f32x4*)a.g; f32x4* m4 = (f32x4*)a.m; f32x4* v4 = (f32x4*)a.v;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (int64_t)gridDim.x * 256) {
        const float decay = a.chunk_decay[i >> 6] ? (1.0f - a.lr * a.wd) : 1.0f;   // 64 float4 per 256-element chunk
        // the moments and the (zeroed) gradients are not touched again before the next step: streamed past the caches (step 4.575 ->
        // 4.543 ms, three alternations); the parameters are read next by the bf16 refresh and stay plain (streamed: 4.600 -> 4.627)
        f32x4 p = p4[i], g = ld_stream(g4 + i), m = ld_stream(m4 + i), v = ld_stream(v4 + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = g[k] * coef;
            p[k] *= decay;
            m[k] = m[k] + (1.0f - a.beta1) * (gg - m[k]);                // exp_avg.lerp_(grad, 1-beta1)
            v[k] = a.beta2 * v[k] + (1.0f - a.beta2) * gg * gg;
            const float denom = sqrtf(v[k]) / a.bc2_sqrt + a.eps;
            p[k] -= step * (m[k] / denom);
            g[k] = 0.f;                                                  // optimizer.zero_grad()
        }
        p4[i] = p; st_stream(m4 + i, m); st_stream(v4 + i, v); st_stream(g4 + i, g);
    }
}

}  // namespace

extern "C" int qst_adamw_launch(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                const uint8_t* chunk_decay, int64_t n, float lr, float beta1, float beta2, float eps,
                                float weight_decay, float max_grad_norm, float grad_scale, int64_t step,
                                float* norm_out, float* scratch, hipStream_t st) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !chunk_decay || n <= 0 || (n & 255) || step < 1)
        return QST_ERR_BAD_ARG;
    if (max_grad_norm > 0.f || norm_out) {
        if (!norm_out || !scratch) return QST_ERR_BAD_ARG;
        sumsq_kernel<<<kNormBlocks, 256, 0, st>>>(grads, n / 4, scratch);
        QST_LAUNCH_CHECK();
        norm_finish_kernel<<<1, 1024, 0, st>>>(scratch, kNormBlocks, grad_scale, norm_out);
        QST_LAUNCH_CHECK();
    }
    AdamArgs a;
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.chunk_decay = chunk_decay;
    a.dyn = nullptr; a.amp = 0;
    a.norm = (max_grad_norm > 0.f) ? norm_out : nullptr;
    a.n4 = n / 4;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.max_norm = max_grad_norm;
    a.grad_scale = grad_scale;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    adamw_kernel<<<2048, 256, 0, st>>>(a);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_adamw_launch_sched(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                      const uint8_t* chunk_decay, int64_t n, float base_lr, float beta1, float beta2,
                                      float eps, float weight_decay, float max_grad_norm, float grad_scale,
                                      int64_t warmup_steps, int64_t total_steps, int64_t* step_dev, float* norm_out,
                                      float* scratch, hipStream_t st) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !chunk_decay || n <= 0 || (n & 255) || !step_dev || !norm_out ||
        !scratch)
        return QST_ERR_BAD_ARG;
    sumsq_kernel<<<kNormBlocks, 256, 0, st>>>(grads, n / 4, scratch);
    QST_LAUNCH_CHECK();
    SchedArgs sc;
    sc.step = step_dev; sc.base_lr = base_lr; sc.beta1 = beta1; sc.beta2 = beta2; sc.warmup = warmup_steps;
    sc.total = total_steps; sc.dyn = scratch + kNormBlocks;
    norm_finish_sched_kernel<<<1, 1024, 0, st>>>(scratch, kNormBlocks, grad_scale, norm_out, sc);
    QST_LAUNCH_CHECK();
    AdamArgs a;
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.chunk_decay = chunk_decay;
    a.dyn = sc.dyn; a.amp = 0;
    a.norm = (max_grad_norm > 0.f) ? norm_out : nullptr;
    a.n4 = n / 4;
    a.lr = 0.f; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.max_norm = max_grad_norm;
    a.grad_scale = grad_scale; a.bc1 = 1.f; a.bc2_sqrt = 1.f;
    adamw_kernel<<<2048, 256, 0, st>>>(a);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_amp_scaler_init(float* scaler_dev, float init_scale, void* stream) {
    if (!scaler_dev || !(init_scale > 0.f) || !(init_scale < 3.0e38f)) return QST_ERR_BAD_ARG;
    amp_scaler_init_kernel<<<1, 1, 0, (hipStream_t)stream>>>(scaler_dev, init_scale);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_adamw_launch_amp(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                    const uint8_t* chunk_decay, int64_t n, float base_lr, float beta1, float beta2,
                                    float eps, float weight_decay, float max_grad_norm, float grad_scale,
                                    int64_t warmup_steps, int64_t total_steps, int64_t* step_dev, float* scaler_dev,
                                    float growth, float backoff, int growth_interval, float* norm_out, float* scratch,
                                    hipStream_t st) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !chunk_decay || n <= 0 || (n & 255) || !step_dev || !scaler_dev ||
        !norm_out || !scratch || !(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f))
        return QST_ERR_BAD_ARG;
    sumsq_kernel<<<kNormBlocks, 256, 0, st>>>(grads, n / 4, scratch);
    QST_LAUNCH_CHECK();
    SchedArgs sc;
    sc.step = step_dev; sc.base_lr = base_lr; sc.beta1 = beta1; sc.beta2 = beta2; sc.warmup = warmup_steps;
    sc.total = total_steps; sc.dyn = scratch + kNormBlocks;
    AmpArgs am;
    am.scaler = scaler_dev; am.growth = growth; am.backoff = backoff; am.interval = growth_interval;
    norm_finish_amp_kernel<<<1, 1024, 0, st>>>(scratch, kNormBlocks, grad_scale, norm_out, sc, am);
    QST_LAUNCH_CHECK();
    AdamArgs a;
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.chunk_decay = chunk_decay;
    a.dyn = sc.dyn; a.amp = 1;
    a.norm = (max_grad_norm > 0.f) ? norm_out : nullptr;
    a.n4 = n / 4;
    a.lr = 0.f; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.max_norm = max_grad_norm;
    a.grad_scale = grad_scale; a.bc1 = 1.f; a.bc2_sqrt = 1.f;
    adamw_kernel<<<2048, 256, 0, st>>>(a);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
