// loss.hip -- fused gamma-quadruplet loss forward + backward (one wave64 per quadruplet).
//
// Replaces /root/reference/models/losses/losses.py:9-69 (gamma_quadruplet_loss) and the
// ~60 small torch kernels autograd runs for it:
//   a = TML(A,P,N; m_pn)  b = TML(A,Q,N; m_qn)  c = TML(A,P,Q; m_pq)
//   TML(x,y,z;m) = max(m + ||x-y+eps||_p - dneg, 0),  dneg = ||x-z+eps||_p, swap: min(dneg, ||y-z+eps||_p)
//   row = a + gamma*b + (1-gamma)*c
// HBM-bound: algorithmic bytes = 4*D*4 read per row (+ 4*D*4 written with gradients). The four
// rows are read once with 16-byte coalesced loads and kept in registers for the gradient pass.
#include "qst_common.h"

namespace {

constexpr float kEps = 1e-6f;       // torch pairwise_distance default eps
constexpr int kMaxVecAll = 8;       // float4 per lane kept in registers -> D <= 64*4*8 = 2048

struct LossArgs {
    const float* x[4];              // A, P, Q, N
    float* g[4];                    // grads (may be null)
    const float* grad_out;          // upstream (may be null)
    float* row_out;                 // [B] per-row loss (out_loss for 'none', scratch otherwise)
    int B, D;
    float gamma, m_pn, m_pq, m_qn, p;
    int swap, reduction;
};

template <int PMODE>  // 1: p==1, 2: p==2, 0: general p
__device__ __forceinline__ float pw(float v, float p) {
    float a = fabsf(v);
    if (PMODE == 2) return v * v;
    if (PMODE == 1) return a;
    return a > 0.f ? __powf(a, p) : 0.f;
}
template <int PMODE>
__device__ __forceinline__ float root(float s, float p) {
    if (PMODE == 2) return sqrtf(s);
    if (PMODE == 1) return s;
    return s > 0.f ? __powf(s, 1.0f / p) : 0.f;
}
// d||v||_p / dv_i given norm d (0 where d == 0, as torch's norm backward masks)
template <int PMODE>
__device__ __forceinline__ float dnorm(float v, float d, float p) {
    if (d == 0.f) return 0.f;
    if (PMODE == 2) return v / d;
    if (PMODE == 1) return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
    float a = fabsf(v);
    if (a == 0.f) return 0.f;
    float s = v > 0.f ? 1.f : -1.f;
    return s * __powf(a, p - 1.f) / __powf(d, p - 1.f);
}

// pair indices: 0 AP, 1 AN, 2 AQ, 3 PN, 4 QN, 5 PQ  (x1 - x2 + eps, in torch's argument order)
// NV: float4 per lane kept in registers (rows of up to 256*NV floats); 0 = scalar path for odd shapes. Sizing the
// arrays for the row at hand instead of the 2048-float maximum takes the D = 384 kernel from 194 to ~70 VGPRs:
// the kernel is latency-bound, waves per SIMD is what feeds HBM.
template <int PMODE, int NV>
__global__ __launch_bounds__(256) void quad_loss_kernel(LossArgs a) {
    constexpr bool VEC = NV > 0;
    constexpr int kMaxVec = NV > 0 ? NV : 1;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.B) return;
    const int D = a.D;
    const size_t base = (size_t)row * D;

    float xa[VEC ? kMaxVec * 4 : 1], xp[VEC ? kMaxVec * 4 : 1], xq[VEC ? kMaxVec * 4 : 1], xn[VEC ? kMaxVec * 4 : 1];
    float s[6] = {0, 0, 0, 0, 0, 0};
    const bool need_swap = a.swap != 0;

    if (VEC) {
        const int nv = D >> 2;
#pragma unroll
        for (int i = 0; i < kMaxVec; ++i) {
            const int v = lane + i * 64;
            f32x4 A = {0, 0, 0, 0}, P = A, Q = A, N = A;
            const bool in = v < nv;
            if (in) {
                A = *(const f32x4*)(a.x[0] + base + v * 4);
                P = *(const f32x4*)(a.x[1] + base + v * 4);
                Q = *(const f32x4*)(a.x[2] + base + v * 4);
                N = *(const f32x4*)(a.x[3] + base + v * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xa[i * 4 + j] = A[j]; xp[i * 4 + j] = P[j]; xq[i * 4 + j] = Q[j]; xn[i * 4 + j] = N[j];
                if (in) {
                    s[0] += pw<PMODE>(A[j] - P[j] + kEps, a.p);
                    s[1] += pw<PMODE>(A[j] - N[j] + kEps, a.p);
                    s[2] += pw<PMODE>(A[j] - Q[j] + kEps, a.p);
                    if (need_swap) {
                        s[3] += pw<PMODE>(P[j] - N[j] + kEps, a.p);
                        s[4] += pw<PMODE>(Q[j] - N[j] + kEps, a.p);
                        s[5] += pw<PMODE>(P[j] - Q[j] + kEps, a.p);
                    }
                }
            }
        }
    } else {
        for (int i = lane; i < D; i += 64) {
            const float A = a.x[0][base + i], P = a.x[1][base + i], Q = a.x[2][base + i], N = a.x[3][base + i];
            s[0] += pw<PMODE>(A - P + kEps, a.p);
            s[1] += pw<PMODE>(A - N + kEps, a.p);
            s[2] += pw<PMODE>(A - Q + kEps, a.p);
            if (need_swap) {
                s[3] += pw<PMODE>(P - N + kEps, a.p);
                s[4] += pw<PMODE>(Q - N + kEps, a.p);
                s[5] += pw<PMODE>(P - Q + kEps, a.p);
            }
        }
    }
    float d[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = (k < 3 || need_swap) ? root<PMODE>(wave_sum(s[k]), a.p) : 0.f;

    // negative distance per term and how its gradient splits between the direct and the swapped pair
    // (torch.min ties send half the gradient to each side)
    float wd[3], ws[3];
    const int dir[3] = {1, 1, 2};     // direct negative pair: AN, AN, AQ
    const int swp[3] = {3, 4, 5};     // swapped pair:        PN, QN, PQ
    const int pos[3] = {0, 2, 0};     // positive pair:       AP, AQ, AP
    const float marg[3] = {a.m_pn, a.m_qn, a.m_pq};
    const float wt[3] = {1.f, a.gamma, 1.f - a.gamma};
    float act[3];
    float rowloss = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float dneg = d[dir[t]];
        wd[t] = 1.f; ws[t] = 0.f;
        if (need_swap) {
            const float dsw = d[swp[t]];
            if (dsw < dneg) { wd[t] = 0.f; ws[t] = 1.f; dneg = dsw; }
            else if (dsw == dneg) { wd[t] = 0.5f; ws[t] = 0.5f; }
        }
        const float h = marg[t] + d[pos[t]] - dneg;
        act[t] = (h >= 0.f) ? 1.f : 0.f;      // clamp_min backward passes the gradient at h == 0
        const float term = fmaxf(h, 0.f);
        // same association as the reference: a + gamma*b + (1-gamma)*c
        rowloss = (t == 0) ? term : rowloss + wt[t] * term;
    }
    if (lane == 0) a.row_out[row] = rowloss;
    if (a.g[0] == nullptr) return;

    // upstream gradient per row
    float up = 1.f;
    if (a.grad_out) up = (a.reduction == QST_REDUCE_NONE) ? a.grad_out[row] : a.grad_out[0];
    if (a.reduction == QST_REDUCE_MEAN) up /= (float)a.B;
    // coefficient of each pair's norm in the row loss
    float c[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float w = up * wt[t] * act[t];
        c[pos[t]] += w;
        c[dir[t]] -= w * wd[t];
        c[swp[t]] -= w * ws[t];
    }
    auto emit = [&](float A, float P, float Q, float N, float& gA, float& gP, float& gQ, float& gN) {
        const float e0 = c[0] * dnorm<PMODE>(A - P + kEps, d[0], a.p);
        const float e1 = c[1] * dnorm<PMODE>(A - N + kEps, d[1], a.p);
        const float e2 = c[2] * dnorm<PMODE>(A - Q + kEps, d[2], a.p);
        float e3 = 0.f, e4 = 0.f, e5 = 0.f;
        if (need_swap) {
            e3 = c[3] * dnorm<PMODE>(P - N + kEps, d[3], a.p);
            e4 = c[4] * dnorm<PMODE>(Q - N + kEps, d[4], a.p);
            e5 = c[5] * dnorm<PMODE>(P - Q + kEps, d[5], a.p);
        }
        gA = e0 + e1 + e2;
        gP = -e0 + e3 + e5;
        gQ = -e2 + e4 - e5;
        gN = -e1 - e3 - e4;
    };
    if (VEC) {
        const int nv = D >> 2;
#pragma unroll
        for (int i = 0; i < kMaxVec; ++i) {
            const int v = lane + i * 64;
            if (v < nv) {
                f32x4 gA, gP, gQ, gN;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float ga, gp, gq, gn;
                    emit(xa[i * 4 + j], xp[i * 4 + j], xq[i * 4 + j], xn[i * 4 + j], ga, gp, gq, gn);
                    gA[j] = ga; gP[j] = gp; gQ[j] = gq; gN[j] = gn;
                }
                *(f32x4*)(a.g[0] + base + v * 4) = gA;
                *(f32x4*)(a.g[1] + base + v * 4) = gP;
                *(f32x4*)(a.g[2] + base + v * 4) = gQ;
                *(f32x4*)(a.g[3] + base + v * 4) = gN;
            }
        }
    } else {
        for (int i = lane; i < D; i += 64) {
            float ga, gp, gq, gn;
            emit(a.x[0][base + i], a.x[1][base + i], a.x[2][base + i], a.x[3][base + i], ga, gp, gq, gn);
            a.g[0][base + i] = ga; a.g[1][base + i] = gp; a.g[2][base + i] = gq; a.g[3][base + i] = gn;
        }
    }
}

// Deterministic second stage for 'sum'/'mean': one block, fixed summation tree.
__global__ __launch_bounds__(1024) void quad_loss_reduce_kernel(const float* rows, int B, float scale, float* out) {
    __shared__ float part[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 1024) s += rows[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? part[threadIdx.x] : 0.f;
        v = wave_sum(v);
        if (threadIdx.x == 0) out[0] = v * scale;
    }
}

template <int PMODE>
void launch_loss(const LossArgs& a, bool vec, hipStream_t st) {
    const int grid = (a.B + 3) / 4;
    const int nv = (a.D + 255) / 256;
    if (!vec) quad_loss_kernel<PMODE, 0><<<grid, 256, 0, st>>>(a);
    else if (nv <= 1) quad_loss_kernel<PMODE, 1><<<grid, 256, 0, st>>>(a);
    else if (nv <= 2) quad_loss_kernel<PMODE, 2><<<grid, 256, 0, st>>>(a);
    else if (nv <= 3) quad_loss_kernel<PMODE, 3><<<grid, 256, 0, st>>>(a);
    else if (nv <= 4) quad_loss_kernel<PMODE, 4><<<grid, 256, 0, st>>>(a);
    else quad_loss_kernel<PMODE, 8><<<grid, 256, 0, st>>>(a);
}

}  // namespace

extern "C" int qst_quadruplet_loss(const float* xa, const float* xp, const float* xq, const float* xn,
                                   int B, int D, float gamma, float margin_pos_neg, float margin_pos_part,
                                   float margin_part_neg, float p, int swap, int reduction,
                                   float* out_loss, const float* grad_out,
                                   float* grad_a, float* grad_p, float* grad_q, float* grad_n,
                                   float* scratch, void* stream) {
    if (!xa || !xp || !xq || !xn || !out_loss || B <= 0 || D <= 0) return QST_ERR_BAD_ARG;
    if (reduction < QST_REDUCE_NONE || reduction > QST_REDUCE_MEAN) return QST_ERR_BAD_ARG;
    if (reduction != QST_REDUCE_NONE && !scratch) return QST_ERR_BAD_ARG;
    // same domain checks as losses.py:20-32 (the Python shim raises ValueError first; this is the ABI guard)
    if (!(gamma >= 0.f && gamma <= 1.f) || !(margin_pos_neg > 0.f) || !(margin_pos_part > 0.f) ||
        !(margin_part_neg > 0.f) || !(p > 0.f))
        return QST_ERR_BAD_ARG;
    const bool any_g = grad_a || grad_p || grad_q || grad_n;
    if (any_g && !(grad_a && grad_p && grad_q && grad_n)) return QST_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    LossArgs a;
    a.x[0] = xa; a.x[1] = xp; a.x[2] = xq; a.x[3] = xn;
    a.g[0] = grad_a; a.g[1] = grad_p; a.g[2] = grad_q; a.g[3] = grad_n;
    a.grad_out = grad_out;
    a.row_out = (reduction == QST_REDUCE_NONE) ? out_loss : scratch;
    a.B = B; a.D = D; a.gamma = gamma; a.m_pn = margin_pos_neg; a.m_pq = margin_pos_part; a.m_qn = margin_part_neg;
    a.p = p; a.swap = swap; a.reduction = reduction;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    bool vec = (D % 4 == 0) && D <= 64 * 4 * kMaxVecAll && al16(xa) && al16(xp) && al16(xq) && al16(xn);
    if (any_g) vec = vec && al16(grad_a) && al16(grad_p) && al16(grad_q) && al16(grad_n);
    if (p == 2.0f) launch_loss<2>(a, vec, st);
    else if (p == 1.0f) launch_loss<1>(a, vec, st);
    else launch_loss<0>(a, vec, st);
    QST_LAUNCH_CHECK();
    if (reduction != QST_REDUCE_NONE) {
        const float scale = (reduction == QST_REDUCE_MEAN) ? 1.0f / (float)B : 1.0f;
        quad_loss_reduce_kernel<<<1, 1024, 0, st>>>(scratch, B, scale, out_loss);
        QST_LAUNCH_CHECK();
    }
    return QST_OK;
}
