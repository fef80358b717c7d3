// phc_learn.hip -- learner-side kernels (P1 observation normaliser).
//
// The PPO update calls the running normaliser four times per optimizer step (obs + three AMP batches); as torch ops that is
// ~40 launches per call (normalise, batch mean / var, the fp64 moment update on D-element vectors): 85 ms of a 250 ms update
// at 4096 envs (scripts/profile_step.py).  Here it is one bandwidth-bound pass + one tiny finishing block.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>
#include "../../include/phc_amd.h"

#define RN_COLS 256   // columns per block == threads per block (thread <-> column: loads coalesce across the block)
#define RN_ROWS 128   // rows per block
#define RN_UNROLL 8   // independent loads in flight per thread

// One pass over x [rows, cols]: y = clamp((x - mean) / sqrt(var + eps), -c, c)  (running_mean_std.py:95-96, fp32 like the reference:
// the fp64 statistics are rounded to fp32 first) and, when `partial` is given, per-block column sums of x and x^2 in fp64.
template <bool BF16>
__global__ __launch_bounds__(RN_COLS) void k_running_norm(const float* __restrict__ x, int64_t rows, int cols, const double* __restrict__ mean,
                                                          const double* __restrict__ var, float eps, float clampv, void* __restrict__ out,
                                                          double* __restrict__ partial) {
    const int c = blockIdx.x * RN_COLS + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * RN_ROWS;
    const int64_t r1 = r0 + RN_ROWS < rows ? r0 + RN_ROWS : rows;
    const float m = (float)mean[c];
    const float s = sqrtf((float)var[c] + eps);
    double sum = 0.0, sq = 0.0;
    auto one = [&](int64_t r, float v) {
        if (partial) { sum += (double)v; sq += (double)v * (double)v; }
        if (out) {
            const float t = (v - m) / s;
            float y = fminf(fmaxf(t, -clampv), clampv);
            if (t != t) y = t;   // torch.clamp propagates NaN; fminf / fmaxf do not
            if (BF16) reinterpret_cast<__hip_bfloat16*>(out)[r * cols + c] = __float2bfloat16(y);
            else reinterpret_cast<float*>(out)[r * cols + c] = y;
        }
    };
    int64_t r = r0;
    for (; r + RN_UNROLL <= r1; r += RN_UNROLL) {
        float v[RN_UNROLL];
#pragma unroll
        for (int k = 0; k < RN_UNROLL; ++k) v[k] = x[(r + k) * cols + c];
#pragma unroll
        for (int k = 0; k < RN_UNROLL; ++k) one(r + k, v[k]);
    }
    for (; r < r1; ++r) one(r, x[r * cols + c]);
    if (partial) {
        partial[((int64_t)blockIdx.y * 2 + 0) * cols + c] = sum;
        partial[((int64_t)blockIdx.y * 2 + 1) * cols + c] = sq;
    }
}

// Batch moments from the partial sums and the parallel-variance update of the running statistics
// (running_mean_std.py:56-67,100-104).  Block = 64 columns x 16 slices of the chunk list; `run_count` is only READ (the caller
// adds the batch size afterwards, stream-ordered), so blocks need no ordering among themselves.
__global__ __launch_bounds__(1024) void k_running_norm_finish(const double* __restrict__ partial, int nchunks, int64_t rows, int cols,
                                                              double* __restrict__ run_mean, double* __restrict__ run_var,
                                                              const double* __restrict__ run_count) {
    __shared__ double ls[16][64], lq[16][64];
    const int cx = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double s = 0.0, q = 0.0;
    if (c < cols)
        for (int k = sy; k < nchunks; k += 16) {
            s += partial[((int64_t)k * 2 + 0) * cols + c];
            q += partial[((int64_t)k * 2 + 1) * cols + c];
        }
    ls[sy][cx] = s; lq[sy][cx] = q;
    __syncthreads();
    if (sy != 0 || c >= cols) return;
    s = 0.0; q = 0.0;
    for (int k = 0; k < 16; ++k) { s += ls[k][cx]; q += lq[k][cx]; }
    const double count = *run_count;
    const double n = (double)rows;
    const double tot = count + n;
    // input.mean(0), input.var(0) are fp32 tensors in the reference: round the batch moments to fp32 before the fp64 update
    const double bm = (double)(float)(s / n);
    const double bv = (double)(float)((q - s * s / n) / (n - 1.0));
    const double mean = run_mean[c], var = run_var[c];
    const double delta = bm - mean;
    run_mean[c] = mean + delta * n / tot;
    run_var[c] = (var * count + bv * n + delta * delta * count * n / tot) / tot;
}

// ------------------------------------------------------------------------------------------
// Column sums of a bf16 matrix [rows, cols] -> fp32 [cols]: the bias gradient of a linear layer (torch's generic reduce takes
// 25 us for 16384 x 1024 and 95 us for 16384 x 69).  Two deterministic stages: 64 columns x 4 row lanes per block over a
// 256-row chunk, then one sum over the chunks.
// ------------------------------------------------------------------------------------------
#define CS_ROWS 256
__global__ __launch_bounds__(256) void k_colsum_bf16(const __hip_bfloat16* __restrict__ x, int64_t rows, int cols, float* __restrict__ partial) {
    __shared__ float l[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    const int64_t r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
    float a = 0.f;
    if (c < cols) {
        int64_t r = r0 + ry;
        for (; r + 28 < r1; r += 32) {   // 8 independent loads in flight
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __bfloat162float(x[(r + 4 * k) * cols + c]);
#pragma unroll
            for (int k = 0; k < 8; ++k) a += v[k];
        }
        for (; r < r1; r += 4) a += __bfloat162float(x[r * cols + c]);
    }
    l[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && c < cols) partial[(int64_t)blockIdx.y * cols + c] = (l[0][cx] + l[1][cx]) + (l[2][cx] + l[3][cx]);
}
__global__ void k_colsum_finish(const float* __restrict__ partial, int nchunks, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float a = 0.f;
    for (int k = 0; k < nchunks; ++k) a += partial[(int64_t)k * cols + c];
    out[c] = a;
}

// ------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam on the flat fp32 parameter (FlatGradBucket): torch.nn.utils.clip_grad_norm_ (coefficient
// min(1, max_norm / (|g| + 1e-6))) followed by torch.optim.Adam's update (L2 weight decay, bias corrections, eps outside the
// square root of the corrected second moment) -- two launches over 4 arrays instead of norm + scale + fused multi-tensor Adam.
// ------------------------------------------------------------------------------------------
#define AD_BLOCK 256
#define AD_PER_THREAD 8
__global__ __launch_bounds__(AD_BLOCK) void k_sumsq(const float* __restrict__ g, int64_t n, double* __restrict__ partial) {
    __shared__ double l[AD_BLOCK / 64];
    const int64_t base = ((int64_t)blockIdx.x * AD_BLOCK + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * AD_BLOCK * 4;
    float a = 0.f;
    for (int64_t i = base; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
            for (int64_t k = i; k < n; ++k) a += g[k] * g[k];
        }
    }
    double d = (double)a;
    for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m, 64);
    if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < AD_BLOCK / 64; ++k) t += l[k];
        partial[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(AD_BLOCK) void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, float bias1,
                                                   float bias2_sqrt, float max_norm, const double* __restrict__ partial, int npartial,
                                                   float* __restrict__ norm_out) {
    float coef = 1.f;
    if (max_norm > 0.f) {   // every block re-reduces the npartial (512) L2-resident partial sums: cheaper than a third launch
        __shared__ double l[AD_BLOCK / 64];
        double t = 0.0;
        for (int k = threadIdx.x; k < npartial; k += AD_BLOCK) t += partial[k];
        for (int m2 = 32; m2 >= 1; m2 >>= 1) t += __shfl_xor(t, m2, 64);
        if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6] = t;
        __syncthreads();
        t = 0.0;
        for (int k = 0; k < AD_BLOCK / 64; ++k) t += l[k];
        const float total = (float)sqrt(t);
        coef = fminf(max_norm / (total + 1e-6f), 1.0f);
        if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total;
    }
    const float step_size = lr / bias1;
    const int64_t i0 = ((int64_t)blockIdx.x * AD_BLOCK + threadIdx.x) * 4;
    if (i0 >= n) return;
    const int cnt = n - i0 >= 4 ? 4 : (int)(n - i0);
    for (int k = 0; k < cnt; ++k) {
        const int64_t i = i0 + k;
        float gi = g[i] * coef;
        g[i] = gi;                                    // clip_grad_norm_ scales the gradient in place
        const float pi = p[i];
        if (weight_decay != 0.f) gi += weight_decay * pi;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;      // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrtf(vi) / bias2_sqrt + eps));
    }
}

extern "C" {

int64_t phc_running_norm_workspace(int64_t rows, int32_t cols) {
    return ((rows + RN_ROWS - 1) / RN_ROWS) * 2 * (int64_t)cols * (int64_t)sizeof(double);
}

int32_t phc_running_norm(const float* x, int64_t rows, int32_t cols, const double* norm_mean, const double* norm_var, float epsilon,
                         float clamp, void* out, int32_t out_bf16, double* run_mean, double* run_var, const double* run_count,
                         double* workspace, void* stream) {
    if (!x || rows < 0 || cols < 1 || !norm_mean || !norm_var) return PHC_EINVAL;
    const bool update = run_mean != nullptr;
    if (update && (!run_var || !run_count || !workspace)) return PHC_EINVAL;
    if (!update && !out) return PHC_EINVAL;
    if (rows == 0) return 0;
    const int64_t nchunks = (rows + RN_ROWS - 1) / RN_ROWS;
    if (nchunks > 65535) return PHC_EUNSUPPORTED;
    const dim3 grid((cols + RN_COLS - 1) / RN_COLS, (unsigned)nchunks);
    hipStream_t st = (hipStream_t)stream;
    if (out_bf16)
        hipLaunchKernelGGL(k_running_norm<true>, grid, dim3(RN_COLS), 0, st, x, rows, cols, norm_mean, norm_var, epsilon, clamp, out, update ? workspace : nullptr);
    else
        hipLaunchKernelGGL(k_running_norm<false>, grid, dim3(RN_COLS), 0, st, x, rows, cols, norm_mean, norm_var, epsilon, clamp, out, update ? workspace : nullptr);
    if (update)
        hipLaunchKernelGGL(k_running_norm_finish, dim3((cols + 63) / 64), dim3(1024), 0, st, workspace, (int)nchunks, rows, cols, run_mean, run_var, run_count);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

int64_t phc_colsum_workspace(int64_t rows, int32_t cols) { return ((rows + CS_ROWS - 1) / CS_ROWS) * (int64_t)cols * (int64_t)sizeof(float); }

int32_t phc_colsum_bf16(const void* x, int64_t rows, int32_t cols, float* out, float* workspace, void* stream) {
    if (!x || !out || !workspace || rows < 1 || cols < 1) return PHC_EINVAL;
    const int64_t nchunks = (rows + CS_ROWS - 1) / CS_ROWS;
    if (nchunks > 65535) return PHC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_colsum_bf16, dim3((cols + 63) / 64, (unsigned)nchunks), dim3(256), 0, st, reinterpret_cast<const __hip_bfloat16*>(x), rows, cols, workspace);
    hipLaunchKernelGGL(k_colsum_finish, dim3((cols + 255) / 256), dim3(256), 0, st, workspace, (int)nchunks, cols, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

#define AD_NORM_BLOCKS 512
int64_t phc_adam_workspace(void) { return AD_NORM_BLOCKS * (int64_t)sizeof(double); }

int32_t phc_adam_clip_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int64_t step, float max_norm, double* workspace, float* grad_norm_out, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1 || (max_norm > 0.f && !workspace)) return PHC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (max_norm > 0.f) hipLaunchKernelGGL(k_sumsq, dim3(AD_NORM_BLOCKS), dim3(AD_BLOCK), 0, st, grad, n, workspace);
    const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t blocks = (n + AD_BLOCK * 4 - 1) / (AD_BLOCK * 4);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(AD_BLOCK), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                       (float)b1, (float)sqrt(b2), max_norm, workspace, AD_NORM_BLOCKS, grad_norm_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

}  // extern "C"
