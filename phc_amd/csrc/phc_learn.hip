// phc_learn.hip -- learner-side kernels (P1 observation normaliser).
//
// The PPO update calls the running normaliser four times per optimizer step (obs + three AMP batches); as torch ops that is
// ~40 launches per call (normalise, batch mean / var, the fp64 moment update on D-element vectors): 85 ms of a 250 ms update
// at 4096 envs (scripts/profile_step.py).  Here it is one bandwidth-bound pass + one tiny finishing block.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/phc_amd.h"

#define RN_COLS 256   // columns per block == threads per block (thread <-> column: loads coalesce across the block)
#define RN_ROWS 64    // rows per block

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
    for (int64_t r = r0; r < r1; ++r) {
        const float v = x[r * cols + c];
        if (partial) { sum += (double)v; sq += (double)v * (double)v; }
        if (out) {
            const float t = (v - m) / s;
            float y = fminf(fmaxf(t, -clampv), clampv);
            if (t != t) y = t;   // torch.clamp propagates NaN; fminf / fmaxf do not
            if (BF16) reinterpret_cast<__hip_bfloat16*>(out)[r * cols + c] = __float2bfloat16(y);
            else reinterpret_cast<float*>(out)[r * cols + c] = y;
        }
    }
    if (partial) {
        partial[((int64_t)blockIdx.y * 2 + 0) * cols + c] = sum;
        partial[((int64_t)blockIdx.y * 2 + 1) * cols + c] = sq;
    }
}

// Batch moments from the partial sums and the parallel-variance update of the running statistics
// (running_mean_std.py:56-67,100-104); ONE block, so that the old count is read by every thread before thread 0 replaces it.
__global__ __launch_bounds__(1024) void k_running_norm_finish(const double* __restrict__ partial, int nchunks, int64_t rows, int cols,
                                                              double* __restrict__ run_mean, double* __restrict__ run_var,
                                                              double* __restrict__ run_count) {
    const double count = *run_count;
    const double n = (double)rows;
    const double tot = count + n;
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < nchunks; ++k) {
            s += partial[((int64_t)k * 2 + 0) * cols + c];
            q += partial[((int64_t)k * 2 + 1) * cols + c];
        }
        // input.mean(0), input.var(0) are fp32 tensors in the reference: round the batch moments to fp32 before the fp64 update
        const double bm = (double)(float)(s / n);
        const double bv = (double)(float)((q - s * s / n) / (n - 1.0));
        const double mean = run_mean[c], var = run_var[c];
        const double delta = bm - mean;
        run_mean[c] = mean + delta * n / tot;
        run_var[c] = (var * count + bv * n + delta * delta * count * n / tot) / tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) *run_count = tot;
}

extern "C" {

int64_t phc_running_norm_workspace(int64_t rows, int32_t cols) {
    return ((rows + RN_ROWS - 1) / RN_ROWS) * 2 * (int64_t)cols * (int64_t)sizeof(double);
}

int32_t phc_running_norm(const float* x, int64_t rows, int32_t cols, const double* norm_mean, const double* norm_var, float epsilon,
                         float clamp, void* out, int32_t out_bf16, double* run_mean, double* run_var, double* run_count,
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
        hipLaunchKernelGGL(k_running_norm_finish, dim3(1), dim3(1024), 0, st, workspace, (int)nchunks, rows, cols, run_mean, run_var, run_count);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

}  // extern "C"
