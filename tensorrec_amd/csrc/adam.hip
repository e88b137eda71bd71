// tensorrec_amd/csrc/adam.hip -- K8: fused L2-regularisation gradient + TensorFlow-1.x-form Adam.
//
// Replaces tf.train.AdamOptimizer(lr).minimize(tf_loss) (tensorrec/tensorrec.py:489) together with the gradient of
// alpha * sum(tf.nn.l2_loss(w)) (tensorrec.py:487-488).  DENSE on purpose: TF's gradient of
// sparse_tensor_dense_matmul w.r.t. the weights is a dense [F, d] tensor, so every row's moments decay every step
// (SURVEY.md 3.4).  One pass: read w, m, v, g; write w, m, v (28 B per element; HBM-bound).
//
// Arithmetic follows the TF CPU functor element by element [external: TF 1.x training_ops ApplyAdam]:
//     g' = g + w * l2_coef
//     m += (g' - m) * (1 - beta1);   v += (g'*g' - v) * (1 - beta2);   w -= (m * lr_t) / (sqrt(v) + eps)
// with lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) computed by the caller.  Every operation is an individually
// rounded fp32 op (__f*_rn: no FMA contraction), so the update is bit-identical to oracle.adam_tf_step.
#include "common.hpp"

// HIP's __f*_rn intrinsics are plain operators, which hipcc would contract into FMAs (-ffp-contract=fast is the
// HIP default and ignores contract pragmas); the TF functor and the oracle round every operation separately, so this
// file is compiled with -ffp-contract=off (see the Makefile).

__device__ __forceinline__ void adam_elem(float& w, float& m, float& v, float g, float lr_t, float omb1, float omb2,
                                          float eps, float l2)
{
    const float gg = (l2 != 0.f) ? __fadd_rn(g, __fmul_rn(w, l2)) : g;
    m = __fadd_rn(m, __fmul_rn(__fsub_rn(gg, m), omb1));
    v = __fadd_rn(v, __fmul_rn(__fsub_rn(__fmul_rn(gg, gg), v), omb2));
    w = __fsub_rn(w, __fdiv_rn(__fmul_rn(m, lr_t), __fadd_rn(sqrtf(v), eps)));
}

__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                     const float* __restrict__ g, int64_t n, float lr_t, float beta1,
                                                     float beta2, float eps, float l2,
                                                     const float* __restrict__ sched = nullptr)
{
    if (sched) lr_t = sched[2];                 // schedule state on the device (HIP-graph replays)
    const float omb1 = __fsub_rn(1.0f, beta1), omb2 = __fsub_rn(1.0f, beta2);
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 ww = ((f32x4*)w)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        const f32x4 gg = ((const f32x4*)g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float we = ww[e], me = mm[e], ve = vv[e];
            adam_elem(we, me, ve, gg[e], lr_t, omb1, omb2, eps, l2);
            ww[e] = we; mm[e] = me; vv[e] = ve;
        }
        ((f32x4*)w)[i] = ww; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float ww = w[i], mm = m[i], vv = v[i];
        adam_elem(ww, mm, vv, g[i], lr_t, omb1, omb2, eps, l2);
        w[i] = ww; m[i] = mm; v[i] = vv;
    }
}

extern "C" int trec_adam_tf_step(float* w, float* m, float* v, const float* grad, int64_t n, float lr_t, float beta1,
                                 float beta2, float epsilon, float l2_coef, void* stream)
{
    TREC_REQUIRE(w && m && v && grad, "trec_adam_tf_step: null pointer");
    if (n == 0) return TREC_OK;
    int64_t blocks = ceil_div64(ceil_div64(n, 4), 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, m, v, grad, n, lr_t,
                       beta1, beta2, epsilon, l2_coef);
    return trec_check_launch("trec_adam_tf_step");
}

// ---- the schedule on the device ------------------------------------------------------------------------------------
// state = { beta1_power, beta2_power, lr_t, sample step (uint32 bits) }.  TF keeps the beta powers as float32 variables
// multiplied by beta once per step and forms lr_t = lr * sqrt(1 - b2p) / (1 - b1p) from them; doing exactly that in a
// one-thread kernel (every op rounded separately, this file is built with -ffp-contract=off) gives the host's lr_t bit
// for bit, and lets optimiser and sampler launches live inside a HIP graph whose replays advance their own counters.
__global__ void adam_schedule_kernel(float* __restrict__ state, float lr, float beta1, float beta2, int bump_sample_step)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float b1p = __fmul_rn(state[0], beta1), b2p = __fmul_rn(state[1], beta2);
    state[0] = b1p;
    state[1] = b2p;
    state[2] = __fdiv_rn(__fmul_rn(lr, sqrtf(__fsub_rn(1.0f, b2p))), __fsub_rn(1.0f, b1p));
    if (bump_sample_step) ((unsigned int*)state)[3] += 1u;
}

extern "C" int trec_adam_schedule_advance(float* state, float learning_rate, float beta1, float beta2,
                                          int32_t bump_sample_step, void* stream)
{
    TREC_REQUIRE(state, "trec_adam_schedule_advance: null pointer");
    hipLaunchKernelGGL(adam_schedule_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, learning_rate, beta1, beta2,
                       bump_sample_step);
    return trec_check_launch("trec_adam_schedule_advance");
}

extern "C" int trec_adam_tf_step_dev(float* w, float* m, float* v, const float* grad, int64_t n, const float* state,
                                     float beta1, float beta2, float epsilon, float l2_coef, void* stream)
{
    TREC_REQUIRE(w && m && v && grad && state, "trec_adam_tf_step_dev: null pointer");
    if (n == 0) return TREC_OK;
    int64_t blocks = ceil_div64(ceil_div64(n, 4), 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, m, v, grad, n, 0.f,
                       beta1, beta2, epsilon, l2_coef, state);
    return trec_check_launch("trec_adam_tf_step_dev");
}
