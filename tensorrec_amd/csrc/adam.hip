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

// V (tuning "adam_variant", measured in profiles/r05_adam_ab.txt): bit 0 = non-temporal stores, bit 1 = non-temporal loads,
// bit 2 = two float4 per array in flight per thread.
template <int V>
__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                     const float* __restrict__ g, int64_t n, float lr_t, float beta1,
                                                     float beta2, float eps, float l2,
                                                     const float* __restrict__ sched = nullptr)
{
    if (sched) lr_t = sched[2];                 // schedule state on the device (HIP-graph replays)
    const float omb1 = __fsub_rn(1.0f, beta1), omb2 = __fsub_rn(1.0f, beta2);
    const int64_t n4 = n >> 2;
    constexpr int UN = (V & 4) ? 2 : 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * UN) {
        f32x4 ww[UN], mm[UN], vv[UN], gg[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int64_t i = i0 + q * stride;
            if (i < n4) {
                if (V & 2) {
                    ww[q] = __builtin_nontemporal_load((f32x4*)w + i); mm[q] = __builtin_nontemporal_load((f32x4*)m + i);
                    vv[q] = __builtin_nontemporal_load((f32x4*)v + i); gg[q] = __builtin_nontemporal_load((const f32x4*)g + i);
                } else {
                    ww[q] = ((f32x4*)w)[i]; mm[q] = ((f32x4*)m)[i]; vv[q] = ((f32x4*)v)[i]; gg[q] = ((const f32x4*)g)[i];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int64_t i = i0 + q * stride;
            if (i < n4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float we = ww[q][e], me = mm[q][e], ve = vv[q][e];
                    adam_elem(we, me, ve, gg[q][e], lr_t, omb1, omb2, eps, l2);
                    ww[q][e] = we; mm[q][e] = me; vv[q][e] = ve;
                }
                if (V & 1) {
                    __builtin_nontemporal_store(ww[q], (f32x4*)w + i); __builtin_nontemporal_store(mm[q], (f32x4*)m + i);
                    __builtin_nontemporal_store(vv[q], (f32x4*)v + i);
                } else {
                    ((f32x4*)w)[i] = ww[q]; ((f32x4*)m)[i] = mm[q]; ((f32x4*)v)[i] = vv[q];
                }
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float ww = w[i], mm = m[i], vv = v[i];
        adam_elem(ww, mm, vv, g[i], lr_t, omb1, omb2, eps, l2);
        w[i] = ww; m[i] = mm; v[i] = vv;
    }
}

static int adam_launch(float* w, float* m, float* v, const float* grad, int64_t n, float lr_t, float beta1, float beta2,
                       float epsilon, float l2_coef, const float* state, hipStream_t st)
{
    int64_t blocks = ceil_div64(ceil_div64(n, 4), 256);
    const int cap = trec_get_tuning("adam_blocks", 1 << 20);     // (a 4,096-workgroup grid-stride loop: 0.745 ms per 1M x 128 table; one float4 per thread: 0.615)
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
#define TREC_ADAM(VV) hipLaunchKernelGGL(adam_tf_kernel<VV>, dim3((unsigned)blocks), dim3(256), 0, st, w, m, v, grad, n, lr_t, beta1, \
                                         beta2, epsilon, l2_coef, state)
    switch (trec_get_tuning("adam_variant", 7)) {
    case 1: TREC_ADAM(1); break;
    case 2: TREC_ADAM(2); break;
    case 3: TREC_ADAM(3); break;
    case 4: TREC_ADAM(4); break;
    case 5: TREC_ADAM(5); break;
    case 0: TREC_ADAM(0); break;
    default: TREC_ADAM(7); break;
    }
#undef TREC_ADAM
    return TREC_OK;
}

extern "C" int trec_adam_tf_step(float* w, float* m, float* v, const float* grad, int64_t n, float lr_t, float beta1,
                                 float beta2, float epsilon, float l2_coef, void* stream)
{
    TREC_REQUIRE(w && m && v && grad, "trec_adam_tf_step: null pointer");
    if (n == 0) return TREC_OK;
    adam_launch(w, m, v, grad, n, lr_t, beta1, beta2, epsilon, l2_coef, nullptr, (hipStream_t)stream);
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
    adam_launch(w, m, v, grad, n, 0.f, beta1, beta2, epsilon, l2_coef, state, (hipStream_t)stream);
    return trec_check_launch("trec_adam_tf_step_dev");
}
