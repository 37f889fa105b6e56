// slices.hpp -- the one place where a policy logit becomes a probability in the sliced heads.
//
// For batches of at most crl_heads_set_sliced_max boards the policy head (model.py:44-48: Dense(1968,
// softmax)) runs as 8 label slices (csrc/heads.hpp: k_heads_sliced) that leave, per board,
//     logits                              and
//     stats[slice] = (m = max logit of the slice, s = sum exp(logit - m) over the slice).
// A probability is then  exp(l - M) / S  with  M = max_k m_k,  S = sum_k s_k exp(m_k - M)  (k = 0..7 in
// that order).  Whoever turns a logit into a probability -- the normalising pass over full policy
// vectors (k_policy_normalise) or the search kernels reading the legal moves' logits directly
// (CRL_POLICY_LEGAL_RAW: csrc/search.hpp gather_priors / argmax_policy) -- does it through these two
// functions, so the values are the same bits wherever they are computed.
#pragma once
#include <hip/hip_runtime.h>

namespace crl_slices {

constexpr int N_SLICES = 8;

struct Norm { float M, inv; };

__device__ __forceinline__ Norm norm_of(const float2 *__restrict__ stats_row)
{
    float2 st[N_SLICES];
#pragma unroll
    for (int k = 0; k < N_SLICES; k++) st[k] = stats_row[k];
    float M = st[0].x;
#pragma unroll
    for (int k = 1; k < N_SLICES; k++) M = fmaxf(M, st[k].x);
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < N_SLICES; k++) S += st[k].y * __expf(st[k].x - M);
    Norm n;
    n.M = M;
    n.inv = 1.0f / S;
    return n;
}

__device__ __forceinline__ float prob(float logit, const Norm &n) { return __expf(logit - n.M) * n.inv; }

}  // namespace crl_slices
