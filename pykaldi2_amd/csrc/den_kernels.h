// Pieces of the denominator kernels shared by chain_den.hip (a launch per frame) and chain_den_persist.hip (one launch
// per call): the parameter block and the cross-lane segmented-scan step.
#pragma once
#include "chain_internal.h"

namespace pk2 {

// DPP cross-lane moves (full-rate VALU, no LDS round trip): lanes whose source lies outside the 16-lane row / is masked
// receive 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }

template <int NG, int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_scan_step(float (&x)[NG], int& fl) {
  float y[NG];
#pragma unroll
  for (int n = 0; n < NG; ++n) y[n] = dpp_f<CTRL, ROW_MASK>(x[n]);
  const int g = dpp_i<CTRL, ROW_MASK>(fl);
#pragma unroll
  for (int n = 0; n < NG; ++n) x[n] += fl ? 0.f : y[n];
  fl |= g;
}

// Uniform floor of the backward normaliser's weights, relative to the mean of pi (see den_beta_frame_sx).
constexpr double kBetaFloor = 1e-8;

struct DenParams {
  DevOrdering fwd, bwd, gam;
  const float* pi;
  float* alpha; float* beta; float* xs; float* gamma;
  float* apart; float* bpart; float* asum; float* inv_tot;
  const int32_t* lengths;
  // state-x path (chain_internal.h: peeled self-loops, virtual states, occupancy states); fwd / bwd are then the
  // fwdv / bwdv orderings
  const int32_t* ps_off;    // occupancy states grouped by pdf (CSR over P)
  const int32_t* ps_state;
  const int32_t* voff;      // [S+1] first virtual state of a state
  const int32_t* ooff;      // [S+1] first occupancy state of a state
  const int32_t* opdf;      // [Vo] pdf of an occupancy state (-1: none)
  const int32_t* ovirt;     // [Vo] first virtual state of the occupancy state's state (its record holds btilde')
  const float* loop_prob;   // [S] probability of the peeled self-loop (0: none)
  float* alphav;            // [G][Tmax+1][Vo][NG]; == alpha when Vo == S
  const float* xl;          // [G][Tmax][S][NG] exp(logit) of the peeled self-loop's pdf
  int V, Vo;
  int S, P, Tmax;
  float leaky, pi_sum;
  float wu;    // kBetaFloor * sum(pi)/S: uniform floor of the backward normaliser's weights (state-x path)
  int brec;    // floats per beta record and sequence: 2 = {btilde', x} (frame kernels, first persistent kernel), 1 = btilde' alone
               // (second persistent kernel, round 4: nobody reads the x halves there; the occupancy pass gathers half the lines)
  float beta_seed;   // test hook (PK2_DEN_DEBUG_BETA_SEED, default 1): beta'(T) = beta_seed / tot -- scales every beta, the occupancies
                     // and the alpha-beta consistency product, so that a test can drive the abandon rule on real inputs
  int debug;   // PK2_DEN_DEBUG ablation bits (profiling only): 1 = all gathers hit state 0, 2 = skip the arc loop, 4 = load half of the arc records
};

}  // namespace pk2
