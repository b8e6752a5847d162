// decision_device.hpp -- device functions for the f64 decision arithmetic, shared by
// decision.hip and predictor.hip (the fused epilogue).  Every TU that includes this header MUST
// be compiled with -ffp-contract=off (see build.py): the reference is CPython float arithmetic,
// one IEEE rounding per operator.
#pragma once

#include "common.hpp"

namespace asd {

// dp_solver.py:106-130
__device__ __forceinline__ double bayes_adjust1(double p_hat, double n_obs, double alpha, double beta) {
    const double posterior_alpha = n_obs * p_hat + alpha;        // :122
    const double posterior_beta = n_obs * (1 - p_hat) + beta;    // :123
    return posterior_alpha / (posterior_alpha + posterior_beta); // :126
}

// dp_solver.py:12-71 for one request; p, C: L values; J: L+1 values out.  Returns k*.  MAXS bounds L at compile time (the
// arrays live in registers: a kernel that only ever sees 3- or 4-tier hierarchies instantiates MAXS = 4).
template <int MAXS>
__device__ __forceinline__ int optimal_stopping_n(const double (&p_in)[MAXS], const double (&C)[MAXS],
                                                  double lam, int L, int risk, double alpha, double beta,
                                                  double (&J)[MAXS + 1]) {
    double p_bar[MAXS + 1];
    p_bar[0] = 1.0;                                              // :42
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        if (i < L) {
            const double pi = risk ? bayes_adjust1(p_in[i], 100.0, alpha, beta) : p_in[i];  // :38-39
            p_bar[i + 1] = p_bar[i] * pi;                        // :44
        }
    }
    int k_star = L - 1;                                          // :67 fallback
    double next = 0.0;                                           // J[L] = 0 :47
#pragma unroll
    for (int i = MAXS; i >= 0; --i)
        if (i == L) J[i] = 0.0;
#pragma unroll
    for (int i = MAXS - 1; i >= 0; --i) {                        // :51 reversed(range(L))
        if (i < L) {
            const double cost_if_stop = C[i] + lam * (1 - p_bar[i + 1]);   // :53
            const double cost_if_continue = C[i] + next;                   // :56
            const bool stop = cost_if_stop <= cost_if_continue;            // :59
            next = stop ? cost_if_stop : cost_if_continue;
            J[i] = next;
            if (stop) k_star = i;                                // lowest stopping index wins (:67 first True)
        }
    }
    return k_star;
}

// the same rule for a hierarchy whose depth is a compile-time constant: no per-stage predicates (the in-kernel epilogue
// branches once on the wave-uniform depth and runs ~25 f64 instructions instead of ~70).  Operation for operation
// optimal_stopping_n's arithmetic with risk = 0.
template <int L>
__device__ __forceinline__ int optimal_stopping_exact(const double* p_in, const double* C, double lam) {
    double p_bar[L + 1];
    p_bar[0] = 1.0;                                              // :42
#pragma unroll
    for (int i = 0; i < L; ++i) p_bar[i + 1] = p_bar[i] * p_in[i];   // :44
    int k_star = L - 1;                                          // :67 fallback
    double next = 0.0;                                           // J[L] = 0 :47
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {                           // :51 reversed(range(L))
        const double cost_if_stop = C[i] + lam * (1 - p_bar[i + 1]);   // :53
        const double cost_if_continue = C[i] + next;                   // :56
        const bool stop = cost_if_stop <= cost_if_continue;            // :59
        next = stop ? cost_if_stop : cost_if_continue;
        if (stop) k_star = i;                                    // lowest stopping index wins (:67 first True)
    }
    return k_star;
}

__device__ __forceinline__ int optimal_stopping1(const double (&p_in)[ASD_MAX_STAGES], const double (&C)[ASD_MAX_STAGES],
                                                 double lam, int L, int risk, double alpha, double beta,
                                                 double (&J)[ASD_MAX_STAGES + 1]) {
    return optimal_stopping_n<ASD_MAX_STAGES>(p_in, C, lam, L, risk, alpha, beta, J);
}

// minimal_adaptive_decoder.py:153-164
__device__ __forceinline__ int threshold_stop1(float score, const double* theta, int L) {
    const double q = static_cast<double>(score);                 // .item() widens f32 -> Python float
    for (int s = 0; s < L; ++s)
        if (q >= theta[s] || s == L - 1) return s;
    return L - 1;
}

}  // namespace asd
