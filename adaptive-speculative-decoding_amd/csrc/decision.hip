// decision.hip -- the reference's float64 decision arithmetic as gfx950 kernels, batched over
// requests.  This translation unit is compiled with -ffp-contract=off: CPython evaluates
// `C[i] + lam * (1 - p_bar)` as two rounded operations, so an FMA here would change k* on ties.
//
//   A1  optimal_stopping_rule      src/algorithms/dp_solver.py:12-71
//   A2  bayesian_adjustment        src/algorithms/dp_solver.py:106-130
//   A3  compute_expected_cost      src/algorithms/dp_solver.py:74-103
//   A10 derive_optimal_policy      src/theory/optimal_stopping.py:45-82   (host, O(n))
//   A11 stop test                  src/minimal_adaptive_decoder.py:153-164
//
// All kernels are one-thread-per-request, L <= ASD_MAX_STAGES values per thread in registers:
// the work is tiny (O(L) f64 per request) and latency-bound; batching it is what removes the
// reference's one-Python-call-per-stage-per-request pattern (pipeline.py:234-256).

#include "decision_device.hpp"

#include <math.h>

namespace asd {
namespace {

__global__ void k_bayes(const double* __restrict__ p, double n_obs, double alpha, double beta, int n,
                        double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bayes_adjust1(p[i], n_obs, alpha, beta);
}

__global__ void k_optimal_stopping(const double* __restrict__ p, const double* __restrict__ C, double lam, int B,
                                   int L, int risk, double alpha, double beta, int32_t* __restrict__ k_star,
                                   double* __restrict__ J) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double pp[ASD_MAX_STAGES], cc[ASD_MAX_STAGES], Jb[ASD_MAX_STAGES + 1];
#pragma unroll
    for (int i = 0; i < ASD_MAX_STAGES; ++i) {
        if (i < L) {
            pp[i] = p[static_cast<int64_t>(b) * L + i];
            cc[i] = C[i];
        }
    }
    const int ks = optimal_stopping1(pp, cc, lam, L, risk, alpha, beta, Jb);
    k_star[b] = ks;
    if (J) {
#pragma unroll
        for (int i = 0; i <= ASD_MAX_STAGES; ++i)
            if (i <= L) J[static_cast<int64_t>(b) * (L + 1) + i] = Jb[i];
    }
}

// N4: the DP rule for a whole grid of lambda values -- one thread per (lambda, request).  p_ok multiplies
// left to right like compute_expected_cost (dp_solver.py:92-95), cost adds left to right like sum(C[:k+1]).
__global__ void k_lambda_sweep(const double* __restrict__ p, const double* __restrict__ C, const double* __restrict__ lam,
                               int B, int L, int G, int risk, double alpha, double beta, int32_t* __restrict__ k_star,
                               double* __restrict__ cost, double* __restrict__ p_ok) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<int64_t>(G) * B) return;
    const int g = static_cast<int>(i / B);
    const int b = static_cast<int>(i % B);
    double pp[ASD_MAX_STAGES], cc[ASD_MAX_STAGES], Jb[ASD_MAX_STAGES + 1];
#pragma unroll
    for (int s = 0; s < ASD_MAX_STAGES; ++s) {
        if (s < L) {
            pp[s] = p[static_cast<int64_t>(b) * L + s];
            cc[s] = C[s];
        }
    }
    const int ks = optimal_stopping1(pp, cc, lam[g], L, risk, alpha, beta, Jb);
    double pb = 1.0, cs = 0.0;
#pragma unroll
    for (int s = 0; s < ASD_MAX_STAGES; ++s) {
        if (s <= ks) {
            pb *= pp[s];
            cs += cc[s];
        }
    }
    k_star[i] = ks;
    if (cost) cost[i] = cs;
    if (p_ok) p_ok[i] = pb;
}

__global__ void k_expected_cost(const double* __restrict__ p, const double* __restrict__ C, double lam,
                                const int32_t* __restrict__ k, int B, int L, double* __restrict__ cost) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int kk = k[b];
    if (kk < 0) kk = -1;
    if (kk > L - 1) kk = L - 1;
    double p_bar = 1.0;                                        // dp_solver.py:92-94
    for (int i = 0; i <= kk; ++i) p_bar = p_bar * p[static_cast<int64_t>(b) * L + i];
    double comp = 0.0;                                         // sum(C[:k+1]) :97 (int 0 + floats)
    for (int i = 0; i <= kk; ++i) comp = comp + C[i];
    const double quality_loss = lam * (1 - p_bar);             // :100
    cost[b] = comp + quality_loss;                             // :102
}

__global__ void k_threshold_stop(const float* __restrict__ score, const double* __restrict__ theta, int B, int L,
                                 int32_t* __restrict__ stage) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    stage[b] = threshold_stop1(score[b], theta, L);
}

inline dim3 grid_for(int n, int block) { return dim3(static_cast<uint32_t>((n + block - 1) / block)); }

}  // namespace
}  // namespace asd

using namespace asd;

ASD_EXPORT int asd_bayes_adjust(const double* p, int64_t n_obs, double alpha, double beta, int n, double* out,
                                void* stream) {
    if (n < 0) return ASD_ERR_INVALID_ARG;
    if (n == 0) return ASD_OK;
    if (!p || !out) return ASD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_bayes, grid_for(n, 256), dim3(256), 0, static_cast<hipStream_t>(stream), p,
                       static_cast<double>(n_obs), alpha, beta, n, out);
    return launch_status();
}

ASD_EXPORT int asd_optimal_stopping(const double* p, const double* C, double lam, int B, int L, int risk_adjustment,
                                    double alpha, double beta, int32_t* k_star, double* J, void* stream) {
    if (B < 0 || L < 1) return ASD_ERR_INVALID_ARG;
    if (L > ASD_MAX_STAGES) return ASD_ERR_UNSUPPORTED;
    if (B == 0) return ASD_OK;
    if (!p || !C || !k_star) return ASD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_optimal_stopping, grid_for(B, 64), dim3(64), 0, static_cast<hipStream_t>(stream), p, C, lam,
                       B, L, risk_adjustment ? 1 : 0, alpha, beta, k_star, J);
    return launch_status();
}

ASD_EXPORT int asd_lambda_sweep(const double* p, const double* C, const double* lam, int B, int L, int G,
                                int risk_adjustment, double alpha, double beta, int32_t* k_star, double* cost,
                                double* p_ok, void* stream) {
    if (B < 0 || G < 0 || L < 1) return ASD_ERR_INVALID_ARG;
    if (L > ASD_MAX_STAGES) return ASD_ERR_UNSUPPORTED;
    if (B == 0 || G == 0) return ASD_OK;
    if (!p || !C || !lam || !k_star) return ASD_ERR_INVALID_ARG;
    const int64_t n = static_cast<int64_t>(G) * B;
    if (n >= (1ll << 31)) return ASD_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_lambda_sweep, grid_for(static_cast<int>(n), 64), dim3(64), 0, static_cast<hipStream_t>(stream), p,
                       C, lam, B, L, G, risk_adjustment ? 1 : 0, alpha, beta, k_star, cost, p_ok);
    return launch_status();
}

ASD_EXPORT int asd_expected_cost(const double* p, const double* C, double lam, const int32_t* k, int B, int L,
                                 double* cost, void* stream) {
    if (B < 0 || L < 1) return ASD_ERR_INVALID_ARG;
    if (B == 0) return ASD_OK;
    if (!p || !C || !k || !cost) return ASD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_expected_cost, grid_for(B, 64), dim3(64), 0, static_cast<hipStream_t>(stream), p, C, lam, k,
                       B, L, cost);
    return launch_status();
}

ASD_EXPORT int asd_threshold_stop(const float* score, const double* theta, int B, int L, int32_t* stage,
                                  void* stream) {
    if (B < 0 || L < 1) return ASD_ERR_INVALID_ARG;
    if (B == 0) return ASD_OK;
    if (!score || !theta || !stage) return ASD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_threshold_stop, grid_for(B, 256), dim3(256), 0, static_cast<hipStream_t>(stream), score,
                       theta, B, L, stage);
    return launch_status();
}

// Host-side: an O(n) f64 recursion evaluated once per set_lambda (optimal_stopping.py:45-82).
// Kept in this TU so it inherits -ffp-contract=off.
ASD_EXPORT int asd_derive_thresholds(const double* q, const double* c, int n, double lam, double* theta, double* V_out) {
    if (n < 1 || !q || !c || !theta) return ASD_ERR_INVALID_ARG;
    if (n > 64) return ASD_ERR_UNSUPPORTED;
    double V[65];
    for (int i = 0; i <= n; ++i) V[i] = 0.0;                                   // :57
    for (int s = n - 1; s >= 0; --s) {                                         // :61
        const double r_stop = q[s] - lam * c[s];                               // :63
        double r_continue;
        if (s < n - 1) {
            const double p_improve = 0.6 * (1 - q[s]);                         // :91
            r_continue = p_improve * V[s + 1] + (1 - p_improve) * r_stop;      // :69
        } else {
            r_continue = -INFINITY;                                            // :71
        }
        V[s] = (r_continue > r_stop) ? r_continue : r_stop;                    // max(r_stop, r_continue) :73
        if (s < n - 1)
            theta[s] = (V[s + 1] + lam * c[s]) / (1 + lam * (c[s + 1] - c[s])); // :77
        else
            theta[s] = 0.0;                                                    // :79
    }
    if (V_out)
        for (int i = 0; i <= n; ++i) V_out[i] = V[i];
    return ASD_OK;
}
