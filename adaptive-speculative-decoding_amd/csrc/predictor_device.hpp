// predictor_device.hpp -- device code of the stopping-rule epilogue shared by predictor.hip (stand-alone
// launches) and verify_accept.hip (the epilogue run by the verify kernel's last arriver, N1 second form).
// Every TU including this header MUST be compiled with -ffp-contract=off (numpy / CPython parity).
#pragma once

#include "decision_device.hpp"
#include "lse_device.hpp"

#include <math.h>

namespace asd {

// ---- numpy pairwise summation (numpy/_core/src/umath/loops_utils.h.src: pairwise_sum) -----
// n < 8: sequential from 0; n <= 128: eight interleaved accumulators, fixed tree, sequential
// tail; n > 128: split at (n/2 rounded down to a multiple of 8).  Recursion unrolled by depth
// (kStatsMaxK = 128 * 2^3).
static __device__ double np_sum_leaf(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
}
__device__ __forceinline__ int np_split(int n) { int h = n / 2; return h - (h % 8); }
static __device__ double np_sum_d1(const double* a, int n) {
    if (n <= 128) return np_sum_leaf(a, n);
    const int h = np_split(n);
    return np_sum_leaf(a, h) + np_sum_leaf(a + h, n - h);
}
static __device__ double np_sum_d2(const double* a, int n) {
    if (n <= 128) return np_sum_leaf(a, n);
    const int h = np_split(n);
    return np_sum_d1(a, h) + np_sum_d1(a + h, n - h);
}
static __device__ double np_sum(const double* a, int n) {  // n <= 1024
    if (n <= 128) return np_sum_leaf(a, n);
    const int h = np_split(n);
    return np_sum_d2(a, h) + np_sum_d2(a + h, n - h);
}

// lane 0's double in every lane: two v_readfirstlane (scalar path) instead of the ds_bpermute pair of __shfl
__device__ __forceinline__ double bcast_lane0(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b >> 32));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}

// numpy _lerp (numpy/lib/_function_base_impl.py)
__device__ __forceinline__ double np_lerp(double a, double b, double t) {
    const double d = b - a;
    double r = a + d * t;
    if (t >= 0.5) r = b - d * (1.0 - t);
    if (d == 0.0) r = a;
    return r;
}

// One wave computes the five statistics of vals[0..n) (f64, in LDS); `sorted` and `sq` are LDS
// scratch of n doubles each.  Result valid in lane 0.
template <bool SMALL>
static __device__ void wave_logprob_stats_t(double* vals, double* sorted, double* sq, int n, int lane, double (&out)[5]) {
    if (n <= 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) out[i] = 0.0;             // features.extend([0.0]*5) :174-175
        return;
    }
    // rank sort (stable): every value lands at #{smaller} + #{equal and earlier}
    for (int i = lane; i < n; i += 64) {
        const double v = vals[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double w = vals[j];
            rank += (w < v) || (w == v && j < i);
        }
        sorted[rank] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double mean = 0.0;
    // SMALL: n <= 128 is known (the in-kernel epilogue: n = K <= 64), where numpy's pairwise sum IS its leaf routine
    if (lane == 0) mean = (SMALL ? np_sum_leaf(vals, n) : np_sum(vals, n)) / static_cast<double>(n);          // np.mean :168
    mean = bcast_lane0(mean);
    for (int i = lane; i < n; i += 64) {
        const double t = vals[i] - mean;
        sq[i] = t * t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
        const double var = (SMALL ? np_sum_leaf(sq, n) : np_sum(sq, n)) / static_cast<double>(n);            // np.std :169 (population)
        const double vi = static_cast<double>(n - 1) * 0.25;                  // np.percentile(.,25) :171
        const int lo = static_cast<int>(floor(vi));
        const int hi = lo + 1 < n ? lo + 1 : n - 1;
        out[0] = mean;
        out[1] = sqrt(var);
        out[2] = sorted[0];                                                   // np.min :170
        out[3] = np_lerp(sorted[lo], sorted[hi], vi - static_cast<double>(lo));
        out[4] = (n & 1) ? sorted[n / 2] : (sorted[n / 2 - 1] + sorted[n / 2]) / 2.0;  // np.median :172
    }
}

[[maybe_unused]] static __device__ void wave_logprob_stats(double* vals, double* sorted, double* sq, int n, int lane, double (&out)[5]) {
    wave_logprob_stats_t<false>(vals, sorted, sq, n, lane, out);
}

// ---- the five statistics again, for n <= 64 values held ONE PER LANE (the in-kernel epilogue: n = K), without LDS: every
// value is fetched with v_readlane (the lane index is wave-uniform), every lane computes the same sums.  Operation for
// operation the arithmetic of wave_logprob_stats (numpy's pairwise leaf routine, population std, linear-interpolated
// percentile, median), so the results are bit-identical; what goes away is seven LDS round trips and three wave barriers on
// the tail of the verify kernel (~0.4 us of its 2.5 us epilogue).
__device__ __forceinline__ double lane_value(double v, int j) {      // j wave-uniform
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(b), j);
    const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(b >> 32), j);
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ double np_sum_leaf_lanes(double v, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + lane_value(v, i);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = lane_value(v, j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = r[j] + lane_value(v, i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + lane_value(v, i);
    return res;
}
// v: this lane's value (lanes < n); out: the same five numbers in EVERY lane
__device__ __forceinline__ void wave_logprob_stats_lanes(double v, int n, int lane, double (&out)[5]) {
    if (n <= 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) out[i] = 0.0;
        return;
    }
    const double mean = np_sum_leaf_lanes(v, n) / static_cast<double>(n);
    const double t = v - mean;
    const double var = np_sum_leaf_lanes(t * t, n) / static_cast<double>(n);
    int rank = 0;                                                    // stable rank of this lane's value
    for (int j = 0; j < n; ++j) {
        const double w = lane_value(v, j);
        rank += (w < v) || (w == v && j < lane);
    }
    auto sorted_at = [&](int r) -> double {                          // r wave-uniform, 0 <= r < n
        const unsigned long long m = __ballot(lane < n && rank == r);
        return lane_value(v, static_cast<int>(__builtin_ctzll(m)));
    };
    const double vi = static_cast<double>(n - 1) * 0.25;
    const int lo = static_cast<int>(floor(vi));
    const int hi = lo + 1 < n ? lo + 1 : n - 1;
    out[0] = mean;
    out[1] = sqrt(var);
    out[2] = sorted_at(0);
    out[3] = np_lerp(sorted_at(lo), sorted_at(hi), vi - static_cast<double>(lo));
    out[4] = (n & 1) ? sorted_at(n / 2) : (sorted_at(n / 2 - 1) + sorted_at(n / 2)) / 2.0;
}

// ---- fused epilogue: parameters and the lane-0 decision tail -----------------------------
struct FusedParams {
    const float* lp; int64_t ld_lp; const int32_t* n_valid; int K;
    const float* feat; int64_t ldf; int stats_col;
    const float* packed; int in_dim, hidden, use_lds;
    int risk; double n_obs, alpha, beta;
    double* p_hist; const double* C; double lam; int L, stage_idx, prefix;
    const double* theta; int B;
    float* score; int32_t* k_star; uint8_t* stop; uint8_t* thr_stop; double* stats;
};

// What the decision tail reads besides the score: the sequence's probability history and the stage costs.  The latency
// forms fetch them up front (DecidePrefetch, lane 0) so that the tail behind the predictor is arithmetic only -- as loads
// issued there they were two more dependent memory round trips at the very end of the kernel.
constexpr int kDecidePrefetch = 4;   // stages whose history / cost are fetched up front (the reference's hierarchies: 3 or 4)
struct DecidePrefetch {
    double ph[kDecidePrefetch], cc[kDecidePrefetch], theta;
};
__device__ __forceinline__ void decide_prefetch(const FusedParams& p, int b, DecidePrefetch& d) {
    int z = 0;
    asm volatile("" : "+v"(z));   // vector loads: a scalarised s_load shares lgkmcnt with the LDS traffic of the caller's loop
    const int n_dp = p.prefix ? p.stage_idx + 1 : p.L;
    const bool dp = p.p_hist && (p.k_star || p.stop) && n_dp <= kDecidePrefetch;
#pragma unroll
    for (int i = 0; i < kDecidePrefetch; ++i) {
        d.ph[i] = 0.0;
        d.cc[i] = 0.0;
        if (dp && i < n_dp) {
            if (i != p.stage_idx) d.ph[i] = p.p_hist[static_cast<int64_t>(b) * p.L + i + z];
            d.cc[i] = p.C[i + z];
        }
    }
    d.theta = (p.theta && p.thr_stop) ? p.theta[p.stage_idx + z] : 0.0;
}

// lane-0 tail of the fused epilogue: Bayes adjustment, history update, DP rule, theta test
template <bool PREFETCHED>
__device__ __forceinline__ void decide_and_store_impl(const FusedParams& p, int b, float sc, const DecidePrefetch& d) {
    if (p.score) p.score[b] = sc;
    // pipeline.py:225-238: prob = predictor.predict(...); prob = bayesian_adjustment(prob, n_obs, a, b)
    double prob = static_cast<double>(sc);
    if (p.risk) prob = bayes_adjust1(prob, p.n_obs, p.alpha, p.beta);
    if (p.p_hist) {
        double* ph = p.p_hist + static_cast<int64_t>(b) * p.L;
        ph[p.stage_idx] = prob;
        if (p.k_star || p.stop) {
            const int n_dp = p.prefix ? p.stage_idx + 1 : p.L;       // pipeline.py:248-256 uses the prefix
            double pp[ASD_MAX_STAGES], cc[ASD_MAX_STAGES], J[ASD_MAX_STAGES + 1];
            if (PREFETCHED && n_dp <= kDecidePrefetch) {
#pragma unroll
                for (int i = 0; i < kDecidePrefetch; ++i) {
                    if (i < n_dp) { pp[i] = (i == p.stage_idx) ? prob : d.ph[i]; cc[i] = d.cc[i]; }
                }
            } else {
#pragma unroll
                for (int i = 0; i < ASD_MAX_STAGES; ++i) {
                    if (i < n_dp) { pp[i] = (i == p.stage_idx) ? prob : ph[i]; cc[i] = p.C[i]; }
                }
            }
            const int ks = optimal_stopping1(pp, cc, p.lam, n_dp, 0, 1.0, 1.0, J);
            if (p.k_star) p.k_star[b] = ks;
            if (p.stop) p.stop[b] = (ks == p.stage_idx) ? 1 : 0;     // pipeline.py:259
        }
    }
    if (p.theta && p.thr_stop) {
        const double q = static_cast<double>(sc);                    // minimal_adaptive_decoder.py:159-161
        const double th = PREFETCHED ? d.theta : p.theta[p.stage_idx];
        p.thr_stop[b] = (q >= th || p.stage_idx == p.L - 1) ? 1 : 0;
    }
}
// the same tail for hierarchies of at most kDecidePrefetch stages, everything from the prefetch: a fraction of the registers
// (the general form keeps 16-stage arrays alive; k_verify<FUSED> is launched for L <= kDecidePrefetch only)
__device__ __forceinline__ void decide_and_store_small(const FusedParams& p, int b, float sc, const DecidePrefetch& d) {
    if (p.score) p.score[b] = sc;
    double prob = static_cast<double>(sc);
    if (p.risk) prob = bayes_adjust1(prob, p.n_obs, p.alpha, p.beta);
    if (p.p_hist) {
        p.p_hist[static_cast<int64_t>(b) * p.L + p.stage_idx] = prob;
        if (p.k_star || p.stop) {
            const int n_dp = p.prefix ? p.stage_idx + 1 : p.L;
            double pp[kDecidePrefetch], cc[kDecidePrefetch], J[kDecidePrefetch + 1];
#pragma unroll
            for (int i = 0; i < kDecidePrefetch; ++i) {
                pp[i] = (i == p.stage_idx) ? prob : d.ph[i];
                cc[i] = d.cc[i];
            }
            const int ks = optimal_stopping_n<kDecidePrefetch>(pp, cc, p.lam, n_dp, 0, 1.0, 1.0, J);
            if (p.k_star) p.k_star[b] = ks;
            if (p.stop) p.stop[b] = (ks == p.stage_idx) ? 1 : 0;
        }
    }
    if (p.theta && p.thr_stop) {
        const double q = static_cast<double>(sc);
        p.thr_stop[b] = (q >= d.theta || p.stage_idx == p.L - 1) ? 1 : 0;
    }
}
__device__ __forceinline__ void decide_and_store(const FusedParams& p, int b, float sc) {
    DecidePrefetch none;
    decide_and_store_impl<false>(p, b, sc, none);
}

// ---- the reference's predictor (64 -> 32 -> 1), one wave per sequence ---------------------
// Everything that does not depend on the log-probs, fetched up front: the first layer is split over
// the two half-waves (lane = hidden unit j, half = input half), its weights live in registers.
struct EpiPrefetch {
    float w[32];
    float b1, w2, b2, xv;
    DecidePrefetch d;      // lane 0 only
};

__device__ __forceinline__ void epi_prefetch(const FusedParams& p, int b, int lane, EpiPrefetch& e) {
    e.xv = p.feat[static_cast<int64_t>(b) * p.ldf + lane];
    const int j = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 32; ++i) e.w[i] = p.packed[(half * 32 + i) * 32 + j];
    e.b1 = p.packed[64 * 32 + j];
    e.w2 = p.packed[64 * 32 + 32 + j];
    e.b2 = p.packed[64 * 32 + 64];
    if (lane == 0) decide_prefetch(p, b, e.d);
}

// lpv: this lane's log-prob (lanes < n).  dvals: 3*64 doubles of LDS, xs: 64 floats of LDS (16-byte aligned).
__device__ __forceinline__ void epi_finish(const FusedParams& p, int b, int lane, float lpv, int n, bool want_stats,
                                           const EpiPrefetch& e, double* dvals, float* xs) {
    float xv = e.xv;
    if (want_stats) {
        double st[5];
        wave_logprob_stats_lanes(static_cast<double>(lpv), n, lane, st);      // n <= 64 values, one per lane: registers + readlane only
        if (p.stats && lane == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) p.stats[5 * static_cast<int64_t>(b) + i] = st[i];
        }
        const int si = lane - p.stats_col;
        if (p.stats_col >= 0 && si >= 0 && si < 5)
            xv = static_cast<float>(si == 0 ? st[0] : si == 1 ? st[1] : si == 2 ? st[2] : si == 3 ? st[3] : st[4]);
    }
    xs[lane] = xv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int half = lane >> 5;
    float h = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) h = fmaf(e.w[i], xs[half * 32 + i], h);
    h += __shfl_xor(h, 32, 64);
    h = fmaxf(h + e.b1, 0.0f);
    // the output unit: one DPP wave sum (fixed order; the ds_bpermute butterfly it replaces was five dependent LDS round trips)
    const float z = wave_sum(half == 0 ? e.w2 * h : 0.0f);
    const float sc = 1.0f / (1.0f + expf(-(z + e.b2)));
    if (lane == 0) decide_and_store_impl<true>(p, b, sc, e.d);
}


// ---- the same epilogue for a kernel that cannot afford the weights in registers while it streams (k_verify<FUSED>:
// 55 more VGPRs took it to 135 and to ONE workgroup per CU).  The packed weights sit in LDS (put there by LDS-DMA at the
// kernel's start: no register is held across the stream); the feature and the decision inputs are loaded LATE -- issued by
// the finisher right before it waits for the hand-off slots, i.e. under a wait that is there anyway.  Same operations in
// the same order as epi_finish: bit-identical scores.
struct EpiLate {
    float xv;
    DecidePrefetch d;      // lane 0 only
};
__device__ __forceinline__ void epi_late_prefetch(const FusedParams& p, int b, int lane, EpiLate& e) {
    e.xv = p.feat[static_cast<int64_t>(b) * p.ldf + lane];
    if (lane == 0) decide_prefetch(p, b, e.d);
}
// wl: the 64 * 32 + 65 packed floats in LDS (W1^T [in][hidden], b1, W2, b2)
__device__ __forceinline__ void epi_finish_lds(const FusedParams& p, int b, int lane, float lpv, int n, bool want_stats,
                                               const float* wl, const EpiLate& e, double* dvals, float* xs) {
    float xv = e.xv;
    if (want_stats) {
        double st[5];
        wave_logprob_stats_lanes(static_cast<double>(lpv), n, lane, st);      // registers + readlane only (dvals unused)
        if (p.stats && lane == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) p.stats[5 * static_cast<int64_t>(b) + i] = st[i];
        }
        const int si = lane - p.stats_col;
        if (p.stats_col >= 0 && si >= 0 && si < 5)
            xv = static_cast<float>(si == 0 ? st[0] : si == 1 ? st[1] : si == 2 ? st[2] : si == 3 ? st[3] : st[4]);
    }
    xs[lane] = xv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int j = lane & 31, half = lane >> 5;
    float h = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) h = fmaf(wl[(half * 32 + i) * 32 + j], xs[half * 32 + i], h);
    h += __shfl_xor(h, 32, 64);
    h = fmaxf(h + wl[64 * 32 + j], 0.0f);
    const float z = wave_sum(half == 0 ? wl[64 * 32 + 32 + j] * h : 0.0f);      // (same DPP order as epi_finish: bit-identical scores)
    const float sc = 1.0f / (1.0f + expf(-(z + wl[64 * 32 + 64])));
    if (lane == 0) decide_and_store_small(p, b, sc, e.d);
}

}  // namespace asd
