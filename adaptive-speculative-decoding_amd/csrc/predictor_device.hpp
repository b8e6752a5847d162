// predictor_device.hpp -- device code of the stopping-rule epilogue shared by predictor.hip (stand-alone
// launches) and verify_accept.hip (the epilogue run by the verify kernel's last arriver, N1 second form).
// Every TU including this header MUST be compiled with -ffp-contract=off (numpy / CPython parity).
#pragma once

#include "decision_device.hpp"
#include "lse_device.hpp"

#include <math.h>

#ifndef ASD_EPI_STAMP
#define ASD_EPI_STAMP(slot) do { } while (0)   // stage stamps: the diagnostic build of verify_accept.hip only (tools/stamp_verify.py)
#endif

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

// ---- the five statistics once more, for n <= 16 values held one per lane of the wave's first row (the serving shapes:
// K = 4 ... 16).  A single wave runs the in-kernel epilogue behind the last row of its sequence, so what it costs is its
// INSTRUCTION COUNT (one VALU instruction per ~4-8 cycles, nothing to overlap with): the readlane form above spends
// ~1.3 us of a 2.2 us epilogue on ~700 instructions (profiles/r04_stamps_c3_fused_stages.log).  Here:
//  * numpy's pairwise leaf sum as a DPP tree -- quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror give lane 0
//    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), operand for operand numpy's expression (a + b = b + a bit for bit, so the other
//    lanes' mirrored operands do not matter); n = 16 first folds a[8+j] onto r[j] (row_shl:8); the n % 8 tail stays a
//    sequential chain of v_readlane adds;
//  * sum / n as a multiplication by 2^-k when n is a power of two (both are the correctly rounded value of the same real
//    number: identical bits, one instruction instead of the ~25 of an f64 division);
//  * the order statistics from a bitonic sorting network (with flips) over order-preserving integer keys of the f32
//    log-probs -- v_min_u32 / v_max_u32 with a DPP operand + one v_cndmask per stage, 6 stages for n <= 8, 10 for n <= 16 --
//    instead of the O(n) readlane rank loop.  The sorted VALUES are those of the stable rank sort (only the order among
//    +0 / -0, which numpy does not define either, can differ).
// Same arithmetic as wave_logprob_stats / wave_logprob_stats_lanes for the sums, the std, the percentile and the median.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(b)), CTRL, 0xf, 0xf, true));
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(b >> 32)), CTRL, 0xf, 0xf, true));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}
// numpy's pairwise_sum leaf over n <= 16 doubles, lane i < n of row 0 holding a[i]; the result is wave-uniform
__device__ __forceinline__ double np_sum_leaf_row16(double v, int n) {
    double res = 0.0;
    int i = 0;
    if (n >= 8) {
        double r = v;                                    // r[j] = a[j]
        if (n == 16) r = r + dpp_f64<0x108>(v);          // row_shl:8   r[j] += a[8 + j]
        r = r + dpp_f64<0xB1>(r);                        // quad_perm [1,0,3,2]   (r0+r1) (r2+r3) (r4+r5) (r6+r7)
        r = r + dpp_f64<0x4E>(r);                        // quad_perm [2,3,0,1]   (r0+r1)+(r2+r3) | (r4+r5)+(r6+r7)
        r = r + dpp_f64<0x141>(r);                       // row_half_mirror       ((..)+(..)) + ((..)+(..))
        res = bcast_lane0(r);
        i = n - (n & 7);
    }
    for (; i < n; ++i) res = res + lane_value(v, i);     // n < 8: numpy's sequential loop from 0.0; otherwise the n % 8 tail
    return res;
}
// f32 <-> unsigned key with the same order (-inf < ... < -0 < +0 < ... < +inf < NaN)
__device__ __forceinline__ uint32_t f32_order_key(float x) {
    const uint32_t b = __float_as_uint(x);
    return b ^ (static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(uint32_t k) {
    return __uint_as_float(k ^ ((k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu));
}
template <int CTRL>
__device__ __forceinline__ uint32_t sort_step(uint32_t k, bool low) {       // compare-exchange with the lane CTRL names
    const uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(k), static_cast<int>(k), CTRL, 0xf, 0xf, false));
    const uint32_t lo = k < o ? k : o, hi = k < o ? o : k;
    return low ? lo : hi;
}
// lpv: this lane's log-prob (lanes < n of row 0), n <= 16; out: the five numbers, the same in every lane
__device__ __forceinline__ void wave_logprob_stats_row16(float lpv, int n, int lane, double (&out)[5]) {
    if (n <= 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) out[i] = 0.0;
        return;
    }
    const bool pow2 = (n & (n - 1)) == 0;
    const double dn = static_cast<double>(n);
    const double inv_n = __builtin_bit_cast(double, static_cast<unsigned long long>(1023 - __builtin_ctz(static_cast<unsigned>(n))) << 52);
    const double v = static_cast<double>(lpv);
    const double sum = np_sum_leaf_row16(v, n);
    double mean, var;
    if (pow2) mean = sum * inv_n; else mean = sum / dn;                       // np.mean
    const double t = v - mean;
    const double ssq = np_sum_leaf_row16(t * t, n);
    if (pow2) var = ssq * inv_n; else var = ssq / dn;                         // np.std (population)
    // sort (ascending, lanes 0..15 of the row; lanes >= n hold the largest key)
    uint32_t k = lane < n ? f32_order_key(lpv) : 0xFFFFFFFFu;
    const bool b0 = (lane & 1) == 0, b1 = (lane & 2) == 0, b2 = (lane & 4) == 0, b3 = (lane & 8) == 0;
    k = sort_step<0xB1>(k, b0);                                               // blocks of 2
    k = sort_step<0x1B>(k, b1);                                               // blocks of 4: mirror [3,2,1,0], then stride 1
    k = sort_step<0xB1>(k, b0);
    k = sort_step<0x141>(k, b2);                                              // blocks of 8: row_half_mirror, strides 2, 1
    k = sort_step<0x4E>(k, b1);
    k = sort_step<0xB1>(k, b0);
    if (n > 8) {
        k = sort_step<0x140>(k, b3);                                          // blocks of 16: row_mirror, strides 4, 2, 1
        {   // stride 4 has no single DPP pattern: row_ror:4 (lane i <- i - 4) serves the lanes with bit 2 set (banks 1, 3),
            // row_ror:12 (lane i <- i + 4) the others (banks 0, 2)
            uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(k), static_cast<int>(k), 0x124, 0xf, 0xA, false));
            o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(o), static_cast<int>(k), 0x12C, 0xf, 0x5, false));
            const uint32_t lo = k < o ? k : o, hi = k < o ? o : k;
            k = b2 ? lo : hi;
        }
        k = sort_step<0x4E>(k, b1);
        k = sort_step<0xB1>(k, b0);
    }
    auto sorted_at = [&](int r) -> double {                                   // r wave-uniform, 0 <= r < n
        return static_cast<double>(f32_from_order_key(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(k), r))));
    };
    const int lo = (n - 1) >> 2;                                              // floor((n - 1) * 0.25)
    const int hi = lo + 1 < n ? lo + 1 : n - 1;
    const double frac = static_cast<double>((n - 1) & 3) * 0.25;              // (n - 1) * 0.25 - lo, exactly
    out[0] = mean;
    out[1] = sqrt(var);
    out[2] = sorted_at(0);
    out[3] = np_lerp(sorted_at(lo), sorted_at(hi), frac);
    out[4] = (n & 1) ? sorted_at(n / 2) : (sorted_at(n / 2 - 1) + sorted_at(n / 2)) / 2.0;
}
// n <= 64 values one per lane: the DPP form where it applies, the readlane form otherwise (wave-uniform choice)
__device__ __forceinline__ void wave_logprob_stats_regs(float lpv, int n, int lane, double (&out)[5]) {
    if (n <= 16) wave_logprob_stats_row16(lpv, n, lane, out);
    else wave_logprob_stats_lanes(static_cast<double>(lpv), n, lane, out);
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
            int ks;
            static_assert(kDecidePrefetch == 4, "the depth switch below lists 1 ... 4");
            (void)J;
            switch (n_dp) {                                              // wave-uniform
                case 1: ks = optimal_stopping_exact<1>(pp, cc, p.lam); break;
                case 2: ks = optimal_stopping_exact<2>(pp, cc, p.lam); break;
                case 3: ks = optimal_stopping_exact<3>(pp, cc, p.lam); break;
                default: ks = optimal_stopping_exact<4>(pp, cc, p.lam); break;
            }
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
// CANONICAL ORDER of the latency forms (k_predictor_stop_w64x32 and the in-kernel epilogue of k_verify<FUSED>; round 4).
// The first layer is split over the two half-waves (lane = hidden unit j, half = input half).  Only the five statistics
// columns [stats_col, stats_col + 5) of the feature vector depend on the log-probs, so they are accumulated LAST:
//   phase A   h = 0;  for i = 0..31:  h = fma(W1[j][32 half + i], x'[32 half + i], h)      x' = the feature row with the
//                                                                                           statistics columns zeroed
//   phase B   for d = 0..4:  c = stats_col + d;  h = fma(W1[j][c], (half == c / 32) ? stat[d] : 0, h)
//   then      h = h(half 0) + h(half 1);  h = relu(h + b1[j]);  z = wave_sum(half == 0 ? W2[j] h : 0);  sigmoid(z + b2)
// Phase A needs nothing the verify pass produces: the in-kernel epilogue runs it UNDER the wait for the hand-off slots,
// and what is left behind the statistics is five FMAs instead of a 32-step chain.  Both latency forms use this order, so
// their scores stay bit-identical (tests/test_gpu_predictor.py); against the oracle the bar is 1e-5 as before.
struct EpiPrefetch {
    float w[32];
    float wd[ASD_NUM_LP_STATS];   // W1[j][stats_col + d]: the deferred columns' weights for this lane's hidden unit
    float b1, w2, b2, xv;
    DecidePrefetch d;      // lane 0 only
};

__device__ __forceinline__ void epi_prefetch(const FusedParams& p, int b, int lane, EpiPrefetch& e) {
    e.xv = p.feat[static_cast<int64_t>(b) * p.ldf + lane];
    const int j = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 32; ++i) e.w[i] = p.packed[(half * 32 + i) * 32 + j];
#pragma unroll
    for (int d = 0; d < ASD_NUM_LP_STATS; ++d) e.wd[d] = p.stats_col >= 0 ? p.packed[(p.stats_col + d) * 32 + j] : 0.0f;
    e.b1 = p.packed[64 * 32 + j];
    e.w2 = p.packed[64 * 32 + 32 + j];
    e.b2 = p.packed[64 * 32 + 64];
    if (lane == 0) decide_prefetch(p, b, e.d);
}

// the feature row with the statistics columns zeroed (phase A's input), one value per lane
__device__ __forceinline__ float epi_phase_a_input(const FusedParams& p, int lane, float xv) {
    const int si = lane - p.stats_col;
    return (p.stats_col >= 0 && si >= 0 && si < ASD_NUM_LP_STATS) ? 0.0f : xv;
}
// both halves' sums in every lane: v_permlane32_swap (gfx950) exchanges the upper half of one copy with the lower half of the
// other -- one VALU instruction where __shfl_xor(h, 32) is a ds_bpermute round trip
__device__ __forceinline__ float add_halves(float h) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(h), __float_as_uint(h), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);      // (half 0's h) + (half 1's h) in every lane
}
// phase B and everything behind it; h: phase A's accumulator, wd: the deferred columns' weights, st: the statistics (uniform)
template <typename WD>
__device__ __forceinline__ float epi_score(const FusedParams& p, int lane, float h, const WD& wd, const double (&st)[5], bool overlay,
                                           float b1, float w2, float b2) {
    const int half = lane >> 5;
    if (overlay) {
#pragma unroll
        for (int d = 0; d < ASD_NUM_LP_STATS; ++d) {
            const float sd = (half == ((p.stats_col + d) >> 5)) ? static_cast<float>(st[d]) : 0.0f;
            h = fmaf(wd(d), sd, h);
        }
    }
    h = add_halves(h);
    h = fmaxf(h + b1, 0.0f);
    const float z = wave_sum(half == 0 ? w2 * h : 0.0f);     // one DPP wave sum, fixed order
    ASD_EPI_STAMP(10);   // both layers done
    // sigmoid on the hardware's 2^x and 1/x (v_exp_f32, v_rcp_f32: 1 ulp each; the score moves by < 2e-7, the bar against the
    // reference is 1e-5): 5 instructions where expf + an IEEE division are ~33 -- and a single wave's instruction count is what
    // this epilogue costs.  Both latency forms share this function, so their scores stay bit-identical.
    const float e = __builtin_amdgcn_exp2f(-(z + b2) * kLog2e);
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// lpv: this lane's log-prob (lanes < n).  xs: 64 floats of LDS (16-byte aligned).
__device__ __forceinline__ void epi_finish(const FusedParams& p, int b, int lane, float lpv, int n, bool want_stats,
                                           const EpiPrefetch& e, double* dvals, float* xs) {
    xs[lane] = epi_phase_a_input(p, lane, e.xv);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int half = lane >> 5;
    float h = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) h = fmaf(e.w[i], xs[half * 32 + i], h);
    double st[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (want_stats) {
        wave_logprob_stats_regs(lpv, n, lane, st);           // n <= 64 values, one per lane: registers, DPP / readlane only
        if (p.stats && lane == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) p.stats[5 * static_cast<int64_t>(b) + i] = st[i];
        }
    }
    const float sc = epi_score(p, lane, h, [&](int d) { return e.wd[d]; }, st, want_stats && p.stats_col >= 0, e.b1, e.w2, e.b2);
    if (lane == 0) decide_and_store_impl<true>(p, b, sc, e.d);
}


// ---- the same epilogue for a kernel that cannot afford the weights in registers while it streams (k_verify<FUSED>:
// 55 more VGPRs took it to 135 and to ONE workgroup per CU).  The packed weights sit in LDS (put there by LDS-DMA at the
// kernel's start: no register is held across the stream); the decision inputs are loaded LATE -- issued by the finisher right
// before it waits for the hand-off slots, i.e. under a wait that is there anyway -- and phase A of the first layer runs under
// that wait too (epi_phase_a_lds).  Same operations in the same order as epi_finish: bit-identical scores.
constexpr int kEpiInKernelMaxK = 16;   // draft lengths the in-kernel epilogue takes: its statistics are the DPP form (one row of 16 lanes)
// The decision inputs of the in-kernel epilogue (the sequence's p_hist row, the stage costs, theta[stage]): 9 doubles that lane 0
// needs at the very end.  As registers (round 3: DecidePrefetch, issued before the hand-off wait) they are 18 VGPRs live across
// the statistics -- the register peak of k_verify<FUSED>, which has to stay within 64 for four workgroups per CU at B = 128 -- so
// they go to LDS by DMA instead (global_load_lds, one dword per lane, three instructions, no VGPR), issued under the hand-off wait,
// and are read back when the score exists (the compiler orders the read behind the DMA: vmcnt counts in order).
// dl: kEpiDecideLds doubles of LDS.
constexpr int kEpiDecideLds = 12;   // [0,4) p_hist row, [4,8) C, [8] theta[stage]
__device__ __forceinline__ void epi_decide_dma(const FusedParams& p, int b, int lane, double* dl) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int n_dp = p.prefix ? p.stage_idx + 1 : p.L;            // <= kDecidePrefetch (launcher)
    if (p.p_hist && (p.k_star || p.stop) && lane < 2 * n_dp) {
        __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(p.p_hist + static_cast<int64_t>(b) * p.L) + lane * 4),
                                         (lds_void*)reinterpret_cast<char*>(dl), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(p.C) + lane * 4),
                                         (lds_void*)(reinterpret_cast<char*>(dl) + 32), 4, 0, 0);
    }
    if (p.theta && p.thr_stop && lane < 2)
        __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(p.theta + p.stage_idx) + lane * 4),
                                         (lds_void*)(reinterpret_cast<char*>(dl) + 64), 4, 0, 0);
}
// lane 0: the registers decide_and_store_small reads (stages beyond the hierarchy's depth hold whatever the LDS held: unused)
__device__ __forceinline__ void epi_decide_from_lds(const double* dl, DecidePrefetch& d) {
#pragma unroll
    for (int i = 0; i < kDecidePrefetch; ++i) {
        d.ph[i] = dl[i];
        d.cc[i] = dl[kDecidePrefetch + i];
    }
    d.theta = dl[2 * kDecidePrefetch];
}

// what phase A leaves in registers for the rest: its accumulator and every weight the steps behind the statistics read (the
// deferred columns', b1, W2, b2), fetched from LDS NOW so that no LDS round trip stands behind the hand-off
struct EpiPhaseA {
    float h, wd[ASD_NUM_LP_STATS], b1, w2, b2;
};
// wl: the 64 * 32 + 65 packed floats in LDS (W1^T [in][hidden], b1, W2, b2); xs: 64 floats of LDS scratch
__device__ __forceinline__ void epi_phase_a_lds(const FusedParams& p, int lane, float xv, const float* wl, float* xs, EpiPhaseA& a) {
    xs[lane] = epi_phase_a_input(p, lane, xv);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int j = lane & 31, half = lane >> 5;
    const int col = p.stats_col >= 0 ? p.stats_col : 0;
#pragma unroll
    for (int d = 0; d < ASD_NUM_LP_STATS; ++d) a.wd[d] = wl[(col + d) * 32 + j];
    a.b1 = wl[64 * 32 + j];
    a.w2 = wl[64 * 32 + 32 + j];
    a.b2 = wl[64 * 32 + 64];
    float h = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) h = fmaf(wl[(half * 32 + i) * 32 + j], xs[half * 32 + i], h);
    a.h = h;
}
// a: epi_phase_a_lds's result
__device__ __forceinline__ void epi_finish_lds(const FusedParams& p, int b, int lane, float lpv, int n, bool want_stats,
                                               const double* decide_lds, const EpiPhaseA& a) {
    double st[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (want_stats) {
        wave_logprob_stats_row16(lpv, n, lane, st);      // n <= kEpiInKernelMaxK (the launcher sends longer drafts down the two-launch route)
        if (p.stats && lane == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) p.stats[5 * static_cast<int64_t>(b) + i] = st[i];
        }
    }
    ASD_EPI_STAMP(9);    // statistics done
    const float sc = epi_score(p, lane, a.h, [&](int d) { return a.wd[d]; }, st, want_stats && p.stats_col >= 0, a.b1, a.w2, a.b2);
    ASD_EPI_STAMP(11);   // sigmoid done
    if (lane == 0) {
        DecidePrefetch d;
        epi_decide_from_lds(decide_lds, d);
        decide_and_store_small(p, b, sc, d);
    }
}

// ---- the 256 -> 128 -> 1 predictor (the reference's server instantiates `QualityPredictor(feature_dim=256)`,
// src/serving/server.py:168; spec docs/guides/RESEARCH_PROTOCOL.md:315-364), latency forms: k_predictor_stop_w256x128 and the
// in-kernel epilogue of k_verify<..., EPI = 2>.  33 025 weights (129 KiB) do not fit the LDS a streaming kernel can spare and one
// wave would need ~1300 instructions for the first layer, so the layer is cut over the EIGHT waves of the (finisher's) workgroup
// -- in the verify kernel they have finished streaming and would otherwise exit:
//   wave w        columns [32 w, 32 w + 32) of the feature row (statistics columns zeroed: they come last, like the 64 x 32 form)
//                 lane j: h0 = sum_i W1[j][c_i] x[c_i],  h1 = the same for hidden unit 64 + j   (FMA chains in column order;
//                 the feature values travel lane -> SGPR by v_readlane, the weights are coalesced global loads, L2-resident)
//                 -> part[w][j], part[w][64 + j] in LDS;  workgroup barrier
//   wave 0        h = ((((((part[0] + part[1]) + part[2]) + ...) + part[7]);  the five statistics columns (phase B);
//                 + b1; ReLU;  t = fma(W2[64 + j], h1, W2[j] h0);  z = wave_sum(t);  sigmoid(z + b2)  (epi_score's sigmoid)
// CANONICAL ORDER of both latency forms (bit-identical scores); against the oracle the bar is 1e-5.
constexpr int kEpi2In = 256, kEpi2Hid = 128, kEpi2Waves = 8, kEpi2Cols = kEpi2In / kEpi2Waves;
// xv: lanes 0..31 hold the feature values of this wave's 32 columns (statistics columns zeroed); packed: GLOBAL
__device__ __forceinline__ void epi2_partial(const float* packed, int w, int lane, float xv, float& h0, float& h1) {
    h0 = 0.0f;
    h1 = 0.0f;
    const float* wp = packed + static_cast<int64_t>(w) * kEpi2Cols * kEpi2Hid + lane;
#pragma unroll 8
    for (int i = 0; i < kEpi2Cols; ++i) {
        const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xv), i));
        h0 = fmaf(wp[i * kEpi2Hid], x, h0);
        h1 = fmaf(wp[i * kEpi2Hid + 64], x, h1);
    }
}
// this wave's slice of the feature row, one value per lane (lanes < 32), statistics columns zeroed
__device__ __forceinline__ float epi2_feature(const float* feat_row, int stats_col, int w, int lane) {
    const int c = w * kEpi2Cols + (lane & 31);
    const float x = feat_row[c];
    const int si = c - stats_col;
    return (stats_col >= 0 && si >= 0 && si < ASD_NUM_LP_STATS) ? 0.0f : x;
}
struct Epi2Tail {          // what wave 0 reads from global memory for the steps behind the statistics (issued early)
    float wd0[ASD_NUM_LP_STATS], wd1[ASD_NUM_LP_STATS], b1a, b1b, w2a, w2b, b2;
};
__device__ __forceinline__ void epi2_tail_prefetch(const float* packed, int stats_col, int lane, Epi2Tail& t) {
    const int col = stats_col >= 0 ? stats_col : 0;
#pragma unroll
    for (int d = 0; d < ASD_NUM_LP_STATS; ++d) {
        t.wd0[d] = packed[(col + d) * kEpi2Hid + lane];
        t.wd1[d] = packed[(col + d) * kEpi2Hid + 64 + lane];
    }
    const float* b1 = packed + kEpi2In * kEpi2Hid;
    t.b1a = b1[lane]; t.b1b = b1[64 + lane];
    t.w2a = b1[kEpi2Hid + lane]; t.w2b = b1[kEpi2Hid + 64 + lane];
    t.b2 = b1[2 * kEpi2Hid];
}
// part: [8][128] floats of LDS (all eight partials written, workgroup barrier passed); st: the statistics (uniform)
__device__ __forceinline__ float epi2_score(const float* part, int lane, const Epi2Tail& t, const double (&st)[5], bool overlay) {
    float h0 = part[lane], h1 = part[64 + lane];
#pragma unroll
    for (int w = 1; w < kEpi2Waves; ++w) {
        h0 = h0 + part[w * kEpi2Hid + lane];
        h1 = h1 + part[w * kEpi2Hid + 64 + lane];
    }
    if (overlay) {
#pragma unroll
        for (int d = 0; d < ASD_NUM_LP_STATS; ++d) {
            const float sd = static_cast<float>(st[d]);
            h0 = fmaf(t.wd0[d], sd, h0);
            h1 = fmaf(t.wd1[d], sd, h1);
        }
    }
    h0 = fmaxf(h0 + t.b1a, 0.0f);
    h1 = fmaxf(h1 + t.b1b, 0.0f);
    const float z = wave_sum(fmaf(t.w2b, h1, t.w2a * h0));
    const float e = __builtin_amdgcn_exp2f(-(z + t.b2) * kLog2e);
    return __builtin_amdgcn_rcpf(1.0f + e);
}
// the in-kernel finish for the 256 -> 128 -> 1 form (wave 0 of the finisher's workgroup, behind the partials' barrier)
__device__ __forceinline__ void epi2_finish_lds(const FusedParams& p, int b, int lane, float lpv, int n, bool want_stats,
                                                const double* decide_lds, const float* part, const Epi2Tail& t) {
    double st[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (want_stats) {
        wave_logprob_stats_row16(lpv, n, lane, st);      // n <= kEpiInKernelMaxK
        if (p.stats && lane == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) p.stats[5 * static_cast<int64_t>(b) + i] = st[i];
        }
    }
    const float sc = epi2_score(part, lane, t, st, want_stats && p.stats_col >= 0);
    if (lane == 0) {
        DecidePrefetch d;
        epi_decide_from_lds(decide_lds, d);
        decide_and_store_small(p, b, sc, d);
    }
}

}  // namespace asd
