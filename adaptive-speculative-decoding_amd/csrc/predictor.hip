// predictor.hip -- stopping-rule quality predictor and its feature epilogue, gfx950.
// Compiled with -ffp-contract=off (numpy / CPython parity of the f64 parts); the f32 MLP uses
// explicit fmaf.
//
//   A7  log-prob statistics   src/training/generate_training_data.py:166-175  (numpy float64)
//   A8  MinimalQualityPredictor.forward, eval mode   src/minimal_adaptive_decoder.py:38-49
//   fused epilogue: stats -> features -> MLP -> Bayes -> DP rule / theta test
//
// No MFMA: the predictor is 64*32+32 = 2080 MACs per row; at B = 128 that is 0.27 M
// MACs, far below one MFMA tile's worth of launch latency.  One wave per row, hidden units on
// lanes, weights staged once per workgroup in LDS, f32 FMAs in registers.

#include "predictor_device.hpp"

namespace asd {
namespace {

constexpr int kStatsMaxK = 1024;     // values per sequence the stats path stages in LDS
constexpr int kWavesPerBlock = 4;
constexpr size_t kWeightLdsLimit = 32 * 1024;  // keep every launch under the 64 KiB default LDS window
constexpr int kFusedMaxK = 128;                   // fused epilogue: the reference caps generations at 128 tokens

__global__ __launch_bounds__(64) void k_logprob_stats(const float* __restrict__ lp, int64_t ld,
                                                      const int32_t* __restrict__ n_valid, int B, int K,
                                                      double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* vals = reinterpret_cast<double*>(smem);
    double* sorted = vals + K;
    double* sq = sorted + K;
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    int n = n_valid ? n_valid[b] : K;
    n = n < 0 ? 0 : (n > K ? K : n);
    for (int i = lane; i < n; i += 64) vals[i] = static_cast<double>(lp[static_cast<int64_t>(b) * ld + i]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double out[5];
    wave_logprob_stats(vals, sorted, sq, n, lane, out);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) stats[5 * static_cast<int64_t>(b) + i] = out[i];
    }
}

// ---- MLP: one wave per row --------------------------------------------------------------
// packed: W1T [in][hidden], b1 [hidden], W2 [hidden], b2 [1].  xs: the row, in_dim floats in LDS.
__device__ __forceinline__ float wave_mlp(const float* __restrict__ w, const float* xs, int in_dim, int hidden,
                                          int lane) {
    const float* b1 = w + static_cast<int64_t>(in_dim) * hidden;
    const float* w2 = b1 + hidden;
    float z = 0.0f;
    for (int j = lane; j < hidden; j += 64) {
        float h = b1[j];
        for (int i = 0; i < in_dim; ++i) h = fmaf(w[static_cast<int64_t>(i) * hidden + j], xs[i], h);
        h = fmaxf(h, 0.0f);                                   // ReLU; Dropout(0.1) is identity in eval
        z = fmaf(w2[j], h, z);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) z += __shfl_xor(z, off, 64);
    z += w2[hidden];                                          // b2
    return 1.0f / (1.0f + expf(-z));                          // Sigmoid
}

__device__ __forceinline__ const float* stage_weights(const float* packed, float* wlds, int n_floats, bool use_lds) {
    if (!use_lds) return packed;
    for (int i = threadIdx.x; i < n_floats; i += blockDim.x) wlds[i] = packed[i];
    __syncthreads();
    return wlds;
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void k_mlp_predict(const float* __restrict__ x, int64_t ldx,
                                                                     const float* __restrict__ packed, int B,
                                                                     int in_dim, int hidden, int use_lds,
                                                                     float* __restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* xs = reinterpret_cast<float*>(smem) + wave * in_dim;
    float* wlds = reinterpret_cast<float*>(smem) + kWavesPerBlock * in_dim;
    const int n_w = in_dim * hidden + 2 * hidden + 1;
    const float* w = stage_weights(packed, wlds, n_w, use_lds != 0);
    for (int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave; row < B;
         row += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
        for (int i = lane; i < in_dim; i += 64) xs[i] = x[row * ldx + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float sc = wave_mlp(w, xs, in_dim, hidden, lane);
        if (lane == 0) score[row] = sc;
        __builtin_amdgcn_wave_barrier();
    }
}

// Throughput form for the reference's predictor (64 -> 32 -> 1) at large batch: ONE LANE PER ROW.
// The row lives in 64 VGPRs, the 32 hidden accumulators in 32 more, and the weights never touch
// a vector register or LDS: W1T[i][0..32) is wave-uniform, so it arrives through the scalar cache
// (s_load) and feeds v_fmac as an SGPR operand.  2112 FMAs per row => VALU-bound at ~28 us per
// million rows on 256 CUs, i.e. at the HBM rate of reading the 260 B rows (the wave-per-row
// latency form above spends its time in LDS round trips: 811 us per million rows, measured).
__global__ __launch_bounds__(256) void k_mlp_rows_64x32(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ packed, int B,
                                                        float* __restrict__ score) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (row >= B) return;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const float* xr = x + row * ldx;
    float xv[64];
    if ((reinterpret_cast<uintptr_t>(xr) & 15u) == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * i);
            xv[4 * i] = v[0]; xv[4 * i + 1] = v[1]; xv[4 * i + 2] = v[2]; xv[4 * i + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 64; ++i) xv[i] = xr[i];
    }
    const float* b1 = packed + 64 * 32;
    const float* w2 = b1 + 32;
    float h[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) h[j] = b1[j];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) h[j] = fmaf(packed[i * 32 + j], xv[i], h[j]);   // packed[...] is uniform: SGPR
    }
    float z = w2[32];                                        // b2
#pragma unroll
    for (int j = 0; j < 32; ++j) z = fmaf(w2[j], fmaxf(h[j], 0.0f), z);
    score[row] = 1.0f / (1.0f + expf(-z));
}


// ---- fused epilogue: one wave per sequence (shared pieces: predictor_device.hpp) ----------
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_predictor_stop(const FusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // carve: per-wave doubles [3*K], per-wave floats [in_dim], then shared weights
    double* dbase = reinterpret_cast<double*>(smem) + static_cast<size_t>(wave) * 3 * p.K;
    float* fbase = reinterpret_cast<float*>(smem + sizeof(double) * 3 * static_cast<size_t>(p.K) * kWavesPerBlock);
    float* xs = fbase + wave * p.in_dim;
    float* wlds = fbase + kWavesPerBlock * p.in_dim;
    const int n_w = p.in_dim * p.hidden + 2 * p.hidden + 1;
    const float* w = stage_weights(p.packed, wlds, n_w, p.use_lds != 0);

    for (int b = blockIdx.x * kWavesPerBlock + wave; b < p.B; b += gridDim.x * kWavesPerBlock) {
        double st[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        const bool want_stats = p.lp && (p.stats_col >= 0 || p.stats);
        if (want_stats) {
            int n = p.n_valid ? p.n_valid[b] : p.K;
            n = n < 0 ? 0 : (n > p.K ? p.K : n);
            for (int i = lane; i < n; i += 64) dbase[i] = static_cast<double>(p.lp[static_cast<int64_t>(b) * p.ld_lp + i]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            wave_logprob_stats(dbase, dbase + p.K, dbase + 2 * p.K, n, lane, st);
#pragma unroll
            for (int i = 0; i < 5; ++i) st[i] = __shfl(st[i], 0, 64);
            if (p.stats && lane == 0) {
#pragma unroll
                for (int i = 0; i < 5; ++i) p.stats[5 * static_cast<int64_t>(b) + i] = st[i];
            }
        }
        for (int i = lane; i < p.in_dim; i += 64) {
            float v = p.feat[static_cast<int64_t>(b) * p.ldf + i];
            const int si = i - p.stats_col;
            if (want_stats && p.stats_col >= 0 && si >= 0 && si < 5) {
                v = static_cast<float>(si == 0 ? st[0] : si == 1 ? st[1] : si == 2 ? st[2] : si == 3 ? st[3] : st[4]);
            }
            xs[i] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float sc = wave_mlp(w, xs, p.in_dim, p.hidden, lane);
        if (lane == 0) decide_and_store(p, b, sc);
        __builtin_amdgcn_wave_barrier();
    }
}

// Latency form for the reference's predictor (64 -> 32 -> 1) and K <= 64: one wave per sequence,
// every global load issued up front (predictor_device.hpp: epi_prefetch / epi_finish).  This is
// what a decode step calls (B = tens of sequences): the work is a few thousand flops, so the only
// thing that matters is the length of the dependency chain.
// The first six arguments repeat the FusedParams fields the prefetch needs first (log-probs, features, packed weights): this
// file is compiled with -amdgpu-kernarg-preload-count, so they reach every wave in SGPRs at wave start and the 40-odd loads of
// the kernel are issued without waiting for an s_load of the kernarg segment (the struct is fetched meanwhile) -- the same
// device as k_verify's prologue.
__global__ __launch_bounds__(64) void k_predictor_stop_w64x32(const float* a_lp, int64_t a_ld_lp, const float* a_feat, int64_t a_ldf,
                                                              const float* a_packed, int a_K, const FusedParams p_in) {
    __shared__ double dvals[3 * 64];
    __shared__ __attribute__((aligned(16))) float xs[64];
    FusedParams p = p_in;
    p.lp = a_lp; p.ld_lp = a_ld_lp; p.feat = a_feat; p.ldf = a_ldf; p.packed = a_packed; p.K = a_K;
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const bool want_stats = p.lp && (p.stats_col >= 0 || p.stats);
    // Every load of the kernel is issued before the first is consumed.  The log-prob and the valid length are kept as raw
    // bits behind an empty asm: the compiler otherwise converts lp to f64 right behind its load -- an `s_waitcnt vmcnt(0)`,
    // one whole memory round trip, in front of the 40 other loads (ISA of the first version).
    int vb = b;
    asm volatile("" : "+v"(vb));          // vector loads: nothing here waits on the scalar path
    uint32_t raw_n = static_cast<uint32_t>(p.K), raw_lp = 0u;
    if (want_stats && p.n_valid) raw_n = reinterpret_cast<const uint32_t*>(p.n_valid)[vb];
    if (want_stats && lane < p.K) raw_lp = reinterpret_cast<const uint32_t*>(p.lp)[static_cast<int64_t>(vb) * p.ld_lp + lane];
    EpiPrefetch e;
    epi_prefetch(p, b, lane, e);
    asm volatile("" : "+v"(raw_n), "+v"(raw_lp));
    int n = __builtin_amdgcn_readfirstlane(static_cast<int>(raw_n));
    const float lpv = __uint_as_float(raw_lp);
    n = n < 0 ? 0 : (n > p.K ? p.K : n);
    epi_finish(p, b, lane, lpv, n, want_stats, e, dvals, xs);
}

// Latency form for the 256 -> 128 -> 1 predictor (the reference's server: QualityPredictor(feature_dim=256), server.py:168) and
// K <= 64: ONE workgroup of eight waves per sequence -- the first layer cut over the waves (predictor_device.hpp: epi2_partial),
// wave 0 does the statistics and the rest.  The in-kernel epilogue of k_verify<..., EPI = 2> runs the same functions in the same
// order: bit-identical scores.
__global__ __launch_bounds__(64 * kEpi2Waves) void k_predictor_stop_w256x128(const FusedParams p) {
    __shared__ float part[kEpi2Waves * kEpi2Hid];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int b = blockIdx.x;
    const bool want_stats = p.lp && (p.stats_col >= 0 || p.stats);
    const float xv = epi2_feature(p.feat + static_cast<int64_t>(b) * p.ldf, p.stats_col, wave, lane);
    uint32_t raw_n = static_cast<uint32_t>(p.K), raw_lp = 0u;
    Epi2Tail t{};
    if (wave == 0) {
        int vb = b;
        asm volatile("" : "+v"(vb));
        if (want_stats && p.n_valid) raw_n = reinterpret_cast<const uint32_t*>(p.n_valid)[vb];
        if (want_stats && lane < p.K) raw_lp = reinterpret_cast<const uint32_t*>(p.lp)[static_cast<int64_t>(vb) * p.ld_lp + lane];
        epi2_tail_prefetch(p.packed, p.stats_col, lane, t);
    }
    float h0, h1;
    epi2_partial(p.packed, wave, lane, xv, h0, h1);
    part[wave * kEpi2Hid + lane] = h0;
    part[wave * kEpi2Hid + 64 + lane] = h1;
    __syncthreads();
    if (wave != 0) return;
    asm volatile("" : "+v"(raw_n), "+v"(raw_lp));
    int n = __builtin_amdgcn_readfirstlane(static_cast<int>(raw_n));
    n = n < 0 ? 0 : (n > p.K ? p.K : n);
    double st[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (want_stats) {
        wave_logprob_stats_regs(__uint_as_float(raw_lp), n, lane, st);
        if (p.stats && lane == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) p.stats[5 * static_cast<int64_t>(b) + i] = st[i];
        }
    }
    const float sc = epi2_score(part, lane, t, st, want_stats && p.stats_col >= 0);
    if (lane == 0) decide_and_store(p, b, sc);
}

inline bool mlp_dims_ok(int in_dim, int hidden) {
    return in_dim >= 1 && hidden >= 1 && in_dim <= ASD_MAX_MLP_DIM && hidden <= ASD_MAX_MLP_DIM;
}

}  // namespace
}  // namespace asd

using namespace asd;

ASD_EXPORT size_t asd_mlp_packed_floats(int in_dim, int hidden) {
    if (in_dim < 1 || hidden < 1) return 0;
    return static_cast<size_t>(in_dim) * hidden + 2 * static_cast<size_t>(hidden) + 1;
}

ASD_EXPORT int asd_mlp_pack_weights(const float* w1, const float* b1, const float* w2, const float* b2, int in_dim,
                                    int hidden, float* packed) {
    if (!w1 || !b1 || !w2 || !b2 || !packed) return ASD_ERR_INVALID_ARG;
    if (!mlp_dims_ok(in_dim, hidden)) return ASD_ERR_UNSUPPORTED;
    for (int i = 0; i < in_dim; ++i)
        for (int j = 0; j < hidden; ++j) packed[static_cast<size_t>(i) * hidden + j] = w1[static_cast<size_t>(j) * in_dim + i];
    float* o = packed + static_cast<size_t>(in_dim) * hidden;
    for (int j = 0; j < hidden; ++j) o[j] = b1[j];
    for (int j = 0; j < hidden; ++j) o[hidden + j] = w2[j];
    o[2 * hidden] = b2[0];
    return ASD_OK;
}

ASD_EXPORT int asd_logprob_stats(const float* lp, int64_t ld, const int32_t* n_valid, int B, int K, double* stats,
                                 void* stream) {
    if (B < 0 || K < 0) return ASD_ERR_INVALID_ARG;
    if (B == 0) return ASD_OK;
    if (!stats || (K > 0 && !lp)) return ASD_ERR_INVALID_ARG;
    if (K > kStatsMaxK) return ASD_ERR_UNSUPPORTED;
    if (K > 0 && ld < K) return ASD_ERR_INVALID_ARG;
    const size_t lds = sizeof(double) * 3 * static_cast<size_t>(K > 0 ? K : 1);
    hipLaunchKernelGGL(k_logprob_stats, dim3(B), dim3(64), lds, static_cast<hipStream_t>(stream), lp, ld, n_valid, B, K,
                       stats);
    return launch_status();
}

ASD_EXPORT int asd_mlp_predict(const float* x, int64_t ldx, const float* packed_w, int B, int in_dim, int hidden,
                               float* score, void* stream) {
    if (B < 0) return ASD_ERR_INVALID_ARG;
    if (!mlp_dims_ok(in_dim, hidden)) return ASD_ERR_UNSUPPORTED;
    if (B == 0) return ASD_OK;
    if (!x || !packed_w || !score || ldx < in_dim) return ASD_ERR_INVALID_ARG;
    if (in_dim == 64 && hidden == 32 && B >= 4096) {   // throughput form: one lane per row, weights on the scalar path
        hipLaunchKernelGGL(k_mlp_rows_64x32, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx,
                           packed_w, B, score);
        return launch_status();
    }
    const size_t wbytes = asd_mlp_packed_floats(in_dim, hidden) * sizeof(float);
    const int use_lds = wbytes <= kWeightLdsLimit;
    const size_t lds = sizeof(float) * kWavesPerBlock * in_dim + (use_lds ? wbytes : 0);
    int blocks = (B + kWavesPerBlock - 1) / kWavesPerBlock;
    const int cap = current_device_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_mlp_predict, dim3(blocks), dim3(64 * kWavesPerBlock), lds, static_cast<hipStream_t>(stream), x,
                       ldx, packed_w, B, in_dim, hidden, use_lds, score);
    return launch_status();
}

ASD_EXPORT int asd_predictor_stop(const float* lp, int64_t ld_lp, const int32_t* n_valid, int K, const float* feat,
                                  int64_t ldf, int stats_col, const float* packed_w, int in_dim, int hidden,
                                  int risk_adjustment, int64_t n_obs, double alpha, double beta, double* p_hist,
                                  const double* C, double lam, int L, int stage_idx, int prefix_rule,
                                  const double* theta, int B, float* score, int32_t* k_star, uint8_t* stop,
                                  uint8_t* thr_stop, double* stats, void* stream) {
    if (B < 0 || K < 0 || L < 1 || stage_idx < 0 || stage_idx >= L) return ASD_ERR_INVALID_ARG;
    if (L > ASD_MAX_STAGES || K > kFusedMaxK) return ASD_ERR_UNSUPPORTED;
    if (!mlp_dims_ok(in_dim, hidden)) return ASD_ERR_UNSUPPORTED;
    if (B == 0) return ASD_OK;
    if (!feat || !packed_w || ldf < in_dim) return ASD_ERR_INVALID_ARG;
    if (stats_col >= 0 && stats_col + ASD_NUM_LP_STATS > in_dim) return ASD_ERR_INVALID_ARG;
    if ((stats_col >= 0 || stats) && (!lp || ld_lp < K)) return ASD_ERR_INVALID_ARG;
    if ((k_star || stop) && (!p_hist || !C)) return ASD_ERR_INVALID_ARG;
    FusedParams p{};
    p.lp = lp; p.ld_lp = ld_lp; p.n_valid = n_valid; p.K = K;
    p.feat = feat; p.ldf = ldf; p.stats_col = stats_col;
    p.packed = packed_w; p.in_dim = in_dim; p.hidden = hidden;
    const size_t wbytes = asd_mlp_packed_floats(in_dim, hidden) * sizeof(float);
    p.use_lds = wbytes <= kWeightLdsLimit;
    p.risk = risk_adjustment ? 1 : 0; p.n_obs = static_cast<double>(n_obs); p.alpha = alpha; p.beta = beta;
    p.p_hist = p_hist; p.C = C; p.lam = lam; p.L = L; p.stage_idx = stage_idx; p.prefix = prefix_rule ? 1 : 0;
    p.theta = theta; p.B = B;
    p.score = score; p.k_star = k_star; p.stop = stop; p.thr_stop = thr_stop; p.stats = stats;
    const size_t lds = sizeof(double) * 3 * static_cast<size_t>(K > 0 ? K : 1) * kWavesPerBlock +
                       sizeof(float) * kWavesPerBlock * in_dim + (p.use_lds ? wbytes : 0);
    int blocks = (B + kWavesPerBlock - 1) / kWavesPerBlock;
    const int cap = current_device_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (in_dim == 64 && hidden == 32 && K <= 64 && B <= 8192) {
        hipLaunchKernelGGL(k_predictor_stop_w64x32, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), p.lp, p.ld_lp, p.feat,
                           p.ldf, p.packed, p.K, p);
        return launch_status();
    }
    if (in_dim == kEpi2In && hidden == kEpi2Hid && K <= 64 && B <= 8192) {
        hipLaunchKernelGGL(k_predictor_stop_w256x128, dim3(B), dim3(64 * kEpi2Waves), 0, static_cast<hipStream_t>(stream), p);
        return launch_status();
    }
    hipLaunchKernelGGL(k_predictor_stop, dim3(blocks), dim3(64 * kWavesPerBlock), lds, static_cast<hipStream_t>(stream), p);
    return launch_status();
}
