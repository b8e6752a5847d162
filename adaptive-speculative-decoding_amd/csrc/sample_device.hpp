// sample_device.hpp -- device helpers shared by the sampling kernels (residual_sample.hip: the commit draw;
// draft_sample.hip: the draft tier's proposal): element unpacking, nucleus-masked accumulation, two-float row
// normalisers, the per-vector probability weights, order-preserving float keys and fixed-point masses.
#pragma once

#include "lse_device.hpp"

namespace asd {

template <int DT>
__device__ __forceinline__ void unpack(const u32x4& v, float (&x)[Elem<DT>::kPerVec]);
template <>
__device__ __forceinline__ void unpack<ASD_DTYPE_BF16>(const u32x4& v, float (&x)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[2 * i] = __uint_as_float(v[i] << 16);
        x[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
    }
}
template <>
__device__ __forceinline__ void unpack<ASD_DTYPE_F16>(const u32x4& v, float (&x)[8]) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = v[i];
        const h2 h = __builtin_bit_cast(h2, w);
        x[2 * i] = static_cast<float>(h[0]);
        x[2 * i + 1] = static_cast<float>(h[1]);
    }
}
template <>
__device__ __forceinline__ void unpack<ASD_DTYPE_F32>(const u32x4& v, float (&x)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __uint_as_float(v[i]);
}

// (m2, s) of the elements >= thr only (the nucleus of a top-p draft row); thr = -inf is the plain accumulate
template <int DT>
__device__ __forceinline__ void accum_nucleus(const u32x4& v, float thr, float c2, float& m2, float& s) {
    using E = Elem<DT>;
    if (thr == -INFINITY) {
        E::accum(v, c2, m2, s);
        return;
    }
    float x[E::kPerVec];
    unpack<DT>(v, x);
#pragma unroll
    for (int i = 0; i < E::kPerVec; ++i) x[i] = x[i] >= thr ? x[i] : -INFINITY;
    if constexpr (E::kPerVec == 8) accum8(x, c2, m2, s);
    else accum4(x, c2, m2, s);
}

// A row's normaliser L = m2 + log2(s) as TWO floats (hi + lo = the f64 value to ~1e-14).  With L rounded to one float every
// probability of the row carries the same relative error (|L| * 6e-8 * ln 2, ~4e-6 at |L| ~ 100), which is harmless for p_t or
// p_d alone but not for their DIFFERENCE on a token that holds nearly all the mass of both rows: max(0, p_t - p_d) then has
// that absolute error against a true value of maybe 1e-3.  The exponent is formed as fma(x, c2, -hi) - lo: the fma result is
// exact to its own (small) magnitude, so the residual keeps ~1e-7 relative accuracy on near-deterministic rows too.
struct Norm2 { float hi, lo; };
__device__ __forceinline__ Norm2 norm2_of(float m2, float s) {
    const double L = static_cast<double>(m2) + log2_split(s);
    Norm2 n;
    n.hi = static_cast<float>(L);
    n.lo = static_cast<float>(L - static_cast<double>(n.hi));
    return n;
}
// weights of one 16-byte vector: w_i = max(0, p_t - p_d), and p_t itself; returns the lane's sums
template <int DT>
__device__ __forceinline__ void vector_weights(const u32x4& vt, const u32x4& vd, bool has_d, float c2, Norm2 Lt, Norm2 Ld,
                                               float tthr, float dthr,
                                               float (&w)[Elem<DT>::kPerVec], float (&pt)[Elem<DT>::kPerVec]) {
    constexpr int N = Elem<DT>::kPerVec;
    float xt[N], xd[N];
    unpack<DT>(vt, xt);
    unpack<DT>(vd, xd);             // (by value, not through an optional pointer: that form went through scratch memory)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // outside a row's nucleus the probability is exactly 0 (Lt / Ld are then the nucleus normalisers)
        pt[i] = xt[i] >= tthr ? fast_exp2(fmaf(xt[i], c2, -Lt.hi) - Lt.lo) : 0.0f;
        const float pd = (has_d && xd[i] >= dthr) ? fast_exp2(fmaf(xd[i], c2, -Ld.hi) - Ld.lo) : 0.0f;
        w[i] = fmaxf(pt[i] - pd, 0.0f);
    }
}

constexpr int kDsDigits = 4096;                 // histogram slots (12-bit digit)
constexpr float kDsFix = 1099511627776.0f;      // 2^40
constexpr int kDrThreads = 1024;
constexpr int kDrWaves = kDrThreads / 64;
constexpr int kDrMaxTiles = 2048;               // 64-vector tiles per row the LDS mass array holds (V <= 1 M bf16 elements)
constexpr int kDrSeg = 1536;                    // candidate tokens one wave can list in LDS (16 waves x 6 KB)

__device__ __forceinline__ uint32_t order_key(float x) {
    x += 0.0f;                                            // -0 -> +0: equal values share one key
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_floor_value(uint32_t key) {   // smallest float whose key is >= `key`
    const uint32_t u = (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key;
    return __uint_as_float(u);
}
// floor(pr * 2^40) for 0 <= pr <~ 1 without the generic (software) f32 -> u64 conversion: two f32 -> u32 conversions
__device__ __forceinline__ unsigned long long mass_fixed40(float pr) {
    const float y = pr * 1048576.0f;                       // * 2^20
    const uint32_t hi = static_cast<uint32_t>(y);
    const uint32_t lo = static_cast<uint32_t>((y - static_cast<float>(hi)) * 1048576.0f);
    return (static_cast<unsigned long long>(hi) << 20) | lo;
}
// inclusive prefix sum over the 64 lanes (the DPP sequence of wave_sum: row_shr 1,2,4,8, row_bcast 15 / 31)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    v += dpp_move_i32<0x111, 0xf>(v);
    v += dpp_move_i32<0x112, 0xf>(v);
    v += dpp_move_i32<0x114, 0xf>(v);
    v += dpp_move_i32<0x118, 0xf>(v);
    v += dpp_move_i32<0x142, 0xa>(v);
    v += dpp_move_i32<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}


// ---- pieces shared by both kernels -----------------------------------------------------------------------------------
// The CANONICAL pair of one 64-vector tile: (M, s) with  sum over the tile of 2^(x c2) = s 2^M.  Every lane reduces its own
// vector against its own maximum, the wave combines max-first (DPP): the value depends on the tile's bytes only, not on
// which wave of which workgroup computes it.  A ragged last tile is padded with -inf vectors (no mass).
template <int DT>
__device__ __forceinline__ void tile_pair(const u32x4& vec, float c2, float& M, float& sw, float thr = -INFINITY) {
    constexpr int N = Elem<DT>::kPerVec;
    float x[N];
    unpack<DT>(vec, x);
    if (thr != -INFINITY) {                                // a nucleus-truncated row: elements below the threshold carry no mass
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = x[i] >= thr ? x[i] : -INFINITY;
    }
    float vmax = x[0];
#pragma unroll
    for (int i = 1; i < N; ++i) vmax = fmaxf(vmax, x[i]);
    const float ml = fmaxf(vmax * c2, kSentinel);          // a lane of -inf logits: finite sentinel, every term 0
    float sl = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) sl += fast_exp2(fmaf(x[i], c2, -ml));
    M = wave_max(ml);
    sw = wave_sum(sl * fast_exp2(ml - M));
}
// (m2, s) of the row from its tile pairs in a FIXED order (wave w folds tiles w, w + 16, ... one per lane, then the lanes,
// then the 16 waves): the same bits whichever kernel / geometry produced the pairs.  All 1024 threads call it.
__device__ __forceinline__ void fold_tile_pairs(const float* tm, const float* ts, int n_tiles, float (*red)[2], int wave, int lane,
                                                float& m2, float& s) {
    float wm = kSentinel, ws = 0.0f;
    for (int tile = wave + kDrWaves * lane; tile < n_tiles; tile += kDrWaves * 64) ms_merge(wm, ws, tm[tile], ts[tile]);
    wave_merge(wm, ws);
    if (lane == 0) { red[wave][0] = wm; red[wave][1] = ws; }
    __syncthreads();
    m2 = red[0][0];
    s = red[0][1];
#pragma unroll
    for (int w = 1; w < kDrWaves; ++w) ms_merge(m2, s, red[w][0], red[w][1]);
}
__device__ __forceinline__ double wave_incl_scan_f64(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// ---- cross-workgroup mailboxes of the group kernels (k_draft_group, k_residual_group): single writer, single reader,
// self-tagging words (non-zero = published), handed back empty by the reader
constexpr int kDgMaxGroups = 32;          // workgroups per row
constexpr int kDgMaxLevels = 3;           // histogram rounds (f32 keys: 12 + 12 + 8 bits)
constexpr int kDgMsgs = 1 + kDgMaxLevels; // mailbox messages per partner: (m2, s), then one decision per round
constexpr int kDgSlots = 256;             // rows * groups the histogram exchange area is sized for (one workgroup per CU)
constexpr int kDgSpinLimit = 1 << 19;    // polls (~1 us each) of a word whose store is in flight before the row is given up
constexpr unsigned long long kDgValid = 1ull << 63;

// bounded wait for a mailbox word (non-zero = published).  `lost` (LDS) is raised by the first wait that runs out and makes
// every later wait of the workgroup return at once: a row whose partner never shows up costs ONE timeout, not one per word.
// `status`: the workspace's sticky status word (its first 32 bits; include/asd_hip.h, asd_workspace_status) -- the wait that runs
// out or-s ASD_WS_LOST_HANDOFF into it, so that the host learns of the loss at its next synchronisation: the poisoned outputs
// (tok = -1, lp = NaN) say WHICH row, the status word says THAT the workspace is no longer all-zero (the late word will never be
// handed back empty) and must be re-initialised before its next use.
__device__ __forceinline__ unsigned long long dg_poll(unsigned long long* slot, volatile int* lost, uint32_t* status) {
    unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spins = 0; v == 0ull && spins < kDgSpinLimit && !*lost; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (v == 0ull) {
        if (!*lost && status) __hip_atomic_fetch_or(status, static_cast<uint32_t>(ASD_WS_LOST_HANDOFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lost = 1;
    }
    return v;
}
constexpr size_t kWorkspaceHeaderBytes = 256;   // every hand-off workspace begins with this block: u32 status word at +0, rest reserved
__device__ __forceinline__ void dg_put(unsigned long long* slot, unsigned long long v) {
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


}  // namespace asd
