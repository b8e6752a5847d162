// residual_sample.hip -- the token a speculative step COMMITS after the accepted prefix, gfx950.
//
// Standard speculative sampling: at the first rejected position j = n_acc[b] the emitted token is
// drawn from the residual distribution  w(v) = max(0, p_t(v) - p_d(v))  (p_t, p_d: target / draft
// softmax at that position, temperature folded in); when all K drafted tokens were accepted it is
// drawn from the target's next-token distribution (`bonus_logits`).  The draw is an inverse CDF
// in vocabulary order against a caller-supplied uniform r[b]:  token = min{v : cdf(v) > r * total}.
// No reference symbol exists for this (the reference has no token-level verification, SURVEY.md
// F2): specified here and in DESIGN.md §2, checked against oracle/asd_oracle.c (parity unpinned).
//
// Only B rows (not B*K) are touched, so each row is cut over S workgroups and the work is three
// stream-ordered launches (B = 32, V = 152064: 2 x 9.7 MB streamed twice, the second time from
// L2 / Infinity Cache):
//   k_rs_lse   per (sequence, slice): log2-domain (m2, s) of the target row and the draft row
//   k_rs_mass  per (sequence, slice): fold the slices -> L_t, L_d; per 64-vector TILE the residual
//              mass  Z = sum max(0, 2^(x_t c2 - L_t) - 2^(x_d c2 - L_d))  and the target mass P
//   k_rs_pick  per sequence, one wave: prefix over the tile masses -> tile of the draw; recompute
//              that tile's weights with the SAME float operations; prefix inside the tile -> token
// Rows must be 16-byte aligned and a whole number of 16-byte vectors (true for every lm_head
// output; V = 152064 bf16 is 19008 vectors); otherwise ASD_ERR_ALIGNMENT.

#include "lse_device.hpp"

namespace asd {
namespace {

constexpr int kRsThreads = 256;
constexpr int kRsWaves = kRsThreads / 64;

struct RsParams {
    const void* t_logits; int64_t ld_t;
    const void* d_logits; int64_t ld_d;
    const void* bonus; int64_t ld_b;
    const int32_t* n_acc;
    const float* r;
    int B, K, V, S;
    float c2;
    int nvec, n_tiles;        // 16-byte vectors per row; tiles of 64 vectors
    float4* partial;          // [B][S]  (m2_t, s_t, m2_d, s_d)
    float2* tiles;            // [B][n_tiles]  (Z, P)
    int32_t* token;
    const float* d_thr;       // [B*K] nucleus thresholds of the draft rows (logit < thr => p_d = 0, renormalised) or nullptr
};

template <int DT>
struct Rows {
    const char* xt;   // target row (or the bonus row), nullptr => nothing to sample from
    const char* xd;   // draft row, nullptr => p_d == 0 (bonus draw)
    float tthr, dthr; // nucleus thresholds of the two rows (-inf: the whole vocabulary)
};

template <int DT>
__device__ __forceinline__ Rows<DT> select_rows(const RsParams& p, int b) {
    using E = Elem<DT>;
    const int j = p.n_acc[b];
    Rows<DT> r{nullptr, nullptr, -INFINITY, -INFINITY};
    if (j >= 0 && j < p.K) {
        r.xt = static_cast<const char*>(p.t_logits) + (static_cast<int64_t>(b) * p.K + j) * p.ld_t * E::kBytes;
        r.xd = static_cast<const char*>(p.d_logits) + (static_cast<int64_t>(b) * p.K + j) * p.ld_d * E::kBytes;
        if (p.d_thr) r.dthr = p.d_thr[static_cast<int64_t>(b) * p.K + j];
    } else if (p.bonus) {
        r.xt = static_cast<const char*>(p.bonus) + static_cast<int64_t>(b) * p.ld_b * E::kBytes;
    }
    return r;
}

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

__device__ __forceinline__ void slice_tiles(int n_tiles, int s, int S, int& t0, int& t1) {
    // 32-bit: n_tiles * S < 2^31 for any row the launcher accepts (a 64-bit software division is ~150 scalar
    // instructions in front of the first load of a latency-bound kernel)
    t0 = static_cast<int>(static_cast<uint32_t>(n_tiles) * static_cast<uint32_t>(s) / static_cast<uint32_t>(S));
    t1 = static_cast<int>(static_cast<uint32_t>(n_tiles) * static_cast<uint32_t>(s + 1) / static_cast<uint32_t>(S));
}

// ---- pass 1: per-slice (m2, s) of both rows -------------------------------------------------
template <int DT>
__global__ __launch_bounds__(kRsThreads) void k_rs_lse(const RsParams p) {
    __shared__ float red[kRsWaves][4];
    const int b = blockIdx.y, s = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Rows<DT> rows = select_rows<DT>(p, b);
    int t0, t1;
    slice_tiles(p.n_tiles, s, p.S, t0, t1);
    float mt = kSentinel, st = 0.0f, md = kSentinel, sd = 0.0f;
    if (rows.xt) {
        const u32x4* vt = reinterpret_cast<const u32x4*>(rows.xt);
        const u32x4* vd = reinterpret_cast<const u32x4*>(rows.xd);
        for (int t = t0 + wave; t < t1; t += kRsWaves) {
            const int v = t * 64 + lane;
            if (v < p.nvec) {
                accum_nucleus<DT>(vt[v], rows.tthr, p.c2, mt, st);
                if (vd) accum_nucleus<DT>(vd[v], rows.dthr, p.c2, md, sd);
            }
        }
    }
    wave_merge(mt, st);
    wave_merge(md, sd);
    if (lane == 0) { red[wave][0] = mt; red[wave][1] = st; red[wave][2] = md; red[wave][3] = sd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float amt = red[0][0], ast = red[0][1], amd = red[0][2], asd_ = red[0][3];
        for (int w = 1; w < kRsWaves; ++w) {
            ms_merge(amt, ast, red[w][0], red[w][1]);
            ms_merge(amd, asd_, red[w][2], red[w][3]);
        }
        p.partial[static_cast<int64_t>(b) * p.S + s] = make_float4(amt, ast, amd, asd_);
    }
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
// L of a whole row from its S slice partials (fixed order; every caller gets the same bits)
__device__ __forceinline__ void row_norms(const RsParams& p, int b, Norm2& Lt, Norm2& Ld) {
    float mt = kSentinel, st = 0.0f, md = kSentinel, sd = 0.0f;
    for (int s = 0; s < p.S; ++s) {
        const float4 q = p.partial[static_cast<int64_t>(b) * p.S + s];
        ms_merge(mt, st, q.x, q.y);
        ms_merge(md, sd, q.z, q.w);
    }
    Lt = norm2_of(mt, st);
    Ld = norm2_of(md, sd);
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

// ---- pass 2: per-tile residual mass Z and target mass P ----------------------------------------
template <int DT>
__global__ __launch_bounds__(kRsThreads) void k_rs_mass(const RsParams p) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    const int b = blockIdx.y, s = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Rows<DT> rows = select_rows<DT>(p, b);
    int t0, t1;
    slice_tiles(p.n_tiles, s, p.S, t0, t1);
    Norm2 Lt, Ld;
    row_norms(p, b, Lt, Ld);
    const u32x4* vt = reinterpret_cast<const u32x4*>(rows.xt);
    const u32x4* vd = reinterpret_cast<const u32x4*>(rows.xd);
    for (int t = t0 + wave; t < t1; t += kRsWaves) {
        const int v = t * 64 + lane;
        float z = 0.0f, q = 0.0f;
        if (rows.xt && v < p.nvec) {
            float w[N], pt[N];
            const u32x4 a = vt[v];
            u32x4 d = {0u, 0u, 0u, 0u};
            if (vd) d = vd[v];
            vector_weights<DT>(a, d, vd != nullptr, p.c2, Lt, Ld, rows.tthr, rows.dthr, w, pt);
#pragma unroll
            for (int i = 0; i < N; ++i) { z += w[i]; q += pt[i]; }
        }
        z = wave_sum(z);
        q = wave_sum(q);
        if (lane == 0) p.tiles[static_cast<int64_t>(b) * p.n_tiles + t] = make_float2(z, q);
    }
}

// ---- pass 3: inverse CDF (one wave) --------------------------------------------------------------
// tl: the row's per-tile masses (Z, P) -- in the workspace (k_rs_pick) or in LDS (k_residual_row); Lt / Ld: the rows' normalisers
struct RsPickScratch {
    double chunk[64];
    double lane_mass[64];
    int sel_tile;
    double sel_rest;
};
template <int DT>
__device__ __forceinline__ void rs_pick_wave(const RsParams& p, int b, int lane, const Rows<DT>& rows, const float2* tl, Norm2 Lt,
                                             Norm2 Ld, RsPickScratch& sc) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    double (&chunk)[64] = sc.chunk;
    double (&lane_mass)[64] = sc.lane_mass;
    int& sel_tile = sc.sel_tile;
    double& sel_rest = sc.sel_rest;
    // lane l owns tiles [l*per, (l+1)*per): chunk sums of Z and of P
    const int per = (p.n_tiles + 63) / 64;
    const int c0 = lane * per, c1 = min(c0 + per, p.n_tiles);
    double cz = 0.0, cp = 0.0;
    for (int t = c0; t < c1; ++t) { cz += static_cast<double>(tl[t].x); cp += static_cast<double>(tl[t].y); }
    // totals (fixed order through LDS)
    chunk[lane] = cz;
    lane_mass[lane] = cp;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double totz = 0.0, totp = 0.0;
    for (int l = 0; l < 64; ++l) { totz += chunk[l]; totp += lane_mass[l]; }
    const bool use_p = !(totz > 0.0);                  // empty residual (p_t <= p_d everywhere): draw from p_t
    const double total = use_p ? totp : totz;
    const double mine = use_p ? cp : cz;
    double target = static_cast<double>(p.r[b]) * total;
    if (!(target >= 0.0)) target = 0.0;
    // exclusive prefix of chunk masses over lanes, lane of the draw
    __builtin_amdgcn_wave_barrier();
    chunk[lane] = mine;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double before = 0.0;
    for (int l = 0; l < lane; ++l) before += chunk[l];
    const bool holds = mine > 0.0 && target >= before && target < before + mine;
    unsigned long long bal = __ballot(holds);
    if (bal == 0) {                                    // rounding pushed the draw past the end: last chunk with mass
        bal = __ballot(mine > 0.0);
        if (bal == 0) { if (lane == 0) p.token[b] = -1; return; }
        bal = 1ull << (63 - __builtin_clzll(bal));
    }
    const int owner = __builtin_ctzll(bal);
    if (lane == owner) {
        double acc = before;
        int pick = -1, last_pos = -1;
        for (int t = c0; t < c1; ++t) {
            const double m = static_cast<double>(use_p ? tl[t].y : tl[t].x);
            if (m > 0.0) {
                last_pos = t;
                if (target < acc + m) { pick = t; break; }
                acc += m;
            }
        }
        if (pick < 0) { pick = last_pos; acc -= static_cast<double>(use_p ? tl[pick].y : tl[pick].x); }
        sel_tile = pick;
        sel_rest = target - acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int tile = sel_tile;
    const double rest = sel_rest;

    // the chosen tile: the same float weights as pass 2, prefix over lanes then inside the lane
    const u32x4* vt = reinterpret_cast<const u32x4*>(rows.xt);
    const u32x4* vd = reinterpret_cast<const u32x4*>(rows.xd);
    const int v = tile * 64 + lane;
    float w[N], pt[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { w[i] = 0.0f; pt[i] = 0.0f; }
    if (v < p.nvec) {
        const u32x4 a = vt[v];
        u32x4 d = {0u, 0u, 0u, 0u};
        if (vd) d = vd[v];
        vector_weights<DT>(a, d, vd != nullptr, p.c2, Lt, Ld, rows.tthr, rows.dthr, w, pt);
    }
    double lm = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) lm += static_cast<double>(use_p ? pt[i] : w[i]);
    __builtin_amdgcn_wave_barrier();
    lane_mass[lane] = lm;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double lb = 0.0;
    for (int l = 0; l < lane; ++l) lb += lane_mass[l];
    const bool lholds = lm > 0.0 && rest >= lb && rest < lb + lm;
    unsigned long long lbal = __ballot(lholds);
    if (lbal == 0) {
        lbal = __ballot(lm > 0.0);
        if (lbal == 0) { if (lane == 0) p.token[b] = -1; return; }
        lbal = 1ull << (63 - __builtin_clzll(lbal));
    }
    if (lane == __builtin_ctzll(lbal)) {
        double acc = lb;
        int pick = -1, last_pos = -1;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double m = static_cast<double>(use_p ? pt[i] : w[i]);
            if (m > 0.0 && pick < 0) {
                last_pos = i;
                if (rest < acc + m) pick = i;
                else acc += m;
            }
        }
        if (pick < 0) pick = last_pos;
        p.token[b] = v * N + pick;
    }
}

template <int DT>
__global__ __launch_bounds__(64) void k_rs_pick(const RsParams p) {
    __shared__ RsPickScratch sc;
    const int b = blockIdx.x, lane = threadIdx.x;
    const Rows<DT> rows = select_rows<DT>(p, b);
    if (!rows.xt) {
        if (lane == 0) p.token[b] = -1;
        return;
    }
    Norm2 Lt, Ld;
    row_norms(p, b, Lt, Ld);
    rs_pick_wave<DT>(p, b, lane, rows, p.tiles + static_cast<int64_t>(b) * p.n_tiles, Lt, Ld, sc);
}

// ---- asd_draft_sample: ONE launch, one 1024-lane workgroup per row ---------------------------------------------
// The proposal step touches one row (304 KB at V = 152064 bf16) per sequence, so it is latency- and issue-bound, not
// bandwidth-bound: the first version spread every row over S workgroups and needed 8-9 dependent launches (72 us at
// B = 32, 171 us at B = 128, mostly launch gaps and cross-workgroup atomics).  Here a row never leaves its workgroup:
// after the first sweep it is L2-resident, the phases are separated by workgroup barriers, nothing crosses workgroups,
// nothing is atomic outside LDS.  One instruction per 16-byte vector per sweep costs the CU ~0.15 us (19 vectors per lane,
// 4 waves per SIMD, 4 cycles per wave64 op), so the design rule is: sweep the row as few times as possible and keep the
// per-element work in a sweep to compares.
//   1. (m2, s) of the row                                   -> L, the softmax normaliser            (sweep, exp per element)
//   2. top-p only: tokens below the mass floor (1 - top_p) / V cannot be inside the nucleus; a compare-only sweep lists the
//      rest (the CANDIDATES: a few hundred tokens of a peaked LLM row, ~5 % of a Gaussian one) in LDS, per wave, tile by
//      tile, in a fixed order                                                                         (sweep, compares)
//   3. radix select on probability MASS over the candidates.  Per level (12 + 12 [+ 8] bits of the order-preserving key
//      of the f32 logit; the last level is constant for 16-bit logits and skipped) every candidate adds its probability,
//      as 2^-40 fixed point, to the LDS histogram slot of its digit (ds_add_u64: integer adds commute => bitwise
//      reproducible); the histogram is scanned from the top digit down for the digit where the cumulative mass reaches
//      top_p * total.  The nucleus is { v : x_v >= x* },  x* = the smallest value of the selected key bucket; the mass the
//      last scan has accumulated is the nucleus normaliser L_N (no pass of its own).
//   4. per 64-vector tile the mass of the nucleus-restricted softmax, from the candidate lists (no truncation: a second
//      sweep with an exp per element)
//   5. one wave: prefix over the tile masses -> tile of the draw -> recompute that tile -> lane -> element; log q(tok)
// A row too flat for the lists (a wave would hold more than kDrSeg candidates) runs 3 and 4 as sweeps instead.
// Phase times of workgroup 0 (tools/stamp_draft.py, B = 32, V = 152064 bf16, T = 0.7, top_p = 0.9, N(0, 3) logits):
// 1: 8.2 us, 2: 13.6, 3: 4.0 + 2.0 (scan) + 3.0 + 1.9, 4: 4.7, 5: 3.3  => 41 us; the all-sweep form of 2-4 took 70.
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

#ifdef ASD_STAMP
// Diagnostic build only (tools/stamp_draft.py builds a separate .so with -DASD_STAMP): phase boundaries of workgroup 0.
__device__ unsigned long long g_dr_stamp[16];
#define ASD_DR_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dr_stamp[slot] = wall_clock64(); } while (0)
#else
#define ASD_DR_STAMP(slot) do { } while (0)
#endif

struct DrParams {
    const void* logits; int64_t ld;
    const float* r;
    int B, V, nvec, n_tiles;
    float c2, top_p;
    int levels;              // 0: no truncation; 2: 16-bit logits; 3: f32 logits
    int32_t* tok; float* lp; float* thr;
};

template <int DT>
__global__ __launch_bounds__(kDrThreads) void k_draft_row(const DrParams p) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    __shared__ unsigned long long hist[kDsDigits];
    __shared__ float tile_mass[kDrMaxTiles];
    __shared__ float red[kDrWaves][2];
    __shared__ unsigned long long wave_tot[kDrWaves];
    __shared__ unsigned long long sel_above, sel_incl;
    __shared__ int sel_digit;
    __shared__ int pick_tile;
    __shared__ double pick_rest;
    __shared__ uint32_t cand[kDrWaves][kDrSeg];       // per wave: ids of the tokens above the mass floor, tile by tile
    __shared__ uint32_t tile_span[kDrMaxTiles];       // (first candidate << 16) | candidates of the tile, in its wave's list
    __shared__ int overflow;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const u32x4* row = reinterpret_cast<const u32x4*>(static_cast<const char*>(p.logits) + static_cast<int64_t>(b) * p.ld * E::kBytes);
    // fn(v, vector) for this thread's vectors v = t, t + 1024, ... (after the first sweep the row is L2-resident).  The trip
    // count is the same for all lanes of a wave -- a ragged last tile is padded with -inf vectors, which carry no mass
    // anywhere -- so the wave reductions inside `fn` always run with every lane active.  Four of a thread's vectors are
    // loaded before the first is consumed: with one load in flight per thread a sweep is 19 dependent L2 round trips.
    // (Holding the row in registers instead, 19 x 16 B per lane, was tried: it spills at the 128-VGPR budget of a
    // 1024-lane workgroup and was slower.)
    auto for_each = [&](auto&& fn) {
        constexpr int kAhead = 4;
        const u32x4 neg = {E::kNegInfWord, E::kNegInfWord, E::kNegInfWord, E::kNegInfWord};
        for (int v0 = t - lane; v0 < p.nvec; v0 += kAhead * kDrThreads) {
            u32x4 q[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const int v = v0 + j * kDrThreads + lane;
                q[j] = v < p.nvec ? row[v] : neg;
            }
#pragma unroll
            for (int j = 0; j < kAhead; ++j)
                if (v0 + j * kDrThreads < p.nvec) fn(v0 + j * kDrThreads + lane, q[j]);     // wave-uniform guard
        }
    };
    auto any_at_least = [&](const float (&x)[N], float bound) -> bool {
        bool any = false;
#pragma unroll
        for (int i = 0; i < N; ++i) any = any || (x[i] >= bound);
        return __ballot(any) != 0ull;
    };
    // (m2, s) of the row, combined over the workgroup in a fixed order; every thread returns the same pair
    auto block_lse = [&](float& m2, float& s) {
        float lm = kSentinel, ls = 0.0f;
        for_each([&](int, const u32x4& q) { E::accum(q, p.c2, lm, ls); });
        wave_merge(lm, ls);
        if (lane == 0) { red[wave][0] = lm; red[wave][1] = ls; }
        __syncthreads();
        m2 = red[0][0];
        s = red[0][1];
#pragma unroll
        for (int w = 1; w < kDrWaves; ++w) ms_merge(m2, s, red[w][0], red[w][1]);
    };

    float m2, s;
    ASD_DR_STAMP(0);
    // Without truncation the first sweep also leaves every tile's OWN (max, sum) pair in LDS: the tile masses then follow
    // from L without a second exp-per-element sweep of the row (24 -> 15 us at B <= 32).
    const bool tiles_from_sweep1 = p.levels == 0;
    if (tiles_from_sweep1) {
        for_each([&](int v, const u32x4& vec) {
            float x[N];
            unpack<DT>(vec, x);
            float vmax = x[0];
#pragma unroll
            for (int i = 1; i < N; ++i) vmax = fmaxf(vmax, x[i]);
            const float ml = fmaxf(vmax * p.c2, kSentinel);   // a lane of -inf logits: finite sentinel, every term 0
            float sl = 0.0f;
#pragma unroll
            for (int i = 0; i < N; ++i) sl += fast_exp2(fmaf(x[i], p.c2, -ml));
            const float M = wave_max(ml);
            const float sw = wave_sum(sl * fast_exp2(ml - M));
            if (lane == 0) {                                  // nothing is carried from tile to tile: the four tiles of a
                tile_span[v >> 6] = __float_as_uint(M);       // for_each step reduce side by side
                tile_mass[v >> 6] = sw;
            }
        });
        // the wave's pair from its own tiles (wave + 16 j), one per lane, in a fixed order
        float wm = kSentinel, ws = 0.0f;
        for (int tile = wave + kDrWaves * lane; tile < p.n_tiles; tile += kDrWaves * 64)
            ms_merge(wm, ws, __uint_as_float(tile_span[tile]), tile_mass[tile]);
        wave_merge(wm, ws);
        if (lane == 0) { red[wave][0] = wm; red[wave][1] = ws; }
        __syncthreads();
        m2 = red[0][0];
        s = red[0][1];
#pragma unroll
        for (int w = 1; w < kDrWaves; ++w) ms_merge(m2, s, red[w][0], red[w][1]);
    } else {
        block_lse(m2, s);
    }
    ASD_DR_STAMP(1);
    double L64 = static_cast<double>(m2) + log2_split(s);      // log2 of the normaliser of the distribution drawn from
    float thr = -INFINITY;
    bool listed = false;       // the candidates of the row are in `cand`: the remaining phases walk the list, not the row
    int wcnt = 0;              // candidates in this wave's list (wave-uniform)
    // fn(slot, x) for this wave's candidates cand[wave][slot], slot in [0, count): 64 per step, four steps' logits gathered
    // (L2 hits) before the first is used; lanes past the end see slot = -1, x = -inf
    auto for_cand = [&](int count, auto&& fn) {
        constexpr int kAhead = 4;
        for (int e0 = 0; e0 < count; e0 += kAhead * 64) {
            float x[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const int e = e0 + j * 64 + lane;
                x[j] = E::scalar(row, e < count ? cand[wave][e] : 0u);
            }
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const int e = e0 + j * 64 + lane;
                if (e0 + j * 64 < count) fn(e < count ? e : -1, e < count ? x[j] : -INFINITY);
            }
        }
    };
    if (p.levels > 0 && s > 0.0f) {
        const float L = static_cast<float>(L64);
        const float p_floor = fmaxf((1.0f - p.top_p) / static_cast<float>(p.V), 1.0f / kDsFix);
        const float x_floor = (L + __builtin_amdgcn_logf(p_floor)) / p.c2;      // p >= p_floor  <=>  x >= x_floor
        // ---- candidates.  Tokens below p_floor = (1 - top_p) / V carry < 1 - top_p together, so the threshold lies above
        // all of them and they can never be drawn.  One compare-only sweep lists the others (per-lane counts, DPP prefix
        // sum, lane-major inside a tile: a fixed order) in the LDS segment of the wave that owns their tile; the histogram
        // levels, the nucleus normaliser and the tile masses then cost a few candidates per lane instead of a sweep of
        // divergent per-element work.  A row too flat for the lists (some wave holds more than kDrSeg candidates) keeps
        // the sweeps.
        if (t == 0) overflow = 0;
        __syncthreads();
        bool over = false;
        for_each([&](int v, const u32x4& vec) {
            float x[N];
            unpack<DT>(vec, x);
            uint32_t keep = 0u;                           // bit i: element i of this lane's vector is a candidate
#pragma unroll
            for (int i = 0; i < N; ++i) keep |= (x[i] >= x_floor ? 1u : 0u) << i;
            const int first = wcnt;
            if (__ballot(keep != 0u) != 0ull) {
                const int cnt = __builtin_popcount(keep);
                const int incl = wave_incl_scan_i32(cnt);
                const int n = __builtin_amdgcn_readlane(incl, 63);
                if (wcnt + n <= kDrSeg) {
                    int at = wcnt + incl - cnt;           // lane-major inside the tile: a fixed order
                    while (keep != 0u) {                  // as many rounds as the fullest lane has candidates (1-3, not N)
                        cand[wave][at++] = static_cast<uint32_t>(v * N + __builtin_ctz(keep));
                        keep &= keep - 1u;
                    }
                    wcnt += n;
                } else {
                    over = true;
                }
            }
            if (lane == 0) tile_span[v >> 6] = (static_cast<uint32_t>(first) << 16) | static_cast<uint32_t>(wcnt - first);
        });
        if (over && lane == 0) overflow = 1;
        __syncthreads();
        listed = overflow == 0;
        ASD_DR_STAMP(12);

        unsigned long long above = 0ull, target = 0ull;
        uint32_t prefix = 0u;
        bool empty = false;
        const int shifts[3] = {20, 8, 0}, widths[3] = {12, 12, 8};
        for (int lv = 0; lv < p.levels; ++lv) {
            const int shift = shifts[lv], digits = 1 << widths[lv], hi_shift = shifts[lv] + widths[lv];
            for (int i = t; i < digits; i += kDrThreads) hist[i] = 0ull;
            if (t == 0) sel_digit = -1;
            __syncthreads();
            ASD_DR_STAMP(2 + 2 * lv);
            // every candidate adds its probability, 2^-40 fixed point, to the slot of its digit
            auto add_mass = [&](int, float x) {
                if (!(x >= x_floor)) return;
                const uint32_t key = order_key(x);
                const bool mine = hi_shift >= 32 || (key >> hi_shift) == (prefix >> hi_shift);
                if (mine) atomicAdd(&hist[(key >> shift) & (digits - 1)], mass_fixed40(fast_exp2(fmaf(x, p.c2, -L))));
            };
            if (listed) {
                for_cand(wcnt, add_mass);
            } else {
                for_each([&](int, const u32x4& vec) {
                    float x[N];
                    unpack<DT>(vec, x);
                    if (!any_at_least(x, x_floor)) return;
#pragma unroll
                    for (int i = 0; i < N; ++i) add_mass(0, x[i]);
                });
            }
            __syncthreads();
            ASD_DR_STAMP(3 + 2 * lv);
            // thread t owns the t-th chunk of digits counted from the TOP
            const int per = digits >= kDrThreads ? digits / kDrThreads : 1;
            const int hi = digits - t * per, lo = hi - per;
            unsigned long long mine = 0ull;
            if (lo >= 0)
                for (int j = lo; j < hi; ++j) mine += hist[j];
            const unsigned long long incl = wave_incl_scan_u64(mine, lane);
            if (lane == 63) wave_tot[wave] = incl;
            __syncthreads();
            unsigned long long base = 0ull, total = 0ull;
#pragma unroll
            for (int w = 0; w < kDrWaves; ++w) {
                if (w < wave) base += wave_tot[w];
                total += wave_tot[w];
            }
            if (lv == 0) {   // the probabilities sum to 1 = 2^40 fixed point (the histogram only holds the tokens above p_floor)
                target = static_cast<unsigned long long>(static_cast<double>(p.top_p) * static_cast<double>(kDsFix));
                if (target > total) target = total;      // fixed-point truncation: never ask for more than is there
                if (target == 0ull) target = 1ull;
                empty = total == 0ull;
            }
            const unsigned long long before = above + base + incl - mine;
            if (lo >= 0 && mine > 0ull && before < target && target <= before + mine) {   // exactly one thread
                unsigned long long acc = before;
                int pick = lo;
                for (int j = hi - 1; j >= lo; --j) {
                    const unsigned long long m = hist[j];
                    if (m > 0ull && acc + m >= target) { pick = j; break; }
                    acc += m;
                }
                sel_digit = pick;
                sel_above = acc;
                sel_incl = acc + hist[pick];
            }
            __syncthreads();
            const int dg = sel_digit;
            if (dg < 0) empty = true;
            else {
                prefix |= static_cast<uint32_t>(dg) << shift;
                above = sel_above;
            }
            __syncthreads();                              // sel_* and hist are rewritten by the next level
        }
        // 16-bit logits: the low 8 key bits were never examined because they are constant -- zeros for x >= 0, ones for
        // x < 0 (the key of a negative float is its complement) -- so the threshold is the logit value itself
        if (p.levels == 2 && !(prefix & 0x80000000u)) prefix |= 0xffu;
        thr = empty ? -INFINITY : key_floor_value(prefix);
        // The nucleus normaliser needs no pass of its own: the last level's scan has summed the masses of exactly the tokens
        // >= thr (2^-40 fixed point relative to L, an integer sum: reproducible, |error| < candidates * 2^-40).
        if (!empty) L64 = static_cast<double>(L) + log2_split(static_cast<float>(sel_incl)) - 40.0;
        ASD_DR_STAMP(8);
    }
    if (t == 0 && p.thr) p.thr[b] = thr;

    // ---- tile masses of the (nucleus-restricted) softmax.  Thread t's j-th vector is v = t + 1024 j = 64 (wave + 16 j) + lane:
    // the lanes of a wave hold tile (wave + 16 j), so wave w owns tiles w, w + 16, ... in the sweeps and in the lists alike.
    const float Lt = static_cast<float>(L64);
    if (listed && thr != -INFINITY) {
        // every candidate's mass replaces its id (the ids are not needed again); then one lane per tile adds its tile's
        // masses in list order.  All of it is wave-local: a wave lists, weighs and sums its own tiles.
        for_cand(wcnt, [&](int slot, float x) {
            if (slot >= 0) cand[wave][slot] = __float_as_uint(x >= thr ? fast_exp2(fmaf(x, p.c2, -Lt)) : 0.0f);
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int tile = wave + kDrWaves * lane; tile < p.n_tiles; tile += kDrWaves * 64) {
            const uint32_t span = tile_span[tile];
            const int first = static_cast<int>(span >> 16), last = first + static_cast<int>(span & 0xffffu);
            float z = 0.0f;
            for (int e = first; e < last; ++e) z += __uint_as_float(cand[wave][e]);
            tile_mass[tile] = z;
        }
    } else if (tiles_from_sweep1) {
        for (int i = t; i < p.n_tiles; i += kDrThreads)       // (written before the barrier inside the first sweep's combine)
            tile_mass[i] *= fast_exp2(__uint_as_float(tile_span[i]) - Lt);
    } else {
        for (int i = t; i < p.n_tiles; i += kDrThreads) tile_mass[i] = 0.0f;
        __syncthreads();
        for_each([&](int v, const u32x4& vec) {
            float x[N];
            unpack<DT>(vec, x);
            if (thr != -INFINITY && !any_at_least(x, thr)) return;      // a tile without a survivor keeps mass 0
            float w[N], pt[N];
            vector_weights<DT>(vec, vec, false, p.c2, Norm2{Lt, 0.0f}, Norm2{0.0f, 0.0f}, thr, -INFINITY, w, pt);
            float z = 0.0f;
#pragma unroll
            for (int i = 0; i < N; ++i) z += pt[i];
            z = wave_sum(z);
            if (lane == 0) tile_mass[v >> 6] = z;
        });
    }
    __syncthreads();
    ASD_DR_STAMP(10);
    if (wave != 0) return;

    // ---- inverse CDF by one wave (the arithmetic of k_rs_pick's bonus draw)
    __shared__ double chunk[64];
    const int per = (p.n_tiles + 63) / 64;
    const int c0 = lane * per, c1 = min(c0 + per, p.n_tiles);
    double mine = 0.0;
    for (int i = c0; i < c1; ++i) mine += static_cast<double>(tile_mass[i]);
    chunk[lane] = mine;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double total = 0.0, before = 0.0;
    for (int l = 0; l < 64; ++l) {
        if (l < lane) before += chunk[l];
        total += chunk[l];
    }
    double target = static_cast<double>(p.r[b]) * total;
    if (!(target >= 0.0)) target = 0.0;
    const bool holds = mine > 0.0 && target >= before && target < before + mine;
    unsigned long long bal = __ballot(holds);
    if (bal == 0) {                                    // rounding pushed the draw past the end: last chunk with mass
        bal = __ballot(mine > 0.0);
        if (bal == 0) {
            if (lane == 0) { p.tok[b] = -1; if (p.lp) p.lp[b] = -INFINITY; }
            return;
        }
        bal = 1ull << (63 - __builtin_clzll(bal));
    }
    if (lane == __builtin_ctzll(bal)) {
        double acc = before;
        int pick = -1, last_pos = -1;
        for (int i = c0; i < c1; ++i) {
            const double m = static_cast<double>(tile_mass[i]);
            if (m > 0.0) {
                last_pos = i;
                if (target < acc + m) { pick = i; break; }
                acc += m;
            }
        }
        if (pick < 0) { pick = last_pos; acc -= static_cast<double>(tile_mass[pick]); }
        pick_tile = pick;
        pick_rest = target - acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int tile = pick_tile;
    const double rest = pick_rest;
    const int v = tile * 64 + lane;
    float w[N], pt[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { w[i] = 0.0f; pt[i] = 0.0f; }
    if (v < p.nvec) { const u32x4 q = row[v]; vector_weights<DT>(q, q, false, p.c2, Norm2{Lt, 0.0f}, Norm2{0.0f, 0.0f}, thr, -INFINITY, w, pt); }
    double lm = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) lm += static_cast<double>(pt[i]);
    __builtin_amdgcn_wave_barrier();
    chunk[lane] = lm;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double lb = 0.0;
    for (int l = 0; l < lane; ++l) lb += chunk[l];
    const bool lholds = lm > 0.0 && rest >= lb && rest < lb + lm;
    unsigned long long lbal = __ballot(lholds);
    if (lbal == 0) {
        lbal = __ballot(lm > 0.0);
        if (lbal == 0) {
            if (lane == 0) { p.tok[b] = -1; if (p.lp) p.lp[b] = -INFINITY; }
            return;
        }
        lbal = 1ull << (63 - __builtin_clzll(lbal));
    }
    if (lane == __builtin_ctzll(lbal)) {
        double acc = lb;
        int pick = -1, last_pos = -1;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double m = static_cast<double>(pt[i]);
            if (m > 0.0 && pick < 0) {
                last_pos = i;
                if (rest < acc + m) pick = i;
                else acc += m;
            }
        }
        if (pick < 0) pick = last_pos;
        p.tok[b] = v * N + pick;
        if (p.lp) {
            const double x = static_cast<double>(E::scalar(row, static_cast<int64_t>(v) * N + pick));
            p.lp[b] = static_cast<float>(kLn2d * (x * static_cast<double>(p.c2) - L64));
        }
    }
    if (lane == 0) ASD_DR_STAMP(11);
}

// ---- asd_residual_sample, many sequences: ONE launch, one 1024-lane workgroup per sequence ------------------------
// The three-launch form above cuts every row over S workgroups: right while B x S about fills the chip (B <= 32: 21-23 us),
// but S falls to 2 at B = 128 and the three launches take 64 us.  From B = 96 on a sequence's two rows stay in ONE
// workgroup (the structure of k_draft_row): sweep 1 = (m2, s) of the target and of the (nucleus-masked) draft row, sweep 2
// (L2 hits) = per-tile residual / target masses into LDS, then the inverse CDF by wave 0 -- the same rs_pick_wave.
template <int DT>
__global__ __launch_bounds__(kDrThreads) void k_residual_row(const RsParams p) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    __shared__ float2 tiles[kDrMaxTiles];
    __shared__ float red[kDrWaves][4];
    __shared__ RsPickScratch sc;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const Rows<DT> rows = select_rows<DT>(p, b);        // block-uniform
    if (!rows.xt) {
        if (t == 0) p.token[b] = -1;
        return;
    }
    const u32x4* vt = reinterpret_cast<const u32x4*>(rows.xt);
    const u32x4* vd = reinterpret_cast<const u32x4*>(rows.xd);
    // thread t's j-th vector is v = t + 1024 j = 64 (wave + 16 j) + lane: a wave step is one 64-vector tile.  Two steps' loads
    // are issued before the first is consumed; the trip count is wave-uniform (a ragged last tile is padded with -inf).
    auto for_each = [&](auto&& fn) {
        constexpr int kAhead = 2;
        const u32x4 neg = {E::kNegInfWord, E::kNegInfWord, E::kNegInfWord, E::kNegInfWord};
        for (int v0 = t - lane; v0 < p.nvec; v0 += kAhead * kDrThreads) {
            u32x4 qt[kAhead], qd[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const int v = v0 + j * kDrThreads + lane;
                qt[j] = v < p.nvec ? vt[v] : neg;
                qd[j] = (vd && v < p.nvec) ? vd[v] : neg;
            }
#pragma unroll
            for (int j = 0; j < kAhead; ++j)
                if (v0 + j * kDrThreads < p.nvec) fn(v0 + j * kDrThreads + lane, qt[j], qd[j]);
        }
    };
    float mt = kSentinel, st = 0.0f, md = kSentinel, sd = 0.0f;
    for_each([&](int, const u32x4& a, const u32x4& d) {
        accum_nucleus<DT>(a, rows.tthr, p.c2, mt, st);
        if (vd) accum_nucleus<DT>(d, rows.dthr, p.c2, md, sd);
    });
    wave_merge(mt, st);
    wave_merge(md, sd);
    if (lane == 0) { red[wave][0] = mt; red[wave][1] = st; red[wave][2] = md; red[wave][3] = sd; }
    __syncthreads();
    mt = red[0][0]; st = red[0][1]; md = red[0][2]; sd = red[0][3];
#pragma unroll
    for (int w = 1; w < kDrWaves; ++w) {
        ms_merge(mt, st, red[w][0], red[w][1]);
        ms_merge(md, sd, red[w][2], red[w][3]);
    }
    const Norm2 Lt = norm2_of(mt, st), Ld = norm2_of(md, sd);
    for_each([&](int v, const u32x4& a, const u32x4& d) {
        float w[N], pt[N];
        vector_weights<DT>(a, d, vd != nullptr, p.c2, Lt, Ld, rows.tthr, rows.dthr, w, pt);
        float z = 0.0f, q = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) { z += w[i]; q += pt[i]; }
        z = wave_sum(z);
        q = wave_sum(q);
        if (lane == 0) tiles[v >> 6] = make_float2(z, q);
    });
    __syncthreads();
    if (wave != 0) return;
    rs_pick_wave<DT>(p, b, lane, rows, tiles, Lt, Ld, sc);
}

// sequences from which the one-workgroup-per-sequence form is used (tools/rs_threshold_ab.sh, V = 152064 bf16: three launches
// 31.6 / 37.8 / 50.1 / 63.9 / 116.4 us at B = 48 / 64 / 96 / 128 / 256, this form 49-55 us at every B)
constexpr int kRsRowMinBatch = 96;

template <int DT>
int launch_rs(const RsParams& p, hipStream_t st) {
    hipLaunchKernelGGL(k_rs_lse<DT>, dim3(p.S, p.B), dim3(kRsThreads), 0, st, p);
    hipLaunchKernelGGL(k_rs_mass<DT>, dim3(p.S, p.B), dim3(kRsThreads), 0, st, p);
    hipLaunchKernelGGL(k_rs_pick<DT>, dim3(p.B), dim3(64), 0, st, p);
    return launch_status();
}

inline int rs_splits(int B, int n_tiles, int cus) {
    int s = (cus + B - 1) / (B > 0 ? B : 1);
    if (s > 32) s = 32;
    if (s > n_tiles) s = n_tiles;
    return s < 1 ? 1 : s;
}

}  // namespace
}  // namespace asd

using namespace asd;

ASD_EXPORT size_t asd_residual_sample_workspace_bytes(int B, int V, int dtype) {
    const int esz = dtype_size(dtype);
    if (B <= 0 || V <= 0 || esz == 0) return 256;
    const size_t nvec = (static_cast<size_t>(V) * esz + 15) / 16;
    const size_t n_tiles = (nvec + 63) / 64;
    return round_up(static_cast<size_t>(B) * 32 * sizeof(float4), 256) + round_up(static_cast<size_t>(B) * n_tiles * sizeof(float2), 256);
}

namespace {
int residual_launch(const void* t_logits, int64_t ld_t, const void* d_logits, int64_t ld_d, const void* bonus_logits,
                    int64_t ld_b, int dtype, const int32_t* n_acc, const float* r, int B, int K, int V,
                    float inv_temperature, const float* d_threshold, int32_t* token, void* workspace,
                    size_t workspace_bytes, void* stream) {
    if (B < 0 || K < 0 || V < 1) return ASD_ERR_INVALID_ARG;
    if (B == 0) return ASD_OK;
    const int esz = dtype_size(dtype);
    if (esz == 0) return ASD_ERR_UNSUPPORTED;
    if (!n_acc || !r || !token || !workspace) return ASD_ERR_INVALID_ARG;
    if (K > 0 && (!t_logits || !d_logits || ld_t < V || ld_d < V)) return ASD_ERR_INVALID_ARG;
    if (bonus_logits && ld_b < V) return ASD_ERR_INVALID_ARG;
    if (!(inv_temperature > 0.0f) || !(inv_temperature < 3.0e38f)) return ASD_ERR_INVALID_ARG;
    // whole 16-byte vectors, 16-byte aligned rows
    if ((static_cast<int64_t>(V) * esz) % 16) return ASD_ERR_ALIGNMENT;
    if ((t_logits && (!aligned_to(t_logits, 16) || (ld_t * esz) % 16)) || (d_logits && (!aligned_to(d_logits, 16) || (ld_d * esz) % 16)) ||
        (bonus_logits && (!aligned_to(bonus_logits, 16) || (ld_b * esz) % 16)))
        return ASD_ERR_ALIGNMENT;
    if (!aligned_to(workspace, 256) || workspace_bytes < asd_residual_sample_workspace_bytes(B, V, dtype)) return ASD_ERR_WORKSPACE;
    RsParams p{};
    p.t_logits = t_logits; p.ld_t = ld_t; p.d_logits = d_logits; p.ld_d = ld_d; p.bonus = bonus_logits; p.ld_b = ld_b;
    p.n_acc = n_acc; p.r = r; p.B = B; p.K = K; p.V = V;
    p.c2 = static_cast<float>(1.4426950408889634074 * static_cast<double>(inv_temperature));
    p.nvec = static_cast<int>(static_cast<int64_t>(V) * esz / 16);
    p.n_tiles = (p.nvec + 63) / 64;
    p.S = rs_splits(B, p.n_tiles, current_device_cus());
    p.partial = static_cast<float4*>(workspace);
    p.tiles = reinterpret_cast<float2*>(static_cast<char*>(workspace) + round_up(static_cast<size_t>(B) * 32 * sizeof(float4), 256));
    p.token = token;
    p.d_thr = d_threshold;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B >= kRsRowMinBatch && p.n_tiles <= kDrMaxTiles) {
        const dim3 grid(static_cast<unsigned>(B)), block(kDrThreads);
        switch (dtype) {
            case ASD_DTYPE_BF16: hipLaunchKernelGGL(k_residual_row<ASD_DTYPE_BF16>, grid, block, 0, st, p); break;
            case ASD_DTYPE_F16: hipLaunchKernelGGL(k_residual_row<ASD_DTYPE_F16>, grid, block, 0, st, p); break;
            default: hipLaunchKernelGGL(k_residual_row<ASD_DTYPE_F32>, grid, block, 0, st, p); break;
        }
        return launch_status();
    }
    switch (dtype) {
        case ASD_DTYPE_BF16: return launch_rs<ASD_DTYPE_BF16>(p, st);
        case ASD_DTYPE_F16: return launch_rs<ASD_DTYPE_F16>(p, st);
        default: return launch_rs<ASD_DTYPE_F32>(p, st);
    }
}

}  // namespace

ASD_EXPORT int asd_residual_sample(const void* t_logits, int64_t ld_t, const void* d_logits, int64_t ld_d,
                                   const void* bonus_logits, int64_t ld_b, int dtype, const int32_t* n_acc,
                                   const float* r, int B, int K, int V, float inv_temperature, int32_t* token,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    return residual_launch(t_logits, ld_t, d_logits, ld_d, bonus_logits, ld_b, dtype, n_acc, r, B, K, V, inv_temperature,
                           nullptr, token, workspace, workspace_bytes, stream);
}

ASD_EXPORT int asd_residual_sample_ex(const void* t_logits, int64_t ld_t, const void* d_logits, int64_t ld_d,
                                      const void* bonus_logits, int64_t ld_b, int dtype, const int32_t* n_acc,
                                      const float* r, int B, int K, int V, float inv_temperature,
                                      const float* d_threshold, int32_t* token, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    return residual_launch(t_logits, ld_t, d_logits, ld_d, bonus_logits, ld_b, dtype, n_acc, r, B, K, V, inv_temperature,
                           d_threshold, token, workspace, workspace_bytes, stream);
}

#ifdef ASD_STAMP
ASD_EXPORT int asd_debug_draft_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(asd::g_dr_stamp), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

ASD_EXPORT size_t asd_draft_sample_workspace_bytes(int B, int V, int dtype) {
    (void)B; (void)V; (void)dtype;
    return 256;   // the one-launch kernel keeps everything in LDS; the argument stays for ABI stability
}

ASD_EXPORT int asd_draft_sample(const void* logits, int64_t ld, int dtype, const float* r, int B, int V,
                                float inv_temperature, float top_p, int32_t* tok, float* lp, float* nucleus_logit,
                                void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    if (B < 0 || V < 1) return ASD_ERR_INVALID_ARG;
    if (B == 0) return ASD_OK;
    const int esz = dtype_size(dtype);
    if (esz == 0) return ASD_ERR_UNSUPPORTED;
    if (!logits || !r || !tok || ld < V) return ASD_ERR_INVALID_ARG;
    if (!(inv_temperature > 0.0f) || !(inv_temperature < 3.0e38f) || top_p != top_p) return ASD_ERR_INVALID_ARG;
    if ((static_cast<int64_t>(V) * esz) % 16 || !aligned_to(logits, 16) || (ld * esz) % 16) return ASD_ERR_ALIGNMENT;
    DrParams p{};
    p.logits = logits; p.ld = ld; p.r = r; p.B = B; p.V = V;
    p.nvec = static_cast<int>(static_cast<int64_t>(V) * esz / 16);
    p.n_tiles = (p.nvec + 63) / 64;
    if (p.n_tiles > kDrMaxTiles) return ASD_ERR_UNSUPPORTED;
    p.c2 = static_cast<float>(1.4426950408889634074 * static_cast<double>(inv_temperature));
    p.top_p = top_p;
    const bool nucleus = top_p > 0.0f && top_p < 1.0f;
    p.levels = nucleus ? (dtype == ASD_DTYPE_F32 ? 3 : 2) : 0;
    p.tok = tok; p.lp = lp; p.thr = nucleus_logit;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(B)), block(kDrThreads);
    switch (dtype) {
        case ASD_DTYPE_BF16: hipLaunchKernelGGL(k_draft_row<ASD_DTYPE_BF16>, grid, block, 0, st, p); break;
        case ASD_DTYPE_F16: hipLaunchKernelGGL(k_draft_row<ASD_DTYPE_F16>, grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL(k_draft_row<ASD_DTYPE_F32>, grid, block, 0, st, p); break;
    }
    return launch_status();
}
