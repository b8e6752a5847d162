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
    const int32_t* n_acc;     // nullptr: every sequence draws from its `bonus` row (asd_draft_sample)
    const float* r;
    int B, K, V, S;
    float c2;
    int nvec, n_tiles;        // 16-byte vectors per row; tiles of 64 vectors
    float4* partial;          // [B][S]  (m2_t, s_t, m2_d, s_d)
    float2* tiles;            // [B][n_tiles]  (Z, P)
    int32_t* token;
    const float* d_thr;       // [B*K] nucleus thresholds of the draft rows (logit < thr => p_d = 0, renormalised) or nullptr
    const float* b_thr;       // [B]   nucleus thresholds of the bonus rows (asd_draft_sample) or nullptr
    float* lp_out;            // [B]   log-probability of the drawn token under the distribution it was drawn from, or nullptr
    float* thr_fill;          // [B]   asd_draft_sample without top-p: receives -inf ("no truncation"), or nullptr
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
    const int j = p.n_acc ? p.n_acc[b] : p.K;
    Rows<DT> r{nullptr, nullptr, -INFINITY, -INFINITY};
    if (j >= 0 && j < p.K) {
        r.xt = static_cast<const char*>(p.t_logits) + (static_cast<int64_t>(b) * p.K + j) * p.ld_t * E::kBytes;
        r.xd = static_cast<const char*>(p.d_logits) + (static_cast<int64_t>(b) * p.K + j) * p.ld_d * E::kBytes;
        if (p.d_thr) r.dthr = p.d_thr[static_cast<int64_t>(b) * p.K + j];
    } else if (p.bonus) {
        r.xt = static_cast<const char*>(p.bonus) + static_cast<int64_t>(b) * p.ld_b * E::kBytes;
        if (p.b_thr) r.tthr = p.b_thr[b];
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
    t0 = static_cast<int>(static_cast<int64_t>(n_tiles) * s / S);
    t1 = static_cast<int>(static_cast<int64_t>(n_tiles) * (s + 1) / S);
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

// L = m2 + log2(s) of a whole row from its S slice partials (fixed order; every caller gets the same bits)
__device__ __forceinline__ void row_norms(const RsParams& p, int b, float& Lt, float& Ld) {
    float mt = kSentinel, st = 0.0f, md = kSentinel, sd = 0.0f;
    for (int s = 0; s < p.S; ++s) {
        const float4 q = p.partial[static_cast<int64_t>(b) * p.S + s];
        ms_merge(mt, st, q.x, q.y);
        ms_merge(md, sd, q.z, q.w);
    }
    Lt = static_cast<float>(static_cast<double>(mt) + log2_split(st));
    Ld = static_cast<float>(static_cast<double>(md) + log2_split(sd));
}

// weights of one 16-byte vector: w_i = max(0, p_t - p_d), and p_t itself; returns the lane's sums
template <int DT>
__device__ __forceinline__ void vector_weights(const u32x4& vt, const u32x4* vdp, float c2, float Lt, float Ld,
                                               float tthr, float dthr,
                                               float (&w)[Elem<DT>::kPerVec], float (&pt)[Elem<DT>::kPerVec]) {
    constexpr int N = Elem<DT>::kPerVec;
    float xt[N], xd[N];
    unpack<DT>(vt, xt);
    if (vdp) unpack<DT>(*vdp, xd);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // outside a row's nucleus the probability is exactly 0 (Lt / Ld are then the nucleus normalisers)
        pt[i] = xt[i] >= tthr ? fast_exp2(fmaf(xt[i], c2, -Lt)) : 0.0f;
        const float pd = (vdp && xd[i] >= dthr) ? fast_exp2(fmaf(xd[i], c2, -Ld)) : 0.0f;
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
    float Lt, Ld;
    row_norms(p, b, Lt, Ld);
    const u32x4* vt = reinterpret_cast<const u32x4*>(rows.xt);
    const u32x4* vd = reinterpret_cast<const u32x4*>(rows.xd);
    for (int t = t0 + wave; t < t1; t += kRsWaves) {
        const int v = t * 64 + lane;
        float z = 0.0f, q = 0.0f;
        if (rows.xt && v < p.nvec) {
            float w[N], pt[N];
            const u32x4 a = vt[v];
            u32x4 d;
            if (vd) d = vd[v];
            vector_weights<DT>(a, vd ? &d : nullptr, p.c2, Lt, Ld, rows.tthr, rows.dthr, w, pt);
#pragma unroll
            for (int i = 0; i < N; ++i) { z += w[i]; q += pt[i]; }
        }
        z = wave_sum(z);
        q = wave_sum(q);
        if (lane == 0) p.tiles[static_cast<int64_t>(b) * p.n_tiles + t] = make_float2(z, q);
    }
}

// ---- pass 3: inverse CDF --------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(64) void k_rs_pick(const RsParams p) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    __shared__ double chunk[64];
    __shared__ double lane_mass[64];
    __shared__ int sel_tile;
    __shared__ double sel_rest;
    const int b = blockIdx.x, lane = threadIdx.x;
    const Rows<DT> rows = select_rows<DT>(p, b);
    if (lane == 0 && p.thr_fill) p.thr_fill[b] = -INFINITY;
    if (!rows.xt) {
        if (lane == 0) {
            p.token[b] = -1;
            if (p.lp_out) p.lp_out[b] = -INFINITY;
        }
        return;
    }
    const float2* tl = p.tiles + static_cast<int64_t>(b) * p.n_tiles;
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
    float Lt, Ld;
    row_norms(p, b, Lt, Ld);
    const u32x4* vt = reinterpret_cast<const u32x4*>(rows.xt);
    const u32x4* vd = reinterpret_cast<const u32x4*>(rows.xd);
    const int v = tile * 64 + lane;
    float w[N], pt[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { w[i] = 0.0f; pt[i] = 0.0f; }
    if (v < p.nvec) {
        const u32x4 a = vt[v];
        u32x4 d;
        if (vd) d = vd[v];
        vector_weights<DT>(a, vd ? &d : nullptr, p.c2, Lt, Ld, rows.tthr, rows.dthr, w, pt);
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
        if (p.lp_out) {
            // log p(token) under the (nucleus-renormalised) softmax of the row the draw came from: x*c2 - L with the
            // row normaliser in f64 -- the same expression finish_row uses on the verify side (bonus / draft draws
            // only: a residual draw has no single-row log-probability, its slot receives log p_t(token))
            float mt = kSentinel, st = 0.0f;
            for (int sl = 0; sl < p.S; ++sl) {
                const float4 q = p.partial[static_cast<int64_t>(b) * p.S + sl];
                ms_merge(mt, st, q.x, q.y);
            }
            const double L = static_cast<double>(mt) + log2_split(st);
            const double x = static_cast<double>(E::scalar(rows.xt, static_cast<int64_t>(v) * N + pick));
            p.lp_out[b] = static_cast<float>(kLn2d * (x * static_cast<double>(p.c2) - L));
        }
    }
}

// ---- asd_draft_sample: nucleus (top-p) threshold by a radix select over probability mass ------------------
// The nucleus of a row is { v : x_v >= x* } with x* the largest logit value whose upper set carries >= top_p of
// the softmax mass.  x* is found digit by digit on the order-preserving 32-bit key of the f32 logit (12 + 12 + 8
// bits; the last digit is constant for 16-bit logits and skipped): per level one streaming pass builds, per row,
// a histogram of probability MASS per digit value (fixed point 2^-40, u64 integer adds: order-independent, so the
// result is bitwise reproducible although the adds are atomic), and one wave scans it from the top for the digit
// where the cumulative mass reaches top_p.  Rows are L2 / Infinity-Cache resident after the first pass.
constexpr int kDsDigits = 4096;                 // histogram slots per row (12-bit digit)
constexpr float kDsFix = 1099511627776.0f;      // 2^40

struct DsState {
    unsigned long long above;    // mass (fixed point) of the digits above the selected prefix
    unsigned long long target;   // top_p * total mass
    uint32_t prefix;             // selected key bits so far
    uint32_t empty;              // 1: the row has no mass at all (all -inf)
};

struct DsParams {
    unsigned long long* hist;    // [B][kDsDigits], zero on entry of every level (the scan re-zeroes it)
    DsState* state;              // [B]
    float* thr;                  // [B] out: x* as a float (-inf: no truncation)
    float top_p;
    int shift, bits;             // this level's digit = (key >> shift) & ((1 << bits) - 1)
    int last;                    // != 0: this level completes the key
};

__device__ __forceinline__ uint32_t order_key(float x) {
    x += 0.0f;                                            // -0 -> +0: equal values share one key
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_floor_value(uint32_t key) {   // smallest float whose key is >= `key`
    const uint32_t u = (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key;
    return __uint_as_float(u);
}

template <int DT>
__global__ __launch_bounds__(kRsThreads) void k_ds_hist(const RsParams p, const DsParams d) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    __shared__ unsigned long long h[kDsDigits];
    const int b = blockIdx.y, s = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int digits = 1 << d.bits;
    for (int i = threadIdx.x; i < digits; i += kRsThreads) h[i] = 0ull;
    __syncthreads();
    const Rows<DT> rows = select_rows<DT>(p, b);
    int t0, t1;
    slice_tiles(p.n_tiles, s, p.S, t0, t1);
    float Lt, Ld;
    row_norms(p, b, Lt, Ld);
    const uint32_t prefix = d.state[b].prefix;
    const int hi_shift = d.shift + d.bits;               // key bits at and above hi_shift are fixed by earlier levels
    const u32x4* vt = reinterpret_cast<const u32x4*>(rows.xt);
    if (rows.xt) {
        for (int t = t0 + wave; t < t1; t += kRsWaves) {
            const int v = t * 64 + lane;
            if (v >= p.nvec) continue;
            float x[N];
            unpack<DT>(vt[v], x);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t key = order_key(x[i]);
                const bool mine = hi_shift >= 32 || (key >> hi_shift) == (prefix >> hi_shift);
                const unsigned long long q = static_cast<unsigned long long>(fast_exp2(fmaf(x[i], p.c2, -Lt)) * kDsFix);
                if (mine && q) atomicAdd(&h[(key >> d.shift) & (digits - 1)], q);
            }
        }
    }
    __syncthreads();
    unsigned long long* out = d.hist + static_cast<int64_t>(b) * kDsDigits;
    for (int i = threadIdx.x; i < digits; i += kRsThreads)
        if (h[i]) atomicAdd(out + i, h[i]);
}

// one workgroup per row: scan the level's histogram from the top digit down.  The histogram is staged in LDS with
// coalesced loads (and zeroed in global memory on the way: ready for the next level / the next call); thread t owns
// the t-th chunk of digits counted from the TOP, an exclusive prefix over the threads (wave scan + one LDS hop) finds
// the chunk in which the cumulative mass reaches the target, and its owner walks the chunk's digits.
constexpr int kScanThreads = 256;
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

__global__ __launch_bounds__(kScanThreads) void k_ds_scan(const DsParams d, int first) {
    __shared__ unsigned long long h[kDsDigits];
    __shared__ unsigned long long wave_tot[kScanThreads / 64];
    __shared__ unsigned long long sel_above;
    __shared__ int sel_digit;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int digits = 1 << d.bits;
    unsigned long long* hist = d.hist + static_cast<int64_t>(b) * kDsDigits;
    for (int i = t; i < digits; i += kScanThreads) {
        h[i] = hist[i];
        hist[i] = 0ull;
    }
    if (t == 0) sel_digit = -1;
    __syncthreads();
    const int per = digits >= kScanThreads ? digits / kScanThreads : 1;      // 16 (12-bit level) or 1 (8-bit level)
    const int hi = digits - t * per, lo = hi - per;                           // thread 0 owns the TOP digits
    unsigned long long mine = 0ull;
    if (lo >= 0)
        for (int j = lo; j < hi; ++j) mine += h[j];
    const unsigned long long incl = wave_incl_scan_u64(mine, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned long long base = 0ull, total = 0ull;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        if (w < wave) base += wave_tot[w];
        total += wave_tot[w];
    }
    DsState st = d.state[b];
    if (first) {
        unsigned long long tg = static_cast<unsigned long long>(static_cast<double>(d.top_p) * static_cast<double>(total));
        if (tg > total) tg = total;
        if (tg == 0ull) tg = 1ull;
        st.above = 0ull;
        st.target = tg;
        st.prefix = 0u;
        st.empty = total == 0ull ? 1u : 0u;
    }
    const unsigned long long before = st.above + base + incl - mine;
    if (mine > 0ull && before < st.target && st.target <= before + mine) {     // exactly one thread
        unsigned long long acc = before;
        int pick = lo;
        for (int j = hi - 1; j >= lo; --j) {
            const unsigned long long m = h[j];
            if (m > 0ull && acc + m >= st.target) { pick = j; break; }
            acc += m;
        }
        sel_digit = pick;
        sel_above = acc;
    }
    __syncthreads();
    if (t == 0) {
        const int dg = sel_digit;
        if (dg < 0) st.empty = 1u;                        // no digit reaches the target: nothing to truncate
        else {
            st.prefix |= static_cast<uint32_t>(dg) << d.shift;
            st.above = sel_above;
        }
        d.state[b] = st;
        if (d.last) d.thr[b] = st.empty ? -INFINITY : key_floor_value(st.prefix);
    }
}

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
    switch (dtype) {
        case ASD_DTYPE_BF16: return launch_rs<ASD_DTYPE_BF16>(p, st);
        case ASD_DTYPE_F16: return launch_rs<ASD_DTYPE_F16>(p, st);
        default: return launch_rs<ASD_DTYPE_F32>(p, st);
    }
}

size_t ds_sample_bytes(int B, int V, int dtype) { return asd_residual_sample_workspace_bytes(B, V, dtype); }
size_t ds_hist_bytes(int B) { return round_up(static_cast<size_t>(B) * kDsDigits * sizeof(unsigned long long), 256); }
size_t ds_state_bytes(int B) { return round_up(static_cast<size_t>(B) * sizeof(DsState), 256); }

template <int DT>
int launch_ds(RsParams p, DsParams d, bool nucleus, bool wide_key, hipStream_t st) {
    const dim3 grid(p.S, p.B), block(kRsThreads);
    if (nucleus) {
        // normaliser of the whole row, then one (hist, scan) pair per key digit
        hipLaunchKernelGGL(k_rs_lse<DT>, grid, block, 0, st, p);
        const int shifts[3] = {20, 8, 0}, widths[3] = {12, 12, 8};
        const int levels = wide_key ? 3 : 2;
        for (int lv = 0; lv < levels; ++lv) {
            d.shift = shifts[lv];
            d.bits = widths[lv];
            d.last = lv == levels - 1;
            hipLaunchKernelGGL(k_ds_hist<DT>, grid, block, 0, st, p, d);
            hipLaunchKernelGGL(k_ds_scan, dim3(p.B), dim3(kScanThreads), 0, st, d, lv == 0 ? 1 : 0);
        }
        p.b_thr = d.thr;
    }
    hipLaunchKernelGGL(k_rs_lse<DT>, grid, block, 0, st, p);     // nucleus normaliser (or the plain one)
    hipLaunchKernelGGL(k_rs_mass<DT>, grid, block, 0, st, p);
    hipLaunchKernelGGL(k_rs_pick<DT>, dim3(p.B), dim3(64), 0, st, p);
    return launch_status();
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

ASD_EXPORT size_t asd_draft_sample_workspace_bytes(int B, int V, int dtype) {
    if (B <= 0 || V <= 0 || dtype_size(dtype) == 0) return 256;
    return ds_sample_bytes(B, V, dtype) + ds_hist_bytes(B) + ds_state_bytes(B) + round_up(static_cast<size_t>(B) * sizeof(float), 256);
}

ASD_EXPORT int asd_draft_sample(const void* logits, int64_t ld, int dtype, const float* r, int B, int V,
                                float inv_temperature, float top_p, int32_t* tok, float* lp, float* nucleus_logit,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || V < 1) return ASD_ERR_INVALID_ARG;
    if (B == 0) return ASD_OK;
    const int esz = dtype_size(dtype);
    if (esz == 0) return ASD_ERR_UNSUPPORTED;
    if (!logits || !r || !tok || !workspace || ld < V) return ASD_ERR_INVALID_ARG;
    if (!(inv_temperature > 0.0f) || !(inv_temperature < 3.0e38f) || top_p != top_p) return ASD_ERR_INVALID_ARG;
    if ((static_cast<int64_t>(V) * esz) % 16 || !aligned_to(logits, 16) || (ld * esz) % 16) return ASD_ERR_ALIGNMENT;
    if (!aligned_to(workspace, 256) || workspace_bytes < asd_draft_sample_workspace_bytes(B, V, dtype)) return ASD_ERR_WORKSPACE;
    const bool nucleus = top_p > 0.0f && top_p < 1.0f;
    char* ws = static_cast<char*>(workspace);
    RsParams p{};
    p.bonus = logits; p.ld_b = ld; p.n_acc = nullptr; p.r = r; p.B = B; p.K = 0; p.V = V;
    p.c2 = static_cast<float>(1.4426950408889634074 * static_cast<double>(inv_temperature));
    p.nvec = static_cast<int>(static_cast<int64_t>(V) * esz / 16);
    p.n_tiles = (p.nvec + 63) / 64;
    p.S = rs_splits(B, p.n_tiles, current_device_cus());
    p.partial = reinterpret_cast<float4*>(ws);
    p.tiles = reinterpret_cast<float2*>(ws + round_up(static_cast<size_t>(B) * 32 * sizeof(float4), 256));
    p.token = tok;
    p.lp_out = lp;
    DsParams d{};
    char* extra = ws + ds_sample_bytes(B, V, dtype);
    d.hist = reinterpret_cast<unsigned long long*>(extra);
    d.state = reinterpret_cast<DsState*>(extra + ds_hist_bytes(B));
    d.thr = nucleus_logit ? nucleus_logit : reinterpret_cast<float*>(extra + ds_hist_bytes(B) + ds_state_bytes(B));
    d.top_p = top_p;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (nucleus) {
        // histogram + state start from zero; the scans leave the histogram zeroed again
        if (hipMemsetAsync(extra, 0, ds_hist_bytes(B) + ds_state_bytes(B), st) != hipSuccess) return ASD_ERR_HIP;
    } else {
        p.thr_fill = nucleus_logit;
    }
    int rc;
    switch (dtype) {
        case ASD_DTYPE_BF16: rc = launch_ds<ASD_DTYPE_BF16>(p, d, nucleus, false, st); break;
        case ASD_DTYPE_F16: rc = launch_ds<ASD_DTYPE_F16>(p, d, nucleus, false, st); break;
        default: rc = launch_ds<ASD_DTYPE_F32>(p, d, nucleus, true, st); break;
    }
    return rc;
}
