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

#include "sample_device.hpp"

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
// tl: the row's per-tile masses (Z, P) -- in the workspace (k_rs_pick) or in LDS (k_residual_row, k_residual_group); Lt / Ld: the
// rows' normalisers.  Prefixes are f64 wave SCANS over f32 masses (exact unless a sum spans more than 53 bits).  Round 2 walked
// 64-entry LDS arrays in 64-step loops (rs_pick_wave): inlined into a 1024-lane kernel that alone asked for more than the 128-VGPR
// budget (k_residual_row spilled 24 registers) and cost ~3 us of LDS round trips on the tail.
struct RsScanScratch {
    int tile;
    double rest;
};
template <int DT>
__device__ __forceinline__ void rs_pick_scan(const RsParams& p, int b, int lane, const Rows<DT>& rows, const float2* tl, Norm2 Lt,
                                             Norm2 Ld, RsScanScratch& sc) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    const int per = (p.n_tiles + 63) / 64;
    const int c0 = lane * per, c1 = min(c0 + per, p.n_tiles);
    double cz = 0.0, cp = 0.0;
    for (int t = c0; t < c1; ++t) { cz += static_cast<double>(tl[t].x); cp += static_cast<double>(tl[t].y); }
    const double totz = __shfl(wave_incl_scan_f64(cz, lane), 63, 64);
    const bool use_p = !(totz > 0.0);                  // empty residual (p_t <= p_d everywhere): draw from p_t
    const double mine = use_p ? cp : cz;
    const double incl = wave_incl_scan_f64(mine, lane);
    const double total = __shfl(incl, 63, 64);
    double before = __shfl_up(incl, 1, 64);
    if (lane == 0) before = 0.0;
    double target = static_cast<double>(p.r[b]) * total;
    if (!(target >= 0.0)) target = 0.0;
    const bool holds = mine > 0.0 && target >= before && target < before + mine;
    unsigned long long bal = __ballot(holds);
    if (bal == 0) {                                    // rounding pushed the draw past the end: last chunk with mass
        bal = __ballot(mine > 0.0);
        if (bal == 0) { if (lane == 0) p.token[b] = -1; return; }
        bal = 1ull << (63 - __builtin_clzll(bal));
    }
    if (lane == __builtin_ctzll(bal)) {
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
        sc.tile = pick;
        sc.rest = target - acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int tile = sc.tile;
    const double rest = sc.rest;
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
    const double lincl = wave_incl_scan_f64(lm, lane);
    double lb = __shfl_up(lincl, 1, 64);
    if (lane == 0) lb = 0.0;
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
    __shared__ RsScanScratch sc;
    const int b = blockIdx.x, lane = threadIdx.x;
    const Rows<DT> rows = select_rows<DT>(p, b);
    if (!rows.xt) {
        if (lane == 0) p.token[b] = -1;
        return;
    }
    Norm2 Lt, Ld;
    row_norms(p, b, Lt, Ld);
    rs_pick_scan<DT>(p, b, lane, rows, p.tiles + static_cast<int64_t>(b) * p.n_tiles, Lt, Ld, sc);
}

// ---- asd_residual_sample, many sequences: ONE launch, one 1024-lane workgroup per sequence ------------------------
// The three-launch form above cuts every row over S workgroups: right while B x S about fills the chip (B <= 32: 21-23 us),
// but S falls to 2 at B = 128 and the three launches take 64 us.  From B = 96 on a sequence's two rows stay in ONE
// workgroup (the structure of k_draft_row): sweep 1 = (m2, s) of the target and of the (nucleus-masked) draft row, sweep 2
// (L2 hits) = per-tile residual / target masses into LDS, then the inverse CDF by wave 0 -- the same rs_pick_scan.
template <int DT>
__global__ __launch_bounds__(kDrThreads) void k_residual_row(const RsParams p) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    __shared__ float2 tiles[kDrMaxTiles];
    __shared__ float red[kDrWaves][4];
    __shared__ RsScanScratch sc;
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
    rs_pick_scan<DT>(p, b, lane, rows, tiles, Lt, Ld, sc);
}

// ---- asd_residual_sample for FEW sequences (round 3): G workgroups per sequence inside ONE launch -----------------------------
// The structure of k_draft_group (draft_sample.hip): the sequence's target row and draft row are cut into G contiguous runs of
// 64-vector tiles, one 1024-lane workgroup each, resident in registers (read from HBM once; the three-launch form reads them
// twice and pays two launch gaps: 24 us at B = 32).  Per tile the CANONICAL pairs (M, s) of both rows -> the leader (workgroup 0)
// folds them in a fixed order -> both normalisers back to the partners -> per tile the residual mass Z and the target mass P ->
// the leader runs the same rs_pick_scan.  Mailboxes: single writer, single reader, self-tagging, handed back empty (workspace
// all-zero between calls); a lost word poisons the sequence (token = -1).
struct RgParams {
    RsParams r;
    int G, n_pad;
    unsigned long long* small;       // per sequence: pairs_t[n_pad], pairs_d[n_pad], mass[n_pad], mail[kDgMaxGroups][2]
    uint32_t* status;                // the workspace's sticky status word
};

template <int DT, int TPW>
__global__ __launch_bounds__(kDrThreads) void k_residual_group(const RgParams q) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    __shared__ float tm[kDrMaxTiles], ts[kDrMaxTiles], dm[kDrMaxTiles], dsum[kDrMaxTiles];
    __shared__ float2 tiles[kDrMaxTiles];
    __shared__ float red[kDrWaves][2];
    __shared__ unsigned long long bc[2];
    __shared__ volatile int lost;
    __shared__ RsScanScratch sc;
    const RsParams& p = q.r;
    const int g = blockIdx.x, b = blockIdx.y, G = q.G;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool leader = g == 0;
    const Rows<DT> rows = select_rows<DT>(p, b);            // the same for every workgroup of the sequence
    if (!rows.xt) {
        if (leader && t == 0) p.token[b] = -1;
        return;
    }
    unsigned long long* small = q.small + static_cast<int64_t>(b) * (3 * q.n_pad + kDgMaxGroups * 2);
    unsigned long long* pt_x = small;
    unsigned long long* pd_x = small + q.n_pad;
    unsigned long long* mass_x = small + 2 * q.n_pad;
    unsigned long long* mail = small + 3 * q.n_pad + static_cast<int64_t>(g) * 2;
    if (t == 0) lost = 0;
    __syncthreads();
    const u32x4* vt = reinterpret_cast<const u32x4*>(rows.xt);
    const u32x4* vd = reinterpret_cast<const u32x4*>(rows.xd);
    const bool has_d = vd != nullptr;
    int t0, t1;
    {
        const uint32_t nt = static_cast<uint32_t>(p.n_tiles);
        t0 = static_cast<int>(nt * static_cast<uint32_t>(g) / static_cast<uint32_t>(G));
        t1 = static_cast<int>(nt * static_cast<uint32_t>(g + 1) / static_cast<uint32_t>(G));
    }
    const u32x4 neg = {E::kNegInfWord, E::kNegInfWord, E::kNegInfWord, E::kNegInfWord};
    u32x4 qa[TPW], qb[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tile = t0 + wave + kDrWaves * j;
        const int v = tile * 64 + lane;
        const bool in = tile < t1 && v < p.nvec;
        qa[j] = in ? vt[v] : neg;
        qb[j] = (in && has_d) ? vd[v] : neg;
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tile = t0 + wave + kDrWaves * j;
        if (tile < t1) {                                   // wave-uniform
            float Mt, St, Md = kSentinel, Sd = 0.0f;
            tile_pair<DT>(qa[j], p.c2, Mt, St, rows.tthr);
            if (has_d) tile_pair<DT>(qb[j], p.c2, Md, Sd, rows.dthr);
            if (lane == 0) {
                if (leader) { tm[tile] = Mt; ts[tile] = St; dm[tile] = Md; dsum[tile] = Sd; }
                else {
                    dg_put(pt_x + tile, (static_cast<unsigned long long>(__float_as_uint(St)) << 32) | __float_as_uint(Mt));
                    dg_put(pd_x + tile, (static_cast<unsigned long long>(__float_as_uint(Sd)) << 32) | __float_as_uint(Md));
                }
            }
        }
    }
    float mt, st, md, sd;
    if (leader) {
        for (int tile = t1 + t; tile < p.n_tiles; tile += kDrThreads) {
            const unsigned long long a = dg_poll(pt_x + tile, &lost, q.status);
            const unsigned long long c = dg_poll(pd_x + tile, &lost, q.status);
            dg_put(pt_x + tile, 0ull);
            dg_put(pd_x + tile, 0ull);
            tm[tile] = __uint_as_float(static_cast<uint32_t>(a));
            ts[tile] = __uint_as_float(static_cast<uint32_t>(a >> 32));
            dm[tile] = __uint_as_float(static_cast<uint32_t>(c));
            dsum[tile] = __uint_as_float(static_cast<uint32_t>(c >> 32));
        }
        __syncthreads();
        fold_tile_pairs(tm, ts, p.n_tiles, red, wave, lane, mt, st);
        __syncthreads();                                   // `red` is reused
        fold_tile_pairs(dm, dsum, p.n_tiles, red, wave, lane, md, sd);
        if (t >= 1 && t < G) {                             // both normalisers to every partner (never all-zero: s >= 1 or m2 = sentinel)
            unsigned long long* box = small + 3 * q.n_pad + static_cast<int64_t>(t) * 2;
            dg_put(box, (static_cast<unsigned long long>(__float_as_uint(st)) << 32) | __float_as_uint(mt));
            dg_put(box + 1, (static_cast<unsigned long long>(__float_as_uint(sd)) << 32) | __float_as_uint(md));
        }
    } else {
        if (t < 2) {
            bc[t] = dg_poll(mail + t, &lost, q.status);
            dg_put(mail + t, 0ull);
        }
        __syncthreads();
        if (lost) return;
        mt = __uint_as_float(static_cast<uint32_t>(bc[0]));
        st = __uint_as_float(static_cast<uint32_t>(bc[0] >> 32));
        md = __uint_as_float(static_cast<uint32_t>(bc[1]));
        sd = __uint_as_float(static_cast<uint32_t>(bc[1] >> 32));
    }
    const Norm2 Lt = norm2_of(mt, st), Ld = norm2_of(md, sd);
    // ---- per tile the residual mass Z and the target mass P (the arithmetic of k_rs_mass)
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        asm volatile("" : "+v"(qa[j]), "+v"(qb[j]));
        const int tile = t0 + wave + kDrWaves * j;
        if (tile < t1) {                                   // wave-uniform
            float w[N], pt[N];
            vector_weights<DT>(qa[j], qb[j], has_d, p.c2, Lt, Ld, rows.tthr, rows.dthr, w, pt);
            float z = 0.0f, pm = 0.0f;
#pragma unroll
            for (int i = 0; i < N; ++i) { z += w[i]; pm += pt[i]; }
            z = wave_sum(z);
            pm = wave_sum(pm);
            if (lane == 0) {
                if (leader) tiles[tile] = make_float2(z, pm);
                else dg_put(mass_x + tile, kDgValid | (static_cast<unsigned long long>(__float_as_uint(pm)) << 32) | __float_as_uint(z));   // P >= 0: its sign bit is the tag
            }
        }
    }
    if (!leader) return;
    for (int tile = t1 + t; tile < p.n_tiles; tile += kDrThreads) {
        const unsigned long long v = dg_poll(mass_x + tile, &lost, q.status);
        dg_put(mass_x + tile, 0ull);
        tiles[tile] = make_float2(__uint_as_float(static_cast<uint32_t>(v)), __uint_as_float(static_cast<uint32_t>((v & ~kDgValid) >> 32)));
    }
    __syncthreads();
    if (wave != 0) return;
    if (lost) {
        if (lane == 0) p.token[b] = -1;                    // a hand-off never arrived: poisoned, not guessed
        return;
    }
    rs_pick_scan<DT>(p, b, lane, rows, tiles, Lt, Ld, sc);
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

namespace {
#ifdef ASD_TEST_HOOKS          // (process-global, not thread-safe: the TEST build of the library only)
int g_debug_rs_groups = 0;     // asd_debug_residual_groups (tests only): force the workgroups per sequence; 0 = heuristic, -1 = never the group form
#else
constexpr int g_debug_rs_groups = 0;
#endif
struct RgLayout {
    size_t legacy_bytes, small_bytes;
    int n_pad;
};
RgLayout rg_layout(int B, size_t n_tiles) {
    RgLayout l;
    l.legacy_bytes = round_up(static_cast<size_t>(B) * 32 * sizeof(float4), 256) + round_up(static_cast<size_t>(B) * n_tiles * sizeof(float2), 256);
    l.n_pad = static_cast<int>(round_up(n_tiles, 32));
    l.small_bytes = static_cast<size_t>(B) * (3 * static_cast<size_t>(l.n_pad) + kDgMaxGroups * 2) * sizeof(unsigned long long);
    return l;
}
}  // namespace

ASD_EXPORT size_t asd_residual_sample_workspace_bytes(int B, int V, int dtype) {
    const int esz = dtype_size(dtype);
    if (B <= 0 || V <= 0 || esz == 0) return 256;
    const size_t nvec = (static_cast<size_t>(V) * esz + 15) / 16;
    const RgLayout l = rg_layout(B, (nvec + 63) / 64);
    return kWorkspaceHeaderBytes + l.legacy_bytes + round_up(l.small_bytes, 256);
}

#ifdef ASD_TEST_HOOKS
/* tests only: force the workgroups a sequence's rows are spread over (1 ... 32; -1 = the multi-launch / one-workgroup forms;
 * 0 = heuristic). */
ASD_EXPORT int asd_debug_residual_groups(int groups) {
    g_debug_rs_groups = groups;
    return ASD_OK;
}
#endif

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
    char* const body = static_cast<char*>(workspace) + kWorkspaceHeaderBytes;       // (behind the status block)
    p.partial = reinterpret_cast<float4*>(body);
    p.tiles = reinterpret_cast<float2*>(body + round_up(static_cast<size_t>(B) * 32 * sizeof(float4), 256));
    p.token = token;
    p.d_thr = d_threshold;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // few sequences: G workgroups per sequence, rows resident in registers, one launch (the mailboxes live behind the legacy
    // part of the workspace and must have been zeroed once: asd_workspace_init)
    {
        const RgLayout l = rg_layout(B, static_cast<size_t>(p.n_tiles));
        const bool have_ws = workspace_bytes >= kWorkspaceHeaderBytes + l.legacy_bytes + l.small_bytes && p.n_tiles <= kDrMaxTiles;
        int G = 0;
        if (have_ws && g_debug_rs_groups >= 0 && B <= 128) {         // (B <= 128: two workgroups per sequence still fit the CUs)
            const int cus = current_device_cus();
            G = cus / B;                          // any count, not only powers of two (round 4: B = 33 ... 63 left half the CUs idle)
            if (G > kDgMaxGroups) G = kDgMaxGroups;
            while (G > 1 && (static_cast<int64_t>(B) * G > 256 || p.n_tiles / G < 8)) --G;
            if (G < 1) G = 1;
            if (g_debug_rs_groups > 0 && g_debug_rs_groups <= kDgMaxGroups && static_cast<int64_t>(B) * g_debug_rs_groups <= 256) G = g_debug_rs_groups;
            const int tpw = ((p.n_tiles + G - 1) / G + kDrWaves - 1) / kDrWaves;
            if (tpw > 10) G = 0;
            if (G >= 1) {
                RgParams q{};
                q.r = p;
                q.G = G;
                q.n_pad = l.n_pad;
                q.small = reinterpret_cast<unsigned long long*>(body + l.legacy_bytes);
                q.status = static_cast<uint32_t*>(workspace);
                const dim3 grid(static_cast<unsigned>(G), static_cast<unsigned>(B)), block(kDrThreads);
#define ASD_LAUNCH_RG(DT)                                                                                 \
    do {                                                                                                  \
        if (tpw <= 3) hipLaunchKernelGGL((k_residual_group<DT, 3>), grid, block, 0, st, q);               \
        else if (tpw <= 5) hipLaunchKernelGGL((k_residual_group<DT, 5>), grid, block, 0, st, q);          \
        else if (tpw <= 7) hipLaunchKernelGGL((k_residual_group<DT, 7>), grid, block, 0, st, q);          \
        else hipLaunchKernelGGL((k_residual_group<DT, 10>), grid, block, 0, st, q);                       \
    } while (0)
                switch (dtype) {
                    case ASD_DTYPE_BF16: ASD_LAUNCH_RG(ASD_DTYPE_BF16); break;
                    case ASD_DTYPE_F16: ASD_LAUNCH_RG(ASD_DTYPE_F16); break;
                    default: ASD_LAUNCH_RG(ASD_DTYPE_F32); break;
                }
#undef ASD_LAUNCH_RG
                return launch_status();
            }
        }
    }
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
