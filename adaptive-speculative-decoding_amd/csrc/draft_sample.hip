// draft_sample.hip -- X1: the draft tier's per-step token proposal (temperature, top-p nucleus, inverse-CDF draw, log q(tok)), gfx950.
// The reference delegates this step to HF generate(do_sample=True, temperature=0.7, top_p=0.9)
// (src/training/generate_training_data.py:110-119); specified in include/asd_hip.h and DESIGN.md.

#include "sample_device.hpp"

namespace asd {
namespace {

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


// ---- inverse CDF by ONE wave: prefix over the tile masses -> tile of the draw -> that tile's 64 vectors (re-read: 1 KB)
// -> lane -> element, with the same float weights as the tile masses; log q(tok) in f64.  The prefixes are wave scans in f64
// over f32 masses (exact unless a sum spans more than 53 bits), not 64-step loops over LDS (3.3 us of the round-2 kernel).
struct DraftPickScratch {
    int tile;
    double rest;
};
template <int DT>
__device__ __forceinline__ void draft_pick_wave(const u32x4* row, int nvec, const float* tile_mass, int n_tiles, float r, float c2,
                                                float Lt, double L64, float thr, int lane, DraftPickScratch& sc, int32_t* tok,
                                                float* lp) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    const int per = (n_tiles + 63) / 64;
    const int c0 = lane * per, c1 = min(c0 + per, n_tiles);
    double mine = 0.0;
    for (int i = c0; i < c1; ++i) mine += static_cast<double>(tile_mass[i]);
    const double incl = wave_incl_scan_f64(mine, lane);
    const double total = __shfl(incl, 63, 64);
    double before = __shfl_up(incl, 1, 64);
    if (lane == 0) before = 0.0;
    double target = static_cast<double>(r) * total;
    if (!(target >= 0.0)) target = 0.0;
    const bool holds = mine > 0.0 && target >= before && target < before + mine;
    unsigned long long bal = __ballot(holds);
    if (bal == 0) {                                    // rounding pushed the draw past the end: last chunk with mass
        bal = __ballot(mine > 0.0);
        if (bal == 0) {
            if (lane == 0) { *tok = -1; if (lp) *lp = -INFINITY; }
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
        sc.tile = pick;
        sc.rest = target - acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int tile = sc.tile;
    const double rest = sc.rest;
    const int v = tile * 64 + lane;
    float w[N], pt[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { w[i] = 0.0f; pt[i] = 0.0f; }
    if (v < nvec) { const u32x4 q = row[v]; vector_weights<DT>(q, q, false, c2, Norm2{Lt, 0.0f}, Norm2{0.0f, 0.0f}, thr, -INFINITY, w, pt); }
    double lm = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) lm += static_cast<double>(pt[i]);
    const double lincl = wave_incl_scan_f64(lm, lane);
    double lb = __shfl_up(lincl, 1, 64);
    if (lane == 0) lb = 0.0;
    const bool lholds = lm > 0.0 && rest >= lb && rest < lb + lm;
    unsigned long long lbal = __ballot(lholds);
    if (lbal == 0) {
        lbal = __ballot(lm > 0.0);
        if (lbal == 0) {
            if (lane == 0) { *tok = -1; if (lp) *lp = -INFINITY; }
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
        *tok = v * N + pick;
        if (lp) {
            const double x = static_cast<double>(E::scalar(row, static_cast<int64_t>(v) * N + pick));
            *lp = static_cast<float>(kLn2d * (x * static_cast<double>(c2) - L64));
        }
    }
}

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
    float m2, s;
    ASD_DR_STAMP(0);
    // Sweep 1 (the only pass that reads HBM) leaves every tile's CANONICAL (max, sum) pair in LDS and folds the pairs in the
    // fixed order of fold_tile_pairs: L has the same bits as in k_draft_group, whatever the batch (round 2 ran a per-lane
    // online softmax here when top-p was on: 5 us less for the sweep, but a value only this geometry could reproduce).
    // Without truncation the tile masses then follow from L without a second exp-per-element sweep of the row.
    const bool tiles_from_sweep1 = p.levels == 0;
    for_each([&](int v, const u32x4& vec) {
        float M, sw;
        tile_pair<DT>(vec, p.c2, M, sw);
        if (lane == 0) {                                      // nothing is carried from tile to tile: the four tiles of a
            tile_span[v >> 6] = __float_as_uint(M);           // for_each step reduce side by side
            tile_mass[v >> 6] = sw;
        }
    });
    __syncthreads();
    fold_tile_pairs(reinterpret_cast<const float*>(tile_span), tile_mass, p.n_tiles, red, wave, lane, m2, s);
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

    // ---- inverse CDF by one wave
    __shared__ DraftPickScratch pick;
    draft_pick_wave<DT>(row, p.nvec, tile_mass, p.n_tiles, p.r[b], p.c2, Lt, L64, thr, lane, pick, p.tok + b, p.lp ? p.lp + b : nullptr);
    if (lane == 0) ASD_DR_STAMP(11);
}

// =====================================================================================================================
// asd_draft_sample for FEW rows (round 3): G workgroups per row inside ONE launch, the row resident in registers.
//
// One 1024-lane workgroup per row leaves 256 - B compute units idle and makes every sweep of the 304 KB row a single CU's
// issue problem (B = 32: 43.8 us with top-p, 2.8 % of the HBM roof).  Here a row is cut into G contiguous runs of 64-vector
// tiles, one workgroup each (B * G ~ the number of CUs); a wave holds its <= TPW tiles in registers for the whole kernel,
// so the row is read from HBM exactly once and no phase re-sweeps memory.  What crosses workgroups is small and goes
// through SINGLE-WRITER / SINGLE-READER mailbox words in the workspace, self-tagging as in k_verify (a published word is
// never zero; the reader polls it, bounded, and hands it back empty): no epoch, no memset node, hipGraph-replay safe, and
// the workspace is all-zero between calls.  Workgroup 0 of a row is its LEADER:
//
//   every workgroup   per tile the pair (M, s) with  sum 2^(x c2) = s 2^M  (wave_max / wave_sum: a CANONICAL value of the
//                     tile, whoever computes it)                                              -> pairs[tile]      (-> leader)
//   leader            folds the n_tiles pairs in a fixed order -> (m2, s) of the row          -> mail[g][0]       (-> partners)
//   no truncation     partners are done after publishing their pairs; the leader scales the pairs into tile masses, draws
//   top-p             every workgroup: histogram of its candidates' probability MASS (2^-40 fixed point: integer adds
//                     commute => reproducible) over the bins of the current key range (<= 4096)  -> hist[g][bin]   (-> leader)
//                     leader: sum over workgroups, scan from the top for the bin where the cumulative mass reaches
//                     top_p * total                                                            -> mail[g][1 + level]
//                     (the range is narrowed and the step repeated while a bin still spans several keys: never for bf16
//                     rows whose candidates span < 32 binades, once for f16, twice for f32)
//                     every workgroup: nucleus-restricted tile masses                          -> mass[tile]       (-> leader)
//   leader            prefix over the tile masses -> tile of the draw -> re-reads that ONE tile (1 KB) -> lane -> element
//
// Because the pairs are canonical and the fold order is fixed, L -- and with it the nucleus threshold (an integer decision),
// log q(tok) and the token -- do not depend on G: the same row gives the same bits in a batch of 8 and in a batch of 200
// (k_draft_row, the B > 128 form, uses the same pairs, fold and draw).
struct DgParams {
    DrParams d;
    int G;                    // workgroups per row
    unsigned long long* hist_x;   // [rows * G][kDgMaxLevels][kDsDigits]   histogram exchange (partner -> leader), one area per round
    unsigned long long* small;    // per row: pairs[n_pad], mass[n_pad], mail[kDgMaxGroups][kDgMsgs][2]
    int n_pad;                // n_tiles rounded up to a whole 256-byte block of words
    int base_shift;           // key bits that carry no information for this dtype (bf16: 16, f16: 13, f32: 0)
    uint32_t* status;         // the workspace's sticky status word (a wait that runs out or-s ASD_WS_LOST_HANDOFF into it)
    int withhold1;            // test build (asd_debug_draft_withhold): 1 + (row * G + g) of the partner workgroup that never publishes its tile pairs; 0 = off
};

template <int DT, int TPW>
__global__ __launch_bounds__(kDrThreads) void k_draft_group(const DgParams p) {
    using E = Elem<DT>;
    constexpr int N = E::kPerVec;
    __shared__ union {
        struct { float m[kDrMaxTiles], s[kDrMaxTiles]; } pair;      // leader: the row's tile pairs (before the rounds)
        unsigned long long hist[kDsDigits];                          // every workgroup: the round's histogram
    } u;
    __shared__ float tile_mass[kDrMaxTiles];                         // leader
    __shared__ float red[kDrWaves][2];
    __shared__ unsigned long long wave_tot[kDrWaves];
    __shared__ unsigned long long sel_above, sel_hist, bc[2];
    __shared__ int sel_digit;
    __shared__ volatile int lost;
    __shared__ DraftPickScratch pick;
    const DrParams& d = p.d;
    const int g = blockIdx.x, b = blockIdx.y, G = p.G;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool leader = g == 0;
    const u32x4* row = reinterpret_cast<const u32x4*>(static_cast<const char*>(d.logits) + static_cast<int64_t>(b) * d.ld * E::kBytes);
    unsigned long long* small = p.small + static_cast<int64_t>(b) * (2 * p.n_pad + kDgMaxGroups * kDgMsgs * 2);
    unsigned long long* pairs_x = small;
    unsigned long long* mass_x = small + p.n_pad;
    unsigned long long* mail = small + 2 * p.n_pad + static_cast<int64_t>(g) * kDgMsgs * 2;          // this workgroup's inbox
    unsigned long long* hist_mine = p.hist_x + (static_cast<int64_t>(b) * G + g) * kDgMaxLevels * kDsDigits;
    if (t == 0) lost = 0;
    __syncthreads();

    // ---- this workgroup's tiles, loaded once (the only HBM read of the row) and kept in registers
    int t0, t1;
    {
        const uint32_t nt = static_cast<uint32_t>(d.n_tiles);
        t0 = static_cast<int>(nt * static_cast<uint32_t>(g) / static_cast<uint32_t>(G));
        t1 = static_cast<int>(nt * static_cast<uint32_t>(g + 1) / static_cast<uint32_t>(G));
    }
    const u32x4 neg = {E::kNegInfWord, E::kNegInfWord, E::kNegInfWord, E::kNegInfWord};
    u32x4 q[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tile = t0 + wave + kDrWaves * j;
        const int v = tile * 64 + lane;
        q[j] = (tile < t1 && v < d.nvec) ? row[v] : neg;
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tile = t0 + wave + kDrWaves * j;
        if (tile < t1) {                                   // wave-uniform
            float M, sw;
            tile_pair<DT>(q[j], d.c2, M, sw);
            if (lane == 0) {
                if (leader) { u.pair.m[tile] = M; u.pair.s[tile] = sw; }
                else if (b * G + g + 1 != p.withhold1)
                    dg_put(pairs_x + tile, (static_cast<unsigned long long>(__float_as_uint(sw)) << 32) | __float_as_uint(M));
            }
        }
    }
    if (!leader && d.levels == 0) return;                  // no truncation: the leader does the rest from the pairs

    float m2 = kSentinel, s = 0.0f;
    if (leader) {
        for (int tile = t1 + t; tile < d.n_tiles; tile += kDrThreads) {     // the partners' tiles (the leader's run is [0, t1))
            const unsigned long long v = dg_poll(pairs_x + tile, &lost, p.status);
            dg_put(pairs_x + tile, 0ull);
            u.pair.m[tile] = __uint_as_float(static_cast<uint32_t>(v));
            u.pair.s[tile] = __uint_as_float(static_cast<uint32_t>(v >> 32));
        }
        __syncthreads();
        fold_tile_pairs(u.pair.m, u.pair.s, d.n_tiles, red, wave, lane, m2, s);
        if (d.levels > 0 && t >= 1 && t < G)               // (m2, s) to every partner: s >= 1, or m2 is the sentinel -- never all-zero bits
            dg_put(small + 2 * p.n_pad + static_cast<int64_t>(t) * kDgMsgs * 2,
                   (static_cast<unsigned long long>(__float_as_uint(s)) << 32) | __float_as_uint(m2));
    } else {
        if (t == 0) {
            const unsigned long long v = dg_poll(mail, &lost, p.status);
            dg_put(mail, 0ull);
            bc[0] = v;
        }
        __syncthreads();
        if (lost) return;                                  // the leader never answered: nothing of this row is written by a partner
        m2 = __uint_as_float(static_cast<uint32_t>(bc[0]));
        s = __uint_as_float(static_cast<uint32_t>(bc[0] >> 32));
    }
    double L64 = static_cast<double>(m2) + log2_split(s);
    float thr = -INFINITY;
    if (!(s > 0.0f)) {                                     // nothing but -inf logits: no distribution to draw from
        if (leader && t == 0) {
            d.tok[b] = -1;
            if (d.lp) d.lp[b] = -INFINITY;
            if (d.thr) d.thr[b] = -INFINITY;
        }
        return;
    }

    if (d.levels > 0) {
        const float L = static_cast<float>(L64);
        const float p_floor = fmaxf((1.0f - d.top_p) / static_cast<float>(d.V), 1.0f / kDsFix);
        const float x_floor = (L + __builtin_amdgcn_logf(p_floor)) / d.c2;      // p >= p_floor  <=>  x >= x_floor
        const uint32_t low_mask = (1u << p.base_shift) - 1u;
        // the key range of the candidates: from the floor (aligned down to the dtype's key grid) up to the row maximum.  m2 is
        // fl(x_max c2), so x_max <= m2 / c2 up to two roundings: the estimate is pushed up by more than that and by one grid
        // step -- bins above the true maximum stay empty
        const float xm = m2 / d.c2;
        const uint32_t k_top = order_key(xm + fabsf(xm) * 1.0e-6f + 1.0e-30f);
        uint32_t lo = order_key(x_floor) & ~low_mask;
        uint32_t hi = (k_top | low_mask) > 0xffffffffu - (low_mask + 1u) ? 0xffffffffu : (k_top | low_mask) + (low_mask + 1u);
        if (hi < lo) hi = lo | low_mask;
        unsigned long long above = 0ull, target = 0ull, sel_incl = 0ull;
        bool empty = false;
        for (int lv = 0; lv < kDgMaxLevels; ++lv) {
            int shift = p.base_shift;
            while (((hi - lo) >> shift) >= static_cast<uint32_t>(kDsDigits)) ++shift;
            const int bins = static_cast<int>((hi - lo) >> shift) + 1;
            for (int i = t; i < bins; i += kDrThreads) u.hist[i] = 0ull;
            if (t == 0) sel_digit = -1;
            // the tiles in registers are loop-invariant and so is every key and mass derived from them: left alone, the
            // compiler hoists all of it out of the rounds (24-80 live values per lane, 30-227 VGPRs spilled).  The empty asm
            // makes the registers "new" in every round.
#pragma unroll
            for (int j = 0; j < TPW; ++j) asm volatile("" : "+v"(q[j]));
            __syncthreads();                               // (also: the leader's pairs are dead, `u` is the histogram now)
            // every candidate of the registers adds its probability, 2^-40 fixed point, to the slot of its bin
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (t0 + wave + kDrWaves * j < t1) {       // wave-uniform
                    float x[N];
                    unpack<DT>(q[j], x);
                    bool any = false;
#pragma unroll
                    for (int i = 0; i < N; ++i) any = any || (x[i] >= x_floor);
                    if (__ballot(any) != 0ull) {
#pragma unroll
                        for (int i = 0; i < N; ++i) {
                            if (x[i] >= x_floor) {
                                const uint32_t key = order_key(x[i]);
                                if (key >= lo && key <= hi) {
                                    uint32_t bin = (key - lo) >> shift;
                                    bin = bin < static_cast<uint32_t>(bins) ? bin : static_cast<uint32_t>(bins - 1);
                                    atomicAdd(&u.hist[bin], mass_fixed40(fast_exp2(fmaf(x[i], d.c2, -L))));
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
            unsigned long long w0 = 0ull, w1 = 0ull;       // the round's decision
            if (!leader) {
                unsigned long long* out = hist_mine + lv * kDsDigits;     // a round has its own area: no word is reused inside a call
                for (int i = t; i < bins; i += kDrThreads) dg_put(out + i, u.hist[i] | kDgValid);
                if (t < 2) {                               // the decision's two words, polled side by side
                    bc[t] = dg_poll(mail + 2 * (1 + lv) + t, &lost, p.status);
                    dg_put(mail + 2 * (1 + lv) + t, 0ull);
                }
                __syncthreads();
                if (lost) return;
                w0 = bc[0];
                w1 = bc[1];
            } else {
                // gather: every (partner, bin) word has ONE reader -- thread (item mod 1024) -- and all of a thread's loads are in
                // flight together: the words are added into the leader's own histogram with LDS atomics (integer adds commute).
                // (First version: one thread per bin polling its G - 1 partners one after the other = G - 1 DEPENDENT memory
                // round trips, 7 us at G = 8 and 25 us at G = 32.)
                const int items = (G - 1) * bins;
                for (int it0 = t; it0 < items; it0 += 4 * kDrThreads) {
                    unsigned long long* in[4];
                    unsigned long long v[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int it = it0 + q4 * kDrThreads;
                        const int pg = 1 + it / bins, bin = it - (pg - 1) * bins;
                        in[q4] = p.hist_x + ((static_cast<int64_t>(b) * G + pg) * kDgMaxLevels + lv) * kDsDigits + bin;
                        v[q4] = it < items ? __hip_atomic_load(in[q4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kDgValid;
                    }
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int it = it0 + q4 * kDrThreads;
                        if (it < items) {
                            if (v[q4] == 0ull) v[q4] = dg_poll(in[q4], &lost, p.status);      // still in flight: wait for this one
                            dg_put(in[q4], 0ull);
                            const int pg = 1 + it / bins;
                            const unsigned long long add = v[q4] & ~kDgValid;
                            if (add) atomicAdd(&u.hist[it - (pg - 1) * bins], add);
                        }
                    }
                }
                __syncthreads();
                // scan from the top: thread t owns the t-th chunk of bins counted from the TOP
                const int per = (bins + kDrThreads - 1) / kDrThreads;
                const int bhi = bins - t * per, blo = bhi - per > 0 ? bhi - per : 0;
                unsigned long long mine = 0ull;
                for (int j = blo; j < bhi; ++j) mine += u.hist[j];
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
                    target = static_cast<unsigned long long>(static_cast<double>(d.top_p) * static_cast<double>(kDsFix));
                    if (target > total) target = total;      // fixed-point truncation: never ask for more than is there
                    if (target == 0ull) target = 1ull;
                    empty = total == 0ull;
                }
                const unsigned long long before = above + base + incl - mine;
                if (bhi > blo && mine > 0ull && before < target && target <= before + mine) {   // exactly one thread
                    unsigned long long acc = before;
                    int pk = blo;
                    for (int j = bhi - 1; j >= blo; --j) {
                        const unsigned long long m = u.hist[j];
                        if (m > 0ull && acc + m >= target) { pk = j; break; }
                        acc += m;
                    }
                    sel_digit = pk;
                    sel_above = acc;
                    sel_hist = u.hist[pk];
                }
                __syncthreads();
                const int dg = sel_digit;
                if (dg < 0) empty = true;
                w0 = kDgValid | (empty ? (1ull << 62) : 0ull) | (empty ? 0ull : sel_above);
                w1 = kDgValid | (empty ? 0ull : (sel_hist | (static_cast<unsigned long long>(dg) << 44)));
                if (t >= 1 && t < G) {
                    unsigned long long* box = small + 2 * p.n_pad + static_cast<int64_t>(t) * kDgMsgs * 2 + 2 * (1 + lv);
                    dg_put(box, w0);
                    dg_put(box + 1, w1);
                }
                __syncthreads();                           // sel_* are rewritten by the next round
            }
            if (w0 & (1ull << 62)) { empty = true; break; }
            const uint32_t dg = static_cast<uint32_t>((w1 >> 44) & 0xfffu);
            above = w0 & ((1ull << 62) - 1ull);
            sel_incl = above + (w1 & ((1ull << 44) - 1ull));
            const uint32_t nlo = lo + (dg << shift);
            const uint32_t span = shift >= 32 ? 0xffffffffu : ((1u << shift) - 1u);
            hi = (nlo + span < nlo || nlo + span > hi) ? hi : nlo + span;
            lo = nlo;
            if (shift == p.base_shift) break;              // the bin is one key of this dtype: x* is found
        }
        // a 16-bit logit's key carries constant low bits: zeros for x >= 0, ones for x < 0 (the key of a negative float is
        // its complement); the threshold is the logit value itself
        const uint32_t kthr = (lo & 0x80000000u) ? lo : (lo | low_mask);
        thr = empty ? -INFINITY : key_floor_value(kthr);
        // the nucleus normaliser needs no pass of its own: the last round has summed the masses of exactly the tokens >= x*
        if (!empty) L64 = static_cast<double>(L) + log2_split(static_cast<float>(sel_incl)) - 40.0;
    }
    const float Lt = static_cast<float>(L64);

    // ---- tile masses of the (nucleus-restricted) softmax
    if (d.levels == 0) {                                   // leader only: from the pairs, no exp per element
        for (int i = t; i < d.n_tiles; i += kDrThreads) tile_mass[i] = u.pair.s[i] * fast_exp2(u.pair.m[i] - Lt);
    } else {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            asm volatile("" : "+v"(q[j]));
            const int tile = t0 + wave + kDrWaves * j;
            if (tile < t1) {                               // wave-uniform
                float x[N];
                unpack<DT>(q[j], x);
                bool any = false;
#pragma unroll
                for (int i = 0; i < N; ++i) any = any || (x[i] >= thr);
                float z = 0.0f;
                if (__ballot(any) != 0ull) {               // a tile without a survivor keeps mass 0
#pragma unroll
                    for (int i = 0; i < N; ++i) z += x[i] >= thr ? fast_exp2(fmaf(x[i], d.c2, -Lt)) : 0.0f;
                    z = wave_sum(z);
                }
                if (lane == 0) {
                    if (leader) tile_mass[tile] = z;
                    else dg_put(mass_x + tile, kDgValid | __float_as_uint(z));
                }
            }
        }
        if (!leader) return;
        for (int tile = t1 + t; tile < d.n_tiles; tile += kDrThreads) {
            const unsigned long long v = dg_poll(mass_x + tile, &lost, p.status);
            dg_put(mass_x + tile, 0ull);
            tile_mass[tile] = __uint_as_float(static_cast<uint32_t>(v));
        }
    }
    __syncthreads();
    if (wave != 0) return;
    if (lost) {                                            // a hand-off never arrived: the row is poisoned, not guessed
        if (lane == 0) {
            d.tok[b] = -1;
            if (d.lp) d.lp[b] = NAN;
            if (d.thr) d.thr[b] = NAN;
        }
        return;
    }
    if (lane == 0 && d.thr) d.thr[b] = thr;
    draft_pick_wave<DT>(row, d.nvec, tile_mass, d.n_tiles, d.r[b], d.c2, Lt, L64, thr, lane, pick, d.tok + b, d.lp ? d.lp + b : nullptr);
}

}  // namespace
}  // namespace asd

using namespace asd;

#ifdef ASD_STAMP
ASD_EXPORT int asd_debug_draft_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(asd::g_dr_stamp), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

namespace {
#ifdef ASD_TEST_HOOKS         // (process-global, not thread-safe: the TEST build of the library only)
int g_debug_groups = 0;       // asd_debug_draft_groups (tests only): force the workgroups per row; 0 = heuristic
int g_debug_draft_withhold = -1;   // asd_debug_draft_withhold: fault injection for the hand-off tests
#else
constexpr int g_debug_groups = 0;
constexpr int g_debug_draft_withhold = -1;
#endif

struct DgLayout {
    size_t hist_bytes, small_bytes;
    int n_pad;
};
DgLayout dg_layout(int B, int n_tiles) {
    DgLayout l;
    l.n_pad = static_cast<int>(round_up(static_cast<size_t>(n_tiles), 32));
    l.hist_bytes = static_cast<size_t>(kDgSlots) * kDgMaxLevels * kDsDigits * sizeof(unsigned long long);
    l.small_bytes = static_cast<size_t>(B) * (2 * static_cast<size_t>(l.n_pad) + kDgMaxGroups * kDgMsgs * 2) * sizeof(unsigned long long);
    return l;
}
inline int tiles_per_wave(int n_tiles, int G) { return ((n_tiles + G - 1) / G + kDrWaves - 1) / kDrWaves; }

// workgroups per row: ~one workgroup per CU while every workgroup keeps >= 8 tiles; 0 = the row does not fit the registers of
// G workgroups (k_draft_row streams it instead).  ANY count, not only powers of two (round 4; the results do not depend on G):
// B = 33 took G = 4 -- 132 workgroups on 256 CUs, 26.2 us with top-p -- where G = 7 takes 21.1; B = 65 ... 80 G = 2, 39 us, where
// G = 3 takes 31.5 (profiles/r04_draft_groups.json: the time falls monotonically with G at every batch size measured).
int choose_groups(int B, int n_tiles, int cus, bool have_ws) {
    int G = 1;
    if (have_ws && B <= kDgSlots / 2) {
        G = cus / (B > 0 ? B : 1);
        if (G > kDgMaxGroups) G = kDgMaxGroups;
        while (G > 1 && (static_cast<int64_t>(B) * G > kDgSlots || n_tiles / G < 8)) --G;
        if (G < 1) G = 1;
    }
    if (g_debug_groups > 0 && have_ws && g_debug_groups <= kDgMaxGroups && static_cast<int64_t>(B) * g_debug_groups <= kDgSlots) G = g_debug_groups;
    return tiles_per_wave(n_tiles, G) <= 10 ? G : 0;
}

template <int DT>
void launch_group(const DgParams& p, int tpw, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(p.G), static_cast<unsigned>(p.d.B)), block(kDrThreads);
    if (tpw <= 3) hipLaunchKernelGGL((k_draft_group<DT, 3>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((k_draft_group<DT, 10>), grid, block, 0, st, p);
}
}  // namespace

#ifdef ASD_TEST_HOOKS
/* tests only: force the number of workgroups a row is spread over (1, 2, 4, ... 32; -1 = k_draft_row, the one-workgroup
 * streaming form; 0 = heuristic).  Results must not depend on it. */
ASD_EXPORT int asd_debug_draft_groups(int groups) {
    g_debug_groups = groups;
    return ASD_OK;
}
/* tests only: in the asd_draft_sample calls that follow, partner workgroup g (>= 1) of row b, index = b * G + g with G the
 * launch's workgroups per row, does not publish its tile pairs: the leader's bounded wait must end in tok = -1 / lp = NaN for
 * THAT row and ASD_WS_LOST_HANDOFF in the workspace's status word.  index < 0 = off. */
ASD_EXPORT int asd_debug_draft_withhold(int index) {
    g_debug_draft_withhold = index < 0 ? -1 : index;
    return ASD_OK;
}
#endif

ASD_EXPORT size_t asd_draft_sample_workspace_bytes(int B, int V, int dtype) {
    const int esz = dtype_size(dtype);
    if (B <= 0 || V <= 0 || esz == 0) return 256;
    const size_t nvec = (static_cast<size_t>(V) * esz + 15) / 16;
    const DgLayout l = dg_layout(B, static_cast<int>((nvec + 63) / 64));
    return kWorkspaceHeaderBytes + round_up(l.hist_bytes + l.small_bytes, 256);
}

ASD_EXPORT int asd_draft_sample(const void* logits, int64_t ld, int dtype, const float* r, int B, int V,
                                float inv_temperature, float top_p, int32_t* tok, float* lp, float* nucleus_logit,
                                void* workspace, size_t workspace_bytes, void* stream) {
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
    // few rows: spread every row over G workgroups (the workspace carries their mailboxes; without one, or with a row too
    // long for the registers of its workgroups, one streaming workgroup per row)
    const DgLayout l = dg_layout(B, p.n_tiles);
    const bool have_ws = workspace && aligned_to(workspace, 256) && workspace_bytes >= kWorkspaceHeaderBytes + l.hist_bytes + l.small_bytes;
    const int G = g_debug_groups < 0 ? 0 : choose_groups(B, p.n_tiles, current_device_cus(), have_ws);
    if (G >= 1) {
        DgParams q{};
        q.d = p;
        q.G = G;
        char* const body = have_ws ? static_cast<char*>(workspace) + kWorkspaceHeaderBytes : nullptr;    // (behind the status block)
        q.status = have_ws ? static_cast<uint32_t*>(workspace) : nullptr;
        q.hist_x = reinterpret_cast<unsigned long long*>(body);
        q.small = have_ws ? reinterpret_cast<unsigned long long*>(body + l.hist_bytes) : nullptr;
        q.n_pad = l.n_pad;
        q.withhold1 = g_debug_draft_withhold + 1;
        q.base_shift = dtype == ASD_DTYPE_BF16 ? 16 : (dtype == ASD_DTYPE_F16 ? 13 : 0);
        const int tpw = tiles_per_wave(p.n_tiles, G);
        switch (dtype) {
            case ASD_DTYPE_BF16: launch_group<ASD_DTYPE_BF16>(q, tpw, st); break;
            case ASD_DTYPE_F16: launch_group<ASD_DTYPE_F16>(q, tpw, st); break;
            default: launch_group<ASD_DTYPE_F32>(q, tpw, st); break;
        }
        return launch_status();
    }
    const dim3 grid(static_cast<unsigned>(B)), block(kDrThreads);
    switch (dtype) {
        case ASD_DTYPE_BF16: hipLaunchKernelGGL(k_draft_row<ASD_DTYPE_BF16>, grid, block, 0, st, p); break;
        case ASD_DTYPE_F16: hipLaunchKernelGGL(k_draft_row<ASD_DTYPE_F16>, grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL(k_draft_row<ASD_DTYPE_F32>, grid, block, 0, st, p); break;
    }
    return launch_status();
}
