// verify_accept.hip -- batched logits-gather + log-sum-exp + acceptance-ratio test, gfx950.
//
// One streaming pass over target logits [B*K rows][V] (bf16 / f16 / f32).  HBM-bound: every
// logit is read exactly once with 16-byte loads; arithmetic is ~1 v_exp_f32 + 4 VALU per element.
//
// Geometry.  A row (one verified position) is cut into S contiguous vocab slices; one workgroup
// of THREADS lanes reduces one slice to a pair (m2, s) with   sum_v exp(x_v) = s * 2^m2
// (log2 domain: the per-element work is one FMA + one v_exp_f32, and the pair stays exactly
// consistent whatever rounding m2 itself carries).  Each lane keeps a running (m2, s) over its
// 16-byte vectors (online softmax, one rescale per vector); lanes are folded max-first with DPP
// row shifts (one rescale per lane), waves through LDS and the same fold on wave 0.
//
// Streaming loop.  16-byte buffer loads through a per-slice descriptor whose num_records is the
// slice end: lanes past the end are dropped by the range check (no memory access), so loads are
// issued unconditionally.  The slice is cut into UNROLL-KiB tiles that waves claim from an LDS
// counter (two tiles in flight per wave); each tile is reduced to its own (m2, s) slot and the
// slots are folded in tile order, so the result is bitwise independent of the claim order.
//
// One workgroup per row (S == 1, K <= 32; the regime rows >= CUs).  The row's workgroup finishes
// its own row: lane 0 prefetches tok / lp_d / u and the drafted token's logit while the row
// streams, computes lp_t and the accept flag, stores them, and contributes to the sequence's
// ballot with ONE returning 64-bit atomic add of (1 << 32 | flag << k).  The arrival whose old
// count is K-1 holds the complete mask: it writes n_acc / accept_bits and zeroes the word.  No
// payload store, no drain, no read-back on the tail.
//
// Hand-off between workgroups (round 3).  Every word that crosses workgroups has ONE writer and ONE reader and tags itself:
// once published it is never all-zero (a granule's s is > 0 or its m2 is the sentinel; slots carry a tag bit), an empty
// word is.  It is stored write-through (agent scope) and NOT drained; the reader re-reads a word that is still empty --
// bounded, 2^20 polls -- and stores zero back.  A word that never arrives POISONS what depended on it (lp_t = NaN and
// reject for the row; score = NaN, k* = L - 1, stop = 0 for the sequence's epilogue), it is never folded as a zero.
// Who reads is decided by POSITION: a sequence's designated finisher is its LAST workgroup in dispatch order (last row;
// with split rows: last row, last slice); its siblings were dispatched before it and wait for nothing.  No ticket, no epoch
// argument, no memset node: the workspace is all-zero between calls whatever B / K / S they had, so calls of different
// shapes may share one workspace in stream order, and launches replay from a hipGraph.
//
// Split rows (S > 1; few rows, many CUs).  Each slice publishes ONE 8-byte granule {m2, s}; slice 0 of a row, which
// fetched the token id and then the token's logit under its stream, also publishes that logit in the row's slot.  The
// finisher stages the K*S granules in LDS, combines the S slices of every row in slice order (=> bitwise deterministic,
// independent of arrival order), runs the K acceptance tests on K lanes and turns the flags into the accept mask /
// accepted-prefix length with one wave ballot.  With the in-kernel epilogue (FUSED) the same wave, which now holds all K
// lp_t in its lanes, runs the predictor / stop rule on the spot.  (Round 2: a ticket per slice, then the last arriver
// loaded tok and then the logit -- three dependent memory round trips behind the last slice.)
// Granule regions are per sequence and padded to whole 256-byte blocks; every sequence's block starts on its own
// 128-byte line (32 tickets packed in one line made 4096 arrivals cost 80 us -- measured in round 1).
//
// In-kernel epilogue on the one-workgroup-per-row path (FUSED): no ballot atomic -- every row stores
// (1 << 63 | flag << 32 | bits of lp_t) into its slot; the designated finisher issues its feature / p_hist / cost loads,
// polls the K - 1 slots, zeroes them, forms the mask with a wave ballot and its whole wave runs epi_finish_lds (the
// predictor's weights were put into LDS by DMA under the stream; the log-prob statistics come from registers).
//
// Reference arithmetic this replaces: src/training/generate_training_data.py:128-136
// (softmax -> index -> log -> .item(), one token per iteration).  The acceptance test has no
// reference symbol; it is specified in include/asd_hip.h and DESIGN.md.

#include <hip/hip_runtime.h>

#ifdef ASD_STAMP
// Diagnostic build only (tools/stamp_verify.py builds a separate .so with -DASD_STAMP): per-workgroup
// phase stamps of the 100 MHz realtime counter, written to a buffer nothing else reads.
__device__ unsigned long long* g_asd_stamps = nullptr;
#define ASD_STAMP_AT(slot)                                                                       \
    do {                                                                                         \
        if (threadIdx.x == 0 && g_asd_stamps)                                                    \
            g_asd_stamps[static_cast<size_t>(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define ASD_EPI_STAMP(slot) ASD_STAMP_AT(slot)   // stage stamps inside the in-kernel epilogue (predictor_device.hpp)
#else
#define ASD_STAMP_AT(slot) do { } while (0)
#endif

#define ASD_DPP_ASM_REDUCTIONS 1   // wave_max / wave_sum as one DPP instruction per step (lse_device.hpp)
#include "lse_device.hpp"
#include "predictor_device.hpp"   // this TU is built with -ffp-contract=off (numpy / CPython parity of the epilogue)

namespace asd {
namespace {

constexpr int kMaxStage = 1024;        // granules one finisher stages in LDS (K*S <= kMaxStage)
constexpr int kTicketStride = 160;     // u32 units: 640 bytes per sequence = ballot line + four lines of up to 64 hand-off slots (8 bytes each)
constexpr int kLpLineOffset = 32;      // u32 units: where the slots start -- (lp_t, flag) per row with one workgroup per row + in-kernel
                                       // epilogue; the drafted token's logit per row with split rows
constexpr int kSpinLimit = 1 << 20;    // polls of a slot whose store is in flight before its row is poisoned
constexpr int kFastMaxK = 32;          // ballot-by-atomic packs K flags + a 32-bit count in one u64

struct VerifyParams {
    const void* logits;
    int64_t ld_row;
    const int32_t* tok;
    const float* lp_d;
    const float* u;
    int B, K, V, S;
    int64_t v_offset;
    float* lp_t;
    uint8_t* accept;
    int32_t* n_acc;
    uint64_t* bits;
    float* msg;
    uint32_t* tickets;   // one 128-byte line per sequence: u32 ticket at +0, u64 ballot word at +8
    uint64_t* granules;
    uint32_t region;     // granules per sequence region
    float scale2;        // log2(e) / temperature
    int mode;            // 0: accept, 1: emit (m2, s, g) partials
    float* row_max_lp;   // [B,K] out or nullptr: max_v log softmax(x / T)[v]  (= -ln s: free in the epilogue)
    float* row_entropy;  // [B,K] out or nullptr: entropy (nats) of softmax(x / T); needs the STATS instantiation
    int fused;           // != 0: the sequence's last arriver also runs the predictor / stop epilogue (N1)
    int withhold1;       // debug (asd_debug_verify_withhold): 1 + (row * S + split) of the workgroup that never publishes its slot; 0 = off
    FusedParams epi;     // asd_predictor_stop's parameters (lp / n_valid unused: the kernel's own lp_t, all K)
};

// bounded wait for a self-tagging slot (non-zero = published); 0 after kSpinLimit polls = lost
__device__ __forceinline__ uint64_t poll_slot(uint64_t* slot) {
    uint64_t v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spins = 0; v == 0ull && spins < kSpinLimit; ++spins) {   // in flight, not lost: see the publishing side
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v;
}

// the rest of a poll whose first probe `v` was issued earlier (work was placed under it)
__device__ __forceinline__ uint64_t poll_more(uint64_t* slot, uint64_t v) {
    for (int spins = 0; v == 0ull && spins < kSpinLimit; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v;
}

// A hand-off word that never arrived poisons its row / sequence AND is reported: the workspace's first word (the unused u32 at
// +0 of sequence 0's ballot line) is a STICKY status, or-ed by whoever gave up, read by the host with asd_workspace_status
// (or straight from the buffer) at a synchronisation it performs anyway, cleared only by asd_workspace_init.  A workspace
// whose status is non-zero must be re-initialised before it is used again: the word that arrived late was never handed back
// empty, so the "all-zero between calls" invariant no longer holds for it.
__device__ __forceinline__ void report_lost(uint32_t* workspace_words) {
    __hip_atomic_fetch_or(workspace_words, static_cast<uint32_t>(ASD_WS_LOST_HANDOFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the epilogue's outputs for a sequence whose hand-off was lost: nothing downstream may read them as a verdict
__device__ __forceinline__ void epi_poison(const FusedParams& e, int b) {
    if (e.score) e.score[b] = NAN;
    if (e.k_star) e.k_star[b] = e.L - 1;
    if (e.stop) e.stop[b] = 0;
    if (e.thr_stop) e.thr_stop[b] = 0;
    if (e.stats) {
        for (int i = 0; i < ASD_NUM_LP_STATS; ++i) e.stats[ASD_NUM_LP_STATS * static_cast<int64_t>(b) + i] = NAN;
    }
}

// k_verify's kernarg segment: (void*, int32_t*, int64_t, int, int, int, float, int) = 44 bytes, then the by-value VerifyParams
// at the next multiple of its alignment
constexpr int kKernargParamsOffset = 48;
static_assert(alignof(VerifyParams) == 8, "kKernargParamsOffset assumes an 8-byte aligned parameter struct");
// an opaque definition of a wave-uniform value: whatever is derived from it below cannot be re-loaded from the kernarg segment
#define ASD_PIN(x) asm volatile("" : "+s"(x))
// ... and for a pointer: its bits are pinned, then turned back into a GLOBAL pointer.  A pointer that comes out of an asm (or out
// of a hand-written load of the kernarg segment) is a generic pointer to the compiler, and what it dereferences becomes flat_load /
// flat_store -- which count on lgkmcnt as well as vmcnt, i.e. every wait for an LDS operation also waits for them (two such loads
// in front of the tile loop, whose tile claims are LDS atomics, cost the whole kernel 1 us: profiles/r04_lab_flat_loads.json).
template <typename T>
__device__ __forceinline__ T* pin_global_ptr(T* p) {
    uint64_t bits = reinterpret_cast<uint64_t>(p);
    asm volatile("" : "+s"(bits));
    return (T*)(__attribute__((address_space(1))) T*)bits;     // integer -> GLOBAL pointer (-> generic, which the compiler sees through)
}
#define ASD_PIN_PTR(x) do { x = pin_global_ptr(x); } while (0)
__device__ __forceinline__ void pin_epilogue_params(FusedParams& e) {
    ASD_PIN(e.K); ASD_PIN_PTR(e.feat); ASD_PIN(e.ldf); ASD_PIN(e.stats_col); ASD_PIN_PTR(e.packed);
    ASD_PIN(e.risk); ASD_PIN(e.n_obs); ASD_PIN(e.alpha); ASD_PIN(e.beta);
    ASD_PIN_PTR(e.p_hist); ASD_PIN_PTR(e.C); ASD_PIN(e.lam); ASD_PIN(e.L); ASD_PIN(e.stage_idx); ASD_PIN(e.prefix);
    ASD_PIN_PTR(e.theta); ASD_PIN_PTR(e.score); ASD_PIN_PTR(e.k_star); ASD_PIN_PTR(e.stop); ASD_PIN_PTR(e.thr_stop); ASD_PIN_PTR(e.stats);
}

template <int DT, int UNROLL, bool CHECK, bool STATS>
__device__ __forceinline__ void consume(const u32x4 (&r)[UNROLL], uint32_t off, uint32_t end, float c2, float& m2, float& s,
                                        float& t) {
    using E = Elem<DT>;
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
        u32x4 v = r[j];
        if (CHECK) {  // past the slice end the range check returned zeros: make them -inf
            const bool ok = off + static_cast<uint32_t>(j) * 1024u < end;
            const u32x4 neg = {E::kNegInfWord, E::kNegInfWord, E::kNegInfWord, E::kNegInfWord};
            v = ok ? v : neg;
        }
        if (STATS) E::accum3(v, c2, m2, s, t);
        else E::accum(v, c2, m2, s);
    }
}

// FUSED: compile-time variant that also carries the in-kernel epilogue (N1 second form).  It is a separate
// instantiation on purpose: the epilogue's prefetch registers and code took the plain kernel from ~40 to 111
// VGPRs and cost it 6 % (A/B in one process), so the plain kernel does not contain it at all.
// max_v log softmax(x/T)[v]: the running maximum IS m2 (up to the rounding of x_max * c2, |m2| * 6e-8 in log2 units), so
// the largest term of s is 1 and the row's top log-probability is -ln s.  s == 0 (nothing but -inf): NaN.
__device__ __forceinline__ float row_max_logprob(float s) {
    return s > 0.0f ? static_cast<float>(-kLn2d * log2_split(s)) : NAN;
}
// H = -sum p ln p = ln2 * (m2 + log2 s - c2 * t / s)  with  p_v = 2^(x_v c2 - m2) / s,  t = sum 2^(x_v c2 - m2) x_v
__device__ __forceinline__ float row_entropy_nats(float m2, float s, float t, float c2) {
    if (!(s > 0.0f)) return NAN;
    const double L = static_cast<double>(m2) + log2_split(s);
    return static_cast<float>(kLn2d * (L - static_cast<double>(c2) * static_cast<double>(t) / static_cast<double>(s)));
}

// The first eight arguments repeat the VerifyParams fields the streaming prologue needs (row base, extents, tok).
// They are separate scalars ON PURPOSE: this file is compiled with -amdgpu-kernarg-preload-count, so the command
// processor hands them to every wave in SGPRs at wave start and the first tile loads are issued without waiting for
// any s_load of the kernarg segment (the by-value struct is fetched meanwhile and first used behind those loads).
// STATS: additionally carries t = sum e * x through the stream (one more FMA per element, a third slot value per tile)
// for the row entropy (asd_verify_accept_stats); its own instantiation so that the plain kernel pays nothing.
// EPI (FUSED only): 1 = the reference's 64 -> 32 -> 1 predictor (weights in LDS by DMA, one wave), 2 = the 256 -> 128 -> 1 predictor
// of the reference's server (first layer cut over the eight waves of the finisher's workgroup: predictor_device.hpp, epi2_*; one
// workgroup per row only -- the launcher sends split-row launches down the two-launch route).
template <int DT, int THREADS, int UNROLL, bool NT, bool FUSED, bool STATS = false, int EPI = 1>
__global__ __launch_bounds__(THREADS) void k_verify(const void* a_logits, const int32_t* a_tok, int64_t a_ld_row, int a_V,
                                                    int a_K, int a_S, float a_scale2, int a_own, const VerifyParams p) {
    using E = Elem<DT>;
    constexpr int kWaves = THREADS / 64;
    __shared__ uint32_t next_tile;
    __shared__ __attribute__((aligned(16))) uint64_t stage[kMaxStage];   // tile slots while streaming, then scratch of the finisher
    __shared__ float stage_t[STATS ? kMaxStage : 1];                      // STATS: the tiles' third value
    static_assert(EPI == 1 || (EPI == 2 && THREADS == 64 * kEpi2Waves), "the 256 x 128 epilogue is cut over eight waves");
    // FUSED: the predictor's packed weights (EPI 1, LDS-DMA) / the eight waves' partial first-layer sums (EPI 2)
    __shared__ __attribute__((aligned(16))) float wlds[FUSED ? (EPI == 2 ? kEpi2Waves * kEpi2Hid : 64 * 32 + 68) : 4];
    __shared__ __attribute__((aligned(16))) double dlds[FUSED ? kEpiDecideLds : 1];  // FUSED: the decision inputs (LDS-DMA)

    ASD_STAMP_AT(0);
#ifdef ASD_STAMP
    if (threadIdx.x == 0 && g_asd_stamps)   // HW_REG_XCC_ID (id 20), all 32 bits
        g_asd_stamps[static_cast<size_t>(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int S = a_S;
    const float c2 = a_scale2;
    // grid = (rows, splits): no division stands between the wave's start and its first loads (as a 1-D grid the prologue
    // held three 64-bit and one 32-bit software divisions -- ~570 scalar instructions, ~2.8 us in the stamped build)
    const int row = static_cast<int>(blockIdx.x);
    const int split = static_cast<int>(blockIdx.y);

    const char* rowp = static_cast<const char*>(a_logits) + static_cast<int64_t>(row) * a_ld_row * E::kBytes;
    const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(rowp) & 15u);
    int head = mis ? static_cast<int>((16u - mis) / E::kBytes) : 0;
    if (head > a_V) head = a_V;
    const int nvec = (a_V - head) / E::kPerVec;
    const int tail = a_V - head - nvec * E::kPerVec;
    const char* body = rowp + static_cast<int64_t>(head) * E::kBytes;
    int v0 = 0, v1 = nvec;
    if (S > 1) {   // launcher: V * S < 2^31, so the products fit 32 bits
        v0 = static_cast<int>(static_cast<uint32_t>(nvec) * static_cast<uint32_t>(split) / static_cast<uint32_t>(S));
        v1 = static_cast<int>(static_cast<uint32_t>(nvec) * static_cast<uint32_t>(split + 1) / static_cast<uint32_t>(S));
    }

    // the row's own scalars, fetched by lane 0 under the stream (one workgroup per row only).  Every one of these
    // small loads is ISSUED early and CONSUMED after the streaming loop (the compiler waits at the first use, and the
    // loop's LDS atomics keep it from sinking the loads).  Nothing before the barrier below waits for memory: the
    // earlier form converted the gathered logit on the spot, which put an `s_waitcnt vmcnt(0)` in front of the
    // barrier -- wave 0 sat there until its first two tiles AND the gather (queued behind the whole CU's initial
    // requests) had landed, and with it every wave of the workgroup (stamps: first tile consumed 4.2 us after the start).
    const bool own_row = a_own != 0;   // launcher: (S == 1) && (mode == 1 || K <= kFastMaxK)
    float x_tok = -INFINITY, lpd = 0.0f, uu = 1.0f;
    uint32_t raw_tok = 0, raw_lpd = 0, raw_u = 0x3f800000u, raw_x = 0;
    bool have_x = false;
    double lu_row = 0.0;
    // vector loads (the index is laundered through a VGPR): a scalarised s_load would share lgkmcnt with the LDS
    // traffic of the loop and stall wave 0 at its first tile claim
    int vrow = row;
    asm volatile("" : "+v"(vrow));
    // split rows: slice 0 of a row fetches the drafted token's logit under its stream and hands it to the finisher (round 2: the
    // finisher loaded tok, then the logit -- two dependent round trips on the tail of every sequence)
    const bool gather = own_row || split == 0;
    if (gather && tid == 0) raw_tok = static_cast<uint32_t>(a_tok[vrow]);   // the oldest load of wave 0: waited for alone

    // ---- streaming: waves claim UNROLL-KiB tiles from an LDS counter ---------------------------
    // Static striding lets the oldest wave group run ahead (age-priority arbitration: measured
    // stream ends 11.9 / 12.6 / 13.4 / 14.6 us for the four groups of a 1024-lane workgroup), so
    // the CU's memory pipe idles while the youngest waves drain alone.  With dynamic claims all
    // waves end within one tile of each other.  Determinism is kept by construction: every tile
    // is reduced to its OWN (m2, s) slot, and slots are combined in tile order afterwards, so the
    // result does not depend on which wave processed which tile.
    constexpr uint32_t kTileBytes = static_cast<uint32_t>(UNROLL) * 1024u;
    const uint32_t end = v1 > v0 ? static_cast<uint32_t>(v1 - v0) * 16u : 0u;
    const uint32_t n_tiles = (end + kTileBytes - 1) / kTileBytes;   // launcher: n_tiles + 1 <= kMaxStage
    const uint32_t n_full = end / kTileBytes;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(body) + static_cast<int64_t>(v0) * 16, 0, static_cast<int>(end), 0x00020000);
    const uint32_t lane_off = static_cast<uint32_t>(lane) * 16u;
    if (tid == 0) next_tile = 2u * kWaves;
    uint32_t ta = static_cast<uint32_t>(wave), tb = static_cast<uint32_t>(wave) + kWaves;
    u32x4 ra[UNROLL], rb[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) ra[j] = load16<NT>(rsrc, ta * kTileBytes + static_cast<uint32_t>(j) * 1024u + lane_off);
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) rb[j] = load16<NT>(rsrc, tb * kTileBytes + static_cast<uint32_t>(j) * 1024u + lane_off);
    ASD_STAMP_AT(7);
    __syncthreads();   // next_tile is visible; the first two tiles of every wave are already in flight

    // behind the barrier, still ahead of the loop: the loads whose ADDRESS needs a loaded value (the drafted token's
    // logit) and the <= 7-element unaligned head / ragged tail of the row -- issued, not waited for
    asm volatile("" : "+v"(raw_tok));   // first use of the token id: here, not hoisted in front of the tile loads
    const int64_t t_tok = static_cast<int64_t>(static_cast<int32_t>(raw_tok)) - p.v_offset;
    if (gather && tid == 0 && t_tok >= 0 && t_tok < a_V) {
        raw_x = E::raw(rowp, t_tok);
        have_x = true;
    }
    if (own_row && tid == 0 && p.mode == 0) {   // (the by-value struct is first used behind the barrier)
        raw_lpd = reinterpret_cast<const uint32_t*>(p.lp_d)[vrow];
        raw_u = reinterpret_cast<const uint32_t*>(p.u)[vrow];
    }
    uint32_t raw_head = 0, raw_tail = 0, raw_feat = 0;
    const bool do_head = wave == 0 && split == 0 && lane < head;
    const bool do_tail = wave == 0 && split == S - 1 && lane >= 32 && lane - 32 < tail;
    if (do_head) raw_head = E::raw(rowp, lane);
    if (do_tail) raw_tail = E::raw(rowp, static_cast<int64_t>(head) + static_cast<int64_t>(nvec) * E::kPerVec + (lane - 32));
    const int b = static_cast<int>(static_cast<uint32_t>(row) / static_cast<uint32_t>(a_K));
    const int k = row - b * a_K;
    const bool fused = FUSED && p.mode == 0;   // one workgroup per row: the sequence's designated finisher; split rows: the last arriver
#define ASD_CLAIM(dst)                                                                                   \
    do {                                                                                                 \
        uint32_t c_ = 0;                                                                                 \
        if (lane == 0) c_ = __hip_atomic_fetch_add(&next_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        dst = __builtin_amdgcn_readfirstlane(c_);                                                        \
    } while (0)
#define ASD_REDUCE_TILE(reg, t)                                                                          \
    do {                                                                                                 \
        float tm_ = kSentinel, ts_ = 0.0f, tt_ = 0.0f;                                                   \
        if ((t) < n_full) consume<DT, UNROLL, false, STATS>(reg, (t) * kTileBytes + lane_off, end, c2, tm_, ts_, tt_);   \
        else consume<DT, UNROLL, true, STATS>(reg, (t) * kTileBytes + lane_off, end, c2, tm_, ts_, tt_);                 \
        if (STATS) wave_merge3(tm_, ts_, tt_);                                                           \
        else wave_merge(tm_, ts_);                                                                       \
        if (lane == 0) {                                                                                 \
            stage[(t)] = (static_cast<uint64_t>(__float_as_uint(ts_)) << 32) | __float_as_uint(tm_);     \
            if (STATS) stage_t[(t)] = tt_;                                                               \
        }                                                                                                \
    } while (0)

    bool first = true;
    while (ta < n_tiles) {   // claims are monotonic per wave: ta < tb at the top of every iteration
        uint32_t tn;
        ASD_CLAIM(tn);
        ASD_REDUCE_TILE(ra, ta);
        if (first) { ASD_STAMP_AT(1); first = false; }
        ta = tn;
#if defined(ASD_LAB) && ASD_LAB == 10    // lab: NO loads in the loop (the registers of the first two tiles are consumed again and again): the loop's VALU + claim time alone
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) asm volatile("" : "+v"(ra[j]));
#else
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) ra[j] = load16<NT>(rsrc, ta * kTileBytes + static_cast<uint32_t>(j) * 1024u + lane_off);
#endif
        if (tb >= n_tiles) break;
        ASD_CLAIM(tn);
        ASD_REDUCE_TILE(rb, tb);
        tb = tn;
#if defined(ASD_LAB) && ASD_LAB == 10
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) asm volatile("" : "+v"(rb[j]));
#else
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) rb[j] = load16<NT>(rsrc, tb * kTileBytes + static_cast<uint32_t>(j) * 1024u + lane_off);
#endif
    }
#undef ASD_CLAIM
#undef ASD_REDUCE_TILE

    ASD_STAMP_AT(2);
#ifdef ASD_STAMP
    if (lane == 0 && g_asd_stamps)   // per-wave stream end, after the per-workgroup records
        g_asd_stamps[static_cast<size_t>(gridDim.x * gridDim.y) * 16 + static_cast<size_t>(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave] = __builtin_amdgcn_s_memrealtime();
#endif
#if defined(ASD_LAB) && (ASD_LAB == 8 || ASD_LAB == 10)     // lab: the kernel ends behind the stream (no barrier, no fold, no tail): ramp + stream + per-tile math alone
    if (lane == 0 && stage[0] == 0x1234567812345678ull) p.lp_t[row] = 0.0f;
    return;
#endif
    FusedParams ep;   // FUSED: the epilogue's parameters (wave 0 only; fetched below)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // EPI 2: every wave of the finisher's workgroup computes a slice of the predictor's first layer behind the barrier below; what
    // it needs of the parameters (feature row, packed weights, statistics column) it fetches itself, here, behind its stream
    const bool fin_wg = FUSED && fused && k == a_K - 1 && (own_row || split == S - 1);
    float xv2 = 0.0f;
    const float* packed2 = nullptr;
    int col2 = -1;
    if constexpr (FUSED && EPI == 2) {
        if (fin_wg) {
            typedef const __attribute__((address_space(4))) char* ka_ptr2;
            typedef const __attribute__((address_space(4))) FusedParams* ka_epi_ptr2;
            ka_ptr2 ka2 = (ka_ptr2)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka2));
            const ka_epi_ptr2 kp2 = (ka_epi_ptr2)(ka2 + kKernargParamsOffset + __builtin_offsetof(VerifyParams, epi));
            const float* feat2 = pin_global_ptr(kp2->feat);
            int64_t ldf2 = kp2->ldf;
            col2 = kp2->stats_col;
            packed2 = pin_global_ptr(kp2->packed);
            ASD_PIN(ldf2); ASD_PIN(col2);
            xv2 = epi2_feature(feat2 + static_cast<int64_t>(b) * ldf2, col2, wave_u, lane);
        }
    }
    if (wave_u == 0) {   // (wave-uniform for the compiler too: the parameters below live in SGPRs)
        // the small loads issued ahead of the loop are consumed here: head / tail elements -> slot n_tiles,
        // the drafted token's logit and log(u) for lane 0
        // pin the first use of the loaded values HERE: without it the compiler hoists the (cheap, speculatable)
        // conversions up to the loads and waits for them in front of the loop
        asm volatile("" : "+v"(raw_x), "+v"(raw_head), "+v"(raw_tail), "+v"(raw_lpd), "+v"(raw_u));
        if (FUSED) {
            // The epilogue's parameters are fetched HERE -- behind the stream, by wave 0 alone, in one batch of scalar loads that
            // completes under the barrier below -- through a kernarg pointer the compiler cannot see through, and pinned in SGPRs.
            // Left to the compiler, the loads of the by-value struct land wherever their first use is: in round 3 right behind
            // the FIRST barrier (the weight DMA used three of the fields there).  The kernarg segment is not served from a cache: a
            // scalar load issued once the chip-wide burst of tile loads exists waits ~1 us behind 12 MB of streaming requests,
            // and EVERY wave of EVERY workgroup waited for it before its first tile claim (profiles/r04_lab_bisect.json: fused
            // without hand-off and epilogue 16.33 us, the same with that block compiled out 15.07 us, plain 15.37 us); the
            // finisher's tail paid three more dependent waits of the same kind.
            typedef const __attribute__((address_space(4))) char* ka_ptr;
            typedef const __attribute__((address_space(4))) FusedParams* ka_epi_ptr;
            ka_ptr ka = (ka_ptr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const ka_epi_ptr kp = (ka_epi_ptr)(ka + kKernargParamsOffset + __builtin_offsetof(VerifyParams, epi));
            ep.lp = nullptr; ep.ld_lp = 0; ep.n_valid = nullptr; ep.in_dim = 64; ep.hidden = 32; ep.use_lds = 0; ep.B = 0;   // (unused by the in-kernel epilogue)
            ep.K = kp->K; ep.feat = kp->feat; ep.ldf = kp->ldf; ep.stats_col = kp->stats_col; ep.packed = kp->packed;
            ep.risk = kp->risk; ep.n_obs = kp->n_obs; ep.alpha = kp->alpha; ep.beta = kp->beta;
            ep.p_hist = kp->p_hist; ep.C = kp->C; ep.lam = kp->lam; ep.L = kp->L; ep.stage_idx = kp->stage_idx; ep.prefix = kp->prefix;
            ep.theta = kp->theta; ep.score = kp->score; ep.k_star = kp->k_star; ep.stop = kp->stop; ep.thr_stop = kp->thr_stop;
            ep.stats = kp->stats;
            pin_epilogue_params(ep);
            // The predictor's 2113 packed weights go STRAIGHT into LDS (global_load_lds: no VGPR is held; as registers across the
            // stream they took this instantiation to 135 VGPRs = one workgroup per CU), issued by the wave that can end up running
            // the epilogue: the designated finisher's (the sequence's last row; with split rows its last slice).  8 x 1 KiB (W1^T),
            // then b1 / W2 (64 floats) and b2 as dwords; with them the sequence's feature row, one value per lane.  They fly under
            // the barrier, the slot fold and finish_row; phase A of the first layer waits for them (vmcnt) -- behind the stream the
            // chip is quiet and the round trip short.  (Round 3 issued them in front of the stream: see above.)
#if defined(ASD_LAB) && ASD_LAB == 5   // lab: no weight DMA, no feature load
            if (false) {
#else
            if (EPI == 1 && fin_wg) {
#endif
                typedef __attribute__((address_space(3))) void lds_void;
                typedef const __attribute__((address_space(1))) void glb_void;
                const char* src = reinterpret_cast<const char*>(ep.packed);
#pragma unroll
                for (int ps = 0; ps < 8; ++ps)
                    __builtin_amdgcn_global_load_lds((glb_void*)(src + ps * 1024 + lane * 16), (lds_void*)(reinterpret_cast<char*>(wlds) + ps * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void*)(src + 8192 + lane * 4), (lds_void*)(reinterpret_cast<char*>(wlds) + 8192), 4, 0, 0);
                if (lane == 0)
                    __builtin_amdgcn_global_load_lds((glb_void*)(src + 8192 + 256), (lds_void*)(reinterpret_cast<char*>(wlds) + 8192 + 256), 4, 0, 0);
                raw_feat = reinterpret_cast<const uint32_t*>(ep.feat)[static_cast<int64_t>(b) * ep.ldf + lane];
            }
        }
        float hm = kSentinel, hs = 0.0f, ht = 0.0f;
        if (do_head) {
            const float xv = E::from_raw(raw_head);
            accum_scalar(xv, c2, hm, hs);
            if (STATS) ht = hs > 0.0f ? hs * xv : 0.0f;      // one element per lane: s = 2^(x c2 - m2) = 1 (or 0 for -inf)
        }
        if (do_tail) {
            const float xv = E::from_raw(raw_tail);
            accum_scalar(xv, c2, hm, hs);
            if (STATS) ht = hs > 0.0f ? hs * xv : 0.0f;
        }
        if (STATS) wave_merge3(hm, hs, ht);
        else wave_merge(hm, hs);
        if (lane == 0) {
            stage[n_tiles] = (static_cast<uint64_t>(__float_as_uint(hs)) << 32) | __float_as_uint(hm);
            if (STATS) stage_t[n_tiles] = ht;
        }
        if (gather && tid == 0) {
            if (have_x) x_tok = E::from_raw(raw_x);
            if (own_row && p.mode == 0) {
                lpd = __uint_as_float(raw_lpd);
                uu = __uint_as_float(raw_u);
                lu_row = log_u(uu);
            }
        }
    }
    // tile slots -> slice: wave 0 folds slots lane, lane+64, ... in order, then across lanes
    __syncthreads();
    if (wave_u != 0) {
        if constexpr (FUSED && EPI == 2) {
            if (fin_wg) {   // this wave's 32 columns of the predictor's first layer -> LDS; wave 0 joins the barrier behind its own slice
                float h0, h1;
                epi2_partial(packed2, wave_u, lane, xv2, h0, h1);
                wlds[wave_u * kEpi2Hid + lane] = h0;
                wlds[wave_u * kEpi2Hid + 64 + lane] = h1;
                __syncthreads();
            }
        }
        return;
    }
    float m2 = kSentinel, s = 0.0f, tsum = 0.0f;
    for (uint32_t t = static_cast<uint32_t>(lane); t <= n_tiles; t += 64) {
        const uint64_t g = stage[t];
        if (STATS) ms_merge3(m2, s, tsum, __uint_as_float(static_cast<uint32_t>(g)), __uint_as_float(static_cast<uint32_t>(g >> 32)), stage_t[t]);
        else ms_merge(m2, s, __uint_as_float(static_cast<uint32_t>(g)), __uint_as_float(static_cast<uint32_t>(g >> 32)));
    }
    if (STATS) wave_merge3(m2, s, tsum);
    else wave_merge(m2, s);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the slots are dead; `stage` is reused below
    __builtin_amdgcn_wave_barrier();
    // only wave 0 is left here
    ASD_STAMP_AT(3);

    if (own_row) {
        // ---- one workgroup per row: finish the row here ---------------------------------------
        if (!(FUSED && fused) && lane != 0) return;
        if (p.mode == 1) {
            p.msg[3 * row + 0] = m2;
            p.msg[3 * row + 1] = s;
            p.msg[3 * row + 2] = x_tok;
            return;
        }
        uint32_t* line = p.tickets + static_cast<int64_t>(b) * kTicketStride;
        if (FUSED && fused) {
            // ---- in-kernel epilogue, one workgroup per row.  No ballot atomic at all: every row hands (lp_t, accept flag) to
            // the sequence's DESIGNATED finisher -- the workgroup of its last row, dispatched after its siblings -- as one
            // self-tagging 8-byte slot, write-through and not drained; the finisher polls the K - 1 slots (bounded; a lost
            // slot poisons the epilogue), hands them back empty, forms the accept mask with a wave ballot and runs the
            // predictor / stop rule on the spot.  (Round 2: slot store + DRAIN + returning ballot atomic + read-back by the
            // last arriver = two more memory round trips on the tail.)
            float lp = 0.0f;
            bool flag = false;
            if (lane == 0) flag = finish_row(m2, s, x_tok, c2, lpd, lu_row, lp);
            uint64_t* slots = reinterpret_cast<uint64_t*>(line + kLpLineOffset);
            const uint64_t mine = (1ull << 63) | (static_cast<uint64_t>(flag ? 1u : 0u) << 32) | __float_as_uint(lp);
            if (k != p.K - 1) {
                if (lane == 0) {   // the hand-off first: it is what the sequence's finisher waits for
                    if (row + 1 != p.withhold1) __hip_atomic_store(slots + k, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    p.lp_t[row] = lp;
                    p.accept[row] = flag ? 1 : 0;
                    if (p.row_max_lp) p.row_max_lp[row] = row_max_logprob(s);
                }
                return;
            }
            ASD_STAMP_AT(4);
            uint64_t sv = __shfl(mine, 0, 64);              // the finisher's own row (lane K - 1 keeps it)
            EpiPhaseA pa{};
            Epi2Tail tail2{};
            if constexpr (EPI == 1) {
                // the weight DMA and the feature row were issued behind the stream, a barrier + slot fold + finish_row ago: landed
                // (nothing younger is outstanding yet, so this wait names exactly them)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw_feat) :: "memory");
                if (lane < p.K - 1) sv = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // first probe: issued ...
                // ... and phase A of the predictor's first layer (everything but the five statistics columns) runs under it
                epi_phase_a_lds(ep, lane, __uint_as_float(raw_feat), wlds, reinterpret_cast<float*>(stage + 192), pa);
                uint32_t sv_lo = static_cast<uint32_t>(sv), sv_hi = static_cast<uint32_t>(sv >> 32);
                asm volatile("" : "+v"(pa.h), "+v"(pa.wd[0]), "+v"(pa.wd[1]), "+v"(pa.wd[2]), "+v"(pa.wd[3]), "+v"(pa.wd[4]), "+v"(pa.b1), "+v"(pa.w2),
                             "+v"(pa.b2), "+v"(sv_lo), "+v"(sv_hi));   // the probe's first use is HERE, behind phase A
                sv = (static_cast<uint64_t>(sv_hi) << 32) | sv_lo;
            } else {
                // EPI 2: the first probe, then wave 0's own 32 columns of the first layer and the weights the tail needs -- the
                // other seven waves have been computing theirs since the barrier; all eight meet at the barrier below
                if (lane < p.K - 1) sv = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                epi2_tail_prefetch(packed2, col2, lane, tail2);
                float h0, h1;
                epi2_partial(packed2, 0, lane, xv2, h0, h1);
                wlds[lane] = h0;
                wlds[64 + lane] = h1;
                __syncthreads();
                uint32_t sv_lo = static_cast<uint32_t>(sv), sv_hi = static_cast<uint32_t>(sv >> 32);
                asm volatile("" : "+v"(sv_lo), "+v"(sv_hi));
                sv = (static_cast<uint64_t>(sv_hi) << 32) | sv_lo;
            }
            epi_decide_dma(ep, b, lane, dlds);           // the decision inputs (p_hist row, stage costs, theta) -> LDS while the slots are polled
            if (lane == 0) {
                p.lp_t[row] = lp;
                p.accept[row] = flag ? 1 : 0;
                if (p.row_max_lp) p.row_max_lp[row] = row_max_logprob(s);
            }
#if defined(ASD_LAB) && (ASD_LAB == 2 || ASD_LAB == 5)      // lab build (tools/lab_fused_tail.py): no wait for the siblings at all
            sv = __shfl(mine, 0, 64);
            if (lane < p.K - 1) __hip_atomic_store(slots + lane, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
            if (lane < p.K - 1) {
                sv = poll_more(slots + lane, sv);
                __hip_atomic_store(slots + lane, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // empty again for the next call
            }
#endif
            const bool lost = lane < p.K && sv == 0ull;
            const unsigned long long bal = __ballot(lane < p.K && ((sv >> 32) & 1ull));
            if (lane == 0) {
                const unsigned long long inv = ~bal;
                const int n = inv ? __builtin_ctzll(inv) : 64;
                p.n_acc[b] = n < p.K ? n : p.K;
                if (p.bits) p.bits[b] = bal;
            }
            ASD_STAMP_AT(5);
            if (__ballot(lost) != 0ull) {
                if (lane == 0) {
                    epi_poison(ep, b);
                    report_lost(p.tickets);
                }
                return;
            }
#if defined(ASD_LAB) && (ASD_LAB == 1 || ASD_LAB == 2 || ASD_LAB == 5)   // lab build: no epilogue (what the step costs without it)
            if (lane == 0 && ep.score) ep.score[b] = pa.h + pa.wd[0] + pa.b1 + pa.w2 + pa.b2 + __uint_as_float(static_cast<uint32_t>(sv));
            return;
#endif
            if constexpr (EPI == 1)
                epi_finish_lds(ep, b, lane, __uint_as_float(static_cast<uint32_t>(sv)), p.K, ep.stats_col >= 0 || ep.stats != nullptr, dlds, pa);
            else
                epi2_finish_lds(ep, b, lane, __uint_as_float(static_cast<uint32_t>(sv)), p.K, ep.stats_col >= 0 || ep.stats != nullptr, dlds, wlds, tail2);
            ASD_STAMP_AT(8);
            return;
        }
        if (lane == 0) {
            float lp;
            const bool flag = finish_row(m2, s, x_tok, c2, lpd, lu_row, lp);
            p.lp_t[row] = lp;
            p.accept[row] = flag ? 1 : 0;
            if (p.row_max_lp) p.row_max_lp[row] = row_max_logprob(s);
            if (STATS) {
                if (p.row_entropy) p.row_entropy[row] = row_entropy_nats(m2, s, tsum, c2);
            }
            // ballot by atomic: count in the high word, this row's flag at bit k of the low word
            uint64_t* word = reinterpret_cast<uint64_t*>(line + 2);
            const uint64_t mine = (1ull << 32) | (static_cast<uint64_t>(flag ? 1u : 0u) << k);
            ASD_STAMP_AT(4);
#if defined(ASD_LAB) && ASD_LAB == 9     // lab: no ballot atomic (what the tail's one memory round trip costs)
            if (mine == 0x1234ull) p.n_acc[b] = 0;
            return;
#endif
            const uint64_t old = __hip_atomic_fetch_add(word, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (static_cast<uint32_t>(old >> 32) == static_cast<uint32_t>(p.K - 1)) {
                const uint32_t mask = static_cast<uint32_t>(old | mine);
                const uint32_t inv = ~mask;
                const int n = inv ? __builtin_ctz(inv) : 32;
                p.n_acc[b] = n < p.K ? n : p.K;
                if (p.bits) p.bits[b] = mask;
                __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#ifdef ASD_STAMP
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (g_asd_stamps) g_asd_stamps[static_cast<size_t>(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 5] = __builtin_amdgcn_s_memrealtime() + (old & 0);
#endif
        }
        return;
    }

    // ---- split rows: publish the slice; the sequence's designated finisher collects -----------------------------
    // No ticket (round 3): the finisher is the LAST workgroup of the sequence in dispatch order -- last row, last slice -- and
    // simply polls every slot of its sequence (self-tagging, bounded, poisoning; each slot has this one reader).  Round 2
    // took a ticket per slice and let the last arriver load tok and then the token's logit: three dependent memory round
    // trips (ticket, tok, logit) behind the last slice of every sequence, ~3 us of an 8 us launch at B = 8.
    const int KS = p.K * S;
    uint64_t* region = p.granules + static_cast<int64_t>(b) * p.region;
    uint64_t* xslots = reinterpret_cast<uint64_t*>(p.tickets + static_cast<int64_t>(b) * kTicketStride + kLpLineOffset);
    if (lane == 0 && row * S + split + 1 != p.withhold1) {
        // a granule is never all-zero bits (s > 0, or m2 is the sentinel), an EMPTY slot is: the granule tags itself
        const uint64_t g = (static_cast<uint64_t>(__float_as_uint(s)) << 32) | __float_as_uint(m2);
        __hip_atomic_store(region + k * S + split, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (split == 0)   // x_tok may be -inf (token outside the shard) or NaN: the tag bit keeps the slot non-zero
            __hip_atomic_store(xslots + k, (1ull << 63) | __float_as_uint(x_tok), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (k != p.K - 1 || split != S - 1) return;

    // ---- designated finisher of sequence b: finish its K rows ------------------------------------
    const int frow = b * p.K + lane;  // lane <-> draft position
    // first probes of the K logit slots and of the first 64 granules: issued, then (FUSED) phase A of the predictor's first
    // layer runs under them
    if (FUSED) {   // the weight DMA and the feature row (issued behind the stream) have landed; nothing younger is outstanding yet
        if (fused) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw_feat) :: "memory");
    }
    uint64_t xv0 = 0ull, gv0 = 0ull;
    if (lane < p.K) {
        if (p.mode == 0) { lpd = p.lp_d[frow]; uu = p.u[frow]; }
        xv0 = __hip_atomic_load(xslots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane < KS) gv0 = __hip_atomic_load(region + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    EpiPhaseA pa{};
    if constexpr (FUSED && EPI == 1) {      // (EPI 2 is launched with one workgroup per row only)
        if (fused) {
            epi_phase_a_lds(ep, lane, __uint_as_float(raw_feat), wlds, reinterpret_cast<float*>(stage + 192), pa);
            uint32_t a0 = static_cast<uint32_t>(xv0), a1 = static_cast<uint32_t>(xv0 >> 32), g0 = static_cast<uint32_t>(gv0), g1 = static_cast<uint32_t>(gv0 >> 32);
            asm volatile("" : "+v"(pa.h), "+v"(pa.wd[0]), "+v"(pa.wd[1]), "+v"(pa.wd[2]), "+v"(pa.wd[3]), "+v"(pa.wd[4]), "+v"(pa.b1), "+v"(pa.w2), "+v"(pa.b2),
                         "+v"(a0), "+v"(a1), "+v"(g0), "+v"(g1));   // the probes' first use: behind phase A
            xv0 = (static_cast<uint64_t>(a1) << 32) | a0;
            gv0 = (static_cast<uint64_t>(g1) << 32) | g0;
            epi_decide_dma(ep, b, lane, dlds);      // the decision inputs -> LDS, under the polls below
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // phase A's scratch (stage + 192 ...) is dead: granules are staged below
            __builtin_amdgcn_wave_barrier();
        }
    }
    bool lost_x = false;
    if (lane < p.K) {
        const uint64_t xv = poll_more(xslots + lane, xv0);
        __hip_atomic_store(xslots + lane, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lost_x = xv == 0ull;
        x_tok = __uint_as_float(static_cast<uint32_t>(xv));
    }
    for (int g = lane; g < KS; g += 64) {
        stage[g] = g == lane ? poll_more(region + g, gv0) : poll_slot(region + g);       // 0 = lost: poisons its row below
        __hip_atomic_store(region + g, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // empty again for the next call
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float fm = kSentinel, fs = 0.0f;
    bool lost = lost_x;
    if (lane < p.K) {
#pragma unroll 4      // (the compiler's own choice, 8, takes the FUSED instantiation past 64 VGPRs: phase A's results are live here)
        for (int j = 0; j < S; ++j) {
            const uint64_t g = stage[lane * S + j];
            lost = lost || g == 0ull;
            ms_merge(fm, fs, __uint_as_float(static_cast<uint32_t>(g)), __uint_as_float(static_cast<uint32_t>(g >> 32)));
        }
        if (lost) { fm = NAN; fs = NAN; }       // a slice never arrived: lp_t = NaN, the row is rejected
    }
    if (__ballot(lost) != 0ull && lane == 0) report_lost(p.tickets);

    if (p.mode == 1) {
        if (lane < p.K) {
            p.msg[3 * frow + 0] = fm;
            p.msg[3 * frow + 1] = fs;
            p.msg[3 * frow + 2] = x_tok;
        }
        return;
    }
    bool flag = false;
    float lp = 0.0f;
    if (lane < p.K) {
        flag = finish_row(fm, fs, x_tok, c2, lpd, log_u(uu), lp);
        p.lp_t[frow] = lp;
        p.accept[frow] = flag ? 1 : 0;
        if (p.row_max_lp) p.row_max_lp[frow] = row_max_logprob(fs);
    }
    finish_sequence(flag, lane, p.K, b, p.n_acc, p.bits);
    if constexpr (FUSED && EPI == 1) {
        if (!fused) return;
        // the finisher's lanes hold all K lp_t: the predictor / stop epilogue runs right here (no hand-off at all)
        if (__ballot(lost) != 0ull) {
            if (lane == 0) epi_poison(ep, b);
            return;
        }
        epi_finish_lds(ep, b, lane, lp, p.K, ep.stats_col >= 0 || ep.stats != nullptr, dlds, pa);
        ASD_STAMP_AT(8);
    }
}

// combine all-gathered per-shard partials; one wave per sequence
__global__ __launch_bounds__(64) void k_accept_from_partials(const float* msg_all, int n_shards, const float* lp_d,
                                                             const float* u, int B, int K, float c2, float* lp_t,
                                                             uint8_t* accept, int32_t* n_acc, uint64_t* bits) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    bool flag = false;
    if (lane < K) {
        const int row = b * K + lane;
        float m2 = kSentinel, s = 0.0f, g = -INFINITY;
        for (int r = 0; r < n_shards; ++r) {
            const float* t = msg_all + (static_cast<int64_t>(r) * B * K + row) * 3;
            ms_merge(m2, s, t[0], t[1]);
            const float gr = t[2];
            g = (gr != gr) ? gr : fmaxf(g, gr);  // a NaN logit must not be dropped by max
        }
        float lp;
        flag = finish_row(m2, s, g, c2, lp_d[row], log_u(u[row]), lp);
        lp_t[row] = lp;
        accept[row] = flag ? 1 : 0;
    }
    finish_sequence(flag, lane, K, b, n_acc, bits);
}

inline int max_splits_for(int K) {
    int s = kMaxStage / (K > 0 ? K : 1);
    if (s > ASD_MAX_SPLITS) s = ASD_MAX_SPLITS;
    return s < 1 ? 1 : s;
}

struct Geometry {
    int splits, threads, unroll, nt;
};

// Launch geometry (numbers: profiles/r02_sweep_*.json, MI355X, 256 CUs, dynamic-tile kernel, preloaded arguments).
//   rows >= CUs : one workgroup per row, no split: 512 lanes x 3-KiB tiles (B=32: 14.8-14.9 us = 5.24 TB/s against 15.0 for
//                 2- and 4-KiB tiles and 15.3 for 1024 lanes x 2 KiB; B=128: 48.1 us = 6.47 TB/s).  A 304 KB row is 101.4 tiles
//                 of 3 KiB: the eight waves of the workgroup end within one tile of each other, and a 3-KiB tile is a third
//                 less to drain at the end than a 4-KiB one, while two tiles per wave still keep 48 KiB per CU in flight.
//   slices (round 4: ONE rule for every row count, fitted to forced-slice sweeps at 22 row counts from 32 to 1040 rows,
//                 profiles/r04_sweep_small_batch_splits.json, r04_sweep_batch_splits{,2}.json; it picks the measured best or a count
//                 within 1.3 % of it in all of them): the rows are cut into the S slices that minimise
//                     est(S) = a * max(ceil(R S / CUs) / S, 1.05 R / CUs)  +  0.2 us * R S / CUs ,   a = 6 us per 304-KB row
//                 -- the bytes of the fullest CU (in rows), floored by the chip-wide HBM time, plus a fixed cost per workgroup a CU
//                 runs (ramp, granule hand-off).  With one workgroup per row the CUs that hold one row more than the others finish
//                 alone, and the few workgroups left cannot keep the HBM busy (B = 33: 21.7 us against 15.5 at B = 32; B = 65: 33.4
//                 against 26.6); few rows leave CUs idle.  Full rounds (B = 32, 64, 96, 128: R a multiple of the CUs) stay whole.
//                 A slice is never cut below one tile per wave.  Geometry: whole rows 512 lanes x 3 KiB; slices of fewer rows than
//                 CUs 512 x 2 KiB (B = 8: S = 4), of more 512 x 3 KiB.
Geometry choose_geometry(int R, int K, int V, int dtype, int cus, bool whole_rows_only = false) {
    Geometry g{1, 512, 3, 1};
    if (R <= 0 || cus <= 0 || whole_rows_only) return g;
    // full rounds; a last round that fills >= 60 % of the CUs still saturates the HBM on its own (B = 52, 56: whole rows measured
    // best, the estimate's one systematic miss); beyond six rounds the dispatcher balances (unmeasured: left whole)
    if (R >= cus && (R % cus == 0 || 10 * (R % cus) >= 6 * cus || R / cus >= 6)) return g;
    const int64_t row_bytes = static_cast<int64_t>(V) * dtype_size(dtype);
    const int unroll_sliced = R >= cus ? 3 : 2;
    int64_t cap = row_bytes / (static_cast<int64_t>(g.threads / 64) * unroll_sliced * 1024);
    if (cap > max_splits_for(K)) cap = max_splits_for(K);
    if (cap > (R >= cus ? 8 : 16)) cap = R >= cus ? 8 : 16;
    if (cap < 1) cap = 1;
    const double a = 6.0 * static_cast<double>(row_bytes) / 304128.0, q = static_cast<double>(R) / cus;
    int best = 1;
    double best_est = 0.0;
    for (int S = 1; S <= cap; ++S) {
        const int64_t wgs = static_cast<int64_t>(R) * S;
        const double load = static_cast<double>((wgs + cus - 1) / cus) / S;
        const double est = a * (load > 1.05 * q ? load : 1.05 * q) + 0.2 * q * S;
        if (S == 1 || est < best_est - 1e-9) { best = S; best_est = est; }
    }
    g.splits = best;
    if (best > 1) g.unroll = unroll_sliced;
    return g;
}

inline int own_row_of(const VerifyParams& p) { return (p.S == 1 && (p.mode == 1 || p.K <= kFastMaxK)) ? 1 : 0; }

template <int DT, int THREADS, int UNROLL>
void launch_nt(const VerifyParams& p, int64_t grid, hipStream_t st, int nt) {
    if (nt)
        hipLaunchKernelGGL((k_verify<DT, THREADS, UNROLL, true, false>), dim3(static_cast<uint32_t>(grid / p.S), static_cast<uint32_t>(p.S)), dim3(THREADS), 0, st,
                           p.logits, p.tok, p.ld_row, p.V, p.K, p.S, p.scale2, own_row_of(p), p);
    else
        hipLaunchKernelGGL((k_verify<DT, THREADS, UNROLL, false, false>), dim3(static_cast<uint32_t>(grid / p.S), static_cast<uint32_t>(p.S)), dim3(THREADS), 0, st,
                           p.logits, p.tok, p.ld_row, p.V, p.K, p.S, p.scale2, own_row_of(p), p);
}

template <int DT, int THREADS>
int launch_unroll(const VerifyParams& p, int64_t grid, hipStream_t st, const Geometry& g) {
    switch (g.unroll) {
        case 2: launch_nt<DT, THREADS, 2>(p, grid, st, g.nt); return ASD_OK;
        case 3: launch_nt<DT, THREADS, 3>(p, grid, st, g.nt); return ASD_OK;
        case 4: launch_nt<DT, THREADS, 4>(p, grid, st, g.nt); return ASD_OK;
        case 8: launch_nt<DT, THREADS, 8>(p, grid, st, g.nt); return ASD_OK;
        default: return ASD_ERR_UNSUPPORTED;
    }
}

template <int DT>
int launch_threads(const VerifyParams& p, int64_t grid, hipStream_t st, const Geometry& g) {
    switch (g.threads) {
        case 256: return launch_unroll<DT, 256>(p, grid, st, g);
        case 512: return launch_unroll<DT, 512>(p, grid, st, g);
        case 1024: return launch_unroll<DT, 1024>(p, grid, st, g);
        default: return ASD_ERR_UNSUPPORTED;
    }
}

#ifdef ASD_TEST_HOOKS        // (process-global, not thread-safe: the TEST build of the library only)
int g_debug_withhold = -1;   // asd_debug_verify_withhold: fault injection for the hand-off tests
#else
constexpr int g_debug_withhold = -1;
#endif

int launch_verify(VerifyParams p, int dtype, void* workspace, size_t workspace_bytes, void* stream, Geometry g) {
    p.withhold1 = g_debug_withhold + 1;
    if (p.B < 0 || p.K < 0 || p.V < 0) return ASD_ERR_INVALID_ARG;
    if (p.B == 0 || p.K == 0) return ASD_OK;
    if (p.K > ASD_MAX_DRAFT_LEN) return ASD_ERR_UNSUPPORTED;
    const int esz = dtype_size(dtype);
    if (esz == 0) return ASD_ERR_UNSUPPORTED;
    if (!p.logits || !p.tok || !workspace) return ASD_ERR_INVALID_ARG;
    if (p.ld_row < p.V) return ASD_ERR_INVALID_ARG;
    if (!aligned_to(p.logits, static_cast<size_t>(esz))) return ASD_ERR_ALIGNMENT;
    if (!aligned_to(workspace, 256)) return ASD_ERR_WORKSPACE;
    const int64_t R = static_cast<int64_t>(p.B) * p.K;
    if (R > INT32_MAX) return ASD_ERR_UNSUPPORTED;

    // (the (m2, s, t) instantiation and the eight-wave 256 x 128 epilogue exist for one workgroup per row only)
    const Geometry h = choose_geometry(static_cast<int>(R), p.K, p.V, dtype, current_device_cus(), p.row_entropy != nullptr || p.fused == 2);
    const bool auto_unroll = g.unroll <= 0, auto_splits = g.splits <= 0;
    if (g.splits <= 0) g.splits = h.splits;
    if (g.threads <= 0) g.threads = h.threads;
    if (g.unroll <= 0) g.unroll = h.unroll;
    if (g.nt < 0) g.nt = h.nt;
    if (g.splits > max_splits_for(p.K)) return ASD_ERR_UNSUPPORTED;
    // every tile of a slice owns an LDS slot: tiles + 1 <= kMaxStage (a 152064-wide bf16 row is 149 tiles)
    for (;;) {
        const int64_t slice_bytes = (static_cast<int64_t>(p.V) * esz + g.splits - 1) / g.splits + 16;
        const int64_t tiles = (slice_bytes + g.unroll * 1024 - 1) / (g.unroll * 1024);
        if (tiles + 1 <= kMaxStage) break;
        if (auto_unroll && g.unroll < 8) { g.unroll *= 2; continue; }
        if (auto_splits && g.splits * 2 <= max_splits_for(p.K)) { g.splits *= 2; continue; }
        return ASD_ERR_UNSUPPORTED;
    }

    const size_t ticket_bytes = round_up(static_cast<size_t>(p.B) * kTicketStride * sizeof(uint32_t), 256);
    const size_t region_bytes = round_up(static_cast<size_t>(p.K) * g.splits * sizeof(uint64_t), 256);
    if (workspace_bytes < ticket_bytes + region_bytes * static_cast<size_t>(p.B)) return ASD_ERR_WORKSPACE;
    p.S = g.splits;
    p.tickets = static_cast<uint32_t*>(workspace);
    p.granules = reinterpret_cast<uint64_t*>(static_cast<char*>(workspace) + ticket_bytes);
    p.region = static_cast<uint32_t>(region_bytes / sizeof(uint64_t));

    const int64_t grid = R * g.splits;
    if (grid > INT32_MAX || static_cast<int64_t>(p.V) * g.splits > INT32_MAX) return ASD_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
    if (p.row_entropy) {   // the (m2, s, t) instantiation exists for one workgroup per row, 512 lanes x 3-KiB tiles
        if (g.splits != 1 || g.threads != 512 || g.unroll != 3 || p.K > kFastMaxK || p.fused || p.mode != 0) return ASD_ERR_UNSUPPORTED;
        const dim3 gd(static_cast<uint32_t>(grid / p.S), static_cast<uint32_t>(p.S));
#define ASD_LAUNCH_STATS(DT)                                                                                      \
    hipLaunchKernelGGL((k_verify<DT, 512, 3, true, false, true>), gd, dim3(512), 0, st, p.logits, p.tok, p.ld_row, p.V, \
                       p.K, p.S, p.scale2, own_row_of(p), p)
        switch (dtype) {
            case ASD_DTYPE_BF16: ASD_LAUNCH_STATS(ASD_DTYPE_BF16); break;
            case ASD_DTYPE_F16: ASD_LAUNCH_STATS(ASD_DTYPE_F16); break;
            default: ASD_LAUNCH_STATS(ASD_DTYPE_F32); break;
        }
#undef ASD_LAUNCH_STATS
        return launch_status();
    }
    if (p.fused) {   // the in-kernel epilogue is instantiated for the two geometries the heuristic uses: rows >= CUs
                     // (512 lanes x 3-KiB tiles, one workgroup per row) and rows < CUs (512 x 2 KiB, split rows)
        const dim3 gd(static_cast<uint32_t>(grid / p.S), static_cast<uint32_t>(p.S));
        const bool wide = (g.threads == 512 && g.unroll == 3);
        if (!wide && !(g.threads == 512 && g.unroll == 2)) return ASD_ERR_UNSUPPORTED;
        if (p.fused == 2) {   // the 256 -> 128 -> 1 epilogue: one workgroup per row, 512 lanes x 3-KiB tiles only
            if (g.splits != 1 || !wide || !own_row_of(p)) return ASD_ERR_UNSUPPORTED;
#define ASD_LAUNCH_FUSED2(DT)                                                                                     \
    hipLaunchKernelGGL((k_verify<DT, 512, 3, true, true, false, 2>), gd, dim3(512), 0, st, p.logits, p.tok, p.ld_row, p.V, \
                       p.K, p.S, p.scale2, own_row_of(p), p)
            switch (dtype) {
                case ASD_DTYPE_BF16: ASD_LAUNCH_FUSED2(ASD_DTYPE_BF16); break;
                case ASD_DTYPE_F16: ASD_LAUNCH_FUSED2(ASD_DTYPE_F16); break;
                default: ASD_LAUNCH_FUSED2(ASD_DTYPE_F32); break;
            }
#undef ASD_LAUNCH_FUSED2
            return launch_status();
        }
#define ASD_LAUNCH_FUSED(DT)                                                                                      \
    do {                                                                                                          \
        if (wide) hipLaunchKernelGGL((k_verify<DT, 512, 3, true, true>), gd, dim3(512), 0, st, p.logits, p.tok, p.ld_row, \
                                     p.V, p.K, p.S, p.scale2, own_row_of(p), p);                                  \
        else hipLaunchKernelGGL((k_verify<DT, 512, 2, true, true>), gd, dim3(512), 0, st, p.logits, p.tok, p.ld_row,  \
                                p.V, p.K, p.S, p.scale2, own_row_of(p), p);                                       \
    } while (0)
        switch (dtype) {
            case ASD_DTYPE_BF16: ASD_LAUNCH_FUSED(ASD_DTYPE_BF16); break;
            case ASD_DTYPE_F16: ASD_LAUNCH_FUSED(ASD_DTYPE_F16); break;
            default: ASD_LAUNCH_FUSED(ASD_DTYPE_F32); break;
        }
#undef ASD_LAUNCH_FUSED
        return launch_status();
    }
    switch (dtype) {
        case ASD_DTYPE_BF16: rc = launch_threads<ASD_DTYPE_BF16>(p, grid, st, g); break;
        case ASD_DTYPE_F16: rc = launch_threads<ASD_DTYPE_F16>(p, grid, st, g); break;
        default: rc = launch_threads<ASD_DTYPE_F32>(p, grid, st, g); break;
    }
    if (rc != ASD_OK) return rc;
    return launch_status();
}

}  // namespace
}  // namespace asd

using namespace asd;

#ifdef ASD_STAMP
ASD_EXPORT int asd_debug_set_stamp_buffer(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_asd_stamps), &buf, sizeof(buf)) == hipSuccess ? ASD_OK : ASD_ERR_HIP;
}
#endif

ASD_EXPORT size_t asd_verify_accept_workspace_bytes(int B, int K, int V, int dtype) {
    (void)V;
    (void)dtype;
    if (B <= 0 || K <= 0) return 256;
    const size_t ticket_bytes = round_up(static_cast<size_t>(B) * kTicketStride * sizeof(uint32_t), 256);
    const size_t region_bytes = round_up(static_cast<size_t>(K) * max_splits_for(K) * sizeof(uint64_t), 256);
    return ticket_bytes + region_bytes * static_cast<size_t>(B);
}

ASD_EXPORT int asd_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace) return ASD_ERR_INVALID_ARG;
    if (!aligned_to(workspace, 256)) return ASD_ERR_WORKSPACE;
    if (hipMemsetAsync(workspace, 0, workspace_bytes, static_cast<hipStream_t>(stream)) != hipSuccess) return ASD_ERR_HIP;
    return ASD_OK;
}

ASD_EXPORT int asd_workspace_status(const void* workspace, uint32_t* status_host, void* stream) {
    if (!workspace || !status_host) return ASD_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(status_host, workspace, sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess) return ASD_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return ASD_ERR_HIP;
    return ASD_OK;
}

namespace {
// options -> (scale2, geometry); NULL = defaults
int unpack_options(const asd_verify_options* opt, float& scale2, asd::Geometry& g) {
    g = asd::Geometry{0, 0, 0, -1};
    double inv_t = 1.0;
    if (opt) {
        if (!(opt->inv_temperature > 0.0f) || !(opt->inv_temperature < 3.0e38f)) return ASD_ERR_INVALID_ARG;
        inv_t = static_cast<double>(opt->inv_temperature);
        g = asd::Geometry{opt->splits, opt->threads, opt->unroll, opt->nontemporal};
    }
    scale2 = static_cast<float>(1.4426950408889634074 * inv_t);
    return ASD_OK;
}
}  // namespace

ASD_EXPORT int asd_verify_accept_ex(const void* logits, int dtype, int64_t ld_row, const int32_t* tok,
                                    const float* lp_draft, const float* u, int B, int K, int V, float* lp_target,
                                    uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits, void* workspace,
                                    size_t workspace_bytes, const asd_verify_options* opt, void* stream) {
    if (B > 0 && K > 0 && (!lp_draft || !u || !lp_target || !accept || !n_acc)) return ASD_ERR_INVALID_ARG;
    VerifyParams p{};
    Geometry g;
    const int rc = unpack_options(opt, p.scale2, g);
    if (rc != ASD_OK) return rc;
    p.logits = logits; p.ld_row = ld_row; p.tok = tok; p.lp_d = lp_draft; p.u = u;
    p.B = B; p.K = K; p.V = V; p.v_offset = 0;
    p.lp_t = lp_target; p.accept = accept; p.n_acc = n_acc; p.bits = accept_bits;
    p.msg = nullptr; p.mode = 0;
    return launch_verify(p, dtype, workspace, workspace_bytes, stream, g);
}

ASD_EXPORT int asd_verify_accept_stats(const void* logits, int dtype, int64_t ld_row, const int32_t* tok,
                                       const float* lp_draft, const float* u, int B, int K, int V, float* lp_target,
                                       uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits, float* row_max_lp,
                                       float* row_entropy, void* workspace, size_t workspace_bytes,
                                       float inv_temperature, void* stream) {
    if (B > 0 && K > 0 && (!lp_draft || !u || !lp_target || !accept || !n_acc)) return ASD_ERR_INVALID_ARG;
    asd_verify_options opt{inv_temperature, 0, 0, 0, -1};
    VerifyParams p{};
    Geometry g;
    const int rc = unpack_options(&opt, p.scale2, g);
    if (rc != ASD_OK) return rc;
    if (row_entropy) g = Geometry{1, 512, 3, 1};   // one workgroup per row whatever the batch: the entropy's third sum is not handed across slices
    p.logits = logits; p.ld_row = ld_row; p.tok = tok; p.lp_d = lp_draft; p.u = u;
    p.B = B; p.K = K; p.V = V; p.v_offset = 0;
    p.lp_t = lp_target; p.accept = accept; p.n_acc = n_acc; p.bits = accept_bits;
    p.msg = nullptr; p.mode = 0;
    p.row_max_lp = row_max_lp; p.row_entropy = row_entropy;
    return launch_verify(p, dtype, workspace, workspace_bytes, stream, g);
}

ASD_EXPORT int asd_verify_accept_fused_ex(const void* logits, int dtype, int64_t ld_row, const int32_t* tok,
                                          const float* lp_draft, const float* u, int B, int K, int V, float* lp_target,
                                          uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits, void* workspace,
                                          size_t workspace_bytes, const float* feat, int64_t ldf, int stats_col,
                                          const float* packed_w, int in_dim, int hidden, int risk_adjustment, int64_t n_obs,
                                          double alpha, double beta, double* p_hist, const double* C, double lam, int L,
                                          int stage_idx, int prefix_rule, const double* theta, float* score,
                                          int32_t* k_star, uint8_t* stop, uint8_t* thr_stop, double* stats,
                                          const asd_verify_options* opt, void* stream) {
    if (B > 0 && K > 0 && (!lp_draft || !u || !lp_target || !accept || !n_acc)) return ASD_ERR_INVALID_ARG;
    if (L < 1 || stage_idx < 0 || stage_idx >= L) return ASD_ERR_INVALID_ARG;
    if (L > ASD_MAX_STAGES) return ASD_ERR_UNSUPPORTED;
    if (B > 0 && (!feat || !packed_w || ldf < in_dim)) return ASD_ERR_INVALID_ARG;
    if (stats_col >= 0 && stats_col + ASD_NUM_LP_STATS > in_dim) return ASD_ERR_INVALID_ARG;
    if ((k_star || stop) && (!p_hist || !C)) return ASD_ERR_INVALID_ARG;
    VerifyParams p{};
    Geometry g;
    const int rc0 = unpack_options(opt, p.scale2, g);
    if (rc0 != ASD_OK) return rc0;
    // the in-kernel epilogue is the reference's 64 -> 32 -> 1 predictor over a hierarchy of <= 4 tiers (the reference's: 3 or 4),
    // draft lengths <= 16, at the heuristic's geometries; any other predictor shape, depth, draft length or forced geometry gets the
    // same results from two launches
    const bool geometry_ok = (g.threads == 0 || g.threads == 512) && (g.unroll == 0 || g.unroll == 2 || g.unroll == 3);
    const bool in_kernel = in_dim == 64 && hidden == 32 && geometry_ok && L <= kDecidePrefetch && K <= kEpiInKernelMaxK;
    // ... and, with one workgroup per row (rows >= CUs, nothing forced), the 256 -> 128 -> 1 predictor of the reference's server
    // (src/serving/server.py:168): its first layer is cut over the eight waves of the finisher's workgroup
    const bool default_geometry = !opt || (opt->splits <= 0 && opt->threads <= 0 && opt->unroll <= 0);
    // -- where the heuristic keeps whole rows: a row count it would cut into slices takes the two-launch route, so that the
    // results stay those of asd_verify_accept_ex at the same arguments bit for bit
    const bool in_kernel2 = in_dim == kEpi2In && hidden == kEpi2Hid && default_geometry && L <= kDecidePrefetch && K <= kEpiInKernelMaxK &&
                            static_cast<int64_t>(B) * K >= current_device_cus() && static_cast<int64_t>(B) * K <= INT32_MAX &&
                            choose_geometry(static_cast<int>(static_cast<int64_t>(B) * K), K, V, dtype, current_device_cus()).splits == 1;
    if (!in_kernel && !in_kernel2) {
        const int rc = asd_verify_accept_ex(logits, dtype, ld_row, tok, lp_draft, u, B, K, V, lp_target, accept, n_acc,
                                            accept_bits, workspace, workspace_bytes, opt, stream);
        if (rc != ASD_OK) return rc;
        return asd_predictor_stop(lp_target, K, nullptr, K, feat, ldf, stats_col, packed_w, in_dim, hidden,
                                  risk_adjustment, n_obs, alpha, beta, p_hist, C, lam, L, stage_idx, prefix_rule, theta, B,
                                  score, k_star, stop, thr_stop, stats, stream);
    }
    p.logits = logits; p.ld_row = ld_row; p.tok = tok; p.lp_d = lp_draft; p.u = u;
    p.B = B; p.K = K; p.V = V; p.v_offset = 0;
    p.lp_t = lp_target; p.accept = accept; p.n_acc = n_acc; p.bits = accept_bits;
    p.msg = nullptr; p.mode = 0; p.fused = in_kernel2 ? 2 : 1;
    FusedParams& e = p.epi;
    e.lp = nullptr; e.ld_lp = K; e.n_valid = nullptr; e.K = K;
    e.feat = feat; e.ldf = ldf; e.stats_col = stats_col;
    e.packed = packed_w; e.in_dim = in_dim; e.hidden = hidden; e.use_lds = 0;
    e.risk = risk_adjustment ? 1 : 0; e.n_obs = static_cast<double>(n_obs); e.alpha = alpha; e.beta = beta;
    e.p_hist = p_hist; e.C = C; e.lam = lam; e.L = L; e.stage_idx = stage_idx; e.prefix = prefix_rule ? 1 : 0;
    e.theta = theta; e.B = B;
    e.score = score; e.k_star = k_star; e.stop = stop; e.thr_stop = thr_stop; e.stats = stats;
    return launch_verify(p, dtype, workspace, workspace_bytes, stream, g);
}

ASD_EXPORT int asd_verify_accept_fused(const void* logits, int dtype, int64_t ld_row, const int32_t* tok,
                                       const float* lp_draft, const float* u, int B, int K, int V, float* lp_target,
                                       uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits, void* workspace,
                                       size_t workspace_bytes, const float* feat, int64_t ldf, int stats_col,
                                       const float* packed_w, int in_dim, int hidden, int risk_adjustment, int64_t n_obs,
                                       double alpha, double beta, double* p_hist, const double* C, double lam, int L,
                                       int stage_idx, int prefix_rule, const double* theta, float* score,
                                       int32_t* k_star, uint8_t* stop, uint8_t* thr_stop, double* stats, void* stream) {
    return asd_verify_accept_fused_ex(logits, dtype, ld_row, tok, lp_draft, u, B, K, V, lp_target, accept, n_acc,
                                      accept_bits, workspace, workspace_bytes, feat, ldf, stats_col, packed_w, in_dim,
                                      hidden, risk_adjustment, n_obs, alpha, beta, p_hist, C, lam, L, stage_idx,
                                      prefix_rule, theta, score, k_star, stop, thr_stop, stats, nullptr, stream);
}

#ifdef ASD_TEST_HOOKS
/* tests only: the workgroup with linear index row * S + split (S = splits of the launch) does not publish its hand-off
 * slot in the following launches (its row / sequence must come back poisoned, never silently wrong); -1 = off. */
ASD_EXPORT int asd_debug_verify_withhold(int index) {
    asd::g_debug_withhold = index < 0 ? -1 : index;
    return ASD_OK;
}
#endif

ASD_EXPORT int asd_verify_accept(const void* logits, int dtype, int64_t ld_row, const int32_t* tok,
                                 const float* lp_draft, const float* u, int B, int K, int V, float* lp_target,
                                 uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    return asd_verify_accept_ex(logits, dtype, ld_row, tok, lp_draft, u, B, K, V, lp_target, accept, n_acc, accept_bits,
                                workspace, workspace_bytes, nullptr, stream);
}

ASD_EXPORT int asd_lse_partial(const void* logits_shard, int dtype, int64_t ld_row, const int32_t* tok, int B, int K,
                               int V_shard, int64_t v_offset, float inv_temperature, float* msg, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (B > 0 && K > 0 && !msg) return ASD_ERR_INVALID_ARG;
    if (!(inv_temperature > 0.0f) || !(inv_temperature < 3.0e38f)) return ASD_ERR_INVALID_ARG;
    VerifyParams p{};
    p.scale2 = static_cast<float>(1.4426950408889634074 * static_cast<double>(inv_temperature));
    p.logits = logits_shard; p.ld_row = ld_row; p.tok = tok;
    p.B = B; p.K = K; p.V = V_shard; p.v_offset = v_offset;
    p.msg = msg; p.mode = 1;
    return launch_verify(p, dtype, workspace, workspace_bytes, stream, Geometry{0, 0, 0, -1});
}

ASD_EXPORT int asd_accept_from_partials(const float* msg_all, int n_shards, const float* lp_draft, const float* u,
                                        int B, int K, float inv_temperature, float* lp_target, uint8_t* accept,
                                        int32_t* n_acc, uint64_t* accept_bits, void* stream) {
    if (B < 0 || K < 0 || n_shards < 1) return ASD_ERR_INVALID_ARG;
    if (!(inv_temperature > 0.0f) || !(inv_temperature < 3.0e38f)) return ASD_ERR_INVALID_ARG;
    const float c2 = static_cast<float>(1.4426950408889634074 * static_cast<double>(inv_temperature));
    if (B == 0 || K == 0) return ASD_OK;
    if (K > ASD_MAX_DRAFT_LEN) return ASD_ERR_UNSUPPORTED;
    if (!msg_all || !lp_draft || !u || !lp_target || !accept || !n_acc) return ASD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_accept_from_partials, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), msg_all,
                       n_shards, lp_draft, u, B, K, c2, lp_target, accept, n_acc, accept_bits);
    return launch_status();
}
