// lm_head_verify.hip -- SURVEY §8(f) N2: the target tier's lm_head projection fused with the verify
// pass, so that the [B, K, V] logits never reach HBM.
//
//   logits[m][v] = sum_k hidden[m][k] * weight[v][k]          (bf16 x bf16 -> f32, m = b*K + k)
//
// k_lm_head_partials: one workgroup (8 waves, one per CU) owns 256 vocabulary columns x up to 256 rows.
// It keeps the 256 x 256 logit tile in MFMA accumulators (v_mfma_f32_32x32x16_bf16, the WEIGHT tile as the
// A operand so that a lane holds 16 vocabulary entries of ONE row and the row reduction is lane-local),
// folds it into the log2-domain partial (m2, s) of lse_device.hpp, gathers logit[tok] where the block
// owns it and writes the (m2, s, g) triple that asd_lse_partial emits for a vocabulary shard -- here a
// "shard" is a 128-column unit (one wave column).  k_accept_from_blocks merges the units of every row and
// applies the accept rule (finish_row / finish_sequence: the same code the streaming kernel ends in).
//
// Pipeline: 32-deep stages, two LDS buffers, two register sets; the loads of stage k+3 are issued while
// stage k is multiplied and are written to LDS two iterations later, one barrier per stage.
//
// The weight matrix is streamed from HBM exactly once per 256 rows (nt loads), the hidden states are
// re-read by every block out of L2 / MALL.  Algorithmic HBM bytes: V*D*2 + M*D*2; flops: 2*M*D*V.
#include "lse_device.hpp"

#ifndef ASD_LMHEAD_W_AUX
#define ASD_LMHEAD_W_AUX 0
#endif
#ifndef ASD_LMHEAD_LAB
#define ASD_LMHEAD_LAB 0   // tools/lm_head_lab.py: 1 = no loads in the loop, 2 = no math (timing experiments)
#endif
#ifndef ASD_LMHEAD_KERNEL
#define ASD_LMHEAD_KERNEL k_lm_head_partials_dma
#endif

namespace asd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 256;                 // rows (draft positions) per workgroup: 4 wave rows x 64
constexpr int kBN = 256;                 // vocabulary columns per workgroup: 2 wave columns x 128
constexpr int kUnit = 128;               // columns behind one (m2, s, g) triple
constexpr int kBK = 32;                  // reduction depth per stage (two 32x32x16 k-steps)
constexpr int kThreads = 512;
constexpr int kRowBytes = 80;            // LDS row pitch: 64 data bytes + 16 pad (conflict-free b128 fragment reads)
constexpr int kHBytes = kBM * kRowBytes;
constexpr int kWBytes = kBN * kRowBytes;
constexpr int kBufBytes = kHBytes + kWBytes;   // 40960; two stages = 81920 bytes of LDS

struct LmHeadParams {
    const void* hidden;
    int64_t ld_h;
    const void* weight;
    int64_t ld_w;
    int D, M, V;
    const int32_t* tok;
    float c2;
    float* msg;          // [n_units][M][3]
    int m_blocks;
};

struct StageRegs {
    u32x4 h[2], w[2];
};

// D[vocab row][m column]: lane (r, h) holds row m = row0 + 32 * mt + r and, per 32-column tile, the 16
// vocabulary ids  n = tile + (i & 3) + 8 * (i >> 2) + 4 * h.  Folds a wave's 64 rows x 128 columns into one
// (m2, s, g) triple per row.
__device__ __forceinline__ void lm_head_epilogue(const LmHeadParams& p, const f32x16 (&acc)[2][4], int row0, int unit,
                                                 int u0, int r, int h) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = row0 + 32 * mt + r;
        const int tk = m < p.M ? p.tok[m] : -1;
        float m2 = kSentinel, s = 0.0f, g = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = u0 + 32 * nt + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float v = n < p.V ? acc[mt][nt][i] : -INFINITY;   // padded weight rows are not vocabulary
                if (n == tk) g = v;
                x[i] = v;
            }
            float lo[8], hi[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { lo[i] = x[i]; hi[i] = x[8 + i]; }
            accum8(lo, p.c2, m2, s);
            accum8(hi, p.c2, m2, s);
        }
        // the row's other 64 columns sit in lane r ^ 32
        const float m2o = __shfl_xor(m2, 32, 64);
        const float so = __shfl_xor(s, 32, 64);
        const float go = __shfl_xor(g, 32, 64);
        ms_merge(m2, s, m2o, so);
        g = (g != g) ? g : ((go != go) ? go : fmaxf(g, go));   // a NaN logit must not be dropped by max
        if (h == 0 && m < p.M) {
            float* out = p.msg + (static_cast<int64_t>(unit) * p.M + m) * 3;
            out[0] = m2;
            out[1] = s;
            out[2] = g;
        }
    }
}

__global__ __launch_bounds__(kThreads, 1) void k_lm_head_partials(LmHeadParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kBufBytes];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = t >> 6;
    const int wm = wv & 3;     // wave row: rows 64 * wm ..
    const int wn = wv >> 2;    // wave column: columns 128 * wn ..
    const int r = lane & 31;
    const int h = lane >> 5;
    // consecutive workgroups share a weight tile (its re-read for M > 256 stays close in time)
    const int mb = static_cast<int>(blockIdx.x) % p.m_blocks;
    const int nb = static_cast<int>(blockIdx.x) / p.m_blocks;
    const int n0 = nb * kBN;
    const int m0 = mb * kBM;
    const int rows_w = min(kBN, p.V - n0);
    const int rows_h = min(kBM, p.M - m0);

    // per-block descriptors: rows past the matrix edge fall outside num_records and read as zero
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(p.weight)) + static_cast<int64_t>(n0) * p.ld_w * 2, 0,
        static_cast<int>((static_cast<int64_t>(rows_w - 1) * p.ld_w + p.D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(p.hidden)) + static_cast<int64_t>(m0) * p.ld_h * 2, 0,
        static_cast<int>((static_cast<int64_t>(rows_h - 1) * p.ld_h + p.D) * 2), 0x00020000);

    // staging map: four lanes cover the 64 bytes one row contributes to a stage; 128 rows per pass
    const int srow = t >> 2;
    const int seg = t & 3;
    const uint32_t g_h = static_cast<uint32_t>(srow) * static_cast<uint32_t>(p.ld_h) * 2u + seg * 16u;
    const uint32_t g_w = static_cast<uint32_t>(srow) * static_cast<uint32_t>(p.ld_w) * 2u + seg * 16u;
    const uint32_t step_h = 128u * static_cast<uint32_t>(p.ld_h) * 2u;
    const uint32_t step_w = 128u * static_cast<uint32_t>(p.ld_w) * 2u;
    const int l_st = srow * kRowBytes + seg * 16;

    auto gload = [&](StageRegs& q, int stage) {
        const uint32_t kb = static_cast<uint32_t>(stage) * (kBK * 2u);
#pragma unroll
        for (int i = 0; i < 2; ++i) q.h[i] = load16<false>(rsrc_h, g_h + i * step_h + kb);
#pragma unroll
        for (int i = 0; i < 2; ++i) q.w[i] = load16<true>(rsrc_w, g_w + i * step_w + kb);
    };
    auto lstore = [&](const StageRegs& q, int buf) {
        unsigned char* hb = lds + buf * kBufBytes;
        unsigned char* wb = hb + kHBytes;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(hb + l_st + i * 128 * kRowBytes) = q.h[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(wb + l_st + i * 128 * kRowBytes) = q.w[i];
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;

    // wave-uniform: a wave whose rows or columns are all padding only stages
    const bool wave_works = 64 * wm < rows_h && kUnit * wn < rows_w;
    const int frag_off = r * kRowBytes + h * 16;
    const int h_off = (64 * wm) * kRowBytes + frag_off;
    const int w_off = kHBytes + (kUnit * wn) * kRowBytes + frag_off;

    auto compute = [&](int buf) {
        if (!wave_works) return;
        const unsigned char* hb = lds + buf * kBufBytes + h_off;
        const unsigned char* wb = lds + buf * kBufBytes + w_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 hf[2], wf[4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                hf[mt] = *reinterpret_cast<const bf16x8*>(hb + mt * 32 * kRowBytes + ks * 32);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                wf[nt] = *reinterpret_cast<const bf16x8*>(wb + nt * 32 * kRowBytes + ks * 32);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], hf[mt], acc[mt][nt], 0, 0, 0);
        }
    };

    // Invariant at the top of iteration kt: LDS[kt & 1] holds stage kt; the register set "older" holds
    // stage kt + 1 (issued two iterations ago), "newer" holds stage kt + 2 (issued one iteration ago).
    // The loop body is branch-free on purpose: the compiler's counted vmcnt waits then leave the newer
    // set's four loads in flight while the older set is written (a conditional load collapses them to
    // vmcnt(0)).  Past the last stage the loads re-read stage nk - 1 (an L2 hit) and the store goes to the
    // buffer nobody reads again.
    const int nk = p.D / kBK;
    const int last = nk - 1;
    StageRegs q0, q1;
    gload(q0, 0);
    gload(q1, min(1, last));
    lstore(q0, 0);
    gload(q0, min(2, last));
    __syncthreads();
    auto iteration = [&](StageRegs& older, int kt) {
        lstore(older, (kt + 1) & 1);
        gload(older, min(kt + 3, last));
        compute(kt & 1);
        __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        iteration(q1, kt);
        iteration(q0, kt + 1);
    }
    if (kt < nk) iteration(q1, kt);
    if (!wave_works) return;
    lm_head_epilogue(p, acc, m0 + 64 * wm, nb * (kBN / kUnit) + wn, n0 + kUnit * wn, r, h);
}

// ---- LDS-DMA form: global_load_lds_dwordx4 into a 4-stage LDS ring, no staging registers -------------
// A stage is 32 reduction columns of the 256 hidden rows + 256 weight rows: 512 x 64 B = 32 KiB, unpadded
// (one wave-instruction writes 16 rows x 64 B contiguously).  Bank conflicts of the fragment reads are
// removed by an XOR swizzle of the 16-byte segment index with (row >> 2) & 3, applied to the per-lane SOURCE
// address and to the read address alike.  Three stages (96 KiB per CU) are in flight while one is multiplied;
// the counted vmcnt + barrier at the top of an iteration retires the stage about to be read, and the slot
// refilled right after it is the one every wave finished reading before that barrier.
constexpr int kRing = 4;
constexpr int kStageBytes = (kBM + kBN) * 64;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__global__ __launch_bounds__(kThreads, 1) void k_lm_head_partials_dma(LmHeadParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kRing * kStageBytes];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = t >> 6;
    const int wm = wv & 3;
    const int wn = wv >> 2;
    const int r = lane & 31;
    const int h = lane >> 5;
    const int mb = static_cast<int>(blockIdx.x) % p.m_blocks;
    const int nb = static_cast<int>(blockIdx.x) / p.m_blocks;
    const int n0 = nb * kBN;
    const int m0 = mb * kBM;
    const int rows_w = min(kBN, p.V - n0);
    const int rows_h = min(kBM, p.M - m0);

    // per-lane sources: pass ps covers rows ps * 128 + wv * 16 + (lane >> 2); rows past the edge re-read the
    // last valid row (their products are masked in the epilogue), so no lane ever leaves the matrices
    const int srow = wv * 16 + (lane >> 2);
    const int seg = (lane & 3) ^ ((lane >> 4) & 3);
    const char* hsrc[2];
    const char* wsrc[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int row = ps * 128 + srow;
        hsrc[ps] = static_cast<const char*>(p.hidden) + (static_cast<int64_t>(m0) + min(row, rows_h - 1)) * p.ld_h * 2 + seg * 16;
        wsrc[ps] = static_cast<const char*>(p.weight) + (static_cast<int64_t>(n0) + min(row, rows_w - 1)) * p.ld_w * 2 + seg * 16;
    }
    const int nk = p.D / kBK;
    const int last = nk - 1;
#if ASD_LMHEAD_LAB & 4
    // timing experiment: the same bytes per stage as whole 128-byte lines (128 rows x 128 B, row halves alternating)
    for (int ps = 0; ps < 2; ++ps) {
        const int row = ps * 64 + wv * 8 + (lane >> 3);
        hsrc[ps] = static_cast<const char*>(p.hidden) + (static_cast<int64_t>(m0) + min(row, rows_h - 1)) * p.ld_h * 2 + (lane & 7) * 16;
        wsrc[ps] = static_cast<const char*>(p.weight) + (static_cast<int64_t>(n0) + min(row, rows_w - 1)) * p.ld_w * 2 + (lane & 7) * 16;
    }
    const int64_t half_h = 128 * p.ld_h * 2, half_w = 128 * p.ld_w * 2;
#endif
    auto issue = [&](int stage) {
#if ASD_LMHEAD_LAB & 4
        const int st = min(stage, last);
        const int64_t kb = static_cast<int64_t>(st >> 1) * 128;
        unsigned char* dst = lds + (stage & (kRing - 1)) * kStageBytes + wv * 1024;
        const int64_t oh = (st & 1) * (rows_h > 128 ? half_h : 0), ow = (st & 1) * (rows_w > 128 ? half_w : 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(hsrc[0] + kb + oh), (lds_void*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(hsrc[1] + kb + oh), (lds_void*)(dst + 8192), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(wsrc[0] + kb + ow), (lds_void*)(dst + 16384), 16, 0, ASD_LMHEAD_W_AUX);
        __builtin_amdgcn_global_load_lds((glb_void*)(wsrc[1] + kb + ow), (lds_void*)(dst + 24576), 16, 0, ASD_LMHEAD_W_AUX);
#else
        const int64_t kb = static_cast<int64_t>(min(stage, last)) * (kBK * 2);
        unsigned char* dst = lds + (stage & (kRing - 1)) * kStageBytes + wv * 1024;
        __builtin_amdgcn_global_load_lds((glb_void*)(hsrc[0] + kb), (lds_void*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(hsrc[1] + kb), (lds_void*)(dst + 8192), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(wsrc[0] + kb), (lds_void*)(dst + 16384), 16, 0, ASD_LMHEAD_W_AUX);
        __builtin_amdgcn_global_load_lds((glb_void*)(wsrc[1] + kb), (lds_void*)(dst + 24576), 16, 0, ASD_LMHEAD_W_AUX);
#endif
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;

    const bool wave_works = 64 * wm < rows_h && kUnit * wn < rows_w;
    const int key = (r >> 2) & 3;
    const int h_off = (64 * wm + r) * 64;
    const int w_off = kBM * 64 + (kUnit * wn + r) * 64;
    auto compute = [&](int slot) {
        if (!wave_works) return;
        const unsigned char* hb = lds + slot * kStageBytes + h_off;
        const unsigned char* wb = lds + slot * kStageBytes + w_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = ((2 * ks + h) ^ key) * 16;
            bf16x8 hf[2], wf[4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) hf[mt] = *reinterpret_cast<const bf16x8*>(hb + mt * 32 * 64 + so);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(wb + nt * 32 * 64 + so);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], hf[mt], acc[mt][nt], 0, 0, 0);
        }
    };

#if ASD_LMHEAD_LAB & 8
    // timing experiment: stages issued in (even, odd) pairs so that both halves of a 128-byte line are requested together
    issue(0);
    issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        if ((kt & 1) == 0) {
            asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            issue(kt + 2);
            issue(kt + 3);
        } else {
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        }
#if !(ASD_LMHEAD_LAB & 2)
        compute(kt & (kRing - 1));
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!wave_works) return;
    lm_head_epilogue(p, acc, m0 + 64 * wm, nb * (kBN / kUnit) + wn, n0 + kUnit * wn, r, h);
    return;
#endif
    issue(0);
    issue(1);
    issue(2);
    for (int kt = 0; kt < nk; ++kt) {
        // 12 loads outstanding (stages kt, kt+1, kt+2): retire this wave's share of stage kt, then meet the others
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
#if !(ASD_LMHEAD_LAB & 1)
        issue(kt + 3);   // slot (kt - 1) & 3: every wave's reads of it completed before it reached the barrier
#endif
#if !(ASD_LMHEAD_LAB & 2)
        compute(kt & (kRing - 1));
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail loads must land before the LDS is released
    if (!wave_works) return;
    lm_head_epilogue(p, acc, m0 + 64 * wm, nb * (kBN / kUnit) + wn, n0 + kUnit * wn, r, h);
}

// merge the per-block triples of every row of sequence b, then the accept rule.  4 waves; wave w
// takes draft positions w, w + 4, ...; its lanes stride over the blocks (fixed order: deterministic).
__global__ __launch_bounds__(256) void k_accept_from_blocks(const float* msg, int n_blocks, const float* lp_d,
                                                            const float* u, int B, int K, float c2, float* lp_t,
                                                            uint8_t* accept, int32_t* n_acc, uint64_t* bits) {
    __shared__ float red[ASD_MAX_DRAFT_LEN][3];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t M = static_cast<int64_t>(B) * K;
    for (int k = w; k < K; k += 4) {
        const int64_t row = static_cast<int64_t>(b) * K + k;
        float m2 = kSentinel, s = 0.0f, g = -INFINITY;
        bool gnan = false;
        for (int j = lane; j < n_blocks; j += 64) {
            const float* t = msg + (static_cast<int64_t>(j) * M + row) * 3;
            ms_merge(m2, s, t[0], t[1]);
            const float gj = t[2];
            gnan = gnan || (gj != gj);
            g = fmaxf(g, gj);
        }
        wave_merge(m2, s);
        g = wave_max(g);
        if (__ballot(gnan) != 0ull) g = NAN;
        if (lane == 0) {
            red[k][0] = m2;
            red[k][1] = s;
            red[k][2] = g;
        }
    }
    __syncthreads();
    if (w != 0) return;
    bool flag = false;
    if (lane < K) {
        const int64_t row = static_cast<int64_t>(b) * K + lane;
        float lp;
        flag = finish_row(red[lane][0], red[lane][1], red[lane][2], c2, lp_d[row], log_u(u[row]), lp);
        lp_t[row] = lp;
        accept[row] = flag ? 1 : 0;
    }
    finish_sequence(flag, lane, K, b, n_acc, bits);
}

inline int n_units_for(int V) { return (V + kUnit - 1) / kUnit; }

}  // namespace
}  // namespace asd

using namespace asd;

ASD_EXPORT size_t asd_lm_head_verify_workspace_bytes(int B, int K, int V) {
    if (B <= 0 || K <= 0 || V <= 0) return 0;
    return round_up(static_cast<size_t>(n_units_for(V)) * static_cast<size_t>(B) * K * 3 * sizeof(float), 256);
}

ASD_EXPORT int asd_lm_head_verify(const void* hidden, int64_t ld_h, const void* weight, int64_t ld_w, int dtype, int D,
                                  const int32_t* tok, const float* lp_draft, const float* u, int B, int K, int V,
                                  float inv_temperature, float* lp_target, uint8_t* accept, int32_t* n_acc,
                                  uint64_t* accept_bits, void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || K < 0 || V < 1 || D < 1) return ASD_ERR_INVALID_ARG;
    if (!(inv_temperature > 0.0f) || !(inv_temperature < 3.0e38f)) return ASD_ERR_INVALID_ARG;
    if (B == 0 || K == 0) return ASD_OK;
    if (K > ASD_MAX_DRAFT_LEN) return ASD_ERR_UNSUPPORTED;
    if (dtype != ASD_DTYPE_BF16 || D % kBK != 0) return ASD_ERR_UNSUPPORTED;
    if (!hidden || !weight || !tok || !lp_draft || !u || !lp_target || !accept || !n_acc) return ASD_ERR_INVALID_ARG;
    if (ld_h < D || ld_w < D) return ASD_ERR_INVALID_ARG;
    if (!aligned_to(hidden, 16) || !aligned_to(weight, 16) || ld_h % 8 != 0 || ld_w % 8 != 0) return ASD_ERR_ALIGNMENT;
    // a block's descriptor spans at most 256 rows: its byte count must fit the 32-bit num_records field
    if ((static_cast<int64_t>(kBM) * ld_h + D) * 2 >= (1ll << 31) || (static_cast<int64_t>(kBN) * ld_w + D) * 2 >= (1ll << 31))
        return ASD_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < asd_lm_head_verify_workspace_bytes(B, K, V)) return ASD_ERR_WORKSPACE;
    if (!aligned_to(workspace, 16)) return ASD_ERR_ALIGNMENT;

    const int64_t M = static_cast<int64_t>(B) * K;
    const int n_units = n_units_for(V);
    const int n_blocks = (V + kBN - 1) / kBN;
    const int64_t m_blocks = (M + kBM - 1) / kBM;
    if (M >= (1ll << 31) || m_blocks * n_blocks >= (1ll << 31)) return ASD_ERR_UNSUPPORTED;
    LmHeadParams p{};
    p.hidden = hidden; p.ld_h = ld_h; p.weight = weight; p.ld_w = ld_w;
    p.D = D; p.M = static_cast<int>(M); p.V = V; p.tok = tok;
    p.c2 = static_cast<float>(1.4426950408889634074 * static_cast<double>(inv_temperature));
    p.msg = static_cast<float*>(workspace);
    p.m_blocks = static_cast<int>(m_blocks);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ASD_LMHEAD_KERNEL, dim3(static_cast<unsigned>(m_blocks * n_blocks)), dim3(kThreads), 0, st, p);
    hipLaunchKernelGGL(k_accept_from_blocks, dim3(B), dim3(256), 0, st, p.msg, n_units, lp_draft, u, B, K, p.c2,
                       lp_target, accept, n_acc, accept_bits);
    return launch_status();
}
