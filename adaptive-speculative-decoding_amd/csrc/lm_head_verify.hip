// lm_head_verify.hip -- SURVEY §8(f) N2: the target tier's lm_head projection fused with the verify
// pass, so that the [B, K, V] logits never reach HBM.
//
//   logits[m][v] = sum_k hidden[m][k] * weight[v][k]          (bf16 x bf16 -> f32, m = b*K + k)
//
// k_lm_head_partials: one workgroup owns 128 vocabulary columns x up to 256 rows.  It keeps the
// 128 x 256 logit tile in MFMA accumulators (v_mfma_f32_32x32x16_bf16, the WEIGHT tile as the A operand
// so that a lane holds 16 vocabulary entries of ONE row and the row reduction is lane-local), folds it
// into the log2-domain partial (m2, s) of lse_device.hpp, gathers logit[tok] where the block owns it and
// writes the (m2, s, g) triple that asd_lse_partial emits for a vocabulary shard -- here a "shard" is a
// 128-column block.  k_accept_from_blocks merges the n_blocks triples of every row and applies the
// accept rule (finish_row / finish_sequence: the same code the streaming kernel ends in).
//
// The weight matrix is streamed from HBM exactly once per 256 rows (nt loads), the hidden states are
// re-read by every block out of L2 / MALL.  Algorithmic HBM bytes: V*D*2 + M*D*2; flops: 2*M*D*V.
#include "lse_device.hpp"

namespace asd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 256;                 // rows (draft positions) per workgroup: 4 waves x 64
constexpr int kBN = 128;                 // vocabulary columns per workgroup
constexpr int kBK = 32;                  // reduction depth per stage (two 32x32x16 k-steps)
constexpr int kRowBytes = 80;            // LDS row pitch: 64 data bytes + 16 pad (conflict-free b128 fragment reads)
constexpr int kHBytes = kBM * kRowBytes;
constexpr int kWBytes = kBN * kRowBytes;
constexpr int kBufBytes = kHBytes + kWBytes;   // 30720; two stages = 61440 bytes of LDS

struct LmHeadParams {
    const void* hidden;
    int64_t ld_h;
    const void* weight;
    int64_t ld_w;
    int D, M, V;
    const int32_t* tok;
    float c2;
    float* msg;          // [n_blocks][M][3]
    int m_blocks;
};

__global__ __launch_bounds__(256, 2) void k_lm_head_partials(LmHeadParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kBufBytes];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;
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

    // staging map: four lanes cover the 64 bytes one row contributes to a stage
    const int srow = t >> 2;
    const int seg = t & 3;
    const uint32_t g_h = static_cast<uint32_t>(srow) * static_cast<uint32_t>(p.ld_h) * 2u + seg * 16u;
    const uint32_t g_w = static_cast<uint32_t>(srow) * static_cast<uint32_t>(p.ld_w) * 2u + seg * 16u;
    const uint32_t step_h = 64u * static_cast<uint32_t>(p.ld_h) * 2u;
    const uint32_t step_w = 64u * static_cast<uint32_t>(p.ld_w) * 2u;
    const int l_st = srow * kRowBytes + seg * 16;

    u32x4 hreg[4], wreg[2];
    auto gload = [&](int k0) {
        const uint32_t kb = static_cast<uint32_t>(k0) * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) hreg[i] = load16<false>(rsrc_h, g_h + i * step_h + kb);
#pragma unroll
        for (int i = 0; i < 2; ++i) wreg[i] = load16<true>(rsrc_w, g_w + i * step_w + kb);
    };
    auto lstore = [&](int buf) {
        unsigned char* hb = lds + buf * kBufBytes;
        unsigned char* wb = hb + kHBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(hb + l_st + i * 64 * kRowBytes) = hreg[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(wb + l_st + i * 64 * kRowBytes) = wreg[i];
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;

    const bool wave_has_rows = 64 * w < rows_h;   // wave-uniform: a wave whose 64 rows are all padding only stages
    const int frag_off = r * kRowBytes + h * 16;

    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = p.D / kBK;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * kBK);
        if (wave_has_rows) {
            const unsigned char* hb = lds + cur * kBufBytes + (64 * w) * kRowBytes + frag_off;
            const unsigned char* wb = lds + cur * kBufBytes + kHBytes + frag_off;
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
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }
    if (!wave_has_rows) return;

    // ---- epilogue: D[vocab row][m column]; lane (r, h) holds row m = .. + r and, per 32-column tile,
    // the 16 vocabulary ids  n = tile + (i & 3) + 8 * (i >> 2) + 4 * h
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + 64 * w + 32 * mt + r;
        const int tk = m < p.M ? p.tok[m] : -1;
        float m2 = kSentinel, s = 0.0f, g = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = n0 + 32 * nt + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float v = n < p.V ? acc[mt][nt][i] : -INFINITY;   // zero-padded weight rows are not vocabulary
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
            float* out = p.msg + (static_cast<int64_t>(nb) * p.M + m) * 3;
            out[0] = m2;
            out[1] = s;
            out[2] = g;
        }
    }
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

inline int n_blocks_for(int V) { return (V + kBN - 1) / kBN; }

}  // namespace
}  // namespace asd

using namespace asd;

ASD_EXPORT size_t asd_lm_head_verify_workspace_bytes(int B, int K, int V) {
    if (B <= 0 || K <= 0 || V <= 0) return 0;
    return round_up(static_cast<size_t>(n_blocks_for(V)) * static_cast<size_t>(B) * K * 3 * sizeof(float), 256);
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
    const int n_blocks = n_blocks_for(V);
    const int64_t m_blocks = (M + kBM - 1) / kBM;
    if (M >= (1ll << 31) || m_blocks * n_blocks >= (1ll << 31)) return ASD_ERR_UNSUPPORTED;
    LmHeadParams p{};
    p.hidden = hidden; p.ld_h = ld_h; p.weight = weight; p.ld_w = ld_w;
    p.D = D; p.M = static_cast<int>(M); p.V = V; p.tok = tok;
    p.c2 = static_cast<float>(1.4426950408889634074 * static_cast<double>(inv_temperature));
    p.msg = static_cast<float*>(workspace);
    p.m_blocks = static_cast<int>(m_blocks);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_lm_head_partials, dim3(static_cast<unsigned>(m_blocks * n_blocks)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_accept_from_blocks, dim3(B), dim3(256), 0, st, p.msg, n_blocks, lp_draft, u, B, K, p.c2,
                       lp_target, accept, n_acc, accept_bits);
    return launch_status();
}
