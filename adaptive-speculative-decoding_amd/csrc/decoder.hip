// decoder.hip -- X3 (the callers' side of the path): one pass of a Qwen2.5-shape decoder stack over the M = B * T
// positions the token-level loop feeds a tier, with a per-sequence (ragged) KV cache.  The reference delegates this to
// transformers / vLLM (src/serving/real_model_pipeline.py:135, src/models/stage.py); in the bench's loop it was ~1500 torch
// launches per pass of the 7B shape (15.5 ms even replayed from a hipGraph, 6.4 ms of them GEMMs).  Here a layer is NINE launches:
//
//   k_rmsnorm -> asd_linear (qkv, bias) -> k_rope_kv_store -> k_attn_ragged -> asd_linear (o, + residual, in place)
//   k_rmsnorm -> asd_linear (gate | up)  -> k_silu_mul     -> asd_linear (down, + residual, in place)
//
// and asd_decoder_forward issues all layers from ONE host call.  Where a projection's plan cuts the reduction into slices
// (narrow matrices: 14 column blocks cannot feed 256 CUs) the kernel that FOLLOWS it adds the f32 partials itself -- rope after
// q|k|v, the next norm (with the residual connection) after o and down, silu * up after gate|up -- so no reduce kernel runs.
//
// KV cache per layer: K as [rows][KVH][Tmax][128] and V TRANSPOSED, [rows][KVH][128][Tmax]: with the keys of a 32-key tile
// permuted (kappa below) both products of the attention take their cache operand straight from global memory in the MFMA
// operand layout -- 16 contiguous bytes per lane and k-step, no LDS, no transposed read -- and the score tile is handed to the
// second product in registers (accumulator -> operand, cdna_hip_programming.md "An accumulator tile as the next MFMA's operand").
#include "common.hpp"

namespace asd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHd = 128;            // head_dim of every Qwen2.5 shape

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf(float a, float b) {
    const __bf16 x = static_cast<__bf16>(a), y = static_cast<__bf16>(b);
    return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, x)) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, y)) << 16);
}

// A sliced asd_linear_partial leaves f32 partials [k_slices][M][N]; the kernels below can take their input from there instead
// of from the rounded [M][N] matrix: slices added in slice order, then the bias, then (where the layer has one) the residual,
// ONE rounding to bf16 -- the order and the rounding of asd_linear_ex's own reduce kernel, so either way gives the same bits.
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Slabs {
    const float* base;     // NULL: no slabs, read the matrix
    int k_slices;
    int64_t stride;        // M * N
    int N;
};
__device__ __forceinline__ void slab_sum8(const Slabs& z, int m, int n, float (&v)[8]) {
    const float* p = z.base + static_cast<int64_t>(m) * z.N + n;
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll 4
    for (int sl = 1; sl < z.k_slices; ++sl) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(p + sl * z.stride), d = *reinterpret_cast<const f32x4*>(p + sl * z.stride + 4);
        a[0] += c[0]; a[1] += c[1]; a[2] += c[2]; a[3] += c[3];
        b[0] += d[0]; b[1] += d[1]; b[2] += d[2]; b[3] += d[3];
    }
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// ---- RMSNorm: out = (x * rsqrt(mean(x^2) + eps) * w) in f32, one rounding.  One workgroup of 1024 threads per row, 4-element
// pieces (a row of 3584 is 896 pieces: one per thread), D <= 8192, D % 4 == 0.
// SLABS: first x[m] = round(sum of the slabs + x[m]) is formed and written back -- the o / down projection's residual connection,
// fused with the norm that follows it (each thread adds the k_slices partials of its piece).
__device__ __forceinline__ void slab_sum4(const Slabs& z, int m, int n, float (&v)[4]) {
    const float* p = z.base + static_cast<int64_t>(m) * z.N + n;
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll 4
    for (int sl = 1; sl < z.k_slices; ++sl) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(p + sl * z.stride);
        a[0] += c[0]; a[1] += c[1]; a[2] += c[2]; a[3] += c[3];
    }
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
template <bool SLABS>
__global__ __launch_bounds__(1024) void k_rmsnorm(char* x, int64_t ld_x, const char* __restrict__ w, float eps,
                                                  char* __restrict__ out, int64_t ld_out, int D, Slabs z) {
    __shared__ float part[16];
    const int t = threadIdx.x, m = blockIdx.x;
    const int pieces = D / 4;
    char* row = x + static_cast<int64_t>(m) * ld_x * 2;
    uint2 v[2];
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = t + 1024 * i;
        v[i] = uint2{0u, 0u};
        if (c < pieces) {
            v[i] = *reinterpret_cast<const uint2*>(row + c * 8);
            if constexpr (SLABS) {
                float p[4];
                slab_sum4(z, m, c * 4, p);
                v[i].x = pack_bf(p[0] + bf_lo(v[i].x), p[1] + bf_hi(v[i].x));
                v[i].y = pack_bf(p[2] + bf_lo(v[i].y), p[3] + bf_hi(v[i].y));
                *reinterpret_cast<uint2*>(row + c * 8) = v[i];
            }
        }
        ss = fmaf(bf_lo(v[i].x), bf_lo(v[i].x), ss);
        ss = fmaf(bf_hi(v[i].x), bf_hi(v[i].x), ss);
        ss = fmaf(bf_lo(v[i].y), bf_lo(v[i].y), ss);
        ss = fmaf(bf_hi(v[i].y), bf_hi(v[i].y), ss);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((t & 63) == 0) part[t >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += part[i];
    const float inv = 1.0f / sqrtf(tot / static_cast<float>(D) + eps);
    char* orow = out + static_cast<int64_t>(m) * ld_out * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = t + 1024 * i;
        if (c >= pieces) continue;
        const uint2 wv = *reinterpret_cast<const uint2*>(w + c * 8);
        uint2 o;
        o.x = pack_bf(bf_lo(v[i].x) * inv * bf_lo(wv.x), bf_hi(v[i].x) * inv * bf_hi(wv.x));
        o.y = pack_bf(bf_lo(v[i].y) * inv * bf_lo(wv.y), bf_hi(v[i].y) * inv * bf_hi(wv.y));
        *reinterpret_cast<uint2*>(orow + c * 8) = o;
    }
}

// ---- rotary embedding + KV-cache write.  qkv: [M][ld] = q heads | k heads | v heads of position m.  Pairs (i, i + 64), i < 64:
// q heads are rotated in place; k heads are rotated into k_cache[row][kvh][pos][:]; v heads go to vt_cache[row][kvh][:][pos].
// pos[m] is clamped into the cache (padding behind a ragged feed lands in the last slot, which no real token uses).
// SLABS: the projection's values come from the slabs (+ bias) instead of from qkv; the rotated q still goes to qkv.
template <bool SLABS>
__global__ __launch_bounds__(256) void k_rope_kv_store(char* __restrict__ qkv, int64_t ld, const int32_t* __restrict__ pos,
                                                       const int32_t* __restrict__ rows, const float* __restrict__ inv_freq,
                                                       char* __restrict__ k_cache, char* __restrict__ vt_cache, int M, int T,
                                                       int H, int KVH, int t_max, Slabs z, const uint16_t* __restrict__ bias) {
    // thread = (position m, head, four pairs i = 4 i4 .. 4 i4 + 3 with their partners i + 64): 16-byte slab loads, 8-byte stores
    const int64_t id = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int heads = H + 2 * KVH;
    if (id >= static_cast<int64_t>(M) * heads * 16) return;
    const int i0 = static_cast<int>(id & 15) * 4;
    const int head = static_cast<int>((id >> 4) % heads);
    const int m = static_cast<int>((id >> 4) / heads);
    const int p = min(max(pos[m], 0), t_max - 1);
    char* const src = qkv + (static_cast<int64_t>(m) * ld + static_cast<int64_t>(head) * kHd) * 2;
    float x1[4], x2[4];                 // the projection's values, already rounded to bf16
    if constexpr (SLABS) {
        const int col = head * kHd + i0;
        slab_sum4(z, m, col, x1);
        slab_sum4(z, m, col + 64, x2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (bias) {
                x1[j] += __uint_as_float(static_cast<uint32_t>(bias[col + j]) << 16);
                x2[j] += __uint_as_float(static_cast<uint32_t>(bias[col + 64 + j]) << 16);
            }
            const uint32_t pr = pack_bf(x1[j], x2[j]);
            x1[j] = bf_lo(pr);
            x2[j] = bf_hi(pr);
        }
    } else {
        const uint2 a = *reinterpret_cast<const uint2*>(src + i0 * 2), b = *reinterpret_cast<const uint2*>(src + (i0 + 64) * 2);
        x1[0] = bf_lo(a.x); x1[1] = bf_hi(a.x); x1[2] = bf_lo(a.y); x1[3] = bf_hi(a.y);
        x2[0] = bf_lo(b.x); x2[1] = bf_hi(b.x); x2[2] = bf_lo(b.y); x2[3] = bf_hi(b.y);
    }
    const int seq = m / T;
    const int64_t row = rows ? rows[seq] : seq;
    if (head >= H + KVH) {                 // v: transposed store (the values are bf16 already: pack_bf is exact on them)
        const int kvh = head - H - KVH;
        uint16_t* dst = reinterpret_cast<uint16_t*>(vt_cache) + ((row * KVH + kvh) * kHd) * static_cast<int64_t>(t_max) + p;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dst[static_cast<int64_t>(i0 + j) * t_max] = static_cast<uint16_t>(__float_as_uint(x1[j]) >> 16);
            dst[static_cast<int64_t>(i0 + j + 64) * t_max] = static_cast<uint16_t>(__float_as_uint(x2[j]) >> 16);
        }
        return;
    }
    float o1[4], o2[4];
    const f32x4 fr = *reinterpret_cast<const f32x4*>(inv_freq + i0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float ang = static_cast<float>(p) * fr[j];
        const float c = cosf(ang), sn = sinf(ang);
        o1[j] = x1[j] * c - x2[j] * sn;
        o2[j] = x2[j] * c + x1[j] * sn;
    }
    uint2 lo, hi;
    lo.x = pack_bf(o1[0], o1[1]); lo.y = pack_bf(o1[2], o1[3]);
    hi.x = pack_bf(o2[0], o2[1]); hi.y = pack_bf(o2[2], o2[3]);
    char* dst = src;
    if (head >= H) {
        const int kvh = head - H;
        dst = k_cache + ((((row * KVH + kvh) * static_cast<int64_t>(t_max)) + p) * kHd) * 2;
    }
    *reinterpret_cast<uint2*>(dst + i0 * 2) = lo;
    *reinterpret_cast<uint2*>(dst + (i0 + 64) * 2) = hi;
}

// ---- act[m][i] = silu(gu[m][i]) * gu[m][I + i], f32 arithmetic, one rounding.  Thread = 8 elements.
template <bool SLABS>
__global__ __launch_bounds__(256) void k_silu_mul(const char* __restrict__ gu, int64_t ld_gu, char* __restrict__ act, int64_t ld_act,
                                                  int M, int I, Slabs z) {
    const int64_t id = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int chunks = I / 8;
    if (id >= static_cast<int64_t>(M) * chunks) return;
    const int m = static_cast<int>(id / chunks), c = static_cast<int>(id % chunks);
    u32x4 g, u;
    if constexpr (SLABS) {
        float pg[8], pu[8];
        slab_sum8(z, m, c * 8, pg);
        slab_sum8(z, m, I + c * 8, pu);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g[j] = pack_bf(pg[2 * j], pg[2 * j + 1]);
            u[j] = pack_bf(pu[2 * j], pu[2 * j + 1]);
        }
    } else {
        const char* row = gu + static_cast<int64_t>(m) * ld_gu * 2;
        g = *reinterpret_cast<const u32x4*>(row + c * 16);
        u = *reinterpret_cast<const u32x4*>(row + (static_cast<int64_t>(I) + c * 8) * 2);
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float g0 = bf_lo(g[j]), g1 = bf_hi(g[j]);
        o[j] = pack_bf(g0 / (1.0f + expf(-g0)) * bf_lo(u[j]), g1 / (1.0f + expf(-g1)) * bf_hi(u[j]));
    }
    *reinterpret_cast<u32x4*>(act + (static_cast<int64_t>(m) * ld_act + c * 8) * 2) = o;
}

// ---- ragged attention over the cache.  Workgroup = (32-row tile of the sequence's rep * T query rows, kv head, sequence), 4
// waves; wave w takes the 32-key tiles w, w + 4, ... of keys [0, max pos] with its own online-softmax state, the four states
// meet in LDS.  Query row rho = t * rep + g: position t of the feed, head kvh * rep + g; it attends keys j <= pos[b * T + t].
//   S^T tile  X[a][rho] = sum_d K[key0 + kappa(a)][d] Q[rho][d]      A = K rows (cache, 16 B per lane and k-step), B = Q rows
//   O^T[d][rho] += sum_a Vt[d][key0 + kappa(a)] P[a][rho]            A = Vt rows (cache, 16 B), B = P = the X registers as bf16
// kappa swaps bits 2 and 3 of a: the accumulator registers 8s .. 8s+7 of lane half h hold X rows 16s + 8(j>>2) + 4h + (j&3), and
// with the permutation those are the 8 CONSECUTIVE keys key0 + 16s + 8h + j -- one 16-byte piece of a Vt row.
__device__ __forceinline__ int kappa(int a) { return (a & ~12) | ((a & 4) << 1) | ((a & 8) >> 1); }

constexpr float kNegBig = -1.0e30f;

__global__ __launch_bounds__(256) void k_attn_ragged(const char* __restrict__ qkv, int64_t ld_q, const char* __restrict__ k_cache,
                                                     const char* __restrict__ vt_cache, const int32_t* __restrict__ pos,
                                                     const int32_t* __restrict__ rows, char* __restrict__ out, int64_t ld_o,
                                                     int T, int H, int KVH, int t_max, float scale_log2) {
    __shared__ float obuf[4][4][16][64];      // [wave][d tile][register][lane]: 64 KiB
    __shared__ float mbuf[4][32], lbuf[4][32];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rt = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int rep = H / KVH;
    const int n_rows = rep * T;
    const int rho = 32 * rt + r;
    const bool valid = rho < n_rows;
    const int rho_c = min(rho, n_rows - 1);
    const int tq = rho_c / rep, g = rho_c % rep;
    const int m = b * T + tq;
    const int head = kvh * rep + g;
    const int my_pos = valid ? min(pos[m], t_max - 1) : -1;
    // keys in use by this tile's rows
    int lmax = my_pos;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lmax = max(lmax, __shfl_xor(lmax, o, 64));
    const int n_tiles = lmax / 32 + 1;          // lmax >= 0: row 0 of every tile is valid
    const int64_t row = rows ? rows[b] : b;
    const char* const kbase = k_cache + ((row * KVH + kvh) * static_cast<int64_t>(t_max)) * kHd * 2;
    const char* const vbase = vt_cache + ((row * KVH + kvh) * static_cast<int64_t>(kHd)) * t_max * 2;

    bf16x8 qf[8];
    {
        const char* qrow = qkv + (static_cast<int64_t>(m) * ld_q + static_cast<int64_t>(head) * kHd) * 2;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + (16 * ks + 8 * h) * 2);
    }
    f32x16 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.0f;
    float m_run = kNegBig, l_run = 0.0f;

    for (int tile = wv; tile < n_tiles; tile += 4) {
        const int key0 = 32 * tile;
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.0f;
        {
            const char* krow = kbase + static_cast<int64_t>(key0 + kappa(r)) * kHd * 2 + 16 * h;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + 32 * ks);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            }
        }
        float tmax = kNegBig;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int a = (i & 3) + 8 * (i >> 2) + 4 * h;
            const bool ok = key0 + kappa(a) <= my_pos;
            s[i] = ok ? s[i] * scale_log2 : -INFINITY;
            tmax = fmaxf(tmax, s[i]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = __builtin_amdgcn_exp2f(s[i] - m_new);
            psum += s[i];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = static_cast<__bf16>(s[8 * st + j]);
            const char* vcol = vbase + (static_cast<int64_t>(r) * t_max + key0 + 16 * st + 8 * h) * 2;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vcol + static_cast<int64_t>(32 * dt) * t_max * 2);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
            }
        }
    }
    // the four waves' states meet: wave w finishes d tile w
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) obuf[wv][dt][i][lane] = oacc[dt][i];
    if (h == 0) { mbuf[wv][r] = m_run; lbuf[wv][r] = l_run; }
    __syncthreads();
    float mw[4], f[4];
    float m_all = kNegBig;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = mbuf[w][r]; m_all = fmaxf(m_all, mw[w]); }
    float l_all = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { f[w] = __builtin_amdgcn_exp2f(mw[w] - m_all); l_all += lbuf[w][r] * f[w]; }
    const float inv_l = l_all > 0.0f ? 1.0f / l_all : 0.0f;
    if (!valid) return;
    const int dt = wv;
    char* orow = out + (static_cast<int64_t>(m) * ld_o + static_cast<int64_t>(head) * kHd + 32 * dt + 4 * h) * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float acc = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) acc += obuf[w][dt][4 * q + e][lane] * f[w];      // fixed wave order
            v[e] = acc * inv_l;
        }
        uint2 o;
        o.x = pack_bf(v[0], v[1]);
        o.y = pack_bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(orow + 16 * q) = o;        // d = 32 dt + 8 q + 4 h + e
    }
}

}  // namespace
}  // namespace asd

using namespace asd;

namespace {
int check_bf16(int dtype) { return dtype == ASD_DTYPE_BF16 ? ASD_OK : ASD_ERR_UNSUPPORTED; }
}

namespace {
Slabs slabs_of(const void* ws, int k_slices, int M, int N) {
    Slabs z{};
    z.base = k_slices > 1 ? static_cast<const float*>(ws) : nullptr;
    z.k_slices = k_slices;
    z.stride = static_cast<int64_t>(M) * N;
    z.N = N;
    return z;
}
int rmsnorm_run(void* x, int64_t ld_x, const void* weight, float eps, int dtype, int M, int D, void* out, int64_t ld_out,
                void* stream, const Slabs& z) {
    if (M < 0 || D < 1 || !(eps >= 0.0f)) return ASD_ERR_INVALID_ARG;
    if (M == 0) return ASD_OK;
    if (check_bf16(dtype) != ASD_OK || D % 4 != 0 || D > 8192) return ASD_ERR_UNSUPPORTED;
    if (!x || !weight || !out || ld_x < D || ld_out < D) return ASD_ERR_INVALID_ARG;
    if (!aligned_to(x, 8) || !aligned_to(weight, 8) || !aligned_to(out, 8) || ld_x % 4 != 0 || ld_out % 4 != 0) return ASD_ERR_ALIGNMENT;
    if (z.base)
        hipLaunchKernelGGL(k_rmsnorm<true>, dim3(static_cast<unsigned>(M)), dim3(1024), 0, static_cast<hipStream_t>(stream),
                           static_cast<char*>(x), ld_x, static_cast<const char*>(weight), eps, static_cast<char*>(out), ld_out, D, z);
    else
        hipLaunchKernelGGL(k_rmsnorm<false>, dim3(static_cast<unsigned>(M)), dim3(1024), 0, static_cast<hipStream_t>(stream),
                           static_cast<char*>(x), ld_x, static_cast<const char*>(weight), eps, static_cast<char*>(out), ld_out, D, z);
    return launch_status();
}
}  // namespace

ASD_EXPORT int asd_rmsnorm(const void* x, int64_t ld_x, const void* weight, float eps, int dtype, int M, int D, void* out,
                           int64_t ld_out, void* stream) {
    return rmsnorm_run(const_cast<void*>(x), ld_x, weight, eps, dtype, M, D, out, ld_out, stream, Slabs{});
}

namespace {
int silu_mul_run(const void* gate_up, int64_t ld_gu, int dtype, int M, int I, void* act, int64_t ld_act, void* stream, const Slabs& z);
}
ASD_EXPORT int asd_silu_mul(const void* gate_up, int64_t ld_gu, int dtype, int M, int I, void* act, int64_t ld_act, void* stream) {
    return silu_mul_run(gate_up, ld_gu, dtype, M, I, act, ld_act, stream, Slabs{});
}
namespace {
int silu_mul_run(const void* gate_up, int64_t ld_gu, int dtype, int M, int I, void* act, int64_t ld_act, void* stream, const Slabs& z) {
    if (M < 0 || I < 1) return ASD_ERR_INVALID_ARG;
    if (M == 0) return ASD_OK;
    if (check_bf16(dtype) != ASD_OK || I % 8 != 0) return ASD_ERR_UNSUPPORTED;
    if (!gate_up || !act || ld_gu < 2 * static_cast<int64_t>(I) || ld_act < I) return ASD_ERR_INVALID_ARG;
    if (!aligned_to(gate_up, 16) || !aligned_to(act, 16) || ld_gu % 8 != 0 || ld_act % 8 != 0) return ASD_ERR_ALIGNMENT;
    const int64_t n = static_cast<int64_t>(M) * (I / 8);
    if ((n + 255) / 256 >= (1ll << 31)) return ASD_ERR_UNSUPPORTED;
    if (z.base)
        hipLaunchKernelGGL(k_silu_mul<true>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const char*>(gate_up), ld_gu, static_cast<char*>(act), ld_act, M, I, z);
    else
        hipLaunchKernelGGL(k_silu_mul<false>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const char*>(gate_up), ld_gu, static_cast<char*>(act), ld_act, M, I, z);
    return launch_status();
}
}  // namespace

namespace {
int check_attn_shape(int B, int T, int H, int KVH, int head_dim, int t_max) {
    if (B < 0 || T < 1 || H < 1 || KVH < 1 || t_max < 1) return ASD_ERR_INVALID_ARG;
    if (head_dim != kHd || H % KVH != 0 || t_max % 32 != 0 || B > 65535 || KVH > 65535) return ASD_ERR_UNSUPPORTED;
    return ASD_OK;
}
}  // namespace

namespace {
int rope_kv_run(void* qkv, int64_t ld_qkv, const int32_t* pos, const int32_t* rows, const float* inv_freq, int dtype, int B, int T,
                int H, int KVH, int head_dim, void* k_cache, void* vt_cache, int t_max, void* stream, const Slabs& z, const void* bias);
}
ASD_EXPORT int asd_rope_kv_store(void* qkv, int64_t ld_qkv, const int32_t* pos, const int32_t* rows, const float* inv_freq,
                                 int dtype, int B, int T, int H, int KVH, int head_dim, void* k_cache, void* vt_cache, int t_max,
                                 void* stream) {
    return rope_kv_run(qkv, ld_qkv, pos, rows, inv_freq, dtype, B, T, H, KVH, head_dim, k_cache, vt_cache, t_max, stream, Slabs{}, nullptr);
}
namespace {
int rope_kv_run(void* qkv, int64_t ld_qkv, const int32_t* pos, const int32_t* rows, const float* inv_freq, int dtype, int B, int T,
                int H, int KVH, int head_dim, void* k_cache, void* vt_cache, int t_max, void* stream, const Slabs& z, const void* bias) {
    if (int rc = check_attn_shape(B, T, H, KVH, head_dim, t_max)) return rc;
    if (B == 0) return ASD_OK;
    if (check_bf16(dtype) != ASD_OK) return ASD_ERR_UNSUPPORTED;
    const int64_t width = static_cast<int64_t>(H + 2 * KVH) * kHd;
    if (!qkv || !pos || !inv_freq || !k_cache || !vt_cache || ld_qkv < width) return ASD_ERR_INVALID_ARG;
    if (!aligned_to(qkv, 16) || !aligned_to(k_cache, 16) || !aligned_to(vt_cache, 16) || ld_qkv % 8 != 0) return ASD_ERR_ALIGNMENT;
    const int64_t M = static_cast<int64_t>(B) * T;
    const int64_t n = M * (H + 2 * KVH) * 16;
    if (M >= (1ll << 31) || (n + 255) / 256 >= (1ll << 31)) return ASD_ERR_UNSUPPORTED;
    if (z.base)
        hipLaunchKernelGGL(k_rope_kv_store<true>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<char*>(qkv), ld_qkv, pos, rows, inv_freq, static_cast<char*>(k_cache), static_cast<char*>(vt_cache),
                           static_cast<int>(M), T, H, KVH, t_max, z, static_cast<const uint16_t*>(bias));
    else
        hipLaunchKernelGGL(k_rope_kv_store<false>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<char*>(qkv), ld_qkv, pos, rows, inv_freq, static_cast<char*>(k_cache), static_cast<char*>(vt_cache),
                           static_cast<int>(M), T, H, KVH, t_max, z, static_cast<const uint16_t*>(bias));
    return launch_status();
}
}  // namespace

ASD_EXPORT int asd_attn_ragged(const void* qkv, int64_t ld_qkv, const void* k_cache, const void* vt_cache, const int32_t* pos,
                               const int32_t* rows, int dtype, int B, int T, int H, int KVH, int head_dim, int t_max,
                               void* out, int64_t ld_out, void* stream) {
    if (int rc = check_attn_shape(B, T, H, KVH, head_dim, t_max)) return rc;
    if (B == 0) return ASD_OK;
    if (check_bf16(dtype) != ASD_OK) return ASD_ERR_UNSUPPORTED;
    if (!qkv || !k_cache || !vt_cache || !pos || !out || ld_qkv < static_cast<int64_t>(H) * kHd || ld_out < static_cast<int64_t>(H) * kHd)
        return ASD_ERR_INVALID_ARG;
    if (!aligned_to(qkv, 16) || !aligned_to(k_cache, 16) || !aligned_to(vt_cache, 16) || !aligned_to(out, 8) || ld_qkv % 8 != 0 ||
        ld_out % 4 != 0)
        return ASD_ERR_ALIGNMENT;
    const int n_rows = (H / KVH) * T;
    const int row_tiles = (n_rows + 31) / 32;
    const float scale_log2 = static_cast<float>(1.4426950408889634074 / sqrt(static_cast<double>(kHd)));
    hipLaunchKernelGGL(k_attn_ragged, dim3(static_cast<unsigned>(row_tiles), static_cast<unsigned>(KVH), static_cast<unsigned>(B)),
                       dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const char*>(qkv), ld_qkv,
                       static_cast<const char*>(k_cache), static_cast<const char*>(vt_cache), pos, rows, static_cast<char*>(out),
                       ld_out, T, H, KVH, t_max, scale_log2);
    return launch_status();
}

// ---- the whole stack from one host call
namespace {
struct Scratch {
    size_t hn, qkv, attn, gu, act, lin, total;
};
Scratch scratch_layout(const asd_decoder_shape_t& s, int M) {
    Scratch z{};
    const size_t m = static_cast<size_t>(M);
    const size_t kv = static_cast<size_t>(s.kv_heads) * s.head_dim;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += round_up(bytes, 256); return at; };
    z.hn = take(m * s.hidden * 2);
    z.qkv = take(m * (s.hidden + 2 * kv) * 2);
    z.attn = take(m * s.hidden * 2);
    z.gu = take(m * 2 * static_cast<size_t>(s.intermediate) * 2);
    z.act = take(m * static_cast<size_t>(s.intermediate) * 2);
    size_t lin = asd_linear_workspace_bytes(M, static_cast<int>(s.hidden + 2 * kv), s.hidden);
    lin = lin > asd_linear_workspace_bytes(M, s.hidden, s.hidden) ? lin : asd_linear_workspace_bytes(M, s.hidden, s.hidden);
    lin = lin > asd_linear_workspace_bytes(M, 2 * s.intermediate, s.hidden) ? lin : asd_linear_workspace_bytes(M, 2 * s.intermediate, s.hidden);
    lin = lin > asd_linear_workspace_bytes(M, s.hidden, s.intermediate) ? lin : asd_linear_workspace_bytes(M, s.hidden, s.intermediate);
    z.lin = take(lin);
    z.total = off;
    return z;
}
int check_shape(const asd_decoder_shape_t* s) {
    if (!s || s->hidden < 1 || s->heads < 1 || s->kv_heads < 1 || s->intermediate < 1 || s->t_max < 1 || !s->inv_freq) return ASD_ERR_INVALID_ARG;
    if (s->head_dim != kHd || s->hidden != s->heads * s->head_dim || s->heads % s->kv_heads != 0 || s->hidden % 64 != 0 ||
        s->intermediate % 64 != 0 || s->hidden > 8192 || s->t_max % 32 != 0)
        return ASD_ERR_UNSUPPORTED;
    return ASD_OK;
}
}  // namespace

ASD_EXPORT size_t asd_decoder_scratch_bytes(const asd_decoder_shape_t* shape, int M) {
    if (check_shape(shape) != ASD_OK || M < 1) return 0;
    return scratch_layout(*shape, M).total;
}

ASD_EXPORT int asd_decoder_forward(const asd_layer_t* layers, int n_layers, const asd_decoder_shape_t* shape, void* x, int64_t ld_x,
                                   const int32_t* pos, const int32_t* rows, int B, int T, const void* final_norm_w,
                                   void* normed_out, int64_t ld_normed, void* scratch, size_t scratch_bytes, void* stream) {
    if (int rc = check_shape(shape)) return rc;
    if (n_layers < 0 || B < 0 || T < 1 || (n_layers > 0 && !layers)) return ASD_ERR_INVALID_ARG;
    if ((final_norm_w != nullptr) != (normed_out != nullptr)) return ASD_ERR_INVALID_ARG;
    if (B == 0) return ASD_OK;
    const asd_decoder_shape_t& s = *shape;
    const int64_t M64 = static_cast<int64_t>(B) * T;
    if (M64 >= (1 << 24)) return ASD_ERR_UNSUPPORTED;
    const int M = static_cast<int>(M64);
    if (!x || !pos || ld_x < s.hidden || (normed_out && ld_normed < s.hidden)) return ASD_ERR_INVALID_ARG;
    const Scratch z = scratch_layout(s, M);
    if (!scratch || scratch_bytes < z.total) return ASD_ERR_WORKSPACE;
    if (!aligned_to(scratch, 256) || !aligned_to(x, 16) || ld_x % 8 != 0) return ASD_ERR_ALIGNMENT;
    char* const base = static_cast<char*>(scratch);
    void* const hn = base + z.hn;
    void* const qkv = base + z.qkv;
    void* const attn = base + z.attn;
    void* const gu = base + z.gu;
    void* const act = base + z.act;
    void* const lin = base + z.lin;
    const size_t lin_bytes = z.total - z.lin;
    const int kvw = s.kv_heads * s.head_dim;
    const int qkv_w = s.hidden + 2 * kvw;
    const int dt = ASD_DTYPE_BF16;
    const int64_t gu_w = 2 * static_cast<int64_t>(s.intermediate);
    // Where a projection's plan cuts the reduction into slices, the kernel that FOLLOWS it adds the partials itself (no reduce
    // launch): rope + cache write after q|k|v, the next norm (with the residual connection) after o and down, silu * up after
    // gate|up.  Nine launches per layer either way.
    int rc = ASD_OK;
    bool have_hn = false;              // hn already holds norm(x) for the layer about to start
    for (int l = 0; l < n_layers && rc == ASD_OK; ++l) {
        const asd_layer_t& L = layers[l];
        if (!L.ln1_w || !L.qkv_w || !L.o_w || !L.ln2_w || !L.gate_up_w || !L.down_w || !L.k_cache || !L.vt_cache) return ASD_ERR_INVALID_ARG;
        int ks = 1;
        const int64_t ldw_h = L.weights_packed ? 0 : s.hidden, ldw_i = L.weights_packed ? 0 : s.intermediate;   // 0: tile-major image
        if (!have_hn) rc = rmsnorm_run(x, ld_x, L.ln1_w, s.rms_eps, dt, M, s.hidden, hn, s.hidden, stream, Slabs{});
        if (rc == ASD_OK) rc = asd_linear_partial(hn, s.hidden, L.qkv_w, ldw_h, L.qkv_b, nullptr, 0, dt, M, qkv_w, s.hidden, qkv, qkv_w, lin, lin_bytes, stream, &ks);
        if (rc == ASD_OK) rc = rope_kv_run(qkv, qkv_w, pos, rows, s.inv_freq, dt, B, T, s.heads, s.kv_heads, s.head_dim, L.k_cache, L.vt_cache, s.t_max, stream,
                                           slabs_of(lin, ks, M, qkv_w), L.qkv_b);
        if (rc == ASD_OK) rc = asd_attn_ragged(qkv, qkv_w, L.k_cache, L.vt_cache, pos, rows, dt, B, T, s.heads, s.kv_heads, s.head_dim, s.t_max, attn, s.hidden, stream);
        if (rc == ASD_OK) rc = asd_linear_partial(attn, s.hidden, L.o_w, ldw_h, nullptr, x, ld_x, dt, M, s.hidden, s.hidden, x, ld_x, lin, lin_bytes, stream, &ks);
        if (rc == ASD_OK) rc = rmsnorm_run(x, ld_x, L.ln2_w, s.rms_eps, dt, M, s.hidden, hn, s.hidden, stream, slabs_of(lin, ks, M, s.hidden));
        if (rc == ASD_OK) rc = asd_linear_partial(hn, s.hidden, L.gate_up_w, ldw_h, nullptr, nullptr, 0, dt, M, 2 * s.intermediate, s.hidden, gu, gu_w, lin, lin_bytes, stream, &ks);
        if (rc == ASD_OK) rc = silu_mul_run(gu, gu_w, dt, M, s.intermediate, act, s.intermediate, stream, slabs_of(lin, ks, M, 2 * s.intermediate));
        if (rc == ASD_OK) rc = asd_linear_partial(act, s.intermediate, L.down_w, ldw_i, nullptr, x, ld_x, dt, M, s.hidden, s.intermediate, x, ld_x, lin, lin_bytes, stream, &ks);
        if (rc != ASD_OK) break;
        // the norm that follows the down projection: the next layer's ln1, or the final norm
        const bool last = l + 1 == n_layers;
        const void* next_w = last ? final_norm_w : layers[l + 1].ln1_w;
        void* next_out = last ? normed_out : hn;
        const int64_t next_ld = last ? ld_normed : s.hidden;
        if (next_w) {
            rc = rmsnorm_run(x, ld_x, next_w, s.rms_eps, dt, M, s.hidden, next_out, next_ld, stream, slabs_of(lin, ks, M, s.hidden));
            have_hn = true;
        } else if (ks > 1) {           // no norm follows: finish the residual connection on its own
            rc = rmsnorm_run(x, ld_x, L.ln2_w, s.rms_eps, dt, M, s.hidden, hn, s.hidden, stream, slabs_of(lin, ks, M, s.hidden));
        }
    }
    if (rc == ASD_OK && n_layers == 0 && final_norm_w)
        rc = rmsnorm_run(x, ld_x, final_norm_w, s.rms_eps, dt, M, s.hidden, normed_out, ld_normed, stream, Slabs{});
    return rc;
}
