// lse_device.hpp -- log2-domain online log-sum-exp building blocks shared by the streaming kernels
// (verify_accept.hip, residual_sample.hip, lm_head_verify.hip): element unpacking for bf16 / f16 / f32,
// the per-vector accumulate, DPP wave reductions, the range-checked 16-byte buffer load, the split
// logarithm and the accept test of a finished row.
// A partial sum is a pair (m2, s) with  sum_v exp(x_v / T) = s * 2^m2 ;  c2 = log2(e) / T.
#pragma once

#include "common.hpp"

#include <math.h>

namespace asd {

constexpr float kLog2e = 1.4426950408889634f;
constexpr double kLn2d = 0.693147180559945309417232121458;
constexpr float kSentinel = -1.0e30f;  // "minus infinity" that stays finite under subtraction

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// v_max3_f32 / v_max_f32 written out: `fmaxf` on a value that came out of integer unpacking (or out of a loop-carried
// register) is preceded by a quieting `v_max_f32 x, x` per operand on this compiler; the instructions themselves return the
// non-NaN operand like fmaxf does (IEEE mode), so the values are the same and two to three issues per 16-byte vector go away.
static __device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
static __device__ __forceinline__ float max2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// (m2a, sa) (+) (m2b, sb) in the log2 domain
static __device__ __forceinline__ void ms_merge(float& m2, float& s, float m2b, float sb) {
    const float M = fmaxf(m2, m2b);
    const float ea = fast_exp2(m2 - M);
    const float eb = fast_exp2(m2b - M);
    s = fmaf(s, ea, sb * eb);
    m2 = M;
}

// ---- wave64 reductions on DPP (row_shr 1,2,4,8 then row_bcast 15 / 31: the GFX9 sequence; the
// total lands in lane 63 and is broadcast with v_readlane).  `__shfl_xor` lowers to ds_bpermute
// (an LDS round trip per step); a 6-step butterfly of (m2, s) pairs cost ~2 us on the kernel's
// tail (gpurun stamps, profiles/r01_stamps_*.log), this costs a few hundred cycles.
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ float dpp_move(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
#if defined(ASD_DPP_ASM_REDUCTIONS)
// ASD_DPP_ASM_REDUCTIONS (verify_accept.hip, lm_head_verify.hip: compiled with -ffp-contract=off, where it is bit-neutral) --
// one DPP instruction per step: `v_max_f32_dpp v, v, v <ctrl>` / `v_add_f32_dpp` with bound_ctrl off leave a lane whose source
// is out of range (or whose row is masked off) UNCHANGED -- which is max(v, -inf) / v + 0, what the builtin form (a v_mov_dpp
// into an identity register, a quieting v_max, the operation: 3-4 issues per step) computed.  Same values, 6 issues instead of
// 24 (max) / 14 (sum) per reduction -- the streaming kernels run one of each per tile.  `s_nop 1`: a DPP operand written by the
// previous VALU instruction needs two wait states, and the compiler does not look into an asm block.
static __device__ __forceinline__ float wave_max(float v) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
static __device__ __forceinline__ float wave_sum(float v) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#else
// The builtin form.  Translation units compiled with floating-point contraction (the samplers) keep it: there the compiler fuses
// the multiplication that feeds a wave_sum into the first step's addition, and their recorded outputs pin those bits.
static __device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0x111, 0xf>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x112, 0xf>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x114, 0xf>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x118, 0xf>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x142, 0xa>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x143, 0xc>(-INFINITY, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
static __device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0x111, 0xf>(0.0f, v);
    v += dpp_move<0x112, 0xf>(0.0f, v);
    v += dpp_move<0x114, 0xf>(0.0f, v);
    v += dpp_move<0x118, 0xf>(0.0f, v);
    v += dpp_move<0x142, 0xa>(0.0f, v);
    v += dpp_move<0x143, 0xc>(0.0f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#endif
// all 64 lanes' (m2, s) -> one pair, valid (uniform) in every lane: max first, ONE rescale per lane
static __device__ __forceinline__ void wave_merge(float& m2, float& s) {
    const float M = wave_max(m2);
    s = wave_sum(s * fast_exp2(m2 - M));
    m2 = M;
}

// ---- (m2, s, t) triples: additionally t = sum_v exp(x_v / T - ...) * x_v on the same scale as s, from which the
// entropy of the row's softmax follows (N1 / A14: RESEARCH_PROTOCOL.md:378-385): H = ln2 * (m2 + log2 s - c2 * t / s).
static __device__ __forceinline__ void ms_merge3(float& m2, float& s, float& t, float m2b, float sb, float tb) {
    const float M = fmaxf(m2, m2b);
    const float ea = fast_exp2(m2 - M);
    const float eb = fast_exp2(m2b - M);
    s = fmaf(s, ea, sb * eb);
    t = fmaf(t, ea, tb * eb);
    m2 = M;
}
static __device__ __forceinline__ void wave_merge3(float& m2, float& s, float& t) {
    const float M = wave_max(m2);
    const float f = fast_exp2(m2 - M);
    s = wave_sum(s * f);
    t = wave_sum(t * f);
    m2 = M;
}
// same (m2, s) arithmetic as accum8 / accum4 below, operation for operation (the STATS kernel's verify outputs are
// bit-identical to the plain kernel's), plus the weighted sum
template <int N>
static __device__ __forceinline__ void accum_t(const float (&x)[N], float c2, float& m2, float& s, float& t) {
    float vmax;
    if constexpr (N == 8) {
        vmax = max3(x[0], x[1], x[2]);
        vmax = max3(vmax, x[3], x[4]);
        vmax = max3(vmax, x[5], x[6]);
        vmax = max2(vmax, x[7]);
    } else {
        vmax = max2(max3(x[0], x[1], x[2]), x[3]);
    }
    const float M = max2(m2, vmax * c2);
    const float scale = fast_exp2(m2 - M);
    float e[N];
    float w = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        e[i] = fast_exp2(fmaf(x[i], c2, -M));
        w = fmaf(e[i], e[i] > 0.0f ? x[i] : 0.0f, w);      // a -inf logit carries no mass: 0 * -inf must not poison t
    }
    float sum;
    if constexpr (N == 8) sum = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
    else sum = (e[0] + e[1]) + (e[2] + e[3]);
    s = fmaf(s, scale, sum);
    t = fmaf(t, scale, w);
    m2 = M;
}

// c2 = log2(e) / temperature: the per-element FMA constant, so temperature scaling costs nothing
static __device__ __forceinline__ void accum_scalar(float x, float c2, float& m2, float& s) {
    const float M = max2(m2, x * c2);
    s = fmaf(s, fast_exp2(m2 - M), fast_exp2(fmaf(x, c2, -M)));
    m2 = M;
}

static __device__ __forceinline__ void accum8(const float (&x)[8], float c2, float& m2, float& s) {
    float vmax = max3(x[0], x[1], x[2]);
    vmax = max3(vmax, x[3], x[4]);
    vmax = max3(vmax, x[5], x[6]);
    vmax = max2(vmax, x[7]);
    const float M = max2(m2, vmax * c2);
    const float scale = fast_exp2(m2 - M);
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = fast_exp2(fmaf(x[i], c2, -M));
    const float sum = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
    s = fmaf(s, scale, sum);
    m2 = M;
}

static __device__ __forceinline__ void accum4(const float (&x)[4], float c2, float& m2, float& s) {
    const float vmax = max2(max3(x[0], x[1], x[2]), x[3]);
    const float M = max2(m2, vmax * c2);
    const float scale = fast_exp2(m2 - M);
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = fast_exp2(fmaf(x[i], c2, -M));
    s = fmaf(s, scale, (e[0] + e[1]) + (e[2] + e[3]));
    m2 = M;
}

template <int DT>
struct Elem;

template <>
struct Elem<ASD_DTYPE_BF16> {
    static constexpr int kBytes = 2;
    static constexpr int kPerVec = 8;
    static constexpr uint32_t kNegInfWord = 0xFF80FF80u;
    static __device__ __forceinline__ float scalar(const void* p, int64_t i) {
        return __uint_as_float(static_cast<uint32_t>(static_cast<const uint16_t*>(p)[i]) << 16);
    }
    // the load and its first use split apart: raw() issues the load, from_raw() converts where the value is needed
    // (the s_waitcnt lands at the first USE; a volatile load would be followed by a full vmcnt(0) on this target)
    static __device__ __forceinline__ uint32_t raw(const void* p, int64_t i) {
        return static_cast<const uint16_t*>(p)[i];
    }
    static __device__ __forceinline__ float from_raw(uint32_t r) { return __uint_as_float(r << 16); }
    static __device__ __forceinline__ void accum(const u32x4& v, float c2, float& m2, float& s) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = __uint_as_float(v[i] << 16);
            x[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
        }
        accum8(x, c2, m2, s);
    }
    static __device__ __forceinline__ void accum3(const u32x4& v, float c2, float& m2, float& s, float& t) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = __uint_as_float(v[i] << 16);
            x[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
        }
        accum_t<8>(x, c2, m2, s, t);
    }
};

template <>
struct Elem<ASD_DTYPE_F16> {
    static constexpr int kBytes = 2;
    static constexpr int kPerVec = 8;
    static constexpr uint32_t kNegInfWord = 0xFC00FC00u;
    static __device__ __forceinline__ float scalar(const void* p, int64_t i) {
        return static_cast<float>(static_cast<const _Float16*>(p)[i]);
    }
    static __device__ __forceinline__ uint32_t raw(const void* p, int64_t i) {
        return static_cast<const uint16_t*>(p)[i];
    }
    static __device__ __forceinline__ float from_raw(uint32_t r) {
        return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(r)));
    }
    static __device__ __forceinline__ void accum(const u32x4& v, float c2, float& m2, float& s) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        float x[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = v[i];  // bit_cast of the vector-element lvalue itself reads element 0
            const h2 h = __builtin_bit_cast(h2, w);
            x[2 * i] = static_cast<float>(h[0]);
            x[2 * i + 1] = static_cast<float>(h[1]);
        }
        accum8(x, c2, m2, s);
    }
    static __device__ __forceinline__ void accum3(const u32x4& v, float c2, float& m2, float& s, float& t) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        float x[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = v[i];
            const h2 h = __builtin_bit_cast(h2, w);
            x[2 * i] = static_cast<float>(h[0]);
            x[2 * i + 1] = static_cast<float>(h[1]);
        }
        accum_t<8>(x, c2, m2, s, t);
    }
};

template <>
struct Elem<ASD_DTYPE_F32> {
    static constexpr int kBytes = 4;
    static constexpr int kPerVec = 4;
    static constexpr uint32_t kNegInfWord = 0xFF800000u;
    static __device__ __forceinline__ float scalar(const void* p, int64_t i) {
        return static_cast<const float*>(p)[i];
    }
    static __device__ __forceinline__ uint32_t raw(const void* p, int64_t i) {
        return static_cast<const uint32_t*>(p)[i];
    }
    static __device__ __forceinline__ float from_raw(uint32_t r) { return __uint_as_float(r); }
    static __device__ __forceinline__ void accum(const u32x4& v, float c2, float& m2, float& s) {
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = __uint_as_float(v[i]);
        accum4(x, c2, m2, s);
    }
    static __device__ __forceinline__ void accum3(const u32x4& v, float c2, float& m2, float& s, float& t) {
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = __uint_as_float(v[i]);
        accum_t<4>(x, c2, m2, s, t);
    }
};

// 16-byte buffer load; lanes whose offset is >= the descriptor's num_records return 0 without a
// memory access.  aux: 0 = default cache policy, 2 = nt (streamed, read-once data).
template <bool NT>
static __device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, NT ? 2 : 0));
}

// log2(x) for x >= 0 with ~1e-7 ABSOLUTE error at any magnitude: exponent exactly, v_log_f32 on
// the mantissa only, summed in f64.  x == 0 -> -inf; inf / NaN / negative propagate like log2f.
static __device__ __forceinline__ double log2_split(float x) {
    if (!(x > 0.0f) || !(x < INFINITY)) return static_cast<double>(__builtin_amdgcn_logf(x));
    int e;
    const float mant = frexpf(x, &e);
    return static_cast<double>(e) + static_cast<double>(__builtin_amdgcn_logf(mant));
}

// ---- the accept test of one row / one sequence (DESIGN.md "accept rule") --------------
// The sums are f64 (they mirror the oracle's structure); the two logarithms are split into an
// exact exponent and a v_log_f32 of the mantissa, which keeps their absolute error ~1e-7 without
// a software f64 log on the kernel's tail.
static __device__ __forceinline__ double log_u(float u) {
    return kLn2d * log2_split(u);  // u == 0 -> -inf (accepts any token of non-zero target probability), u < 0 -> NaN (rejects)
}
static __device__ __forceinline__ bool finish_row(float m2, float s, float x_tok, float c2, float lp_d, double lu,
                                                  float& lp_out) {
    // everything in the log2 domain with the SAME constant c2 the stream used, so its rounding cancels:
    // lp = ln2 * (x_tok*c2 - (m2 + log2 s)) = log softmax(a*x)[tok] exactly for a = c2*ln2 (= 1/T to 6e-8)
    const double l2 = static_cast<double>(m2) + log2_split(s);
    const double lp = kLn2d * (static_cast<double>(x_tok) * static_cast<double>(c2) - l2);
    lp_out = static_cast<float>(lp);
    // p_t(tok) == 0 (a -inf target logit: top-p / top-k masked rows, an id outside the shard) never accepts, not
    // even at u == 0 where log u = -inf <= -inf would: the rule is u < p_t / p_d.  NaN rejects by comparison.
    return lp > -INFINITY && lu <= lp - static_cast<double>(lp_d);
}

static __device__ __forceinline__ void finish_sequence(bool flag, int lane, int K, int b, int32_t* n_acc,
                                                       uint64_t* bits) {
    const unsigned long long bal = __ballot(flag && lane < K);
    if (lane == 0) {
        const unsigned long long inv = ~bal;
        int n = inv ? __builtin_ctzll(inv) : 64;
        n_acc[b] = n < K ? n : K;
        if (bits) bits[b] = bal;
    }
}

}  // namespace asd
