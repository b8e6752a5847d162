// verify_accept.hip -- batched logits-gather + log-sum-exp + acceptance-ratio test, gfx950.
//
// One streaming pass over target logits [B*K rows][V] (bf16 / f16 / f32).  HBM-bound: every
// logit is read exactly once with 16-byte loads; arithmetic is ~1 v_exp_f32 + 4 VALU per element.
//
// Geometry.  A row (one verified position) is cut into S contiguous vocab slices; one workgroup
// of THREADS lanes reduces one slice to a pair (m2, s) with   sum_v exp(x_v) = s * 2^m2
// (log2 domain: the per-element work is one FMA + one v_exp_f32, and the pair stays exactly
// consistent whatever rounding m2 itself carries).  Each lane keeps a running (m2, s) over its
// 16-byte vectors (online softmax, one rescale per vector), lanes are folded with a fixed
// xor-butterfly, waves through LDS.
//
// Hand-off.  The slice result is published as ONE 8-byte granule (write-through, agent scope),
// the publishing lane drains its store (s_waitcnt vmcnt(0)) and takes a ticket on the
// sequence's counter.  The workgroup whose ticket is the last of the sequence's K*S tickets
// stages the K*S granules of the K candidate rows in LDS, combines the S slices of every row in
// slice order (=> bitwise deterministic, independent of arrival order), gathers the drafted
// token's logit, runs the acceptance test, and turns the K accept flags into the sequence's
// accept mask / accepted-prefix length with one wave ballot.  Nothing spins: there is no wait
// anywhere in the kernel.  Granule regions are per sequence and padded to whole 256-byte blocks,
// so every line of them has exactly one reader per launch (MI355X_MICROARCH.md, inter-workgroup
// visibility: sc1 stores + drained ticket + sc1 loads).  The last arriver resets the ticket, so a
// workspace zeroed once serves every later stream-ordered call (hipGraph-replay safe: no epoch
// argument, no memset node).
//
// Reference arithmetic this replaces: src/training/generate_training_data.py:128-136
// (softmax -> index -> log -> .item(), one token per iteration).  The acceptance test has no
// reference symbol; it is specified in include/asd_hip.h and DESIGN.md.

#include "common.hpp"

namespace asd {
namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr double kLn2d = 0.693147180559945309417232121458;
constexpr float kSentinel = -1.0e30f;  // "minus infinity" that stays finite under subtraction
constexpr int kMaxStage = 1024;        // granules one finisher stages in LDS (K*S <= kMaxStage)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
    uint32_t* tickets;
    uint64_t* granules;
    uint32_t region;  // granules per sequence region
    int mode;         // 0: accept, 1: emit (m2, s, g) partials
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// (m2a, sa) (+) (m2b, sb) in the log2 domain
__device__ __forceinline__ void ms_merge(float& m2, float& s, float m2b, float sb) {
    const float M = fmaxf(m2, m2b);
    const float ea = fast_exp2(m2 - M);
    const float eb = fast_exp2(m2b - M);
    s = fmaf(s, ea, sb * eb);
    m2 = M;
}

__device__ __forceinline__ void accum_scalar(float x, float& m2, float& s) {
    const float M = fmaxf(m2, x * kLog2e);
    s = fmaf(s, fast_exp2(m2 - M), fast_exp2(fmaf(x, kLog2e, -M)));
    m2 = M;
}

__device__ __forceinline__ void accum8(const float (&x)[8], float& m2, float& s) {
    float vmax = max3(x[0], x[1], x[2]);
    vmax = max3(vmax, x[3], x[4]);
    vmax = max3(vmax, x[5], x[6]);
    vmax = fmaxf(vmax, x[7]);
    const float M = fmaxf(m2, vmax * kLog2e);
    const float scale = fast_exp2(m2 - M);
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = fast_exp2(fmaf(x[i], kLog2e, -M));
    const float sum = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
    s = fmaf(s, scale, sum);
    m2 = M;
}

__device__ __forceinline__ void accum4(const float (&x)[4], float& m2, float& s) {
    const float vmax = fmaxf(max3(x[0], x[1], x[2]), x[3]);
    const float M = fmaxf(m2, vmax * kLog2e);
    const float scale = fast_exp2(m2 - M);
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = fast_exp2(fmaf(x[i], kLog2e, -M));
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
    static __device__ __forceinline__ void accum(const u32x4& v, float& m2, float& s) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = __uint_as_float(v[i] << 16);
            x[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
        }
        accum8(x, m2, s);
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
    static __device__ __forceinline__ void accum(const u32x4& v, float& m2, float& s) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        float x[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = v[i];  // bit_cast of the vector-element lvalue itself reads element 0
            const h2 h = __builtin_bit_cast(h2, w);
            x[2 * i] = static_cast<float>(h[0]);
            x[2 * i + 1] = static_cast<float>(h[1]);
        }
        accum8(x, m2, s);
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
    static __device__ __forceinline__ void accum(const u32x4& v, float& m2, float& s) {
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = __uint_as_float(v[i]);
        accum4(x, m2, s);
    }
};

template <bool NT>
__device__ __forceinline__ u32x4 load16(const u32x4* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// lse, log-prob and acceptance test of one row from its combined (m2, s) and gathered logit.
// f64 for the K-per-sequence epilogue only: it mirrors the oracle's structure, so the only
// difference left between the two is the f32 accumulation of s.
__device__ __forceinline__ bool finish_row(float m2, float s, float x_tok, float lp_d, float u,
                                           float& lp_out) {
    const double lse = kLn2d * (static_cast<double>(m2) + log2(static_cast<double>(s)));
    const double lp = static_cast<double>(x_tok) - lse;
    lp_out = static_cast<float>(lp);
    const double lu = log(static_cast<double>(u));  // u == 0 -> -inf, u < 0 -> NaN (rejects)
    return lu <= lp - static_cast<double>(lp_d);
}

__device__ __forceinline__ void finish_sequence(bool flag, int lane, int K, int b, int32_t* n_acc,
                                                uint64_t* bits) {
    const unsigned long long bal = __ballot(flag && lane < K);
    if (lane == 0) {
        const unsigned long long inv = ~bal;
        int n = inv ? __builtin_ctzll(inv) : 64;
        n_acc[b] = n < K ? n : K;
        if (bits) bits[b] = bal;
    }
}

template <int DT, int THREADS, int UNROLL, bool NT>
__global__ __launch_bounds__(THREADS) void k_verify(const VerifyParams p) {
    using E = Elem<DT>;
    constexpr int kWaves = THREADS / 64;
    __shared__ float red_m[kWaves];
    __shared__ float red_s[kWaves];
    __shared__ uint64_t stage[kMaxStage];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int S = p.S;
    const int64_t bid = blockIdx.x;
    const int row = static_cast<int>(bid / S);
    const int split = static_cast<int>(bid - static_cast<int64_t>(row) * S);
    const int b = row / p.K;
    const int k = row - b * p.K;

    const char* rowp = static_cast<const char*>(p.logits) + static_cast<int64_t>(row) * p.ld_row * E::kBytes;
    const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(rowp) & 15u);
    int head = mis ? static_cast<int>((16u - mis) / E::kBytes) : 0;
    if (head > p.V) head = p.V;
    const int nvec = (p.V - head) / E::kPerVec;
    const int tail = p.V - head - nvec * E::kPerVec;
    const u32x4* vec = reinterpret_cast<const u32x4*>(rowp + static_cast<int64_t>(head) * E::kBytes);
    const int v0 = static_cast<int>(static_cast<int64_t>(nvec) * split / S);
    const int v1 = static_cast<int>(static_cast<int64_t>(nvec) * (split + 1) / S);

    float m2 = kSentinel, s = 0.0f;

    // unaligned head / ragged tail (<= 7 elements each), folded into the first / last slice
    if (split == 0 && tid < head) accum_scalar(E::scalar(rowp, tid), m2, s);
    if (split == S - 1 && tid < tail)
        accum_scalar(E::scalar(rowp, static_cast<int64_t>(head) + static_cast<int64_t>(nvec) * E::kPerVec + tid), m2, s);

    int i = v0 + tid;
    for (; i + (UNROLL - 1) * THREADS < v1; i += UNROLL * THREADS) {
        u32x4 r[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) r[j] = load16<NT>(vec + i + j * THREADS);
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) E::accum(r[j], m2, s);
    }
    if (i < v1) {  // ragged last batch: predicated loads, still issued back to back
        u32x4 r[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int idx = i + j * THREADS;
            if (idx < v1) {
                r[j] = load16<NT>(vec + idx);
            } else {
                r[j] = u32x4{E::kNegInfWord, E::kNegInfWord, E::kNegInfWord, E::kNegInfWord};
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) E::accum(r[j], m2, s);
    }

    // lanes -> wave (fixed xor butterfly), waves -> workgroup (LDS)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float om = __shfl_xor(m2, off, 64);
        const float os = __shfl_xor(s, off, 64);
        ms_merge(m2, s, om, os);
    }
    if (kWaves > 1) {
        if (lane == 0) { red_m[wave] = m2; red_s[wave] = s; }
        __syncthreads();
        if (wave != 0) return;
        m2 = red_m[0];
        s = red_s[0];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) ms_merge(m2, s, red_m[w], red_s[w]);
    }

    // publish the slice, take a ticket; only wave 0 is left here
    const int KS = p.K * S;
    uint64_t* region = p.granules + static_cast<int64_t>(b) * p.region;
    int last = 0;
    if (lane == 0) {
        const uint64_t g = (static_cast<uint64_t>(__float_as_uint(s)) << 32) | __float_as_uint(m2);
        __hip_atomic_store(region + k * S + split, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(p.tickets + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (old == static_cast<uint32_t>(KS - 1));
    }
    last = __shfl(last, 0, 64);
    if (!last) return;

    // ---- last arriver of sequence b: finish its K rows ------------------------------------
    const int frow = b * p.K + lane;  // lane <-> draft position
    float x_tok = -INFINITY, lpd = 0.0f, uu = 1.0f;
    if (lane < p.K) {
        const int64_t t = static_cast<int64_t>(p.tok[frow]) - p.v_offset;
        if (t >= 0 && t < p.V)
            x_tok = E::scalar(static_cast<const char*>(p.logits) + static_cast<int64_t>(frow) * p.ld_row * E::kBytes, t);
        if (p.mode == 0) { lpd = p.lp_d[frow]; uu = p.u[frow]; }
    }
    for (int g = lane; g < KS; g += 64)
        stage[g] = __hip_atomic_load(region + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float fm = kSentinel, fs = 0.0f;
    if (lane < p.K) {
        for (int j = 0; j < S; ++j) {
            const uint64_t g = stage[lane * S + j];
            ms_merge(fm, fs, __uint_as_float(static_cast<uint32_t>(g)), __uint_as_float(static_cast<uint32_t>(g >> 32)));
        }
    }
    if (lane == 0) __hip_atomic_store(p.tickets + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    if (p.mode == 1) {
        if (lane < p.K) {
            p.msg[3 * frow + 0] = fm;
            p.msg[3 * frow + 1] = fs;
            p.msg[3 * frow + 2] = x_tok;
        }
        return;
    }
    bool flag = false;
    if (lane < p.K) {
        float lp;
        flag = finish_row(fm, fs, x_tok, lpd, uu, lp);
        p.lp_t[frow] = lp;
        p.accept[frow] = flag ? 1 : 0;
    }
    finish_sequence(flag, lane, p.K, b, p.n_acc, p.bits);
}

// combine all-gathered per-shard partials; one wave per sequence
__global__ __launch_bounds__(64) void k_accept_from_partials(const float* msg_all, int n_shards, const float* lp_d,
                                                             const float* u, int B, int K, float* lp_t,
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
        flag = finish_row(m2, s, g, lp_d[row], u[row], lp);
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

// Launch geometry.  Rows are independent, so the only questions are how many workgroups share a
// row and how many 16-byte loads each lane keeps in flight.  Policy ("verify kernel
// geometry"; numbers from the gpurun sweeps under profiles/):  aim for ~8 workgroups of 256
// lanes per CU, never cut a slice below one full unrolled batch per workgroup.
Geometry choose_geometry(int R, int K, int V, int dtype, int cus) {
    Geometry g{1, 256, 4, 1};
    const int64_t row_vecs = static_cast<int64_t>(V) * dtype_size(dtype) / 16;
    const int64_t batch = static_cast<int64_t>(g.threads) * g.unroll;
    int64_t want = (static_cast<int64_t>(cus) * 8 + R - 1) / (R > 0 ? R : 1);
    int64_t cap = row_vecs / batch;
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    const int smax = max_splits_for(K);
    if (want > smax) want = smax;
    if (want < 1) want = 1;
    g.splits = static_cast<int>(want);
    return g;
}

template <int DT, int THREADS, int UNROLL>
void launch_nt(const VerifyParams& p, int64_t grid, hipStream_t st, int nt) {
    if (nt)
        hipLaunchKernelGGL((k_verify<DT, THREADS, UNROLL, true>), dim3(static_cast<uint32_t>(grid)), dim3(THREADS), 0, st, p);
    else
        hipLaunchKernelGGL((k_verify<DT, THREADS, UNROLL, false>), dim3(static_cast<uint32_t>(grid)), dim3(THREADS), 0, st, p);
}

template <int DT, int THREADS>
int launch_unroll(const VerifyParams& p, int64_t grid, hipStream_t st, const Geometry& g) {
    switch (g.unroll) {
        case 2: launch_nt<DT, THREADS, 2>(p, grid, st, g.nt); return ASD_OK;
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

int launch_verify(VerifyParams p, int dtype, void* workspace, size_t workspace_bytes, void* stream, Geometry g) {
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

    const Geometry h = choose_geometry(static_cast<int>(R), p.K, p.V, dtype, current_device_cus());
    if (g.splits <= 0) g.splits = h.splits;
    if (g.threads <= 0) g.threads = h.threads;
    if (g.unroll <= 0) g.unroll = h.unroll;
    if (g.nt < 0) g.nt = h.nt;
    if (g.splits > max_splits_for(p.K)) return ASD_ERR_UNSUPPORTED;

    const size_t ticket_bytes = round_up(static_cast<size_t>(p.B) * sizeof(uint32_t), 256);
    const size_t region_bytes = round_up(static_cast<size_t>(p.K) * g.splits * sizeof(uint64_t), 256);
    if (workspace_bytes < ticket_bytes + region_bytes * static_cast<size_t>(p.B)) return ASD_ERR_WORKSPACE;
    p.S = g.splits;
    p.tickets = static_cast<uint32_t*>(workspace);
    p.granules = reinterpret_cast<uint64_t*>(static_cast<char*>(workspace) + ticket_bytes);
    p.region = static_cast<uint32_t>(region_bytes / sizeof(uint64_t));

    const int64_t grid = R * g.splits;
    if (grid > INT32_MAX) return ASD_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
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

ASD_EXPORT size_t asd_verify_accept_workspace_bytes(int B, int K, int V, int dtype) {
    (void)V;
    (void)dtype;
    if (B <= 0 || K <= 0) return 256;
    const size_t ticket_bytes = round_up(static_cast<size_t>(B) * sizeof(uint32_t), 256);
    const size_t region_bytes = round_up(static_cast<size_t>(K) * max_splits_for(K) * sizeof(uint64_t), 256);
    return ticket_bytes + region_bytes * static_cast<size_t>(B);
}

ASD_EXPORT int asd_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace) return ASD_ERR_INVALID_ARG;
    if (!aligned_to(workspace, 256)) return ASD_ERR_WORKSPACE;
    if (hipMemsetAsync(workspace, 0, workspace_bytes, static_cast<hipStream_t>(stream)) != hipSuccess) return ASD_ERR_HIP;
    return ASD_OK;
}

ASD_EXPORT int asd_verify_accept_tuned(const void* logits, int dtype, int64_t ld_row, const int32_t* tok,
                                       const float* lp_draft, const float* u, int B, int K, int V,
                                       float* lp_target, uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits,
                                       void* workspace, size_t workspace_bytes, void* stream, int splits,
                                       int threads, int unroll, int nontemporal) {
    if (B > 0 && K > 0 && (!lp_draft || !u || !lp_target || !accept || !n_acc)) return ASD_ERR_INVALID_ARG;
    VerifyParams p{};
    p.logits = logits; p.ld_row = ld_row; p.tok = tok; p.lp_d = lp_draft; p.u = u;
    p.B = B; p.K = K; p.V = V; p.v_offset = 0;
    p.lp_t = lp_target; p.accept = accept; p.n_acc = n_acc; p.bits = accept_bits;
    p.msg = nullptr; p.mode = 0;
    return launch_verify(p, dtype, workspace, workspace_bytes, stream, Geometry{splits, threads, unroll, nontemporal});
}

ASD_EXPORT int asd_verify_accept(const void* logits, int dtype, int64_t ld_row, const int32_t* tok,
                                 const float* lp_draft, const float* u, int B, int K, int V, float* lp_target,
                                 uint8_t* accept, int32_t* n_acc, uint64_t* accept_bits, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    return asd_verify_accept_tuned(logits, dtype, ld_row, tok, lp_draft, u, B, K, V, lp_target, accept, n_acc,
                                   accept_bits, workspace, workspace_bytes, stream, 0, 0, 0, -1);
}

ASD_EXPORT int asd_lse_partial(const void* logits_shard, int dtype, int64_t ld_row, const int32_t* tok, int B, int K,
                               int V_shard, int64_t v_offset, float* msg, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (B > 0 && K > 0 && !msg) return ASD_ERR_INVALID_ARG;
    VerifyParams p{};
    p.logits = logits_shard; p.ld_row = ld_row; p.tok = tok;
    p.B = B; p.K = K; p.V = V_shard; p.v_offset = v_offset;
    p.msg = msg; p.mode = 1;
    return launch_verify(p, dtype, workspace, workspace_bytes, stream, Geometry{0, 0, 0, -1});
}

ASD_EXPORT int asd_accept_from_partials(const float* msg_all, int n_shards, const float* lp_draft, const float* u,
                                        int B, int K, float* lp_target, uint8_t* accept, int32_t* n_acc,
                                        uint64_t* accept_bits, void* stream) {
    if (B < 0 || K < 0 || n_shards < 1) return ASD_ERR_INVALID_ARG;
    if (B == 0 || K == 0) return ASD_OK;
    if (K > ASD_MAX_DRAFT_LEN) return ASD_ERR_UNSUPPORTED;
    if (!msg_all || !lp_draft || !u || !lp_target || !accept || !n_acc) return ASD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_accept_from_partials, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), msg_all,
                       n_shards, lp_draft, u, B, K, lp_target, accept, n_acc, accept_bits);
    return launch_status();
}
