// lm_head_verify.hip -- SURVEY §8(f) N2: the target tier's lm_head projection fused with the verify
// pass, so that the [B, K, V] logits never reach HBM.
//
//   logits[m][v] = sum_k hidden[m][k] * weight[v][k]          (bf16 x bf16 -> f32, m = b*K + k)
//
// k_lm_head_tile<NTW>: one workgroup (8 waves as 4 x 2, one workgroup per CU) owns 64*NTW vocabulary
// columns x up to 256 rows; a wave keeps its 64 rows x 32*NTW columns of logits in MFMA accumulators
// (v_mfma_f32_32x32x16_bf16 with the WEIGHT tile as the A operand, so a lane holds 16 vocabulary entries
// of ONE row and the row reduction is lane-local).  At the end the accumulators are folded into the
// log2-domain partial (m2, s) of lse_device.hpp, logit[tok] is gathered where the block owns it, and the
// record (m2, s, g, arg-max value, arg-max id) of the block is written per row -- (m2, s, g) is the message
// asd_lse_partial emits for a vocabulary shard.  k_accept_from_blocks merges the blocks' records of every
// row and applies the accept rule (finish_row / finish_sequence: the code the streaming kernel ends in).
//
// Data movement, per 64 reduction columns ("superstage"):
// Both operands move as whole 128-byte lines by LDS-DMA (global_load_lds_dwordx4: 8 lanes per row, no
// staging registers, nothing for the compiler to mis-wait on):
//   weights  nt, 3-slot LDS ring: two superstages in flight while one is multiplied;
//   hidden   (L2 / MALL hits) 2-slot ring: one superstage in flight.
// (64-byte pieces per row and instruction -- a 32-column stage -- moved ~20 % fewer bytes per second.)
// With 256-column blocks the rings take 3 x 32 + 2 x 32 = 160 KiB: all of a CU's LDS.
// MFMA k-slot (ks, h, j) of a superstage is reduction column 32h + 8ks + j for BOTH operands (any
// bijection works as long as A and B agree), so a lane's fragment is one 16-byte segment 4h + ks of its
// row; the LDS image is swizzled (segment ^ (row >> 1) & 7, on the DMA source address and on the fragment
// read alike) so that ds_read_b128 is conflict-free.
//
// Column blocks: 256 wide while they fill whole rounds of the CUs; the remainder as 128-wide blocks (V = 152064
// on 256 CUs: 512 + 164 blocks = 2.5 rounds instead of 3) or, for deep reductions, as 256-wide blocks cut into
// reduction slices that meet through f32 slabs and a ticket (see the launcher).
//
// Algorithmic HBM bytes: V*D*2 (weights once per 256 rows) + M*D*2; flops: 2*M*D*V.
#define ASD_DPP_ASM_REDUCTIONS 1   // wave_max / wave_sum as one DPP instruction per step (lse_device.hpp)
#include "lse_device.hpp"

#ifndef ASD_LMHEAD_LAB
#define ASD_LMHEAD_LAB 0   // tools/lm_head_lab.py timing experiments: 1 = no loads in the loop, 2 = no math
#endif

namespace asd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// one 32 x 32 x 16 MFMA on fragments held as raw 16-byte vectors: bf16 or f16 operands (same lane maps, same LDS images --
// only this instruction differs between the two element types of the call)
template <bool F16>
__device__ __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int kBM = 256;        // rows (draft positions) per workgroup: 4 wave rows x 64
constexpr int kSuper = 64;      // reduction columns per superstage = one 128-byte line per row
constexpr int kThreads = 512;
constexpr int kMsg = 5;         // floats per (block, row) record: m2, s, g, arg-max value, arg-max id (int bits)
constexpr int kNoIndex = 0x7fffffff;
constexpr int kWRing = 3;       // LDS slots of one weight superstage each (two in flight, one multiplied)
constexpr int kHRing = 2;       // LDS slots of one hidden superstage each

struct LmHeadParams {
    const void* hidden;
    int64_t ld_h;
    const void* weight;
    int64_t ld_w;
    int D, M, V;
    const int32_t* tok;
    float c2;
    float* msg;          // [blocks of both launches][M][kMsg]
    int m_blocks;
    int n_blocks;        // column blocks of this launch
    int v_offset;        // global vocabulary id of column 0 (a vocabulary shard of a tensor-parallel lm_head)
    int col0;            // first vocabulary column of this launch
    int unit0;           // index of this launch's first block in msg (one record per row and block)
    int packed;          // != 0: `weight` is the tile-major image of asd_lm_head_pack_weights ([V/256][D/64][256 rows][64 cols])
    int need_argmax;     // the caller wants the row arg-max (argmax_out / greedy): otherwise the epilogue skips its bookkeeping
    int k_slices;        // > 1: every column block is cut into this many reduction slices (one workgroup each)
    float* slabs;        // [n_blocks][k_slices][8 waves][2 * NTW * 4][64 lanes] float4: partial accumulators
    uint32_t* tickets;   // [n_blocks], zero before the launch; the slice that draws k_slices - 1 finishes the block (and zeroes it)
    // STORE kernels (asd_linear: the same products written out as a [M][N] matrix instead of folded into log-sum-exp records)
    void* out;           // [M][ld_out], element type of the operands
    int64_t ld_out;
    const void* bias;    // [V] (the N of the linear layer), element type of the operands, or NULL
    const void* residual;   // [M][ld_res] added to the product (may alias out: every element is read, then written, by one lane), or NULL
    int64_t ld_res;
};

// ---- STORE epilogue: lane (r, h) holds, per 32 x 32 accumulator tile, 4 x 4 consecutive output columns
// n_first + 8 q .. + 3 (q = 0..3) of ONE row m: four 8-byte stores (the two half-waves' pieces of a row are adjacent, the
// four q of a tile make a 64-byte run; L2 merges them).  One rounding f32 -> bf16 / f16 (RNE) at the store; bias added in f32.
template <bool F16>
__device__ __forceinline__ uint32_t pack_pair(float a, float b) {
    if constexpr (F16) {
        const _Float16 x = static_cast<_Float16>(a), y = static_cast<_Float16>(b);
        return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, x)) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, y)) << 16);
    } else {
        const __bf16 x = static_cast<__bf16>(a), y = static_cast<__bf16>(b);
        return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, x)) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, y)) << 16);
    }
}
template <bool F16>
__device__ __forceinline__ float unpack_elem(uint32_t w, int hi) {
    const uint16_t u = static_cast<uint16_t>(hi ? (w >> 16) : (w & 0xffffu));
    if constexpr (F16) return static_cast<float>(__builtin_bit_cast(_Float16, u));
    else return __uint_as_float(static_cast<uint32_t>(u) << 16);
}
template <bool F16>
__device__ __forceinline__ void store_tile16(const f32x16& a, int m, int n_first, int slice, const LmHeadParams& p) {
    if (m >= p.M) return;
    if (p.k_slices > 1) {
        // a reduction slice: f32 partials into this slice's [M][V] slab; k_linear_reduce adds the slabs in slice order
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        float* const row = p.slabs + (static_cast<int64_t>(slice) * p.M + m) * p.V;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n_first + 8 * q;
            if (n >= p.V) continue;
            f32x4 v;
            v[0] = a[4 * q]; v[1] = a[4 * q + 1]; v[2] = a[4 * q + 2]; v[3] = a[4 * q + 3];
            *reinterpret_cast<f32x4*>(row + n) = v;
        }
        return;
    }
    char* const row = static_cast<char*>(p.out) + static_cast<int64_t>(m) * p.ld_out * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n_first + 8 * q;
        if (n >= p.V) continue;                      // N % 4 == 0 (checked by the launcher): a piece is all in or all out
        float v0 = a[4 * q], v1 = a[4 * q + 1], v2 = a[4 * q + 2], v3 = a[4 * q + 3];
        if (p.bias) {
            const uint2 bw = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.bias) + static_cast<int64_t>(n) * 2);
            v0 += unpack_elem<F16>(bw.x, 0); v1 += unpack_elem<F16>(bw.x, 1);
            v2 += unpack_elem<F16>(bw.y, 0); v3 += unpack_elem<F16>(bw.y, 1);
        }
        if (p.residual) {
            const uint2 rw = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.residual) +
                                                             (static_cast<int64_t>(m) * p.ld_res + n) * 2);
            v0 += unpack_elem<F16>(rw.x, 0); v1 += unpack_elem<F16>(rw.x, 1);
            v2 += unpack_elem<F16>(rw.y, 0); v3 += unpack_elem<F16>(rw.y, 1);
        }
        uint2 o;
        o.x = pack_pair<F16>(v0, v1);
        o.y = pack_pair<F16>(v2, v3);
        *reinterpret_cast<uint2*>(row + static_cast<int64_t>(n) * 2) = o;
    }
}

// the slabs of a sliced STORE launch -> out: thread = 4 consecutive columns of one row; slices added in slice order
template <bool F16>
__global__ __launch_bounds__(256) void k_linear_reduce(const float* __restrict__ slabs, int k_slices, int M, int N,
                                                       const void* __restrict__ bias, const void* residual, int64_t ld_res,
                                                       void* out, int64_t ld_out) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int n4 = N / 4;
    if (i >= static_cast<int64_t>(M) * n4) return;
    const int m = static_cast<int>(i / n4), n = static_cast<int>(i % n4) * 4;
    const float* src = slabs + static_cast<int64_t>(m) * N + n;
    const int64_t stride = static_cast<int64_t>(M) * N;
    f32x4 acc = *reinterpret_cast<const f32x4*>(src);
    for (int sl = 1; sl < k_slices; ++sl) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + sl * stride);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    if (bias) {
        const uint2 bw = *reinterpret_cast<const uint2*>(static_cast<const char*>(bias) + static_cast<int64_t>(n) * 2);
        acc[0] += unpack_elem<F16>(bw.x, 0); acc[1] += unpack_elem<F16>(bw.x, 1);
        acc[2] += unpack_elem<F16>(bw.y, 0); acc[3] += unpack_elem<F16>(bw.y, 1);
    }
    if (residual) {
        const uint2 rw = *reinterpret_cast<const uint2*>(static_cast<const char*>(residual) + (static_cast<int64_t>(m) * ld_res + n) * 2);
        acc[0] += unpack_elem<F16>(rw.x, 0); acc[1] += unpack_elem<F16>(rw.x, 1);
        acc[2] += unpack_elem<F16>(rw.y, 0); acc[3] += unpack_elem<F16>(rw.y, 1);
    }
    uint2 o;
    o.x = pack_pair<F16>(acc[0], acc[1]);
    o.y = pack_pair<F16>(acc[2], acc[3]);
    *reinterpret_cast<uint2*>(static_cast<char*>(out) + (static_cast<int64_t>(m) * ld_out + n) * 2) = o;
}

template <int PENDING>
__device__ __forceinline__ void wait_and_meet() {
    // lgkmcnt(0): this wave's LDS reads (the fragments it carries across the barrier) have returned, so the
    // slots it read may be refilled by anyone once the barrier is passed
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PENDING) : "memory");
}

// One 32 x 32 accumulator tile of a lane -- 16 vocabulary entries n = n_first + (i & 3) + 8 * (i >> 2) of ONE row -- folded
// into the row's running (m2, s), the drafted token's logit g and, if wanted, the arg-max (value, id; ties -> lowest id, NaN
// never wins, padding never wins a tie at -inf).  Selects, not branches: as short-circuit `if`s this was ~35 instructions and
// two exec-mask branches per logit -- 17 us per 256 x 256 block, 6-15 % of a block's time.
template <bool ARGMAX>
__device__ __forceinline__ void fold_tile16(const f32x16& a, int n_first, int tk, int V, float c2, float& m2, float& s, float& g,
                                            float& bv, int& bi) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = n_first + (i & 3) + 8 * (i >> 2);
        const bool valid = n < V;
        const float v = valid ? a[i] : -INFINITY;            // padded weight rows are not vocabulary
        g = (n == tk) ? v : g;
        if (ARGMAX) {
            const bool better = (v > bv) | ((v == bv) & (n < bi) & valid);
            bv = better ? v : bv;
            bi = better ? n : bi;
        }
        x[i] = v;
    }
    float lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { lo[i] = x[i]; hi[i] = x[8 + i]; }
    accum8(lo, c2, m2, s);
    accum8(hi, c2, m2, s);
}

// NTW: 32-column accumulator tiles per wave: 4 (256-column block) or 2 (128-column block).
// HPASSES: 64-row DMA passes over the hidden rows, ceil(rows / 64) for a call with M <= 256 rows (1, 2 or 4):
// at M = 64 three quarters of the hidden-state traffic into LDS would be padding.
// WM: wave rows (4: the lm_head form, 4 x 2 waves over 256 rows; STORE launches also run 2 x 4 waves over a row block of
// <= 128 rows and 1 x 8 waves over one of <= 64 rows, so that a short LAST row block -- M = 288 = 256 + 32, the K + 1 = 9
// positions of 32 sequences -- costs its share of MFMAs instead of a full block's).  A wave always owns 64 rows x 32 * NTW
// columns; the block is 32 * NTW * (8 / WM) columns wide.
// MT: 32-row accumulator tiles per wave (2 everywhere but the "tall" STORE form: 1 x 8 waves of 9 x 1 tiles = ONE row block
// of 288 rows x 256 columns -- the K + 1 = 9 positions of 32 sequences without a second, mostly empty row block; its hidden
// slot is 320 rows, so the weight ring is WRING = 2 slots: one superstage in flight, which its 2.5 us of MFMAs per superstage hide).
template <int NTW, int HPASSES, bool F16, bool STORE, int WM, int MT = 2, int WRING = kWRing>
__device__ __forceinline__ void tile_body(const LmHeadParams& p, unsigned char* const lds) {
    constexpr int WN = 8 / WM;               // wave columns
    constexpr int BN = 32 * NTW * WN;
    constexpr int kWSlot = BN * 128;         // one weight superstage
    constexpr int kRowBlock = WM * MT * 32 > kBM ? WM * MT * 32 : kBM;        // rows a workgroup owns
    constexpr int kHSlot = (64 * HPASSES > kBM ? 64 * HPASSES : kBM) * 128;   // one hidden superstage
    constexpr int WPASSES = BN / 64;         // DMA instructions per thread and superstage (64 rows per pass)
    unsigned char* const lds_h = lds + WRING * kWSlot;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = t >> 6;
    const int wm = wv % WM;    // wave row: hidden rows 64 * wm ..
    const int wn = wv / WM;    // wave column: weight rows 32 * NTW * wn ..
    const int r = lane & 31;
    const int h = lane >> 5;
    // M > 256: the row blocks that share a weight tile get ids 8 apart, i.e. the same XCD (workgroups are
    // dealt round-robin to the 8 XCDs) and the same dispatch round: one of them pulls the tile from HBM,
    // the others find it in that XCD's L2.  Column blocks past the last multiple of 8 keep the plain order.
    int mb, nb, slice = 0;
    {
        int id = static_cast<int>(blockIdx.x);
        if (p.k_slices > 1) {           // the slices of a block are neighbours: they finish at about the same time
            slice = id % p.k_slices;
            id /= p.k_slices;
        }
        const int group = 8 * p.m_blocks;
        const int swizzled = (p.n_blocks / 8) * group;
        if (id < swizzled) {
            const int in_group = id % group;
            nb = (id / group) * 8 + in_group % 8;
            mb = in_group / 8;
        } else {
            mb = (id - swizzled) % p.m_blocks;
            nb = (p.n_blocks / 8) * 8 + (id - swizzled) / p.m_blocks;
        }
    }
    const int n0 = p.col0 + nb * BN;
    const int m0 = mb * kRowBlock;
    const int rows_w = min(BN, p.V - n0);
    const int rows_h = min(kRowBlock, p.M - m0);

    // DMA sources: one instruction moves 8 rows x 128 B (lane -> row lane >> 3, 16-byte segment lane & 7,
    // swizzled).  Rows past a matrix edge re-read the last valid row (their products are masked in the
    // epilogue), so no lane ever addresses outside the operands.
    // Addresses are a block-uniform 64-bit base (+ the superstage's byte offset, scalar arithmetic) plus a
    // loop-invariant 32-bit lane offset, so the DMA instructions take the saddr + voffset form and the loop
    // spends no vector instructions on addressing (SQ counters: the DMA issue was ~8 % of an iteration).
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);   // wave id as a scalar: LDS destinations stay in SGPRs
    const int drow = wv * 8 + (lane >> 3);
    // weights as given ([V][ld_w]: a superstage of a block is 256 strided 128-byte lines, every one in another DRAM page)
    // or packed tile-major (asd_lm_head_pack_weights: the block's superstage is ONE contiguous 32 KiB run)
    const int64_t w_stage_stride = p.packed ? 256 * 128 : kSuper * 2;
    const uint32_t w_row_stride = p.packed ? 128u : static_cast<uint32_t>(p.ld_w * 2);
    const char* const wbase = p.packed
        ? static_cast<const char*>(p.weight) + (static_cast<int64_t>(n0 / 256) * (p.D / kSuper)) * (256 * 128) + static_cast<int64_t>(n0 % 256) * 128
        : static_cast<const char*>(p.weight) + static_cast<int64_t>(n0) * p.ld_w * 2;
    const char* const hbase = static_cast<const char*>(p.hidden) + static_cast<int64_t>(m0) * p.ld_h * 2;
    uint32_t woff[WPASSES], hoff[HPASSES];
#pragma unroll
    for (int ps = 0; ps < WPASSES; ++ps) {
        const int row = ps * 64 + drow;
        const int seg = (lane & 7) ^ ((row >> 1) & 7);
        woff[ps] = static_cast<uint32_t>(min(row, rows_w - 1)) * w_row_stride + seg * 16;
    }
#pragma unroll
    for (int ps = 0; ps < HPASSES; ++ps) {
        const int row = ps * 64 + drow;
        const int seg = (lane & 7) ^ ((row >> 1) & 7);
        hoff[ps] = static_cast<uint32_t>(min(row, rows_h - 1)) * static_cast<uint32_t>(p.ld_h * 2) + seg * 16;
    }
    // this workgroup's superstages [s_begin, n_super): all of them, or its reduction slice
    const int total_super = p.D / kSuper;
    const int s_begin = static_cast<int>(static_cast<int64_t>(slice) * total_super / p.k_slices);        // balanced: no empty slice
    const int n_super = static_cast<int>(static_cast<int64_t>(slice + 1) * total_super / p.k_slices);
    auto issue_w = [&](int stage) {
        const char* src = wbase + static_cast<int64_t>(stage) * w_stage_stride;
        asm volatile("" : "+s"(src));   // keep the base in SGPRs: without it LLVM folds the lane offset into a 64-bit VGPR pointer
        unsigned char* dst = lds + (stage % WRING) * kWSlot + wv_s * 1024;
#pragma unroll
        for (int ps = 0; ps < WPASSES; ++ps)
            __builtin_amdgcn_global_load_lds((glb_void*)(src + woff[ps]), (lds_void*)(dst + ps * 8192), 16, 0, 2);
    };
    auto issue_h = [&](int stage) {
        const char* src = hbase + static_cast<int64_t>(stage) * (kSuper * 2);
        asm volatile("" : "+s"(src));
        unsigned char* dst = lds_h + (stage & (kHRing - 1)) * kHSlot + wv_s * 1024;
#pragma unroll
        for (int ps = 0; ps < HPASSES; ++ps)
            __builtin_amdgcn_global_load_lds((glb_void*)(src + hoff[ps]), (lds_void*)(dst + ps * 8192), 16, 0, 0);
    };

    f32x16 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;

    // wave-uniform: a wave whose rows or columns are all padding only stages
    const bool wave_works = 32 * MT * wm < rows_h && 32 * NTW * wn < rows_w;
    const int key = (r >> 1) & 7;
    const int h_off = (32 * MT * wm + r) * 128;
    const int w_off = (32 * NTW * wn + r) * 128;
    // The fragment reads of k-step ks + 1 are issued before the MFMAs of k-step ks (two register sets; the
    // sched_barriers pin that order -- left alone the scheduler emits read, read, wait, mfma, mfma).  The LAST
    // k-step of a superstage is multiplied only after the next barrier: its fragments are in registers by
    // then, so the MFMA pipe has work while the wave issues the DMA of the coming superstages and waits for
    // the first fragments of the next one (the barrier -> first MFMA bubble was ~12 % of an iteration).
    bf16x8 wf[2][NTW], hf[2][MT];
    auto read_frags = [&](int S, int ks, int set) {
        const unsigned char* wb = lds + (S % WRING) * kWSlot + w_off;
        const unsigned char* hb = lds_h + (S & (kHRing - 1)) * kHSlot + h_off;
        const int so = ((4 * h + ks) ^ key) * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hf[set][mt] = *reinterpret_cast<const bf16x8*>(hb + mt * 32 * 128 + so);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) wf[set][nt] = *reinterpret_cast<const bf16x8*>(wb + nt * 32 * 128 + so);
    };
    auto multiply = [&](int set) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);   // MFMA issue ahead of the other wave's reads / DMA issue: -10 % (7B head)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                acc[mt][nt] = mfma32<F16>(wf[set][nt], hf[set][mt], acc[mt][nt]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // the same MFMAs as ONE scheduling region with the fragment reads of the next k-step issued just before them: a read
    // goes out behind each of the first 2 + NTW MFMAs instead of all of them in front
    auto multiply_interleaved = [&](int set) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                acc[mt][nt] = mfma32<F16>(wf[set][nt], hf[set][mt], acc[mt][nt]);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < MT * NTW; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < MT + NTW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // k-steps 0..2 of superstage S are multiplied; the fragments of k-step 3 are left in register set 1
    auto head = [&](int S, auto&& late_issue) {
        if (!wave_works) {
            late_issue();
            return;
        }
        read_frags(S, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(S, 1, 1);
        multiply(0);
        late_issue();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(S, 2, 0);
        multiply_interleaved(1);
        read_frags(S, 3, 1);
        multiply_interleaved(0);
    };
    auto tail = [&]() {
        if (wave_works) multiply(1);
    };

    // Issue order: prologue W(0), H(0), W(1); after the barrier that opens superstage S: H(S+1), W(S+2).
    // At that barrier the youngest loads are  H(S), W(S+1) : vmcnt(WPASSES) retires this wave's share of H(S)
    // and of everything older (W(S)) and leaves W(S+1) in flight; the barrier makes every wave's share visible.
    // The slots refilled after it (hidden: slot of S-1; weights: slot (S+2) % 3 = (S-1) % 3) were read into
    // registers -- k-step 3 included, see wait_and_meet's lgkmcnt(0) -- before their readers reached it.
    int S = s_begin;
    if constexpr (WRING == 2) {
        // two weight slots: W(S + 1) and H(S + 1) go out behind the barrier that opens superstage S (their slots held S - 1,
        // whose readers passed that barrier) and must have landed at the next one: vmcnt(0)
        if (S < n_super) { issue_w(S); issue_h(S); }
        for (; S < n_super; ++S) {
            wait_and_meet<0>();
            if (S > s_begin) tail();
            const bool more = S + 1 < n_super;
            if (more && wn < WN / 2) { issue_h(S + 1); issue_w(S + 1); }       // (the two waves of a SIMD issue at different points)
            head(S, [&] { if (more && wn >= WN / 2) { issue_h(S + 1); issue_w(S + 1); } });
        }
        if (n_super > s_begin) tail();
    } else {
#if ASD_LMHEAD_LAB & 1
    // lab, "math alone": no load in the loop, so EVERY ring slot is filled once with real operands up front (round 3: with one
    // weight slot and one hidden slot left as whatever the previous kernel had in LDS the variant's time depended on that
    // garbage -- 513 us in round 1, 1508 us under the counter passes of round 3)
    if (S + 2 < n_super) {
        issue_w(S + 2);
        issue_h(S + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#endif
    if (S < n_super) {
        issue_w(S);
        issue_h(S);
        if (S + 1 < n_super) issue_w(S + 1);
    }
    for (; S + 2 < n_super; ++S) {   // steady state: branch-free
        wait_and_meet<WPASSES>();
#if !(ASD_LMHEAD_LAB & 2)
        if (S > s_begin) tail();     // k-step 3 of superstage S-1
#endif
        // The two waves that share a SIMD (wave column 0 and 1) issue their DMA at different points of the
        // iteration -- right after the barrier, and behind the first MFMA group -- so that a DMA instruction
        // held up by a full memory pipeline never stops both of them from feeding the MFMA pipe (-1..-2.5 %;
        // moving the second point behind k-step 1 loses 4 %).
#if ASD_LMHEAD_LAB & 2
        issue_h(S + 1);
        issue_w(S + 2);
#elif ASD_LMHEAD_LAB & 1
        head(S, [] {});
#else
        if (wn == 0) { issue_h(S + 1); issue_w(S + 2); }
        head(S, [&] { if (wn != 0) { issue_h(S + 1); issue_w(S + 2); } });
#endif
    }
    for (; S < n_super; ++S) {       // the last two superstages
        if (S + 1 < n_super) wait_and_meet<WPASSES>();
        else wait_and_meet<0>();
#if !(ASD_LMHEAD_LAB & 2)
        if (S > s_begin) tail();
#endif
        if (S + 1 < n_super) issue_h(S + 1);
#if !(ASD_LMHEAD_LAB & 2)
        head(S, [] {});
#endif
    }
#if !(ASD_LMHEAD_LAB & 2)
    if (n_super > s_begin) tail();   // k-step 3 of the last superstage
#endif

    }
    // ---- reduction slices: every slice publishes its partial accumulators; the one that draws the last
    // ticket adds the others' and goes on to the epilogue.  Hand-off: plain 16-byte stores, every wave drains
    // them (vmcnt), workgroup barrier, ONE agent-scope release + ticket by lane 0; the finisher acquires once,
    // barrier, then plain loads.  The ticket word is zeroed by the launcher before every call.
    if (!STORE && p.k_slices > 1) {      // (a sliced STORE launch writes its partials as [slice][M][N] slabs: store_tile16)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        constexpr int kChunks = 2 * NTW * 4;                       // float4 chunks per lane
        const int unit = nb * p.m_blocks + mb;                     // (column block, row block): one ticket, k_slices slabs
        const int64_t block_slabs = static_cast<int64_t>(unit) * p.k_slices;
        auto slab = [&](int sl) {
            return reinterpret_cast<f32x4*>(p.slabs) + ((block_slabs + sl) * 8 + wv) * (kChunks * 64) + lane;
        };
        {
            f32x4* out = slab(slice);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
                        v[0] = acc[mt][nt][4 * q]; v[1] = acc[mt][nt][4 * q + 1];
                        v[2] = acc[mt][nt][4 * q + 2]; v[3] = acc[mt][nt][4 * q + 3];
                        out[((mt * NTW + nt) * 4 + q) * 64] = v;
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t* const flag = reinterpret_cast<uint32_t*>(lds);   // the rings are idle now
        if (t == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t old = __hip_atomic_fetch_add(p.tickets + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old == static_cast<uint32_t>(p.k_slices - 1);
            if (last) {
                __hip_atomic_store(p.tickets + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every slice has drawn: zero for the next call
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flag = last ? 1u : 0u;
        }
        __syncthreads();
        if (*flag == 0u) return;
        // every slice is added from its slab in slice order, the finisher's own included: the sum does not
        // depend on which slice happened to arrive last (bit-reproducible results)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;
        for (int sl = 0; sl < p.k_slices; ++sl) {
            const f32x4* in = slab(sl);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = in[((mt * NTW + nt) * 4 + q) * 64];
                        acc[mt][nt][4 * q] += v[0]; acc[mt][nt][4 * q + 1] += v[1];
                        acc[mt][nt][4 * q + 2] += v[2]; acc[mt][nt][4 * q + 3] += v[3];
                    }
        }
    }
    // ---- epilogue: D[vocab row][m column]; lane (r, h) holds row m and, per 32-column tile, the 16
    // vocabulary ids  n = tile + (i & 3) + 8 * (i >> 2) + 4 * h.  A wave folds its 32 * NTW columns per row,
    // the two wave columns meet in LDS (free now: every DMA was retired by the last wait), and the block
    // writes ONE record per row: (m2, s, g) and the block's arg-max (value, vocabulary id; ties -> lowest id,
    // NaN logits never win).
    if constexpr (STORE) {
        if (wave_works) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    store_tile16<F16>(acc[mt][nt], m0 + 32 * MT * wm + 32 * mt + r, n0 + 32 * NTW * wn + 32 * nt + 4 * h, slice, p);
        }
        return;
    }
    if constexpr (WM == 1) {
        // the "tall" form (ONE row block of 32 MT rows, eight wave columns of 32 NTW columns; asd_lm_head_verify for 256 < M <= 288):
        // every wave folds its columns of every row, the eight partial records of a row meet in LDS and one thread per row merges
        // them in wave order
        constexpr int kRows = 32 * MT;
        float* const meet8 = reinterpret_cast<float*>(lds);   // [8 wave columns][kRows][kMsg]  (the rings are idle now)
        __syncthreads();                                       // the last superstage's fragment reads are done
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + 32 * mt + r;
            const int tk = (wave_works && m < p.M && p.tok[m] >= p.v_offset) ? p.tok[m] - p.v_offset : -1;
            float m2 = kSentinel, s = 0.0f, g = -INFINITY, bv = -INFINITY;
            int bi = kNoIndex;
            if (wave_works) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int n_first = n0 + 32 * NTW * wn + 32 * nt + 4 * h;
                    if (p.need_argmax) fold_tile16<true>(acc[mt][nt], n_first, tk, p.V, p.c2, m2, s, g, bv, bi);
                    else fold_tile16<false>(acc[mt][nt], n_first, tk, p.V, p.c2, m2, s, g, bv, bi);
                }
            }
            const float m2o = __shfl_xor(m2, 32, 64);          // the row's other columns of these tiles sit in lane r ^ 32
            const float so = __shfl_xor(s, 32, 64);
            const float go = __shfl_xor(g, 32, 64);
            const float bvo = __shfl_xor(bv, 32, 64);
            const int bio = __shfl_xor(bi, 32, 64);
            ms_merge(m2, s, m2o, so);
            g = (g != g) ? g : ((go != go) ? go : fmaxf(g, go));
            if (bvo > bv || (bvo == bv && bio < bi)) { bv = bvo; bi = bio; }
            if (h == 0) {
                float* q = meet8 + (wn * kRows + 32 * mt + r) * kMsg;
                q[0] = m2; q[1] = s; q[2] = g; q[3] = bv; q[4] = __int_as_float(bi);
            }
        }
        __syncthreads();
        if (t < kRows && m0 + t < p.M) {
            float m2 = kSentinel, s = 0.0f, g = -INFINITY, bv = -INFINITY;
            int bi = kNoIndex;
            for (int w = 0; w < WN; ++w) {
                const float* q = meet8 + (w * kRows + t) * kMsg;
                ms_merge(m2, s, q[0], q[1]);
                const float go = q[2];
                g = (g != g) ? g : ((go != go) ? go : fmaxf(g, go));
                const float bvo = q[3];
                const int bio = __float_as_int(q[4]);
                if (bvo > bv || (bvo == bv && bio < bi)) { bv = bvo; bi = bio; }
            }
            float* out = p.msg + (static_cast<int64_t>(p.unit0 + nb) * p.M + (m0 + t)) * kMsg;
            out[0] = m2; out[1] = s; out[2] = g; out[3] = bv; out[4] = __int_as_float(bi);
        }
        return;
    }
    float* const meet = reinterpret_cast<float*>(lds);   // [256 rows][kMsg]
    __syncthreads();                                      // the last superstage's fragment reads are done
    float tm2[2], ts[2], tg[2], tbv[2];
    int tbi[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + 64 * wm + 32 * mt + r;
        const int tk = (wave_works && m < p.M && p.tok[m] >= p.v_offset) ? p.tok[m] - p.v_offset : -1;
        float m2 = kSentinel, s = 0.0f, g = -INFINITY, bv = -INFINITY;
        int bi = kNoIndex;
        if (wave_works) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n_first = n0 + 32 * NTW * wn + 32 * nt + 4 * h;
                if (p.need_argmax) fold_tile16<true>(acc[mt][nt], n_first, tk, p.V, p.c2, m2, s, g, bv, bi);
                else fold_tile16<false>(acc[mt][nt], n_first, tk, p.V, p.c2, m2, s, g, bv, bi);
            }
        }
        // the row's other columns of these tiles sit in lane r ^ 32
        const float m2o = __shfl_xor(m2, 32, 64);
        const float so = __shfl_xor(s, 32, 64);
        const float go = __shfl_xor(g, 32, 64);
        const float bvo = __shfl_xor(bv, 32, 64);
        const int bio = __shfl_xor(bi, 32, 64);
        ms_merge(m2, s, m2o, so);
        g = (g != g) ? g : ((go != go) ? go : fmaxf(g, go));   // a NaN logit must not be dropped by max
        if (bvo > bv || (bvo == bv && bio < bi)) { bv = bvo; bi = bio; }
        tm2[mt] = m2; ts[mt] = s; tg[mt] = g; tbv[mt] = bv; tbi[mt] = bi;
        if (wn == 1 && h == 0) {
            float* q = meet + (64 * wm + 32 * mt + r) * kMsg;
            q[0] = m2; q[1] = s; q[2] = g; q[3] = bv; q[4] = __int_as_float(bi);
        }
    }
    __syncthreads();
    if (wn != 0 || h != 0) return;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + 64 * wm + 32 * mt + r;
        if (m >= p.M) continue;
        const float* q = meet + (64 * wm + 32 * mt + r) * kMsg;
        float m2 = tm2[mt], s = ts[mt], g = tg[mt], bv = tbv[mt];
        int bi = tbi[mt];
        ms_merge(m2, s, q[0], q[1]);
        const float go = q[2];
        g = (g != g) ? g : ((go != go) ? go : fmaxf(g, go));
        const float bvo = q[3];
        const int bio = __float_as_int(q[4]);
        if (bvo > bv || (bvo == bv && bio < bi)) { bv = bvo; bi = bio; }
        float* out = p.msg + (static_cast<int64_t>(p.unit0 + nb) * p.M + m) * kMsg;
        out[0] = m2;
        out[1] = s;
        out[2] = g;
        out[3] = bv;
        out[4] = __int_as_float(bi);
    }
}

template <int NTW, int HPASSES, bool F16>
__global__ __launch_bounds__(kThreads, 1) void k_lm_head_tile(LmHeadParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kWRing * 64 * NTW * 128 + kHRing * kBM * 128];
    tile_body<NTW, HPASSES, F16, false, 4>(p, lds);
}

// asd_linear, M > 64: every row block of 256 rows in the 4 x 2 form, a last row block of <= 128 / <= 64 rows in the 2 x 4 /
// 1 x 8 form -- in ONE launch, so that the row blocks of a weight tile still share it through the XCD's L2
template <bool F16>
__global__ __launch_bounds__(kThreads, 1) void k_linear_tile(LmHeadParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kWRing * 256 * 128 + kHRing * kBM * 128];
    // (the row block of this workgroup, as tile_body decodes it)
    int id = static_cast<int>(blockIdx.x) / p.k_slices;
    const int group = 8 * p.m_blocks;
    const int swizzled = (p.n_blocks / 8) * group;
    const int mb = id < swizzled ? (id % group) / 8 : (id - swizzled) % p.m_blocks;
    const int rows_h = min(kBM, p.M - mb * kBM);
    if (rows_h > 128) tile_body<4, 4, F16, true, 4>(p, lds);
    else if (rows_h > 64) tile_body<2, 2, F16, true, 2>(p, lds);
    else tile_body<1, 1, F16, true, 1>(p, lds);
}

// asd_linear, 256 < M <= 288: ONE row block of 288 rows, 1 x 8 waves of 9 x 1 tiles (see tile_body's MT)
template <bool F16, int MT>
__global__ __launch_bounds__(kThreads, 1) void k_linear_tall(LmHeadParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * 256 * 128 + kHRing * 320 * 128];
    tile_body<1, (32 * MT + 63) / 64, F16, true, 1, MT, 2>(p, lds);
}

// asd_lm_head_verify, 256 < M <= 288 (B = 33 ... 36 at K = 8; K + 1 = 9 positions of 32 sequences): the same ONE row block with the
// log-sum-exp epilogue -- a second row block would stream the whole matrix again for a handful of rows (72B head: 1 197 us at
// M = 264 against 685 us at M = 256)
template <bool F16, int MT>
__global__ __launch_bounds__(kThreads, 1) void k_lm_head_tall(LmHeadParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * 256 * 128 + kHRing * 320 * 128];
    tile_body<1, (32 * MT + 63) / 64, F16, false, 1, MT, 2>(p, lds);
}

// ---- M > 256 rows (several row blocks share every weight tile): FOUR waves per workgroup, 128 x 128 logits per wave.
// With the weights re-read from L2 by the row blocks of a tile the call is MFMA-bound, and the 8-wave kernel above spends
// its issue slots on fragment reads: 6 ds_read_b128 per 8 MFMAs and wave, two waves per SIMD taking turns, 0.94-0.97 PF.
// Here a wave owns 16 accumulator tiles (256 AGPRs; 16 MFMAs per 8 fragment reads), the operands are staged
// global -> VGPR -> ds_write_b128 into a double-buffered XOR-swizzled LDS image (plain loads and stores: the compiler counts
// vmcnt / lgkmcnt itself -- the steady state is branch-free so that it CAN count), and every k-step is ONE scheduling region
// in which sched_group_barrier deals the 8 fragment reads of the next k-step, 4 staging stores and 4 staging loads between
// the 16 MFMAs, so the wave never issues a long run of non-MFMA instructions while the matrix pipe drains.  The last k-step
// of a superstage is multiplied behind the barrier that ends it, over the first fragment reads of the next one.
// tools/lab_gemm4.hip is this loop as a stand-alone program (profiles/r02_lab_gemm4.txt: 0.92 PF without the pinned
// read-ahead, 1.00 with it, 1.09 with the interleave, 1.25 with the XCD-aware block order -- the library GEMM's rate).
constexpr int kQThreads = 256;

constexpr int kQSlot = kBM * 128;             // one operand, one superstage: 256 rows x 128 bytes

// The body of one 256-column tile.  WM x WN waves, MT x NT accumulator tiles of 32 x 32 per wave:
//   2 x 2 waves, 4 x 4 tiles   a full row block (256 rows)
//   1 x 4 waves, 4 x 2 tiles   the last row block when it holds <= 128 rows  (half the MFMAs)
//   1 x 4 waves, 2 x 2 tiles   ...                       <= 64 rows           (a quarter)
// The partial forms live in the SAME launch as the full ones: the row blocks of a weight tile run side by side on one XCD
// and share the tile through its L2 -- a separate launch for the last row block would stream the weights from HBM again.
template <int WM, int MT, int NT, bool F16, bool STORE>
__device__ __forceinline__ void quad_tile(const LmHeadParams& p, int mb, int nb, int slice, unsigned char* lds) {
    constexpr int kSlot = kQSlot;
    constexpr int WN = 4 / WM;
    static_assert(WM * MT * 32 <= kBM && WN * NT * 32 == 256, "tile shape");
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv % WM, wn = wv / WM;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = p.col0 + nb * 256, m0 = mb * kBM;
    const int rows_w = min(256, p.V - n0), rows_h = min(kBM, p.M - m0);
    const int64_t w_stage_stride = p.packed ? 256 * 128 : kSuper * 2;
    const uint32_t w_row_stride = p.packed ? 128u : static_cast<uint32_t>(p.ld_w * 2);
    const char* const wbase = p.packed
        ? static_cast<const char*>(p.weight) + (static_cast<int64_t>(n0 / 256) * (p.D / kSuper)) * (256 * 128)
        : static_cast<const char*>(p.weight) + static_cast<int64_t>(n0) * p.ld_w * 2;
    const char* const hbase = static_cast<const char*>(p.hidden) + static_cast<int64_t>(m0) * p.ld_h * 2;
    // staging piece ps * 256 + t: row id >> 3, 16-byte segment id & 7.  Rows past a matrix edge re-read the last valid row
    // (their products are masked in the epilogue): no lane addresses outside the operands.
    uint32_t woff[8], hoff[8], loff[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int id = ps * 256 + t;
        const int row = id >> 3, seg = id & 7;
        woff[ps] = static_cast<uint32_t>(min(row, rows_w - 1)) * w_row_stride + seg * 16;
        hoff[ps] = static_cast<uint32_t>(min(row, rows_h - 1)) * static_cast<uint32_t>(p.ld_h * 2) + seg * 16;
        loff[ps] = row * 128 + ((seg ^ ((row >> 1) & 7)) * 16);
    }
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    u32x4v sw[8], sh[8];
    // this workgroup's superstages: all of them, or (sliced STORE launches) its balanced share; S below counts from s_first
    const int total_super = p.D / kSuper;
    const int s_first = static_cast<int>(static_cast<int64_t>(slice) * total_super / p.k_slices);
    const int n_super = static_cast<int>(static_cast<int64_t>(slice + 1) * total_super / p.k_slices) - s_first;
    auto load_piece = [&](int S, int ps) {
        sw[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(wbase + static_cast<int64_t>(s_first + S) * w_stage_stride + woff[ps]));
        sh[ps] = *reinterpret_cast<const u32x4v*>(hbase + static_cast<int64_t>(s_first + S) * (kSuper * 2) + hoff[ps]);
    };
    auto store_piece = [&](int buf, int ps) {
        unsigned char* wb = lds + buf * 2 * kSlot;
        *reinterpret_cast<u32x4v*>(wb + loff[ps]) = sw[ps];
        *reinterpret_cast<u32x4v*>(wb + kSlot + loff[ps]) = sh[ps];
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;
    // wave-uniform (made scalar: a uniform jump, not an exec mask around the k-steps): a wave of padding only stages
    const bool wave_works = __builtin_amdgcn_readfirstlane((32 * MT * wm < rows_h && 32 * NT * wn < rows_w) ? 1 : 0) != 0;
    const int key = (r >> 1) & 7;
    const int w_off = (32 * NT * wn + r) * 128, h_off = (32 * MT * wm + r) * 128;
    bf16x8 wf[2][NT], hf[2][MT];
    auto read_frags = [&](int buf, int ks, int set) {
        const unsigned char* wb = lds + buf * 2 * kSlot + w_off;
        const unsigned char* hb = lds + buf * 2 * kSlot + kSlot + h_off;
        const int so = ((4 * h + ks) ^ key) * 16;
#pragma unroll
        for (int i = 0; i < NT; ++i) wf[set][i] = *reinterpret_cast<const bf16x8*>(wb + i * 32 * 128 + so);
#pragma unroll
        for (int i = 0; i < MT; ++i) hf[set][i] = *reinterpret_cast<const bf16x8*>(hb + i * 32 * 128 + so);
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = mfma32<F16>(wf[set][nt], hf[set][mt], acc[mt][nt]);
    };
    // one scheduling region: MT * NT MFMAs, MT + NT DS reads, 4 DS writes, 4 VMEM reads (masks 0x008 / 0x100 / 0x200 / 0x020),
    // dealt one non-MFMA instruction behind each MFMA while they last (reads first: the next k-step needs them soonest)
    auto interleave = [&](bool stores, bool loads) {
        constexpr int kMfma = MT * NT, kReads = MT + NT;
#pragma unroll
        for (int i = 0; i < kMfma; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < kReads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (i < kReads + 4) { if (stores) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
            else if (i < kReads + 8) { if (loads) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // a quarter of the staging: two pieces of stage S + 1 go to the other LDS buffer, the same registers take stage S + 2
    auto restage = [&](int S, int part, bool store, bool load) {
#pragma unroll
        for (int ps = 2 * part; ps < 2 * part + 2; ++ps) {
            if (store) store_piece((S + 1) & 1, ps);
            if (load) load_piece(S + 2, ps);
        }
    };
    // `first` / `store` / `load` are compile-time constants at every call: the loop below is branch-free in steady state.
    // (A wave of padding rows / columns only stages; the test is hoisted so that a working wave's k-step -- reads, MFMAs,
    // staging -- stays ONE basic block, which is what sched_group_barrier orders.)
    auto superstage = [&](int S, bool first, bool store, bool load) {
        const int buf = S & 1;
        if (wave_works) {
            read_frags(buf, 0, 0);
            if (!first) multiply(1);                   // (S - 1, k-step 3): its fragments were read before the barrier
            restage(S, 0, store, load);
            interleave(store, load);
            read_frags(buf, 1, 1);
            multiply(0);
            restage(S, 1, store, load);
            interleave(store, load);
            read_frags(buf, 2, 0);
            multiply(1);
            restage(S, 2, store, load);
            interleave(store, load);
            read_frags(buf, 3, 1);
            multiply(0);
            restage(S, 3, store, load);
            interleave(store, load);
        } else {
#pragma unroll
            for (int part = 0; part < 4; ++part) restage(S, part, store, load);
        }
        __syncthreads();
    };
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) load_piece(0, ps);
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) store_piece(0, ps);
    __syncthreads();
    if (n_super >= 3) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) load_piece(1, ps);
        superstage(0, true, true, true);
        int S = 1;
        for (; S + 2 < n_super; ++S) superstage(S, false, true, true);
        superstage(S, false, true, false);
        superstage(S + 1, false, false, false);
    } else if (n_super == 2) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) load_piece(1, ps);
        superstage(0, true, true, false);
        superstage(1, false, false, false);
    } else {
        superstage(0, true, false, false);
    }
    if (wave_works) multiply(1);                       // k-step 3 of the last superstage

    // ---- epilogue (as k_lm_head_tile): D[vocab row][m column]; lane (r, h) holds row m of four 32-row tiles and, per
    // 32-column tile, the vocabulary ids n = tile + (i & 3) + 8 * (i >> 2) + 4 * h.  A wave folds its 128 columns per row,
    // the two wave columns meet in LDS, the block writes ONE record per row.
    if constexpr (STORE) {
        if (wave_works) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    store_tile16<F16>(acc[mt][nt], m0 + 32 * MT * wm + 32 * mt + r, n0 + 32 * NT * wn + 32 * nt + 4 * h, slice, p);
        }
        return;
    }
    float* const meet = reinterpret_cast<float*>(lds);   // [WN - 1 wave columns][256 rows][kMsg]; every fragment read was waited for
    float tm2[MT], ts[MT], tg[MT], tbv[MT];
    int tbi[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = 32 * MT * wm + 32 * mt + r;
        const int m = m0 + row;
        const int tk = (wave_works && m < p.M && p.tok[m] >= p.v_offset) ? p.tok[m] - p.v_offset : -1;
        float m2 = kSentinel, s = 0.0f, g = -INFINITY, bv = -INFINITY;
        int bi = kNoIndex;
        if (wave_works) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n_first = n0 + 32 * NT * wn + 32 * nt + 4 * h;
                if (p.need_argmax) fold_tile16<true>(acc[mt][nt], n_first, tk, p.V, p.c2, m2, s, g, bv, bi);
                else fold_tile16<false>(acc[mt][nt], n_first, tk, p.V, p.c2, m2, s, g, bv, bi);
            }
        }
        const float m2o = __shfl_xor(m2, 32, 64);
        const float so = __shfl_xor(s, 32, 64);
        const float go = __shfl_xor(g, 32, 64);
        const float bvo = __shfl_xor(bv, 32, 64);
        const int bio = __shfl_xor(bi, 32, 64);
        ms_merge(m2, s, m2o, so);
        g = (g != g) ? g : ((go != go) ? go : fmaxf(g, go));   // a NaN logit must not be dropped by max
        if (bvo > bv || (bvo == bv && bio < bi)) { bv = bvo; bi = bio; }
        tm2[mt] = m2; ts[mt] = s; tg[mt] = g; tbv[mt] = bv; tbi[mt] = bi;
        if (wn != 0 && h == 0) {
            float* q = meet + ((wn - 1) * kBM + row) * kMsg;
            q[0] = m2; q[1] = s; q[2] = g; q[3] = bv; q[4] = __int_as_float(bi);
        }
    }
    __syncthreads();
    if (wn != 0 || h != 0) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = 32 * MT * wm + 32 * mt + r;
        const int m = m0 + row;
        if (m >= p.M) continue;
        float m2 = tm2[mt], s = ts[mt], g = tg[mt], bv = tbv[mt];
        int bi = tbi[mt];
#pragma unroll
        for (int w = 0; w < WN - 1; ++w) {             // the other wave columns, in a fixed order
            const float* q = meet + (w * kBM + row) * kMsg;
            ms_merge(m2, s, q[0], q[1]);
            const float go = q[2];
            g = (g != g) ? g : ((go != go) ? go : fmaxf(g, go));
            const float bvo = q[3];
            const int bio = __float_as_int(q[4]);
            if (bvo > bv || (bvo == bv && bio < bi)) { bv = bvo; bi = bio; }
        }
        float* out = p.msg + (static_cast<int64_t>(p.unit0 + nb) * p.M + m) * kMsg;
        out[0] = m2;
        out[1] = s;
        out[2] = g;
        out[3] = bv;
        out[4] = __int_as_float(bi);
    }
}

template <bool F16, bool STORE = false>
__global__ __launch_bounds__(kQThreads, 1) void k_lm_head_quad(LmHeadParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * 2 * kQSlot];    // [buffer][weights | hidden]
    int mb, nb, slice = 0;
    {   // the row blocks that share a weight tile get ids 8 apart: one XCD, one dispatch round (see k_lm_head_tile)
        int id = static_cast<int>(blockIdx.x);
        if (p.k_slices > 1) {           // (STORE launches only) the slices of a tile are neighbours
            slice = id % p.k_slices;
            id /= p.k_slices;
        }
        const int group = 8 * p.m_blocks;
        const int swizzled = (p.n_blocks / 8) * group;
        if (id < swizzled) {
            const int in_group = id % group;
            nb = (id / group) * 8 + in_group % 8;
            mb = in_group / 8;
        } else {
            mb = (id - swizzled) % p.m_blocks;
            nb = (p.n_blocks / 8) * 8 + (id - swizzled) / p.m_blocks;
        }
    }
    const int rows_h = min(kBM, p.M - mb * kBM);       // block-uniform: only the last row block can be partial
    if (rows_h > 128) quad_tile<2, 4, 4, F16, STORE>(p, mb, nb, slice, lds);
    else if (rows_h > 64) quad_tile<1, 4, 2, F16, STORE>(p, mb, nb, slice, lds);
    else quad_tile<1, 2, 2, F16, STORE>(p, mb, nb, slice, lds);
}

// ---- M <= 64 rows (B*K <= 64: BASELINE configs[1], batch 8 x draft_len 8): the call is HBM-bound (64 flop per weight byte),
// so the tile is shaped for the STREAM, not for the MFMA pipe: the 8 waves sit side by side on the 256 vocabulary columns of a
// block (32 columns each, all 64 rows), the hidden slot shrinks to 8 KiB, and the LDS that frees goes into deeper rings --
// 4 weight slots + 3 hidden slots (152 KiB).  Loads are issued  H(S+2), W(S+3)  behind the barrier that opens superstage S;
// because a wave's vmcnt retires in order, the wait for H(S) leaves W(S+1), H(S+1), W(S+2) in flight (9 instructions):
// TWO whole weight superstages per CU stay outstanding at every barrier instead of one in the 256-row kernel.
constexpr int kSkWRing = 4;
constexpr int kSkHRing = 3;
constexpr int kSkRows = 64;

template <bool F16, bool STORE = false>
__global__ __launch_bounds__(kThreads, 1) void k_lm_head_skinny(LmHeadParams p) {
    constexpr int BN = 256;
    constexpr int kWSlot = BN * 128;
    constexpr int kHSlot = kSkRows * 128;
    constexpr int WPASSES = BN / 64;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kSkWRing * kWSlot + kSkHRing * kHSlot];
    unsigned char* const lds_h = lds + kSkWRing * kWSlot;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = t >> 6;          // wave column: vocabulary rows 32 * wv .. of the block
    const int r = lane & 31;
    const int h = lane >> 5;
    // reduction slices (STORE launches of narrow matrices: 14 column blocks cannot feed 256 CUs): the slices of a block are
    // neighbours in the grid, slice i owns the superstages [i * total / k, (i + 1) * total / k)
    const int nb = static_cast<int>(blockIdx.x) / p.k_slices;
    const int slice = static_cast<int>(blockIdx.x) % p.k_slices;
    const int n0 = p.col0 + nb * BN;
    const int rows_w = min(BN, p.V - n0);
    const int rows_h = min(kSkRows, p.M);

    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    const int drow = wv * 8 + (lane >> 3);
    const int64_t w_stage_stride = p.packed ? 256 * 128 : kSuper * 2;
    const uint32_t w_row_stride = p.packed ? 128u : static_cast<uint32_t>(p.ld_w * 2);
    const char* const wbase = p.packed
        ? static_cast<const char*>(p.weight) + (static_cast<int64_t>(n0 / 256) * (p.D / kSuper)) * (256 * 128)
        : static_cast<const char*>(p.weight) + static_cast<int64_t>(n0) * p.ld_w * 2;
    const char* const hbase = static_cast<const char*>(p.hidden);
    uint32_t woff[WPASSES], hoff;
#pragma unroll
    for (int ps = 0; ps < WPASSES; ++ps) {
        const int row = ps * 64 + drow;
        const int seg = (lane & 7) ^ ((row >> 1) & 7);
        woff[ps] = static_cast<uint32_t>(min(row, rows_w - 1)) * w_row_stride + seg * 16;
    }
    {
        const int seg = (lane & 7) ^ ((drow >> 1) & 7);
        hoff = static_cast<uint32_t>(min(drow, rows_h - 1)) * static_cast<uint32_t>(p.ld_h * 2) + seg * 16;
    }
    const int total_super = p.D / kSuper;
    const int s_begin = static_cast<int>(static_cast<int64_t>(slice) * total_super / p.k_slices);
    const int n_super = static_cast<int>(static_cast<int64_t>(slice + 1) * total_super / p.k_slices);
    auto issue_w = [&](int stage) {
        const char* src = wbase + static_cast<int64_t>(stage) * w_stage_stride;
        asm volatile("" : "+s"(src));
        unsigned char* dst = lds + (stage % kSkWRing) * kWSlot + wv_s * 1024;
#pragma unroll
        for (int ps = 0; ps < WPASSES; ++ps)
            __builtin_amdgcn_global_load_lds((glb_void*)(src + woff[ps]), (lds_void*)(dst + ps * 8192), 16, 0, 2);
    };
    auto issue_h = [&](int stage) {
        const char* src = hbase + static_cast<int64_t>(stage) * (kSuper * 2);
        asm volatile("" : "+s"(src));
        unsigned char* dst = lds_h + (stage % kSkHRing) * kHSlot + wv_s * 1024;
        __builtin_amdgcn_global_load_lds((glb_void*)(src + hoff), (lds_void*)dst, 16, 0, 0);
    };

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;
    const bool wave_works = 32 * wv < rows_w;
    const int key = (r >> 1) & 7;
    const int h_off = r * 128;
    const int w_off = (32 * wv + r) * 128;
    bf16x8 wf[2], hf[2][2];
    auto read_frags = [&](int S, int ks, int set) {
        const unsigned char* wb = lds + (S % kSkWRing) * kWSlot + w_off;
        const unsigned char* hb = lds_h + (S % kSkHRing) * kHSlot + h_off;
        const int so = ((4 * h + ks) ^ key) * 16;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) hf[set][mt] = *reinterpret_cast<const bf16x8*>(hb + mt * 32 * 128 + so);
        wf[set] = *reinterpret_cast<const bf16x8*>(wb + so);
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            acc[mt] = mfma32<F16>(wf[set], hf[set][mt], acc[mt]);
    };
    auto compute = [&](int S) {
        if (!wave_works) return;
        read_frags(S, 0, 0);
        read_frags(S, 1, 1);
        multiply(0);
        read_frags(S, 2, 0);
        multiply(1);
        read_frags(S, 3, 1);
        multiply(0);
        multiply(1);
    };
    // prologue in steady-state order: W(0) | H(0) W(1) | H(1) W(2)
    int S = s_begin;
    if (S < n_super) issue_w(S);
    if (S < n_super) issue_h(S);
    if (S + 1 < n_super) issue_w(S + 1);
    if (S + 1 < n_super) issue_h(S + 1);
    if (S + 2 < n_super) issue_w(S + 2);
    for (; S + 3 < n_super; ++S) {          // steady state: W(S+1), H(S+1), W(S+2) stay in flight across the barrier
        wait_and_meet<2 * WPASSES + 1>();
        issue_h(S + 2);
        issue_w(S + 3);
        compute(S);
    }
    for (; S < n_super; ++S) {              // the last three superstages of the block: drain
        wait_and_meet<0>();
        if (S + 2 < n_super) issue_h(S + 2);
        compute(S);
    }

    if constexpr (STORE) {
        if (wave_works) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                store_tile16<F16>(acc[mt], 32 * mt + r, n0 + 32 * wv + 4 * h, slice, p);
        }
        return;
    }
    // ---- epilogue: per wave 64 rows x 32 columns -> (m2, s, g, arg-max); waves 1..7 hand theirs to wave 0 through LDS
    float* const meet = reinterpret_cast<float*>(lds);   // [7 waves][64 rows][kMsg]
    __syncthreads();
    float tm2[2], ts[2], tg[2], tbv[2];
    int tbi[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 32 * mt + r;
        const int tk = (wave_works && m < p.M && p.tok[m] >= p.v_offset) ? p.tok[m] - p.v_offset : -1;
        float m2 = kSentinel, s = 0.0f, g = -INFINITY, bv = -INFINITY;
        int bi = kNoIndex;
        if (wave_works) {
            const int n_first = n0 + 32 * wv + 4 * h;
            if (p.need_argmax) fold_tile16<true>(acc[mt], n_first, tk, p.V, p.c2, m2, s, g, bv, bi);
            else fold_tile16<false>(acc[mt], n_first, tk, p.V, p.c2, m2, s, g, bv, bi);
        }
        const float m2o = __shfl_xor(m2, 32, 64);
        const float so = __shfl_xor(s, 32, 64);
        const float go = __shfl_xor(g, 32, 64);
        const float bvo = __shfl_xor(bv, 32, 64);
        const int bio = __shfl_xor(bi, 32, 64);
        ms_merge(m2, s, m2o, so);
        g = (g != g) ? g : ((go != go) ? go : fmaxf(g, go));
        if (bvo > bv || (bvo == bv && bio < bi)) { bv = bvo; bi = bio; }
        tm2[mt] = m2; ts[mt] = s; tg[mt] = g; tbv[mt] = bv; tbi[mt] = bi;
        if (wv != 0 && h == 0) {
            float* q = meet + ((wv - 1) * kSkRows + m) * kMsg;
            q[0] = m2; q[1] = s; q[2] = g; q[3] = bv; q[4] = __int_as_float(bi);
        }
    }
    __syncthreads();
    if (wv != 0 || h != 0) return;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 32 * mt + r;
        if (m >= p.M) continue;
        float m2 = tm2[mt], s = ts[mt], g = tg[mt], bv = tbv[mt];
        int bi = tbi[mt];
        for (int w = 0; w < 7; ++w) {          // fixed order: deterministic
            const float* q = meet + (w * kSkRows + m) * kMsg;
            ms_merge(m2, s, q[0], q[1]);
            const float go = q[2];
            g = (g != g) ? g : ((go != go) ? go : fmaxf(g, go));
            const float bvo = q[3];
            const int bio = __float_as_int(q[4]);
            if (bvo > bv || (bvo == bv && bio < bi)) { bv = bvo; bi = bio; }
        }
        float* out = p.msg + (static_cast<int64_t>(p.unit0 + nb) * p.M + m) * kMsg;
        out[0] = m2;
        out[1] = s;
        out[2] = g;
        out[3] = bv;
        out[4] = __int_as_float(bi);
    }
}

// merge the per-block records of every row of sequence b, then the accept rule.  Up to 16 waves; wave w
// takes draft positions w, w + waves, ...; its lanes stride over the blocks (fixed order: deterministic).
// greedy != 0: accept[b,k] = (tok[b,k] == argmax_v logits[b,k,v]) instead of the sampling test
// (lp_d / u unused); argmax_out (may be NULL) receives the row arg-max either way (-1: no finite logit).
// emit != NULL: write the merged (m2, s, g) per row ([B,K,3], the asd_lse_partial message) and stop there.
__global__ __launch_bounds__(1024) void k_accept_from_blocks(const float* msg, int n_blocks, const float* lp_d,
                                                            const float* u, const int32_t* tok, int greedy, int B, int K,
                                                            float c2, float* lp_t, uint8_t* accept, int32_t* n_acc,
                                                            uint64_t* bits, int32_t* argmax_out, float* emit) {
    __shared__ float red[ASD_MAX_DRAFT_LEN][3];
    __shared__ int best[ASD_MAX_DRAFT_LEN];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int64_t M = static_cast<int64_t>(B) * K;
    for (int k = w; k < K; k += waves) {
        const int64_t row = static_cast<int64_t>(b) * K + k;
        float m2 = kSentinel, s = 0.0f, g = -INFINITY, bv = -INFINITY;
        int bi = kNoIndex;
        bool gnan = false;
        for (int j = lane; j < n_blocks; j += 64) {
            const float* t = msg + (static_cast<int64_t>(j) * M + row) * kMsg;
            ms_merge(m2, s, t[0], t[1]);
            const float gj = t[2];
            gnan = gnan || (gj != gj);
            g = fmaxf(g, gj);
            const float bvj = t[3];
            const int bij = __float_as_int(t[4]);
            if (bvj > bv || (bvj == bv && bij < bi)) { bv = bvj; bi = bij; }
        }
        wave_merge(m2, s);
        g = wave_max(g);
        if (__ballot(gnan) != 0ull) g = NAN;
        const float top = wave_max(bv);
        int cand = (bv == top) ? bi : kNoIndex;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 64));
        if (lane == 0) {
            red[k][0] = m2;
            red[k][1] = s;
            red[k][2] = g;
            best[k] = cand == kNoIndex ? -1 : cand;
        }
    }
    __syncthreads();
    if (w != 0) return;
    if (emit) {   // vocabulary shard of a tensor-parallel lm_head: hand the merged (m2, s, g) to the all-gather
        if (lane < K) {
            float* out = emit + (static_cast<int64_t>(b) * K + lane) * 3;
            out[0] = red[lane][0];
            out[1] = red[lane][1];
            out[2] = red[lane][2];
        }
        return;
    }
    bool flag = false;
    if (lane < K) {
        const int64_t row = static_cast<int64_t>(b) * K + lane;
        float lp;
        const float lpd = greedy ? 0.0f : lp_d[row];
        const double lu = greedy ? 0.0 : log_u(u[row]);
        flag = finish_row(red[lane][0], red[lane][1], red[lane][2], c2, lpd, lu, lp);
        if (greedy) flag = best[lane] >= 0 && tok[row] == best[lane];
        lp_t[row] = lp;
        accept[row] = flag ? 1 : 0;
        if (argmax_out) argmax_out[row] = best[lane];
    }
    finish_sequence(flag, lane, K, b, n_acc, bits);
}

// [V][ld_w] bf16 -> tile-major [ceil(V/256)][D/64][256][64]: thread = one 16-byte segment; rows >= V are zero
__global__ __launch_bounds__(256) void k_pack_lm_head(const char* __restrict__ w, int64_t ld_w, int V, int D,
                                                      char* __restrict__ out, int64_t n_segments) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n_segments) return;
    const int n_super = D / kSuper;
    const int seg = static_cast<int>(i & 7);                 // 8 segments of 16 B per 128-byte row piece
    const int64_t line = i >> 3;                              // (block, superstage, row)
    const int row = static_cast<int>(line & 255);
    const int64_t bs = line >> 8;
    const int st = static_cast<int>(bs % n_super);
    const int64_t blk = bs / n_super;
    const int64_t v = blk * 256 + row;
    u32x4 val = {0u, 0u, 0u, 0u};
    if (v < V) val = *reinterpret_cast<const u32x4*>(w + (v * ld_w + static_cast<int64_t>(st) * kSuper) * 2 + seg * 16);
    *reinterpret_cast<u32x4*>(out + i * 16) = val;
}

// upper bound of the column blocks of one call (all of them narrow)
inline int max_blocks_for(int V) { return (V + 127) / 128; }

}  // namespace
}  // namespace asd

using namespace asd;

namespace {
constexpr size_t kSlabBytes = 8 * 32 * 64 * 16;   // one 256-column block's accumulators: 256 KiB
constexpr size_t kTicketBytes = 4096;             // up to 1024 tail blocks

size_t records_bytes(int B, int K, int V) {
    return round_up(static_cast<size_t>(max_blocks_for(V)) * static_cast<size_t>(B) * K * kMsg * sizeof(float), 256);
}
// reduction slices of the tail blocks never outnumber the CUs: one slab each
size_t slab_budget() {
    const int cus = current_device_cus();
    return static_cast<size_t>(cus > 0 ? cus : 512) * kSlabBytes;
}
}  // namespace

ASD_EXPORT size_t asd_lm_head_verify_workspace_bytes(int B, int K, int V) {
    if (B <= 0 || K <= 0 || V <= 0) return 0;
    return records_bytes(B, K, V) + kTicketBytes + slab_budget();
}

namespace {
template <int NTW, bool F16>
void launch_tile_t(int h_passes, dim3 grid, hipStream_t st, const LmHeadParams& p) {
    if (h_passes == 1) hipLaunchKernelGGL((k_lm_head_tile<NTW, 1, F16>), grid, dim3(kThreads), 0, st, p);
    else if (h_passes == 2) hipLaunchKernelGGL((k_lm_head_tile<NTW, 2, F16>), grid, dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL((k_lm_head_tile<NTW, 4, F16>), grid, dim3(kThreads), 0, st, p);
}
template <int NTW>
void launch_tile(bool f16, int h_passes, dim3 grid, hipStream_t st, const LmHeadParams& p) {
    if (f16) launch_tile_t<NTW, true>(h_passes, grid, st, p);
    else launch_tile_t<NTW, false>(h_passes, grid, st, p);
}

struct LmHeadCall {
    const void* hidden; int64_t ld_h; const void* weight; int64_t ld_w; int dtype, D;
    const int32_t* tok; const float* lp_draft; const float* u; int B, K, V; int64_t v_offset; float inv_temperature;
    int greedy; float* lp_target; uint8_t* accept; int32_t* n_acc; uint64_t* accept_bits; int32_t* argmax_out;
    float* emit; void* workspace; size_t workspace_bytes; void* stream;
};

int lm_head_launch(const LmHeadCall& c) {
    if (c.B < 0 || c.K < 0 || c.V < 1 || c.D < 1 || c.v_offset < 0) return ASD_ERR_INVALID_ARG;
    if (!(c.inv_temperature > 0.0f) || !(c.inv_temperature < 3.0e38f)) return ASD_ERR_INVALID_ARG;
    if (c.B == 0 || c.K == 0) return ASD_OK;
    if (c.K > ASD_MAX_DRAFT_LEN) return ASD_ERR_UNSUPPORTED;
    if ((c.dtype != ASD_DTYPE_BF16 && c.dtype != ASD_DTYPE_F16) || c.D % kSuper != 0 || c.v_offset + c.V >= (1ll << 31))
        return ASD_ERR_UNSUPPORTED;
    const bool f16 = c.dtype == ASD_DTYPE_F16;     // hidden states and weights share the element type
    if (!c.hidden || !c.weight || !c.tok) return ASD_ERR_INVALID_ARG;
    if (c.emit == nullptr) {
        if (!c.lp_target || !c.accept || !c.n_acc) return ASD_ERR_INVALID_ARG;
        if (!c.greedy && (!c.lp_draft || !c.u)) return ASD_ERR_INVALID_ARG;
    }
    const bool packed = c.ld_w == 0;       // ld_w == 0: `weight` is the image written by asd_lm_head_pack_weights
    if (c.ld_h < c.D || (!packed && c.ld_w < c.D)) return ASD_ERR_INVALID_ARG;
    if (!aligned_to(c.hidden, 16) || !aligned_to(c.weight, 16) || c.ld_h % 8 != 0 || c.ld_w % 8 != 0) return ASD_ERR_ALIGNMENT;
    if (!c.workspace || c.workspace_bytes < asd_lm_head_verify_workspace_bytes(c.B, c.K, c.V)) return ASD_ERR_WORKSPACE;
    if (!aligned_to(c.workspace, 16)) return ASD_ERR_ALIGNMENT;

    const int64_t M = static_cast<int64_t>(c.B) * c.K;
    const int64_t m_blocks = (M + kBM - 1) / kBM;
    if (M >= (1ll << 31) || m_blocks * max_blocks_for(c.V) >= (1ll << 31)) return ASD_ERR_UNSUPPORTED;
    // 256-column blocks while they fill whole rounds of the CUs, 128-column blocks for the rest
    const int64_t cus = current_device_cus();
    const int64_t wide = (static_cast<int64_t>(c.V / 256) * m_blocks / cus) * cus / m_blocks;
    const int tail_col = static_cast<int>(wide * 256);
    const int64_t rem = static_cast<int64_t>(c.V) - tail_col;
    int64_t narrow = (rem + 127) / 128;
    // The tail -- the columns that do not fill a whole round of 256-column blocks -- as 256-column blocks cut
    // into reduction slices, one workgroup each, when that fills the CUs better than 128-column blocks do
    // (V = 152064, 256 CUs: 82 blocks x 3 slices = 246 workgroups of a third of the depth instead of 164
    // workgroups of half the width).  The slices meet through f32 slabs and a ticket per block.  Only for deep
    // reductions (>= 24 superstages per slice): measured -3 % at D = 8192, -2 % at 5120, +0..2 % at 3584, where a
    // slice is too short to amortise its pipeline ramp and the slab traffic.
    const int64_t tail_tiles = (rem + 255) / 256;
    int64_t slices = tail_tiles > 0 ? cus / tail_tiles : 0;
    if (slices > 4) slices = 4;
    while (slices >= 2 && c.D / kSuper < 24 * slices) --slices;   // a slice needs >= 24 superstages to pay (see below)
    const bool split_k = m_blocks == 1 && M > 128 && slices >= 2 && tail_tiles * slices > narrow &&
                         tail_tiles <= static_cast<int64_t>(kTicketBytes / 4);
    LmHeadParams p{};
    p.hidden = c.hidden; p.ld_h = c.ld_h; p.weight = c.weight; p.ld_w = c.ld_w;
    p.D = c.D; p.M = static_cast<int>(M); p.V = c.V; p.tok = c.tok;
    p.v_offset = static_cast<int>(c.v_offset);
    p.c2 = static_cast<float>(1.4426950408889634074 * static_cast<double>(c.inv_temperature));
    p.msg = static_cast<float*>(c.workspace);
    p.m_blocks = static_cast<int>(m_blocks);
    p.packed = packed ? 1 : 0;
    p.k_slices = 1;
    p.need_argmax = (c.argmax_out != nullptr || c.greedy) ? 1 : 0;
    hipStream_t st = static_cast<hipStream_t>(c.stream);
    const int hp = M <= 64 ? 1 : (M <= 128 ? 2 : 4);
    if (M <= kSkRows) {   // the stream-shaped kernel: every 256-column block, one record per block
        const int64_t blocks = (static_cast<int64_t>(c.V) + 255) / 256;
        p.col0 = 0;
        p.unit0 = 0;
        p.n_blocks = static_cast<int>(blocks);
        if (f16) hipLaunchKernelGGL(k_lm_head_skinny<true>, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, st, p);
        else hipLaunchKernelGGL(k_lm_head_skinny<false>, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, st, p);
        hipLaunchKernelGGL(k_accept_from_blocks, dim3(c.B), dim3(64 * (c.K < 16 ? c.K : 16)), 0, st, p.msg,
                           static_cast<int>(blocks), c.lp_draft, c.u, c.tok, c.greedy ? 1 : 0, c.B, c.K, p.c2,
                           c.lp_target, c.accept, c.n_acc, c.accept_bits, c.argmax_out, c.emit);
        return launch_status();
    }
    if (M > kBM && M <= 288) {   // ONE tall row block (1 x 8 waves of 9 x 1 tiles) over every 256-column block
        const int64_t blocks = (static_cast<int64_t>(c.V) + 255) / 256;
        p.col0 = 0;
        p.unit0 = 0;
        p.m_blocks = 1;
        p.n_blocks = static_cast<int>(blocks);
        if (f16) hipLaunchKernelGGL((k_lm_head_tall<true, 9>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, st, p);
        else hipLaunchKernelGGL((k_lm_head_tall<false, 9>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, st, p);
        hipLaunchKernelGGL(k_accept_from_blocks, dim3(c.B), dim3(64 * (c.K < 16 ? c.K : 16)), 0, st, p.msg,
                           static_cast<int>(blocks), c.lp_draft, c.u, c.tok, c.greedy ? 1 : 0, c.B, c.K, p.c2,
                           c.lp_target, c.accept, c.n_acc, c.accept_bits, c.argmax_out, c.emit);
        return launch_status();
    }
    if (m_blocks >= 2) {   // several row blocks per weight tile: MFMA-bound, the 4-wave kernel over every 256-column block
        const int64_t blocks = (static_cast<int64_t>(c.V) + 255) / 256;
        // (Cutting the tiles of the last, partial round of the CUs into reduction slices -- V = 152064, M = 1024: 2376 tiles =
        // 9.28 rounds -- was built and measured: no gain at D = 8192, +4 % at D = 3584.  The 72 tiles of that round run
        // faster than a tile of a full round, and a second launch + the slab traffic cost what the slices save.)
        p.col0 = 0;
        p.unit0 = 0;
        p.n_blocks = static_cast<int>(blocks);
        if (f16) hipLaunchKernelGGL(k_lm_head_quad<true>, dim3(static_cast<unsigned>(blocks * m_blocks)), dim3(kQThreads), 0, st, p);
        else hipLaunchKernelGGL(k_lm_head_quad<false>, dim3(static_cast<unsigned>(blocks * m_blocks)), dim3(kQThreads), 0, st, p);
        hipLaunchKernelGGL(k_accept_from_blocks, dim3(c.B), dim3(64 * (c.K < 16 ? c.K : 16)), 0, st, p.msg,
                           static_cast<int>(blocks), c.lp_draft, c.u, c.tok, c.greedy ? 1 : 0, c.B, c.K, p.c2,
                           c.lp_target, c.accept, c.n_acc, c.accept_bits, c.argmax_out, c.emit);
        return launch_status();
    }
    if (wide > 0) {
        p.col0 = 0;
        p.unit0 = 0;
        p.n_blocks = static_cast<int>(wide);
        launch_tile<4>(f16, hp, dim3(static_cast<unsigned>(wide * m_blocks)), st, p);
    }
    if (split_k) {
        char* const base = static_cast<char*>(c.workspace) + records_bytes(c.B, c.K, c.V);
        p.tickets = reinterpret_cast<uint32_t*>(base);
        p.slabs = reinterpret_cast<float*>(base + kTicketBytes);
        if (hipMemsetAsync(p.tickets, 0, static_cast<size_t>(tail_tiles) * 4, st) != hipSuccess) return ASD_ERR_HIP;
        p.col0 = tail_col;
        p.unit0 = static_cast<int>(wide);
        p.n_blocks = static_cast<int>(tail_tiles);
        p.k_slices = static_cast<int>(slices);
        launch_tile<4>(f16, hp, dim3(static_cast<unsigned>(tail_tiles * slices)), st, p);
        narrow = tail_tiles;          // blocks that wrote a record
    } else if (narrow > 0) {
        p.col0 = tail_col;
        p.unit0 = static_cast<int>(wide);
        p.n_blocks = static_cast<int>(narrow);
        launch_tile<2>(f16, hp, dim3(static_cast<unsigned>(narrow * m_blocks)), st, p);
    }
    hipLaunchKernelGGL(k_accept_from_blocks, dim3(c.B), dim3(64 * (c.K < 16 ? c.K : 16)), 0, st, p.msg,
                       static_cast<int>(wide + narrow), c.lp_draft, c.u, c.tok, c.greedy ? 1 : 0, c.B, c.K, p.c2,
                       c.lp_target, c.accept, c.n_acc, c.accept_bits, c.argmax_out, c.emit);
    return launch_status();
}
}  // namespace

ASD_EXPORT int asd_lm_head_verify_ex(const void* hidden, int64_t ld_h, const void* weight, int64_t ld_w, int dtype, int D,
                                     const int32_t* tok, const float* lp_draft, const float* u, int B, int K, int V,
                                     float inv_temperature, int greedy, float* lp_target, uint8_t* accept,
                                     int32_t* n_acc, uint64_t* accept_bits, int32_t* argmax_out, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    return lm_head_launch(LmHeadCall{hidden, ld_h, weight, ld_w, dtype, D, tok, lp_draft, u, B, K, V, 0, inv_temperature,
                                     greedy, lp_target, accept, n_acc, accept_bits, argmax_out, nullptr, workspace,
                                     workspace_bytes, stream});
}

ASD_EXPORT int asd_lm_head_verify(const void* hidden, int64_t ld_h, const void* weight, int64_t ld_w, int dtype, int D,
                                  const int32_t* tok, const float* lp_draft, const float* u, int B, int K, int V,
                                  float inv_temperature, float* lp_target, uint8_t* accept, int32_t* n_acc,
                                  uint64_t* accept_bits, void* workspace, size_t workspace_bytes, void* stream) {
    return asd_lm_head_verify_ex(hidden, ld_h, weight, ld_w, dtype, D, tok, lp_draft, u, B, K, V, inv_temperature, 0,
                                 lp_target, accept, n_acc, accept_bits, nullptr, workspace, workspace_bytes, stream);
}

ASD_EXPORT int asd_lm_head_partial(const void* hidden, int64_t ld_h, const void* weight_shard, int64_t ld_w, int dtype,
                                   int D, const int32_t* tok, int B, int K, int V_shard, int64_t v_offset,
                                   float inv_temperature, float* msg, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (B > 0 && K > 0 && !msg) return ASD_ERR_INVALID_ARG;
    return lm_head_launch(LmHeadCall{hidden, ld_h, weight_shard, ld_w, dtype, D, tok, nullptr, nullptr, B, K, V_shard,
                                     v_offset, inv_temperature, 0, nullptr, nullptr, nullptr, nullptr, nullptr, msg,
                                     workspace, workspace_bytes, stream});
}

ASD_EXPORT size_t asd_lm_head_packed_bytes(int V, int D) {
    if (V <= 0 || D <= 0 || D % kSuper != 0) return 0;
    return static_cast<size_t>((V + 255) / 256) * static_cast<size_t>(D / kSuper) * 256 * 128;
}

ASD_EXPORT int asd_lm_head_pack_weights(const void* weight, int64_t ld_w, int dtype, int V, int D, void* packed,
                                        size_t packed_bytes, void* stream) {
    if (V < 1 || D < 1 || !weight || !packed || ld_w < D) return ASD_ERR_INVALID_ARG;
    if ((dtype != ASD_DTYPE_BF16 && dtype != ASD_DTYPE_F16) || D % kSuper != 0) return ASD_ERR_UNSUPPORTED;   // 2-byte elements: the re-layout moves bytes
    if (!aligned_to(weight, 16) || !aligned_to(packed, 16) || ld_w % 8 != 0) return ASD_ERR_ALIGNMENT;
    const size_t need = asd_lm_head_packed_bytes(V, D);
    if (packed_bytes < need) return ASD_ERR_WORKSPACE;
    const int64_t n_seg = static_cast<int64_t>(need / 16);
    const int64_t blocks = (n_seg + 255) / 256;
    if (blocks >= (1ll << 31)) return ASD_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_pack_lm_head, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char*>(weight), ld_w, V, D, static_cast<char*>(packed), n_seg);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// asd_linear: y[M][N] = x[M][D] . w[N][D]^T (+ bias) -- the projections of the decoder layers around the path (the callers'
// side, X3), on the SAME three kernels with the STORE epilogue:
//   M <= 64    k_lm_head_skinny (stream-shaped: the call is HBM-bound)     column blocks x reduction slices
//   M  > 64    k_linear_tile (8 waves; 256-row blocks as 4 x 2 waves, a last block of <= 128 / <= 64 rows as 2 x 4 / 1 x 8)
//              row blocks x column blocks x reduction slices
//   M  > 256 and an unsliced grid of >= 2 x CUs:  k_lm_head_quad (4 waves x 128 x 128, the MFMA-bound form)
// A layer's matrices are narrow (N = 3584 .. 57344: 14 .. 224 column blocks of 256) next to the lm_head's 594, so the
// reduction is cut into slices until the grid fills the CUs; the slices write f32 partials ([slice][M][N], L2 / MALL
// resident) and k_linear_reduce -- or, inside asd_decoder_forward, the kernel that consumes the product (asd_linear_partial) --
// adds them in slice order (bit-reproducible), adds bias and residual and rounds once.
namespace {
// test / lab switches: process-global and not thread-safe, so they exist in the TEST build of the library only
// (-DASD_TEST_HOOKS -> lib/libasd_hip_test.so); the product library has the constants
#ifdef ASD_TEST_HOOKS
int g_linear_tall = 1;     // asd_debug_linear_tall(0): 256 < M <= 288 as 256 + 32 rows (A/B measurements)
int g_force_linear_slices = 0;
#else
constexpr int g_linear_tall = 1;
constexpr int g_force_linear_slices = 0;
#endif
struct LinearPlan {
    int kind;        // 0 skinny, 1 tile, 2 quad, 3 tall (one 288-row block)
    int k_slices;
    int64_t units;   // workgroups before slicing
};

LinearPlan linear_plan(int M, int N, int D) {
    LinearPlan pl{};
    const int64_t blocks = (static_cast<int64_t>(N) + 255) / 256;
    const int64_t m_blocks = (static_cast<int64_t>(M) + kBM - 1) / kBM;
    const int64_t cus = current_device_cus() > 0 ? current_device_cus() : 256;
    // M > 64: the 8-wave kernel with reduction slices (a short last row block runs the 2 x 4 / 1 x 8 wave forms); the 4-wave
    // kernel only where its unsliced grid fills the CUs twice over (it is the MFMA-bound form: 0.95-1.0 PF on the lm_head)
    pl.kind = M <= kSkRows ? 0 : ((m_blocks >= 2 && blocks * m_blocks >= 2 * cus) ? 2 : 1);
    pl.units = blocks * m_blocks;
    pl.k_slices = 1;
    if ((M > kBM && M <= 288 && g_linear_tall) || (M > 192 && M <= kBM && g_linear_tall >= 2)) {   // one 224 / 256 / 288-row block per column block
        pl.kind = 3;
        pl.units = blocks;
    }
    // cost of a plan in superstages per CU: rounds x (superstages of a slice + pipeline fill) (+ the slab round trip)
    const int total = D / kSuper;
    // Cost of a plan in superstage-times of one CU.  A workgroup of a sliced launch pays its share of the reduction, the
    // pipeline fill, and the slab round trip: rows x 256 columns x 4 B written and read again = rows / 16 superstages' worth
    // of bytes (2 at M = 32, 16 for a full 256-row block -- slicing a wide, tall product costs more than it balances).
    // Makespan: whole rounds of the FULL workgroups (a CU holds one), or the total work spread over the CUs if that is
    // more; a last row block of <= 64 / <= 128 rows counts as a quarter / half of a workgroup.
    const int fill = 3;
    const int rows_eff = pl.kind == 3 ? (M + 31) / 32 * 32 : (M < kBM ? M : kBM);
    const int slab = (rows_eff + 15) / 16;
    int64_t full_units = pl.units, rem_units = 0, rem_quarters = 0;
    if (pl.kind == 1 || pl.kind == 2) {
        const int rem = M - static_cast<int>(m_blocks - 1) * kBM;
        if (rem <= 128) {
            full_units = blocks * (m_blocks - 1);
            rem_units = blocks;
            rem_quarters = rem <= 64 ? 1 : 2;
        }
    }
    // (measured, tools/sweep_linear_slices.py -> profiles/r03_linear_slices_*.json: while the partials stay in the 256 MB MALL
    // beside the weight stream their round trip costs a quarter of that; and whoever adds them walks the slices one after the
    // other -- 0.45 us, three quarters of a superstage-time, per slice)
    double best = -1.0;
    // (stream-shaped kernel with more than half a round of column blocks: unsliced -- forced counts measured level or worse
    // there, even counts 8-15 % worse: gate|up at M = 32, profiles/r03_linear_slices_*.json)
    const int max_slices = (pl.kind == 0 && pl.units * 2 > cus) ? 1 : 32;
    for (int sl = 1; sl <= max_slices && sl <= total; ++sl) {
        const double slab_bytes = static_cast<double>(sl) * M * static_cast<double>(N) * 4.0;
        const double slab_cost = sl > 1 ? (slab_bytes > 128.0e6 ? slab : (slab >= 4 ? slab / 4.0 : 1.0)) : 0.0;
        const double t_unit = static_cast<double>((total + sl - 1) / sl + fill) + slab_cost;
        const double by_rounds = static_cast<double>((full_units * sl + cus - 1) / cus) * t_unit;
        const double by_work = (static_cast<double>(full_units) + 0.25 * rem_quarters * static_cast<double>(rem_units)) * sl * t_unit / cus;
        double cost = by_rounds > by_work ? by_rounds : by_work;
        if (full_units == 0) cost = static_cast<double>((rem_units * sl + cus - 1) / cus) * t_unit;   // one short row block only
        if (pl.kind == 0) {
            // the stream-shaped kernel is HBM-bound: a partial last round is not a whole round (its workgroups share the
            // bandwidth the idle CUs leave) -- the lm_head's 594 blocks run 2.3, not 3, rounds; only the fill is per round
            const double wgs = static_cast<double>(pl.units) * sl;
            const double share = wgs > static_cast<double>(cus) ? wgs / static_cast<double>(cus) : 1.0;
            cost = share * static_cast<double>((total + sl - 1) / sl) + static_cast<double>((pl.units * sl + cus - 1) / cus) * (fill + slab_cost);
        }
        if (sl > 1) cost += 0.75 * sl;
        if (best < 0.0 || cost < best) { best = cost; pl.k_slices = sl; }
    }
    return pl;
}
}  // namespace

ASD_EXPORT size_t asd_linear_workspace_bytes(int M, int N, int D) {
    if (M <= 0 || N <= 0 || D <= 0 || D % kSuper != 0) return 0;
    const LinearPlan pl = linear_plan(M, N, D);
    const size_t slabs = pl.k_slices > 1 ? static_cast<size_t>(pl.k_slices) * M * static_cast<size_t>(N) * sizeof(float) : 0;
    return round_up(slabs, 256) + 256;
}

ASD_EXPORT int asd_linear_slices(int M, int N, int D) {      // the plan's slice count (a pure query: sizing, tests, tools)
    if (M <= 0 || N <= 0 || D <= 0 || D % kSuper != 0) return 0;
    return linear_plan(M, N, D).k_slices;
}

#ifdef ASD_TEST_HOOKS
ASD_EXPORT int asd_debug_linear_tall(int on) {            // returns the previous value
    const int old = g_linear_tall;
    g_linear_tall = on < 0 ? 0 : (on > 2 ? 2 : on);
    return old;
}
ASD_EXPORT int asd_debug_force_linear_slices(int k) {      // 0: the plan's own choice; returns the previous value
    const int old = g_force_linear_slices;
    g_force_linear_slices = k < 0 ? 0 : k;
    return old;
}
#endif

namespace {
// k_slices_out != NULL: a sliced plan stops after the product -- the f32 partials stay in `workspace` as [k_slices][M][N]
// and the caller's own kernel adds them (bias and residual are then the caller's too); an unsliced plan stores y as usual.
int linear_run(const void* x, int64_t ld_x, const void* w, int64_t ld_w, const void* bias, const void* residual,
               int64_t ld_res, int dtype, int M, int N, int D, void* y, int64_t ld_y, void* workspace,
               size_t workspace_bytes, void* stream, int* k_slices_out) {
    if (k_slices_out) *k_slices_out = 1;
    if (M < 0 || N < 1 || D < 1) return ASD_ERR_INVALID_ARG;
    if (residual && (ld_res < N || ld_res % 4 != 0 || !aligned_to(residual, 8))) return ld_res < N ? ASD_ERR_INVALID_ARG : ASD_ERR_ALIGNMENT;
    if (M == 0) return ASD_OK;
    if ((dtype != ASD_DTYPE_BF16 && dtype != ASD_DTYPE_F16) || D % kSuper != 0 || N % 4 != 0) return ASD_ERR_UNSUPPORTED;
    const bool packed = ld_w == 0;        // ld_w == 0: `w` is the tile-major image asd_lm_head_pack_weights writes (N rows, D columns)
    if (!x || !w || !y || ld_x < D || (!packed && ld_w < D) || ld_y < N) return ASD_ERR_INVALID_ARG;
    if (!aligned_to(x, 16) || !aligned_to(w, 16) || !aligned_to(y, 8) || (bias && !aligned_to(bias, 8)) || ld_x % 8 != 0 ||
        ld_w % 8 != 0 || ld_y % 4 != 0)
        return ASD_ERR_ALIGNMENT;
    const bool f16 = dtype == ASD_DTYPE_F16;
    LinearPlan pl = linear_plan(M, N, D);
    if (g_force_linear_slices > 0) pl.k_slices = g_force_linear_slices <= D / kSuper ? g_force_linear_slices : D / kSuper;
    const size_t slab_bytes = pl.k_slices > 1 ? static_cast<size_t>(pl.k_slices) * M * static_cast<size_t>(N) * sizeof(float) : 0;
    if (slab_bytes > 0 && (!workspace || workspace_bytes < slab_bytes)) return ASD_ERR_WORKSPACE;
    if (slab_bytes > 0 && !aligned_to(workspace, 16)) return ASD_ERR_ALIGNMENT;
    if (pl.units * pl.k_slices >= (1ll << 31) || static_cast<int64_t>(M) * N / 4 / 256 >= (1ll << 31)) return ASD_ERR_UNSUPPORTED;

    LmHeadParams p{};
    p.hidden = x; p.ld_h = ld_x; p.weight = w; p.ld_w = ld_w;
    p.D = D; p.M = M; p.V = N;
    p.m_blocks = pl.kind == 3 ? 1 : (M + kBM - 1) / kBM;
    p.n_blocks = (N + 255) / 256;
    p.packed = packed ? 1 : 0;
    p.k_slices = pl.k_slices;
    p.slabs = static_cast<float*>(workspace);
    p.out = y; p.ld_out = ld_y; p.bias = bias; p.residual = residual; p.ld_res = ld_res;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(pl.units * pl.k_slices));
    if (pl.kind == 0) {
        if (f16) hipLaunchKernelGGL((k_lm_head_skinny<true, true>), grid, dim3(kThreads), 0, st, p);
        else hipLaunchKernelGGL((k_lm_head_skinny<false, true>), grid, dim3(kThreads), 0, st, p);
    } else if (pl.kind == 1) {
        if (f16) hipLaunchKernelGGL(k_linear_tile<true>, grid, dim3(kThreads), 0, st, p);
        else hipLaunchKernelGGL(k_linear_tile<false>, grid, dim3(kThreads), 0, st, p);
    } else if (pl.kind == 3) {
        const int mt = (M + 31) / 32;
        if (f16) {
            if (mt <= 7) hipLaunchKernelGGL((k_linear_tall<true, 7>), grid, dim3(kThreads), 0, st, p);
            else if (mt == 8) hipLaunchKernelGGL((k_linear_tall<true, 8>), grid, dim3(kThreads), 0, st, p);
            else hipLaunchKernelGGL((k_linear_tall<true, 9>), grid, dim3(kThreads), 0, st, p);
        } else {
            if (mt <= 7) hipLaunchKernelGGL((k_linear_tall<false, 7>), grid, dim3(kThreads), 0, st, p);
            else if (mt == 8) hipLaunchKernelGGL((k_linear_tall<false, 8>), grid, dim3(kThreads), 0, st, p);
            else hipLaunchKernelGGL((k_linear_tall<false, 9>), grid, dim3(kThreads), 0, st, p);
        }
    } else {
        if (f16) hipLaunchKernelGGL((k_lm_head_quad<true, true>), grid, dim3(kQThreads), 0, st, p);
        else hipLaunchKernelGGL((k_lm_head_quad<false, true>), grid, dim3(kQThreads), 0, st, p);
    }
    if (k_slices_out) *k_slices_out = pl.k_slices;
    if (pl.k_slices > 1 && !k_slices_out) {
        const int64_t n_threads = static_cast<int64_t>(M) * (N / 4);
        const dim3 rgrid(static_cast<unsigned>((n_threads + 255) / 256));
        if (f16) hipLaunchKernelGGL(k_linear_reduce<true>, rgrid, dim3(256), 0, st, p.slabs, pl.k_slices, M, N, bias, residual, ld_res, y, ld_y);
        else hipLaunchKernelGGL(k_linear_reduce<false>, rgrid, dim3(256), 0, st, p.slabs, pl.k_slices, M, N, bias, residual, ld_res, y, ld_y);
    }
    return launch_status();
}
}  // namespace

ASD_EXPORT int asd_linear_ex(const void* x, int64_t ld_x, const void* w, int64_t ld_w, const void* bias, const void* residual,
                             int64_t ld_res, int dtype, int M, int N, int D, void* y, int64_t ld_y, void* workspace,
                             size_t workspace_bytes, void* stream) {
    return linear_run(x, ld_x, w, ld_w, bias, residual, ld_res, dtype, M, N, D, y, ld_y, workspace, workspace_bytes, stream, nullptr);
}

ASD_EXPORT int asd_linear_partial(const void* x, int64_t ld_x, const void* w, int64_t ld_w, const void* bias, const void* residual,
                                  int64_t ld_res, int dtype, int M, int N, int D, void* y, int64_t ld_y, void* workspace,
                                  size_t workspace_bytes, void* stream, int* k_slices) {
    if (!k_slices) return ASD_ERR_INVALID_ARG;
    return linear_run(x, ld_x, w, ld_w, bias, residual, ld_res, dtype, M, N, D, y, ld_y, workspace, workspace_bytes, stream, k_slices);
}

ASD_EXPORT int asd_linear(const void* x, int64_t ld_x, const void* w, int64_t ld_w, const void* bias, int dtype, int M, int N,
                          int D, void* y, int64_t ld_y, void* workspace, size_t workspace_bytes, void* stream) {
    return asd_linear_ex(x, ld_x, w, ld_w, bias, nullptr, 0, dtype, M, N, D, y, ld_y, workspace, workspace_bytes, stream);
}
