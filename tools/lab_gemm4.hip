// Lab (not shipped): what does the MFMA side of the lm_head product reach with FOUR waves per workgroup, each holding a
// 128 x 128 logits tile (16 accumulator tiles, 16 MFMAs per 8 fragment reads), operands staged global -> VGPR ->
// ds_write_b128 into a double-buffered swizzled LDS image, one workgroup barrier per 64 reduction columns?  (The shipped
// k_lm_head_tile: 8 waves, 64 x 128 per wave, 8 MFMAs per 6 fragment reads, LDS-DMA rings.)  No epilogue: every lane adds
// up its accumulators into a sink, which the host checks for ONE workgroup against an f64 product of the same integers.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab_gemm4.hip -o /tmp/lab_gemm4 && /tmp/lab_gemm4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kTile = 256;      // rows and columns of a workgroup's block
constexpr int kSuper = 64;      // reduction columns per stage = one 128-byte line per row
constexpr int kSlot = kTile * 128;

// grid: (N / 256) * (M / 256) workgroups, column blocks fastest; W [N][D], H [M][D] bf16 row-major; D % 64 == 0
__global__ __launch_bounds__(256, 1) void k_gemm4(const uint16_t* __restrict__ W, const uint16_t* __restrict__ H, int M, int N,
                                                  int D, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * 2 * kSlot];     // [buffer][W | H]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv & 1, wn = wv >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int n_blocks = N / kTile;
    // the row blocks that share a weight tile get ids 8 apart: the same XCD (round-robin dispatch) and the same dispatch round,
    // so one of them pulls the tile from HBM and the others find it in that XCD's L2 (as the shipped kernel does)
    const int m_blocks = M / kTile;
    int nb, mb;
    {
        const int id = blockIdx.x, group = 8 * m_blocks, swizzled = (n_blocks / 8) * group;
        if (id < swizzled) {
            const int in_group = id % group;
            nb = (id / group) * 8 + in_group % 8;
            mb = in_group / 8;
        } else {
            mb = (id - swizzled) % m_blocks;
            nb = (n_blocks / 8) * 8 + (id - swizzled) / m_blocks;
        }
    }
    const char* wbase = reinterpret_cast<const char*>(W) + static_cast<int64_t>(nb) * kTile * D * 2;
    const char* hbase = reinterpret_cast<const char*>(H) + static_cast<int64_t>(mb) * kTile * D * 2;
    // staging: piece id = ps * 256 + t -> row id >> 3, 16-byte segment id & 7; LDS image XOR-swizzled like the shipped kernel
    uint32_t goff[8], loff[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int id = ps * 256 + t;
        const int row = id >> 3, seg = id & 7;
        goff[ps] = static_cast<uint32_t>(row) * static_cast<uint32_t>(D * 2) + seg * 16;
        loff[ps] = row * 128 + ((seg ^ ((row >> 1) & 7)) * 16);
    }
    u32x4 sw[8], sh[8];
    auto load_stage = [&](int S) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            sw[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + static_cast<int64_t>(S) * 128 + goff[ps]));
            sh[ps] = *reinterpret_cast<const u32x4*>(hbase + static_cast<int64_t>(S) * 128 + goff[ps]);
        }
    };
    auto store_stage = [&](int buf) {
        unsigned char* wb = lds + buf * 2 * kSlot;
        unsigned char* hb = wb + kSlot;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            *reinterpret_cast<u32x4*>(wb + loff[ps]) = sw[ps];
            *reinterpret_cast<u32x4*>(hb + loff[ps]) = sh[ps];
        }
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    const int key = (r >> 1) & 7;
    const int w_off = (128 * wn + r) * 128, h_off = (128 * wm + r) * 128;
    bf16x8 wf[2][4], hf[2][4];
    auto read_frags = [&](int buf, int ks, int set) {
        const unsigned char* wb = lds + buf * 2 * kSlot + w_off;
        const unsigned char* hb = lds + buf * 2 * kSlot + kSlot + h_off;
        const int so = ((4 * h + ks) ^ key) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wf[set][i] = *reinterpret_cast<const bf16x8*>(wb + i * 32 * 128 + so);
            hf[set][i] = *reinterpret_cast<const bf16x8*>(hb + i * 32 * 128 + so);
        }
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[set][nt], hf[set][mt], acc[mt][nt], 0, 0, 0);
    };
    // One scheduling region per k-step: 16 MFMAs with the 8 fragment reads of the NEXT k-step, 4 staging stores and 4 staging
    // loads dealt between them (sched_group_barrier: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x020 VMEM read), so that the
    // wave never issues a long run of non-MFMA instructions while the matrix pipe drains.
    auto interleave = [&](bool reads, bool stores, bool loads) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (stores) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (loads) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // 16 MFMAs with 8 fragment reads, 8 staging stores and 8 staging loads
    auto interleave2 = [&](bool stores, bool loads) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (stores) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (loads) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int n_super = D / kSuper;
    // a quarter of the staging work -- 4 of the 16 register pieces: ds_write of stage S + 1, then the load of stage S + 2 into
    // the same registers -- behind each of the four MFMA groups of superstage S
    // steady state is branch-free so that the compiler can COUNT the loads in flight (a conditional load makes it wait for
    // vmcnt(0) in front of every ds_write)
    auto restage = [&](int S, int part, bool store, bool load) {
        unsigned char* wb = lds + ((S + 1) & 1) * 2 * kSlot;
        unsigned char* hb = wb + kSlot;
#pragma unroll
        for (int ps = 2 * part; ps < 2 * part + 2; ++ps) {
            if (store) {
                *reinterpret_cast<u32x4*>(wb + loff[ps]) = sw[ps];
                *reinterpret_cast<u32x4*>(hb + loff[ps]) = sh[ps];
            }
            if (load) {
                sw[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + static_cast<int64_t>(S + 2) * 128 + goff[ps]));
                sh[ps] = *reinterpret_cast<const u32x4*>(hbase + static_cast<int64_t>(S + 2) * 128 + goff[ps]);
            }
        }
    };
    // The LAST k-step of a superstage is multiplied behind the barrier that ends it (its fragments are in register set 1 by
    // then), together with the reads of the next superstage's first k-step: the matrix pipe has 16 MFMAs of work while the
    // first fragments of the new buffer arrive.
    auto superstage = [&](int S, bool first, bool store, bool load) {
        const int buf = S & 1;
        read_frags(buf, 0, 0);
        if (!first) multiply(1);                       // (S - 1, k-step 3)
        restage(S, 0, store, load);
        interleave(true, store, load);
        read_frags(buf, 1, 1);
        multiply(0);
        restage(S, 1, store, load);
        interleave(true, store, load);
        read_frags(buf, 2, 0);
        multiply(1);
        restage(S, 2, store, load);
        interleave(true, store, load);
        read_frags(buf, 3, 1);
        multiply(0);
        restage(S, 3, store, load);
        interleave(true, store, load);
        __syncthreads();
    };
    load_stage(0);
    store_stage(0);
    __syncthreads();
    load_stage(1);                       // (n_super >= 3 in every run below)
    superstage(0, true, true, true);
    int S = 1;
    for (; S + 2 < n_super; ++S) superstage(S, false, true, true);
    superstage(S, false, true, false);
    superstage(S + 1, false, false, false);
    multiply(1);
    float sum = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += acc[a][b][i];
    sink[static_cast<int64_t>(blockIdx.x) * 256 + t] = sum;
}

// the 32x32x16 form with TWO weight stages in registers
// grid: (N / 256) * (M / 256) workgroups, column blocks fastest; W [N][D], H [M][D] bf16 row-major; D % 64 == 0
__global__ __launch_bounds__(256, 1) void k_gemm4_deep(const uint16_t* __restrict__ W, const uint16_t* __restrict__ H, int M, int N,
                                                  int D, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * 2 * kSlot];     // [buffer][W | H]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv & 1, wn = wv >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int n_blocks = N / kTile;
    // the row blocks that share a weight tile get ids 8 apart: the same XCD (round-robin dispatch) and the same dispatch round,
    // so one of them pulls the tile from HBM and the others find it in that XCD's L2 (as the shipped kernel does)
    const int m_blocks = M / kTile;
    int nb, mb;
    {
        const int id = blockIdx.x, group = 8 * m_blocks, swizzled = (n_blocks / 8) * group;
        if (id < swizzled) {
            const int in_group = id % group;
            nb = (id / group) * 8 + in_group % 8;
            mb = in_group / 8;
        } else {
            mb = (id - swizzled) % m_blocks;
            nb = (n_blocks / 8) * 8 + (id - swizzled) / m_blocks;
        }
    }
    const char* wbase = reinterpret_cast<const char*>(W) + static_cast<int64_t>(nb) * kTile * D * 2;
    const char* hbase = reinterpret_cast<const char*>(H) + static_cast<int64_t>(mb) * kTile * D * 2;
    // staging: piece id = ps * 256 + t -> row id >> 3, 16-byte segment id & 7; LDS image XOR-swizzled like the shipped kernel
    uint32_t goff[8], loff[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int id = ps * 256 + t;
        const int row = id >> 3, seg = id & 7;
        goff[ps] = static_cast<uint32_t>(row) * static_cast<uint32_t>(D * 2) + seg * 16;
        loff[ps] = row * 128 + ((seg ^ ((row >> 1) & 7)) * 16);
    }
    // TWO weight stages in registers (the hidden states are L2-resident: one)
    u32x4 swa[8], swb[8], sh[8];
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    const int key = (r >> 1) & 7;
    const int w_off = (128 * wn + r) * 128, h_off = (128 * wm + r) * 128;
    bf16x8 wf[2][4], hf[2][4];
    auto read_frags = [&](int buf, int ks, int set) {
        const unsigned char* wb = lds + buf * 2 * kSlot + w_off;
        const unsigned char* hb = lds + buf * 2 * kSlot + kSlot + h_off;
        const int so = ((4 * h + ks) ^ key) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wf[set][i] = *reinterpret_cast<const bf16x8*>(wb + i * 32 * 128 + so);
            hf[set][i] = *reinterpret_cast<const bf16x8*>(hb + i * 32 * 128 + so);
        }
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[set][nt], hf[set][mt], acc[mt][nt], 0, 0, 0);
    };
    // One scheduling region per k-step: 16 MFMAs with the 8 fragment reads of the NEXT k-step, 4 staging stores and 4 staging
    // loads dealt between them (sched_group_barrier: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x020 VMEM read), so that the
    // wave never issues a long run of non-MFMA instructions while the matrix pipe drains.
    auto interleave = [&](bool reads, bool stores, bool loads) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (stores) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (loads) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // 16 MFMAs with 8 fragment reads, 8 staging stores and 8 staging loads
    auto interleave2 = [&](bool stores, bool loads) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (stores) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (loads) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int n_super = D / kSuper;
    // a quarter of the staging work -- 4 of the 16 register pieces: ds_write of stage S + 1, then the load of stage S + 2 into
    // the same registers -- behind each of the four MFMA groups of superstage S
    // superstage S multiplies LDS buffer S & 1; `cur` holds W(S + 1) (stored to the other buffer now, then refilled with
    // W(S + 3)), the other register set holds W(S + 2) untouched; sh holds H(S + 1), refilled with H(S + 2)
    auto restage = [&](u32x4 (&cur)[8], int S, int part, bool store, bool loadw, bool loadh) {
        unsigned char* wb = lds + ((S + 1) & 1) * 2 * kSlot;
        unsigned char* hb = wb + kSlot;
#pragma unroll
        for (int ps = 2 * part; ps < 2 * part + 2; ++ps) {
            if (store) {
                *reinterpret_cast<u32x4*>(wb + loff[ps]) = cur[ps];
                *reinterpret_cast<u32x4*>(hb + loff[ps]) = sh[ps];
            }
            if (loadw) cur[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + static_cast<int64_t>(S + 3) * 128 + goff[ps]));
            if (loadh) sh[ps] = *reinterpret_cast<const u32x4*>(hbase + static_cast<int64_t>(S + 2) * 128 + goff[ps]);
        }
    };
    auto superstage = [&](u32x4 (&cur)[8], int S, bool first, bool store, bool loadw, bool loadh) {
        const int buf = S & 1;
        read_frags(buf, 0, 0);
        if (!first) multiply(1);
        restage(cur, S, 0, store, loadw, loadh);
        interleave(true, store, loadw || loadh);
        read_frags(buf, 1, 1);
        multiply(0);
        restage(cur, S, 1, store, loadw, loadh);
        interleave(true, store, loadw || loadh);
        read_frags(buf, 2, 0);
        multiply(1);
        restage(cur, S, 2, store, loadw, loadh);
        interleave(true, store, loadw || loadh);
        read_frags(buf, 3, 1);
        multiply(0);
        restage(cur, S, 3, store, loadw, loadh);
        interleave(true, store, loadw || loadh);
        __syncthreads();
    };
    // prologue: stage 0 through registers into buffer 0; W(1) -> swa, H(1) -> sh, W(2) -> swb   (n_super >= 6, even, below)
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        swa[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + goff[ps]));
        sh[ps] = *reinterpret_cast<const u32x4*>(hbase + goff[ps]);
    }
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        *reinterpret_cast<u32x4*>(lds + loff[ps]) = swa[ps];
        *reinterpret_cast<u32x4*>(lds + kSlot + loff[ps]) = sh[ps];
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        swa[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + 128 + goff[ps]));
        sh[ps] = *reinterpret_cast<const u32x4*>(hbase + 128 + goff[ps]);
        swb[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + 256 + goff[ps]));
    }
    superstage(swa, 0, true, true, true, true);
    int S = 1;
    for (; S + 4 < n_super; S += 2) {
        superstage(swb, S, false, true, true, true);
        superstage(swa, S + 1, false, true, true, true);
    }
    // n_super even: the loop leaves S = n_super - 3 (odd).  swb holds W(n - 2), swa W(n - 1), sh H(n - 2): no further weights.
    superstage(swb, S, false, true, false, true);         // stores stage n - 2, loads H(n - 1)
    superstage(swa, S + 1, false, true, false, false);    // stores stage n - 1
    superstage(swb, S + 2, false, false, false, false);
    multiply(1);
    float sum = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += acc[a][b][i];
    sink[static_cast<int64_t>(blockIdx.x) * 256 + t] = sum;
}

// the same with v_mfma_f32_16x16x32_bf16
// grid: (N / 256) * (M / 256) workgroups, column blocks fastest; W [N][D], H [M][D] bf16 row-major; D % 64 == 0
__global__ __launch_bounds__(256, 1) void k_gemm4_16(const uint16_t* __restrict__ W, const uint16_t* __restrict__ H, int M, int N,
                                                  int D, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * 2 * kSlot];     // [buffer][W | H]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv & 1, wn = wv >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int n_blocks = N / kTile;
    // the row blocks that share a weight tile get ids 8 apart: the same XCD (round-robin dispatch) and the same dispatch round,
    // so one of them pulls the tile from HBM and the others find it in that XCD's L2 (as the shipped kernel does)
    const int m_blocks = M / kTile;
    int nb, mb;
    {
        const int id = blockIdx.x, group = 8 * m_blocks, swizzled = (n_blocks / 8) * group;
        if (id < swizzled) {
            const int in_group = id % group;
            nb = (id / group) * 8 + in_group % 8;
            mb = in_group / 8;
        } else {
            mb = (id - swizzled) % m_blocks;
            nb = (n_blocks / 8) * 8 + (id - swizzled) / m_blocks;
        }
    }
    const char* wbase = reinterpret_cast<const char*>(W) + static_cast<int64_t>(nb) * kTile * D * 2;
    const char* hbase = reinterpret_cast<const char*>(H) + static_cast<int64_t>(mb) * kTile * D * 2;
    // staging: piece id = ps * 256 + t -> row id >> 3, 16-byte segment id & 7; LDS image XOR-swizzled like the shipped kernel
    uint32_t goff[8], loff[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int id = ps * 256 + t;
        const int row = id >> 3, seg = id & 7;
        goff[ps] = static_cast<uint32_t>(row) * static_cast<uint32_t>(D * 2) + seg * 16;
        loff[ps] = row * 128 + ((seg ^ ((row >> 1) & 7)) * 16);
    }
    u32x4 sw[8], sh[8];
    auto load_stage = [&](int S) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            sw[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + static_cast<int64_t>(S) * 128 + goff[ps]));
            sh[ps] = *reinterpret_cast<const u32x4*>(hbase + static_cast<int64_t>(S) * 128 + goff[ps]);
        }
    };
    auto store_stage = [&](int buf) {
        unsigned char* wb = lds + buf * 2 * kSlot;
        unsigned char* hb = wb + kSlot;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            *reinterpret_cast<u32x4*>(wb + loff[ps]) = sw[ps];
            *reinterpret_cast<u32x4*>(hb + loff[ps]) = sh[ps];
        }
    };
    // v_mfma_f32_16x16x32_bf16: lane (c = lane & 15, g = lane >> 4) supplies row / column c and k = 8 g .. 8 g + 7 of a 32-deep
    // step; its fragment of step q of a superstage is the 16-byte segment 4 q + g of its row (the same swizzled image)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[a][b][i] = 0.0f;
    const int c16 = lane & 15, g16 = lane >> 4;
    const int key = (c16 >> 1) & 7;
    const int w_off = (128 * wn + c16) * 128, h_off = (128 * wm + c16) * 128;
    bf16x8 wf[2][8], hf[2][8];
    auto read_frags = [&](int buf, int q, int set) {
        const unsigned char* wb = lds + buf * 2 * kSlot + w_off;
        const unsigned char* hb = lds + buf * 2 * kSlot + kSlot + h_off;
        const int so = ((4 * q + g16) ^ key) * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            wf[set][i] = *reinterpret_cast<const bf16x8*>(wb + i * 16 * 128 + so);
            hf[set][i] = *reinterpret_cast<const bf16x8*>(hb + i * 16 * 128 + so);
        }
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set][nt], hf[set][mt], acc[nt][mt], 0, 0, 0);
    };
    // One scheduling region per k-step: 16 MFMAs with the 8 fragment reads of the NEXT k-step, 4 staging stores and 4 staging
    // loads dealt between them (sched_group_barrier: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x020 VMEM read), so that the
    // wave never issues a long run of non-MFMA instructions while the matrix pipe drains.
    auto interleave = [&](bool reads, bool stores, bool loads) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (reads && (i & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (stores && (i & 7) == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (loads && (i & 7) == 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // 16 MFMAs with 8 fragment reads, 8 staging stores and 8 staging loads
    auto interleave2 = [&](bool stores, bool loads) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (stores) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (loads) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int n_super = D / kSuper;
    // a quarter of the staging work -- 4 of the 16 register pieces: ds_write of stage S + 1, then the load of stage S + 2 into
    // the same registers -- behind each of the four MFMA groups of superstage S
    // steady state is branch-free so that the compiler can COUNT the loads in flight (a conditional load makes it wait for
    // vmcnt(0) in front of every ds_write)
    auto restage = [&](int S, int part, bool store, bool load) {
        unsigned char* wb = lds + ((S + 1) & 1) * 2 * kSlot;
        unsigned char* hb = wb + kSlot;
#pragma unroll
        for (int ps = 2 * part; ps < 2 * part + 2; ++ps) {
            if (store) {
                *reinterpret_cast<u32x4*>(wb + loff[ps]) = sw[ps];
                *reinterpret_cast<u32x4*>(hb + loff[ps]) = sh[ps];
            }
            if (load) {
                sw[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + static_cast<int64_t>(S + 2) * 128 + goff[ps]));
                sh[ps] = *reinterpret_cast<const u32x4*>(hbase + static_cast<int64_t>(S + 2) * 128 + goff[ps]);
            }
        }
    };
    // The LAST k-step of a superstage is multiplied behind the barrier that ends it (its fragments are in register set 1 by
    // then), together with the reads of the next superstage's first k-step: the matrix pipe has 16 MFMAs of work while the
    // first fragments of the new buffer arrive.
    auto superstage = [&](int S, bool first, bool store, bool load) {
        const int buf = S & 1;
        read_frags(buf, 0, 0);
        if (!first) multiply(1);                       // (S - 1, step 1)
        restage(S, 0, store, load);
        restage(S, 1, store, load);
        interleave(true, store, load);
        read_frags(buf, 1, 1);
        multiply(0);
        restage(S, 2, store, load);
        restage(S, 3, store, load);
        interleave(true, store, load);
        __syncthreads();
    };
    load_stage(0);
    store_stage(0);
    __syncthreads();
    load_stage(1);                       // (n_super >= 3 in every run below)
    superstage(0, true, true, true);
    int S = 1;
    for (; S + 2 < n_super; ++S) superstage(S, false, true, true);
    superstage(S, false, true, false);
    superstage(S + 1, false, false, false);
    multiply(1);
    float sum = 0.0f;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) sum += acc[a][b][i];
    sink[static_cast<int64_t>(blockIdx.x) * 256 + t] = sum;
}

static uint16_t bf16_of(float f) {     // exact for the small dyadic values used here
    uint32_t u;
    memcpy(&u, &f, 4);
    return static_cast<uint16_t>(u >> 16);
}

template <typename KernelT>
static int run_variant(KernelT kernel, bool mfma16, const char* name, const uint16_t* dW, const uint16_t* dH, float* sink, int M,
                       int N, int D) {
    const int grid = (N / 256) * (M / 256);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, dW, dH, M, N, D, sink);
    CHECK(hipDeviceSynchronize());
    // check workgroup `probe` (a block away from the origin) lane by lane
    const int probe = grid - 3;
    std::vector<float> got(256);
    CHECK(hipMemcpy(got.data(), sink + static_cast<size_t>(probe) * 256, 1024, hipMemcpyDeviceToHost));
    int nbk, mbk;
    {
        const int n_blocks = N / 256, m_blocks = M / 256, group = 8 * m_blocks, swizzled = (n_blocks / 8) * group;
        if (probe < swizzled) { const int in_group = probe % group; nbk = (probe / group) * 8 + in_group % 8; mbk = in_group / 8; }
        else { mbk = (probe - swizzled) % m_blocks; nbk = (n_blocks / 8) * 8 + (probe - swizzled) / m_blocks; }
    }
    // C[m][n] of the probed block (f64; the operands are small dyadic numbers: exact)
    std::vector<double> C(256 * 256);
    for (int m = 0; m < 256; ++m)
        for (int n = 0; n < 256; ++n) {
            const int gm = mbk * 256 + m, gn = nbk * 256 + n;
            double c = 0.0;
            for (int k = 0; k < D; ++k) c += (((gn + k) % 7) * 0.25 - 0.75) * (((gm + 2 * k) % 5) * 0.5 - 1.0);
            C[m * 256 + n] = c;
        }
    double worst = 0.0;
    for (int t = 0; t < 256; ++t) {
        const int lane = t & 63, wv = t >> 6, wm = wv & 1, wn = wv >> 1;
        double want = 0.0;
        if (!mfma16) {
            const int r = lane & 31, h = lane >> 5;
            for (int mt = 0; mt < 4; ++mt)
                for (int nt = 0; nt < 4; ++nt)
                    for (int i = 0; i < 16; ++i)
                        want += C[(128 * wm + 32 * mt + r) * 256 + 128 * wn + 32 * nt + (i & 3) + 8 * (i >> 2) + 4 * h];
        } else {
            const int c = lane & 15, g = lane >> 4;
            for (int mt = 0; mt < 8; ++mt)
                for (int nt = 0; nt < 8; ++nt)
                    for (int j = 0; j < 4; ++j) want += C[(128 * wm + 16 * mt + c) * 256 + 128 * wn + 16 * nt + 4 * g + j];
        }
        worst = fmax(worst, fabs(want - got[t]) / fmax(1.0, fabs(want)));
    }
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, dW, dH, M, N, D, sink);
    CHECK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, dW, dH, M, N, D, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    printf("M=%4d N=%d D=%d  %-28s %8.1f us  %6.3f PFLOP/s  weights %5.2f TB/s   lane check %.1e %s\n", M, N, D, name, us,
           2.0 * M * N * D / us * 1e-9, static_cast<double>(N) * D * 2 / us * 1e-6, worst, worst < 1e-4 ? "(ok)" : "(WRONG)");
    return 0;
}

int main() {
    const int N = 152064, D = 8192;
    for (int M : {256, 1024}) {
        std::vector<uint16_t> hW(static_cast<size_t>(N) * D), hH(static_cast<size_t>(M) * D);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < D; ++k) hW[static_cast<size_t>(n) * D + k] = bf16_of(((n + k) % 7) * 0.25f - 0.75f);
        for (int m = 0; m < M; ++m)
            for (int k = 0; k < D; ++k) hH[static_cast<size_t>(m) * D + k] = bf16_of(((m + 2 * k) % 5) * 0.5f - 1.0f);
        uint16_t *dW, *dH;
        float* sink;
        const int grid = (N / 256) * (M / 256);
        CHECK(hipMalloc(&dW, hW.size() * 2));
        CHECK(hipMalloc(&dH, hH.size() * 2));
        CHECK(hipMalloc(&sink, static_cast<size_t>(grid) * 256 * 4));
        CHECK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dH, hH.data(), hH.size() * 2, hipMemcpyHostToDevice));
        if (run_variant(k_gemm4, false, "4 waves, mfma 32x32x16", dW, dH, sink, M, N, D)) return 1;
        if (run_variant(k_gemm4_16, true, "4 waves, mfma 16x16x32", dW, dH, sink, M, N, D)) return 1;
        if (run_variant(k_gemm4_deep, false, "4 waves, 2 W stages in regs", dW, dH, sink, M, N, D)) return 1;
        CHECK(hipFree(dW));
        CHECK(hipFree(dH));
        CHECK(hipFree(sink));
    }
    return 0;
}
