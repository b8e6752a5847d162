// Lab (not shipped): how fast can a CU-resident workgroup stream the lm_head weight matrix straight into VGPRs in the
// MFMA A-operand layout (lane (r, h) holds the 16-byte segments 4h .. 4h+3 of weight row r of a 128-byte line), i.e.
// WITHOUT the LDS ring that bounds the bytes in flight of k_lm_head_tile?  Every load instruction touches 32 bytes of each
// of 32 lines; four consecutive instructions complete the lines.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab_wstream.hip -o /tmp/lab_wstream && /tmp/lab_wstream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// WAVES_N: waves side by side on the block's 256 rows (8: 32 rows each, no duplicate loads; 4: 64 rows each and the two
// wave rows load the same lines).  DEPTH: superstages (64 reduction columns = one line per row) a wave keeps in flight.
template <int WAVES_N, int DEPTH, bool NT, bool PACKED>
__global__ __launch_bounds__(512, 1) void k_stream(const char* w, int V, int D, uint32_t* sink) {
    __shared__ unsigned char pad[96 * 1024];       // one workgroup per CU, like the real kernel
    constexpr int TILES = 8 / WAVES_N;             // 32-row tiles per wave
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wn = wv % WAVES_N;
    const int r = lane & 31, h = lane >> 5;
    const int n_blocks = V / 256, n_super = D / 64;
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (t == 100000) pad[0] = 1;
    for (int nb = blockIdx.x; nb < n_blocks; nb += gridDim.x) {
        const char* base[TILES];
        int64_t stage_stride;
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl) {
            const int row = (wn * TILES + tl) * 32 + r;
            if (PACKED) {
                base[tl] = w + (int64_t)nb * n_super * (256 * 128) + row * 128 + h * 64;
                stage_stride = 256 * 128;
            } else {
                base[tl] = w + ((int64_t)nb * 256 + row) * D * 2 + h * 64;
                stage_stride = 128;
            }
        }
        u32x4 reg[DEPTH][TILES][4];
        auto issue = [&](int s, int slot) {
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4* src = reinterpret_cast<const u32x4*>(base[tl] + (int64_t)s * stage_stride + ks * 16);
                    reg[slot][tl][ks] = NT ? __builtin_nontemporal_load(src) : *src;
                }
        };
        auto consume = [&](int slot) {
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc ^= reg[slot][tl][ks];
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
        int s = 0;
        for (; s + DEPTH <= n_super; s += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + d + DEPTH - 1 < n_super) issue(s + d + DEPTH - 1, (d + DEPTH - 1) % DEPTH);
                __builtin_amdgcn_sched_barrier(0);
                consume(d);
            }
        }
        for (int d = 0; s + d < n_super; ++d) consume(d);   // (n_super % DEPTH == 0 in the runs below)
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int WAVES_N, int DEPTH, bool NT, bool PACKED>
int run(const char* w, int V, int D, uint32_t* sink, const char* name) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<WAVES_N, DEPTH, NT, PACKED>), dim3(256), dim3(512), 0, 0, w, V, D, sink);
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<WAVES_N, DEPTH, NT, PACKED>), dim3(256), dim3(512), 0, 0, w, V, D, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1000.0 / reps;
    const double bytes = (double)(V / 256) * 256 * D * 2;
    printf("%-44s D=%5d  %8.1f us  %6.2f TB/s\n", name, D, us, bytes / us * 1e-6);
    return 0;
}

int main() {
    const int V = 152064;
    for (int D : {3584 + 256, 8192}) {     // 3840 = 60 superstages (divisible by 2, 3, 4, 5, 6); the 7B head has 56
        const size_t bytes = (size_t)V * D * 2;
        char* w;
        uint32_t* sink;
        CHECK(hipMalloc(&w, bytes));
        CHECK(hipMalloc(&sink, 4));
        CHECK(hipMemset(w, 1, bytes));
        if (D == 8192) {
            if (run<8, 4, false, false>(w, V, D, sink, "1x8 waves, depth 4, [V][D]")) return 1;
            if (run<8, 4, true, false>(w, V, D, sink, "1x8 waves, depth 4, [V][D], nt")) return 1;
            if (run<8, 4, false, true>(w, V, D, sink, "1x8 waves, depth 4, packed")) return 1;
            if (run<8, 4, true, true>(w, V, D, sink, "1x8 waves, depth 4, packed, nt")) return 1;
            if (run<8, 8, true, true>(w, V, D, sink, "1x8 waves, depth 8, packed, nt")) return 1;
            if (run<8, 2, true, true>(w, V, D, sink, "1x8 waves, depth 2, packed, nt")) return 1;
            if (run<4, 2, true, true>(w, V, D, sink, "2x4 waves (dup), depth 2, packed, nt")) return 1;
            if (run<4, 4, true, true>(w, V, D, sink, "2x4 waves (dup), depth 4, packed, nt")) return 1;
            if (run<4, 4, false, true>(w, V, D, sink, "2x4 waves (dup), depth 4, packed")) return 1;
            if (run<4, 4, false, false>(w, V, D, sink, "2x4 waves (dup), depth 4, [V][D]")) return 1;
        } else {
            if (run<8, 4, true, true>(w, V, D, sink, "1x8 waves, depth 4, packed, nt")) return 1;
            if (run<8, 4, false, false>(w, V, D, sink, "1x8 waves, depth 4, [V][D]")) return 1;
            if (run<4, 4, true, true>(w, V, D, sink, "2x4 waves (dup), depth 4, packed, nt")) return 1;
            if (run<4, 3, true, true>(w, V, D, sink, "2x4 waves (dup), depth 3, packed, nt")) return 1;
        }
        CHECK(hipFree(w));
        CHECK(hipFree(sink));
    }
    return 0;
}
