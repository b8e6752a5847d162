// lab_stream_floor.hip -- what does a pure read of the verify step's logits cost?  (tools/lab_stream_floor.py)
// The same launch shape as k_verify at rows >= CUs: one workgroup per row, 16-byte buffer loads, `nt`; nothing is computed but an
// xor of the words (one dword stored per workgroup), so the time is ramp + stream + drain of the memory system alone.
//   mode 0  static striding, DEPTH loads in flight per lane
//   mode 1  dynamic tiles claimed from an LDS counter (k_verify's loop structure), two tiles in flight per wave
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
static __device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, NT ? 2 : 0));
}

template <int THREADS, int DEPTH, bool NT>
__global__ __launch_bounds__(THREADS) void k_stream_static(const char* base, int64_t row_bytes, int64_t ld_bytes, int splits, uint32_t* out) {
    const int row = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int64_t slice = (row_bytes / splits) & ~15ll;
    const char* p = base + row * ld_bytes + sp * slice;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, static_cast<int>(slice), 0x00020000);
    const uint32_t step = THREADS * 16u;
    uint32_t off = threadIdx.x * 16u;
    u32x4 r[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) r[j] = load16<NT>(rsrc, off + j * step);
    uint32_t acc = 0;
    for (; off < static_cast<uint32_t>(slice); off += DEPTH * step) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const u32x4 v = r[j];
            r[j] = load16<NT>(rsrc, off + (DEPTH + j) * step);      // past the end: dropped by the range check
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x * THREADS + threadIdx.x] = acc;     // (never: keeps the loads alive)
    if (threadIdx.x == 0) out[blockIdx.x] = 1;
}

template <int THREADS, int UNROLL, bool NT>
__global__ __launch_bounds__(THREADS) void k_stream_dynamic(const char* base, int64_t row_bytes, int64_t ld_bytes, int splits, uint32_t* out) {
    constexpr int kWaves = THREADS / 64;
    __shared__ uint32_t next_tile;
    const int row = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int64_t slice = (row_bytes / splits) & ~15ll;
    const char* p = base + row * ld_bytes + sp * slice;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr uint32_t kTile = UNROLL * 1024u;
    const uint32_t end = static_cast<uint32_t>(slice);
    const uint32_t n_tiles = (end + kTile - 1) / kTile;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, static_cast<int>(slice), 0x00020000);
    const uint32_t lane_off = lane * 16u;
    if (threadIdx.x == 0) next_tile = 2u * kWaves;
    uint32_t ta = wave, tb = wave + kWaves;
    u32x4 ra[UNROLL], rb[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) ra[j] = load16<NT>(rsrc, ta * kTile + j * 1024u + lane_off);
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) rb[j] = load16<NT>(rsrc, tb * kTile + j * 1024u + lane_off);
    __syncthreads();
    uint32_t acc = 0;
    while (ta < n_tiles) {
        uint32_t c = 0;
        if (lane == 0) c = __hip_atomic_fetch_add(&next_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t tn = __builtin_amdgcn_readfirstlane(c);
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) acc ^= ra[j][0] ^ ra[j][1] ^ ra[j][2] ^ ra[j][3];
        ta = tn;
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) ra[j] = load16<NT>(rsrc, ta * kTile + j * 1024u + lane_off);
        if (tb >= n_tiles) break;
        c = 0;
        if (lane == 0) c = __hip_atomic_fetch_add(&next_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tn = __builtin_amdgcn_readfirstlane(c);
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) acc ^= rb[j][0] ^ rb[j][1] ^ rb[j][2] ^ rb[j][3];
        tb = tn;
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) rb[j] = load16<NT>(rsrc, tb * kTile + j * 1024u + lane_off);
    }
    if (acc == 0x12345678u) out[blockIdx.x * THREADS + threadIdx.x] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = 1;
}

extern "C" __attribute__((visibility("default")))
int lab_stream(const void* base, int64_t rows, int64_t row_bytes, int64_t ld_bytes, int mode, int threads, int depth, int nt, int splits,
               void* out, void* stream) {
    const dim3 grid(static_cast<unsigned>(rows * splits));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char* b = static_cast<const char*>(base);
    uint32_t* o = static_cast<uint32_t*>(out);
#define L(K, T, D, N) hipLaunchKernelGGL((K<T, D, N>), grid, dim3(T), 0, st, b, row_bytes, ld_bytes, splits, o)
    if (mode == 0) {
        if (threads == 512 && depth == 6 && nt) L(k_stream_static, 512, 6, true);
        else if (threads == 512 && depth == 6) L(k_stream_static, 512, 6, false);
        else if (threads == 512 && depth == 4 && nt) L(k_stream_static, 512, 4, true);
        else if (threads == 512 && depth == 8 && nt) L(k_stream_static, 512, 8, true);
        else if (threads == 1024 && depth == 4 && nt) L(k_stream_static, 1024, 4, true);
        else if (threads == 256 && depth == 8 && nt) L(k_stream_static, 256, 8, true);
        else return -2;
    } else {
        if (threads == 512 && depth == 3 && nt) L(k_stream_dynamic, 512, 3, true);
        else if (threads == 512 && depth == 3) L(k_stream_dynamic, 512, 3, false);
        else if (threads == 512 && depth == 4 && nt) L(k_stream_dynamic, 512, 4, true);
        else if (threads == 1024 && depth == 2 && nt) L(k_stream_dynamic, 1024, 2, true);
        else return -2;
    }
#undef L
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
