// commit.hip -- SURVEY §8(f) N3: the step after accept.  The reference's cache manager only has the
// name (src/serving/cache_manager.py:149-190 `truncate_at_stage` trims a dict of strings); in a
// token-level loop the rollback of a per-sequence KV cache is a LENGTH update: entries past the accepted
// prefix stay where they are and are overwritten by the next step (queries never look past their own
// position).  This kernel does the whole bookkeeping of one step on the device, so the loop needs no
// host synchronisation to learn n_acc:
//   row b of the token buffer receives tok[b, 0 .. n_acc[b]) followed by drawn[b] at seq_len[b],
//   seq_len[b] += n_acc[b] + 1  (clamped to max_len; tokens past max_len are dropped),
//   n_commit[b] = number of tokens actually appended.
// One wave per sequence (K <= 64): lane k moves draft token k.  Integer work: bit-exact vs the oracle.
#include "common.hpp"

namespace asd {
namespace {

__global__ __launch_bounds__(64) void k_commit_step(const int32_t* tok, const int32_t* n_acc, const int32_t* drawn, int B,
                                                    int K, int32_t* seq_len, int32_t* out_tokens, int64_t ld_out,
                                                    int32_t* n_commit, int32_t max_len) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int len = seq_len[b];                      // every lane reads the old length before lane 0 rewrites it
    int na = n_acc[b];
    na = na < 0 ? 0 : (na > K ? K : na);
    int32_t* row = out_tokens + static_cast<int64_t>(b) * ld_out;
    if (lane < na && len + lane < max_len) row[len + lane] = tok[static_cast<int64_t>(b) * K + lane];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        if (len + na < max_len) row[len + na] = drawn[b];
        int appended = max_len - len;
        appended = appended < 0 ? 0 : (appended > na + 1 ? na + 1 : appended);
        seq_len[b] = len + appended;
        if (n_commit) n_commit[b] = appended;
    }
}

}  // namespace
}  // namespace asd

using namespace asd;

ASD_EXPORT int asd_commit_step(const int32_t* tok, const int32_t* n_acc, const int32_t* drawn, int B, int K,
                               int32_t* seq_len, int32_t* out_tokens, int64_t ld_out, int32_t* n_commit, int32_t max_len,
                               void* stream) {
    if (B < 0 || K < 0 || max_len < 0) return ASD_ERR_INVALID_ARG;
    if (B == 0) return ASD_OK;
    if (K > ASD_MAX_DRAFT_LEN) return ASD_ERR_UNSUPPORTED;
    if ((K > 0 && !tok) || !n_acc || !drawn || !seq_len || !out_tokens) return ASD_ERR_INVALID_ARG;
    if (ld_out < max_len) return ASD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_commit_step, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), tok, n_acc, drawn, B, K,
                       seq_len, out_tokens, ld_out, n_commit, max_len);
    return launch_status();
}
