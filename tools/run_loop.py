#!/usr/bin/env python3
"""BASELINE configs[2]: the full token-level loop with Qwen2.5-SHAPE synthetic models on one MI355X
(random weights generated on the device; no checkpoint, no network).

    python tools/run_loop.py --draft 7b --target 32b --batch 32 --draft-len 8 --new-tokens 48

Per step: K draft forwards (7B shape), one target forward over the K drafted positions, then the
hot path of this repo: asd_verify_accept + asd_predictor_stop.  Reports verified tokens/s for
the WHOLE loop (dominated by the third-party-equivalent model execution: torch / hipBLASLt), the
time share of the two kernels, and checks the first steps' accept masks against the CPU oracle.
Model execution here is plumbing; the headline number of the repo is bench.py's.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from asd_amd.minimal_adaptive_decoder import MinimalQualityPredictor  # noqa: E402
from asd_amd.serving.speculative import (SpeculativeVerifier, speculative_generate,  # noqa: E402
                                             speculative_generate_ragged)
from asd_amd.serving.synthetic_lm import QWEN25_SHAPES, SyntheticLM, tiny  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draft", default="7b")
    ap.add_argument("--target", default="32b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--draft-len", type=int, default=8)
    ap.add_argument("--prompt-len", type=int, default=32)
    ap.add_argument("--new-tokens", type=int, default=48)
    ap.add_argument("--logit-scale", type=float, default=0.6,
                    help="random-weight logits are scaled so that two unrelated models still accept some tokens")
    ap.add_argument("--check-steps", type=int, default=2)
    ap.add_argument("--ragged", action="store_true",
                    help="per-sequence commit (N3: ragged KV + asd_commit_step) instead of the lock-step minimum")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "loop.json"))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    shapes = dict(QWEN25_SHAPES, tiny=tiny())
    t0 = time.time()
    draft = SyntheticLM(shapes[a.draft], dtype=torch.bfloat16, device=dev, seed=1, logit_scale=a.logit_scale)
    target = SyntheticLM(shapes[a.target], dtype=torch.bfloat16, device=dev, seed=2, logit_scale=a.logit_scale)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    V = shapes[a.target].vocab
    torch.manual_seed(0)
    pred = MinimalQualityPredictor().eval()
    ver = SpeculativeVerifier(a.batch, a.draft_len, V, predictor=pred, lambda_value=1.0, stage_costs=(1.0, 4.5, 10.0))
    prompt = torch.randint(0, V, (a.batch, a.prompt_len), device=dev)
    feat = torch.zeros((a.batch, 64), device=dev)
    loop = speculative_generate_ragged if a.ragged else speculative_generate
    loop(draft, target, prompt, a.draft_len + 1, ver, seed=3, feat=feat)      # warm-up (GEMM autotune)
    torch.cuda.synchronize()
    t0 = time.time()
    tr = loop(draft, target, prompt, a.new_tokens, ver, seed=4, feat=feat, keep_inputs=True)
    torch.cuda.synchronize()
    loop_s = time.time() - t0

    # hot-path share: replay the recorded step inputs through the two kernels alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for inp in tr.step_inputs:
        ver.step(inp["logits"], inp["tok"], inp["lp_d"], inp["u"], feat)
    e1.record()
    torch.cuda.synchronize()
    hot_ms = e0.elapsed_time(e1)

    # parity of the first steps against the oracle
    from oracle import oracle as O
    checked = mism = 0
    for inp, mask in list(zip(tr.step_inputs, tr.accept_masks))[: a.check_steps]:
        store = inp["logits"].view(torch.int16).cpu().numpy().view(np.uint16).reshape(a.batch * a.draft_len, V)
        ref = O.verify_accept(store, O.DT_BF16, inp["tok"].cpu().numpy(), inp["lp_d"].cpu().numpy(),
                              inp["u"].cpu().numpy(), a.batch, a.draft_len, V, n_threads=16)
        ok = ~(ref["margin"] < 1e-4)
        mism += int((mask.cpu().numpy()[ok] != ref["accept"][ok]).sum())
        checked += int(ok.sum())
    acc_rate = float(np.mean([m.float().mean().item() for m in tr.accept_masks]))
    committed = int(tr.seq_len.sum().item()) - a.batch * a.prompt_len if a.ragged else tr.tokens.numel()
    out = dict(draft=shapes[a.draft].name, target=shapes[a.target].name, batch=a.batch, draft_len=a.draft_len, vocab=V,
               params_B=[round(shapes[a.draft].param_count() / 1e9, 2), round(shapes[a.target].param_count() / 1e9, 2)],
               hbm_GB_allocated=round(torch.cuda.max_memory_allocated() / 1e9, 1), build_s=round(build_s, 1),
               steps=tr.steps, verified_tokens=tr.verified_tokens, loop_s=loop_s,
               verified_tokens_per_s=tr.verified_tokens / loop_s, committed_tokens=committed,
               committed_tokens_per_s=committed / loop_s, ms_per_step=1e3 * loop_s / tr.steps,
               hot_path_ms_per_step=hot_ms / tr.steps, hot_path_share=hot_ms / 1e3 / loop_s,
               mean_accept_rate=acc_rate, mask_positions_checked=checked, mask_mismatches=mism,
               commit="per-sequence (ragged KV, asd_commit_step)" if a.ragged else "lock-step min_b(n_acc)+1",
               tokens_per_step_per_seq=tr.verified_tokens / tr.steps / a.batch)
    if not a.ragged:
        out["stop_rate"] = float(np.mean([s.float().mean().item() for s in tr.stop_flags if s is not None]))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
    assert mism == 0, "accept mask differs from the oracle"


if __name__ == "__main__":
    main()
