"""One model pass of the token-level loop in isolation (for rocprofv3 --kernel-trace --stats): a Qwen2.5-shape synthetic model,
B sequences, a prefill of `prompt` tokens, then `passes` forward_ragged calls of T tokens each.
    python tools/profile_pass.py --model 7b --batch 32 --tokens 1 --passes 8 [--graphs] [--hip-layers]"""
import argparse
import sys
import time
from importlib import import_module
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=1)
    ap.add_argument("--passes", type=int, default=8)
    ap.add_argument("--prompt", type=int, default=32)
    ap.add_argument("--cap", type=int, default=256)
    ap.add_argument("--graphs", action="store_true")
    ap.add_argument("--hip-layers", action="store_true")
    ap.add_argument("--hidden-only", action="store_true")
    ap.add_argument("--pack", action="store_true", help="projection matrices re-laid tile-major in place")
    a = ap.parse_args()
    lm = SL.SyntheticLM(SL.QWEN25_SHAPES[a.model], device="cuda", seed=1, logit_scale=0.6)
    if a.hip_layers:
        lm.enable_hip_layers(pack_weights=a.pack)
    lm.alloc_ragged(a.batch, a.cap)
    g = torch.Generator(device="cuda").manual_seed(3)
    ids = torch.randint(0, lm.shape.vocab, (a.batch, a.prompt), generator=g, device="cuda")
    pos0 = torch.zeros(a.batch, dtype=torch.int64, device="cuda")
    lm.forward_ragged(ids, pos0, a.prompt, return_hidden=True)
    if a.graphs:
        lm.enable_graphs(True)
    pos = a.prompt
    step_ids = torch.randint(0, lm.shape.vocab, (a.batch, a.tokens), generator=g, device="cuda")
    for i in range(3):
        lm.forward_ragged(step_ids, pos0 + pos, pos + a.tokens, return_hidden=a.hidden_only)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.passes):
        out = lm.forward_ragged(step_ids, pos0 + pos, pos + a.tokens, return_hidden=a.hidden_only)
        pos += a.tokens
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.passes
    print(f"{a.model} B={a.batch} T={a.tokens} graphs={a.graphs} hip_layers={a.hip_layers}: {dt * 1e3:.3f} ms per pass; out {tuple(out.shape)}", flush=True)


if __name__ == "__main__":
    main()
