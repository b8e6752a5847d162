#!/usr/bin/env python3
"""asd_draft_sample: time against the workgroups per row G (forced through the TEST build's asd_debug_draft_groups), any G -- not
only powers of two -- beside the heuristic; results must not depend on G (thr / lp bit for bit, tok except at CDF tile edges).
Dev tool.   python tools/sweep_draft_groups.py [--batches 33,40,48,65] [--groups 1,2,3,4,5,6,7,8] [--out gpurun_out/draft_groups.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from asd_amd import kernels as K  # noqa: E402


def timed(fn, reps=100, settle=30):
    for _ in range(settle):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="33,40,48,65,80,100")
    ap.add_argument("--groups", default="1,2,3,4,5,6,7,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "draft_groups.json"))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    V = 152064
    hooks = K.test_hooks().__enter__()
    res = {}
    for B in [int(x) for x in a.batches.split(",")]:
        g = torch.Generator(device=dev).manual_seed(B)
        rows = [(torch.randn((B, V), generator=g, device=dev) * 3.0).to(torch.bfloat16) for _ in range(6)]
        r = torch.rand((B,), generator=g, device=dev)
        ds = K.DraftSampler(B, V, torch.bfloat16, dev)
        i = [0]

        def draft(top_p):
            i[0] += 1
            return ds(rows[i[0] % 6], r, 1 / 0.7, top_p)
        rec = {}
        hooks.asd_debug_draft_groups(0)
        ref = ds(rows[0], r, 1 / 0.7, 0.9)
        torch.cuda.synchronize()
        rec["heuristic"] = {"top_p": timed(lambda: draft(0.9)), "plain": timed(lambda: draft(1.0))}
        for G in [int(x) for x in a.groups.split(",")]:
            if B * G > 256:
                continue
            hooks.asd_debug_draft_groups(G)
            try:
                got = ds(rows[0], r, 1 / 0.7, 0.9)
                torch.cuda.synchronize()
                same = bool(torch.equal(got.thr, ref.thr)) and bool(torch.equal(got.lp.view(torch.int32), ref.lp.view(torch.int32)))
                tok_diff = int((got.tok != ref.tok).sum())
                rec[f"G{G}"] = {"top_p": timed(lambda: draft(0.9)), "plain": timed(lambda: draft(1.0)), "thr_lp_equal": same, "tok_diff": tok_diff,
                                "status": ds.status()}
            finally:
                hooks.asd_debug_draft_groups(0)
        # the commit draw (asd_residual_sample), groups forced through asd_debug_residual_groups; B <= 64 only (the group form's range)
        if B <= 64:
            Kd = 8
            t3 = torch.stack([rows[j % 6] for j in range(Kd)], 1).contiguous()
            d3 = torch.stack([rows[(j + 3) % 6] for j in range(Kd)], 1).contiguous()
            n_acc = torch.randint(0, Kd + 1, (B,), generator=g, device=dev, dtype=torch.int32)
            rs = K.ResidualSampler(B, V, torch.bfloat16, dev)
            hooks.asd_debug_residual_groups(0)
            ref_tok = rs(t3, d3, n_acc, r, rows[0], 1 / 0.7).clone()
            torch.cuda.synchronize()
            rec["residual_heuristic"] = timed(lambda: rs(t3, d3, n_acc, r, rows[0], 1 / 0.7))
            for G in [int(x) for x in a.groups.split(",")]:
                if B * G > 256:
                    continue
                hooks.asd_debug_residual_groups(G)
                try:
                    got = rs(t3, d3, n_acc, r, rows[0], 1 / 0.7).clone()
                    torch.cuda.synchronize()
                    rec[f"residual_G{G}"] = {"us": timed(lambda: rs(t3, d3, n_acc, r, rows[0], 1 / 0.7)), "tok_equal": bool(torch.equal(got, ref_tok)),
                                             "status": rs.status()}
                finally:
                    hooks.asd_debug_residual_groups(0)
        print(B, json.dumps(rec), flush=True)
        res[f"B{B}"] = rec
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
