import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asd_amd import kernels as K
B, Kk, D, V = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
h = torch.randn((B * Kk, D), device="cuda").to(torch.bfloat16)
w = torch.randn((V, D), device="cuda").to(torch.bfloat16)
tok = torch.zeros((B, Kk), dtype=torch.int32, device="cuda")
f = torch.rand((B, Kk), device="cuda")
ver = K.LmHeadVerifier(w, B, Kk)
print("hidden %x..%x weight %x..%x ws %x..%x" % (h.data_ptr(), h.data_ptr() + h.numel() * 2, w.data_ptr(), w.data_ptr() + w.numel() * 2,
      ver.workspace.data_ptr(), ver.workspace.data_ptr() + ver.workspace.numel()), flush=True)
r = ver(h, tok, -f, f)
torch.cuda.synchronize()
ref = (h.float() @ w.float().T).log_softmax(-1)[:, 0].reshape(B, Kk)
print("max err", (r.lp_target - ref).abs().max().item(), flush=True)
msg = ver.workspace[: ((V + 127) // 128) * B * Kk * 12].view(torch.float32).reshape(-1, B * Kk, 3)
x = (h.float() @ w.float().T).double()
for un in range(msg.shape[0]):
    cols = x[:, un * 128:(un + 1) * 128] * 1.4426950408889634
    m2 = cols.max(-1).values
    s = torch.exp2(cols - m2[:, None]).sum(-1)
    print("unit", un, "gpu", msg[un, 0].tolist(), "ref", [m2[0].item(), s[0].item()], flush=True)
