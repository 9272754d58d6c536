"""Launch the opt-pass backward / loss kernels of the 64^2 self-attention layer (5 heads x f... as in the edit) and of a cross-attention
layer a few times (for rocprofv3 --pmc runs): k_attn_bwd_dq, k_attn_bwd_dk, k_attn_probs, k_corr_max, k_removal_bwd."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from geodiffuser_amd import ops
dev, dt = "cuda", torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(0)
SHAPES = {"self": (5, 4096, 4096), "cross": (5, 4096, 77)}
for (H, N, M) in (SHAPES[os.environ.get("BWD_SHAPE", "self")],):
    q = (torch.randn(H, N, 64, device=dev) * 1.2).to(dt); k = (torch.randn(H, M, 64, device=dev) * 1.2).to(dt); v = torch.randn(H, M, 64, device=dev).to(dt)
    g = (torch.randn(H, N, 64, device=dev) * 0.1).to(dt)
    out = torch.empty_like(q); lse = torch.empty(H, N, device=dev)
    ops.attn_fwd([(q, k, v, out, lse)], 0.125)
    R = int(os.environ.get("BWD_ROWS", "640"))                 # inpaint rows of a typical object mask at 64^2 (bucketed to 256)
    rows = (torch.arange(0, R, device=dev, dtype=torch.int32) * 5 % N).contiguous()
    nv = torch.tensor([int(os.environ.get("BWD_NV", str(R - 100)))], dtype=torch.int32, device=dev)
    m_inp = torch.zeros(N, device=dev); m_inp[rows.long()] = 1; m_wo = 1 - m_inp
    for _ in range(reps):
        dq, dk = ops.attn_bwd(q, k, v, out, lse, g, 0.125, M == 77)
        Pb = ops.attn_probs(q, k, lse, None, 0.125); Pe = ops.attn_probs(q, k, lse, rows, 0.125, n_valid=nv)
        aux, rm = ops.removal_fwd(Pe, Pb, m_inp, m_wo, rows, 64, n_valid=nv)
        dq32 = torch.zeros(H, N, 64, device=dev); dk32 = torch.zeros(H, M, 64, device=dev) if M == 77 else None
        ops.removal_bwd(Pe, Pb, q, k, rows, aux, m_inp, m_wo, 0.01, None, 0.125, dq32, dk32, n_valid=nv)
torch.cuda.synchronize()
