"""The fp64 backward oracle against torch autograd (fp64) -- CPU only. The reference holds no usable
fixture for the backward (its CPU check is broken, main.mm:1100-1101), so this is how it is anchored."""
import numpy as np


def test_bwd_oracle_matches_autograd(oracle_mod):
    import torch
    import torch.nn.functional as F

    rng = np.random.default_rng(0)
    for (B, H, N, D) in ((1, 2, 40, 16), (2, 1, 33, 64)):
        q, k, v, do = (rng.standard_normal((B, H, N, D)).astype(np.float32) for _ in range(4))
        for causal in (False, True):
            dq, dk, dv = oracle_mod.attn_bwd_f64(q, k, v, do, causal)
            tq, tk, tv = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (q, k, v))
            F.scaled_dot_product_attention(tq, tk, tv, is_causal=causal).backward(torch.tensor(do, dtype=torch.float64))
            for g, t in ((dq, tq), (dk, tk), (dv, tv)):
                assert np.abs(g - t.grad.numpy()).max() < 1e-12
