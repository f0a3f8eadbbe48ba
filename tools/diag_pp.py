#!/usr/bin/env python3
"""Localise differences between two variants on one shape: per 32-row block and per 32-column block max |diff|."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_metal_amd as fa
if os.environ.get("FA_LIB"):
    import flash_attention_metal_amd._lib as _l; _l._SO = os.environ["FA_LIB"]
B, H, N, D = map(int, sys.argv[1:5]); dtype = sys.argv[5]; causal = bool(int(sys.argv[6]))
va, vb = sys.argv[7], sys.argv[8]
tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtype]
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = ((torch.rand(B, H, N, D, generator=g, device="cuda") * 2 - 1).to(tdt) for _ in range(3))
oa, la = fa.flash_attention_forward(q, k, v, is_causal=causal, variant=va)
ob, lb = fa.flash_attention_forward(q, k, v, is_causal=causal, variant=vb)
torch.cuda.synchronize()
d = (oa.float() - ob.float()).abs()
print("max diff O", d.max().item(), "LSE", (la - lb).abs().max().item())
for b in range(B):
    for h in range(H):
        rows = [d[b, h, i:i + 32].max().item() for i in range(0, N, 32)]
        cols = [d[b, h, :, j:j + 32].max().item() for j in range(0, D, 32)]
        print(f"b{b} h{h} rows/32:", " ".join(f"{x:.3f}" for x in rows), "| cols/32:", " ".join(f"{x:.3f}" for x in cols))
        print("   lse rows/32:", " ".join(f"{(la[b,h,i:i+32]-lb[b,h,i:i+32]).abs().max().item():.3f}" for i in range(0, N, 32)))
if os.environ.get("FA_DETAIL"):
    e = d[0, 0, :32, :32]
    print("rows with err>0.05:", [i for i in range(32) if e[i].max() > 0.05])
    print("cols with err>0.05:", [j for j in range(32) if e[:, j].max() > 0.05])
    print("ratio pp/ref row0:", (ob[0, 0, 0, :32].float() / oa[0, 0, 0, :32].float()).tolist())
    print("pp row0:", ob[0,0,0,:32].float().tolist())
    print("ref row0:", oa[0,0,0,:32].float().tolist())
