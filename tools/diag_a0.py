import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_metal_amd as fa
import flash_attention_metal_amd._lib as _l; _l._SO = os.environ["FA_LIB"]
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = ((torch.rand(1, 1, 128, 64, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(3))
o, l = fa.flash_attention_forward(q, k, v, is_causal=False, variant="mfma_pp")
torch.cuda.synchronize()
print(os.environ["FA_LIB"], "lane(r,h=0) a0/a1 alternating:", [round(x, 4) for x in l[0, 0, :16].tolist()], "h=1:", [round(x, 4) for x in l[0, 0, 32:40].tolist()])
