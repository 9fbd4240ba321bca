import os, sys, torch
sys.path.insert(0, "/root/repo")
from ssl_wafermap_amd import vit_ops, optim, ops
dev = "cuda:0"
for c, rows_list in ((384, (13824, 25088)), (192, (300,)), (2048, (512,))):
    res = {}
    for flag in ("1", "0"):
        os.environ["WM_LN_SLOTS"] = flag
        torch.manual_seed(0)
        bias = torch.nn.Parameter(torch.randn(c, device=dev) * 0.1)
        opt = optim.AdamW([bias], lr=1e-3)
        opt.zero_grad()
        g = torch.Generator(device=dev).manual_seed(1)
        outs, dys = [], []
        for rows in rows_list:
            x = torch.randn(rows, c, generator=g, device=dev).bfloat16().requires_grad_(True)
            outs.append(vit_ops.bias_act(x, bias))
            dys.append(torch.randn(rows, c, generator=g, device=dev).bfloat16())
        torch.autograd.backward(outs, dys)
        torch.cuda.synchronize()
        res[flag] = bias.grad.clone()
        ref = sum(d.float().sum(0) for d in dys)
        print(c, rows_list, flag, float((bias.grad - ref).norm() / ref.norm()))
