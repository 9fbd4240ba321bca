"""Diagnostic (GPU box): is the gradient-direction gap between our bf16 pipeline and the fp32 oracle
inherent to bf16?  Compares (a) fp32 CPU oracle, (b) torch's own bf16 autocast on the GPU, (c) ours."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import resnet as orn
from ssl_wafermap_amd import ops
from ssl_wafermap_amd.heads import SimCLRProjectionHead
from ssl_wafermap_amd.loss import NTXentLoss
from ssl_wafermap_amd.models import create_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
backbone, head = create_model("resnet18", num_classes=0), SimCLRProjectionHead(512, 512, 128)
for m in backbone.modules():
    if hasattr(m, "bn2"):
        torch.nn.init.constant_(m.bn2.weight, 0.5)
sd = {"backbone." + k: v.clone() for k, v in backbone.state_dict().items()}
sd.update({"projection_head." + k: v.clone() for k, v in head.state_dict().items()})
sd = {k: (v.bfloat16().float() if v.dtype == torch.float32 and v.dim() > 1 else v) for k, v in sd.items()}
backbone.load_state_dict({k[9:]: v for k, v in sd.items() if k.startswith("backbone.")})
head.load_state_dict({k[16:]: v for k, v in sd.items() if k.startswith("projection_head.")})
g = torch.Generator().manual_seed(1)
lut = torch.tensor([-1.5366, 0.1790, 1.8811]).bfloat16().float()
x0 = lut[torch.randint(0, 3, (B, 1, S, S), generator=g)].expand(-1, 3, -1, -1).contiguous()
x1 = lut[torch.randint(0, 3, (B, 1, S, S), generator=g)].expand(-1, 3, -1, -1).contiguous()


def run_oracle(device, autocast):
    params = {k: v.clone().to(device).requires_grad_(v.dtype == torch.float32 and "running" not in k) for k, v in sd.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        loss, _ = orn.simclr_loss(x0.to(device), x1.to(device), params, 0.5, True)
    loss.backward()
    return float(loss), {k: p.grad.float().cpu() for k, p in params.items() if p.requires_grad}


la, ga = run_oracle("cpu", False)
lb, gb = run_oracle("cuda", True)
lb32, gb32 = run_oracle("cuda", False)
backbone.cuda().train(); head.cuda().train()
with ops.bn_groups(2):
    z = head(backbone(torch.cat([x0, x1]).cuda()))
loss = NTXentLoss(0.5)(z[:B], z[B:])
loss.backward()
gc = {"backbone." + n: p.grad.float().cpu() for n, p in backbone.named_parameters()}
gc.update({"projection_head." + n: p.grad.float().cpu() for n, p in head.named_parameters()})
print(f"loss fp32cpu {la:.5f}  fp32gpu {lb32:.5f}  torch-bf16 {lb:.5f}  ours {float(loss):.5f}")
cos = lambda a, b: F.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
print(f"{'param':45s} cpu32~gpu32 cpu32~torchbf16 cpu32~ours torchbf16~ours |ours|/|ref|")
for k in ga:
    print(f"{k:45s} {cos(ga[k], gb32[k]):.4f}      {cos(ga[k], gb[k]):.4f}        {cos(ga[k], gc[k]):.4f}     {cos(gb[k], gc[k]):.4f}      {gc[k].norm() / ga[k].norm():.3f}")
