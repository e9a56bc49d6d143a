import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd"), os.path.join(ROOT, "tests")]
import torch, torch.nn.functional as F
from conftest import rel_err
from gfe_hip.gen_train import _Conv3Fn, _GroupNormFn, _LiftIn1Fn, single_conv
from pytorch3dunet.unet3d.buildingblocks import ResNetBlock
BF = torch.bfloat16
cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()
nc = lambda t: t.permute(0, 4, 1, 2, 3)
g = torch.Generator().manual_seed(17)
cin, c, (B, D, H, W) = 1, 16, (2, 8, 8, 8)
blk = ResNetBlock(cin, c)
with torch.no_grad():
    for p in blk.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    for m in blk.modules():
        if isinstance(m, torch.nn.GroupNorm):
            m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=g)); m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
x = torch.randn(B, cin, D, H, W, generator=g)
w = torch.randn(B, c, D, H, W, generator=g)
ref = {k: v.detach().clone().requires_grad_(True) for k, v in blk.named_parameters()}
r = F.conv3d(x, ref["conv1.weight"], ref["conv1.bias"]); r.retain_grad()
gn = lambda t, p: F.group_norm(t, 8, ref[p + ".groupnorm.weight"], ref[p + ".groupnorm.bias"], 1e-5)
h2 = gn(r, "conv2"); h2.retain_grad()
o2 = F.relu(F.conv3d(h2, ref["conv2.conv.weight"], padding=1)); o2.retain_grad()
h3 = gn(o2, "conv3"); h3.retain_grad()
o = F.relu(F.conv3d(h3, ref["conv3.conv.weight"], padding=1) + r)
(o * w).sum().backward()
blk = blk.cuda()
rg = _LiftIn1Fn.apply(x.cuda(), blk.conv1.weight, blk.conv1.bias); rg.retain_grad()
gnm = blk.conv2.groupnorm
h2g = _GroupNormFn.apply(rg, gnm.weight, gnm.bias, 8, 1e-5); h2g.retain_grad()
o2g = _Conv3Fn.apply(h2g, blk.conv2.conv.weight, None, True); o2g.retain_grad()
gnm3 = blk.conv3.groupnorm
h3g = _GroupNormFn.apply(o2g, gnm3.weight, gnm3.bias, 8, 1e-5); h3g.retain_grad()
og = _Conv3Fn.apply(h3g, blk.conv3.conv.weight, rg, True)
og.backward(cl(w).to(BF).cuda())
for name, a, b in (("r", rg, r), ("h2", h2g, h2), ("o2", o2g, o2), ("h3", h3g, h3), ("o", og, o)):
    print("fwd %-3s %.2e" % (name, rel_err(nc(a), b)), end="  ")
print()
for name, a, b in (("h3", h3g, h3), ("o2", o2g, o2), ("h2", h2g, h2), ("r", rg, r)):
    print("grad %-3s %.2e" % (name, rel_err(nc(a.grad), b.grad)), end="  ")
print()
print({k: "%.1e" % rel_err(p.grad, ref[k].grad) for k, p in blk.named_parameters()})
# same activation pattern: reference with our ReLU masks
m2 = (nc(o2g.detach()) > 0).float().cpu(); m3 = (nc(og.detach()) > 0).float().cpu()
print("flipped o2 %.4f o %.4f" % (((o2 > 0).float() != m2).float().mean(), ((o > 0).float() != m3).float().mean()))
ref2 = {k: v.detach().clone().requires_grad_(True) for k, v in ref.items()}
gn2 = lambda t, p: F.group_norm(t, 8, ref2[p + ".groupnorm.weight"], ref2[p + ".groupnorm.bias"], 1e-5)
r_ = F.conv3d(x, ref2["conv1.weight"], ref2["conv1.bias"])
o2_ = F.conv3d(gn2(r_, "conv2"), ref2["conv2.conv.weight"], padding=1) * m2
o_ = (F.conv3d(gn2(o2_, "conv3"), ref2["conv3.conv.weight"], padding=1) + r_) * m3
(o_ * w).sum().backward()
print({k: "%.1e" % rel_err(p.grad, ref2[k].grad) for k, p in blk.named_parameters()})
l2 = lambda a, b: ((a.detach().cpu().double() - b.detach().double()).norm() / b.detach().double().norm()).item()
print("L2 vs plain ref", {k: "%.1e" % l2(p.grad, ref[k].grad) for k, p in blk.named_parameters()})
print("w.to(bf16) err", rel_err(w.to(BF).float(), w))
