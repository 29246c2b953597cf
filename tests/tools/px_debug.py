import sys, torch
sys.path.insert(0, "/root/repo")
import bench as B
from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
dev = torch.device("cuda:0")
G, n = int(sys.argv[1]), 144
torch.manual_seed(0)
m = GNN_Diffusion(steps=100, sampling="DDIM", rotation=True, visual_pretrained=False, model_mean_type=ModelMeanType.EPSILON, backbone="resnet18equiv", freeze_backbone=False).to(dev).train()
opt = m.configure_optimizers()
gen = torch.Generator(device=dev).manual_seed(99)
crops = torch.rand((G * n, 3, 32, 32), generator=gen, device=dev)
x0 = torch.randn((G * n, 4), generator=gen, device=dev)
ei, batch = B.dense_batch(G, n, dev)
for it in range(8):
    t = torch.randint(0, 100, (G,), generator=gen, device=dev)[batch]
    opt.zero_grad()
    feats = m.visual_features(crops)
    loss = m.p_losses(x0, t, loss_type="huber", cond=None, edge_index=ei, batch=batch, patch_feats=feats)
    loss.backward()
    enc = m.model.visual_backbone
    bad = [k for k, p in enc.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    gmax = max(float(p.grad.abs().max()) for p in enc.parameters() if p.grad is not None)
    print(it, float(loss), "feats", float(feats.abs().max()), "finite feats", bool(torch.isfinite(feats).all()), "bad grads", bad[:3], "gmax %.3g" % gmax, flush=True)
    opt.step()
