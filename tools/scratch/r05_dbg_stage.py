import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import cases as CC
from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
dev = torch.device("cuda:0")
for name in ("rot144_g2_sharp", "exo144_v8_g2"):
    spec = CC.by_name(name); case = CC.build_case(spec)
    outs = {}
    for staged in (False, True):
        kw = dict(architecture="exophormer", virt_nodes=spec.get("virt_nodes", 8)) if spec.get("arch") == "exophormer" else {}
        m = GNN_Diffusion(steps=spec["steps"], sampling="DDIM", rotation=True, visual_pretrained=False, model_mean_type=ModelMeanType.EPSILON, **kw)
        m.model.load_state_dict(case["sd"], strict=False)
        m = m.to(dev).train()
        te = m.model.train_engine(dev); te.force_staged = staged
        g = torch.Generator().manual_seed(5)
        x0 = case["x"].to(dev)
        t = case["t"].to(dev)
        noise = torch.randn(x0.shape, generator=g).to(dev)
        for mb in range(2):
            loss = m.p_losses(x0, t, noise=noise * (1 + 0.25 * mb), loss_type="huber", cond=None, edge_index=case["edge_index"].to(dev), batch=case["batch"].to(dev), patch_feats=case["feats"].to(dev))
            loss.backward()
        torch.cuda.synchronize()
        outs[staged] = te.flat_grad.clone()
        names, offs = te.names, [v.data_ptr() for v in te.grad_views]
    d = (outs[True] - outs[False]).abs()
    print(name, "staged vs all: max diff", float(d.max()), "equal", torch.equal(outs[True], outs[False]), "nan", int(torch.isnan(outs[True]).sum()), int(torch.isnan(outs[False]).sum()))
    base = te.flat_grad.data_ptr()
    for n_, gv in zip(te.names, te.grad_views):
        o = (gv.data_ptr() - base) // 4
        dd = float(d[o:o + gv.numel()].max())
        if dd > 0: print("   ", n_, dd, float(outs[False][o:o + gv.numel()].abs().max()))
