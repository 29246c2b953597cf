"""One rank of an RCCL ("nccl") process group on cuda:0 -- world size 1, which a 1-GPU box can run -- exercising the calls the
8-GPU node makes for BASELINE config 5 (the reference: Lightning's NCCL DDP, train_script.py:215-218):
``sharding.allreduce_gradients`` on the RCCL branch, ``TrainEngine.sync_gradients`` with the bucketed / overlapped exchange
(early bucket all-reduced on a side stream under conv 0's backward) and without it, ``GNN_Diffusion.on_before_optimizer_step``,
``FusedAdafactor.step``.  Prints one JSON line; tests/test_gpu_rccl.py compares it with the same steps run WITHOUT a process
group (argv[1] == "nodist")."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    mode = sys.argv[1]
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if mode != "nodist":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", sys.argv[2])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    os.environ["DIFFASSEMBLE_FUSED_OPTIMIZER"] = "1"
    import cases as CC
    from diffassemble_amd import sharding as S
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    spec = CC.by_name("rot144_g2_sharp")
    case = CC.build_case(spec)
    out = {"mode": mode, "exchange_active": bool(S.exchange_active()), "backend": dist.get_backend() if dist.is_initialized() else None}
    res = {}
    for overlap in ((True, False) if mode != "nodist" else (False,)):
        m = GNN_Diffusion(steps=spec["steps"], sampling="DDIM", rotation=True, visual_pretrained=False,
                          model_mean_type=ModelMeanType.EPSILON)
        m.model.load_state_dict(case["sd"], strict=False)
        m = m.to(dev).train()
        opt = m.configure_optimizers()
        te = m.model.train_engine(dev)
        te.overlap_exchange = overlap
        g = torch.Generator().manual_seed(5)
        x0 = case["x"].to(dev)
        losses, early_seen = [], []
        for it in range(3):
            t = torch.randint(0, spec["steps"], (2,), generator=g)[case["batch"]].to(dev)
            noise = torch.randn(x0.shape, generator=g).to(dev)
            opt.zero_grad()
            # two micro-batches accumulated before the exchange (the early bucket is exchanged after each backward)
            for mb in range(2):
                loss = m.p_losses(x0, t, noise=noise * (1.0 + 0.25 * mb), loss_type="huber", cond=None, edge_index=case["edge_index"].to(dev),
                                  batch=case["batch"].to(dev), patch_feats=case["feats"].to(dev))
                loss.backward()
                early_seen.append(bool(te._early_pending))
            if it % 2 == 0:
                m.on_before_optimizer_step(opt)         # Lightning's hook
                assert te.grads_synced
            if it == 0:
                torch.cuda.synchronize()
                res.setdefault("grad", {})[overlap] = te.flat_grad.detach().clone()
            opt.step()                                   # FusedAdafactor: exchanges by itself when the hook did not
            losses.append(float(loss))
        torch.cuda.synchronize()
        res.setdefault("flat", {})[overlap] = te.flat.detach().clone()
        out[f"losses_overlap_{int(overlap)}"] = losses
        out[f"early_pending_seen_overlap_{int(overlap)}"] = early_seen
        out[f"flat_sum_overlap_{int(overlap)}"] = float(te.flat.double().sum())
        out[f"grad_abs_sum_overlap_{int(overlap)}"] = float(res["grad"][overlap].double().abs().sum())
        out["bucket_split"] = [int(te.early_off), int(te.total)]
    if mode != "nodist":
        # (not bit-equal: k_time_scatter adds the time_emb rows' gradients with atomics, in whatever order the chunks arrive)
        reldiff = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
        out["overlap_vs_serial_grad"] = reldiff(res["grad"][True], res["grad"][False])
        out["overlap_vs_serial_params"] = reldiff(res["flat"][True], res["flat"][False])
        # the RCCL branch of the plain helper on a scratch tensor
        v = torch.arange(1024, dtype=torch.float32, device=dev)
        S.allreduce_gradients(v, average=True)
        torch.cuda.synchronize()
        out["allreduce_identity"] = bool(torch.equal(v, torch.arange(1024, dtype=torch.float32, device=dev)))
        dist.destroy_process_group()
    torch.save({k: {kk: vv.cpu() for kk, vv in d.items()} for k, d in res.items()}, sys.argv[3])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
