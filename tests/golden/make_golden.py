"""Generate tests/golden/golden_v1.npz from the REFERENCE's own code.

BUILD-CONTAINER ONLY: imports /root/reference/puzzle_diff/model/*.py under the stubs of
ref_import.py (PyG's TransformerConv and pytorch3d's quaternion conversions bound to
oracle/pyg_restatement.py -- the reference does not vendor them), loads the seeded
weights of oracle/weights.py into the reference modules, runs the reference's
forward_with_feats / p_sample_loop / p_sample_ddpm / p_losses and stores the OUTPUTS.
Inputs and weights are not stored: tests regenerate them from the seeds (cases.py).

Run:  python tests/golden/make_golden.py
"""
import sys

sys.dont_write_bytecode = True
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases as C  # noqa: E402
from ref_import import import_reference  # noqa: E402

torch.set_num_threads(8)
sd2, sd3 = import_reference()
OUT = {}


def put(case, field, t):
    OUT[f"{case}/{field}"] = (t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t))


def stats(t):
    t = t.double()
    return torch.stack([t.sum(), t.abs().sum(), (t * t).sum()]).float()


def load_weights(module, sd):
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    dead = ("linear1.", "linear2.", "visual_backbone.", "pcd_backbone.", "mean", "std")
    bad = [k for k in missing if not k.startswith(dead)]
    assert not bad, bad


def ref_model_2d(spec, sampling="DDIM", ratio=1, mean="START_X", noise_weight=1.0, steps=None,
                 cf_w=0.0, cf_p=0.0):
    m = sd2.GNN_Diffusion(
        steps=steps or spec["steps"], sampling=sampling, inference_ratio=ratio,
        noise_weight=noise_weight, rotation=(spec["c"] == 4),
        model_mean_type=getattr(sd2.ModelMeanType, mean), visual_pretrained=False,
        architecture=spec["arch"], virt_nodes=spec["V"], classifier_free_w=cf_w,
        classifier_free_prob=cf_p)
    m.eval()
    return m


def ref_model_3d(spec, ratio=1, mean="START_X", noise_weight=1.0):
    m = sd3.GNN_Diffusion(
        steps=spec["steps"], sampling="DDIM", inference_ratio=ratio, noise_weight=noise_weight,
        model_mean_type=getattr(sd3.ModelMeanType, mean), backbone="vn_dgcnn",
        architecture=spec["arch"])
    m.eval()
    return m


def hook_acts(model):
    acts = []
    hs = [model.mlp.register_forward_hook(lambda m, i, o: acts.append(o))]
    for conv in model.gnn_backbone.module_list:
        hs.append(conv.register_forward_hook(
            lambda m, i, o: acts.append(o[0] if isinstance(o, tuple) else o)))
    return acts, hs


# ----------------------------------------------------------------------------- schedules
for T in C.SCHEDULE_T:
    m = sd2.GNN_Diffusion(steps=T, sampling="DDIM", visual_pretrained=False)
    for k in ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas",
              "sqrt_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "posterior_variance"):
        put(f"schedule_T{T}", k, getattr(m, k))
    if T == 50:
        keys = sorted(k for k in m.state_dict() if not k.startswith("model.visual_backbone"))
        put("statedict_2d", "keys", np.array(keys))
        put("statedict_2d", "shapes", np.array([str(tuple(m.state_dict()[k].shape)) for k in keys]))

# ----------------------------------------------------------------------------- 2D forward
for spec in C.FWD2D:
    case = C.build_case(spec)
    m = ref_model_2d(spec)
    load_weights(m.model, case["sd"])
    acts, hs = hook_acts(m.model)
    with torch.no_grad():
        out, att = m.forward_with_feats(case["x"], case["t"], None, case["edge_index"],
                                        case["feats"], case["batch"], return_attentions=True)
    for h in hs:
        h.remove()
    put(spec["name"], "out", out)
    ei_last, alpha_last = att[-1]
    put(spec["name"], "n_att", len(att))
    put(spec["name"], "alpha_last_stats", stats(alpha_last))
    put(spec["name"], "alpha_last_head", alpha_last[:256])
    put(spec["name"], "alpha_last_tail", alpha_last[-256:])
    put(spec["name"], "ei_last_shape", np.array(ei_last.shape))
    put(spec["name"], "ei_last_tail", ei_last[:, -4096:])
    put(spec["name"], "ei_last_checksum",
        np.array([int(ei_last[0].sum()), int(ei_last[1].sum()),
                  int((ei_last[0] * 7 + ei_last[1] * 13).remainder(1000003).sum())]))
    for i, a in enumerate(acts):
        put(spec["name"], f"act{i}_stats", stats(a))
        put(spec["name"], f"act{i}_rows", a[:: max(1, a.shape[0] // 8), :64])
    print("fwd2d", spec["name"], tuple(out.shape), "E'", ei_last.shape[1], flush=True)

# ----------------------------------------------------------------------------- 2D loops
for lp in C.LOOPS2D:
    spec = C.by_name(lp["base"])
    case = C.build_case(spec)
    m = ref_model_2d(spec, sampling=lp["sampling"], ratio=lp["ratio"], mean=lp["mean"],
                     noise_weight=lp["noise_weight"], steps=lp["T"])
    load_weights(m.model, case["sd"])
    m.visual_features = lambda cond: case["feats"]          # encoder bypassed
    if lp.get("max_iters"):
        # shorten by monkey-patching `range` is fragile; run the reference's own p_sample instead
        torch.manual_seed(123)
        img = torch.randn(case["x"].shape) * lp["noise_weight"]
        put(lp["name"], "x_init", img)
        imgs = []
        b = img.shape[0]
        for i in list(reversed(range(0, lp["T"], lp["ratio"])))[: lp["max_iters"]]:
            img, _ = m.p_sample(img, torch.full((b,), i, dtype=torch.long), i, cond=None,
                                edge_index=case["edge_index"], patch_feats=case["feats"],
                                batch=case["batch"])
            imgs.append(img)
    else:
        torch.manual_seed(123)
        x_init = torch.randn(case["x"].shape) * lp["noise_weight"]
        put(lp["name"], "x_init", x_init)
        torch.manual_seed(123)
        with torch.no_grad():
            imgs, _ = m.p_sample_loop(case["x"].shape, None, case["edge_index"], case["batch"])
    put(lp["name"], "imgs", torch.stack(imgs))
    print("loop2d", lp["name"], len(imgs), flush=True)

# DDPM: direct p_sample_ddpm call with saved noise; and the loop's failure mode.
spec = C.by_name("k36_noloop_eps")
case = C.build_case(spec)
m = ref_model_2d(spec, sampling="DDPM", mean="EPSILON")
load_weights(m.model, case["sd"])
t = torch.full((36,), 17, dtype=torch.long)
torch.manual_seed(5)
noise = torch.randn_like(case["x"])
torch.manual_seed(5)
y = m.p_sample_ddpm(case["x"], t, 17, None, case["edge_index"], case["feats"], case["batch"])
put("ddpm_direct", "noise", noise)
put("ddpm_direct", "out_t17", y)
y0 = m.p_sample_ddpm(case["x"], t * 0, 0, None, case["edge_index"], case["feats"], case["batch"])
put("ddpm_direct", "out_t0", y0)
m.visual_features = lambda cond: case["feats"]
try:
    m.p_sample_loop(case["x"].shape, None, case["edge_index"], case["batch"])
    raised = ""
except Exception as e:  # noqa: BLE001
    raised = f"{type(e).__name__}: {e}"
put("ddpm_direct", "loop_raises", np.array(raised))
print("ddpm loop raises:", raised, flush=True)

# classifier-free guidance branch of p_sample_ddim (spatial_diffusion.py:568-589)
spec = C.by_name("k36_loop_sharp")
case = C.build_case(spec)
m = ref_model_2d(spec, mean="START_X", cf_w=0.5, cf_p=0.1)
load_weights(m.model, case["sd"])
t = torch.full((36,), 30, dtype=torch.long)
y, _ = m.p_sample_ddim(case["x"], t, 30, None, case["edge_index"], case["feats"], case["batch"])
put("cfg_ddim", "out_t30", y)

# ----------------------------------------------------------------------------- 3D
for spec in C.FWD3D:
    case = C.build_case(spec, "3d")
    m = ref_model_3d(spec)
    load_weights(m.model, case["sd"])
    acts, hs = hook_acts(m.model)
    with torch.no_grad():
        out, att = m.forward_with_feats(case["x"], case["t"], case["edge_index"], case["feats"],
                                        case["batch"])
    for h in hs:
        h.remove()
    put(spec["name"], "out", out)
    put(spec["name"], "alpha_last_stats", stats(att[-1][1]))
    for i, a in enumerate(acts):
        put(spec["name"], f"act{i}_stats", stats(a))
    print("fwd3d", spec["name"], tuple(out.shape), flush=True)

for lp in C.LOOPS3D:
    spec = C.by_name(lp["base"])
    case = C.build_case(spec, "3d")
    m = ref_model_3d(spec, ratio=lp["ratio"], mean=lp["mean"], noise_weight=lp["noise_weight"])
    load_weights(m.model, case["sd"])
    b = case["x"].shape[0]
    torch.manual_seed(321)
    tr = torch.randn((b, 3)) * lp["noise_weight"]
    quat = sd3.matrix_to_quaternion(torch.eye(3).repeat(b, 1, 1))
    img = torch.cat([quat, tr], 1)
    put(lp["name"], "x_init", img)
    imgs = []
    for i in list(reversed(range(0, lp["T"], lp["ratio"])))[: lp["max_iters"]]:
        img, _ = m.p_sample(img, torch.full((b,), i, dtype=torch.long), i,
                            edge_index=case["edge_index"], pcd_feats=case["feats"],
                            batch=case["batch"])
        imgs.append(img)
    put(lp["name"], "imgs", torch.stack(imgs))
    print("loop3d", lp["name"], len(imgs), flush=True)

# ----------------------------------------------------------------------------- training
for tr in C.TRAIN2D:
    spec = C.by_name(tr["base"])
    case = C.build_case(spec)
    m = ref_model_2d(spec, mean=tr["mean"])
    load_weights(m.model, case["sd"])
    m.train()
    m.visual_features = lambda cond: case["feats"]
    rng = np.random.default_rng(tr["seed"])
    noise = torch.from_numpy(rng.standard_normal(tuple(case["x"].shape)).astype(np.float32))
    x_start = case["x"]
    loss = m.p_losses(x_start, case["t"], noise=noise, loss_type="huber", cond=None,
                      edge_index=case["edge_index"], batch=case["batch"])
    loss.backward()
    put(tr["name"], "loss", loss)
    for k, p in m.model.named_parameters():
        if p.grad is not None and k in case["sd"]:
            put(tr["name"], f"grad_stats/{k}", stats(p.grad))
            put(tr["name"], f"grad_head/{k}", p.grad.flatten()[:64])
    print("train", tr["name"], float(loss), flush=True)

np.savez_compressed(C.GOLDEN_FILE, **OUT)
import os  # noqa: E402
print("wrote", C.GOLDEN_FILE, os.path.getsize(C.GOLDEN_FILE), "bytes,", len(OUT), "arrays")
