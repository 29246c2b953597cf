"""GPU parity tests of the 3D piece encoder: the HIP path through the C ABI (da_pcd_encoder_forward,
da_knn, da_nearest_sq) against the CPU oracle (oracle/vn_dgcnn.py) and against the fixtures produced by the reference's own
vnn/vn_dgcnn.py (tests/golden/make_golden_v3.py).

Tolerance, fp32 end to end, per cloud (max-abs error / max-abs of the reference tensor): below 2e-5 for all clouds but
one (or a quarter of them, when there are more than four) (measured ~1e-6 to 9e-6: summation order, the BatchNorm fold norm * scale + shift, and the first layer of a stage
evaluated per point, W x_j - W x_i instead of W (x_j - x_i)).  The neighbour search is discrete: a near-tie at rank 20 / 21
resolves differently under another summation order and swaps ONE of the 20 N edge terms of a cloud, an O(1 / (20 N))
change of the pooled output (the CPU oracle itself differs from the reference by 4e-5 at N = 1000 for that reason; one
such cloud measured 4.2e-4 at N = 200).  Every cloud must stay below 0.3 / N.  The neighbour lists themselves are
compared as sets, 99.9 % of the entries must agree."""
import numpy as np
import pytest
import torch

from oracle import vn_dgcnn as OV
from oracle import weights as W
import cases as C

pytestmark = pytest.mark.gpu
GOLD = C.load_golden3()


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def assert_clouds_close(got, want, n_points):
    got, want = torch.as_tensor(got).detach().double().cpu().numpy(), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    e = np.abs(got - want).max(1) / np.abs(want).max()
    assert (e >= 2e-5).sum() <= max(1, len(e) // 4) and e.max() < 0.3 / n_points, e


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    return torch.device("cuda:0")


def _set_overlap(a, b):
    """fraction of (point, neighbour) pairs of a that are in b's list of the same point"""
    a, b = np.asarray(a), np.asarray(b)
    hit = (a[..., :, None] == b[..., None, :]).any(-1)
    return float(hit.mean())


@pytest.mark.parametrize("spec", C.PCD_ENC, ids=lambda s: s["name"])
def test_pcd_encoder_matches_reference_fixture(dev, spec):
    from diffassemble_amd.pcd_encoder import PcdEncoderEngine
    sd, pts = C.pcd_encoder_case(spec)
    eng = PcdEncoderEngine(sd, inv=spec["inv"], device=dev)
    out = eng.forward(pts.to(dev))
    want = GOLD[f"pcd_enc/{spec['name']}/out"]
    assert out.dtype == torch.float32
    assert_clouds_close(out, want, spec["N"])
    if not spec["inv"]:
        assert torch.equal(out[:, :384], out[:, 384:])          # cat(x, mean(x)) pooled over the points (vn_dgcnn.py:62-65)


@pytest.mark.parametrize("spec", [s for s in C.PCD_ENC if s["N"] <= 256], ids=lambda s: s["name"])
def test_knn_lists_match_reference(dev, spec):
    from diffassemble_amd.pcd_encoder import knn
    _, pts = C.pcd_encoder_case(spec)
    idx = knn(pts.to(dev)).cpu().numpy()
    want = GOLD[f"pcd_enc/{spec['name']}/idx1"]
    assert idx.shape == want.shape and idx.dtype == np.int32
    assert (idx[:, :, 0] == np.arange(spec["N"])[None, :]).all()           # a point is its own nearest neighbour
    assert _set_overlap(idx, want) > 0.999


def test_knn_feature_space_and_chunked_encoder_vs_oracle(dev):
    """63-dimensional neighbour search (stages 2 and 3), fragment chunking, and a cloud size that is not a multiple of
    the 256-point block nor of the 32-query slab."""
    from diffassemble_amd.pcd_encoder import PcdEncoderEngine, knn
    rng = np.random.default_rng(11)
    x = rng.standard_normal((3, 333, 63)).astype(np.float32)
    idx = knn(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert _set_overlap(idx, OV.knn(x)) > 0.999
    sd, pts = W.make_vn_dgcnn_state(128, 21), W.make_point_clouds(7, 333, 22)
    want, mid = OV.forward(sd, pts.numpy(), return_intermediates=True)
    whole = PcdEncoderEngine(sd, device=dev).forward(pts.to(dev))
    parts = PcdEncoderEngine(sd, device=dev, chunk=3).forward(pts.to(dev))
    assert_clouds_close(whole, want, 333)
    assert torch.equal(whole, parts)                                       # chunking does not change a bit


def test_knn_ties_and_duplicates(dev):
    """Exact grid (many equal distances) and duplicated points: lists stay valid (in range, no repeats, self or its
    duplicate first, distances non-decreasing)."""
    from diffassemble_amd.pcd_encoder import knn
    g = torch.stack(torch.meshgrid(torch.arange(5.), torch.arange(5.), torch.arange(4.), indexing="ij"), -1).reshape(1, 100, 3)
    g = torch.cat([g, g[:, :7]], 1)                                        # 7 duplicated points
    idx = knn(g.to(dev)).cpu()
    n = g.shape[1]
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    assert all(len(set(r.tolist())) == 20 for r in idx[0])
    d = (g[0][:, None, :] - g[0][idx[0].long()]).pow(2).sum(-1)
    assert (d[:, 0] == 0).all() and (d[:, 1:] >= d[:, :-1]).all()
    want = torch.cdist(g[0], g[0]).pow(2).sort(1)[0][:, :20]
    assert torch.allclose(d, want)


def test_nearest_sq_vs_oracle_and_part_accuracy(dev):
    from diffassemble_amd import metrics3d as M
    from diffassemble_amd.pcd_encoder import nearest_sq
    rng = np.random.default_rng(3)
    a = rng.standard_normal((4, 1000, 3)).astype(np.float32) * 0.3
    b = rng.standard_normal((4, 1333, 3)).astype(np.float32) * 0.3
    d_ab, d_ba = nearest_sq(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
    o_ab, o_ba = OV.chamfer_sq(a, b)
    assert np.allclose(d_ab.cpu().numpy(), o_ab, rtol=1e-5, atol=1e-7) and np.allclose(d_ba.cpu().numpy(), o_ba, rtol=1e-5, atol=1e-7)
    gold2 = C.load_golden2()
    for ms in C.METRICS3D:
        pcds, pred, gt = C.metrics3d_inputs(ms)
        acc = M.calc_part_acc(pcds.to(dev), pred[:, 4:].to(dev), gt[:, 4:].to(dev), pred[:, :4].to(dev), gt[:, :4].to(dev))
        assert abs(float(acc) - float(gold2[f"{ms['name']}/part_acc"])) < 1e-6


def test_3d_model_encodes_point_clouds(dev):
    """Eff_GAT_3d(backbone='vn_dgcnn').forward(xy_pos, time, pcd, ...) = forward_with_feats on pcd_features(pcd)
    (efficient_gat_3d.py:160-171): the HIP encoder feeds the HIP denoiser, from raw fragments."""
    from diffassemble_amd.model.backbones.efficient_gat_3d import Eff_GAT_3d
    torch.manual_seed(0)
    m = Eff_GAT_3d(steps=50, backbone="vn_dgcnn").to(dev).eval()
    sd = W.make_vn_dgcnn_state(128, 5)
    m.pcd_backbone.load_state_dict(sd)
    sizes = [5, 8]
    n = sum(sizes)
    pts = W.make_point_clouds(n, 200, 9).to(dev)
    feats = m.pcd_features(pts)
    assert tuple(feats.shape) == (n, 768)
    assert_clouds_close(feats, OV.forward(sd, pts.cpu().numpy()), 200)
    ei, batch = W.collate([W.dense_edge_index(k, True) for k in sizes], sizes)
    x = torch.randn(n, 7, device=dev)
    t = torch.randint(0, 50, (n,), device=dev)
    out_a, _ = m.forward(x, t, pts, ei.to(dev), batch.to(dev))
    out_b, _ = m.forward_with_feats(x, t, ei.to(dev), feats, batch.to(dev))
    assert torch.equal(out_a, out_b) and torch.isfinite(out_a).all()
    with pytest.raises(NotImplementedError):
        m.train()
        m.pcd_features(pts)


def test_3d_module_samples_from_raw_fragments(dev):
    """spatial_diffusion_3d_test_double_diffusion.GNN_Diffusion.p_sample_loop(shape, cond = fragments, ...) with no
    precomputed features (the reference's call, ...double_diffusion.py:689-700): the encoder runs once, the loop is the
    same as the one fed with pcd_features(cond), and validation_step scores the part accuracy through da_nearest_sq."""
    from types import SimpleNamespace
    from diffassemble_amd.model.spatial_diffusion_3d_test_double_diffusion import GNN_Diffusion, ModelMeanType
    torch.manual_seed(1)
    m = GNN_Diffusion(steps=20, sampling="DDIM", inference_ratio=2, model_mean_type=ModelMeanType.START_X,
                      backbone="vn_dgcnn").to(dev).eval()
    m.model.pcd_backbone.load_state_dict(W.make_vn_dgcnn_state(128, 2))
    sizes = [6, 9]
    P = sum(sizes)
    pts = W.make_point_clouds(P, 128, 4).to(dev)
    ei, batch = W.collate([W.dense_edge_index(k, True) for k in sizes], sizes)
    ei, batch = ei.to(dev), batch.to(dev)
    torch.manual_seed(5)
    a, _ = m.p_sample_loop((P, 7), pts, ei, batch)
    torch.manual_seed(5)
    b, _ = m.p_sample_loop((P, 7), None, ei, batch, pcd_feats=m.pcd_features(pts))
    assert len(a) == 10 and all(torch.equal(x, y) for x, y in zip(a, b)) and torch.isfinite(a[-1]).all()
    gt = torch.cat([torch.nn.functional.normalize(torch.randn(P, 4), dim=-1), 0.3 * torch.randn(P, 3)], 1).to(dev)
    bt = SimpleNamespace(x=gt, pcds=pts, edge_index=ei, batch=batch, category=["everyday", "everyday"])
    m.initialize_torchmetrics(["everyday"])
    final = m.validation_step(bt, 0)
    assert final.shape == (P, 7)
    acc = float(m.metrics["part_acc_everyday"].compute())
    assert 0.0 <= acc <= 1.0


def test_large_clouds_and_other_k(dev):
    """Clouds beyond the register-resident selection (N > 1024: the scores stay in LDS, da_knn's ordered path) through
    the whole encoder, and da_knn for k other than 20 (nearest first, against torch's sort of the same scores)."""
    from diffassemble_amd.pcd_encoder import PcdEncoderEngine, knn
    sd, pts = W.make_vn_dgcnn_state(128, 31), W.make_point_clouds(2, 1500, 32)
    assert_clouds_close(PcdEncoderEngine(sd, device=dev).forward(pts.to(dev)), OV.forward(sd, pts.numpy()), 1500)
    x = W.make_point_clouds(3, 700, 33)
    d = torch.cdist(x, x).pow(2)
    for k in (1, 5, 33, 64):
        idx = knn(x.to(dev), k=k).cpu().long()
        got = torch.gather(d, 2, idx)
        want = d.sort(2)[0][:, :, :k]
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-7), k
        assert (got[:, :, 1:] >= got[:, :, :-1] * (1 - 1e-4) - 1e-7).all()        # cdist's own rounding: not exactly monotone


def test_pcd_encoder_is_stream_capturable(dev):
    """da_pcd_encoder_forward inside a hipGraph (as its header promises): capture once, replay on new clouds written into
    the captured input buffer, same bits as the eager call."""
    from diffassemble_amd.pcd_encoder import PcdEncoderEngine
    sd = W.make_vn_dgcnn_state(128, 51)
    eng = PcdEncoderEngine(sd, device=dev)
    buf = W.make_point_clouds(6, 300, 52).to(dev)
    out = torch.empty(6, 768, device=dev)
    eng.forward(buf, out)                                   # warm-up: workspace allocation, one-time attributes
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            eng.forward(buf, out)
    new = W.make_point_clouds(6, 300, 53).to(dev)
    buf.copy_(new)
    g.replay()
    torch.cuda.synchronize()
    got = out.clone()
    assert torch.equal(got, eng.forward(new.clone()))


def test_smallest_cloud_and_single_fragment(dev):
    """N = 20 (every point neighbours every point: the encoder's lower limit), one fragment; nearest_sq on 1-point clouds."""
    from diffassemble_amd import _lib
    from diffassemble_amd.pcd_encoder import PcdEncoderEngine, knn, nearest_sq
    sd, pts = W.make_vn_dgcnn_state(128, 61), W.make_point_clouds(1, 20, 62)
    eng = PcdEncoderEngine(sd, device=dev)
    assert_clouds_close(eng.forward(pts.to(dev)), OV.forward(sd, pts.numpy()), 20)
    idx = knn(pts.to(dev)).cpu()
    assert all(sorted(r.tolist()) == list(range(20)) for r in idx[0])
    with pytest.raises(_lib.DaError):
        eng.forward(W.make_point_clouds(1, 19, 63).to(dev))             # fewer points than neighbours: loud, like topk
    a, b = torch.tensor([[[0.0, 0.0, 0.0]]]), torch.tensor([[[1.0, 2.0, 2.0], [0.5, 0.0, 0.0]]])
    d_ab, d_ba = nearest_sq(a.to(dev), b.to(dev))
    assert torch.allclose(d_ab.cpu(), torch.tensor([[0.25]])) and torch.allclose(d_ba.cpu(), torch.tensor([[9.0, 0.25]]))
