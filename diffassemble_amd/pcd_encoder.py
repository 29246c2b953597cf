"""Host side of the 3D piece encoder: packs the reference's ``VN_DGCNN`` state dict
(/root/reference/puzzle_diff/model/backbones/vnn/vn_dgcnn.py:7-31) into the fp32 blobs ``da_pcd_encoder_forward`` reads
(include/diffassemble_hip.h) and owns the workspace.  Also the two point-cloud primitives around it: ``knn`` (the
encoder's neighbour search, vn_dgcnn.py:114-120) and ``nearest_sq`` (the K = 1 search behind the part-accuracy metric,
chamfer_distance.py:148-149).  No torch arithmetic on the data path, no CPU fallback."""
import torch

from . import _lib

STAGES = (("conv1", "conv2", 1), ("conv3", "conv4", 21), ("conv5", None, 21))
BN_EPS = 1e-5


def _bn_affine(sd, name):
    """eval BatchNorm of the vector norm as  norm * scale + shift  (vn_layers.py:145-154)."""
    g, b = sd[f"{name}.batchnorm.bn.weight"].double(), sd[f"{name}.batchnorm.bn.bias"].double()
    mu, var = sd[f"{name}.batchnorm.bn.running_mean"].double(), sd[f"{name}.batchnorm.bn.running_var"].double()
    scale = g / torch.sqrt(var + BN_EPS)
    return torch.stack([scale, b - mu * scale]).float()


class PcdEncoderEngine:
    """Packed VN-DGCNN.  ``sd``: state dict with the reference's ``VN_DGCNN(feat_dim)`` keys (``pcd_backbone.*`` of an
    ``Eff_GAT_3d`` checkpoint with the prefix stripped).  ``inv`` selects the ``vn_dgcnn_inv`` output (linear0 path)."""

    def __init__(self, sd, *, inv=False, device=None, chunk=None):
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _lib.DaError("PcdEncoderEngine needs a ROCm device (no CPU path in diffassemble_amd)")
        self.lib = _lib.lib()
        self.inv, self.chunk = bool(inv), chunk
        sd = {k: v.detach().to(self.device, torch.float32) for k, v in sd.items() if v.is_floating_point()}
        self._keep = []
        w = _lib.DaPcdEncoderWeights()

        def keep(*ts):
            t = torch.cat([x.reshape(-1) for x in ts]).to(torch.float32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        for s, (a, b, cin) in enumerate(STAGES):
            wf, wd = sd[f"{a}.map_to_feat.weight"], sd[f"{a}.map_to_dir.weight"]          # [21, 2 cin]
            assert wf.shape == (21, 2 * cin) and wd.shape == (21, 2 * cin), (a, tuple(wf.shape), tuple(wd.shape))
            w.premap[s] = keep(wf[:, :cin], wd[:, :cin], wf[:, cin:] - wf[:, :cin], wd[:, cin:] - wd[:, :cin])
            w.bn_a[s] = keep(_bn_affine(sd, a))
            if b is not None:
                # rows zero-padded to 22: a (c, c + 1) weight pair is one 64-bit scalar operand of a packed fp32 FMA
                pad = torch.nn.functional.pad
                w.conv_b[s] = keep(pad(sd[f"{b}.map_to_feat.weight"], (0, 1)), pad(sd[f"{b}.map_to_dir.weight"], (0, 1)),
                                   _bn_affine(sd, b))
        w6, d6 = sd["conv6.map_to_feat.weight"], sd["conv6.map_to_dir.weight"]
        self.feat_dim = int(w6.shape[0])
        assert w6.shape[1] == 63 and tuple(d6.shape) == (1, 63), (tuple(w6.shape), tuple(d6.shape))
        w.feat_dim = self.feat_dim
        w.conv6 = keep(w6, d6, _bn_affine(sd, "conv6"))
        if "linear0.weight" in sd:
            assert tuple(sd["linear0.weight"].shape) == (2 * self.feat_dim, 3)
            w.linear0 = keep(sd["linear0.weight"], sd["linear0.bias"])
        self.w = w
        self.out_dim = 2 * self.feat_dim if self.inv else 6 * self.feat_dim
        self._ws, self._ws_key = None, None

    def _chunk_for(self, n_parts, n_points):
        if self.chunk:
            return int(self.chunk)
        # ~1.9 KB of workspace per point: keep a chunk under ~2 GB
        return max(1, min(n_parts, (1 << 20) // max(n_points, 1)))

    def forward(self, points, out=None):
        """points [P, N, 3] fp32 (device), N >= 20 -> [P, 6 feat_dim] (or [P, 2 feat_dim] for inv) fp32."""
        if points.device.type != "cuda":
            raise _lib.DaError("PcdEncoderEngine.forward: the point clouds must live on the ROCm device")
        assert points.dim() == 3 and points.shape[2] == 3, tuple(points.shape)
        x = points.detach().to(torch.float32).contiguous()
        P, N = int(x.shape[0]), int(x.shape[1])
        if out is None:
            out = torch.empty(P, self.out_dim, dtype=torch.float32, device=self.device)
        if P == 0:
            return out
        chunk = self._chunk_for(P, N)
        key = (N, chunk)
        if self._ws_key != key:
            self._ws = torch.empty(self.lib.da_pcd_encoder_workspace_bytes(N, chunk, self.feat_dim), dtype=torch.uint8,
                                   device=self.device)
            self._ws_key = key
        assert out.dtype == torch.float32 and out.shape == (P, self.out_dim) and out.stride(1) == 1
        _lib.check(self.lib.da_pcd_encoder_forward(self.w, P, N, _lib.ptr(x), int(self.inv), _lib.ptr(out), out.stride(0),
                                                   _lib.ptr(self._ws), self._ws.numel(), chunk,
                                                   _lib.stream_ptr(self.device)))
        return out


def knn(x, k=_lib.PCD_K):
    """x [B, N, F] fp32 on the device, F == 3 or F <= 64 -> idx [B, N, k] int32: the k nearest points of the same cloud
    (self included), nearest first (vn_dgcnn.py:114-120; ties towards the lower index)."""
    if x.device.type != "cuda":
        raise _lib.DaError("knn: the clouds must live on the ROCm device")
    B, N, F = x.shape
    x = x.detach().to(torch.float32)
    if F == 3:
        x, ldx = x.contiguous(), 3
    else:
        assert F <= 64, F
        pad = torch.zeros(B, N, 64, dtype=torch.float32, device=x.device)
        pad[:, :, :F] = x
        x, ldx = pad, 64
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().da_knn(B, N, F, _lib.ptr(x), ldx, k, _lib.ptr(idx), _lib.stream_ptr(x.device)))
    return idx


def nearest_sq(a, b):
    """a [P, N, 3], b [P, M, 3] fp32 on the device -> (d_ab [P, N], d_ba [P, M]): squared distance from every point to the
    nearest point of the other cloud (pytorch3d ``knn_points(K=1)`` both ways, chamfer_distance.py:148-149)."""
    if a.device.type != "cuda" or b.device.type != "cuda":
        raise _lib.DaError("nearest_sq: the clouds must live on the ROCm device")
    a, b = a.detach().to(torch.float32).contiguous(), b.detach().to(torch.float32).contiguous()
    P, N, _ = a.shape
    M = b.shape[1]
    assert b.shape[0] == P and a.shape[2] == 3 and b.shape[2] == 3
    d_ab = torch.empty(P, N, dtype=torch.float32, device=a.device)
    d_ba = torch.empty(P, M, dtype=torch.float32, device=a.device)
    if P:
        _lib.check(_lib.lib().da_nearest_sq(P, N, M, _lib.ptr(a), _lib.ptr(b), _lib.ptr(d_ab), _lib.ptr(d_ba),
                                            _lib.stream_ptr(a.device)))
    return d_ab, d_ba
