"""Training path of the 2D piece encoder (the encoder runs in every training step,
/root/reference/puzzle_diff/model/spatial_diffusion.py:450): the network walk of the reference's P4 ResNet-18 in
train() mode -- BatchNorm3d on batch statistics -- and its backward, driven layer by layer through the fp32 primitives of
libdiffassemble_hip.so (``da_enc_*``, ``da_gemm_tn_f32``, ``da_linear``; include/diffassemble_hip.h).  Host logic only: unit
order, the activation / gradient buffer pool, weight packing; every per-pixel operation is a library call.

Replaces, under /root/reference/puzzle_diff/model/backbones/: ``ResNet.forward`` / ``BasicBlock.forward``
(resnet_equivariant.py:14-38,93-112) under ``model.train()`` and torch autograd through them, ``SplitGConv2D``
(groupy/gconv/pytorch_gconv/splitgconv2d.py:15-22,70-92), and ``Eff_GAT.visual_features`` (efficient_gat.py:149-189).

A "unit" is conv -> BatchNorm [-> + residual] [-> ReLU]; Y is the convolution output, Z the unit output.  Maps are
zero-haloed NHWC [n][H+2][H+2][planes*4]; the halos are zeroed at allocation and never written.
Backward rules used (see da_encoder_train.hip): dgrad = the forward convolution kernel on dY with flipped / transposed
weights, its ``res`` input accumulating the other gradient paths; stride-2 units are differentiated as stride-1 units on a
zero-stuffed dY; wgrad = one TN GEMM per filter tap over the haloed maps as they lie in memory; filter bank -> parameter
gradient = a 4-way gather-sum.
"""
import torch

from . import _lib
from .encoder import PLANES, _halo_linear, conv_keys, p4_filter_bank

GEMM_SCRATCH_FLOATS = 16 << 20          # split-row partials of da_gemm_tn_f32 (64 MB)
MOMENTUM = 0.1                          # nn.BatchNorm3d default


def _units():
    """[(conv key, bn key, in planes, out planes, k, stride, input H)] in forward order; unit 0 is the stem."""
    units = [("conv1", "bn1", None, 32, 3, 1, 32)]
    cin, H = 32, 32
    for li, planes in enumerate(PLANES, start=1):
        for bi in range(2):
            p = f"layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            units.append((p + "conv1", p + "bn1", cin, planes, 3, stride, H))
            units.append((p + "conv2", p + "bn2", planes, planes, 3, 1, H // stride))
            if stride == 2:
                units.append((p + "shortcut.0", p + "shortcut.1", cin, planes, 1, 2, H))
            cin, H = planes, H // stride
    return units


class EncoderTrainEngine:
    """Training-mode forward / backward of ``model.backbones.resnet_equivariant.ResNet`` (a parameter holder).
    ``forward(patches)`` -> patch_feats [n, 1088] fp32 and updates the BatchNorm running statistics like torch does;
    ``backward(d_feats)`` ADDS the parameter gradients into ``param.grad`` (allocated on first use)."""

    def __init__(self, module, device=None, precision="fp32"):
        """``precision``: "fp32" (the parity mode) or "bf16" -- activations and activation gradients stored in bf16, the
        convolutions on the bf16 matrix cores, BatchNorm arithmetic, master weights and every parameter gradient in fp32."""
        assert precision in ("fp32", "bf16"), precision
        self.precision = precision
        self.prec = _lib.PREC_BF16 if precision == "bf16" else _lib.PREC_F32
        self.act = torch.bfloat16 if precision == "bf16" else torch.float32
        self.module = module
        self.params = dict(module.named_parameters())
        self.buffers = dict(module.named_buffers())
        self.device = torch.device(device) if device is not None else self.params["conv1.weight"].device
        if self.device.type != "cuda":
            raise _lib.DaError("EncoderTrainEngine needs a ROCm device (no CPU path in diffassemble_amd)")
        self.lib = _lib.lib()
        self.units = _units()
        assert [u[0] for u in self.units[1:]] == [c for c, _ in conv_keys()]
        dev = self.device
        self.zero_bias = torch.zeros(512, dtype=torch.float32, device=dev)
        self.gemm_scratch = torch.empty(GEMM_SCRATCH_FLOATS, dtype=torch.float32, device=dev)
        self._tables = {}
        for conv, *_ in self.units:
            w = self.params[conv + ".weight"]
            bank = p4_filter_bank(torch.arange(w.numel(), device=dev).view(w.shape))
            src = (bank if conv == "conv1" else self._pack_fwd(bank)).reshape(-1)       # the stem's bank stays [128, c*9 + tap]
            self._tables[conv] = torch.argsort(src, stable=True).to(torch.int32).view(w.numel(), 4).contiguous()
        self._pack_key = None
        self._n = None

    # ------------------------------------------------------------------ packing (once per optimizer step)
    @staticmethod
    def _pack_fwd(bank):                      # [O4, I4, k, k] -> [O4, k*k*I4]  (tap-major, channel-minor)
        return bank.permute(0, 2, 3, 1).reshape(bank.shape[0], -1).contiguous()

    @staticmethod
    def _pack_dgrad(bank):                    # -> [I4, k*k*O4] with the taps flipped: conv(dY, .) is the input gradient
        return bank.flip(2, 3).permute(1, 2, 3, 0).reshape(bank.shape[1], -1).contiguous()

    def _pack(self):
        key = tuple((p.data_ptr(), p._version) for p in self.params.values())
        if key == self._pack_key:
            return
        with torch.no_grad():
            self.wf, self.wt = {}, {}
            for conv, _, cin, *_ in self.units:
                bank = p4_filter_bank(self.params[conv + ".weight"].detach().float())
                self.wf[conv] = bank.reshape(128, 27).contiguous() if cin is None else self._pack_fwd(bank).to(self.act)
                if cin is not None:
                    self.wt[conv] = self._pack_dgrad(bank).to(self.act)
            self.lin = {}
            for name, C, H in (("linear1", 256, 8), ("linear2", 512, 4)):
                wh = _halo_linear(self.params[name + ".weight"].detach().float(), C, H).to(self.act).contiguous()
                # transposed copy for dA = dF W; its K (= 544) is zero-padded to 576 so that the bf16 matrix kernel's
                # 64-element K steps divide it
                wht = torch.zeros(wh.shape[1], 576, dtype=self.act, device=wh.device)
                wht[:, :544] = wh.t()
                self.lin[name] = (wh, wht, self.params[name + ".bias"].detach().float().contiguous())
        self._pack_key = key

    # ------------------------------------------------------------------ buffers
    def _alloc(self, n):
        if self._n == n:
            return
        dev = self.device
        z = lambda *s: torch.zeros(*s, dtype=self.act, device=self.device)  # noqa: E731
        zf = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)  # noqa: E731
        self.Y, self.Z, self.mean, self.var = [], [], [], []
        for _, _, _, planes, _, stride, H in self.units:
            Ho = H // stride
            self.Y.append(z(n, Ho + 2, Ho + 2, planes * 4))
            self.Z.append(z(n, Ho + 2, Ho + 2, planes * 4))
            self.mean.append(zf(planes))
            self.var.append(zf(planes))
        # gradient pool: five maps per resolution (dOut, dY, gRes, dZ of the inner unit, spare) + zero-stuffed maps
        self.pool = {}
        for H, C4 in ((32, 128), (16, 256), (8, 256), (4, 512)):
            self.pool[H] = [z(n, H + 2, H + 2, C4) for _ in range(4)]
        self.up = {H: z(n, H + 2, H + 2, C4) for H, C4 in ((32, 256), (16, 256), (8, 512))}     # keyed by INPUT resolution
        self.cols = z(n, 34, 34, 32)
        self.bn_scratch = torch.empty(self.lib.da_enc_train_scratch_bytes(n), dtype=torch.uint8, device=dev)
        self.colsum_scratch = torch.empty(((n + 127) // 128) * 544 + 64, dtype=torch.float32, device=dev)
        self.feats = torch.empty(n, 1088, dtype=self.act, device=dev)
        self._n = n

    # ------------------------------------------------------------------ primitives
    def _st(self):
        return _lib.stream_ptr(self.device)

    def _conv(self, X, cin4, Hi, W, Y, cout4, k, stride, res=None):
        """Y = conv(X, W) [+ res]; pieces are independent, so the batch goes through in slices whose maps stay below the
        kernel's 4 GB (32-bit offset) limit."""
        per = X.element_size() * max(X[0].numel(), Y[0].numel())
        step = max(1, ((1 << 32) - 1) // per)
        for i in range(0, X.shape[0], step):
            j = min(X.shape[0], i + step)
            _lib.check(self.lib.da_enc_conv(self.prec, j - i, _lib.ptr(X[i:j]), cin4, Hi, _lib.ptr(W), _lib.ptr(self.zero_bias),
                                            _lib.ptr(None if res is None else res[i:j]), _lib.ptr(Y[i:j]), cout4, k, stride, 0,
                                            self._st()))

    def _bn_forward(self, u, res, relu):
        _, bn, _, planes, _, stride, H = self.units[u]
        Ho, n = H // stride, self._n
        _lib.check(self.lib.da_enc_bn_stats(self.prec, n, Ho, planes * 4, _lib.ptr(self.Y[u]), _lib.ptr(self.mean[u]), _lib.ptr(self.var[u]),
                                            _lib.ptr(self.bn_scratch), self._st()))
        _lib.check(self.lib.da_enc_bn_apply(self.prec, n, Ho, planes * 4, _lib.ptr(self.Y[u]), _lib.ptr(self.mean[u]), _lib.ptr(self.var[u]),
                                            _lib.ptr(self.params[bn + ".weight"]), _lib.ptr(self.params[bn + ".bias"]),
                                            _lib.ptr(res), int(relu), _lib.ptr(self.Z[u]), self._st()))
        with torch.no_grad():                                     # running statistics, as torch updates them
            cnt = n * Ho * Ho * 4
            self.buffers[bn + ".running_mean"].mul_(1 - MOMENTUM).add_(self.mean[u], alpha=MOMENTUM)
            self.buffers[bn + ".running_var"].mul_(1 - MOMENTUM).add_(self.var[u], alpha=MOMENTUM * cnt / max(cnt - 1, 1))
            self.buffers[bn + ".num_batches_tracked"].add_(1)

    def _grad(self, name):
        """``param.grad`` as a view of ONE flat fp32 buffer (``flat_grad``, like ``TrainEngine.flat_grad``): the data-parallel
        exchange is then a single all-reduce of that buffer, without re-flattening 11 M values every step.  A gradient the
        caller dropped (``zero_grad(set_to_none=True)``) or replaced is re-attached, zeroed / carried over."""
        if getattr(self, "flat_grad", None) is None:
            offs, off = {}, 0
            for n_, q in self.params.items():
                offs[n_] = off
                off += (q.numel() + 63) // 64 * 64
            self.flat_grad = torch.zeros(off, dtype=torch.float32, device=self.device)
            self._grad_views = {n_: self.flat_grad[o:o + self.params[n_].numel()].view(self.params[n_].shape) for n_, o in offs.items()}
        p, v = self.params[name], self._grad_views[name]
        if p.grad is None:
            v.zero_()
            p.grad = v
        elif p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v
        return v

    def grads_attached(self):
        """True when every parameter's ``.grad`` is its view of ``flat_grad`` (so the flat buffer IS the gradient)."""
        views = getattr(self, "_grad_views", None)
        return views is not None and all(p.grad is not None and p.grad.data_ptr() == views[n_].data_ptr() for n_, p in self.params.items())

    def _bn_backward(self, u, dZ, relu, dY, dRes=None):
        _, bn, _, planes, _, stride, H = self.units[u]
        _lib.check(self.lib.da_enc_bn_backward(self.prec, self._n, H // stride, planes * 4, _lib.ptr(dZ), _lib.ptr(self.Z[u]), _lib.ptr(self.Y[u]),
                                               _lib.ptr(self.mean[u]), _lib.ptr(self.var[u]), _lib.ptr(self.params[bn + ".weight"]),
                                               int(relu), _lib.ptr(self._grad(bn + ".weight")), _lib.ptr(self._grad(bn + ".bias")),
                                               _lib.ptr(dY), _lib.ptr(dRes), _lib.ptr(self.bn_scratch), self._st()))

    def _gemm_tn(self, M, N, K, A, lda, B, ldb, C, ldc):
        fn = self.lib.da_gemm_tn_bf16 if self.precision == "bf16" else self.lib.da_gemm_tn_f32
        _lib.check(fn(M, N, K, _lib.ptr(A), lda, _lib.ptr(B), ldb, _lib.ptr(C), ldc, _lib.ptr(self.gemm_scratch), self._st()))

    def _wgrad(self, conv, dY, X, cin4, cout4, k, H):
        """dBank[o][tap][c] += sum_q dY[q][o] X[q + offset(tap)][c] over the haloed positions (dY's halo is zero), then the
        4-way gather-sum into the parameter's gradient.  dY and X have the same spatial size H (stride-1 form)."""
        Wp = H + 2
        rows = self._n * Wp * Wp - 2 * (Wp + 1)
        dbank = torch.zeros(cout4, k * k * cin4, dtype=torch.float32, device=self.device)
        a = dY.view(-1)[(Wp + 1) * cout4:]
        for tap in range(k * k):
            ky, kx = (tap // 3, tap % 3) if k == 3 else (1, 1)
            off = (Wp + 1) + (ky - 1) * Wp + (kx - 1)
            b = X.view(-1)[off * cin4:]
            self._gemm_tn(rows, cout4, cin4, a, cout4, b, cin4, dbank.view(-1)[tap * cin4:], k * k * cin4)
        w = self.params[conv + ".weight"]
        _lib.check(self.lib.da_enc_bank_grad(w.numel(), _lib.ptr(self._tables[conv]), _lib.ptr(dbank), _lib.ptr(self._grad(conv + ".weight")),
                                             self._st()))

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, patches):
        if patches.device.type != "cuda":
            raise _lib.DaError("EncoderTrainEngine.forward: patches must live on the ROCm device")
        x = patches.detach().to(torch.float32).contiguous()
        n = x.shape[0]
        self._alloc(n)
        self._pack()
        self.x = x
        _lib.check(self.lib.da_enc_stem(self.prec, n, _lib.ptr(x), _lib.ptr(self.wf["conv1"]), _lib.ptr(self.zero_bias), _lib.ptr(self.Y[0]), 0,
                                        self._st()))
        self._bn_forward(0, None, True)
        cur, u = 0, 1                                    # cur: unit whose Z is the running activation
        self.block_io = []                               # (input unit, unit a, unit b, shortcut unit or None)
        while u < len(self.units):
            conv, _, cin, planes, _, stride, H = self.units[u]
            a, b = u, u + 1
            s = u + 2 if stride == 2 else None
            self._conv(self.Z[cur], cin * 4, H, self.wf[conv], self.Y[a], planes * 4, 3, stride)
            self._bn_forward(a, None, True)
            self._conv(self.Z[a], planes * 4, H // stride, self.wf[self.units[b][0]], self.Y[b], planes * 4, 3, 1)
            if s is not None:
                self._conv(self.Z[cur], cin * 4, H, self.wf[self.units[s][0]], self.Y[s], planes * 4, 1, 2)
                self._bn_forward(s, None, False)
                res = self.Z[s]
            else:
                res = self.Z[cur]
            self._bn_forward(b, res, True)
            self.block_io.append((cur, a, b, s))
            cur, u = b, u + (3 if s is not None else 2)
        self.out3, self.out4 = self.block_io[5][2], self.block_io[7][2]
        for name, src, col in (("linear1", self.out3, 0), ("linear2", self.out4, 544)):
            wh, _, bias = self.lin[name]
            A = self.Z[src].view(n, -1)
            _lib.check(self.lib.da_linear(self.prec, n, A.shape[1], 544, _lib.ptr(A), A.shape[1], _lib.ptr(wh), _lib.ptr(bias),
                                          _lib.ACT_NONE, None, _lib.ptr(self.feats[:, col:]), 1088, self._st()))
        return self.feats.float() if self.precision == "bf16" else self.feats

    # ------------------------------------------------------------------ backward
    @torch.no_grad()
    def backward(self, d_feats):
        n = self._n
        d = d_feats.detach().to(torch.float32).contiguous()
        assert d.shape == (n, 1088)
        da = torch.zeros(n, 2 * 576, dtype=self.act, device=self.device)   # operand copy of dF, each head padded to 576 columns
        da[:, :544] = d[:, :544]
        da[:, 576:576 + 544] = d[:, 544:]
        st = self._st
        # linear heads: dA = dF W (halo columns of the packed weight are zero -> the halo of dA is zero), dW += dF^T A
        heads = {}
        for name, src, col, C, H in (("linear1", self.out3, 0, 256, 8), ("linear2", self.out4, 544, 512, 4)):
            wh, wht, _ = self.lin[name]
            A = self.Z[src].view(n, -1)
            dA = self.pool[H][0]
            dF, dFa = d[:, col:col + 544], da[:, (576 if col else 0):]
            _lib.check(self.lib.da_linear(self.prec, n, 576, A.shape[1], _lib.ptr(dFa), 1152, _lib.ptr(wht),
                                          _lib.ptr(torch.zeros(A.shape[1], dtype=torch.float32, device=self.device)),
                                          _lib.ACT_NONE, None, _lib.ptr(dA), A.shape[1], st()))
            dwh = torch.zeros(wh.shape, dtype=torch.float32, device=self.device)
            self._gemm_tn(n, 544, A.shape[1], dFa, 1152, A, A.shape[1], dwh, A.shape[1])
            # back from the haloed NHWC columns to the reference's NCHW flatten (inverse of encoder._halo_linear)
            self._grad(name + ".weight").add_(dwh.view(544, H + 2, H + 2, C)[:, 1:H + 1, 1:H + 1, :].permute(0, 3, 1, 2).reshape(544, -1))
            _lib.check(self.lib.da_colsum_f32(n, 544, _lib.ptr(dF), 1088, _lib.ptr(self._grad(name + ".bias")), _lib.ptr(self.colsum_scratch), st()))
            heads[src] = dA
        # blocks in reverse; pool[H] = [dOut, dY, gRes, dZ_a]
        for cur, a, b, s in reversed(self.block_io):
            conv_a, _, cin, planes, _, stride, Hin = self.units[a]
            conv_b = self.units[b][0]
            H = Hin // stride
            dOut, dY, gRes, dZa = self.pool[H]
            C4 = planes * 4
            self._bn_backward(b, dOut, True, dY, gRes)
            self._wgrad(conv_b, dY, self.Z[a], C4, C4, 3, H)
            self._conv(dY, C4, H, self.wt[conv_b], dZa, C4, 3, 1)
            self._bn_backward(a, dZa, True, dY)
            if s is None:
                self._wgrad(conv_a, dY, self.Z[cur], cin * 4, C4, 3, H)
                # input gradient = dgrad(conv a) + the identity residual's share; lands in dOut = the next block's dOut
                self._conv(dY, C4, H, self.wt[conv_a], dOut, cin * 4, 3, 1, res=gRes)
            else:
                up, dIn = self.up[Hin], self.pool[Hin][0]
                prior = dIn if cur == self.out3 else None            # layer3's output also feeds linear1: accumulate
                _lib.check(self.lib.da_enc_upsample2(self.prec, n, H, C4, _lib.ptr(dY), _lib.ptr(up), st()))
                self._wgrad(conv_a, up, self.Z[cur], cin * 4, C4, 3, Hin)
                self._conv(up, C4, Hin, self.wt[conv_a], dIn, cin * 4, 3, 1, res=prior)
                self._bn_backward(s, gRes, False, dY)
                _lib.check(self.lib.da_enc_upsample2(self.prec, n, H, C4, _lib.ptr(dY), _lib.ptr(up), st()))
                self._wgrad(self.units[s][0], up, self.Z[cur], cin * 4, C4, 1, Hin)
                self._conv(up, C4, Hin, self.wt[self.units[s][0]], dIn, cin * 4, 1, 1, res=dIn)
        # stem: BatchNorm backward, then the weight gradient as one TN GEMM against the im2col of the normalised crops
        dOut, dY = self.pool[32][0], self.pool[32][1]
        self._bn_backward(0, dOut, True, dY)
        _lib.check(self.lib.da_enc_stem_im2col(self.prec, n, _lib.ptr(self.x), _lib.ptr(self.cols), st()))
        dbank = torch.zeros(128, 27, dtype=torch.float32, device=self.device)
        self._gemm_tn(n * 34 * 34, 128, 27, dY, 128, self.cols, 32, dbank, 27)
        w = self.params["conv1.weight"]
        _lib.check(self.lib.da_enc_bank_grad(w.numel(), _lib.ptr(self._tables["conv1"]), _lib.ptr(dbank), _lib.ptr(self._grad("conv1.weight")), st()))
