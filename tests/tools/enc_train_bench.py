"""Timing of the encoder's training forward + backward for n pieces (fp32 or bf16 storage); for rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from oracle import weights as W
from diffassemble_amd.encoder_train import EncoderTrainEngine
from diffassemble_amd.model.backbones.resnet_equivariant import ResNet18
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1152
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
net = ResNet18(precision=prec); net.load_state_dict(W.make_encoder_state(0)); net = net.to(dev).train()
eng = EncoderTrainEngine(net, dev, precision=prec)
x = torch.rand(n, 3, 32, 32, device=dev); G = torch.randn(n, 1088, device=dev)
for _ in range(2):
    eng.forward(x); eng.backward(G)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(reps):
    eng.forward(x)
torch.cuda.synchronize(); t1 = time.time()
for _ in range(reps):
    eng.backward(G)
torch.cuda.synchronize(); t2 = time.time()
print(f"{prec} n={n}: forward {(t1 - t0) / reps * 1e3:.1f} ms, backward {(t2 - t1) / reps * 1e3:.1f} ms, {n * reps / (t2 - t0):.0f} pieces/s (fwd+bwd), mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
