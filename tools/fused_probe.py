"""Ablation timings of k_conv_fused (needs a -DDA_FUSED_PROBE build: DA_HIPCC_EXTRA=-DDA_FUSED_PROBE python __graft_entry__.py).
DA_FUSED_DEBUG bits: 1 no phase 2, 2 no softmax, 4 no PV, 8 no QK, 16 no x loads in phase 2, 32 one key block, 64 no phase 1."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for dbg in (0, 1, 64, 65, 2, 4, 8, 16, 32, 2 | 4 | 8, 6, 10, 12):
    env = dict(os.environ, DA_FUSED_DEBUG=str(dbg), CS="32", ITERS="20")
    r = subprocess.run([sys.executable, os.path.join(here, "attn_probe.py")], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("C=")]
    print(f"debug={dbg:3d}: {line[0] if line else r.stderr[-300:]}", flush=True)
