import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import merv_oracle as O
cfg = O.merv_full_cfgs()[3]; cfg.layers = 2
W = O.random_encoder_weights(cfg, 0)
pix = torch.randn(1, 16, 3, 224, 224)
print("cores", os.cpu_count())
for n in (8, 16, 32, 64, 128, 256):
    if n > (os.cpu_count() or 1): break
    torch.set_num_threads(n)
    with torch.no_grad():
        O.encoder_forward(pix, cfg, W)
        t = time.perf_counter(); O.encoder_forward(pix, cfg, W); dt = time.perf_counter() - t
    print(n, "threads:", round(dt, 3), "s for siglip 2 blocks")
