import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fourierdiffusion_amd.utils.fourier import dft, idft
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
for (B, T, C) in [(4096, 365, 8), (4096, 365, 8), (4096, 219, 8), (4096, 146, 8), (4096, 73, 8), (4096, 185, 8)]:
    x = torch.randn(B, T, C, device="cuda")
    y = dft(x)
    td = timed(lambda: dft(x)); ti = timed(lambda: idft(y)); ti2 = timed(lambda: idft(x))
    print(f"(B={B},T={T},C={C}): dft {td*1e6:7.1f} us | idft(spectrum) {ti*1e6:7.1f} us | idft(randn) {ti2*1e6:7.1f} us  roundtrip err {(idft(y)-x).abs().max().item():.2e}")
