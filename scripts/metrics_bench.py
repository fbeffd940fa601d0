"""Time the Wasserstein evaluation metrics at the reference's ECG scale (87 554 x 187 training set, 10 000 generated samples,
1000 directions -- cmd/conf/sample.yaml + metrics/default.yaml).  usage: python scripts/metrics_bench.py [n_train n_gen d K]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n, m, d, K = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (87554, 10000, 187, 1000)
    from fourierdiffusion_amd.utils.wasserstein import WassersteinDistances, project_rows, sort_rows, w2_sorted_rows
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.randn(n, d, device="cuda", generator=g)
    Y = torch.randn(m, d, device="cuda", generator=g) * 1.1
    wd = WassersteinDistances(X, Y, seed=42)
    for name, fn in (("sliced", lambda: wd.sliced_distances(K)), ("marginal", wd.marginal_distances)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        print(f"{name}: {len(out)} distances between {n} and {m} samples of dim {d}: {(time.perf_counter() - t0) * 1e3:.1f} ms "
              f"(mean {out.mean():.4f})")
    dirs = np.stack(WassersteinDistances(X, Y, seed=42).get_random_directions(K))
    stages = {}
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        P = project_rows(X, dirs); torch.cuda.synchronize(); stages["project (K,n)"] = time.perf_counter() - t; t = time.perf_counter()
        S = sort_rows(P); torch.cuda.synchronize(); stages["sort rows"] = time.perf_counter() - t; t = time.perf_counter()
        Q = sort_rows(project_rows(Y, dirs)); torch.cuda.synchronize(); t = time.perf_counter()
        w2_sorted_rows(S, Q); torch.cuda.synchronize(); stages["w2 rows"] = time.perf_counter() - t
    print({k: f"{v * 1e3:.2f} ms" for k, v in stages.items()}, f"sort: {K * n * 8 / stages['sort rows'] / 1e9:.0f} GB/s (keys in + out)")


if __name__ == "__main__":
    main()
