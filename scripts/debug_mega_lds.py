"""Debug aid: dump workgroup 0's LDS after layer 0's attention and compare K, V^T, attention output with numpy."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FDIFF_MEGA_DUMP"] = "/tmp/lds.bin"
os.environ["FDIFF_MEGA_LAYERS"] = "1"
from oracle import fdiff_oracle as O, weights as W
from oracle.make_golden import CFG_DEFAULT
from tests.gpu_util import make_model, dev, host
from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch

cfg = CFG_DEFAULT
B = 4
T, C, D, H = cfg["T"], cfg["C"], cfg["D"], cfg["H"]
hd = D // H
m, _, sd = make_model(cfg, precision="bf16")
X = W.randn("dbg_x", (B, T, C), 2)
t = W.uniform("dbg_t", (B,), 2, 1e-5, 1.0)
m.eval()
out = host(m(DiffusableBatch(X=dev(X), timesteps=dev(t))))
raw = np.fromfile("/tmp/lds.bin", dtype=np.uint8)
print("lds bytes", raw.size)

def bf16(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)

KT = (T + 15) // 16; S = 1; NTILE = S * KT; NTOK = NTILE * 16; KS1 = 3; KSX = 3; NP = 6; NJ = (KT + 1) // 2
off_xfr = 0
off_wsl = NTILE * KSX * 1024
off_kbf = off_wsl + 3 * NP * KS1 * 1024
off_vbf = off_kbf + NP * NTOK * 32
_, hidden = O.score_forward(sd, X, t, H, return_hidden=True)
h0 = hidden[0][0]                                  # (T, D) layer-0 input of series 0
Win = sd["backbone.layers.0.self_attn.in_proj_weight"].astype(np.float64)
bin_ = sd["backbone.layers.0.self_attn.in_proj_bias"].astype(np.float64)
qkv = h0 @ Win.T + bin_
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
kb = bf16(raw[off_kbf:off_kbf + NP * NTOK * 32].view(np.uint16)).reshape(NP, NTOK, 4, 4)
kerr = 0
for pair in range(NP):
    for g in range(4):
        head = 2 * pair + (g >> 1)
        for r in range(4):
            dim = 4 * (g & 1) + r
            got = kb[pair, :T, g, r]
            want = k[:, hd * head + dim] if dim < hd else np.zeros(T)
            kerr = max(kerr, np.abs(got - want).max())
print("K max err", kerr, "nan in K", np.isnan(kb).sum())
want_k = np.zeros((NP, T, 4, 4))
for pair in range(NP):
    for g in range(4):
        for r in range(4):
            dim = 4 * (g & 1) + r; head = 2 * pair + (g >> 1)
            if dim < hd: want_k[pair, :, g, r] = k[:, hd * head + dim]
bad = np.argwhere(~(np.abs(kb[:, :T] - want_k) < 0.05))
print("K bad count", len(bad), "of", want_k.size, "first", bad[:12].tolist())
np.set_printoptions(precision=3, suppress=True, linewidth=200)
print("kb[0,0]", kb[0,0].ravel()); print("want  ", want_k[0,0].ravel())
print("kb[0,1]", kb[0,1].ravel()); print("want  ", want_k[0,1].ravel())
print("kb[3,50]", kb[3,50].ravel()); print("want   ", want_k[3,50].ravel())
print("K bad by pair", np.bincount(bad[:,0], minlength=NP), "by g", np.bincount(bad[:,2], minlength=4), "by tile", np.bincount(bad[:,1]//16, minlength=KT))
vb = bf16(raw[off_vbf:off_vbf + NP * S * NJ * 4 * 16 * 16].view(np.uint16)).reshape(NP, S, NJ, 4, 16, 8)
verr = 0
for pair in range(NP):
    for jb in range(NJ):
        for g in range(4):
            for d16 in range(16):
                head = 2 * pair + (d16 >> 3); dim = d16 & 7
                for e in range(8):
                    key = 32 * jb + (4 * g + e if e < 4 else 16 + 4 * g + (e - 4))
                    got = vb[pair, 0, jb, g, d16, e]
                    if key < T and dim < hd:
                        verr = max(verr, abs(got - v[key, hd * head + dim]))
                    elif key >= KT * 16 and got != 0:
                        print("nonzero V pad", pair, jb, g, d16, e, got)
print("V max err", verr, "nan in V", np.isnan(vb).sum(), "inf", np.isinf(vb).sum())
off_afr = off_vbf + NP * S * NJ * 4 * 16 * 16
half_ring = 2 * 2 * 11 * 1024
off_afr = max(off_afr, off_wsl + half_ring)
xf = bf16(raw[off_afr:off_afr + NTILE * KSX * 1024].view(np.uint16)).reshape(NTILE, KSX, 4, 16, 8)
qh = q.reshape(T, H, hd).transpose(1, 0, 2); kh = k.reshape(T, H, hd).transpose(1, 0, 2); vh = v.reshape(T, H, hd).transpose(1, 0, 2)
sc = qh @ kh.transpose(0, 2, 1) / np.sqrt(hd)
sc = sc - sc.max(-1, keepdims=True); p = np.exp(sc); p /= p.sum(-1, keepdims=True)
att = p @ vh
aerr = 0; nan_a = 0
for tile in range(NTILE):
    for ks in range(KSX):
        for gq in range(4):
            head = 4 * ks + gq
            for tk in range(16):
                tt = tile * 16 + tk
                if tt >= T: continue
                got = xf[tile, ks, gq, tk, :hd]
                nan_a += np.isnan(got).sum()
                if not np.isnan(got).all():
                    aerr = max(aerr, np.nanmax(np.abs(got - att[head, tt])))
errmap = np.zeros((H, NTILE))
for tile in range(NTILE):
    for ks in range(KSX):
        for gq in range(4):
            head = 4 * ks + gq
            for tk in range(16):
                tt = tile * 16 + tk
                if tt < T: errmap[head, tile] = max(errmap[head, tile], np.abs(xf[tile, ks, gq, tk, :hd] - att[head, tt]).max())
print("attn err by head(rows) x tile(cols)"); print(errmap)
print("attention-out max err", aerr, "nan", nan_a, "of", NTILE*KSX*4*16*hd)
print("sample got/want head0 tok0", xf[0,0,0,0,:hd], att[0,0])
