"""Latency of the training step's gradient exchange on ONE rank (what a single-GPU box can measure): fd_allreduce_grads over a
one-rank RCCL communicator for the flat fp32 gradient of the default model (3 197 744 parameters = 12.8 MB), events on the
caller's stream.  The N-rank number is the driver's (8-GPU node); this one bounds the fixed cost (launch + RCCL kernel + scale)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fourierdiffusion_amd import _C
ctx = _C.ctx(torch.device("cuda", 0)); lib = _C.lib()
uid = (C.c_ubyte * _C.FD_COMM_ID_BYTES)()
assert lib.fd_comm_unique_id(uid) == 0
_C.check(lib.fd_comm_init(ctx, 0, 1, uid), ctx)
for n in (3_197_744, 3_262_000):
    g = torch.randn(n, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        _C.check(lib.fd_allreduce_grads(ctx, g.data_ptr(), n, 1.0, st), ctx)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(); _C.check(lib.fd_allreduce_grads(ctx, g.data_ptr(), n, 1.0, st), ctx); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    print(f"fd_allreduce_grads, 1 rank, {n} floats ({4*n/1e6:.1f} MB): median {ts[len(ts)//2]:.1f} us, min {ts[0]:.1f} us")
_C.check(lib.fd_comm_destroy(ctx), ctx)
