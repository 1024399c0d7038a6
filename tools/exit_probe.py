"""Which load order of RCCL and PyTorch survives interpreter exit on this image (run on the GPU box): modes
   C0 dlopen torch's librccl (RTLD_GLOBAL) then import torch     C1 the same without RTLD_GLOBAL     C2 import torch, then dlopen
   C3 dlopen /opt/rocm's librccl then import torch                 C4 dlopen torch's librccl, never import torch"""
import ctypes, sys
mode = sys.argv[1]
T = "/usr/local/lib/python3.10/dist-packages/torch/lib/librccl.so"
if mode == "C0":
    ctypes.CDLL(T, mode=ctypes.RTLD_GLOBAL); import torch
elif mode == "C1":
    ctypes.CDLL(T); import torch
elif mode == "C2":
    import torch; ctypes.CDLL(T, mode=ctypes.RTLD_GLOBAL)
elif mode == "C3":
    ctypes.CDLL("/opt/rocm/lib/librccl.so.1", mode=ctypes.RTLD_GLOBAL); import torch
elif mode == "C4":
    ctypes.CDLL(T, mode=ctypes.RTLD_GLOBAL)
print("mode", mode, "done", flush=True)
if mode in ("A", "B"):   # the library's own group (A), with an injected failure and ncclCommAbort (B), then import torch
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mmseq_amd import gibbs as G
    prob = G.Problem.synthetic(20000, 500, 5)
    mu0, _ = prob.start_values()
    s = G.Sampler(prob, mu0, seed=1, gibbs_iter=8, trace_len=8)
    grp = G.Group([0])
    if mode == "B":
        with G.options(group_fail=0):
            try:
                grp.run_sharded([s], 4)
            except Exception as e:
                print("expected failure:", str(e)[:60])
    else:
        grp.run_sharded([s], 4)
    grp.close(); s.close(); prob.close()
    import torch
    print("mode", mode, "torch imported", flush=True)
