"""Where does the time of building a graph handle go?  usage: python scripts/probes/graph_build_time.py [shape]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import idgrec_amd.host as H, idgrec_amd.ops as ops, idgrec_amd.synth as S
shape = sys.argv[1] if len(sys.argv) > 1 else "synth-10M"
U, I, E = S.SHAPES[shape]
t = time.time(); users, items = S.generate(U, I, E, seed=0); print("generate %.1f s" % (time.time() - t), flush=True)
t = time.time(); ip, ix, dv = H.build_norm_adj(U, I, users, items); print("build_norm_adj %.1f s" % (time.time() - t), flush=True)
n = U + I
torch.cuda.init(); torch.zeros(1).cuda()
t = time.time(); G = ops.Graph(ip, ix, dv, n, n); torch.cuda.synchronize(); print("Graph (host CSR) %.1f s" % (time.time() - t), G.info(), flush=True)
del G
t = time.time()
dip, dix, ddv = torch.from_numpy(ip).cuda(), torch.from_numpy(ix).cuda(), torch.from_numpy(dv).cuda()
torch.cuda.synchronize(); print("H2D of the CSR %.2f s" % (time.time() - t), flush=True)
A = torch.sparse_csr_tensor(dip, dix.long(), ddv, size=(n, n))
t = time.time(); G2 = ops.Graph.from_torch_sparse(A); torch.cuda.synchronize(); print("Graph.from_torch_sparse (device CSR) %.1f s" % (time.time() - t), flush=True)
