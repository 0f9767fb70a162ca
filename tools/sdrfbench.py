import sys, time, numpy as np, torch
sys.path.insert(0, 'hyper-graph-nets_amd'); sys.path.insert(0, '.')
from hgn_amd import graph_balancer as gb, synthetic
s, r = synthetic.two_way_edges(synthetic.grid_triangles(40, 40))
s, r = s.cuda(), r.cuda()
np.random.seed(0)
gb.sdrf(s, r, 1600, loops=3, remove_edges=True, tau=150)
torch.cuda.synchronize(); t0 = time.perf_counter()
added, removed = gb.sdrf(s, r, 1600, loops=150, remove_edges=True, tau=150)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('SDRF 150 loops on the 40x40 flag mesh (1600 nodes, 9282 directed edges): %.2f s total, %.1f ms/loop; added %d, removed %d' % (dt, dt / 150 * 1e3, len(added['senders']) // 2, len(removed['senders']) // 2))
A = torch.zeros(1600, 1600, device='cuda'); A[s, r] = 1
for _ in range(3): gb.forman_curvature(A)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): gb.forman_curvature(A)
torch.cuda.synchronize(); print('curvature matrix (A*A by rocBLAS + wave-per-edge kernel + nonzero): %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
