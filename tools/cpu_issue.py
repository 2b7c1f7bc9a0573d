"""Host side of a training step: python tools/cpu_issue.py [batch] [profile]

Prints the time the Python thread needs to ISSUE a step (no synchronisation inside the loop) next to the synchronised step time,
and with `profile` a cProfile of 30 steps.  Measured (round 6, one MI355X box): batch 8: 5.34 ms issue / 5.82 ms per step, of which
0.7 ms is RegionLoss._upload waiting on its 4-deep pinned ring - i.e. the host is AHEAD of the GPU and the batch-8 step is bound
by the GPU side (293 dependent launches of 5-20 us), not by Python; batch 64: 24.6 ms per step.
"""
import os, sys, time, torch
sys.path.insert(0, '/root/repo')
os.environ.setdefault('SSP_TUNE_CACHE', '/root/repo/gpurun_out/tune_cache_cpuissue.json')
from singleshotpose_amd.darknet import Darknet
from singleshotpose_amd.region_loss import RegionLoss
from singleshotpose_amd.optim import SGD
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
model = Darknet('/root/repo/cfg/yolo-pose.cfg').to(dev).train()
crit = RegionLoss(); crit.verbose = False
opt = SGD(model.parameters(), lr=1e-4, momentum=0.9, weight_decay=5e-4)
x, tgt = bench.synthetic_batch(B, 416, 416, 1000, dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model(x), tgt, 20)
    loss.backward()
    opt.step()
for _ in range(8): step()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('B=%d: CPU issue %.3f ms/step, total %.3f ms/step' % (B, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
if len(sys.argv) > 2 and sys.argv[2] == 'profile':
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(30): step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)
    st.sort_stats('cumulative').print_stats(25)
