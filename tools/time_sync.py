import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, fun_ofdm_amd as foa
iq, pays = bench.make_workload(np.arange(10000), 7919)
t=time.time(); want = foa.find_alignments(iq); th=time.time()-t
dev = torch.device("cuda", 0)
rx = foa.Receiver(0)
t_iq = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).to(dev)
cap = iq.size // 300 + 16
t_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev); t_ends = torch.zeros(cap, dtype=torch.int64, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t=time.time(); n = rx.sync_dev(t_iq, t_desc, t_ends); torch.cuda.synchronize(); td=time.time()-t
got = t_desc.cpu().numpy()[:n*48].view(foa.frame_desc_dtype)
print('host sync %.2f s, device sync %.3f ms, n %d vs %d, equal pos %s, max phasor diff %.2e' % (th, td*1e3, n, want.size,
      np.array_equal(got['lts1_pos'], want['lts1_pos']) and np.array_equal(got['rot_start'], want['rot_start']), max(np.abs(got[k]-want[k]).max() for k in ('c','s','c_prev','s_prev'))))
print('device sync: %.1f Gsample/s over %d samples' % (iq.size/td/1e9, iq.size))
