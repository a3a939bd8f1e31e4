import sys, time, numpy as np
sys.path[:0]=['/root/repo','/root/repo/kompass-core_amd']
import torch, kompass_hip as kh
n, bins, step = 1_000_000, 2048, 16
rng = np.random.default_rng(0)
xyz = np.zeros((n, 4), np.float32); xyz[:,0]=rng.uniform(-30,30,n); xyz[:,1]=rng.uniform(-30,30,n); xyz[:,2]=rng.uniform(0,1,n)
host = xyz.reshape(-1).view(np.int8); dev = torch.from_numpy(host.copy()).cuda(); torch.cuda.synchronize()
ctx = kh.CloudContext(max_bytes=host.size, max_bins=bins)
call = lambda: ctx.to_laserscan(None, step, n*step, 1, n, 0, 4, 8, 25.0, 0.0, 1.0, num_bins=bins, device_ptr=dev.data_ptr(), nbytes=host.size)
for _ in range(50): call()
ts=[]
for _ in range(300):
    t=time.perf_counter(); call(); ts.append(time.perf_counter()-t)
print('call p50 %.1f us'%(np.percentile(ts,50)*1e6), 'rebinned', ctx.last_rebinned() if hasattr(ctx,'last_rebinned') else None)
ctx.timing_enable(True)
acc={}
for _ in range(50):
    call()
    for name, ms in ctx.timings(): acc.setdefault(name,[]).append(ms)
print({k:round(float(np.mean(v))*1e3,1) for k,v in acc.items()})
