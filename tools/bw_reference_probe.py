import torch, sys
sys.path.insert(0,'/root/repo')
dev=torch.device('cuda:0')
flush = torch.empty(1 << 28, device=dev)
def cold(fn, iters=4, do_flush=True):
    ts=[]
    for _ in range(iters+1):
        if do_flush: flush.fill_(1.0)
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e)*1e3)
    return min(ts[1:])
n134=134217728//4
a=torch.empty(n134,device=dev); b=torch.empty(n134,device=dev)
a84=torch.empty(84*1024*1024//4,device=dev); b84=torch.empty_like(a84)
small=torch.empty(33554432//4,device=dev)
for fl in (True, False):
    print('flush' if fl else 'warm')
    t=cold(lambda: a.fill_(2.0), do_flush=fl); print(' fill 134 MB: %.1f us = %.2f TB/s'%(t,134.2/t))
    t=cold(lambda: b84.copy_(a84), do_flush=fl); print(' copy 84->84 MB: %.1f us = %.2f TB/s'%(t,2*88.1/t))
    t=cold(lambda: b.copy_(a), do_flush=fl); print(' copy 134->134 MB: %.1f us = %.2f TB/s'%(t,2*134.2/t))
    t=cold(lambda: torch.sum(a), do_flush=fl); print(' sum-read 134 MB: %.1f us = %.2f TB/s'%(t,134.2/t))
    t=cold(lambda: a.view(-1,4)[:,0:1].expand(-1,4).contiguous() if False else torch.mul(small,2.0,out=a[:small.numel()]), do_flush=fl); print(' 33->33: %.1f us'%t)
# flush with a READ instead of a write (clean cache lines)
def cold_rd(fn, iters=4):
    ts=[]
    for _ in range(iters+1):
        torch.sum(flush); 
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e)*1e3)
    return min(ts[1:])
print('read-flush (cache left clean)')
t=cold_rd(lambda: a.fill_(2.0)); print(' fill 134 MB: %.1f us = %.2f TB/s'%(t,134.2/t))
t=cold_rd(lambda: b84.copy_(a84)); print(' copy 84->84 MB: %.1f us = %.2f TB/s'%(t,2*88.1/t))
t=cold_rd(lambda: b.copy_(a)); print(' copy 134->134 MB: %.1f us = %.2f TB/s'%(t,2*134.2/t))
