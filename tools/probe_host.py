import os, time, torch
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max','/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, 'n/a')
print('torch threads', torch.get_num_threads())
a=torch.randn(2048,2048); 
for n in (torch.get_num_threads(), 8, 16, 32):
    torch.set_num_threads(n); t=time.time(); 
    for _ in range(5): a@a
    print(n, 'threads', (time.time()-t)/5*1e3, 'ms per 2048^3 matmul')
