import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
torch.cuda.set_device(0); dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from monorun_amd.parallel import RcclAllGather
ag = RcclAllGather(dev)
a = torch.zeros(88 * 1024, dtype=torch.uint8, device=dev); b = torch.zeros_like(a)
for _ in range(20): ag.gather(a, b)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): ag.gather(a, b)
h = time.perf_counter() - t; torch.cuda.synchronize(); tot = time.perf_counter() - t
print(f'direct RCCL gather: host {h/200*1e6:.1f} us per call, incl. completion {tot/200*1e6:.1f} us')
for _ in range(20): dist.all_gather_into_tensor(b, a)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): dist.all_gather_into_tensor(b, a)
h = time.perf_counter() - t; torch.cuda.synchronize(); tot = time.perf_counter() - t
print(f'c10d all_gather_into_tensor: host {h/200*1e6:.1f} us per call, incl. completion {tot/200*1e6:.1f} us')
ag.close(); dist.destroy_process_group()
