import os, sys, time
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
x = torch.zeros(99587 + 23, device='cuda')
for _ in range(5):
    dist.all_reduce(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    dist.all_reduce(x)
    y = x[99587:].cpu()
dt = (time.perf_counter() - t0) / 50
print('all_reduce(400KB)+cpu readback: %.3f ms' % (dt * 1e3), 'OMP_NUM_THREADS', os.environ.get('OMP_NUM_THREADS'))
# with a busy GPU before it
a = torch.randn(8192, 8192, device='cuda')
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    b = a @ a
    dist.all_reduce(x)
    y = x[99587:].cpu()
dt1 = (time.perf_counter() - t0) / 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    b = a @ a
    y = x[99587:].cpu()
dt2 = (time.perf_counter() - t0) / 20
print('matmul + all_reduce + readback: %.3f ms ; matmul + readback: %.3f ms' % (dt1 * 1e3, dt2 * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    b = a @ a
    torch.cuda.current_stream().synchronize()
    dist.all_reduce(x)
    y = x[99587:].cpu()
dt3 = (time.perf_counter() - t0) / 20
print('matmul + SYNC + all_reduce + readback: %.3f ms' % (dt3 * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    b = a @ a
    w = dist.all_reduce(x, async_op=True)
    w.wait()
    y = x[99587:].cpu()
dt4 = (time.perf_counter() - t0) / 20
print('matmul + async all_reduce + wait + readback: %.3f ms' % (dt4 * 1e3))
dist.destroy_process_group()
