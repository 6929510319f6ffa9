'''What plain streaming kernels reach on the device: torch copy_ / add / sum / fill_ on 64 MB .. 2 GB
float32 arrays, HIP events (the calibration quoted in profiles/r02_fft_analysis.txt).
'''
import torch
for mb in (64, 256, 1024, 2048):
    n = mb * (1 << 20) // 4
    a = torch.rand(n, device='cuda'); b = torch.empty_like(a); c = torch.rand(n, device='cuda')
    for name, fn in (('copy', lambda: b.copy_(a)), ('add', lambda: torch.add(a, c, out=b)), ('read-sum', lambda: a.sum()), ('fill', lambda: b.fill_(1.0))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nbytes = {'copy': 2, 'add': 3, 'read-sum': 1, 'fill': 1}[name] * n * 4
        print(f'{mb:5d} MB {name:9s} {ms*1e3:8.1f} us  {nbytes/ms/1e9:6.2f} TB/s')
