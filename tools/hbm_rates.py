import torch, time
n = 1 << 30   # floats: 4 GB
a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda')
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print('fill 4 GB: %.3f ms = %.2f TB/s written' % (ms, 4.295 / ms))
ms = t(lambda: b.copy_(a)); print('copy 4 GB -> 4 GB: %.3f ms = %.2f TB/s moved (read + write)' % (ms, 8.59 / ms))
ms = t(lambda: a.sum()); print('sum of 4 GB: %.3f ms = %.2f TB/s read' % (ms, 4.295 / ms))
c = torch.empty(n // 2, device='cuda')
ms = t(lambda: torch.add(a[:n//2], 1.0, out=c)); print('read 2 GB write 2 GB (add): %.3f ms = %.2f TB/s' % (ms, 4.295 / ms))
