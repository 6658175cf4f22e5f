import torch, time
for gb in (1, 4, 15):
    n = gb * (1 << 30) // 4
    x = torch.empty(n, device='cuda')
    y = torch.empty(n, device='cuda') if gb <= 4 else None
    for name, f in (('zero_', lambda: x.zero_()), ('fill_', lambda: x.fill_(1.5)), ('copy_', (lambda: x.copy_(y)) if y is not None else None)):
        if f is None: continue
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print('%2d GB %-6s %8.3f ms  %7.1f GB/s (bytes written%s)' % (gb, name, ms, gb * 1.073741824 / ms * 1e3 * (1 if name != 'copy_' else 2), '' if name != 'copy_' else ' + read'))
    del x, y
