"""Probe (GPU): does hipSetDeviceFlags(hipDeviceScheduleBlockingSync) before the first HIP call of the process lower the CPU time of a
thread that waits on the GPU?   python tools/r5/blocking_sync_probe.py [flags]   (0 = auto, 1 = spin, 2 = yield, 4 = blocking sync)"""
import ctypes, os, sys, time
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 4
hip = ctypes.CDLL('libamdhip64.so')
rc = hip.hipSetDeviceFlags(ctypes.c_uint(flags)) if flags >= 0 else -1
import torch
x = torch.randn(8192, 8192, device='cuda')
torch.cuda.synchronize()
for mode in ('synchronize', 'blocking event'):
    c0, t0 = time.process_time(), time.perf_counter()
    for _ in range(20):
        for _ in range(10):
            y = x @ x
        if mode == 'synchronize':
            torch.cuda.synchronize()
        else:
            ev = torch.cuda.Event(blocking=True); ev.record(); ev.synchronize()
    print('flags %d rc %d  %-15s wall %.3f s  process CPU %.3f s' % (flags, rc, mode, time.perf_counter() - t0, time.process_time() - c0))
