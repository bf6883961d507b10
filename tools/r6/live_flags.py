"""Round 6: what hipSetDeviceFlags(hipDeviceScheduleBlockingSync) does on a LIVE device (ADVICE round 5 said HIP accepts it)."""
import ctypes
import json
import sys
import time

import torch

sys.path.insert(0, '.')
import se3et_amd

hip = ctypes.CDLL('libamdhip64.so')
torch.cuda.set_device(0)
when = sys.argv[1] if len(sys.argv) > 1 else 'live'
if when == 'early':
    print('early request:', se3et_amd.request_blocking_sync(0))
x = torch.randn(8192, 8192, device='cuda')
torch.cuda.synchronize()


def wait_cost():
    y = x
    for _ in range(40):
        y = (y @ x) * 1e-4
    c0, w0 = time.process_time(), time.perf_counter()
    torch.cuda.synchronize()
    return round(time.process_time() - c0, 3), round(time.perf_counter() - w0, 3)


wait_cost()
print('before:', wait_cost())
if when == 'live':
    print('live request:', se3et_amd.request_blocking_sync(0), 'last error', hip.hipGetLastError(), hip.hipPeekAtLastError())
    print('after :', wait_cost())
    print('second request:', se3et_amd.request_blocking_sync(0), 'last error', hip.hipGetLastError())
    print('after2:', wait_cost())
flags = ctypes.c_uint(0)
print('hipGetDeviceFlags rc', hip.hipGetDeviceFlags(ctypes.byref(flags)), 'flags', flags.value)
