"""MI355X-native SE3ET hot path.  Importing the package has no side effect on the HIP runtime.

`request_blocking_sync(device_index)` asks the runtime to BLOCK host threads that wait for that GPU instead of spinning
(hipDeviceScheduleBlockingSync): a rank's host side is three threads that each wait on the device three times per forward, and a spinning wait
is a busy core (measured on the GPU box: 1.69 s of process CPU time per 1.56 s of waiting with the runtime's default, 0.13 s with the flag,
same wall time; tools/r5/blocking_sync_probe.py).  With eight ranks per node that is the difference between ~30 busy cores and ~8.

The flag belongs to ONE device and must be the process's FIRST GPU call: a rank calls it for the device it will run on (default: LOCAL_RANK)
before anything touches torch.cuda (bench.py, tools/train_bench.py do).  Measured in round 6 (tools/r6/live_flags.py, profiles/r06_blocking_sync.txt):
on a device torch has already initialised the call is ACCEPTED (rc 0, hipGetDeviceFlags reads it back) but waits keep spinning, and set
between torch.cuda.set_device() and the first allocation it leaves the null stream -- created before it -- with waits that do not wait (a
one-pair forward on the default stream then read its pinned counts before the copy had landed; a plain synchronize hung).  So the request
is REFUSED here once torch has initialised CUDA/HIP; SE3_BLOCKING_SYNC=0 turns every request into a no-op."""
import ctypes as _ctypes
import os as _os

_blocking_sync = {}          # device index -> status text


def request_blocking_sync(device_index=None):
    """hipSetDevice(device_index) (None: $LOCAL_RANK, else 0) + hipSetDeviceFlags(hipDeviceScheduleBlockingSync), as the process's first GPU
    call.  Returns the status text for that device: 'set', 'not requested' (SE3_BLOCKING_SYNC=0), 'no GPU', 'too late ...' (torch has
    already initialised the runtime: nothing is changed), or the reason it failed."""
    if device_index is None:
        device_index = int(_os.environ.get('LOCAL_RANK', '0'))
    device_index = int(device_index)
    if _os.environ.get('SE3_BLOCKING_SYNC', '1') == '0':           # ('force': bench.py requests it in multi-rank runs as well)
        status = 'not requested'
    elif not _os.path.exists('/dev/kfd'):            # no GPU in this container: nothing to ask (and no runtime to wake up)
        status = 'no GPU'
    elif _torch_runtime_is_live():
        status = 'too late (torch has initialised the HIP runtime: the flag must precede the first GPU call of the process)'
    elif _blocking_sync.get(device_index) == 'set':
        status = 'set'
    else:
        try:
            hip = _ctypes.CDLL('libamdhip64.so')
            # device 0 is the process's current device before any HIP call: the flag call alone, as the very first call (round 5's form;
            # with hipSetDevice(0) in front of it the waits of a bench rank kept 1.4-1.8 cores busy instead of 0.8)
            rc = hip.hipSetDevice(_ctypes.c_int(device_index)) if device_index != 0 else 0
            if rc == 0:
                rc = hip.hipSetDeviceFlags(_ctypes.c_uint(4))          # hipDeviceScheduleBlockingSync
            status = 'set' if rc == 0 else 'refused (hipError %d)' % rc
        except OSError as e:
            status = 'libamdhip64 not loadable: %s' % e
    _blocking_sync[device_index] = status
    return status


def _torch_runtime_is_live():
    """Has torch initialised CUDA / HIP in this process?  Read from torch.cuda's own Python flag WITHOUT calling into torch:
    torch.cuda.is_initialized() goes through torch._C (_cuda_isInBadFork), and a request made behind it was only half effective (a bench
    rank kept 1.7 cores busy instead of 0.8: round 6, same-box A/B against the round-5 tree)."""
    import sys
    torch = sys.modules.get('torch')
    cuda = getattr(torch, 'cuda', None) if torch is not None else None
    return bool(getattr(cuda, '_initialized', False))


def blocking_sync_status(device_index):
    """Status text of the last request for that device ('not requested' if there was none)."""
    return _blocking_sync.get(int(device_index), 'not requested')
