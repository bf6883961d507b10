"""MI355X-native SE3ET hot path.  Importing the package asks the HIP runtime -- before its first use in this process -- to
BLOCK host threads that wait for the GPU instead of spinning (hipDeviceScheduleBlockingSync): a rank's host side is three threads that
each wait on the device three times per forward, and a spinning wait is a busy core (measured on the GPU box: 1.69 s of process CPU time
per 1.56 s of waiting with the runtime's default, 0.13 s with the flag, same wall time; tools/r5/blocking_sync_probe.py).  With eight ranks
per node that is the difference between ~30 busy cores and ~8.  Has no effect once the runtime is initialised (import the package before the
first torch.cuda call); SE3_BLOCKING_SYNC=0 skips it."""
import ctypes as _ctypes
import os as _os

BLOCKING_SYNC_STATUS = 'not requested'


def _request_blocking_sync():
    global BLOCKING_SYNC_STATUS
    if _os.environ.get('SE3_BLOCKING_SYNC', '1') == '0':
        return
    if not _os.path.exists('/dev/kfd'):              # no GPU in this container: nothing to ask (and no runtime to wake up)
        BLOCKING_SYNC_STATUS = 'no GPU'
        return
    try:
        hip = _ctypes.CDLL('libamdhip64.so')
        rc = hip.hipSetDeviceFlags(_ctypes.c_uint(4))          # hipDeviceScheduleBlockingSync
        BLOCKING_SYNC_STATUS = 'set' if rc == 0 else 'refused (hipError %d: the runtime was initialised before se3et_amd was imported)' % rc
    except OSError as e:
        BLOCKING_SYNC_STATUS = 'libamdhip64 not loadable: %s' % e


_request_blocking_sync()
