"""Host-side mirror of the reference's operator layer `geotransformer.modules` for the SE3ET hot path: same
sub-package names, class/function names, call signatures and state-dict names, implemented on HIP kernels
(libse3et_hip.so) instead of CPU C++ / eager PyTorch chains.  See INTEGRATION.md."""
