import sys, ctypes as C
sys.path.insert(0, '/root/repo')
from ipdm_pytorch_amd import _lib
t = C.c_float()
with _lib.option("conv_dbg", 64):
    _lib.call("ipdm_bench_conv2d", 8, 128, 0, 512, 512, 128, 3, 1, 2, 1, 2, C.byref(t))
print(t.value)
