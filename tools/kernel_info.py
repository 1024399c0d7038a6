import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
from mmseq_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] == "system":
    _lib._share_hip_runtime_with_torch = lambda: None
L = _lib.load()
v, l, s, o = C.c_int(), C.c_int(), C.c_int(), C.c_int()
rc = L.mmg_selftest_kernel_info(0, C.byref(v), C.byref(l), C.byref(s), C.byref(o))
print(sys.argv[1:], "rc", rc, "vgprs", v.value, "lds", l.value, "scratch", s.value, "resident/CU", o.value)
