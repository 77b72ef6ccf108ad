import sys, ctypes, torch
sys.path.insert(0, '.')
from sota_imagenet_amd import native
L = native.lib()
c = ctypes.c_void_p()
native.check(L.mi355_bresnet50_create(ctypes.byref(c), 0, native.BF16, 256, 224, 224, 1000, 1))
print("bresnet workspace GB", L.mi355_bresnet50_workspace_bytes(c) / 1e9)
L.mi355_bresnet50_destroy(c)
