#!/usr/bin/env python3
"""Constant wall-clock rate and CU count the HIP runtime reports (run on the GPU box)."""
import ctypes
import torch
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
v = ctypes.c_int(0)
for name, attr in (("wall clock kHz", 10017), ("clock instruction rate kHz", 10000), ("physical CUs", 10015), ("XCCs", 10018)):
    rc = hip.hipDeviceGetAttribute(ctypes.byref(v), attr, 0)
    print(name, v.value, "rc", rc)
print("multiprocessors", torch.cuda.get_device_properties(0).multi_processor_count)
