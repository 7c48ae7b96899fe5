import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
variant = sys.argv[1]
os.environ["NCCL_DEBUG"] = "INFO"
import torch
import ehyb_spmv_gpu_amd as E
from ehyb_spmv_gpu_amd import dist as D, _lib
lib = _lib.load()
print("variant", variant, "devices", E.device_count(), flush=True)
if variant == "setdev":
    lib.ehyb_device_set(0)
if variant == "torchinit":
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")
if variant == "alloc":
    p = C.c_void_p(); lib.ehyb_dev_alloc(1024, C.byref(p))
try:
    c = D.make_comm()
    print("OK", c.world, flush=True)
    c.destroy()
except Exception as e:
    print("FAIL", e, flush=True)
