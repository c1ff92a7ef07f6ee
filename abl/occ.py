import ctypes, sys
sys.path.insert(0, "/root/repo")
import torch
from facodec_amd import _lib
lib = _lib.load()
lib.fac_debug_bsplit_occupancy.restype = ctypes.c_int
for lds in (0, 32768, 65536, 73000, 75000, 79000, 81920, 100000, 146000):
    print(lds, lib.fac_debug_bsplit_occupancy(lds))
p = torch.cuda.get_device_properties(0)
print(p.name, p.multi_processor_count, getattr(p, "shared_memory_per_multiprocessor", None), getattr(p, "max_threads_per_multi_processor", None), getattr(p, "regs_per_multiprocessor", None))
