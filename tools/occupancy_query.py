"""What the HIP occupancy API answers for the attention kernels (workgroups per CU) at several dynamic-LDS sizes.  usage: python tools/occupancy_query.py"""
import ctypes as C, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
torch.zeros(1, device="cuda:0")
L = hip.lib()
rt = C.CDLL("libamdhip64.so")
rt.hipOccupancyMaxActiveBlocksPerMultiprocessor.argtypes = [C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_size_t]
rt.hipFuncSetAttribute.argtypes = [C.c_void_p, C.c_int, C.c_int]
for sym, block in (("_Z27dit_attention_stream_kernelILi0EEvPKDF16_S1_S1_PDF16_iiiiijf", 512), ("_Z20dit_attention_kernelILi256ELi1ELi64ELi0ELi1ELi1EEvPKDF16_S1_S1_PDF16_iifi", 512)):
    f = C.addressof(C.c_char.in_dll(L, sym))
    print(sym[:40], "set attr rc", rt.hipFuncSetAttribute(f, 8, 163840))  # hipFuncAttributeMaxDynamicSharedMemorySize = 8
    for lds in (0, 32768, 65536, 69632, 73728, 77824, 81920, 98304):
        n = C.c_int(-1)
        rc = rt.hipOccupancyMaxActiveBlocksPerMultiprocessor(C.byref(n), f, block, lds)
        print(f"  {sym[:36]} block {block} dynamic LDS {lds:6d}: rc {rc}, workgroups per CU {n.value}")
p = torch.cuda.get_device_properties(0)
print("CUs", p.multi_processor_count, "shared per block", getattr(p, "shared_memory_per_block", None), "per multiprocessor", getattr(p, "shared_memory_per_multiprocessor", None), "regs per mp", getattr(p, "regs_per_multiprocessor", None))
