"""HipMgr: Python face of the device manager, method-for-method the reference's `GpuMgr` contract
(omniscidb/DataMgr/GpuMgr.h:29-79; CUDA twin omniscidb/CudaMgr/CudaMgr.h:83-260), forwarding to the
C ABI (`hdk_hip_mgr_*`).  Method names keep the reference's camelCase on purpose."""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import check, lib


class DeviceBuffer:
    """A device allocation owned by the manager (cf. GpuBuffer under BufferMgr)."""

    def __init__(self, mgr, ptr, nbytes, device):
        self.mgr, self.ptr, self.nbytes, self.device = mgr, ptr, nbytes, device

    def free(self):
        if self.ptr:
            self.mgr.freeDeviceMem(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HipMgr:
    PLATFORM = "HIP"  # GpuMgrPlatform::HIP (new enumerator beside CUDA / L0, Shared/GpuPlatform.h:22)

    def __init__(self):
        self._lib = lib()
        n = C.c_int32(0)
        check(self._lib.hdk_hip_mgr_get_device_count(C.byref(n)))
        self._count = n.value
        if self._count == 0:
            raise RuntimeError("HipMgr: no HIP device visible")
        self._props = {}

    # --- GpuMgr virtuals ------------------------------------------------------------------
    def getDeviceCount(self):
        return self._count

    def getPlatform(self):
        return self.PLATFORM

    def setContext(self, device_num):
        check(self._lib.hdk_hip_mgr_set_context(device_num))

    def allocateDeviceMem(self, num_bytes, device_num):
        p = C.c_void_p(0)
        check(self._lib.hdk_hip_mgr_allocate_device_mem(num_bytes, device_num, C.byref(p)))
        return p.value

    def freeDeviceMem(self, device_ptr):
        check(self._lib.hdk_hip_mgr_free_device_mem(device_ptr))

    def copyHostToDevice(self, device_ptr, host, num_bytes, device_num):
        check(self._lib.hdk_hip_mgr_copy_host_to_device(device_ptr, _host_ptr(host), num_bytes, device_num))

    def copyHostToDeviceAsync(self, device_ptr, host, num_bytes, device_num):
        check(self._lib.hdk_hip_mgr_copy_host_to_device_async(device_ptr, _host_ptr(host), num_bytes,
                                                              device_num))

    def allocatePinnedHostMem(self, num_bytes) -> int:
        """CudaMgr::allocatePinnedHostMem (CudaMgr/CudaMgr.h:120): page-locked host memory, the source an asynchronous
        host-to-device copy needs to really be asynchronous."""
        p = C.c_void_p()
        check(self._lib.hdk_hip_mgr_allocate_pinned_host_mem(num_bytes, C.byref(p)))
        return p.value

    def freePinnedHostMem(self, host_ptr):
        check(self._lib.hdk_hip_mgr_free_pinned_host_mem(host_ptr))

    def synchronizeStream(self, device_num):
        check(self._lib.hdk_hip_mgr_synchronize_stream(device_num))

    def copyDeviceToHost(self, host, device_ptr, num_bytes, device_num):
        check(self._lib.hdk_hip_mgr_copy_device_to_host(_host_ptr(host), device_ptr, num_bytes, device_num))

    def copyDeviceToDevice(self, dest_ptr, src_ptr, num_bytes, dest_device_num, src_device_num):
        check(self._lib.hdk_hip_mgr_copy_device_to_device(dest_ptr, src_ptr, num_bytes, dest_device_num,
                                                          src_device_num))

    def zeroDeviceMem(self, device_ptr, num_bytes, device_num):
        check(self._lib.hdk_hip_mgr_zero_device_mem(device_ptr, num_bytes, device_num))

    def setDeviceMem(self, device_ptr, uc, num_bytes, device_num):
        check(self._lib.hdk_hip_mgr_set_device_mem(device_ptr, uc, num_bytes, device_num))

    def synchronizeDevices(self):
        check(self._lib.hdk_hip_mgr_synchronize_devices())

    def getDeviceProperties(self, device_num=0) -> A.DeviceProperties:
        if device_num not in self._props:
            p = A.DeviceProperties()
            check(self._lib.hdk_hip_mgr_get_device_properties(device_num, C.byref(p)))
            self._props[device_num] = p
        return self._props[device_num]

    def getTotalMem(self, device_num=0):
        return self.getDeviceProperties(device_num).global_mem

    def getMaxBlockSize(self):
        return self.getDeviceProperties(0).max_threads_per_block

    def getSubGroupSize(self):
        return self.getDeviceProperties(0).wavefront_size  # 64 on CDNA4

    def getGridSize(self):
        return self.getDeviceProperties(0).grid_size

    def getMinEUNumForAllDevices(self):
        return min(self.getDeviceProperties(d).num_cu for d in range(self._count))

    def hasSharedMemoryAtomicsSupport(self):
        return True

    def canLoadAsync(self):
        return True

    def hasFP64Support(self):
        return True

    def getMinSharedMemoryPerBlockForAllDevices(self):
        return min(self.getDeviceProperties(d).shared_mem_per_block for d in range(self._count))

    # --- conveniences above the contract ----------------------------------------------------
    def getStream(self, device_num):
        s = C.c_void_p(0)
        check(self._lib.hdk_hip_mgr_get_stream(device_num, C.byref(s)))
        return s.value

    def alloc(self, num_bytes, device_num) -> DeviceBuffer:
        return DeviceBuffer(self, self.allocateDeviceMem(num_bytes, device_num), num_bytes, device_num)

    def to_device(self, arr: np.ndarray, device_num) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        buf = self.alloc(max(arr.nbytes, 1), device_num)
        if arr.nbytes:
            self.copyHostToDevice(buf.ptr, arr, arr.nbytes, device_num)
        return buf

    def pinned_array(self, shape, dtype):
        """A numpy array over pinned host memory + the pointer to hand to freePinnedHostMem."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = self.allocatePinnedHostMem(max(n, 1))
        arr = np.frombuffer((C.c_char * max(n, 1)).from_address(ptr), dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        return arr, ptr

    def to_host(self, device_ptr, nbytes, device_num, dtype=np.int64) -> np.ndarray:
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        if nbytes:
            self.copyDeviceToHost(out, device_ptr, nbytes, device_num)
        return out


def _host_ptr(host):
    if isinstance(host, np.ndarray):
        return host.ctypes.data
    return host
