// A stub HIP runtime for the CPU sanitizer build of the library's HOST side (tests/test_sanitizers.py): "device" memory comes from malloc (so
// AddressSanitizer sees every table upload, staging copy and ring write of the host code), copies are memcpy, kernel launches do nothing,
// streams and events are tokens. Eight devices are reported so that the sharded handle routes across distinct device ids. No audio comes out
// of this: it exists to run graph construction, event / command-list bookkeeping, topology rebuilds and the handles' routing under ASan + UBSan
// without a GPU. Built host-only (hipcc --cuda-host-only) against the real HIP headers, linked INSTEAD of libamdhip64.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>
#include <cstring>

static thread_local int t_device = 0;
static thread_local dim3 t_grid, t_block;
static thread_local size_t t_shmem = 0;
static thread_local hipStream_t t_stream = nullptr;

extern "C" {

hipError_t hipGetDeviceCount(int* n) { *n = 8; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= 8) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub HIP runtime"; }
hipError_t hipGetLastError(void) { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned int) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned int) { static unsigned n = 0; return (++n & 1) ? hipSuccess : hipErrorPeerAccessAlreadyEnabled; }   // (both answers the handle accepts)
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { static unsigned n = 0; (void)s; return (++n & 1) ? hipErrorNotReady : hipSuccess; }   // (both answers get exercised)
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.1f; return hipSuccess; }

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
hipError_t hipExtLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t, hipEvent_t, hipEvent_t, int) { return hipSuccess; }

// what the compiler-generated kernel stubs and module constructors call
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) { t_grid = grid; t_block = block; t_shmem = shmem; t_stream = stream; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) { *grid = t_grid; *block = t_block; *shmem = t_shmem; *stream = t_stream; return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* handle[1]; return handle; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned int, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}

}  // extern "C"
