// How much LDS can one workgroup be given on this device?  (MI355X: 160 KB per CU; the default limit per workgroup is 64 KB.)
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_probe tools/probes/lds_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void touch(uint32_t *out, int words) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = (uint32_t)i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) acc ^= lds[words - 1 - i];
    atomicXor(out, acc);
}
int main() {
    int v = 0;
    hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0); printf("MaxSharedMemoryPerBlock %d\n", v);
    hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0); printf("MaxSharedMemoryPerMultiprocessor %d\n", v);
    hipDeviceGetAttribute(&v, hipDeviceAttributeMaxThreadsPerBlock, 0); printf("MaxThreadsPerBlock %d\n", v);
    uint32_t *out; hipMalloc(&out, 4);
    for (int kb : {48, 64, 96, 128, 144, 160}) {
        for (int threads : {512, 1024}) {
            hipMemset(out, 0, 4);
            hipError_t a = hipFuncSetAttribute((const void *)touch, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
            hipLaunchKernelGGL(touch, dim3(4), dim3(threads), (size_t)kb * 1024, 0, out, kb * 256);
            hipError_t l = hipGetLastError();
            hipError_t s = hipDeviceSynchronize();
            printf("%3d KB, %4d threads: set-attribute %s, launch %s, sync %s\n", kb, threads, hipGetErrorName(a), hipGetErrorName(l), hipGetErrorName(s));
            if (l != hipSuccess || s != hipSuccess) break;
        }
    }
    return 0;
}
