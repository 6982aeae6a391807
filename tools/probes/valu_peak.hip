// valu_peak.hip -- measures the issue rate of the 32-bit integer VALU instructions the scan kernels are made of
// (v_alignbit_b32, v_or_b32, v_xor_b32, v_bitop3_b32) against v_fma_f32, with 1..8 waves per SIMD.
// The roofline of a bit-parallel kernel is VALU issue; this gives its denominator on the actual part.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int UNROLL = 16;     // independent chains per lane
constexpr int ITERS = 65536;

template <int OP>
__global__ __launch_bounds__(64) void probe(uint32_t *out, uint32_t seed) {
    uint32_t a[UNROLL];
    float f[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { a[i] = seed + threadIdx.x * 131u + i; f[i] = (float)a[i]; }
    uint32_t s = seed | 1u;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) % UNROLL], 7);
            if (OP == 1) asm volatile("v_or_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) % UNROLL]));
            if (OP == 2) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) % UNROLL]));
            if (OP == 3) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x36" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) % UNROLL]), "v"(s));
            if (OP == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i + 1) % UNROLL]), "v"(f[(i + 2) % UNROLL]));
            if (OP == 5) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) % UNROLL]), "v"(s));
            if (OP == 6) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) % UNROLL]));
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) r ^= a[i] ^ (uint32_t)f[i];
    if (r == 0x12345u) out[0] = r;
}

template <int OP>
void run(const char *name, int waves_per_simd, uint32_t *d) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(64), 0, 0, d, 3u);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(64), 0, 0, d, 3u + r);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = 5.0 * blocks * (double)ITERS * UNROLL;          // wave-instructions
    const double per_simd_cycle = instr / (ms * 1e-3) / (p.multiProcessorCount * 4.0) / (p.clockRate * 1e3);
    printf("%-16s waves/SIMD %d  %8.1f G wave-instr/s  %6.2f T lane-ops/s  %.3f wave-instr/cycle/SIMD (clock %d MHz)\n", name,
           waves_per_simd, instr / (ms * 1e-3) / 1e9, instr * 64 / (ms * 1e-3) / 1e12, per_simd_cycle, p.clockRate / 1000);
}

int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 64);
    for (int w : {1, 2, 4, 8}) {   // 4 = what scan_perfect_kernel runs at (111 VGPRs)
        run<0>("v_alignbit_b32", w, d);
        run<1>("v_or_b32", w, d);
        run<2>("v_xor_b32", w, d);
        run<3>("v_bitop3_b32", w, d);
        run<5>("v_and_or_b32", w, d);
        run<6>("v_bcnt_u32_b32", w, d);
        run<4>("v_fma_f32", w, d);
    }
    (void)hipFree(d);
    return 0;
}
