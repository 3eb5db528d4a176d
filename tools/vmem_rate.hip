// Micro-benchmark: how many vector-memory instructions per microsecond does one CU issue?
// 8 waves per CU (2 per SIMD, like the trainer's kernels) loop over buffer loads of an L2-resident
// 64 KB region per workgroup.  hipcc --offload-arch=gfx950 -O3 tools/vmem_rate.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// MODE 0: dwordx4 loads in range, 1: dwordx4 loads out of range (empty descriptor), 2: dword loads in range,
// 3: dwordx4 loads, 32 rows x 32 B per instruction (scattered), 4: dwordx4 stores in range
template <int MODE>
__global__ __launch_bounds__(512) void k(float *buf, int iters, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *base = buf + (size_t)blockIdx.x * 16384;  // 64 KB per workgroup
    const rsrc_t r = mk(base, MODE == 1 ? 0u : 65536u);
    int vo = (wave * 2048 + lane * 16) & 65535;
    if (MODE == 3) vo = (((lane & 31) * 2048) + (lane >> 5) * 16 + wave * 64) & 65535;
    if (MODE == 2) vo = (wave * 1024 + lane * 4) & 65535;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int so = (u * 8192 + (it & 3) * 1024) & 65535;
            if (MODE == 2) {
                acc.x += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
            } else if (MODE == 4) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, acc), r, vo, so, 0);
            } else {
                const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
int main() {
    float *buf, *out;
    CK(hipMalloc(&buf, (size_t)256 * 65536)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, (size_t)256 * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    const char *names[] = {"dwordx4 loads, L2/L1-resident", "dwordx4 loads, out of range (empty descriptor)", "dword loads, resident",
                           "dwordx4 loads, 32 rows x 32 B per instruction", "dwordx4 stores"};
    for (int mode = 0; mode < 5; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double instr_per_cu = 8.0 * iters * 8;  // 8 waves x iters x 8 instructions
        const double ns_per = best * 1e6 / instr_per_cu;
        printf("%-52s %7.1f ns per wave-instruction per CU  (%.1f GB/s per CU, %.1f TB/s chip)\n", names[mode], ns_per,
               (mode == 2 ? 256.0 : 1024.0) / ns_per, (mode == 2 ? 256.0 : 1024.0) / ns_per * 256 / 1000);
    }
    return 0;
}
