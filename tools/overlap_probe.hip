// Micro-benchmark: do MFMA work and vector-memory work overlap on a CU when they are issued by
// DIFFERENT waves (specialised) rather than interleaved in the same waves?
// One 512-thread workgroup per CU.  MFMA job: a chain of dependent v_mfma_f32_32x32x2_f32 per wave.
// Memory job: dwordx4 buffer loads + stores over an L2-resident 64 KB region.
//   mode 0: waves 0-3 MFMA only, waves 4-7 idle        mode 1: waves 4-7 memory only, waves 0-3 idle
//   mode 2: waves 0-3 MFMA, waves 4-7 memory (specialised)
//   mode 3: all 8 waves do half of each, interleaved in their instruction streams (like k_dwp)
// hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o /tmp/op && /tmp/op
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
template <int MODE>
__global__ __launch_bounds__(512) void k(float *buf, int iters, float *out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *base = buf + (size_t)blockIdx.x * 16384;
    const rsrc_t r = mk(base, 65536u);
    const int vo = ((wave & 3) * 2048 + lane * 16) & 65535;
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    float4 st = make_float4(lane, 1, 2, 3), ld = make_float4(0, 0, 0, 0);
    float a = lane * 0.001f, b = 1.0f;
    const bool spec6 = MODE == 6;
    const bool do_mfma = spec6 ? wave < 4 : MODE == 0 ? wave < 4 : MODE == 1 ? false : MODE == 2 ? wave < 4 : (MODE == 4 || MODE == 5) ? wave >= 4 : true;
    const bool do_mem = MODE == 0 ? false : MODE == 1 ? wave >= 4 : MODE == 2 ? wave >= 4 : (MODE == 4 || MODE == 5) ? wave < 4 : spec6 ? wave >= 4 : true;
    if (MODE == 5 && do_mem) __builtin_amdgcn_s_setprio(3);
    if (MODE == 6) { if (wave >= 4) __builtin_amdgcn_s_setprio(3); }
    // per iteration and CU: 4 x 16 MFMAs per SIMD-pair budget, 4 x (6 loads + 2 stores) -- the same totals in every mode
    const int n_mfma = (MODE == 3) ? 8 : 16, n_mem = (MODE == 3) ? 1 : 2;
    for (int it = 0; it < iters; it++) {
        if (MODE == 3) {
            for (int g = 0; g < n_mem; g++) {
                for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                for (int u = 0; u < 3; u++) {
                    const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, (u * 8192 + (it & 3) * 1024) & 65535, 0));
                    ld.x += v.x;
                }
                for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, st), r, vo, (32768 + (it & 7) * 1024) & 65535, 0);
            }
        } else {
            if (do_mfma)
                for (int u = 0; u < n_mfma; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (do_mem)
                for (int g = 0; g < n_mem; g++) {
                    for (int u = 0; u < 3; u++) {
                        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, (u * 8192 + (it & 3) * 1024 + g * 4096) & 65535, 0));
                        ld.x += v.x;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, st), r, vo, (32768 + (it & 7) * 1024 + g * 512) & 65535, 0);
                }
        }
    }
    float s = ld.x;
    for (int i = 0; i < 16; i++) s += acc[i];
    if (s == 123.456f) out[0] = s;
}
int main() {
    float *buf, *out;
    CK(hipMalloc(&buf, (size_t)256 * 65536)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, (size_t)256 * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    const char *names[] = {"MFMA only (waves 0-3: 16 MFMAs per iteration)", "memory only (waves 4-7: 6 loads + 2 stores per iteration)",
                           "specialised: waves 0-3 MFMA, waves 4-7 memory", "interleaved: every wave 8 MFMAs + 3 loads + 1 store",
                           "specialised, roles swapped: waves 0-3 memory, waves 4-7 MFMA", "roles swapped + s_setprio 3 on the memory waves",
                           "waves 0-3 MFMA, waves 4-7 memory with s_setprio 3"};
    for (int mode = 0; mode < 7; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            default: hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, buf, iters, out); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-62s %8.1f ns per iteration\n", names[mode], best * 1e6 / iters);
    }
    return 0;
}
