// LDS-DMA semantics probe for gfx950: buffer_load_dwordx4 ... offen lds issued from inline asm (hidden from
// hipcc's waitcnt bookkeeping), the destination rule (M0 + lane * 16), what an out-of-range buffer offset leaves in LDS, what
// exec-masked lanes leave, and whether an instruction offset moves the LDS side.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/glds_probe.hip -o tools/ub_glds && tools/ub_glds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, i32x4 rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma16_off64(unsigned lds_addr, unsigned voff, i32x4 rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen offset:64 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__global__ __launch_bounds__(256) void k_probe(const float* in, int n_in, float* out, int mode) {
    __shared__ __attribute__((aligned(16))) float smem[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8192; i += 256) smem[i] = -7.0f;                     // sentinel
    __syncthreads();
    const unsigned long long base = (unsigned long long)in;
    const i32x4 rsrc = {int(unsigned(base)), int(unsigned(base >> 32) & 0xFFFF), n_in * 4, 0x00020000};
    const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) float*)smem));
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);   // each wave owns 4 KB: four 1 KB pieces
    // piece 0: lane-reversed source (lane l reads float4 #(63 - l) of its wave's 256-float window)
    dma16(dst, unsigned((wave * 256 + (63 - lane) * 4) * 4), rsrc, 0);
    // piece 1: odd lanes out of range (offset bit 31), soffset = 1024 bytes on the even ones
    dma16(dst + 1024, (lane & 1) ? 0x80000000u : unsigned(lane * 16), rsrc, 1024);
    // piece 2: only lanes < 16 active
    if (lane < 16) dma16(dst + 2048, unsigned(lane * 16), rsrc, 0);
    // piece 3: instruction offset 64: does it move the global side only, or the LDS side too?
    dma16_off64(dst + 3072, unsigned(lane * 16), rsrc, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 8192; i += 256) out[i] = smem[i];
    if (mode == 1 && tid == 0) out[8192] = float(lds0);
}

// DMA issue-rate / pattern cost: every wave streams `pieces` 1 KB pieces with one of three source patterns
//   0: 8 adjacent lanes per 128-B line (a 32-channel halo)   1: 4 adjacent lanes per 64-B half line (a 16-channel piece)
//   2: a pixel's four quads 16 lanes apart (the bank-perfect image)
__global__ __launch_bounds__(256, 3) void k_rate(const float* in, unsigned nbytes, int pieces, int pattern, float* out) {
    __shared__ __attribute__((aligned(16))) float smem[12288];                 // 48 KB: three blocks per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long base = (unsigned long long)in;
    const i32x4 rsrc = {int(unsigned(base)), int(unsigned(base >> 32) & 0xFFFF), int(nbytes), 0x00020000};
    const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) float*)smem));
    const unsigned pix_stride = 512;                                           // 128 channels
    unsigned voff;
    if (pattern == 0) voff = (lane >> 3) * pix_stride + (lane & 7) * 16;
    else if (pattern == 1) voff = (lane >> 2) * 4 * pix_stride + (lane & 3) * 16;
    else voff = (lane & 15) * 4 * pix_stride + (lane >> 4) * 16;
    voff += (blockIdx.x * 4 + wave) * 64 * pix_stride;
    const unsigned span = pattern == 0 ? 8 * pix_stride : 64 * pix_stride;
    for (int p = 0; p < pieces; ++p) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (wave * 12 + (p % 12)) * 1024);
        dma16(dst, (voff + p * span) % (nbytes - 4096), rsrc, 0);
        if ((p & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (out) out[blockIdx.x * 256 + tid] = smem[tid];
}

int main() {
    const int n_in = 4096;
    std::vector<float> h(n_in); for (int i = 0; i < n_in; ++i) h[i] = float(i);
    float *in, *out; CK(hipMalloc(&in, n_in * 4)); CK(hipMalloc(&out, 8200 * 4));
    CK(hipMemcpy(in, h.data(), n_in * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 0, 0, in, n_in, out, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> o(8200); CK(hipMemcpy(o.data(), out, 8200 * 4, hipMemcpyDeviceToHost));
    printf("LDS byte address of the array: %g\n", o[8192]);
    int bad0 = 0, bad1 = 0, zero1 = 0, sent1 = 0, bad2 = 0, sent2 = 0;
    for (int w = 0; w < 4; ++w) {
        const float* p = o.data() + w * 1024;
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                if (p[l * 4 + e] != float(w * 256 + (63 - l) * 4 + e)) ++bad0;
                const float v1 = p[256 + l * 4 + e];
                if (l & 1) { if (v1 == 0.0f) ++zero1; else if (v1 == -7.0f) ++sent1; else ++bad1; }
                else if (v1 != float(256 + l * 4 + e)) ++bad1;
                const float v2 = p[512 + l * 4 + e];
                if (l < 16) { if (v2 != float(l * 4 + e)) ++bad2; } else { if (v2 == -7.0f) ++sent2; else ++bad2; }
            }
    }
    printf("piece 0 (lane-linear destination, per-lane source): %s (%d wrong)\n", bad0 ? "FAIL" : "ok", bad0);
    printf("piece 1 (out-of-range lanes): %d wrong in-range values; out-of-range elements: %d zeros, %d untouched sentinels  -> %s\n", bad1, zero1, sent1,
           zero1 == 4 * 32 * 4 ? "OUT-OF-RANGE LANES WRITE ZEROS" : (sent1 == 4 * 32 * 4 ? "out-of-range lanes write NOTHING" : "mixed"));
    printf("piece 2 (exec-masked lanes): %d wrong, %d untouched sentinels of %d  -> %s\n", bad2, sent2, 4 * 48 * 4, sent2 == 4 * 48 * 4 && !bad2 ? "masked lanes write nothing" : "UNEXPECTED");
    { const float* p = o.data() + 768;
      printf("piece 3 (offset:64): LDS[dst + 0..3] = %g %g %g %g ; LDS[dst + 64 bytes..] = %g %g  (global side +16 floats expected: 16 17 18 19 at the slot the LDS side uses)\n",
             p[0], p[1], p[2], p[3], p[16], p[17]); }

    // rates
    const size_t nbytes = size_t(1) << 28;
    float* big; CK(hipMalloc(&big, nbytes)); CK(hipMemset(big, 0, nbytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pattern = 0; pattern < 3; ++pattern) {
        const int pieces = 96, blocks = 768;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, big, unsigned(nbytes), pieces, pattern, (float*)nullptr);
        CK(hipEventRecord(e0, 0));
        for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, big, unsigned(nbytes), pieces, pattern, (float*)nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = double(blocks) * 4 * pieces * 1024;
        printf("pattern %d: %.1f us per launch, %.2f TB/s into LDS (768 blocks x 4 waves x %d pieces)\n", pattern, ms * 100, bytes / (ms * 1e-4) / 1e12, pieces);
    }
    return 0;
}
