// lds_dma_probe.hip -- can gfx950 load 16 bytes per lane straight from a buffer into LDS (no VGPR staging), where do the
// bytes land, what do out-of-range lanes write, and is a barrier after it enough for other waves to read the data?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)bytes, 0x00027000);
}

__global__ __launch_bounds__(256) void k(const u32x4 *src, unsigned valid_bytes, u32x4 *out) {
    __shared__ u32x4 smem[512];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (int i = tid; i < 512; i += 256)
        smem[i] = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(src, valid_bytes);
    // wave w loads 64 x 16 B: global unit (w*64 + (lane ^ 5)) -> LDS slot w*64 + lane (hardware places lane i at base + 16 i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(smem + wave * 64), 16,
                                         (wave * 64 + (lane ^ 5u)) * 16, 0, 0, 0);
    // second half of LDS with an SGPR offset and an instruction offset
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(smem + 256 + wave * 64), 16,
                                         lane * 16, wave * 1024, 4096, 0);
    __syncthreads();
    // every thread reads a slot written by ANOTHER wave
    const unsigned j = (tid + 64) & 255u;
    out[tid] = smem[j];
    out[256 + tid] = smem[256 + j];
}

int main() {
    const int n = 1024;
    static u32x4 h[n], o[512];
    for (int i = 0; i < n; ++i) h[i] = u32x4{(unsigned)i, (unsigned)i * 3u, 0x1000u + i, 0x2000u + i};
    u32x4 *d, *dout;
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const unsigned valid = 200 * 16 + 4096; // first call: units >= 456 out of range; tests units < 256 -> all in range except none
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, 230u * 16u, dout);   // units >= 230 are out of range
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0, zero1 = 0, zero2 = 0;
    for (int t = 0; t < 256; ++t) {
        const int j = (t + 64) & 255, w = j / 64, l = j % 64;
        const int u = w * 64 + (l ^ 5);                 // expected source unit of slot j
        if (u < 230) { if (o[t][0] != (unsigned)u || o[t][2] != 0x1000u + u) ++bad1; }
        else { if (o[t][0] == 0 && o[t][1] == 0 && o[t][2] == 0 && o[t][3] == 0) ++zero1; else ++bad1; }
        const int u2 = 256 + w * 64 + l;                // 4096 B imm + w*1024 soffset + lane*16 -> unit 256 + ...
        if (u2 < 230) { if (o[256 + t][0] != (unsigned)u2) ++bad2; }
        else { if (o[256 + t][0] == 0 && o[256 + t][3] == 0) ++zero2; else ++bad2; }
    }
    printf("call 1 (voffset swizzle): wrong %d, out-of-range lanes that wrote zeros %d (of %d)\n", bad1, zero1, 256 - 230);
    printf("call 2 (soffset + imm, all out of range): wrong %d, zeros %d (of 256)\n", bad2, zero2);
    printf("sample slot 0: %08x %08x %08x %08x\n", o[192][0], o[192][1], o[192][2], o[192][3]);
    (void)valid;
    return 0;
}
