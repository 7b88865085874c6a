// ubench_cuid.hip -- which physical CU does a block run on?  512 blocks of 256 threads with 73 KB of LDS each (two per CU, the residency of
// conv2d_wide_kernel) stay resident together for ~100 us and record physical_cu_slot() (conv2d_wide_f16.hip: XCC id x the SE / SH / CU fields of
// HW_ID).  Expected: 256 distinct slots, exactly two blocks on each -- what the K-loop token of conv2d_wide_kernel relies on for its pairing
// (an aliased slot would only serialise blocks of different CUs, never deadlock).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_cuid.hip -o /tmp/ubench_cuid && /tmp/ubench_cuid
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <vector>

__global__ void k(unsigned* out, unsigned* raw) {
    extern __shared__ float smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (8 << 6) | (7 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    if (threadIdx.x == 0) {
        out[blockIdx.x] = ((xcc & 7u) << 8) | (hw & 255u);
        raw[blockIdx.x * 2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        raw[blockIdx.x * 2 + 1] = xcc;
    }
    smem[threadIdx.x] = 1.0f;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 10000ull) __builtin_amdgcn_s_sleep(32); // 100 MHz clock: 100 us
}

int main() {
    const int blocks = 512;
    unsigned *d, *r;
    hipMalloc(&d, blocks * 4);
    hipMalloc(&r, blocks * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 73 * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 73 * 1024, 0, d, r);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks), hr(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), r, blocks * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> cnt;
    for (unsigned v : h) cnt[v]++;
    std::map<int, int> hist;
    for (auto& kv : cnt) hist[kv.second]++;
    printf("%d blocks -> %zu distinct slots;", blocks, cnt.size());
    for (auto& kv : hist) printf(" %d slot(s) hold %d block(s);", kv.second, kv.first);
    printf("\nfirst blocks: ");
    for (int i = 0; i < 12; ++i) printf("[b%d hw=%08x xcc=%u slot=%u] ", i, hr[2 * i], hr[2 * i + 1], h[i]);
    printf("\n");
    return 0;
}
