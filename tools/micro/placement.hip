// Where do the two waves of 128-thread workgroups land?  1024 workgroups x 2 waves, 256 VGPRs and 37 KB of LDS each (the pair
// form's footprint): every wave records (XCC, SE, CU, SIMD) from its hardware id registers and its start / end clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
template <int SCRATCH>
__global__ __launch_bounds__(128) __attribute__((amdgpu_num_vgpr(256))) void probe(unsigned* out, int spin, int lds_words) {
    extern __shared__ unsigned lds[];
    volatile double priv[SCRATCH > 0 ? SCRATCH : 1];     // dynamically indexed -> lives in scratch memory
    if (SCRATCH > 0) { for (int k = 0; k < SCRATCH; ++k) priv[(k * 7 + spin) % SCRATCH] = k; }
    const int wave = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_readcyclecounter();
    double acc = threadIdx.x;
    for (int i = 0; i < spin / 100; ++i) __builtin_amdgcn_s_sleep(100);    // idles without using issue slots: co-resident waves do not slow each other
    if (SCRATCH > 0) acc += priv[spin % SCRATCH];
    if (lds_words > 0) lds[threadIdx.x % lds_words] = (unsigned)acc;
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        unsigned* o = out + (blockIdx.x * 2 + wave) * 6;
        o[0] = hw; o[1] = xcc; o[2] = (unsigned)(t0 >> 8); o[3] = (unsigned)(t1 >> 8); o[4] = blockIdx.x; o[5] = wave;
    }
}
int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 1024, lds = argc > 2 ? atoi(argv[2]) : 37392, scratch = argc > 3 ? atoi(argv[3]) : 0;
    unsigned* d; hipMalloc(&d, 4096 * 2 * 6 * 4);
    if (scratch) hipLaunchKernelGGL(probe<70>, dim3(wgs), dim3(128), lds, 0, d, 200000, lds / 4);
    else hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(128), lds, 0, d, 200000, lds / 4);
    hipDeviceSynchronize();
    {   // duration against the number of workgroups: a jump = a second round
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w : {256, 512, 768, 1024, 1280, 2048}) {
            hipEventRecord(e0, 0);
            if (scratch) hipLaunchKernelGGL(probe<70>, dim3(w), dim3(128), lds, 0, d, 200000, lds / 4);
            else hipLaunchKernelGGL(probe<0>, dim3(w), dim3(128), lds, 0, d, 200000, lds / 4);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %5d workgroups: %.3f ms\n", w, ms);
        }
    }
    std::vector<unsigned> h(wgs * 2 * 6);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<unsigned long long, std::vector<int>> simd;   // key (xcc, se, sh, cu, simd) -> list of (block*2+wave)
    unsigned tmin = ~0u;
    for (int i = 0; i < wgs * 2; ++i) tmin = std::min(tmin, h[i * 6 + 2]);
    int late = 0;
    for (int i = 0; i < wgs * 2; ++i) {
        unsigned hw = h[i * 6], xcc = h[i * 6 + 1] & 0xf;
        unsigned long long key = ((unsigned long long)xcc << 32) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 12) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3);
        simd[key].push_back(i);
        if ((h[i * 6 + 2] - tmin) > (h[0 * 6 + 3] - h[0 * 6 + 2]) / 2) ++late;
    }
    std::map<int, int> hist; int same_role = 0, mixed = 0, same_wg = 0;
    for (auto& kv : simd) {
        hist[(int)kv.second.size()]++;
        if (kv.second.size() == 2) {
            int a = kv.second[0], b = kv.second[1];
            if ((a & 1) == (b & 1)) ++same_role; else ++mixed;
            if ((a >> 1) == (b >> 1)) ++same_wg;
        }
    }
    printf("workgroups %d lds %d scratch %d: SIMDs used %zu, waves that started late (second round) %d\n", wgs, lds, scratch, simd.size(), late);
    for (auto& kv : hist) printf("  SIMDs hosting %d waves: %d\n", kv.first, kv.second);
    printf("  of the SIMDs hosting 2: same wave index (0+0 or 1+1) %d, one of each %d, both waves of ONE workgroup %d\n", same_role, mixed, same_wg);
    // first CU: which blocks
    int shown = 0;
    for (auto& kv : simd) { if (shown++ >= 8) break; printf("  simd key %llx:", kv.first); for (int v : kv.second) printf(" wg%d.w%d", v >> 1, v & 1); printf("\n"); }
    return 0;
}
