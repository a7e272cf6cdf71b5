// Which workgroups share a CU?  512 workgroups of 256 threads with 76 KB of LDS each (2 per CU on the
// 256 CUs of an MI355X, like the run kernel), all resident at once; each reports XCC / SE / CU ids.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void k(unsigned *out, int spin) {
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double a = lds[(threadIdx.x + 1) & 255];
    for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;     // stay resident for a while
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    if (a == 12345.0) out[0] = 0;
}
int main() {
    const int nb = 512;
    unsigned *d; (void)hipMalloc(&d, nb * 8);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
    k<<<nb, 256, 76 * 1024>>>(d, 2000000);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 2); (void)hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < nb; ++b) {
        unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("%zu distinct CUs hold the %d workgroups\n", cu.size(), nb);
    int shown = 0;
    for (auto &e : cu) { if (shown++ < 24) { printf("xcc %u se %u sh %u cu %2u:", e.first >> 16, (e.first >> 8) & 0xff, (e.first >> 4) & 0xf, e.first & 0xf); for (int b : e.second) printf(" %d", b); printf("\n"); } }
    std::map<int, int> diff;
    for (auto &e : cu) if (e.second.size() == 2) diff[e.second[1] - e.second[0]]++;
    for (auto &e : diff) printf("pairs with block-index difference %d: %d\n", e.first, e.second);
    return 0;
}
