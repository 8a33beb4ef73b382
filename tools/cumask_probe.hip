// Which XCD / CU does bit i of a hipExtStreamCreateWithCUMask mask select on gfx950?
//   hipcc -O2 --offload-arch=gfx950 tools/cumask_probe.hip -o tools/cumask_probe && tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void where(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
    for (int i = 0; i < 20000; ++i) __builtin_amdgcn_s_sleep(10);  // stay resident: spread over every allowed CU
}
static void run(const char* what, std::vector<unsigned> mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", what); return; }
    const int B = 1024; unsigned* d; hipMalloc(&d, B * 8); hipMemset(d, 0xff, B * 8);
    where<<<B, 64, 0, s>>>(d); hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * B); hipMemcpy(h.data(), d, B * 8, hipMemcpyDeviceToHost);
    int per[16] = {0}; bool seen[16][8][2][16]; memset(seen, 0, sizeof(seen));
    for (int b = 0; b < B; ++b) { unsigned x = h[2*b], hw = h[2*b+1]; unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        if (x < 16 && !seen[x][se][sh][cu]) { seen[x][se][sh][cu] = true; per[x]++; } }
    printf("%-34s distinct CUs used per XCC:", what); for (int x = 0; x < 8; ++x) printf(" %2d", per[x]); printf("\n");
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    std::vector<unsigned> m(8, 0);
    m.assign(8, 0); m[0] = 0xffffffffu; run("bits 0..31", m);
    m.assign(8, 0); for (int w = 0; w < 8; ++w) m[w] = 0xfu; run("bits i%32 < 4", m);
    m.assign(8, 0xffffffffu); m[0] = 0; run("all but bits 0..31", m);
    m.assign(8, 0xffffffffu); run("all 256 bits", m);
    m.assign(8, 0); m[0] = 0xffu; run("bits 0..7", m);
    m.assign(8, 0); m[0] = 0x1u; run("bit 0", m);
    m.assign(8, 0); m[0] = 0x100u; run("bit 8", m);
    return 0;
}
