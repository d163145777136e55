// Where do the workgroups of a launch go under a CU mask?  (hipExtStreamCreateWithCUMask on an 8-XCD MI355X.)
//   hipcc --offload-arch=gfx950 -O2 scripts/lab/cumask_map.hip -o /tmp/cumask_map && timeout 120 /tmp/cumask_map
// Every workgroup records the XCD it runs on (HW_REG_XCC_ID) and its shader engine / CU (HW_REG_HW_ID).  For each mask the
// program prints, per XCD, how many workgroups ran there and on how many distinct CUs.  Questions answered:
//   (1) which bit of the mask is which XCD: interleaved (bit i -> XCD i % 8) or contiguous (bit i -> XCD i / 32)?
//   (2) what happens to a launch when an XCD has NO enabled CU: are its workgroups dealt over the other XCDs, or does the
//       launch wait for ever?  (The risky cases run last, each announced before it starts; run under `timeout`.)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <set>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the workgroup alive for a while so that the launch spreads over every CU it may use
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char* name, const std::vector<uint32_t>& mask, int n_wgs, unsigned* dev, bool order = false) {
    printf("mask %-34s", name); fflush(stdout);
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    CK(hipMemsetAsync(dev, 0xff, sizeof(unsigned) * 2 * n_wgs, s));
    hipLaunchKernelGGL(where_kernel, dim3(n_wgs), dim3(256), 0, s, dev, 2000);          // 20 us per workgroup
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * n_wgs);
    CK(hipMemcpy(h.data(), dev, sizeof(unsigned) * 2 * n_wgs, hipMemcpyDeviceToHost));
    int per_xcd[16] = {0};
    std::set<unsigned> cus[16];
    for (int i = 0; i < n_wgs; ++i) {
        const unsigned x = h[2 * i] & 0xf, hw = h[2 * i + 1];
        per_xcd[x]++;
        cus[x].insert((hw >> 8) & 0xff);      // cu_id [11:8], sh_id [12], se_id [15:13]
    }
    printf(" wgs/xcd:");
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcd[x]);
    printf("   distinct (se,sh,cu)/xcd:");
    for (int x = 0; x < 8; ++x) printf(" %2zu", cus[x].size());
    printf("\n");
    if (order) {
        printf("   xcd of workgroups 0..31:");
        for (int i = 0; i < 32; ++i) printf(" %u", h[2 * i] & 0xf);
        printf("\n");
    }
    fflush(stdout);
    CK(hipStreamDestroy(s));
}

int main(int argc, char** argv) {
    const bool risky = argc > 1 && !strcmp(argv[1], "risky");
    int n_cu = 0;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs: %d\n", n_cu);
    const int words = (n_cu + 31) / 32, n_wgs = 2048;
    unsigned* dev;
    CK(hipMalloc(&dev, sizeof(unsigned) * 2 * n_wgs));
    auto mk = [&](auto pred) { std::vector<uint32_t> m(words, 0); for (int i = 0; i < n_cu; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
    run("all", mk([](int) { return true; }), n_wgs, dev, true);
    // never an empty XCD under either hypothesis: half of every XCD
    run("even bits of each group of 16", mk([](int i) { return (i % 16) < 8 && true; }), n_wgs, dev);
    run("bits with (i/8)%2 == 0", mk([](int i) { return ((i / 8) % 2) == 0; }), n_wgs, dev);
    run("first 96 bits", mk([](int i) { return i < 96; }), n_wgs, dev);
    // one CU left on seven XCDs under the interleaved hypothesis (bit i -> XCD i % 8): bits 0..7 plus all bits of XCD 0
    run("i%8==0 or i<8 (interleaved: XCD0 + 1 CU each)", mk([](int i) { return i % 8 == 0 || i < 8; }), n_wgs, dev);
    // the same under the contiguous hypothesis: bits 0..31 plus one bit per group of 32
    run("i<32 or i%32==0 (contiguous: XCD0 + 1 CU each)", mk([](int i) { return i < 32 || i % 32 == 0; }), n_wgs, dev);
    if (risky) {
        printf("RISKY: masks that may leave an XCD without a CU (a launch that never ends is the answer 'static partition')\n"); fflush(stdout);
        run("i%8==0 (interleaved: only XCD0)", mk([](int i) { return i % 8 == 0; }), n_wgs, dev, true);
        run("i%8<3 (interleaved: XCD0-2)", mk([](int i) { return i % 8 < 3; }), n_wgs, dev, true);
        run("i<32 (contiguous: only XCD0)", mk([](int i) { return i < 32; }), n_wgs, dev, true);
    }
    printf("done\n");
    return 0;
}
