// Which compute units does a stream created with hipExtStreamCreateWithCUMask run on?  (Round 5: the weight-gradient side stream of the backward is
// confined to a few CUs so that its blocks never sit on a CU a one-block-per-CU persistent kernel needs.)  For each mask pattern: a kernel of many
// long-lived blocks on the masked stream; every block records its XCC_ID and HW_ID; the host prints how many distinct (xcc, se, sh, cu) slots
// were used per XCC.  Patterns: the first k bits; every (ncu / k)-th bit; k / 8 bits at the start of every 32-bit word.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/cu_mask_probe tools/micro/cu_mask_probe.hip && ./tools/micro/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);      // stay resident: the dispatcher has to spread the grid over every CU it may use
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
    const int nb = 2048;
    unsigned* d;
    hipMalloc(&d, nb * 8);
    hipMemsetAsync(d, 0xff, nb * 8, s);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, s, d, 200);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per;
    for (int b = 0; b < nb; ++b) {
        const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int bits = 0;
    for (uint32_t w : mask) bits += __builtin_popcount(w);
    printf("%-28s bits %3d ->", name, bits);
    int total = 0;
    for (auto& kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  (distinct CU slots %d)\n", total);
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int words = (ncu + 31) / 32;
    printf("compute units: %d, mask words: %d\n", ncu, words);
    std::vector<uint32_t> all(words, 0xffffffffu);
    run("all bits", all);
    for (int k : {8, 16, 32}) {
        std::vector<uint32_t> a(words, 0u), b(words, 0u), c(words, 0u);
        for (int i = 0; i < k; ++i) a[i >> 5] |= 1u << (i & 31);
        for (int i = 0; i < k; ++i) { const int bit = i * (ncu / k); b[bit >> 5] |= 1u << (bit & 31); }
        for (int w = 0; w < words; ++w) for (int i = 0; i < k / words; ++i) c[w] |= 1u << i;
        char nm[64];
        snprintf(nm, sizeof nm, "first %d bits", k); run(nm, a);
        snprintf(nm, sizeof nm, "every %d-th bit (%d)", ncu / k, k); run(nm, b);
        snprintf(nm, sizeof nm, "%d low bits per word (%d)", k / words, k); run(nm, c);
    }
    return 0;
}
