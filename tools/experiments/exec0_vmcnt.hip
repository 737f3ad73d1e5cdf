// Does a vector-memory instruction issued with EXEC = 0 take part in vmcnt on gfx950?
//   load A (real, slow: a cold line), then load B with EXEC = 0, then s_waitcnt vmcnt(1), then read A's destination register.
// If B counts (in-order return), vmcnt(1) cannot be satisfied before A is back: A's register holds the loaded value.
// If B is dropped without counting, vmcnt(1) is satisfied at once (one operation outstanding): the register still holds the sentinel.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/exec0_vmcnt.hip -o /tmp/exec0_vmcnt && /tmp/exec0_vmcnt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const unsigned *src, unsigned *out, unsigned long long zero_mask)
{
    unsigned a = 0xdeadbeefu, b = 0x12345678u;
    const unsigned voff = (threadIdx.x + blockIdx.x * 64u) * 4096u;      // one cold line per lane
    unsigned long long save;
    asm volatile("global_load_dword %0, %3, %4\n\t"
                 "s_mov_b64 %2, exec\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "global_load_dword %1, %3, %4 offset:64\n\t"
                 "s_mov_b64 exec, %2\n\t"
                 "s_waitcnt vmcnt(1)\n\t"
                 "v_mov_b32 %1, %0\n\t"            // A's register right behind the wait
                 "s_waitcnt vmcnt(0)"
                 : "+v"(a), "+v"(b), "=&s"(save)
                 : "v"(voff), "s"(src), "s"(zero_mask)
                 : "memory");
    out[threadIdx.x + blockIdx.x * 64] = b;
}

int main()
{
    const int blocks = 256, n = blocks * 64;
    unsigned *src, *out;
    hipMalloc(&src, (size_t)n * 4096 + 4096);
    hipMalloc(&out, n * 4);
    hipMemset(src, 0x5a, (size_t)n * 4096 + 4096);
    hipDeviceSynchronize();
    int stale = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, src, out, 0ull);
        std::vector<unsigned> h(n);
        hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
        for (unsigned v : h) stale += (v == 0xdeadbeefu);
    }
    printf("lanes that read the sentinel behind vmcnt(1): %d of %d -> an EXEC = 0 load %s in vmcnt\n", stale, 5 * n, stale ? "does NOT count" : "counts");
    return 0;
}
