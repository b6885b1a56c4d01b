// Same-address 64-bit atomic throughput on MI355X: N workgroups x 256 threads, thread t adds 1 to acc[(wg % R) * 256 + t].
// Answers: what do the fixed-point BatchNorm accumulators cost per producer launch, how much do R replicas help, and is a
// workgroup-scope atomic (a) correct across XCDs and (b) any faster than an agent-scope one.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int SCOPE>
__global__ void k(unsigned long long* acc, int R) {
  unsigned long long* p = acc + (size_t)(blockIdx.x % R) * 256 + threadIdx.x;
  if (SCOPE == 0) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void xcc(unsigned* out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
  unsigned long long* acc; hipMalloc(&acc, 64 * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int scope = 0; scope < 2; ++scope)
    for (int N : {768, 3136, 12544})
      for (int R : {1, 8, 32}) {
        hipMemset(acc, 0, 64 * 256 * 8);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) { if (scope) hipLaunchKernelGGL(k<1>, dim3(N), dim3(256), 0, 0, acc, R); else hipLaunchKernelGGL(k<0>, dim3(N), dim3(256), 0, 0, acc, R); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(64 * 256);
        hipMemcpy(h.data(), acc, 64 * 256 * 8, hipMemcpyDeviceToHost);
        unsigned long long tot = 0; for (auto v : h) tot += v;
        printf("scope %s  N %5d  R %2d : %7.1f us/launch   total %llu (expected %llu) %s\n", scope ? "workgroup" : "agent    ", N, R, ms * 100.f, tot,
               (unsigned long long)N * 256 * 10, tot == (unsigned long long)N * 256 * 10 ? "ok" : "LOST UPDATES");
      }
  unsigned* xo; hipMalloc(&xo, 64 * 4); hipLaunchKernelGGL(xcc, dim3(32), dim3(64), 0, 0, xo);
  unsigned hx[32]; hipMemcpy(hx, xo, 32 * 4, hipMemcpyDeviceToHost);
  printf("XCC_ID of workgroups 0..31:"); for (int i = 0; i < 32; ++i) printf(" %u", hx[i] & 0xf); printf("\n");
  return 0;
}
