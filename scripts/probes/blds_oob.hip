// Probe: what does an out-of-range lane of `buffer_load_dwordx4 ... lds` leave in the LDS on gfx950 -- zeros (usable as
// padding) or the old bytes?  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/blds_oob scripts/probes/blds_oob.hip && /tmp/blds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned nrec_bytes, unsigned soff, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* lds = reinterpret_cast<unsigned*>(smem);
  for (int i = threadIdx.x; i < 64 * 4; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  const unsigned long long b = reinterpret_cast<unsigned long long>(src);
  u32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((unsigned)b);
  rs.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xFFFFu);
  rs.z = nrec_bytes;
  rs.w = 0x00020000u;
  // lanes 0..31 in range, lanes 32..47 beyond num_records, lanes 48..63 offset 0xFFFFFFFF
  unsigned voff = threadIdx.x < 32 ? threadIdx.x * 16 : (threadIdx.x < 48 ? nrec_bytes + threadIdx.x * 16 : 0xFFFFFFFFu);
  const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(ldsb) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4; i += 64) out[i] = lds[i];
}
int main() {
  const int N = 4096;
  std::vector<unsigned> h(N);
  for (int i = 0; i < N; ++i) h[i] = 0x1000u + i;
  unsigned *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  for (unsigned soff : {0u, 1024u}) {
    k<<<1, 64, 64 * 16>>>(d, 2048, soff, o);
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("soffset %u (num_records 2048 B):\n", soff);
    for (int lane : {0, 1, 31, 32, 47, 48, 63})
      printf("  lane %2d: %08x %08x %08x %08x\n", lane, r[lane * 4], r[lane * 4 + 1], r[lane * 4 + 2], r[lane * 4 + 3]);
  }
  return 0;
}
