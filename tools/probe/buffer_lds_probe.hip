// probe: buffer_load_dwordx4 ... lds (raw buffer, offen): do out-of-range lanes write ZEROS into LDS, also with a scalar offset
// (range check on voffset + soffset, no 32-bit wrap-around of the 0xFFFFFFF0 marker)?  (hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* a, int nbytes, unsigned* out, int soff) {
  __shared__ __attribute__((aligned(16))) unsigned smem[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) smem[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a, (short)0, nbytes, 0x00020000);
  // lanes 0..31: in range (lane * 16 bytes), lanes 32..47: exactly past the end, lanes 48..63: 0xFFFFFFF0
  // (with a scalar offset the range check is on voffset + soffset: lanes 32..47 land exactly past the end, lanes 48..63 must not wrap
  //  around 2^32 into the buffer)
  unsigned off = threadIdx.x < 32 ? threadIdx.x * 16 : (threadIdx.x < 48 ? (nbytes - soff) + (threadIdx.x - 32) * 16 : 0xFFFFFFF0u);
  // soff: the instruction's scalar offset (the kernels put the plane offset there): added to the address AFTER the range check
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, off, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 0x1000 + i;
  unsigned *a, *o;
  hipMalloc(&a, 4096 * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(a, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  int bad = 0;
  for (int soff = 0; soff <= 1024; soff += 1024) {   // num_records = 512 + soff bytes: lanes 0..31 in range, reading at soff + lane * 16
    k<<<1, 64>>>(a, 512 + soff, o, soff);
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        unsigned want = l < 32 ? 0x1000 + soff / 4 + l * 4 + j : 0u;
        if (r[l * 4 + j] != want) { if (bad < 8) printf("soffset %d lane %d word %d: got %08x want %08x\n", soff, l, j, r[l * 4 + j], want); ++bad; }
      }
  }
  printf(bad ? "PROBE FAIL (%d words)\n" : "PROBE OK: in-range lanes loaded, out-of-range lanes wrote zeros (%d bad)\n", bad);
  return bad != 0;
}
// build + run:  hipcc --offload-arch=gfx950 -O2 -o tools/probe/buffer_lds_probe tools/probe/buffer_lds_probe.hip && gpurun -- ./tools/probe/buffer_lds_probe
